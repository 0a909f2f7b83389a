// controller_base.cpp -- CPU MPPI, serial by default (see include/controller_base.hpp).
// Stage numbers refer to the reference's intended pipeline, src/controller_base.cpp:61-80, and
// to the GPU path it mirrors, src/point_mass.cu:129-203.
#include "../../include/controller_base.hpp"

#include "../../include/mppi_gpu_amd.h"
#include "../../include/point_mass_gpu.hpp"

#include <rocrand/rocrand_kernel.h>

#include <cmath>
#include <cstdio>
#include <functional>
#include <thread>

namespace {

// fn(begin, end) over [0, n) in at most `threads` contiguous pieces of at least `grain` items (the
// caller's thread takes the first): a worker thread costs tens of microseconds to start, so a piece
// must be worth it -- 256 workers on K = 1e4 samples were 4x one thread, 39 are 20x
void parallel_ranges(int n, int threads, int grain, const std::function<void(int, int)>& fn)
{
    if (threads > n / grain) threads = n / grain;
    if (threads <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> pool;
    const int per = (n + threads - 1) / threads;
    for (int t = 1; t < threads; ++t) {
        const int b = t * per, e = b + per < n ? b + per : n;
        if (b < e) pool.emplace_back(fn, b, e);
    }
    fn(0, per < n ? per : n);
    for (std::thread& th : pool) th.join();
}

// Box-Muller of the engine (kernels: box_muller_hw), evaluated with libm on the host.
void box_muller_host(unsigned int x, unsigned int y, float* z0, float* z1)
{
    const float kInv = 2.3283064e-10f;
    const float u = kInv + (float)x * kInv;
    const float th = kInv + (float)y * kInv;
    const float s = sqrtf(-1.3862943611198906f * log2f(u));
    const double ang = 6.283185307179586476925 * (double)th;
    *z0 = (float)sin(ang) * s;
    *z1 = (float)cos(ang) * s;
}

}  // namespace

ControllerBase::ControllerBase(const int k, const int tau, const float dt, const int sDim,
                               const int aDim)
    : mK(k), mTau(tau), mSDim(sDim), mADim(aDim), mThreads(1), mDt(dt), mLambda(1.0f), mSeed(0), mSolve(0),
      mInjected(false), mBeta(0.0f), mNabla(0.0f)
{
    mU.assign((size_t)tau * aDim, 0.0f);
    mE.assign((size_t)k * tau * aDim, 0.0f);
    mX.assign((size_t)(tau + 1) * sDim, 0.0f);
    mCost.assign(k, 0.0f);
    mWeights.assign(k, 0.0f);
    mGoal.assign(sDim, 0.0f);
    mW.assign(sDim, 1.0f);
    mSigma.assign(aDim, 0.025f);      // reference src/point_mass_gpu.cu:86
    mInvS.assign(aDim, 1.0f);         // reference src/point_mass_gpu.cu:58-61
    // reference src/point_mass.cu:46-51
    mXGain[0] = 1.0f; mXGain[1] = dt; mXGain[2] = 0.0f; mXGain[3] = 1.0f;
    const float dd = dt * dt;
    mUGain[0] = (float)((double)dd / 2.0);
    mUGain[1] = dt;
}

ControllerBase::~ControllerBase() {}

bool ControllerBase::setActions(const float* actions)
{
    if (!actions) return false;
    mU.assign(actions, actions + (size_t)mTau * mADim);
    return true;
}

void ControllerBase::setCost(const float* goal, const float* w)
{
    if (goal) mGoal.assign(goal, goal + mSDim);
    if (w) mW.assign(w, w + mSDim);
}

void ControllerBase::setParams(float lambda, const float* sigma, const float* inv_s)
{
    mLambda = lambda;
    if (sigma) mSigma.assign(sigma, sigma + mADim);
    if (inv_s) mInvS.assign(inv_s, inv_s + mADim);
}

void ControllerBase::setSeed(unsigned long long seed)
{
    mSeed = seed;
    mSolve = 0;
}

void ControllerBase::setThreads(int n) { mThreads = n < 1 ? 1 : (n > 256 ? 256 : n); }

void ControllerBase::setNoise(const float* E)
{
    mInjected = E != nullptr;
    if (E) mE.assign(E, E + (size_t)mK * mTau * mADim);
}

// stage 1 -- noise: Philox block b of sample k in solve j holds normals 4b..4b+3 of the flat
// sequence n = t*aDim + a (same stream as the GPU engine, DESIGN.md "Noise")
void ControllerBase::sampleNoise()
{
    const int TA = mTau * mADim;
    const unsigned long long NBT = (unsigned long long)((TA + 3) / 4);
    parallel_ranges(mK, mThreads, 256, [&](int k0, int k1) {
    for (int k = k0; k < k1; ++k) {
        for (unsigned long long b = 0; b < NBT; ++b) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(mSeed, (unsigned long long)k, 4ull * (mSolve * NBT + b), &st);
            const uint4 r = rocrand4(&st);
            float z[4];
            box_muller_host(r.x, r.y, &z[0], &z[1]);
            box_muller_host(r.z, r.w, &z[2], &z[3]);
            for (int i = 0; i < 4; ++i) {
                const int n = (int)b * 4 + i;
                if (n < TA) mE[(size_t)k * TA + n] = mSigma[n % mADim] * z[i];
            }
        }
    }
    });
}

void ControllerBase::next(const float* x, float* act)
{
    const int TA = mTau * mADim;
    if (!mInjected) sampleNoise();

    // stage 2 -- simulate every sample, one after the other (per worker: its own state trace)
    std::vector<float> x0(x, x + mSDim);
    parallel_ranges(mK, mThreads, 256, [&](int k0, int k1) {
    std::vector<float> trace_local;
    float* X = mX.data();
    if (k0 != 0) {
        trace_local.assign(mX.size(), 0.0f);
        X = trace_local.data();
    }
    std::vector<float> x0w(x0);
    for (int k = k0; k < k1; ++k) {
        PointMassModelGpu sim;
        sim.init(X, x0w.data(), mU.data(), &mE[(size_t)k * TA], mTau, mXGain, mSDim, mUGain,
                 mADim, mW.data(), mGoal.data(), mLambda, k);
        // the per-sample type fixes inv_s = 1 like the reference; honour a custom inv_s by
        // evaluating the control term here when it differs
        float c = sim.run(nullptr);
        bool unit = true;
        for (int a = 0; a < mADim; ++a) unit = unit && (mInvS[a] == 1.0f);
        if (!unit) {
            // recompute with the general control cost (same order of operations)
            Cost stage(mW.data(), mSDim, mGoal.data(), mSDim, mLambda, mInvS.data(), mADim);
            c = 0.0f;
            for (int t = 0; t < mTau; ++t)
                c += stage.step_cost(&X[(size_t)(t + 1) * mSDim], &mU[(size_t)t * mADim],
                                     &mE[(size_t)k * TA + (size_t)t * mADim], k, t);
            c += stage.final_cost(&X[(size_t)mTau * mSDim], k);
        }
        mCost[k] = c;
    }
    });

    // stage 3 -- beta = min cost (numerical stability of the exponentials)
    float beta = INFINITY;
    for (int k = 0; k < mK; ++k) beta = mCost[k] < beta ? mCost[k] : beta;

    // stages 4, 5 -- exp and its sum (double accumulator, rounded once)
    double sum = 0.0;
    for (int k = 0; k < mK; ++k) {
        mWeights[k] = expf(-(1 / mLambda) * (mCost[k] - beta));
        sum += (double)mWeights[k];
    }
    const float nabla = (float)sum;

    // stage 6 -- weights, reference src/point_mass.cu:743-754 (double intermediates)
    for (int k = 0; k < mK; ++k)
        mWeights[k] = (float)(1.0 / (double)nabla * (double)mWeights[k]);

    // stage 7 -- weighted mean of the noise, U += sum_k w_k E_k (double accumulators)
    // (workers split the control values, not the samples: every value is summed over k in order)
    std::vector<double> acc(TA, 0.0);
    parallel_ranges(TA, mThreads, 32, [&](int n0, int n1) {
        for (int k = 0; k < mK; ++k)
            for (int n = n0; n < n1; ++n)
                acc[n] += (double)mWeights[k] * (double)mE[(size_t)k * TA + n];
    });
    for (int n = 0; n < TA; ++n) mU[n] = (float)((double)mU[n] + acc[n]);

    // action out, then shift with the last step repeated (reference src/point_mass.cu:195-199)
    if (act)
        for (int a = 0; a < mADim; ++a) act[a] = mU[a];
    for (int n = 0; n + mADim < TA; ++n) mU[n] = mU[n + mADim];

    mBeta = beta;
    mNabla = nabla;
    mSolve += 1;
}

int ControllerBase::logGraph() const
{
    printf("ControllerBase: serial MPPI, K=%d tau=%d dt=%g sDim=%d aDim=%d lambda=%g\n", mK, mTau,
           mDt, mSDim, mADim, mLambda);
    printf("  1 noise      Philox4x32-10 + Box-Muller, sigma per axis\n");
    printf("  2 simulate   double integrator, quadratic stage + terminal cost\n");
    printf("  3 min        beta = min_k cost\n");
    printf("  4 exp        exp(-(cost - beta) / lambda)\n");
    printf("  5 sum        nabla\n");
    printf("  6 div        weights = exp / nabla\n");
    printf("  7 update     U += sum_k w_k E_k, action = U[0], shift\n");
    return 0;
}

// ---- C entry points (include/mppi_gpu_amd.h, "serial CPU controller") ---------------------

extern "C" {

mppi_cpu_controller* mppi_cpu_create(int k, int tau, float dt, int s_dim, int a_dim)
{
    if (k < 1 || tau < 1 || a_dim < 1 || s_dim != 2 * a_dim || !(dt > 0.f)) return nullptr;
    return reinterpret_cast<mppi_cpu_controller*>(new ControllerBase(k, tau, dt, s_dim, a_dim));
}
void mppi_cpu_destroy(mppi_cpu_controller* c) { delete reinterpret_cast<ControllerBase*>(c); }
int mppi_cpu_set_data(mppi_cpu_controller* c, const float* u, const float* goal, const float* w)
{
    if (!c) return MPPI_EINVAL;
    ControllerBase* b = reinterpret_cast<ControllerBase*>(c);
    if (u && !b->setActions(u)) return MPPI_EINVAL;
    b->setCost(goal, w);
    return MPPI_OK;
}
int mppi_cpu_set_params(mppi_cpu_controller* c, float lambda, const float* sigma,
                        const float* inv_s)
{
    if (!c || !(lambda > 0.f)) return MPPI_EINVAL;
    reinterpret_cast<ControllerBase*>(c)->setParams(lambda, sigma, inv_s);
    return MPPI_OK;
}
int mppi_cpu_set_seed(mppi_cpu_controller* c, unsigned long long seed)
{
    if (!c) return MPPI_EINVAL;
    reinterpret_cast<ControllerBase*>(c)->setSeed(seed);
    return MPPI_OK;
}
int mppi_cpu_set_threads(mppi_cpu_controller* c, int n)
{
    if (!c || n < 1) return MPPI_EINVAL;
    reinterpret_cast<ControllerBase*>(c)->setThreads(n);
    return MPPI_OK;
}
int mppi_cpu_set_noise(mppi_cpu_controller* c, const float* noise)
{
    if (!c) return MPPI_EINVAL;
    reinterpret_cast<ControllerBase*>(c)->setNoise(noise);
    return MPPI_OK;
}
int mppi_cpu_next(mppi_cpu_controller* c, const float* x, float* act)
{
    if (!c || !x) return MPPI_EINVAL;
    reinterpret_cast<ControllerBase*>(c)->next(x, act);
    return MPPI_OK;
}
int mppi_cpu_get(mppi_cpu_controller* c, float* u, float* noise, float* cost, float* beta,
                 float* nabla, float* weight)
{
    if (!c) return MPPI_EINVAL;
    ControllerBase* b = reinterpret_cast<ControllerBase*>(c);
    if (u) for (size_t i = 0; i < b->actions().size(); ++i) u[i] = b->actions()[i];
    if (noise) for (size_t i = 0; i < b->noise().size(); ++i) noise[i] = b->noise()[i];
    if (cost) for (size_t i = 0; i < b->costs().size(); ++i) cost[i] = b->costs()[i];
    if (weight) for (size_t i = 0; i < b->weights().size(); ++i) weight[i] = b->weights()[i];
    if (beta) *beta = b->beta();
    if (nabla) *nabla = b->nabla();
    return MPPI_OK;
}

}  // extern "C"
