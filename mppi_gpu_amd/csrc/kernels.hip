// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the MPPI solve.
//
// Replaces the reference's 12 CUDA kernels (src/point_mass.cu:493-926) and the per-sample
// object PointMassModelGpu (src/point_mass_gpu.cu) with TWO launches per solve:
//
//   k_rollout_fused  sample noise (rocRAND Philox4x32-10 + Box-Muller), roll the point mass
//                    out, accumulate the quadratic cost (src/cost.cu:42-64), store E and
//                    cost, and reduce -- per persistent block -- the running minimum, the
//                    exp-sum and the exp-weighted noise sums relative to that minimum.
//   k_combine        beta = min, nabla = sum, dU = sum(w*E) from the per-block partials;
//                    U += dU, action = U[0], shift (src/point_mass.cu:195-199,805-824).
//
// Work decomposition: C lanes cooperate on one trajectory (C = 1..64, a power of two); lane
// (k, c) owns the time chunk c of trajectory k: it draws that chunk's noise, integrates the
// chunk from a zero state, an affine scan across the C lanes gives every chunk its true
// start state, and a second pass over the SAME register-resident noise evaluates dynamics
// and cost in the reference's operation order.  C = 64 is "one wavefront per trajectory";
// small C keeps more of the recurrence sequential.  All float arithmetic of dynamics and
// cost is compiled without FMA contraction (-ffp-contract=off) so that with C = 1
// (k_rollout_stream) the cost is bit-identical to the reference's host arithmetic.
//
// No MFMA: there is no dense contraction on this path. Bound: HBM (E store) / VALU (Philox).
#include "kernels.hpp"

#include <rocrand/rocrand_kernel.h>

namespace mppi {

// ------------------------------------------------------------------------------------------
// rocRAND Philox4x32-10, addressed by counter.  ten_rounds() is a protected member of
// rocRAND's engine; deriving from it lets a lane evaluate block (counter, key) directly
// -- random access in (sample, time) with no stored generator state.  Identical words to
// rocrand_init(seed, subsequence = k, offset = 4*block) + rocrand4() (tests check this).
// ------------------------------------------------------------------------------------------
struct PhiloxAt : public rocrand_device::philox4x32_10_engine {
    __device__ __forceinline__ static uint4 block(unsigned long long blk, unsigned long long k,
                                                  unsigned long long seed)
    {
        PhiloxAt eng;
        uint4 ctr;
        ctr.x = static_cast<unsigned int>(blk);
        ctr.y = static_cast<unsigned int>(blk >> 32);
        ctr.z = static_cast<unsigned int>(k);
        ctr.w = static_cast<unsigned int>(k >> 32);
        uint2 key;
        key.x = static_cast<unsigned int>(seed);
        key.y = static_cast<unsigned int>(seed >> 32);
        return eng.ten_rounds(ctr, key);
    }
};

template <int A>
struct Dim {
    static_assert(A >= 1 && A <= 4, "act_dim 1..4");
    static constexpr int S = 2 * A;
    static constexpr int SPB = 4 / A;      // time steps per Philox block
    static constexpr int W = SPB * A;      // normals used (and stored) per block
};

struct __attribute__((packed, aligned(4))) F3 {
    float x, y, z;
};

template <int W>
__device__ __forceinline__ void store_block(float* dst, const float* v)
{
    if constexpr (W == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        F3 t{v[0], v[1], v[2]};
        *reinterpret_cast<F3*>(dst) = t;
    }
}

template <int W>
__device__ __forceinline__ void load_block(const float* src, float* v)
{
    if constexpr (W == 4) {
        float4 t = *reinterpret_cast<const float4*>(src);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        F3 t = *reinterpret_cast<const F3*>(src);
        v[0] = t.x; v[1] = t.y; v[2] = t.z;
    }
}

__device__ __forceinline__ float wave_min(float x)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x = fminf(x, __shfl_xor(x, d));
    return x;
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

// One Euler step of the double integrator, reference src/point_mass_gpu.cu:97-106 with
// x_gain = {1, dt, 0, 1}, u_gain = {B0, dt}: the multiplications by 1 and 0 are exact and
// dropped; every remaining product and sum rounds separately, left to right.
template <int A>
__device__ __forceinline__ void lti_step(float (&p)[A], float (&v)[A], const float* u,
                                         const float* e, float dt, float B0)
{
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float a = u[i] + e[i];
        const float pn = (p[i] + dt * v[i]) + B0 * a;
        const float vn = v[i] + dt * a;
        p[i] = pn;
        v[i] = vn;
    }
}

// Cost::step_cost, reference src/cost.cu:42-55, on the state AFTER the step.
template <int A>
__device__ __forceinline__ float stage_cost(const float (&p)[A], const float (&v)[A],
                                            const float* u, const float* e,
                                            const RolloutArgs& g)
{
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) r += (u[i] * g.inv_s[i]) * e[i];
    r *= g.lambda;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float d = p[i] - g.goal[i];
        r += (d * g.w[i]) * d;
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float d = v[i] - g.goal[A + i];
        r += (d * g.w[A + i]) * d;
    }
    return r;
}

// Cost::final_cost, reference src/cost.cu:57-64.
template <int A>
__device__ __forceinline__ float final_cost(const float (&p)[A], const float (&v)[A],
                                            const RolloutArgs& g)
{
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float d = p[i] - g.goal[i];
        r += (d * g.w[i]) * d;
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float d = v[i] - g.goal[A + i];
        r += (d * g.w[A + i]) * d;
    }
    return r;
}

// Draw the W normals of Philox block (solve, bq) of global sample kglob and scale by sigma.
template <int A>
__device__ __forceinline__ void draw_block(unsigned long long blk, unsigned long long kglob,
                                           const RolloutArgs& g, float* e)
{
    constexpr int W = Dim<A>::W;
    const uint4 r = PhiloxAt::block(blk, kglob, g.seed);
    const float4 z = rocrand_device::detail::normal_distribution4(r);
    const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
    for (int i = 0; i < W; ++i) e[i] = g.sigma[i % A] * zz[i];
}

// Block-level tail shared by both rollout kernels: given every lane's path cost, fold this
// tile group into the block's running (min, exp-sum, weighted-noise sums).
//   misc : [8] LDS floats, wsum : [4][TAp] LDS, nrun : [TAp] LDS (thread n owns nrun[n])
struct RunState {
    float M;      // running minimum of the block
    float S;      // running sum of exp(-(c-M)/lambda)
};

__device__ __forceinline__ float tile_min(float cost_or_inf, float* misc, int wave, int lane)
{
    const float m = wave_min(cost_or_inf);
    if (lane == 0) misc[wave] = m;
    __syncthreads();
    return fminf(fminf(misc[0], misc[1]), fminf(misc[2], misc[3]));
}

// after wsum[][] and misc[4..7] are written and a barrier has passed
__device__ __forceinline__ void fold_tile(RunState& rs, float m_t, const float* misc,
                                          const float* wsum, float* nrun, int TAp, int TA,
                                          float inv_lambda, bool first)
{
    const float s_t = ((misc[4] + misc[5]) + misc[6]) + misc[7];
    float alpha, gamma;
    float Mn;
    if (first) {
        Mn = m_t; alpha = 0.0f; gamma = 1.0f;
    } else {
        Mn = fminf(rs.M, m_t);
        alpha = expf(-inv_lambda * (rs.M - Mn));
        gamma = expf(-inv_lambda * (m_t - Mn));
    }
    for (int n = threadIdx.x; n < TA; n += kRolloutThreads) {
        const float tot = ((wsum[n] + wsum[TAp + n]) + wsum[2 * TAp + n]) + wsum[3 * TAp + n];
        const float old = first ? 0.0f : nrun[n];
        nrun[n] = alpha * old + gamma * tot;
    }
    rs.S = first ? s_t : alpha * rs.S + gamma * s_t;
    rs.M = Mn;
}

// ------------------------------------------------------------------------------------------
// Fused rollout: C lanes per trajectory, noise chunk resident in registers.
// ------------------------------------------------------------------------------------------
template <int A, int NQ, bool SAMPLE>
__global__ void __launch_bounds__(kRolloutThreads)
k_rollout_fused(const RolloutArgs g)
{
    const unsigned long long solve_idx = g.solve_idx;
    constexpr int SPB = Dim<A>::SPB;
    constexpr int W = Dim<A>::W;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* ulds = reinterpret_cast<float4*>(smem_raw);          // [NBT] U in block layout
    const int TAp = g.C * g.nq * W;
    float* wsum = reinterpret_cast<float*>(ulds + g.NBT);        // [4][TAp]
    float* nrun = wsum + 4 * TAp;                                // [TAp]
    float* misc = nrun + TAp;                                    // [8]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int C = g.C;
    const int c = lane & (C - 1);
    const int nq = g.nq;

    // stage the nominal controls in LDS, one float4 per Philox block (zero padded)
    {
        const float* Uin = g.U + (solve_idx & 1ull) * g.TA;
        for (int b = threadIdx.x; b < g.NBT; b += kRolloutThreads) {
            float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const int n = b * W + i;
                if (n < g.TA) u[i] = Uin[n];
            }
            ulds[b] = make_float4(u[0], u[1], u[2], u[3]);
        }
    }
    float x0p[A], x0v[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        x0p[i] = g.dev->x0[i];
        x0v[i] = g.dev->x0[A + i];
    }
    __syncthreads();

    // chunk geometry of this lane (same for every tile group)
    const int cbase = c * nq;                                  // first Philox block
    const int L = nq * SPB;                                    // steps per full chunk
    const int nbefore = min(c * L, g.T);
    const int ns_own = min((c + 1) * L, g.T) - nbefore;
    const bool owns_last = (ns_own > 0) && (nbefore + ns_own == g.T);
    const unsigned long long blk0 = solve_idx * (unsigned long long)g.NBT;

    RunState rs{INFINITY, 0.0f};
    bool first = true;

    for (int tb = blockIdx.x; tb < g.n_tileblk; tb += gridDim.x) {
        const long long gid = (long long)tb * kRolloutThreads + threadIdx.x;
        const long long kloc = gid >> g.logC;
        const bool valid = kloc < g.K;
        const unsigned long long kglob = (unsigned long long)(g.k_offset + kloc);
        const size_t tile = (size_t)(gid >> 6);
        float* etile = g.Eint + ((tile * nq) * 64 + lane) * W;   // + q*64*W per block

        // ---- pass 1: draw (or load) the chunk's noise, keep it in registers, store it,
        //      and integrate the chunk from a zero state ------------------------------------
        float e[NQ * W];
        float Pz[A], Vz[A];
#pragma unroll
        for (int i = 0; i < A; ++i) { Pz[i] = 0.f; Vz[i] = 0.f; }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int i = 0; i < W; ++i) e[q * W + i] = 0.f;
            if (q < nq) {
                const int bq = cbase + q;
                if (bq < g.NBT) {
                    if constexpr (SAMPLE) {
                        draw_block<A>(blk0 + (unsigned long long)bq, kglob, g, &e[q * W]);
                        store_block<W>(etile + (size_t)q * 64 * W, &e[q * W]);
                    } else {
                        load_block<W>(etile + (size_t)q * 64 * W, &e[q * W]);
                    }
                    const float4 u4 = ulds[bq];
                    const float u[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                    for (int s = 0; s < SPB; ++s) {
                        if (bq * SPB + s < g.T)
                            lti_step<A>(Pz, Vz, &u[s * A], &e[q * W + s * A], g.dt, g.B0);
                    }
                }
            }
        }

        // ---- affine scan over the C chunks: (n, P, V) o (n', P', V') =
        //      (n + n', P + n'*dt*V + P', V + V') ------------------------------------------
        {
            int nacc = ns_own;
            for (int d = 1; d < C; d <<= 1) {
                const int nl = __shfl_up(nacc, d, C);
                float Pl[A], Vl[A];
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    Pl[i] = __shfl_up(Pz[i], d, C);
                    Vl[i] = __shfl_up(Vz[i], d, C);
                }
                if (c >= d) {
                    const float tau = (float)nacc * g.dt;
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        Pz[i] = (Pl[i] + tau * Vl[i]) + Pz[i];
                        Vz[i] = Vl[i] + Vz[i];
                    }
                    nacc += nl;
                }
            }
        }
        float p[A], v[A];
        {
            const float tau0 = (float)nbefore * g.dt;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float Pex = __shfl_up(Pz[i], 1, C);
                float Vex = __shfl_up(Vz[i], 1, C);
                if (c == 0) { Pex = 0.f; Vex = 0.f; }
                p[i] = (x0p[i] + tau0 * x0v[i]) + Pex;
                v[i] = x0v[i] + Vex;
            }
        }

        // ---- pass 2: dynamics + cost over the own chunk, reference operation order --------
        float cpart = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < nq) {
                const int bq = cbase + q;
                if (bq < g.NBT) {
                    const float4 u4 = ulds[bq];
                    const float u[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                    for (int s = 0; s < SPB; ++s) {
                        if (bq * SPB + s < g.T) {
                            lti_step<A>(p, v, &u[s * A], &e[q * W + s * A], g.dt, g.B0);
                            cpart += stage_cost<A>(p, v, &u[s * A], &e[q * W + s * A], g);
                        }
                    }
                }
            }
        }
        if (owns_last) cpart += final_cost<A>(p, v, g);
        for (int d = 1; d < C; d <<= 1) cpart += __shfl_xor(cpart, d);
        const float cost = cpart;
        if (valid && c == 0) g.cost[kloc] = cost;

        // ---- block tail: min, exp weights, weighted noise sums ----------------------------
        const float m_t = tile_min(valid ? cost : INFINITY, misc, wave, lane);
        const float wt = valid ? expf(-g.inv_lambda * (cost - m_t)) : 0.0f;
        {
            const float sw = wave_sum(c == 0 ? wt : 0.0f);
            if (lane == 0) misc[4 + wave] = sw;
        }
        const float wtN = ((long long)kglob < g.k_cover) ? wt : 0.0f;
        float* wrow = wsum + wave * TAp + (lane * nq) * W;       // valid for lane < C
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < nq) {
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    float val = wtN * e[q * W + i];
                    for (int d = C; d < 64; d <<= 1) val += __shfl_xor(val, d);
                    if (lane < C) wrow[q * W + i] = val;
                }
            }
        }
        __syncthreads();
        fold_tile(rs, m_t, misc, wsum, nrun, TAp, g.TA, g.inv_lambda, first);
        first = false;
        __syncthreads();
    }

    // ---- publish the block partial ----------------------------------------------------------
    float* Nout = g.part_N + (size_t)blockIdx.x * g.TA;
    for (int n = threadIdx.x; n < g.TA; n += kRolloutThreads) Nout[n] = first ? 0.0f : nrun[n];
    if (threadIdx.x == 0) {
        g.part_m[blockIdx.x] = rs.M;
        g.part_s[blockIdx.x] = rs.S;
    }
}

// ------------------------------------------------------------------------------------------
// Strict rollout: one lane per trajectory, fully sequential in time (C = 1, nq = NBT).
// The path cost is accumulated exactly like PointMassModelGpu::run
// (reference src/point_mass_gpu.cu:111-121): bit-identical to the serial host arithmetic.
// Slow by design (noise is re-read for the weighted sums); it is the parity anchor.
// ------------------------------------------------------------------------------------------
template <int A, bool SAMPLE>
__global__ void __launch_bounds__(kRolloutThreads)
k_rollout_stream(const RolloutArgs g)
{
    const unsigned long long solve_idx = g.solve_idx;
    constexpr int SPB = Dim<A>::SPB;
    constexpr int W = Dim<A>::W;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* ulds = reinterpret_cast<float4*>(smem_raw);
    const int TAp = g.NBT * W;
    float* wsum = reinterpret_cast<float*>(ulds + g.NBT);
    float* nrun = wsum + 4 * TAp;
    float* misc = nrun + TAp;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    {
        const float* Uin = g.U + (solve_idx & 1ull) * g.TA;
        for (int b = threadIdx.x; b < g.NBT; b += kRolloutThreads) {
            float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const int n = b * W + i;
                if (n < g.TA) u[i] = Uin[n];
            }
            ulds[b] = make_float4(u[0], u[1], u[2], u[3]);
        }
    }
    __syncthreads();
    const unsigned long long blk0 = solve_idx * (unsigned long long)g.NBT;

    RunState rs{INFINITY, 0.0f};
    bool first = true;
    for (int tb = blockIdx.x; tb < g.n_tileblk; tb += gridDim.x) {
        const long long kloc = (long long)tb * kRolloutThreads + threadIdx.x;
        const bool valid = kloc < g.K;
        const unsigned long long kglob = (unsigned long long)(g.k_offset + kloc);
        const size_t tile = (size_t)(kloc >> 6);
        float* etile = g.Eint + ((tile * g.NBT) * 64 + lane) * W;

        float p[A], v[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            p[i] = g.dev->x0[i];
            v[i] = g.dev->x0[A + i];
        }
        float cost = 0.0f;
        for (int bq = 0; bq < g.NBT; ++bq) {
            float e[W];
            if constexpr (SAMPLE) {
                draw_block<A>(blk0 + (unsigned long long)bq, kglob, g, e);
                store_block<W>(etile + (size_t)bq * 64 * W, e);
            } else {
                load_block<W>(etile + (size_t)bq * 64 * W, e);
            }
            const float4 u4 = ulds[bq];
            const float u[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
            for (int s = 0; s < SPB; ++s) {
                if (bq * SPB + s < g.T) {
                    lti_step<A>(p, v, &u[s * A], &e[s * A], g.dt, g.B0);
                    cost += stage_cost<A>(p, v, &u[s * A], &e[s * A], g);
                }
            }
        }
        cost += final_cost<A>(p, v, g);
        if (valid) g.cost[kloc] = cost;

        const float m_t = tile_min(valid ? cost : INFINITY, misc, wave, lane);
        const float wt = valid ? expf(-g.inv_lambda * (cost - m_t)) : 0.0f;
        {
            const float sw = wave_sum(wt);
            if (lane == 0) misc[4 + wave] = sw;
        }
        const float wtN = ((long long)kglob < g.k_cover) ? wt : 0.0f;
        for (int bq = 0; bq < g.NBT; ++bq) {
            float e[W];
            load_block<W>(etile + (size_t)bq * 64 * W, e);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float val = wave_sum(wtN * e[i]);
                if (lane == 0) wsum[wave * TAp + bq * W + i] = val;
            }
        }
        __syncthreads();
        fold_tile(rs, m_t, misc, wsum, nrun, TAp, g.TA, g.inv_lambda, first);
        first = false;
        __syncthreads();
    }
    float* Nout = g.part_N + (size_t)blockIdx.x * g.TA;
    for (int n = threadIdx.x; n < g.TA; n += kRolloutThreads) Nout[n] = first ? 0.0f : nrun[n];
    if (threadIdx.x == 0) {
        g.part_m[blockIdx.x] = rs.M;
        g.part_s[blockIdx.x] = rs.S;
    }
}

// ------------------------------------------------------------------------------------------
// Combine: beta (src/point_mass.cu:273-322), nabla (:328-377), weighted update
// (:384-480), action read-out and shift (:195-199, :805-824) in one launch.
// Grid = ceil(TA/64) blocks x 1024 threads; every block recomputes beta and nabla from the
// (<= kMaxParts) partials in a fixed order, so the result is deterministic.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kCombineThreads)
k_combine(const CombineArgs a)
{
    const unsigned long long solve_idx = a.solve_idx;
    __shared__ float r_lds[kMaxParts];
    __shared__ float red[16 * kCombineCols];
    __shared__ float scal[32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;   // 0..15

    float mloc = INFINITY;
    for (int p = tid; p < a.n_parts; p += kCombineThreads)
        mloc = fminf(mloc, a.m[(size_t)p * a.m_stride]);
    mloc = wave_min(mloc);
    if (lane == 0) scal[wave] = mloc;
    __syncthreads();
    float beta = scal[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) beta = fminf(beta, scal[i]);

    float sloc = 0.0f;
    for (int p = tid; p < a.n_parts; p += kCombineThreads) {
        const float mp = a.m[(size_t)p * a.m_stride];
        const float r = (mp < INFINITY) ? expf(-a.inv_lambda * (mp - beta)) : 0.0f;
        r_lds[p] = r;
        sloc += r * a.s[(size_t)p * a.s_stride];
    }
    sloc = wave_sum(sloc);
    if (lane == 0) scal[16 + wave] = sloc;
    __syncthreads();
    float nabla = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) nabla += scal[16 + i];

    const int n = blockIdx.x * kCombineCols + lane;
    float acc = 0.0f;
    if (n < a.TA) {
        // 8 independent row loads in flight per lane; the accumulation order stays fixed
        for (int p0 = wave; p0 < a.n_parts; p0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = p0 + 16 * j;
                v[j] = (p < a.n_parts) ? a.N[(size_t)p * a.N_stride + n] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = p0 + 16 * j;
                if (p < a.n_parts) acc = fmaf(r_lds[p], v[j], acc);
            }
        }
    }
    red[wave * kCombineCols + lane] = acc;
    __syncthreads();
    if (wave == 0 && n < a.TA) {
        float tot = 0.0f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) tot += red[wv * kCombineCols + lane];
        if (a.final_mode) {
            const float* Uin = a.U + (solve_idx & 1ull) * a.TA;
            float* Uout = a.U + ((solve_idx + 1ull) & 1ull) * a.TA;
            const float unew = Uin[n] + tot / nabla;
            if (n < a.A) {
                a.act_dev[n] = unew;
                if (a.act_host) a.act_host[n] = unew;
            } else {
                Uout[n - a.A] = unew;
            }
            if (n >= a.TA - a.A) Uout[n] = unew;   // last step repeated
        } else {
            a.partial_out[2 + n] = tot;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (a.final_mode) {
            a.dev->beta = beta;
            a.dev->nabla = nabla;
        } else {
            a.partial_out[0] = beta;
            a.partial_out[1] = nabla;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Off-path kernels: layout conversion, state trace, normalised weights (debug / get_inf).
// ------------------------------------------------------------------------------------------
template <int A>
__device__ __forceinline__ size_t eint_index(long long kloc, int t, int a, int C, int nq)
{
    constexpr int SPB = Dim<A>::SPB;
    constexpr int W = Dim<A>::W;
    const int bq = t / SPB;
    const int c = bq / nq;
    const int q = bq - c * nq;
    const long long gid = kloc * C + c;
    const size_t tile = (size_t)(gid >> 6);
    const int lane = (int)(gid & 63);
    const int i = (t - bq * SPB) * A + a;
    return ((tile * nq + q) * 64 + lane) * W + i;
}

template <int A>
__global__ void k_export_noise(const float* Eint, float* E, int K, int T, int C, int nq)
{
    const size_t total = (size_t)K * T * A;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % A);
        const size_t kt = idx / A;
        const int t = (int)(kt % T);
        const long long k = (long long)(kt / T);
        E[idx] = Eint[eint_index<A>(k, t, a, C, nq)];
    }
}

template <int A>
__global__ void k_import_noise(const float* E, float* Eint, int K, int T, int C, int nq)
{
    const size_t total = (size_t)K * T * A;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % A);
        const size_t kt = idx / A;
        const int t = (int)(kt % T);
        const long long k = (long long)(kt / T);
        Eint[eint_index<A>(k, t, a, C, nq)] = E[idx];
    }
}

// X[k][t][s], t = 0..T, recomputed sequentially from the stored noise and the controls the
// rollout used (reference layout of _x, src/point_mass.cu:63).
template <int A>
__global__ void k_trace_states(const float* Eint, const float* U, const float* x0, float* X,
                               int K, int T, int C, int nq, float dt, float B0)
{
    constexpr int S = 2 * A;
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float p[A], v[A];
    float* xk = X + (size_t)k * (T + 1) * S;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        p[i] = x0[i];
        v[i] = x0[A + i];
        xk[i] = p[i];
        xk[A + i] = v[i];
    }
    for (int t = 0; t < T; ++t) {
        float u[A], e[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            u[i] = U[t * A + i];
            e[i] = Eint[eint_index<A>(k, t, i, C, nq)];
        }
        lti_step<A>(p, v, u, e, dt, B0);
#pragma unroll
        for (int i = 0; i < A; ++i) {
            xk[(size_t)(t + 1) * S + i] = p[i];
            xk[(size_t)(t + 1) * S + A + i] = v[i];
        }
    }
}

// weights_kernel, reference src/point_mass.cu:743-754 (double intermediates kept).
__global__ void k_weights(const float* cost, const DevState* dev, float lambda, float* wts,
                          int K)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double arg = -(1.0 / (double)lambda) * (double)(cost[k] - dev->beta);
    wts[k] = (float)(1.0 / (double)dev->nabla * (double)expf((float)arg));
}

// ------------------------------------------------------------------------------------------
// Host-side dispatch
// ------------------------------------------------------------------------------------------
static const int kNQ4[] = {4, 7, 13, 20};   // W = 4  (A = 1, 2, 4): <= 80 noise registers
static const int kNQ3[] = {4, 7, 13, 25};   // W = 3  (A = 3)

int rollout_pick_nq_template(int A, int nq)
{
    const int* tab = (A == 3) ? kNQ3 : kNQ4;
    for (int i = 0; i < 4; ++i)
        if (nq <= tab[i]) return tab[i];
    return 0;
}

size_t rollout_lds_bytes(int NBT, int TAp)
{
    return (size_t)NBT * 16 + (size_t)(5 * TAp + 8) * sizeof(float);
}

template <int A, int NQ>
static hipError_t launch_fused_t(bool sample, int grid, const RolloutArgs& a,
                                 hipStream_t st)
{
    const size_t lds = rollout_lds_bytes(a.NBT, a.C * a.nq * Dim<A>::W);
    if (sample)
        hipLaunchKernelGGL((k_rollout_fused<A, NQ, true>), dim3(grid), dim3(kRolloutThreads), lds,
                           st, a);
    else
        hipLaunchKernelGGL((k_rollout_fused<A, NQ, false>), dim3(grid), dim3(kRolloutThreads),
                           lds, st, a);
    return hipGetLastError();
}

template <int A>
static hipError_t launch_fused_a(int NQt, bool sample, int grid, const RolloutArgs& a,
                                 hipStream_t st)
{
    constexpr int NQL = (A == 3) ? 25 : 20;
    switch (NQt) {
        case 4: return launch_fused_t<A, 4>(sample, grid, a, st);
        case 7: return launch_fused_t<A, 7>(sample, grid, a, st);
        case 13: return launch_fused_t<A, 13>(sample, grid, a, st);
        case NQL: return launch_fused_t<A, NQL>(sample, grid, a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rollout_fused(int A, int NQt, bool sample, int grid, const RolloutArgs& a,
                                hipStream_t st)
{
    switch (A) {
        case 1: return launch_fused_a<1>(NQt, sample, grid, a, st);
        case 2: return launch_fused_a<2>(NQt, sample, grid, a, st);
        case 3: return launch_fused_a<3>(NQt, sample, grid, a, st);
        case 4: return launch_fused_a<4>(NQt, sample, grid, a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int A>
static hipError_t launch_stream_a(bool sample, int grid, const RolloutArgs& a,
                                  hipStream_t st)
{
    const size_t lds = rollout_lds_bytes(a.NBT, a.NBT * Dim<A>::W);
    if (sample)
        hipLaunchKernelGGL((k_rollout_stream<A, true>), dim3(grid), dim3(kRolloutThreads), lds, st,
                           a);
    else
        hipLaunchKernelGGL((k_rollout_stream<A, false>), dim3(grid), dim3(kRolloutThreads), lds,
                           st, a);
    return hipGetLastError();
}

hipError_t launch_rollout_stream(int A, bool sample, int grid, const RolloutArgs& a,
                                 hipStream_t st)
{
    switch (A) {
        case 1: return launch_stream_a<1>(sample, grid, a, st);
        case 2: return launch_stream_a<2>(sample, grid, a, st);
        case 3: return launch_stream_a<3>(sample, grid, a, st);
        case 4: return launch_stream_a<4>(sample, grid, a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_combine(const CombineArgs& a, hipStream_t st)
{
    const int grid = (a.TA + kCombineCols - 1) / kCombineCols;
    hipLaunchKernelGGL(k_combine, dim3(grid), dim3(kCombineThreads), 0, st, a);
    return hipGetLastError();
}

static int copy_grid(size_t total)
{
    size_t b = (total + 255) / 256;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

hipError_t launch_export_noise(int A, const float* Eint, float* E, int K, int T, int C, int nq,
                               hipStream_t st)
{
    const int grid = copy_grid((size_t)K * T * A);
    switch (A) {
        case 1: hipLaunchKernelGGL(k_export_noise<1>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, C, nq); break;
        case 2: hipLaunchKernelGGL(k_export_noise<2>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, C, nq); break;
        case 3: hipLaunchKernelGGL(k_export_noise<3>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, C, nq); break;
        case 4: hipLaunchKernelGGL(k_export_noise<4>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, C, nq); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_import_noise(int A, const float* E, float* Eint, int K, int T, int C, int nq,
                               hipStream_t st)
{
    const int grid = copy_grid((size_t)K * T * A);
    switch (A) {
        case 1: hipLaunchKernelGGL(k_import_noise<1>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, C, nq); break;
        case 2: hipLaunchKernelGGL(k_import_noise<2>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, C, nq); break;
        case 3: hipLaunchKernelGGL(k_import_noise<3>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, C, nq); break;
        case 4: hipLaunchKernelGGL(k_import_noise<4>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, C, nq); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_trace_states(int A, const float* Eint, const float* U, const float* x0, float* X,
                               int K, int T, int C, int nq, float dt, float B0, hipStream_t st)
{
    const int grid = (K + 255) / 256;
    switch (A) {
        case 1: hipLaunchKernelGGL(k_trace_states<1>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, C, nq, dt, B0); break;
        case 2: hipLaunchKernelGGL(k_trace_states<2>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, C, nq, dt, B0); break;
        case 3: hipLaunchKernelGGL(k_trace_states<3>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, C, nq, dt, B0); break;
        case 4: hipLaunchKernelGGL(k_trace_states<4>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, C, nq, dt, B0); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_weights(const float* cost, const DevState* dev, float lambda, float* wts, int K,
                          hipStream_t st)
{
    const int grid = (K + 255) / 256;
    hipLaunchKernelGGL(k_weights, dim3(grid), dim3(256), 0, st, cost, dev, lambda, wts, K);
    return hipGetLastError();
}

}  // namespace mppi
