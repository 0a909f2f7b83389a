// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the MPPI solve.
//
// Replaces the reference's 12 CUDA kernels (src/point_mass.cu:493-926) and the per-sample
// object PointMassModelGpu (src/point_mass_gpu.cu) with TWO launches per solve:
//
//   k_rollout_fused  sample noise (rocRAND Philox4x32-10 words + a Box-Muller written on the
//                    hardware transcendentals), roll the point mass out, accumulate the
//                    quadratic cost (src/cost.cu:42-64), store E and cost, and reduce -- per
//                    persistent block -- the running minimum, the exp-sum and the
//                    exp-weighted noise sums relative to that minimum.
//   k_combine        beta = min, nabla = sum, dU = sum(w*E) from the per-block partials;
//                    U += dU, action = U[0], shift (src/point_mass.cu:195-199,805-824).
//
// Work decomposition: C lanes cooperate on one trajectory (C = 1..64, a power of two); lane
// (k, c) owns the time chunk c of trajectory k: it draws that chunk's noise, integrates the
// chunk from a zero state, an affine scan across the C lanes gives every chunk its true
// start state, and a second pass over the SAME register-resident noise evaluates dynamics
// and cost in the reference's operation order.  C = 64 is "one wavefront per trajectory".
// Cross-lane traffic is DPP / v_permlane*_swap (no LDS round trips); LDS holds the nominal
// controls and the per-wave partial sums.  All float arithmetic of dynamics and cost is
// compiled without FMA contraction (-ffp-contract=off) so that the sequential kernel
// (k_rollout_stream, C = 1) is bit-identical to the reference's host arithmetic.
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

// Box-Muller on the CDNA4 transcendental units.  Same uniforms as rocRAND's box_muller
// (rocrand_normal.h: u = 2^-32 + x*2^-32), with
//   radius  sqrt(-2 ln u)   = v_sqrt_f32( (-2 ln 2) * v_log_f32(u) )         (v_log is log2)
//   angle   2 pi * theta    : v_sin_f32 / v_cos_f32 take theta in REVOLUTIONS, so the
//                             2 pi multiply (and the 1/2pi inside __sincosf) disappears.
// Agrees with rocRAND's host box_muller to ~1e-6 absolute per normal (tests).
__device__ __forceinline__ void box_muller_hw(unsigned int x, unsigned int y, float& z0, float& z1)
{
    const float kInv = 2.3283064e-10f;                       // 2^-32
    const float u = kInv + (float)x * kInv;                  // (0, 1]
    const float th = kInv + (float)y * kInv;                 // (0, 1] revolutions
    const float r2 = -1.3862943611198906f * __builtin_amdgcn_logf(u);
    const float s = __builtin_amdgcn_sqrtf(r2);
    z0 = __builtin_amdgcn_sinf(th) * s;
    z1 = __builtin_amdgcn_cosf(th) * s;
}

// Geometry by action dimension.  The noise of one sample and one solve is the flat sequence
// n = t*A + a; Philox block b holds normals 4b..4b+3.  A GROUP is the smallest run of whole
// steps that is also a run of whole blocks.
template <int A>
struct Dim {
    static_assert(A >= 1 && A <= 4, "act_dim 1..4");
    static constexpr int SG = (A == 1) ? 4 : (A == 2) ? 2 : (A == 3) ? 4 : 1;   // steps / group
    static constexpr int BPG = SG * A / 4;                                       // blocks / group
};

// ---- cross-lane primitives: DPP and lane-swap instructions, no LDS ------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadXor1 = 0xB1;      // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;      // quad_perm [2,3,0,1]
constexpr int kHalfMirror = 0x141;   // lane i <-> 7-i within 8
constexpr int kRowMirror = 0x140;    // lane i <-> 15-i within 16
#define MPPI_ROW_ROR(n) (0x120 + (n))
#define MPPI_ROW_SHR(n) (0x110 + (n))

__device__ __forceinline__ void swap16(float x, float& a, float& b)
{   // a + b = x[row r] + x[row r^1]   (v_permlane16_swap_b32)
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32(float x, float& a, float& b)
{   // a + b = x[lane] + x[lane ^ 32]  (v_permlane32_swap_b32)
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// sum over the aligned group of 2^LOGC consecutive lanes; every lane of the group ends with the
// same bits
template <int LOGC>
__device__ __forceinline__ float group_sum(float x)
{
    if constexpr (LOGC >= 1) x += dpp<kQuadXor1>(x);
    if constexpr (LOGC >= 2) x += dpp<kQuadXor2>(x);
    if constexpr (LOGC >= 3) x += dpp<kHalfMirror>(x);
    if constexpr (LOGC >= 4) x += dpp<kRowMirror>(x);
    if constexpr (LOGC >= 5) { float a, b; swap16(x, a, b); x = a + b; }
    if constexpr (LOGC >= 6) { float a, b; swap32(x, a, b); x = a + b; }
    return x;
}
__device__ __forceinline__ float wave_sum(float x) { return group_sum<6>(x); }
__device__ __forceinline__ float wave_min(float x)
{
    x = fminf(x, dpp<kQuadXor1>(x));
    x = fminf(x, dpp<kQuadXor2>(x));
    x = fminf(x, dpp<kHalfMirror>(x));
    x = fminf(x, dpp<kRowMirror>(x));
    { float a, b; swap16(x, a, b); x = fminf(a, b); }
    { float a, b; swap32(x, a, b); x = fminf(a, b); }
    return x;
}

// sum over the lanes {l : l % 2^LOGC == lane % 2^LOGC} of the wave; rotation based, so the
// association differs per lane -- callers read fixed lanes only
template <int LOGC>
__device__ __forceinline__ float strided_sum(float x)
{
    if constexpr (LOGC <= 0) x += dpp<MPPI_ROW_ROR(1)>(x);
    if constexpr (LOGC <= 1) x += dpp<MPPI_ROW_ROR(2)>(x);
    if constexpr (LOGC <= 2) x += dpp<MPPI_ROW_ROR(4)>(x);
    if constexpr (LOGC <= 3) x += dpp<MPPI_ROW_ROR(8)>(x);
    if constexpr (LOGC <= 4) { float a, b; swap16(x, a, b); x = a + b; }
    if constexpr (LOGC <= 5) { float a, b; swap32(x, a, b); x = a + b; }
    return x;
}

// value of lane (lane - D) within the aligned group of 2^LOGC lanes; caller masks c < D.
// Groups of <= 16 lanes lie inside one DPP row (row_shr); wider groups cross rows and go
// through ds_bpermute.
template <int D, int LOGC>
__device__ __forceinline__ float lane_up(float x)
{
    if constexpr (D < 16 && LOGC <= 4) return dpp<MPPI_ROW_SHR(D)>(x);
    else return __shfl_up(x, D, 1 << LOGC);
}
template <int D, int LOGC>
__device__ __forceinline__ int lane_up_i(int x)
{
    if constexpr (D < 16 && LOGC <= 4) return dppi<MPPI_ROW_SHR(D)>(x);
    else return __shfl_up(x, D, 1 << LOGC);
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

// Draw the 4 normals of Philox block `blk` of global sample kglob; normal i of the block is
// flat index n = 4*(blk % NBT) + i, i.e. axis (n % A): `a0` = axis of element 0.
template <int A>
__device__ __forceinline__ void draw_block(unsigned long long blk, unsigned long long kglob,
                                           int a0, const RolloutArgs& g, float* e)
{
    const uint4 r = PhiloxAt::block(blk, kglob, g.seed);
    float z[4];
    box_muller_hw(r.x, r.y, z[0], z[1]);
    box_muller_hw(r.z, r.w, z[2], z[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = g.sigma[(a0 + i) % A] * z[i];
}

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

// Fold one tile group into the block's running (min, exp-sum, weighted-noise sums); called
// after wsum[][] and misc[4..7] are written and a barrier has passed.
//   misc : [8] LDS floats, wsum : [4][TAp] LDS, nrun : [TAp] LDS (thread n owns nrun[n])
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

__device__ __forceinline__ void stage_controls(const RolloutArgs& g, float4* ulds)
{   // nominal controls into LDS, one float4 per Philox block, zero padded past T*A
    const float* Uin = g.U + (g.solve_idx & 1ull) * g.TA;
    for (int b = threadIdx.x; b < g.NBTp; b += kRolloutThreads) {
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = b * 4 + i;
            u[i] = (n < g.TA) ? Uin[n] : 0.0f;
        }
        ulds[b] = make_float4(u[0], u[1], u[2], u[3]);
    }
}

// ------------------------------------------------------------------------------------------
// Fused rollout: 2^LOGC lanes per trajectory, NG groups per lane, noise resident in registers.
//
// Arithmetic: unlike the strict kernel this one lets products feed additions as FMAs
// (explicit fmaf), as nvcc does by default for the reference's device code; the chunk
// hand-over and the cost tree re-associate anyway, so its results are the same few-ulp
// class either way (tests state the bound).
// ------------------------------------------------------------------------------------------
template <int A>
struct LaneParams {     // wave-uniform problem constants, deliberately held in VGPRs: as kernel
    float goal[2 * A];  // arguments they and the launch geometry exceed the 102-SGPR file and
    float w[2 * A];     // every spilled scalar costs a v_readlane + s_nop in the hot loop
    float sigma[A];
    float dt, B0, dt2;
};
constexpr int kParamFloats = 32;

template <int A, int NG, bool SAMPLE, int LOGC>
__device__ __forceinline__ void fused_body(const RolloutArgs& g)
{
    constexpr int SG = Dim<A>::SG;
    constexpr int BPG = Dim<A>::BPG;
    constexpr int NE = NG * BPG * 4;          // normals held per lane
    constexpr int C = 1 << LOGC;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* ulds = reinterpret_cast<float4*>(smem_raw);          // [NBTp] U in block layout
    float4* uclds = ulds + g.NBTp;                               // [NBTp] lambda*inv_s*U
    float* plds = reinterpret_cast<float*>(uclds + g.NBTp);      // [kParamFloats]
    const int nq = g.nq;                                         // blocks per lane = ng*BPG
    const int TAp = C * nq * 4;
    float* wsum = plds + kParamFloats;                           // [4][TAp]
    float* nrun = wsum + 4 * TAp;                                // [TAp]
    float* misc = nrun + TAp;                                    // [8]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & (C - 1);
    const int ng = g.ng;

    // ---- stage U, lambda*inv_s*U and the problem constants in LDS --------------------------
    {
        const float* Uin = g.U + (g.solve_idx & 1ull) * g.TA;
        for (int b = threadIdx.x; b < g.NBTp; b += kRolloutThreads) {
            float u[4], uc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = b * 4 + i;
                u[i] = (n < g.TA) ? Uin[n] : 0.0f;
                uc[i] = g.lambda * (u[i] * g.inv_s[(b * 4 + i) % A]);
            }
            ulds[b] = make_float4(u[0], u[1], u[2], u[3]);
            uclds[b] = make_float4(uc[0], uc[1], uc[2], uc[3]);
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 2 * A; ++i) {
                plds[i] = g.goal[i];
                plds[8 + i] = g.w[i];
                plds[16 + i] = g.dev->x0[i];
            }
#pragma unroll
            for (int i = 0; i < A; ++i) plds[24 + i] = g.sigma[i];
            plds[28] = g.dt;
            plds[29] = g.B0;
        }
    }
    __syncthreads();
    LaneParams<A> P;
    float x0p[A], x0v[A];
#pragma unroll
    for (int i = 0; i < 2 * A; ++i) { P.goal[i] = plds[i]; P.w[i] = plds[8 + i]; }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        P.sigma[i] = plds[24 + i];
        x0p[i] = plds[16 + i];
        x0v[i] = plds[16 + A + i];
    }
    P.dt = plds[28];
    P.B0 = plds[29];
    P.dt2 = P.dt * P.dt;

    // chunk geometry of this lane (same for every tile group)
    const int L = g.L;                                         // steps per full chunk
    const int ns_own = (c < g.c_last) ? L : (c == g.c_last ? g.n_last : 0);
    const int nbefore = min(c * L, g.T);
    const unsigned long long blk0 = g.solve_idx * (unsigned long long)g.NBT
                                    + (unsigned long long)(c * nq);
    const float Lm1 = (float)(L - 1);

    RunState rs{INFINITY, 0.0f};
    bool first = true;

    for (int tb = blockIdx.x; tb < g.n_tileblk; tb += gridDim.x) {
        const long long gid = (long long)tb * kRolloutThreads + threadIdx.x;
        const long long kloc = gid >> LOGC;
        const bool valid = kloc < g.K;
        const unsigned long long kglob = (unsigned long long)(g.k_offset + kloc);
        const size_t tile = (size_t)(gid >> 6);
        float* etile = g.Eint + ((tile * nq) * 64 + lane) * 4;    // + q*256 floats per block

        // ---- pass 1: draw (or load) the chunk's noise into registers and store it; the
        //      chunk's zero-state response is two weighted sums of a = u + e:
        //      V = dt*S1,  P = B0*S1 + dt^2*((L-1)*S1 - S2),  S1 = sum a_j, S2 = sum j*a_j.
        //      No masking: what a partial or empty chunk adds past the horizon is not used. ---
        float e[NE];
        float S1[A], S2[A];
#pragma unroll
        for (int i = 0; i < A; ++i) { S1[i] = 0.f; S2[i] = 0.f; }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
            for (int i = 0; i < BPG * 4; ++i) e[gi * BPG * 4 + i] = 0.f;
            if (gi < ng) {
                float u[BPG * 4];
#pragma unroll
                for (int j = 0; j < BPG; ++j) {
                    const int q = gi * BPG + j;
                    float* eq = &e[q * 4];
                    if constexpr (SAMPLE) {
                        const uint4 r = PhiloxAt::block(blk0 + (unsigned long long)q, kglob, g.seed);
                        float z[4];
                        box_muller_hw(r.x, r.y, z[0], z[1]);
                        box_muller_hw(r.z, r.w, z[2], z[3]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) eq[i] = P.sigma[(q * 4 + i) % A] * z[i];
                        *reinterpret_cast<float4*>(etile + (size_t)q * 256) =
                            make_float4(eq[0], eq[1], eq[2], eq[3]);
                    } else {
                        const float4 t = *reinterpret_cast<const float4*>(etile + (size_t)q * 256);
                        eq[0] = t.x; eq[1] = t.y; eq[2] = t.z; eq[3] = t.w;
                    }
                    const float4 u4 = ulds[c * nq + q];
                    u[j * 4 + 0] = u4.x; u[j * 4 + 1] = u4.y; u[j * 4 + 2] = u4.z; u[j * 4 + 3] = u4.w;
                }
#pragma unroll
                for (int s = 0; s < SG; ++s) {
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        const float a = u[s * A + i] + e[gi * BPG * 4 + s * A + i];
                        S1[i] += a;
                        S2[i] = fmaf((float)(gi * SG + s), a, S2[i]);
                    }
                }
            }
        }
        float Pz[A], Vz[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            Vz[i] = P.dt * S1[i];
            Pz[i] = fmaf(P.dt2, fmaf(Lm1, S1[i], -S2[i]), P.B0 * S1[i]);
            if (ns_own == 0) { Pz[i] = 0.f; Vz[i] = 0.f; }
        }

        // ---- affine scan over the C chunks: (n, P, V) o (n', P', V') =
        //      (n + n', P + n'*dt*V + P', V + V') ------------------------------------------
        {
            int nacc = ns_own;
#define MPPI_SCAN_LEVEL(D)                                                          \
            if constexpr (C > (D)) {                                                \
                const int nl = lane_up_i<(D), LOGC>(nacc);                          \
                float Pl[A], Vl[A];                                                 \
                _Pragma("unroll") for (int i = 0; i < A; ++i) {                     \
                    Pl[i] = lane_up<(D), LOGC>(Pz[i]);                              \
                    Vl[i] = lane_up<(D), LOGC>(Vz[i]);                              \
                }                                                                   \
                if (c >= (D)) {                                                     \
                    const float tau = (float)nacc * P.dt;                           \
                    _Pragma("unroll") for (int i = 0; i < A; ++i) {                 \
                        Pz[i] = fmaf(tau, Vl[i], Pl[i]) + Pz[i];                    \
                        Vz[i] = Vl[i] + Vz[i];                                      \
                    }                                                               \
                    nacc += nl;                                                     \
                }                                                                   \
            }
            MPPI_SCAN_LEVEL(1)
            MPPI_SCAN_LEVEL(2)
            MPPI_SCAN_LEVEL(4)
            MPPI_SCAN_LEVEL(8)
            MPPI_SCAN_LEVEL(16)
            MPPI_SCAN_LEVEL(32)
#undef MPPI_SCAN_LEVEL
        }
        float p[A], v[A];
        {
            const float tau0 = (float)nbefore * P.dt;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float Pex = 0.f, Vex = 0.f;
                if constexpr (C > 1) {
                    Pex = lane_up<1, LOGC>(Pz[i]);
                    Vex = lane_up<1, LOGC>(Vz[i]);
                    if (c == 0) { Pex = 0.f; Vex = 0.f; }
                }
                p[i] = fmaf(tau0, x0v[i], x0p[i]) + Pex;
                v[i] = x0v[i] + Vex;
            }
        }

        // ---- pass 2: dynamics + stage cost over the own chunk (src/point_mass_gpu.cu:97-107,
        //      src/cost.cu:42-55).  No per-step masking: the chunk that holds step T-1 takes a
        //      snapshot (cost so far, state) at the wave-uniform step n_last and uses that. ----
        float cpart = 0.0f, cT = 0.0f;
        float pT[A], vT[A];
#pragma unroll
        for (int i = 0; i < A; ++i) { pT[i] = 0.f; vT[i] = 0.f; }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi < ng) {
                float u[BPG * 4], uc[BPG * 4];
#pragma unroll
                for (int j = 0; j < BPG; ++j) {
                    const float4 u4 = ulds[c * nq + gi * BPG + j];
                    const float4 c4 = uclds[c * nq + gi * BPG + j];
                    u[j * 4 + 0] = u4.x; u[j * 4 + 1] = u4.y; u[j * 4 + 2] = u4.z; u[j * 4 + 3] = u4.w;
                    uc[j * 4 + 0] = c4.x; uc[j * 4 + 1] = c4.y; uc[j * 4 + 2] = c4.z; uc[j * 4 + 3] = c4.w;
                }
#pragma unroll
                for (int s = 0; s < SG; ++s) {
                    const int sl = gi * SG + s;
                    const float* es = &e[gi * BPG * 4 + s * A];
                    float r = 0.0f;
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        const float a = u[s * A + i] + es[i];
                        const float pn = fmaf(P.B0, a, fmaf(P.dt, v[i], p[i]));
                        v[i] = fmaf(P.dt, a, v[i]);
                        p[i] = pn;
                        r = fmaf(uc[s * A + i], es[i], r);
                    }
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        const float d = p[i] - P.goal[i];
                        r = fmaf(d * P.w[i], d, r);
                    }
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        const float d = v[i] - P.goal[A + i];
                        r = fmaf(d * P.w[A + i], d, r);
                    }
                    cpart += r;
                    if (sl + 1 == g.n_last) {                  // wave-uniform
                        cT = cpart;
#pragma unroll
                        for (int i = 0; i < A; ++i) { pT[i] = p[i]; vT[i] = v[i]; }
                    }
                }
            }
        }
        {
            float fc = 0.0f;    // Cost::final_cost (src/cost.cu:57-64) on the state after step T-1
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float d = pT[i] - P.goal[i];
                fc = fmaf(d * P.w[i], d, fc);
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float d = vT[i] - P.goal[A + i];
                fc = fmaf(d * P.w[A + i], d, fc);
            }
            cpart = (c < g.c_last) ? cpart : (c == g.c_last ? cT + fc : 0.0f);
        }
        const float cost = group_sum<LOGC>(cpart);
        if (valid && c == 0) g.cost[kloc] = cost;

        // ---- block tail: min, exp weights, weighted noise sums ----------------------------
        const float m_t = tile_min(valid ? cost : INFINITY, misc, wave, lane);
        const float wt = valid ? expf(-g.inv_lambda * (cost - m_t)) : 0.0f;
        {
            const float sw = wave_sum(c == 0 ? wt : 0.0f);
            if (lane == 0) misc[4 + wave] = sw;
        }
        const float wtN = ((long long)kglob < g.k_cover) ? wt : 0.0f;
        float* wrow = wsum + wave * TAp + (lane * nq) * 4;       // valid for lane < C
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi < ng) {
#pragma unroll
                for (int i = 0; i < BPG * 4; ++i) {
                    const float val = strided_sum<LOGC>(wtN * e[gi * BPG * 4 + i]);
                    if (lane < C) wrow[gi * BPG * 4 + i] = val;
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

template <int A, int NG, bool SAMPLE>
__global__ void __launch_bounds__(kRolloutThreads)
k_rollout_fused(const RolloutArgs g)
{
    switch (g.logC) {      // wave-uniform: one specialised body per lanes-per-trajectory
        case 0: fused_body<A, NG, SAMPLE, 0>(g); break;
        case 1: fused_body<A, NG, SAMPLE, 1>(g); break;
        case 2: fused_body<A, NG, SAMPLE, 2>(g); break;
        case 3: fused_body<A, NG, SAMPLE, 3>(g); break;
        case 4: fused_body<A, NG, SAMPLE, 4>(g); break;
        case 5: fused_body<A, NG, SAMPLE, 5>(g); break;
        default: fused_body<A, NG, SAMPLE, 6>(g); break;
    }
}

// ------------------------------------------------------------------------------------------
// Strict rollout: one lane per trajectory, fully sequential in time (C = 1, nq = NBTp).
// The path cost is accumulated exactly like PointMassModelGpu::run
// (reference src/point_mass_gpu.cu:111-121): bit-identical to the serial host arithmetic.
// Slow by design (noise is re-read for the weighted sums); it is the parity anchor.
// ------------------------------------------------------------------------------------------
template <int A, bool SAMPLE>
__global__ void __launch_bounds__(kRolloutThreads)
k_rollout_stream(const RolloutArgs g)
{
    constexpr int SG = Dim<A>::SG;
    constexpr int BPG = Dim<A>::BPG;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* ulds = reinterpret_cast<float4*>(smem_raw);
    const int nq = g.nq;                     // = NBTp
    const int TAp = nq * 4;
    float* wsum = reinterpret_cast<float*>(ulds + g.NBTp);
    float* nrun = wsum + 4 * TAp;
    float* misc = nrun + TAp;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    stage_controls(g, ulds);
    __syncthreads();
    const unsigned long long blk0 = g.solve_idx * (unsigned long long)g.NBT;
    const int n_groups = nq / BPG;

    RunState rs{INFINITY, 0.0f};
    bool first = true;
    for (int tb = blockIdx.x; tb < g.n_tileblk; tb += gridDim.x) {
        const long long kloc = (long long)tb * kRolloutThreads + threadIdx.x;
        const bool valid = kloc < g.K;
        const unsigned long long kglob = (unsigned long long)(g.k_offset + kloc);
        const size_t tile = (size_t)(kloc >> 6);
        float* etile = g.Eint + ((tile * nq) * 64 + lane) * 4;

        float p[A], v[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            p[i] = g.dev->x0[i];
            v[i] = g.dev->x0[A + i];
        }
        float cost = 0.0f;
        for (int gi = 0; gi < n_groups; ++gi) {
            float e[BPG * 4], u[BPG * 4];
#pragma unroll
            for (int j = 0; j < BPG; ++j) {
                const int q = gi * BPG + j;
                float* eq = &e[j * 4];
                if constexpr (SAMPLE) {
                    draw_block<A>(blk0 + (unsigned long long)q, kglob, (j * 4) % A, g, eq);
                    *reinterpret_cast<float4*>(etile + (size_t)q * 256) =
                        make_float4(eq[0], eq[1], eq[2], eq[3]);
                } else {
                    const float4 t = *reinterpret_cast<const float4*>(etile + (size_t)q * 256);
                    eq[0] = t.x; eq[1] = t.y; eq[2] = t.z; eq[3] = t.w;
                }
                const float4 u4 = ulds[q];
                u[j * 4 + 0] = u4.x; u[j * 4 + 1] = u4.y; u[j * 4 + 2] = u4.z; u[j * 4 + 3] = u4.w;
            }
#pragma unroll
            for (int s = 0; s < SG; ++s) {
                if (gi * SG + s < g.T) {                       // wave-uniform
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
        for (int q = 0; q < nq; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(etile + (size_t)q * 256);
            const float ev[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float val = wave_sum(wtN * ev[i]);
                if (lane == 0) wsum[wave * TAp + q * 4 + i] = val;
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
    const int n = t * A + a;         // flat normal index of the sample
    const int b = n >> 2;            // Philox block
    const int c = b / nq;
    const int q = b - c * nq;
    const long long gid = kloc * C + c;
    const size_t tile = (size_t)(gid >> 6);
    const int lane = (int)(gid & 63);
    return ((tile * nq + q) * 64 + lane) * 4 + (n & 3);
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
static const int kNG1[] = {1, 2, 4, 7, 13, 20};   // BPG = 1 (A = 1, 2, 4): <= 80 noise registers
static const int kNG3[] = {1, 2, 4, 7};           // BPG = 3 (A = 3):       <= 84 noise registers

int rollout_group_steps(int A) { return (A == 1) ? 4 : (A == 2) ? 2 : (A == 3) ? 4 : 1; }
int rollout_group_blocks(int A) { return rollout_group_steps(A) * A / 4; }
int rollout_max_groups(int A) { return A == 3 ? 7 : 20; }

int rollout_pick_ng_template(int A, int ng)
{
    const int* tab = (A == 3) ? kNG3 : kNG1;
    const int n = (A == 3) ? 4 : 6;
    for (int i = 0; i < n; ++i)
        if (ng <= tab[i]) return tab[i];
    return 0;
}

size_t rollout_lds_bytes(int NBTp, int TAp)
{   // u and lambda*inv_s*u blocks, constants, 4 per-wave rows + the running row, scratch
    return (size_t)NBTp * 32 + (size_t)(kParamFloats + 5 * TAp + 8) * sizeof(float);
}

template <int A, int NG>
static hipError_t launch_fused_t(bool sample, int grid, const RolloutArgs& a, hipStream_t st)
{
    const size_t lds = rollout_lds_bytes(a.NBTp, a.C * a.nq * 4);
    if (sample)
        hipLaunchKernelGGL((k_rollout_fused<A, NG, true>), dim3(grid), dim3(kRolloutThreads), lds,
                           st, a);
    else
        hipLaunchKernelGGL((k_rollout_fused<A, NG, false>), dim3(grid), dim3(kRolloutThreads),
                           lds, st, a);
    return hipGetLastError();
}

template <int A>
static hipError_t launch_fused_a(int NGt, bool sample, int grid, const RolloutArgs& a,
                                 hipStream_t st)
{
    if constexpr (A == 3) {
        switch (NGt) {
            case 1: return launch_fused_t<A, 1>(sample, grid, a, st);
            case 2: return launch_fused_t<A, 2>(sample, grid, a, st);
            case 4: return launch_fused_t<A, 4>(sample, grid, a, st);
            case 7: return launch_fused_t<A, 7>(sample, grid, a, st);
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (NGt) {
            case 1: return launch_fused_t<A, 1>(sample, grid, a, st);
            case 2: return launch_fused_t<A, 2>(sample, grid, a, st);
            case 4: return launch_fused_t<A, 4>(sample, grid, a, st);
            case 7: return launch_fused_t<A, 7>(sample, grid, a, st);
            case 13: return launch_fused_t<A, 13>(sample, grid, a, st);
            case 20: return launch_fused_t<A, 20>(sample, grid, a, st);
            default: return hipErrorInvalidValue;
        }
    }
}

hipError_t launch_rollout_fused(int A, int NGt, bool sample, int grid, const RolloutArgs& a,
                                hipStream_t st)
{
    switch (A) {
        case 1: return launch_fused_a<1>(NGt, sample, grid, a, st);
        case 2: return launch_fused_a<2>(NGt, sample, grid, a, st);
        case 3: return launch_fused_a<3>(NGt, sample, grid, a, st);
        case 4: return launch_fused_a<4>(NGt, sample, grid, a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int A>
static hipError_t launch_stream_a(bool sample, int grid, const RolloutArgs& a, hipStream_t st)
{
    const size_t lds = rollout_lds_bytes(a.NBTp, a.nq * 4);
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
