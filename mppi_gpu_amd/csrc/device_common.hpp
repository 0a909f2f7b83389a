// device_common.hpp -- device-side building blocks shared by the rollout kernels:
// counter-addressed Philox4x32-10 (rocRAND's stream), the hardware Box-Muller, DPP / lane-swap reductions, the
// reference's step and cost arithmetic, and the per-block running (min, exp-sum, sums) fold.
#pragma once
#include "kernels.hpp"


namespace mppi {


// ------------------------------------------------------------------------------------------
// Philox4x32-10 addressed by counter: a lane evaluates block (counter, key) directly -- random
// access in (sample, time) with no stored generator state.  Identical words to rocRAND's
// rocrand_init(seed, subsequence = k, offset = 4*block) + rocrand4() (tests check this against
// the rocRAND host API and the Random123 known answer).
//
// The ten rounds are written out for gfx950 (Salmon et al., "Parallel random numbers: as easy as
// 1, 2, 3", SC'11: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 /
// 0xBB67AE85): per round two v_mad_u64_u32 (each yields the high AND the low product word) and
// two v_bitop3_b32 with truth table 0x96 = a ^ b ^ c, a gfx950 instruction hipcc does not form
// from `hi ^ ctr ^ key` by itself (it emits two v_xor_b32 each; rocRAND's ten_rounds compiles to
// 6 VALU per round, this to 4).  Measured (tools/ubench_issue, tools/ubench_int): a round issues in
// 16.6 SIMD cycles -- v_mad_u64_u32 4.5-4.9, v_bitop3_b32 with its scalar key 4.25 (2.64 with three
// vector operands: an SGPR operand halves the rate of any VALU instruction) -- and a block with its
// Box-Muller in ~270: that, not the instruction count, is the floor of the noise pass.  The round
// keys are wave-uniform and stay in SGPRs: moved to VGPRs (20 v_mov per tile) the kernels got
// SLOWER, 71.2 against 67.9 us at C3 -- the wave next door issues in the scalar-operand gaps
// (DESIGN 2.5).
struct PhiloxAt {
    __device__ __forceinline__ static unsigned int xor3(unsigned int a, unsigned int b,
                                                        unsigned int c)
    {
        return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    }
    __device__ __forceinline__ static uint4 block(unsigned long long blk, unsigned long long k,
                                                  unsigned long long seed)
    {
        unsigned int c0 = static_cast<unsigned int>(blk);
        unsigned int c1 = static_cast<unsigned int>(blk >> 32);
        unsigned int c2 = static_cast<unsigned int>(k);
        unsigned int c3 = static_cast<unsigned int>(k >> 32);
        unsigned int k0 = static_cast<unsigned int>(seed);
        unsigned int k1 = static_cast<unsigned int>(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
            const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
            const unsigned int n0 = xor3(static_cast<unsigned int>(p1 >> 32), c1, k0);
            const unsigned int n2 = xor3(static_cast<unsigned int>(p0 >> 32), c3, k1);
            c1 = static_cast<unsigned int>(p1);
            c3 = static_cast<unsigned int>(p0);
            c0 = n0;
            c2 = n2;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
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
    // x * 2^-32 is exact, so the fused form below rounds exactly like kInv + x * kInv
    const float u = fmaf((float)x, kInv, kInv);              // (0, 1]
    const float th = fmaf((float)y, kInv, kInv);             // (0, 1] revolutions
    const float r2 = -1.3862943611198906f * __builtin_amdgcn_logf(u);
    const float s = __builtin_amdgcn_sqrtf(r2);
    z0 = __builtin_amdgcn_sinf(th) * s;
    z1 = __builtin_amdgcn_cosf(th) * s;
}

// E = sigma * z for the four normals of one Philox block; element i belongs to axis (a0 + i) % A.
// With ONE sigma for every axis -- the reference's case: its 0.025 is a literal,
// src/point_mass_gpu.cu:86 -- the factor goes under the square root of the Box-Muller radius,
// sqrt(sigma^2 (-2 ln u)), and the four multiplies per block are gone (48 of the ~1 700 VALU
// instructions of a 3-D tile).  `r2c` = -2 ln 2, or -2 ln 2 sigma^2 when `one`: worked out ONCE on
// the host (noise_radius_factor, kernels.hpp) so that every kernel that draws noise -- the three
// rollouts, the regeneration for get_inf, the prefetch -- computes the same bits.
template <int A>
__device__ __forceinline__ void scaled_normals4(const uint4& r, int a0, bool one, float r2c,
                                                const float (&sigma)[A], float* e)
{
    const float kInv = 2.3283064e-10f;                       // 2^-32
    const float u0 = fmaf((float)r.x, kInv, kInv), t0 = fmaf((float)r.y, kInv, kInv);
    const float u1 = fmaf((float)r.z, kInv, kInv), t1 = fmaf((float)r.w, kInv, kInv);
    const float s0 = __builtin_amdgcn_sqrtf(r2c * __builtin_amdgcn_logf(u0));
    const float s1 = __builtin_amdgcn_sqrtf(r2c * __builtin_amdgcn_logf(u1));
    float z[4] = {__builtin_amdgcn_sinf(t0) * s0, __builtin_amdgcn_cosf(t0) * s0,
                  __builtin_amdgcn_sinf(t1) * s1, __builtin_amdgcn_cosf(t1) * s1};
    if (one) {                                   // kernel-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = z[i];
    } else {
        asm volatile("");                        // (a branch, not a select per normal)
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = sigma[(a0 + i) % A] * z[i];
    }
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

// sum over the lanes {l : l % 2^LOGC == lane % 2^LOGC} of the wave.  Rotations are applied in
// DECREASING distance (8, 4, 2, 1): before the rotation by d the data is 2d-periodic within the
// 16-lane row, so lane i and lane i^d add the same two operands (in swapped order) and every
// lane of a group ends with the same bits; the row and half swaps are symmetric by construction.
template <int LOGC>
__device__ __forceinline__ float symmetric_strided_sum(float x)
{
    if constexpr (LOGC <= 3) x += dpp<MPPI_ROW_ROR(8)>(x);
    if constexpr (LOGC <= 2) x += dpp<MPPI_ROW_ROR(4)>(x);
    if constexpr (LOGC <= 1) x += dpp<MPPI_ROW_ROR(2)>(x);
    if constexpr (LOGC <= 0) x += dpp<MPPI_ROW_ROR(1)>(x);
    if constexpr (LOGC <= 4) { float a, b; swap16(x, a, b); x = a + b; }
    if constexpr (LOGC <= 5) { float a, b; swap32(x, a, b); x = a + b; }
    return x;
}

// In-place two-register lane swaps (gfx950):
//   swap16_pair(a, b): a' = {a.r0, b.r0, a.r2, b.r2}, b' = {a.r1, b.r1, a.r3, b.r3}   (16-lane rows)
//   swap32_pair(a, b): a' = {a.lo32, b.lo32},          b' = {a.hi32, b.hi32}
// so a' + b' sums each input over a pair of rows (halves) AND sorts the two inputs into
// different rows (halves): one swap + one add is a reduce-scatter step for two values.
__device__ __forceinline__ void swap16_pair(float& a, float& b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32_pair(float& a, float& b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// Weighted-noise sums of one Philox block: x[0..3] are this lane's four weighted normals of the
// block; the sum of each over the lanes with the same chunk index c (lane % 2^LOGC) is written
// to dst[0..3] (the block's four floats in the wave's LDS row).  Reduce-scatter: inside a
// 16-lane row by DPP rotations (decreasing distance: symmetric), across the four rows by the
// pair swaps above, after which row t of the wave holds the finished sum of x[t]; 2.75
// instructions per value at C = 16 instead of 7 for four independent butterflies.
template <int LOGC>
__device__ __forceinline__ void nreduce_block(float (&x)[4], float* dst, int lane)
{
    constexpr int C = 1 << LOGC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (LOGC <= 3) x[i] += dpp<MPPI_ROW_ROR(8)>(x[i]);
        if constexpr (LOGC <= 2) x[i] += dpp<MPPI_ROW_ROR(4)>(x[i]);
        if constexpr (LOGC <= 1) x[i] += dpp<MPPI_ROW_ROR(2)>(x[i]);
        if constexpr (LOGC <= 0) x[i] += dpp<MPPI_ROW_ROR(1)>(x[i]);
    }
    if constexpr (LOGC <= 4) {
        swap16_pair(x[0], x[1]);
        swap16_pair(x[2], x[3]);
        float y01 = x[0] + x[1];
        float y23 = x[2] + x[3];
        swap32_pair(y01, y23);
        const float z = y01 + y23;                      // row t of the wave: total of x[t]
        if ((lane & 15) < C) dst[lane >> 4] = z;
    } else if constexpr (LOGC == 5) {
        swap32_pair(x[0], x[1]);
        swap32_pair(x[2], x[3]);
        dst[lane >> 5] = x[0] + x[1];                   // half t of the wave: total of x[t]
        dst[2 + (lane >> 5)] = x[2] + x[3];
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
    }
}

// DPP moves used by the chunk scan (all verified on gfx950 by tools/dpp_probe):
//   row_shr:D      lane i <- lane i-D inside its 16-lane row (0 shifted in)
//   row_bcast:15   rows 1 and 3 <- lane 15 of the row below     (row_mask 0xA)
//   row_bcast:31   rows 2 and 3 <- lane 31                      (row_mask 0xC)
//   wave_shr:1     lane i <- lane i-1 across the whole wave
constexpr int kRowBcast15 = 0x142;
constexpr int kRowBcast31 = 0x143;
constexpr int kWaveShr1 = 0x138;
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_rows(float x)
{
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_rows_i(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xf, false);
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
    float sg[A];
#pragma unroll
    for (int i = 0; i < A; ++i) sg[i] = g.sigma[i];
    scaled_normals4<A>(r, a0 % A, g.sigma_one != 0, g.noise_r2c, sg, e);
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
                                          float inv_lambda, bool first, float* nout = nullptr)
{   // nout != null (block-uniform; the block's LAST tile): the folded sums go straight to the block
    // partial in global memory instead of back into nrun -- no LDS round trip, no barrier, no copy
    // loop between the last tile and the end of the block
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
        const float val = alpha * old + gamma * tot;
        if (nout) nout[n] = val;
        else nrun[n] = val;
    }
    rs.S = first ? s_t : alpha * rs.S + gamma * s_t;
    rs.M = Mn;
}

__device__ __forceinline__ void stage_controls(const RolloutArgs& g, unsigned long long solve_idx,
                                               float4* ulds)
{   // nominal controls into LDS, one float4 per Philox block, zero padded past T*A
    const float* Uin = g.U + (solve_idx & 1ull) * g.TA;
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

__device__ __forceinline__ float to_vgpr(float x)
{   // opaque move: afterwards the compiler no longer knows the value is wave-uniform
    asm volatile("" : "+v"(x));
    return x;
}

// nominal controls AND lambda*inv_s*controls into LDS, one float4 per Philox block each, zero
// padded past T*A (n_blocks blocks are staged)
template <int A>
__device__ __forceinline__ void stage_controls_pair(const RolloutArgs& g, const float* Uin,
                                                    float lambda, float4* ulds, float4* uclds,
                                                    int n_blocks, int TA)
{
    for (int b = threadIdx.x; b < n_blocks; b += kRolloutThreads) {
        float u[4], uc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = b * 4 + i;
            u[i] = (n < TA) ? Uin[n] : 0.0f;
            uc[i] = lambda * (u[i] * g.inv_s[(b * 4 + i) % A]);
        }
        ulds[b] = make_float4(u[0], u[1], u[2], u[3]);
        uclds[b] = make_float4(uc[0], uc[1], uc[2], uc[3]);
    }
}

}  // namespace mppi
