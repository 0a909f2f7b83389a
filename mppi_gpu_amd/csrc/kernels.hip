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
#include "device_common.hpp"
#include "combine_impl.hpp"

#include <cstring>

namespace mppi {

// ------------------------------------------------------------------------------------------
// Strict rollout: one lane per trajectory, fully sequential in time (C = 1, nq = NBTp).
// The path cost is accumulated exactly like PointMassModelGpu::run
// (reference src/point_mass_gpu.cu:111-121): bit-identical to the serial host arithmetic.
// Slow by design (noise is re-read for the weighted sums); it is the parity anchor.
// ------------------------------------------------------------------------------------------
struct X0Arg { float v[8]; };      // the current state travels by value (mppi_set_x is host-only)

template <int A, bool SAMPLE>
__global__ void __launch_bounds__(kRolloutThreads)
k_rollout_stream(const RolloutArgs* __restrict__ gp, float* __restrict__ Eint,
                 const unsigned long long solve_idx, const X0Arg x0)
{
    const RolloutArgs& g = *gp;      // Eint travels by value: the device copy does not hold it
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
    stage_controls(g, solve_idx, ulds);
    __syncthreads();
    const unsigned long long blk0 = solve_idx * (unsigned long long)g.NBT;
    const int n_groups = nq / BPG;

    RunState rs{INFINITY, 0.0f};
    bool first = true;
    for (int tb = blockIdx.x; tb < g.n_tileblk; tb += gridDim.x) {
        const long long kloc = (long long)tb * kRolloutThreads + threadIdx.x;
        const bool valid = kloc < g.K;
        const unsigned long long kglob = (unsigned long long)(g.k_offset + kloc);
        const size_t tile = (size_t)(kloc >> 6);
        float* etile = Eint + ((tile * nq) * 64 + lane) * 4;

        float p[A], v[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            p[i] = x0.v[i];
            v[i] = x0.v[A + i];
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
        const float wtN =
            ((long long)kglob < g.k_cover && ((unsigned int)kglob & g.cover_and) == 0u) ? wt : 0.0f;
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
    float* Nout = g.part_N + (size_t)blockIdx.x * g.Nrow;
    for (int n = threadIdx.x; n < g.TA; n += kRolloutThreads) Nout[n] = first ? 0.0f : nrun[n];
    if (threadIdx.x == 0) {
        g.part_m[blockIdx.x] = rs.M;
        g.part_s[blockIdx.x] = rs.S;
    }
}

// ------------------------------------------------------------------------------------------
// Combine launches (the body lives in combine_impl.hpp).
// ------------------------------------------------------------------------------------------
template <int NR>    // NR row loads in flight per lane
__global__ void __launch_bounds__(kCombineThreads)
k_combine(const CombineArgs a)
{
    __shared__ float smem[combine_smem_floats<kCombineThreads>()];
    combine_body<kCombineThreads, NR>(a, (int)blockIdx.x, carve_combine_smem<kCombineThreads>(smem));
}

// The combine in the block shape of the fused rollout (256 threads): what a deferred combine that
// found no next solve to ride with is flushed through -- same device function, same bits.
__global__ void __launch_bounds__(kRolloutThreads)
k_combine_small(const CombineArgs a)
{
    __shared__ float smem[combine_smem_floats<kRolloutThreads>()];
    combine_body<kRolloutThreads, kSmallCombineNR>(a, (int)blockIdx.x,
                                                   carve_combine_smem<kRolloutThreads>(smem));
}

// Final combine of G gathered rank partials (the collective-library transport): one block per 16
// columns, thread (g, col) loads one value, then the same finish as the direct exchange.
__global__ void __launch_bounds__(kCombineThreads)
k_finish_gathered(const CombineArgs a, const float* __restrict__ gathered, int G)
{
    __shared__ float xv[kMaxRanks * kCombineCols];
    __shared__ float xm[kMaxRanks];
    __shared__ float xs[kMaxRanks];
    const int tid = threadIdx.x;
    const int cb = blockIdx.x;
    const int g = tid / kCombineCols;
    const int c = tid & (kCombineCols - 1);
    const int nn = cb * kCombineCols + c;
    const size_t stride = (size_t)a.TA + 2;
    float uin = 0.0f;
    if (tid < kCombineCols && nn < a.TA) uin = a.U[(a.solve_idx & 1ull) * a.TA + nn];
    if (g < G) {
        const float* src = gathered + (size_t)g * stride;
        xv[g * kCombineCols + c] = (nn < a.TA) ? src[2 + nn] : 0.0f;
        if (c == 0) xm[g] = src[0];
        if (c == 1) xs[g] = src[1];
    }
    __syncthreads();
    finish_columns(a, cb, tid, G, xm, xs, xv, uin);
}

// ------------------------------------------------------------------------------------------
// Off-path kernels: layout conversion, state trace, normalised weights (debug / get_inf).
// ------------------------------------------------------------------------------------------
template <int A>
__device__ __forceinline__ size_t eint_index(long long kloc, int t, int a, const ELayout& L)
{
    const int n = t * A + a;         // flat normal index of the sample
    const int b = n >> 2;            // Philox block
    if (L.packed == 2) return ((size_t)kloc * L.NGT + t) * A + a;      // plain [k][t][a], NGT = T
    if (L.packed) {
        constexpr int BPG = Dim<A>::BPG;
        const long long tile = kloc / L.TPW;                  // one wavefront = TPW trajectories
        const int j = (int)(kloc - tile * L.TPW);
        const int r = b / BPG;                                // group of the trajectory
        const int s = j * L.NGT + r;                          // group slot of the wavefront
        const int lane = s / L.NG;
        const int q = (s - lane * L.NG) * BPG + (b - r * BPG);
        return (((size_t)tile * L.nq + q) * 64 + lane) * 4 + (n & 3);
    }
    const int c = b / L.nq;
    const int q = b - c * L.nq;
    const long long gid = kloc * L.C + c;
    const size_t tile = (size_t)(gid >> 6);
    const int lane = (int)(gid & 63);
    return ((tile * L.nq + q) * 64 + lane) * 4 + (n & 3);
}

template <int A>
__global__ void k_export_noise(const float* Eint, float* E, int K, int T, const ELayout L)
{
    const size_t total = (size_t)K * T * A;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % A);
        const size_t kt = idx / A;
        const int t = (int)(kt % T);
        const long long k = (long long)(kt / T);
        E[idx] = Eint[eint_index<A>(k, t, a, L)];
    }
}

template <int A>
__global__ void k_import_noise(const float* E, float* Eint, int K, int T, const ELayout L)
{
    const size_t total = (size_t)K * T * A;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % A);
        const size_t kt = idx / A;
        const int t = (int)(kt % T);
        const long long k = (long long)(kt / T);
        Eint[eint_index<A>(k, t, a, L)] = E[idx];
    }
}

// X[k][t][s], t = 0..T, recomputed sequentially from the stored noise and the controls the
// rollout used (reference layout of _x, src/point_mass.cu:63).
template <int A>
__global__ void k_trace_states(const float* Eint, const float* U, const float* x0, float* X,
                               int K, int T, const ELayout L, float dt, float B0)
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
            e[i] = Eint[eint_index<A>(k, t, i, L)];
        }
        lti_step<A>(p, v, u, e, dt, B0);
#pragma unroll
        for (int i = 0; i < A; ++i) {
            xk[(size_t)(t + 1) * S + i] = p[i];
            xk[(size_t)(t + 1) * S + A + i] = v[i];
        }
    }
}

// E[k][t][a] regenerated from the counters (noise not materialised by the rollout)
template <int A>
__global__ void k_regen_noise(float* E, int K, int TA, int NBT, unsigned long long seed,
                              unsigned long long solve_idx, long long k_offset, float s0, float s1,
                              float s2, float s3, float r2c, int sigma_one)
{
    const float sig4[4] = {s0, s1, s2, s3};
    float sig[A];
#pragma unroll
    for (int i = 0; i < A; ++i) sig[i] = sig4[i];
    const size_t total = (size_t)K * NBT;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const long long k = (long long)(idx / NBT);
        const int b = (int)(idx - (size_t)k * NBT);
        const uint4 r = PhiloxAt::block(solve_idx * (unsigned long long)NBT + (unsigned long long)b,
                                        (unsigned long long)(k_offset + k), seed);
        float ev[4];
        scaled_normals4<A>(r, (b * 4) % A, sigma_one != 0, r2c, sig, ev);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = b * 4 + i;
            if (n < TA) E[(size_t)k * TA + n] = ev[i];
        }
    }
}

// The noise of solve `solve_idx` written AHEAD of its rollout, in the rollout's own tile layout (one
// thread per float4 slot: a wavefront stores 1 KiB contiguous, like pass 1a), by the device
// functions the rollout draws with: the rollout then LOADS it (its injected-noise instantiation)
// and computes the very bits it would have drawn.  It runs as EXTRA BLOCKS of the stand-alone
// combine launch of a blocking call (k_combine_small_prefetch): the combine blocks come first and
// at high priority -- the host polls the action words they write, not the end of the launch -- and
// the prefetch blocks fill the chip behind them while the host holds the action; the next rollout
// follows on the same stream, so no event and no second queue is involved.
struct PrefetchArgs {
    float* Eint;
    ELayout L;
    int K, TA, NBT;
    unsigned long long seed, blk_base;
    long long k_offset;
    float sig[4];
    float r2c;
    int sigma_one;
    long long n_slots;
};

template <int A>
__device__ __forceinline__ void prefetch_body(const PrefetchArgs& p, long long first, long long stride)
{
    constexpr int BPG = Dim<A>::BPG;
    // (the engine prefetches buffers below 1.5 GB only: 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.Eint, 0, (unsigned int)(p.n_slots * 16), 0x00020000);
    for (long long g = first; g < p.n_slots; g += stride) {
        const int lane = (int)(g & 63);
        const long long tq = g >> 6;
        const int q = (int)(tq % p.L.nq);
        const long long tile = tq / p.L.nq;
        long long k;
        int b;
        bool valid;
        if (p.L.packed) {
            const int s = lane * p.L.NG + q / BPG;            // group slot of the wavefront
            const int j = s / p.L.NGT;
            k = tile * p.L.TPW + j;
            b = (s - j * p.L.NGT) * BPG + q % BPG;
            valid = j < p.L.TPW && k < p.K && b < p.NBT;
        } else {
            const int c = lane & (p.L.C - 1);
            k = tile * (64 / p.L.C) + lane / p.L.C;
            b = c * p.L.nq + q;
            valid = k < p.K && b < p.NBT;
        }
        if (!valid) continue;
        const uint4 r = PhiloxAt::block(p.blk_base + (unsigned long long)b,
                                        (unsigned long long)(p.k_offset + k), p.seed);
        float e[4], sg[A];
#pragma unroll
        for (int i = 0; i < A; ++i) sg[i] = p.sig[i];
        scaled_normals4<A>(r, (b * 4) % A, p.sigma_one != 0, p.r2c, sg, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = b * 4 + i;
            e[i] = (n < p.TA) ? e[i] : 0.0f;                   // (a ragged horizon: zero past T)
        }
        // write-through (sc0 sc1), like the rollout's own noise stores: what a plain store leaves
        // dirty in the XCD L2s is written back at the END of the launch, and the next rollout
        // waits for that end
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        const v4u val = {__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2]),
                         __float_as_uint(e[3])};
        __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (unsigned int)(g * 16), 0, 17);
    }
}

template <int A>
__global__ void __launch_bounds__(kRolloutThreads)
k_combine_small_prefetch(const CombineArgs a, const PrefetchArgs p, const int n_comb)
{
    if ((int)blockIdx.x < n_comb) {
        __builtin_amdgcn_s_setprio(3);
        __shared__ float smem[combine_smem_floats<kRolloutThreads>()];
        combine_body<kRolloutThreads, kSmallCombineNR>(a, (int)blockIdx.x,
                                                       carve_combine_smem<kRolloutThreads>(smem));
        return;
    }
    __builtin_amdgcn_s_setprio(0);
    const long long nb = (long long)gridDim.x - n_comb;
    prefetch_body<A>(p, ((long long)blockIdx.x - n_comb) * kRolloutThreads + threadIdx.x,
                     nb * kRolloutThreads);
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
{   // u and lambda*inv_s*u blocks, 4 per-wave rows + the running row, scratch, and one private
    // snapshot slot per thread (cost so far + state at the last step: 1 + 2*4 floats at most)
    return (size_t)NBTp * 32 + (size_t)(5 * TAp + 8 + 9 * kRolloutThreads) * sizeof(float);
}

template <int A>
hipError_t launch_fused_a(int NGt, bool sample, int grid, const RolloutArgs& a,
                          const DeferredCombine& d, hipStream_t st, LaunchTiming tm);
extern template hipError_t launch_fused_a<1>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
extern template hipError_t launch_fused_a<2>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
extern template hipError_t launch_fused_a<3>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
extern template hipError_t launch_fused_a<4>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
template <int A>
int fused_blocks_per_cu_a(int NGt, bool sample, size_t lds, bool ride);
extern template int fused_blocks_per_cu_a<1>(int, bool, size_t, bool);
extern template int fused_blocks_per_cu_a<2>(int, bool, size_t, bool);
extern template int fused_blocks_per_cu_a<3>(int, bool, size_t, bool);
extern template int fused_blocks_per_cu_a<4>(int, bool, size_t, bool);

// ---- packed rollout (rollout_packed_impl.hpp, instantiated in rollout_packed_a{1,2,3,4}.hip) ----
template <int A>
hipError_t launch_packed_a(int NG, bool sample, int grid, const RolloutArgs& a,
                           const DeferredCombine& d, hipStream_t st, LaunchTiming tm);
template <int A>
int packed_blocks_per_cu_a(int NG, bool sample, size_t lds, bool ride, bool ragged);
template <int A>
size_t packed_lds_bytes_a(int NG, int NBT, int TPW);
#define MPPI_PACKED_EXTERN(A_)                                                                       \
    extern template hipError_t launch_packed_a<A_>(int, bool, int, const RolloutArgs&,               \
                                                   const DeferredCombine&, hipStream_t, LaunchTiming); \
    extern template int packed_blocks_per_cu_a<A_>(int, bool, size_t, bool, bool);                           \
    extern template size_t packed_lds_bytes_a<A_>(int, int, int);
MPPI_PACKED_EXTERN(1)
MPPI_PACKED_EXTERN(2)
MPPI_PACKED_EXTERN(3)
MPPI_PACKED_EXTERN(4)
#undef MPPI_PACKED_EXTERN

// groups-per-lane values the packed kernel is instantiated for (rollout_packed_impl.hpp: PackedNG)
const int* packed_ng_list(int A)
{
    static const int l1[] = {4, 0}, l2[] = {5, 8, 0}, l3[] = {4, 0}, l4[] = {10, 0}, none[] = {0};
    switch (A) {
        case 1: return l1;
        case 2: return l2;
        case 3: return l3;
        case 4: return l4;
        default: return none;
    }
}

size_t packed_lds_bytes(int A, int NG, int NBT, int TPW)
{
    switch (A) {
        case 1: return packed_lds_bytes_a<1>(NG, NBT, TPW);
        case 2: return packed_lds_bytes_a<2>(NG, NBT, TPW);
        case 3: return packed_lds_bytes_a<3>(NG, NBT, TPW);
        case 4: return packed_lds_bytes_a<4>(NG, NBT, TPW);
        default: return 0;
    }
}

int packed_blocks_per_cu(int A, int NG, bool sample, size_t lds, bool ride, bool ragged)
{
    switch (A) {
        case 1: return packed_blocks_per_cu_a<1>(NG, sample, lds, ride, ragged);
        case 2: return packed_blocks_per_cu_a<2>(NG, sample, lds, ride, ragged);
        case 3: return packed_blocks_per_cu_a<3>(NG, sample, lds, ride, ragged);
        case 4: return packed_blocks_per_cu_a<4>(NG, sample, lds, ride, ragged);
        default: return 0;
    }
}

hipError_t launch_rollout_packed(int A, int NG, bool sample, int grid, const RolloutArgs& a,
                                 const DeferredCombine& d, hipStream_t st, LaunchTiming tm)
{
    switch (A) {
        case 1: return launch_packed_a<1>(NG, sample, grid, a, d, st, tm);
        case 2: return launch_packed_a<2>(NG, sample, grid, a, d, st, tm);
        case 3: return launch_packed_a<3>(NG, sample, grid, a, d, st, tm);
        case 4: return launch_packed_a<4>(NG, sample, grid, a, d, st, tm);
        default: return hipErrorInvalidValue;
    }
}

int rollout_blocks_per_cu(int A, int NGt, bool sample, size_t lds, bool ride)
{
    switch (A) {
        case 1: return fused_blocks_per_cu_a<1>(NGt, sample, lds, ride);
        case 2: return fused_blocks_per_cu_a<2>(NGt, sample, lds, ride);
        case 3: return fused_blocks_per_cu_a<3>(NGt, sample, lds, ride);
        case 4: return fused_blocks_per_cu_a<4>(NGt, sample, lds, ride);
        default: return 0;
    }
}

hipError_t launch_rollout_fused(int A, int NGt, bool sample, int grid, const RolloutArgs& a,
                                const DeferredCombine& d, hipStream_t st, LaunchTiming tm)
{
    switch (A) {
        case 1: return launch_fused_a<1>(NGt, sample, grid, a, d, st, tm);
        case 2: return launch_fused_a<2>(NGt, sample, grid, a, d, st, tm);
        case 3: return launch_fused_a<3>(NGt, sample, grid, a, d, st, tm);
        case 4: return launch_fused_a<4>(NGt, sample, grid, a, d, st, tm);
        default: return hipErrorInvalidValue;
    }
}

hipError_t probe_code_object(int A)
{
    hipFuncAttributes fa;
    hipError_t rc = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_combine_small));
    if (rc != hipSuccess) return rc;
    switch (A) {
        case 1: rc = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rollout_stream<1, true>)); break;
        case 2: rc = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rollout_stream<2, true>)); break;
        case 3: rc = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rollout_stream<3, true>)); break;
        case 4: rc = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rollout_stream<4, true>)); break;
        default: return hipErrorInvalidValue;
    }
    if (rc != hipSuccess) return rc;
    // the fused rollout lives in its own translation unit: the occupancy query resolves its symbol
    return rollout_blocks_per_cu(A, 1, true, rollout_lds_bytes(8, 16), false) > 0 ? hipSuccess
                                                                             : hipErrorInvalidDeviceFunction;
}

int combine_small_prepare(CombineArgs& a)
{
    constexpr int kRowGroups = (kRolloutThreads / 64) * (64 / kCombineCols);
    const int cols = (a.TA + kCombineCols - 1) / kCombineCols;
    // up to kSmallCombineNR rows per lane and split, all in flight at once, at most kMaxSmallSplits
    // splits (a second split is a store -> poll hop: one split for as long as the rows fit)
    int rs = a.row_splits > 0 ? a.row_splits : (a.n_parts + kRowGroups * kSmallCombineNR - 1) / (kRowGroups * kSmallCombineNR);
    if (rs < 1) rs = 1;
    if (rs > kMaxSmallSplits) rs = kMaxSmallSplits;
    a.n_cols = cols;
    a.RS = rs;
    return cols * rs;
}

size_t combine_small_lds_bytes() { return combine_smem_floats<kRolloutThreads>() * sizeof(float); }

hipError_t launch_combine_small(const CombineArgs& a, hipStream_t st, LaunchTiming tm)
{
    const dim3 grid(a.n_cols * a.RS), block(kRolloutThreads);
    MPPI_LAUNCH(k_combine_small, grid, block, 0, st, tm, a);
    return hipGetLastError();
}

template <int A>
static hipError_t launch_stream_a(bool sample, int grid, const RolloutArgs& a, hipStream_t st,
                                  LaunchTiming tm)
{
    const size_t lds = rollout_lds_bytes(a.NBTp, a.nq * 4);
    X0Arg x0;
    for (int i = 0; i < 8; ++i) x0.v[i] = a.x0[i];
    if (sample)
        MPPI_LAUNCH((k_rollout_stream<A, true>), dim3(grid), dim3(kRolloutThreads), lds, st, tm,
                    a.dev_copy, a.Eint, a.solve_idx, x0);
    else
        MPPI_LAUNCH((k_rollout_stream<A, false>), dim3(grid), dim3(kRolloutThreads), lds, st, tm,
                    a.dev_copy, a.Eint, a.solve_idx, x0);
    return hipGetLastError();
}

hipError_t launch_rollout_stream(int A, bool sample, int grid, const RolloutArgs& a,
                                 hipStream_t st, LaunchTiming tm)
{
    switch (A) {
        case 1: return launch_stream_a<1>(sample, grid, a, st, tm);
        case 2: return launch_stream_a<2>(sample, grid, a, st, tm);
        case 3: return launch_stream_a<3>(sample, grid, a, st, tm);
        case 4: return launch_stream_a<4>(sample, grid, a, st, tm);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finish_gathered(const CombineArgs& a, const float* gathered, int G,
                                  hipStream_t st, LaunchTiming tm)
{
    if (G < 1 || G > kMaxRanks) return hipErrorInvalidValue;
    const dim3 grid((a.TA + kCombineCols - 1) / kCombineCols), block(kCombineThreads);
    MPPI_LAUNCH(k_finish_gathered, grid, block, 0, st, tm, a, gathered, G);
    return hipGetLastError();
}

hipError_t launch_combine(const CombineArgs& a_in, hipStream_t st, LaunchTiming tm)
{
    CombineArgs a = a_in;
    const int cols = (a.TA + kCombineCols - 1) / kCombineCols;
    // Row splits meet through an agent-scope ticket (~3 us of fences), so a single split with
    // every row load in flight is preferred for as long as the rows fit 20 registers per lane.
    constexpr int kRowGroups = 16 * (64 / kCombineCols);
    int rs = a.row_splits > 0 ? a.row_splits : (a.n_parts + kRowGroups * 20 - 1) / (kRowGroups * 20);
    if (rs < 1) rs = 1;
    if (rs > kMaxRowSplits) rs = kMaxRowSplits;
    a.n_cols = cols;
    a.RS = rs;
    const dim3 grid((unsigned)(cols * rs)), block(kCombineThreads);
    const int rows_per_wave = ((a.n_parts + rs - 1) / rs + kRowGroups - 1) / kRowGroups;   // per row group
    if (rows_per_wave <= 8) MPPI_LAUNCH((k_combine<8>), grid, block, 0, st, tm, a);
    else if (rows_per_wave <= 20) MPPI_LAUNCH((k_combine<20>), grid, block, 0, st, tm, a);
    else MPPI_LAUNCH((k_combine<40>), grid, block, 0, st, tm, a);
    return hipGetLastError();
}

static int copy_grid(size_t total)
{
    size_t b = (total + 255) / 256;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

hipError_t launch_export_noise(int A, const float* Eint, float* E, int K, int T,
                               const ELayout& lay, hipStream_t st)
{
    const int grid = copy_grid((size_t)K * T * A);
    switch (A) {
        case 1: hipLaunchKernelGGL(k_export_noise<1>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, lay); break;
        case 2: hipLaunchKernelGGL(k_export_noise<2>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, lay); break;
        case 3: hipLaunchKernelGGL(k_export_noise<3>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, lay); break;
        case 4: hipLaunchKernelGGL(k_export_noise<4>, dim3(grid), dim3(256), 0, st, Eint, E, K, T, lay); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_import_noise(int A, const float* E, float* Eint, int K, int T,
                               const ELayout& lay, hipStream_t st)
{
    const int grid = copy_grid((size_t)K * T * A);
    switch (A) {
        case 1: hipLaunchKernelGGL(k_import_noise<1>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, lay); break;
        case 2: hipLaunchKernelGGL(k_import_noise<2>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, lay); break;
        case 3: hipLaunchKernelGGL(k_import_noise<3>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, lay); break;
        case 4: hipLaunchKernelGGL(k_import_noise<4>, dim3(grid), dim3(256), 0, st, E, Eint, K, T, lay); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_trace_states(int A, const float* Eint, const float* U, const float* x0, float* X,
                               int K, int T, const ELayout& lay, float dt, float B0, hipStream_t st)
{
    const int grid = (K + 255) / 256;
    switch (A) {
        case 1: hipLaunchKernelGGL(k_trace_states<1>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, lay, dt, B0); break;
        case 2: hipLaunchKernelGGL(k_trace_states<2>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, lay, dt, B0); break;
        case 3: hipLaunchKernelGGL(k_trace_states<3>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, lay, dt, B0); break;
        case 4: hipLaunchKernelGGL(k_trace_states<4>, dim3(grid), dim3(256), 0, st, Eint, U, x0, X, K, T, lay, dt, B0); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_regen_noise(int A, float* E, int K, int T, unsigned long long seed,
                              unsigned long long solve_idx, long long k_offset, const float* sg,
                              hipStream_t st)
{
    const int TA = T * A, NBT = (TA + 3) / 4;
    const int grid = copy_grid((size_t)K * NBT);
    int one = 0;
    const float r2c = noise_radius_factor(sg, A, &one);
    switch (A) {
        case 1: hipLaunchKernelGGL(k_regen_noise<1>, dim3(grid), dim3(256), 0, st, E, K, TA, NBT, seed, solve_idx, k_offset, sg[0], sg[1], sg[2], sg[3], r2c, one); break;
        case 2: hipLaunchKernelGGL(k_regen_noise<2>, dim3(grid), dim3(256), 0, st, E, K, TA, NBT, seed, solve_idx, k_offset, sg[0], sg[1], sg[2], sg[3], r2c, one); break;
        case 3: hipLaunchKernelGGL(k_regen_noise<3>, dim3(grid), dim3(256), 0, st, E, K, TA, NBT, seed, solve_idx, k_offset, sg[0], sg[1], sg[2], sg[3], r2c, one); break;
        case 4: hipLaunchKernelGGL(k_regen_noise<4>, dim3(grid), dim3(256), 0, st, E, K, TA, NBT, seed, solve_idx, k_offset, sg[0], sg[1], sg[2], sg[3], r2c, one); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_combine_small_prefetch(int A, const CombineArgs& a, float* Eint, const ELayout& lay,
                                         int K, int T, long long n_tiles, unsigned long long seed,
                                         unsigned long long solve_idx, long long k_offset,
                                         const float* sg, hipStream_t st, LaunchTiming tm)
{
    PrefetchArgs p;
    p.Eint = Eint;
    p.L = lay;
    p.K = K;
    p.TA = T * A;
    p.NBT = (p.TA + 3) / 4;
    p.seed = seed;
    p.blk_base = solve_idx * (unsigned long long)p.NBT;
    p.k_offset = k_offset;
    for (int i = 0; i < 4; ++i) p.sig[i] = sg[i];
    p.r2c = noise_radius_factor(sg, A, &p.sigma_one);
    p.n_slots = n_tiles * lay.nq * 64;
    const int n_comb = a.n_cols * a.RS;
    long long nb = (p.n_slots + kRolloutThreads - 1) / kRolloutThreads;
    if (nb > 4096) nb = 4096;
    const dim3 grid((unsigned)(n_comb + nb)), block(kRolloutThreads);
    switch (A) {
        case 1: MPPI_LAUNCH(k_combine_small_prefetch<1>, grid, block, 0, st, tm, a, p, n_comb); break;
        case 2: MPPI_LAUNCH(k_combine_small_prefetch<2>, grid, block, 0, st, tm, a, p, n_comb); break;
        case 3: MPPI_LAUNCH(k_combine_small_prefetch<3>, grid, block, 0, st, tm, a, p, n_comb); break;
        case 4: MPPI_LAUNCH(k_combine_small_prefetch<4>, grid, block, 0, st, tm, a, p, n_comb); break;
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
