// kernels.hpp -- host-visible launch interface of the gfx950 kernels (kernels.hip).
//
// Data layout in HBM (DESIGN.md "Layout"):
//   DevState      : x0[8] (set_x writes it with one small H2D copy) and the observables
//                   beta / nabla of the last solve.
//   U             : 2 x [T*A] floats, double-buffered by solve_idx parity (in = idx&1).
//   Eint          : noise in TILE layout: float4[tiles][nq][64]; a tile is one wavefront's
//                   64 lanes = 64/C trajectories x C time-chunks, lane = (k%(64/C))*C + c;
//                   one float4 = the 4 normals of one Philox block (flat index n = t*A + a,
//                   block n/4).  Every store and load of it is one full-wave contiguous
//                   1 KiB dwordx4 access (stores: write-through buffer_store ... sc0 sc1).
//                   PACKED layout (rollout_packed_impl.hpp): the same container, but a wavefront holds
//                   TPW whole trajectories laid end to end over its 64*NG group slots (slot
//                   s = j*NGT + r of trajectory j, group r; lane = s / NG, q = (s % NG)*BPG + b),
//                   so a trajectory does not have to fill a power-of-two number of lanes.
//   cost          : [K] floats.
//   part_m/part_s : [grid] per-block running min and exp-sum (relative to that min).
//   part_N        : [grid][Nrow] per-block weighted noise sums (relative to that min); Nrow = TA
//                   rounded up to whole Philox blocks, so that every row starts 16-byte aligned.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdlib>

namespace mppi {

struct DevState {
    float x0[8];
    float beta;    // observables of the last finished solve
    float nabla;
};

// Everything a rollout launch needs.  The kernels read it from device memory (dev_copy) and get
// only the solve index by value.
struct RolloutArgs {
    const RolloutArgs* dev_copy;   // this descriptor in device memory (what the kernels read)
    const DevState* dev;
    const float* U;        // base of the 2 x TA double buffer
    float* Eint;           // tile-layout noise (written when sampling, read when injected)
    float* cost;           // [K]
    float* part_m;         // [grid]
    float* part_s;         // [grid]
    float* part_N;         // [grid][TA]
    long long k_offset;    // global index of local sample 0 (Philox subsequence base)
    long long k_cover;     // samples with GLOBAL index >= k_cover get zero update weight,
    unsigned int cover_and; // and so do samples with (index & cover_and) != 0 (ref_compat masks)
    unsigned long long seed;
    unsigned long long solve_idx;   // solves since set_data: Philox offset, U buffer parity
    int K;                 // local samples
    int T;
    int TA;                // T*A
    int NBT;               // Philox blocks per sample per solve = ceil(T*A/4)
    int NBTp;              // blocks of U staged in LDS = max(NBT, C*nq)
    int C;                 // lanes per trajectory (power of two)
    int logC;
    int ng;                // groups per lane (group = smallest run of whole steps AND blocks)
    int nq;                // Philox blocks per lane = ng * blocks-per-group
    int L;                 // steps per full chunk = ng * steps-per-group
    int c_last;            // chunk holding step T-1; chunks beyond are empty
    int n_last;            // valid steps in chunk c_last (1..L)
    int n_tileblk;         // number of 256-lane tile groups = ceil(K*C/256)
    int packed;            // != 0: packed layout -- NG = ng groups per lane, NGT groups per
    int NGT;               //       trajectory, TPW trajectories per wavefront
    int TPW;
    int pk_nlast;          // packed: steps of the horizon in a trajectory's LAST group (1..SG; SG when
                           // T is a whole number of groups): the steps past T are masked
    int Nrow;              // floats per row of part_N (TA rounded up to whole Philox blocks)
    float dt;
    float B0;              // (float)(dt*dt/2.0)
    float lambda;
    float inv_lambda;      // 1/lambda (float)
    float goal[8];
    float w[8];
    float sigma[4];
    float noise_r2c;       // Box-Muller radius factor and "one sigma for all axes" (noise_radius_factor)
    int sigma_one;
    float inv_s[4];
    // packed kernel: dynamics on scaled state d_p = sp (p - g_p), d_v = sv (v - g_v) with
    // sp = sqrt(w_p), sv = sqrt(w_v) (2^-60 where a weight is 0): d_p' = d_p + k1 d_v + k2 a + cg,
    // d_v' = d_v + k3 a, stage cost d_p'^2 + d_v'^2; computed on the host in double precision
    float pk_sp[4], pk_sv[4], pk_k1[4], pk_k2[4], pk_k3[4], pk_cg[4], pk_gps[4], pk_gvs[4];
    int pk_has_cg;         // some velocity goal != 0
    // row-aligned kernel, same scaled dynamics for weights of EITHER sign: the scales are taken of
    // |w| and the stage-cost terms of an axis enter with sign(w), applied once per chunk
    float fs_sp[4], fs_sv[4], fs_k1[4], fs_k2[4], fs_k3[4], fs_cg[4], fs_gps[4], fs_gvs[4];
    float fs_sgp[4], fs_sgv[4];
    int store_e;           // 0: the sampled noise is not written to Eint (mppi_set_noise_store);
                           // it is a pure function of (seed, solve, sample, step) and is
                           // regenerated on request (launch_regen_noise)
    long long nt_from_tile;// packed kernel, store_e == 2: wavefront tiles below this index keep the
                           // write-through store (the head of the buffer stays resident in the
                           // memory-side cache and is overwritten there by the next solve), the rest
                           // stream past it as non-temporal stores; 0 = all non-temporal
    float x0[8];           // host copy of the current state (travels by value in RolloutHot)
    // read only on the riding path (DeferredCombine), kept here so that they cost no kernel
    // argument registers: the tagged finished controls and the device watchdog words
    const unsigned long long* fin_tag;
    int* err_dev;
    int* err_host;
    unsigned long long ride_timeout_ticks;   // 100 MHz ticks a rollout block waits for the controls
};

// What the first instructions of the fused rollout need, passed BY VALUE as kernel arguments
// (no dependent load in front of the Philox work); everything else is read through `rest`.
struct RolloutHot {
    const RolloutArgs* rest;   // the full descriptor in device memory
    const float* U_in;         // nominal controls of this solve (already offset by parity)
    float* Eint;
    unsigned long long seed;
    unsigned long long solve_idx;
    long long k_offset;
    int K, T, TA, NBT, NBTp;
    int logC, ng, nq, L, c_last, n_last, n_tileblk;
    int NGT, TPW, pk_nlast;    // packed layout only
    int Nrow;
    float x0[8];
#ifdef MPPI_TRACE      // analysis builds only (tools/trace.sh): per-block region time stamps
    unsigned long long* trace;
    int trace_tile;            // which tile of a block the packed kernel stamps (MPPI_TRACE_TILE)
#endif
};
#ifdef MPPI_TRACE
extern unsigned long long* g_mppi_trace_buf;     // [kMaxParts][16], engine.hip
#define MPPI_TRACE_PTR(h) ((h).trace)
#define MPPI_STAMP(i)                                                                    \
    do {                                                                                 \
        if (h.trace && threadIdx.x == 0) {                                               \
            h.trace[(size_t)blockIdx.x * 16 + (i)] = wall_clock64();                     \
            if ((i) == 0) {   /* where the block runs: HW_ID (SE/CU/SIMD/wave slot), XCC_ID */ \
                h.trace[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  \
                h.trace[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_getreg((31 << 11) | 20); \
            }                                                                            \
        }                                                                                \
    } while (0)
#else
#define MPPI_STAMP(i) do { } while (0)
#endif

// The Box-Muller radius factor every noise-drawing kernel uses (device_common.hpp,
// scaled_normals4): -2 ln 2, or -2 ln 2 sigma^2 when all axes share one sigma >= 0 (`one`).
// (MPPI_SIGMA_FOLD=0 in the environment keeps the multiply per normal: an A/B aid, read once)
inline float noise_radius_factor(const float* sigma, int A, int* one)
{
    static const bool allow = !(getenv("MPPI_SIGMA_FOLD") && atoi(getenv("MPPI_SIGMA_FOLD")) == 0);
    bool same = allow && sigma[0] >= 0.0f;
    for (int i = 1; i < A; ++i) same = same && sigma[i] == sigma[0];
    *one = same ? 1 : 0;
    const double c = -1.3862943611198906;
    return same ? (float)(c * (double)sigma[0] * (double)sigma[0]) : (float)c;
}

inline RolloutHot make_hot(const RolloutArgs& a)
{
    RolloutHot h;
    h.rest = a.dev_copy;
    h.U_in = a.U + (a.solve_idx & 1ull) * a.TA;
    h.Eint = a.Eint;
    h.seed = a.seed;
    h.solve_idx = a.solve_idx;
    h.k_offset = a.k_offset;
    h.K = a.K; h.T = a.T; h.TA = a.TA; h.NBT = a.NBT; h.NBTp = a.NBTp;
    h.logC = a.logC; h.ng = a.ng; h.nq = a.nq; h.L = a.L;
    h.c_last = a.c_last; h.n_last = a.n_last; h.n_tileblk = a.n_tileblk;
    h.NGT = a.NGT; h.TPW = a.TPW; h.pk_nlast = a.pk_nlast;
    h.Nrow = a.Nrow;
    for (int i = 0; i < 8; ++i) h.x0[i] = a.x0[i];
#ifdef MPPI_TRACE
    h.trace = g_mppi_trace_buf;
    h.trace_tile = getenv("MPPI_TRACE_TILE") ? atoi(getenv("MPPI_TRACE_TILE")) : 0;
#endif
#ifdef MPPI_TRACE      // latency probes of the analysis build only: the product never reads them
    if (const char* dbg = getenv("MPPI_DEBUG_SKIP")) {
        h.n_tileblk = 0;                                 // 1: prologue + epilogue only
        if (dbg[0] == '2') { h.NBTp = 0; h.TA = 0; }     // 2: (almost) empty kernel
    }
#endif
    return h;
}

// Direct exchange of the rank partials (final_mode 2): every rank owns an INBOX in uncached
// device memory, mapped into every other rank's process (hipIpc over xGMI); the rank-local
// combine stores its partial straight into all G inboxes as 8-byte words {float bits, tag} and
// then polls its own inbox until the G words it needs carry this exchange's tag -- no fence, no
// flag, no collective library on the data path (a 64-bit store is single-copy atomic).
//   inbox word (parity, src, i) at  peers[g] + (parity*G + src)*W + i
//   i = 0: beta_src, 1: S_src, 2+n: N_src[n];  tag = exchange sequence number + 1 (never 0)
struct XchgArgs {
    unsigned long long* const* peers;   // [G] device table of inbox bases, rank order
    int G, rank, W;
    int parity;
    unsigned int tag;
    unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz) before giving up
    int* err_dev;                       // set to 1 on time-out (device + pinned host copy)
    int* err_host;
};

struct CombineArgs {
    DevState* dev;
    const float* m;        // [n_parts] (stride m_stride floats)
    const float* s;
    const float* N;        // row p at N + p*N_stride, TA floats
    long long m_stride, s_stride, N_stride;
    int n_parts;
    int TA;
    int A;
    float inv_lambda;
    // final mode: update + shift U, publish action, bump solve_idx
    float* U;              // base of the 2 x TA double buffer
    float* act_dev;        // [4] device copy of the action
    unsigned long long* act_host;   // pinned, host-mapped words {action bits, act_tag}; may be null
    unsigned int act_tag;  // the engine's count of final combines: lets the host poll for THIS action
    // partial mode: out[0]=beta_g, out[1]=S_g, out[2..2+TA)=N_g
    float* partial_out;
    float* slab;           // [kMaxRowSplits][TA] row-split sums (RS > 1)
    // 256-thread shape only (else null): the row splits' sums as TAGGED 8-byte words
    // {float bits, tag}, [kMaxSmallSplits][TA], then the finished values unew[n] (before the
    // shift) [TA] -- old or complete, never torn, so neither the split meeting nor a rollout
    // riding in the same launch needs a fence
    unsigned long long* slab_tag;
    unsigned int tag;               // this combine's tag: the engine's combine count, never reused
    unsigned int* tickets; // [ceil(TA/64)] arrival counters, zero between launches
    unsigned long long solve_idx;
    int final_mode;        // 0 partial -> partial_out, 1 final, 2 partial -> peer exchange -> final
    XchgArgs x;            // final_mode 2 only
    int row_splits;        // 0 = auto
    int clamp;             // != 0: updated controls are limited to [-max_a[axis], +max_a[axis]]
    float max_a[4];        // (the reference parses max-a and drops it, src/main.cu:566-568: opt-in)
    int n_cols, RS;        // filled by the launcher: column blocks and row splits of the grid
#ifdef MPPI_TRACE
    unsigned long long* trace;
#endif
};
#ifdef MPPI_TRACE
#define MPPI_CSTAMP(i)                                                                   \
    do {                                                                                 \
        if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MPPI_CSTAMP(i) do { } while (0)
#endif

// A combine that rides at the front of the NEXT solve's rollout launch (mppi_solve_async back to
// back): the first n_blocks blocks of the grid play the combine role for solve `solve_idx` while
// the rollout blocks draw their noise, which does not depend on the controls.  A rollout block
// then takes the new controls from the TAGGED words the applying combine blocks publish (value
// n of the update, before the shift: U[i] = unew[i + A], the last step repeats), polling any word
// whose tag is not yet this combine's (bounded).  No fence and no counter is involved, so the
// megabytes of noise the rollout blocks leave dirty in the L2s are never written back for the
// hand-over (measured: agent-scope release/acquire fences here cost 4 us per launch).  Blocks are
// dispatched in index order: the combine blocks hold their slots before any rollout block can
// wait for them.  (Letting the rollout blocks add the row splits' sums themselves saves a hop but
// multiplies the polled words by RS + 2: measured slower, 17.4 vs 17.0 us at C2.)
struct DeferredCombine {
    CombineArgs c;                    // by value: a pointer would put one more memory round
                                      // trip in front of the combine the rollout blocks wait for
    int n_blocks;                     // 0 = nothing rides with this launch
};

constexpr int kRolloutThreads = 256;
constexpr int kCombineThreads = 1024;
constexpr int kCombineCols = 16;   // columns of U per combine block = 64-byte pieces of a partial row
                                   // (four rows per wave-instruction).  With kSmallCombineNR = 40 rows
                                   // in flight per lane the 625 block partials of C2 fit ONE row split
                                   // of the 256-thread combine (no split meeting: a store -> poll hop).
                                   // Measured at C2, riding: 4 columns 13.45, 8 columns 12.03,
                                   // 16 columns 11.8 us per solve -- half-line pieces cost the CUs
                                   // that host the combine role twice the requests, on a memory
                                   // pipeline the rollout blocks' noise stores are using
constexpr int kMaxParts = 4096;   // LDS r[] capacity in the combine kernel
constexpr int kMaxRowSplits = 32;
constexpr int kMaxRanks = 64;     // rank partials one combine block can hold (LDS xv[][16])
constexpr int kSmallCombineNR = 40;   // row loads in flight per lane of the 256-thread combine
                                      // (16 row groups per block: up to 640 rows per split)
constexpr int kMaxSmallSplits = 8;    // row splits of the 256-thread combine (one poll batch)

// Optional dispatch timing: when both events are non-null the launch goes through
// hipExtLaunchKernelGGL, which stamps the events with the dispatch packet's own start / end
// times (the same timestamps rocprofv3 --kernel-trace reports), not with stream-order markers.
struct LaunchTiming {
    hipEvent_t start = nullptr;
    hipEvent_t stop = nullptr;
};

// Where sample k's normal (t, a) sits in the noise buffer: the row-aligned layout is described by
// (C, nq), the packed one by (NG, NGT, TPW) with nq = NG * blocks-per-group.
struct ELayout {
    int packed;            // 0 row-aligned, 1 packed, 2 plain E[k][t][a] (regenerated noise)
    int C, nq;
    int NG, NGT, TPW;
};

// Packed rollout (rollout_packed_impl.hpp): instantiated groups-per-lane values per act_dim
// (0-terminated), its LDS need and its launcher; grid = rollout blocks, d as for the fused kernel.
const int* packed_ng_list(int A);
size_t packed_lds_bytes(int A, int NG, int NBTp, int TPW);
int packed_blocks_per_cu(int A, int NG, bool sample, size_t lds, bool ride, bool ragged);
hipError_t launch_rollout_packed(int A, int NG, bool sample, int grid, const RolloutArgs& a,
                                 const DeferredCombine& d, hipStream_t st,
                                 LaunchTiming tm = LaunchTiming());

// Group geometry by action dimension and the instantiated register-resident chunk lengths
// (template NG = groups per lane); pick returns the smallest instantiated NG >= ng, 0 if none.
int rollout_group_steps(int A);
int rollout_group_blocks(int A);
int rollout_max_groups(int A);
int rollout_pick_ng_template(int A, int ng);
size_t rollout_lds_bytes(int NBTp, int TAp);
int rollout_blocks_per_cu(int A, int NGt, bool sample, size_t lds, bool ride = false);   // occupancy API, 0 = unknown;
                                                 // ride: of the riding variant (blocks that wait for each other)
// hipSuccess iff the code object of this library loads on the current device and holds the kernels
// an engine of this act_dim launches (asked through hipFuncGetAttributes: an error code here,
// where the first launch would abort inside the runtime)
hipError_t probe_code_object(int A);


// grid = rollout blocks; d.n_blocks combine-role blocks are launched in front of them
hipError_t launch_rollout_fused(int A, int NGt, bool sample, int grid, const RolloutArgs& a,
                                const DeferredCombine& d, hipStream_t st,
                                LaunchTiming tm = LaunchTiming());
// The combine in 256-thread blocks (what rides in the rollout launch, and what flushes a deferred
// combine that found no next solve): fills a.n_cols / a.RS, returns the number of blocks.
int combine_small_prepare(CombineArgs& a);
size_t combine_small_lds_bytes();
hipError_t launch_combine_small(const CombineArgs& a_prepared, hipStream_t st,
                                LaunchTiming tm = LaunchTiming());
hipError_t launch_rollout_stream(int A, bool sample, int grid, const RolloutArgs& a,
                                 hipStream_t st, LaunchTiming tm = LaunchTiming());
// The combine in 1024-thread blocks with ticketed row splits (eager mode).
hipError_t launch_combine(const CombineArgs& a, hipStream_t st, LaunchTiming tm = LaunchTiming());

// Final combine of G gathered rank partials ([G][TA+2] floats: beta_g, S_g, N_g[TA]) in rank
// order -- the same arithmetic the direct exchange applies, so both transports give equal bits.
hipError_t launch_finish_gathered(const CombineArgs& a, const float* gathered, int G,
                                  hipStream_t st, LaunchTiming tm = LaunchTiming());

// debug / data-movement kernels (off the timed path)
hipError_t launch_export_noise(int A, const float* Eint, float* E_ktA, int K, int T,
                               const ELayout& lay, hipStream_t st);
hipError_t launch_import_noise(int A, const float* E_ktA, float* Eint, int K, int T,
                               const ELayout& lay, hipStream_t st);
hipError_t launch_trace_states(int A, const float* Eint, const float* U_rollout, const float* x0,
                               float* X, int K, int T, const ELayout& lay, float dt, float B0,
                               hipStream_t st);
hipError_t launch_weights(const float* cost, const DevState* dev, float lambda, float* wts,
                          int K, hipStream_t st);
// E[k][t][a] of solve `solve_idx` straight from the Philox counters: the very device functions the
// rollout kernels draw with, hence the very bits they would have stored
hipError_t launch_regen_noise(int A, float* E_ktA, int K, int T, unsigned long long seed,
                              unsigned long long solve_idx, long long k_offset, const float* sigma4,
                              hipStream_t st);

// LDS one block may use: gfx950 has 160 KB per CU and lets one workgroup have all of it
// (tools/lds_probe.hip); a launch beyond the 64 KB every HIP device grants opts in per kernel.
constexpr size_t kMaxLdsBytes = 160 * 1024;
constexpr size_t kDefaultLdsBytes = 64 * 1024;

// The stand-alone 256-thread combine (prepared like launch_combine_small) with the noise of solve
// `solve_idx` drawn by extra blocks behind it, in the tile layout `lay` (n_tiles wavefront tiles):
// bit for bit what the sampling rollout would draw and store
hipError_t launch_combine_small_prefetch(int A, const CombineArgs& a_prepared, float* Eint,
                                         const ELayout& lay, int K, int T, long long n_tiles,
                                         unsigned long long seed, unsigned long long solve_idx,
                                         long long k_offset, const float* sigma4, hipStream_t st,
                                         LaunchTiming tm = LaunchTiming());

// launch with or without dispatch timing
#define MPPI_LAUNCH(kernel, grid, block, lds, st, tm, ...)                                       \
    do {                                                                                         \
        if ((size_t)(lds) > mppi::kDefaultLdsBytes)                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds));   \
        if ((tm).start && (tm).stop)                                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, (tm).start, (tm).stop, 0,        \
                                  __VA_ARGS__);                                                  \
        else                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                       \
    } while (0)

}  // namespace mppi
