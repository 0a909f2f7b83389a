// engine.hip -- host side of the engine and its C ABI (include/mppi_gpu_amd.h).
//
// One Engine owns every device buffer of one MPPI controller, like the reference's
// PointMassModel (include/point_mass.hpp:23-116) -- but as flat structure-of-arrays buffers:
// no per-sample objects, no device heap, no host loop over the horizon, no synchronisation
// between the stages of a solve (reference src/point_mass.cu:129-203 syncs 8 times and
// launches >= 3*T kernels per solve).  A solve is a rollout launch and a combine launch on one
// stream; with solves back to back the combine rides in the next rollout launch.
#include "../../include/mppi_gpu_amd.h"
#include "kernels.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t err__ = (expr);                                                            \
        if (err__ != hipSuccess)                                                              \
            return fail(MPPI_EHIP, "HIP error %d (%s) at %s:%d: %s", (int)err__,              \
                        hipGetErrorString(err__), __FILE__, __LINE__, #expr);                 \
    } while (0)

int next_pow2(int x)
{
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

int ilog2(int x)
{
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

}  // namespace

#ifdef MPPI_TRACE
unsigned long long* mppi::g_mppi_trace_buf = nullptr;
extern "C" int mppi_debug_trace(unsigned long long* out, int n_blocks)
{   // analysis builds: allocate on first call (out == null), else copy the stamps back
    const size_t bytes = (size_t)mppi::kMaxParts * 16 * sizeof(unsigned long long);
    if (!mppi::g_mppi_trace_buf) {
        if (hipMalloc(&mppi::g_mppi_trace_buf, bytes) != hipSuccess) return -1;
        (void)hipMemset(mppi::g_mppi_trace_buf, 0, bytes);
    }
    if (!out) return 0;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out, mppi::g_mppi_trace_buf, (size_t)n_blocks * 16 * sizeof(unsigned long long),
                     hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

struct mppi_engine {
    // problem
    int K = 0, T = 0, S = 0, A = 0, TA = 0, SG = 0, BPG = 0, NGT = 0, NBT = 0;
    long long k_offset = 0;
    bool sharded = false;
    float dt = 0.f, B0 = 0.f, lambda = 1.f;
    float sigma[4] = {0.025f, 0.025f, 0.025f, 0.025f};
    float inv_s[4] = {1.f, 1.f, 1.f, 1.f};
    float goal[8] = {0}, w[8] = {0}, x0[8] = {0}, x0_last[8] = {0};
    unsigned long long seed = 0;
    unsigned long long solve_idx = 0;       // solves since set_data
    bool data_set = false, have_solve = false;
    bool ref_compat = false;
    bool clamp = false;                     // opt-in action limit (mppi_set_action_limit)
    float max_a[4] = {0.f, 0.f, 0.f, 0.f};
    int verbose = 0;
    int fault = 0;                          // sticky device-watchdog code, cleared by mppi_set_data
    // tuning aids, read from the environment ONCE at mppi_create (never per launch)
    int tune_combine_splits = 0;            // MPPI_COMBINE_SPLITS: row splits of the combine, 0 = auto
    int tune_ride_max_tiles = 2;            // MPPI_RIDE_MAX_TILES: longest launch a combine rides in
    int tune_ride_long = 1;                 // MPPI_RIDE_LONG: packed launches of any length carry it
    int tune_store_mode = 0;                // MPPI_STORE_MODE: 1 write-through, 2 non-temporal noise stores (0 = by size)
    bool tune_nt_resident_set = false;      // (given explicitly: also used beyond 1.5 GB of noise)
    int tune_nt_resident_mb = 192;          // MPPI_NT_RESIDENT_MB: head of the noise buffer kept write-through in
                                            // mode 2 (it stays in the memory-side cache and is overwritten there by
                                            // the next solve: K = 2e5 129.7 -> 127.8 us; packed kernel only)
    long long resident_ride = 0;            // blocks of the riding kernel variant the chip holds at once
    long long n_rollout_launches = 0;       // since creation: rollout launches, those that carried a
    long long n_riding_launches = 0;        // combine, and combines launched on their own
    long long n_combine_launches = 0;       // (mppi_get_launch_counts)

    // geometry
    int user_chunks = 0, user_strict = 0, user_max_blocks = 0;
    int user_packing = 0;       // 0 auto, -1 never, NG > 0: packed kernel with NG groups per lane
    bool geom_ok = false;
    int C = 1, logC = 0, ng = 0, nq = 0, NGt = 0, L = 0, c_last = 0, n_last = 0, NBTp = 0;
    int n_tileblk = 0, grid = 0, strict = 0;
    bool packed = false;        // packed layout (rollout_packed_impl.hpp): ng groups per lane,
    int TPW = 0;                // TPW trajectories per wavefront, NGT groups per trajectory
    // geometry the stored noise / partials belong to
    mppi::ELayout last_lay = {0, 1, 0, 0, 0, 0};
    unsigned long long last_idx = 0;

    // device memory
    mppi::DevState* d_state = nullptr;
    float* d_U = nullptr;       // 2 x TA
    float* d_Eint = nullptr;    // noise (tile layout)
    size_t eint_floats = 0;
    float* last_E = nullptr;    // buffer the last rollout used
    float* d_cost = nullptr;
    float *d_pm = nullptr, *d_ps = nullptr, *d_pN = nullptr;
    int part_cap = 0;
    float* d_act = nullptr;
    unsigned long long* h_act = nullptr;      // pinned + mapped: A words {action bits, act_seq}
    unsigned long long* h_act_dev = nullptr;  // device alias of h_act
    unsigned int act_seq = 0;   // final combines launched; the tag of the newest action
    float* d_local_partial = nullptr;  // TA+2 (sharded path)
    mppi::RolloutArgs* d_args = nullptr;   // launch descriptor in device memory
    mppi::RolloutArgs h_args_last;         // what d_args currently holds
    bool args_valid = false;
    float* d_slab = nullptr;           // combine row-split sums
    unsigned int* d_tickets = nullptr; // combine arrival counters
    float* d_Einj = nullptr;    // injected noise, [K][T][A]
    bool injected = false, inj_dirty = false;
    bool store_noise = true;    // sampled noise is materialised in d_Eint (mppi_set_noise_store)
    bool last_injected = false; // the last rollout ran on the caller's noise (d_Einj holds it in E[k][t][a] order)
    bool last_stored = true;    // ... was, by the last rollout; if not, get_inf regenerates it from
    unsigned long long last_seed = 0;       // (seed, solve index, sample offset, sigma) of that rollout
    float last_sigma[4] = {0.f, 0.f, 0.f, 0.f};
    float* d_scratch = nullptr; // export / trace / weights scratch
    size_t scratch_floats = 0;

    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the most recent enqueue

    // noise prefetch (mppi_set_noise_prefetch): while the host holds the action of solve j, extra
    // blocks of that solve's stand-alone combine launch draw the noise of solve j+1 into a second
    // buffer in the rollout's own tile layout; the next rollout then LOADS it (its injected-noise
    // instantiation: 6.8 instead of 10.2 us at C2) and computes the same bits
    int pf_mode = 1;                     // 0 off, 1 auto, 2 whenever possible
    float* d_Epf = nullptr;
    size_t epf_floats = 0;
    bool pf_valid = false;
    unsigned long long pf_idx = 0, pf_seed = 0;
    float pf_sigma[4] = {0.f, 0.f, 0.f, 0.f};
    mppi::ELayout pf_lay = {0, 1, 0, 0, 0, 0};
    hipStream_t pf_on_stream = nullptr;  // the stream whose combine launch carried the prefetch
    bool pf_warm = false;                // the fused combine + prefetch kernel has been launched once
    long long n_pf_launched = 0, n_pf_used = 0;
    std::chrono::steady_clock::time_point t_return;   // when the last blocking get_act returned
    bool t_return_valid = false;
    double think_ema_us = 0.0;           // host time between a get_act's return and the next call

    // direct peer exchange (mppi_xchg_*)
    int xg_rank = -1, xg_world = 0, xg_W = 0;
    bool xg_connected = false;
    int xg_ranks_on_my_device = 1;                     // ranks (this one included) whose inbox is on THIS GPU
    bool xg_peer_on_my_device = false;                 // a peer's inbox lives on THIS GPU (rehearsals,
                                                       // GPU sharing): see enqueue_rollout
    unsigned long long* xg_inbox = nullptr;            // this rank's inbox (uncached device memory)
    std::vector<void*> xg_opened;                      // hipIpcOpenMemHandle mappings to close
    unsigned long long** d_xg_peers = nullptr;         // device table of the G inbox bases
    unsigned long long xg_seq = 0;                     // exchanges done; never reset
    double xg_timeout_s = 5.0;

    // profiling
    int prof = 0;                   // 0 = off, n = record every n-th solve (and the one after it)
    bool prof_now = false;
    bool prof_prev = false;         // the previous rollout launch was stamped too
    unsigned long long prof_count = 0;
    std::vector<char> ev_clean;     // per rollout pair: its predecessor was stamped as well, so the
                                    // stamp covers this dispatch alone (see mppi_kernel_ms)
    std::vector<hipEvent_t> ev[2];  // start/stop pairs: [0] rollout launches, [1] combine launches
    size_t ev_used[2] = {0, 0};

    // deferred combine (mppi_solve_async back to back): the combine of the last enqueued solve has
    // not been launched yet; it rides at the front of the next solve's rollout launch, or is
    // flushed by whatever needs its results
    int defer = 1;                          // 0 = every solve launches its own combine
    bool degraded = false;                  // ... because a device watchdog tripped (check_watchdog)
    bool pending = false;
    int pending_mode = 1;                   // 1 = final combine, 2 = peer exchange + final
    unsigned long long pending_xseq = 0;    // exchange sequence number of a pending mode-2 combine
    unsigned long long pending_idx = 0;     // solve index of the pending combine
    hipStream_t pending_stream = nullptr;
    unsigned long long* d_slab_tag = nullptr;   // [kMaxSmallSplits][TA] tagged split sums + nabla
    unsigned int u_epoch = 0;               // combines launched through the small path; tag source
    int* d_err = nullptr;                   // device watchdog word: 1 = peer exchange timed out,
    int* h_err = nullptr;                   // 2 = wait for a riding combine timed out
    int* h_err_dev = nullptr;               // (pinned + mapped mirror the kernels also write)
};

namespace {

using mppi_engine_t = mppi_engine;

int ensure_scratch(mppi_engine_t* e, size_t floats)
{
    if (e->scratch_floats >= floats) return MPPI_OK;
    if (e->d_scratch) HIPCHK(hipFree(e->d_scratch));
    e->d_scratch = nullptr;
    e->scratch_floats = 0;
    HIPCHK(hipMalloc(&e->d_scratch, floats * sizeof(float)));
    e->scratch_floats = floats;
    return MPPI_OK;
}

// Choose lanes-per-trajectory C, blocks-per-lane nq, the persistent grid; (re)allocate what
// depends on them.
int ensure_geometry(mppi_engine_t* e)
{
    if (e->geom_ok) return MPPI_OK;
    const int NGT = e->NGT;          // groups per trajectory
    int C, ng, NGt = 0, strict = e->user_strict ? 1 : 0;
    int C_pref = 0;                  // row-aligned lanes per trajectory of a multi-round launch (auto mode)
    bool row_one_round = false;      // the row-aligned launch of the chosen C fits the chip at once
    if (strict) {
        C = 1;
        ng = NGT;
    } else {
        C_pref = 0;
        const int ng_max = mppi::rollout_max_groups(e->A);
        const int Cmin = next_pow2((NGT + ng_max - 1) / ng_max);
        if (Cmin > 64)
            return fail(MPPI_EINVAL, "horizon too long for the register-resident kernel: %d steps",
                        e->T);
        if (e->user_chunks > 0) {
            C = e->user_chunks;
            if (C < Cmin || C > 64 || (C & (C - 1)))
                return fail(MPPI_EINVAL, "chunks must be a power of two in [%d, 64], got %d", Cmin,
                            C);
            C_pref = C;
        } else {
            // measured on MI355X (profiles/): lanes that hold <= 7 groups (<= 4 for act_dim 3,
            // whose groups are 3 Philox blocks) keep the kernel at >= 4 waves per SIMD; beyond
            // that, more lanes per trajectory only help while the chip is under-filled
            const int ng_pref = (e->A == 3) ? 4 : 7;
            C = Cmin;
            while (C < 64 && (NGT + C - 1) / C > ng_pref) C <<= 1;
            // Small launches (tools/chunks_probe.py, 30 shapes x 5 widths): a launch is a latency
            // chain of ~0.3 us per Philox block of a lane, so MORE lanes per trajectory shorten it
            // -- while the launch stays at <= 625 blocks, the lane still holds >= 7 blocks (>= 8 to
            // leave the 16-lane DPP row: the scans then cross rows) and up to 32 lanes (64 never
            // paid for 3-D / 4-D and 0.5 us at K = 1e3 otherwise).  And FEWER lanes when the launch
            // would overfill the chip: beyond ~640 blocks the waves share the SIMDs three deep
            // (4-D K = 7e3: 24.2 us at 32 lanes, 15.2 at 16; 2-D K = 7e3: 14.1 against 10.1).
            auto blocks_of = [&](int c) { return ((long long)e->K * c + mppi::kRolloutThreads - 1) / mppi::kRolloutThreads; };
            auto nq_of = [&](int c) { return ((NGT + c - 1) / c) * e->BPG; };
            C_pref = C;                    // the throughput choice: what a launch of several rounds runs with
            while (C < 32 && blocks_of(2 * C) <= 625 && nq_of(C) >= (2 * C <= 16 ? 7 : 8) &&
                   (NGT + 2 * C - 1) / (2 * C) >= 2)
                C <<= 1;
            // (... but no lane of more than 28 steps, what the throughput rule above allows at most:
            //  the path cost is summed along the lane in one accumulator, and the stated cost bar
            //  0.35 T 2^-24 is calibrated on lanes of that length -- tools/sweep_auto.py met 5.5e-6
            //  against a bar of 5.3e-6 at T = 256 with 32 and 64 steps per lane)
            while (C > Cmin && blocks_of(C) > (nq_of(C) <= 8 ? 640 : 512) &&
                   ((NGT + C / 2 - 1) / (C / 2)) * e->SG <= 28 &&
                   mppi::rollout_pick_ng_template(e->A, (NGT + C / 2 - 1) / (C / 2)) != 0)
                C >>= 1;
            // (C: the candidate for a launch the chip holds at once; whether it does is decided
            //  below with the occupancy of its kernel, and if not, C_pref runs)
        }
        ng = (NGT + C - 1) / C;
        NGt = mppi::rollout_pick_ng_template(e->A, ng);
        if (!NGt) return fail(MPPI_EINVAL, "no kernel for %d groups per lane", ng);
        // does the chip hold the candidate's launch at once?  (occupancy of ITS kernel, and no more
        // than 2.5 blocks per CU -- 2 where a lane holds more than 8 Philox blocks: beyond that
        // the waves share the SIMDs three and four deep and the chain gets as long as a second
        // round)  If not, the launch runs several rounds and the throughput choice C_pref is used.
        if (e->user_chunks == 0) {
            int ncu = 256;
            hipDeviceProp_t prop;
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                ncu = prop.multiProcessorCount;
            const int nq_row = ng * e->BPG;
            const int NBTp_row = (C * nq_row > e->NBT) ? C * nq_row : e->NBT;
            const size_t lds_row = mppi::rollout_lds_bytes(NBTp_row, C * nq_row * 4);
            const int per_cu_row = lds_row <= mppi::kMaxLdsBytes
                                       ? mppi::rollout_blocks_per_cu(e->A, NGt, !e->injected, lds_row) : 0;
            if (per_cu_row <= 0) (void)hipGetLastError();
            const long long blocks_row = ((long long)e->K * C + mppi::kRolloutThreads - 1) / mppi::kRolloutThreads;
            const long long cap = std::min<long long>((long long)per_cu_row * ncu,
                                                      nq_row <= 8 ? 5LL * ncu / 2 : 2LL * ncu);
            row_one_round = per_cu_row > 0 && blocks_row <= cap;
            if (!row_one_round && C != C_pref) {
                C = C_pref;
                ng = (NGT + C - 1) / C;
                NGt = mppi::rollout_pick_ng_template(e->A, ng);
                if (!NGt) return fail(MPPI_EINVAL, "no kernel for %d groups per lane", ng);
            }
        }
    }
    // Packed layout (rollout_packed_impl.hpp): whole trajectories end to end over the lanes of a
    // wavefront.  Needs a horizon of whole groups and non-negative cost weights; taken when it
    // wastes fewer group slots than the power-of-two lanes per trajectory above (or when forced).
    bool packed = false;
    int pk_NG = 0, TPW = 0;
    if (!strict && e->user_packing >= 0 && (e->user_chunks == 0 || e->user_packing > 0)) {
        bool w_ok = true;
        for (int i = 0; i < e->S; ++i) w_ok = w_ok && e->w[i] >= 0.f;
        double best = 0.0;
        for (const int* cand = mppi::packed_ng_list(e->A); *cand; ++cand) {
            const int n = *cand;
            if (e->user_packing > 0 && n != e->user_packing) continue;
            if (!w_ok || NGT < n || NGT > 64 * n) continue;
            const int tpw = 64 * n / NGT;
            // useful fraction of the group slots (a ragged last group counts its real steps)
            const double util = (double)tpw * ((double)e->T / e->SG) / (64.0 * n);
            if (util > best + 1e-9) { best = util; pk_NG = n; TPW = tpw; }
        }
        if (e->user_packing > 0 && !pk_NG)
            return fail(MPPI_EINVAL, "packed kernel with %d groups per lane not available for "
                        "T=%d act_dim=%d (needs weights >= 0, an instantiated size, and %d <= "
                        "ceil(T / %d) <= 64 x that size)",
                        e->user_packing, e->T, e->A, e->user_packing, e->SG);
        const double util_row = ((double)e->T / e->SG) / ((double)C * ng);
        packed = pk_NG > 0 && (e->user_packing > 0 || best > util_row + 0.02);
        // Packing buys throughput: fewer, fuller tiles.  A launch so short that the chip holds all
        // of its blocks at once is a latency problem instead, and there the row-aligned kernel's
        // shorter tail wins (12.0 against 14.4 us at C2) -- as long as ITS blocks all fit at once
        // (row_one_round above): 3-D K = 1e4 is 625 row-aligned blocks of which 512 are resident
        // (a second round: 26.0 us) and 500 packed ones (18.5 us); tools/k_sweep_geometry.py.
        if (packed && e->user_packing == 0 && row_one_round) packed = false;
        const bool pk_fits = pk_NG > 0 &&
                             mppi::packed_lds_bytes(e->A, pk_NG, NGT * e->BPG, TPW) <= mppi::kMaxLdsBytes;
        if (packed && e->user_packing <= 0 && !pk_fits)
            packed = false;          // horizon too long for the LDS slots: row-aligned kernel
        if (!packed && e->user_packing == 0 && pk_fits) {
            // ... and the other way round: a horizon whose row-aligned tile does not fit the LDS
            // (the per-wave weighted-noise rows grow with T*A) may still fit the packed one
            const int nq_row = ng * e->BPG;
            const int NBTp_row = (C * nq_row > e->NBT) ? C * nq_row : e->NBT;
            if (mppi::rollout_lds_bytes(NBTp_row, C * nq_row * 4) > mppi::kMaxLdsBytes) packed = true;
        }
    }
    if (packed) {
        C = 1;                      // (unused by the packed kernel)
        ng = pk_NG;
        NGt = pk_NG;
    }
    const int nq = ng * e->BPG;
    const int L = ng * e->SG;
    const int c_last = (e->T - 1) / L;
    const int n_last = e->T - c_last * L;
    // blocks of the controls staged in LDS: the packed kernel pads a ragged horizon to whole groups
    const int NBTp = packed ? NGT * e->BPG : ((C * nq > e->NBT) ? C * nq : e->NBT);
    const long long lanes = (long long)e->K * C;
    const long long ntb = packed ? (((long long)e->K + TPW - 1) / TPW + 3) / 4
                                 : (lanes + mppi::kRolloutThreads - 1) / mppi::kRolloutThreads;
    if (ntb > 0x7fffffffLL) return fail(MPPI_EINVAL, "too many samples");
    const bool ragged = e->T % e->SG != 0;     // (packed kernel: the RAGGED instantiations)
    const size_t lds_need = packed ? mppi::packed_lds_bytes(e->A, pk_NG, NGT * e->BPG, TPW)
                                   : mppi::rollout_lds_bytes(NBTp, C * nq * 4);
    int max_blocks = e->user_max_blocks;
    if (max_blocks <= 0) {
        // persistent grid = 3 x what the chip holds at once (blocks per CU from the occupancy
        // API x CUs), capped at 3072: measured best on MI355X (1x: -5 %, 5x: -1 % and a slower
        // combine).  Only efficiency depends on this number.
        int ncu = 256;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            ncu = prop.multiProcessorCount;
        const bool in_kernel_sampling = !e->injected;
        const int per_cu = strict ? 0
                           : packed ? mppi::packed_blocks_per_cu(e->A, pk_NG, in_kernel_sampling, lds_need, false, ragged)
                                    : mppi::rollout_blocks_per_cu(e->A, NGt, in_kernel_sampling, lds_need);
        // The occupancy query resolves the very instantiation this geometry launches (the rollout
        // units are translation units of their own): if it fails, the kernel is not in the loaded
        // code object and the launch would abort() inside the HIP runtime -- an error code instead.
        if (!strict && per_cu <= 0) {
            (void)hipGetLastError();
            return fail(MPPI_ENODEV, "the %s rollout kernel for act_dim %d, %d groups per lane (%s "
                        "noise) cannot be resolved in the loaded gfx950 code object",
                        packed ? "packed" : "row-aligned", e->A, NGt,
                        in_kernel_sampling ? "sampled" : "injected");
        }
        // (the packed kernel's blocks meet only once, at their end: one round of resident blocks
        //  walking all tiles is best, measured 81 us at 512 blocks against 87 at 1536, C3)
        max_blocks = strict ? 2048 : (packed ? 1 : 3) * per_cu * ncu;
        if (max_blocks > 3072) max_blocks = 3072;
    }
    {   // what the chip holds AT ONCE of the riding variant of this kernel (its blocks wait for each
        // other inside the launch): a combine rides only in a launch that fits it (enqueue_rollout)
        int ncu = 256;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            ncu = prop.multiProcessorCount;
        const bool in_kernel_sampling = !e->injected;
        const int per_cu_ride = strict ? 0
            : packed ? mppi::packed_blocks_per_cu(e->A, pk_NG, in_kernel_sampling, lds_need, true, ragged)
                     : mppi::rollout_blocks_per_cu(e->A, NGt, in_kernel_sampling, lds_need, true);
        if (!strict && per_cu_ride <= 0) {
            (void)hipGetLastError();
            return fail(MPPI_ENODEV, "the riding variant of the %s rollout kernel (act_dim %d, %d "
                        "groups per lane) cannot be resolved in the loaded gfx950 code object",
                        packed ? "packed" : "row-aligned", e->A, NGt);
        }
        e->resident_ride = (long long)per_cu_ride * ncu;
    }
    if (max_blocks > mppi::kMaxParts) max_blocks = mppi::kMaxParts;
    const int grid = (int)(ntb < max_blocks ? ntb : max_blocks);

    const size_t lds = lds_need;
    if (lds > mppi::kMaxLdsBytes)
        return fail(MPPI_EINVAL, "LDS need %zu B exceeds 160 KiB (T=%d A=%d C=%d)", lds, e->T, e->A,
                    C);

    const size_t need = (size_t)ntb * 4 * nq * 64 * 4;      // 4 wavefront tiles per tile group
    e->pf_valid = false;                                   // (another layout: a prefetch is stale)
    if (e->pf_on_stream) HIPCHK(hipStreamSynchronize(e->pf_on_stream));
    if (need > e->eint_floats) {
        if (e->d_Eint) HIPCHK(hipFree(e->d_Eint));
        e->d_Eint = nullptr;
        e->eint_floats = 0;
        HIPCHK(hipMalloc(&e->d_Eint, need * sizeof(float)));
        e->eint_floats = need;
    }
    e->last_E = e->d_Eint;
    HIPCHK(hipMemsetAsync(e->d_Eint, 0, e->eint_floats * sizeof(float), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // The last solve's noise went with the old layout's buffer: sampled noise is a pure function of
    // (seed, solve, sample, step) and mppi_get_inf / mppi_get_data regenerate it from here on, bit
    // for bit; injected noise is read back from the caller's copy (d_Einj, in E[k][t][a] order).
    e->last_stored = false;
    if (grid > e->part_cap) {
        if (e->d_pm) HIPCHK(hipFree(e->d_pm));
        if (e->d_ps) HIPCHK(hipFree(e->d_ps));
        if (e->d_pN) HIPCHK(hipFree(e->d_pN));
        e->d_pm = e->d_ps = e->d_pN = nullptr;
        HIPCHK(hipMalloc(&e->d_pm, (size_t)grid * sizeof(float)));
        HIPCHK(hipMalloc(&e->d_ps, (size_t)grid * sizeof(float)));
        HIPCHK(hipMalloc(&e->d_pN, (size_t)grid * e->NBT * 4 * sizeof(float)));
        e->part_cap = grid;
    }
    e->C = C;
    e->logC = ilog2(C);
    e->ng = ng;
    e->nq = nq;
    e->NGt = NGt;
    e->L = L;
    e->c_last = c_last;
    e->n_last = n_last;
    e->NBTp = NBTp;
    e->strict = strict;
    e->packed = packed;
    e->TPW = TPW;
    e->n_tileblk = (int)ntb;
    e->grid = grid;
    e->geom_ok = true;
    e->args_valid = false;
    e->inj_dirty = e->injected;
    return MPPI_OK;
}

// ref_compat: which samples the reference's update_act sums.
//   act_dim 3: the grid K/(256*3)+1 covers only the first 512*(K/768+1) samples, each block
//              taking 512 (reference src/point_mass.cu:387,402,839-842);
//   act_dim 1: the in-block tree stops at s > 1 (:893, and :709 in sum_red_adim), so of every
//              256-thread block only the even threads' sums reach the block partial, and at the
//              second level only the even blocks' partials reach the result: samples k with k even
//              and, once there is a second level (K >= 256), (k / 512) even.
long long ref_cover(const mppi_engine_t* e)
{
    if (!e->ref_compat || e->A != 3) return 0x7fffffffffffffffLL;
    long long cov = 512LL * (e->K / 768 + 1);
    return cov < e->K ? cov : e->K;
}

unsigned int ref_cover_and(const mppi_engine_t* e)
{
    if (!e->ref_compat || e->A != 1) return 0u;
    return e->K >= 256 ? 0x201u : 0x1u;
}

void fill_rollout_args(const mppi_engine_t* e, mppi::RolloutArgs& a)
{
    a.dev = e->d_state;
    a.fin_tag = e->d_slab_tag + (size_t)mppi::kMaxSmallSplits * e->TA;
    a.err_dev = e->d_err;
    a.err_host = e->h_err_dev;
    // a riding exchange may itself wait for a late peer: the rollout blocks outwait it
    a.ride_timeout_ticks = (unsigned long long)((2.0 + (e->xg_connected ? e->xg_timeout_s : 0.0)) * 1e8);
    a.U = e->d_U;
    a.Eint = e->d_Eint;
    a.cost = e->d_cost;
    a.part_m = e->d_pm;
    a.part_s = e->d_ps;
    a.part_N = e->d_pN;
    a.k_offset = e->k_offset;
    a.k_cover = ref_cover(e);
    a.cover_and = ref_cover_and(e);
    a.seed = e->seed;
    a.solve_idx = e->solve_idx;
    a.K = e->K;
    a.T = e->T;
    a.TA = e->TA;
    a.NBT = e->NBT;
    a.NBTp = e->NBTp;
    a.C = e->C;
    a.logC = e->logC;
    a.ng = e->ng;
    a.nq = e->nq;
    a.L = e->L;
    a.c_last = e->c_last;
    a.n_last = e->n_last;
    a.n_tileblk = e->n_tileblk;
    a.packed = e->packed ? 1 : 0;
    a.NGT = e->NGT;
    a.TPW = e->TPW;
    a.pk_nlast = e->T - (e->NGT - 1) * e->SG;
    a.Nrow = e->NBT * 4;
    a.pk_has_cg = 0;
    a.noise_r2c = mppi::noise_radius_factor(e->sigma, e->A, &a.sigma_one);
    // 0: the sampled noise is not materialised; 1: write-through stores; 2: non-temporal stores.
    // A launch whose noise fits the 256 MB memory-side cache (C3: 240 MB, rewritten by every solve)
    // is best served by write-through; beyond it the stores stream to HBM, and as write-through
    // they hold the launch at ~3.1 TB/s (K = 1e6: 767 us) where non-temporal ones let it run at its
    // VALU rate (640 us; 15-18 % from K = 2e5 up, equal or 3 % better at the C4 shard's 300 MB, 2 %
    // worse at C3's 240 MB): non-temporal from the cache's own size up.
    {
        const double noise_bytes = 4.0 * (double)e->K * e->T * e->A;
        a.store_e = (e->store_noise || e->strict) ? (noise_bytes > 256.0 * 1048576.0 ? 2 : 1) : 0;
        if (e->tune_store_mode > 0 && a.store_e) a.store_e = e->tune_store_mode;
        // bytes of noise one wavefront tile stores: nq float4 per lane
        const double tile_bytes = 1024.0 * (double)(e->nq > 0 ? e->nq : 1);
        // the head of the buffer that keeps the write-through store (it stays in the memory-side
        // cache and is overwritten there by the next solve): 192 MB, +2 % at K = 2e5 (480 MB), nothing
        // at 960 MB, nothing or 2 % worse at K = 1e6 (2.4 GB: 653 against 641 us on one box, equal
        // on another) -- none beyond 1.5 GB
        const double head_mb = (noise_bytes > 1.5e9 && !e->tune_nt_resident_set) ? 0.0
                                                                                : (double)e->tune_nt_resident_mb;
        a.nt_from_tile = (long long)(head_mb * 1048576.0 / tile_bytes);
    }
    for (int i = 0; i < e->A; ++i) {
        // scaled state of the packed kernel: d_p = sp (p - g_p), d_v = sv (v - g_v); a zero weight
        // gets the scale 2^-60, whose square vanishes against any cost (and is exact to undo)
        const double wp = e->w[i], wv = e->w[e->A + i], gp = e->goal[i], gv = e->goal[e->A + i];
        const double sp = wp > 0.0 ? sqrt(wp) : ldexp(1.0, -60);
        const double sv = wv > 0.0 ? sqrt(wv) : ldexp(1.0, -60);
        a.pk_sp[i] = (float)sp;
        a.pk_sv[i] = (float)sv;
        a.pk_k1[i] = (float)(sp * (double)e->dt / sv);
        a.pk_k2[i] = (float)(sp * (double)e->B0);
        a.pk_k3[i] = (float)(sv * (double)e->dt);
        a.pk_cg[i] = (float)(sp * (double)e->dt * gv);
        a.pk_gps[i] = (float)(sp * gp);
        a.pk_gvs[i] = (float)(sv * gv);
        if (gv != 0.0) a.pk_has_cg = 1;
        // the row-aligned kernel's scaled state: scales of |w|, signs kept aside
        const double fp = wp != 0.0 ? sqrt(fabs(wp)) : ldexp(1.0, -60);
        const double fv = wv != 0.0 ? sqrt(fabs(wv)) : ldexp(1.0, -60);
        a.fs_sp[i] = (float)fp;
        a.fs_sv[i] = (float)fv;
        a.fs_k1[i] = (float)(fp * (double)e->dt / fv);
        a.fs_k2[i] = (float)(fp * (double)e->B0);
        a.fs_k3[i] = (float)(fv * (double)e->dt);
        a.fs_cg[i] = (float)(fp * (double)e->dt * gv);
        a.fs_gps[i] = (float)(fp * gp);
        a.fs_gvs[i] = (float)(fv * gv);
        a.fs_sgp[i] = wp < 0.0 ? -1.0f : 1.0f;
        a.fs_sgv[i] = wv < 0.0 ? -1.0f : 1.0f;
    }
    a.dt = e->dt;
    a.B0 = e->B0;
    a.lambda = e->lambda;
    a.inv_lambda = 1 / e->lambda;
    for (int i = 0; i < 8; ++i) { a.goal[i] = e->goal[i]; a.w[i] = e->w[i]; }
    for (int i = 0; i < 4; ++i) { a.sigma[i] = e->sigma[i]; a.inv_s[i] = e->inv_s[i]; }
    for (int i = 0; i < 8; ++i) a.x0[i] = e->x0[i];
}

// a start/stop event pair for the next launch, or an empty timing when this solve is not sampled
int prof_pair(mppi_engine_t* e, mppi::LaunchTiming& tm, int which)
{
    tm = mppi::LaunchTiming();
    if (!e->prof_now) return MPPI_OK;
    std::vector<hipEvent_t>& ev = e->ev[which];
    size_t& used = e->ev_used[which];
    if (used + 2 > ev.size()) {
        if (ev.size() >= 2 * 8192) return MPPI_OK;   // stop recording, keep running
        for (int i = 0; i < 2; ++i) {
            hipEvent_t ne;
            HIPCHK(hipEventCreate(&ne));
            ev.push_back(ne);
        }
    }
    tm.start = ev[used++];
    tm.stop = ev[used++];
    return MPPI_OK;
}

unsigned int next_act_tag(mppi_engine_t* e)
{
    e->act_seq += 1;
    if (e->act_seq == 0) e->act_seq = 1;       // 0 is the tag of the zero-initialised words
    return e->act_seq;
}

// mode: 0 rank partial -> partial_out, 1 final, 2 rank partial -> peer exchange (sequence number
// xseq) -> final
void fill_own_combine(mppi_engine_t* e, mppi::CombineArgs& ca, unsigned long long idx,
                      unsigned int tag, int mode = 1, unsigned long long xseq = 0,
                      float* partial_out = nullptr)
{
    memset(&ca, 0, sizeof ca);
    ca.dev = e->d_state;
    ca.m = e->d_pm; ca.s = e->d_ps; ca.N = e->d_pN;
    ca.m_stride = 1; ca.s_stride = 1; ca.N_stride = e->NBT * 4;
    ca.n_parts = e->grid;
    ca.TA = e->TA;
    ca.A = e->A;
    ca.inv_lambda = 1 / e->lambda;
    ca.U = e->d_U;
    ca.tag = tag;
    ca.act_dev = e->d_act;
    ca.act_host = e->h_act_dev;
    if (mode != 0) ca.act_tag = next_act_tag(e);
    ca.slab = e->d_slab;
    ca.slab_tag = e->d_slab_tag;
    ca.tickets = e->d_tickets;
    ca.solve_idx = idx;
    ca.final_mode = mode;
    ca.partial_out = partial_out;
    ca.x.timeout_ticks = 200000000ull;         // 2 s: bound on the split meeting's polls
    ca.x.err_dev = e->d_err;
    ca.x.err_host = e->h_err_dev;
    if (mode == 2) {
        ca.x.peers = e->d_xg_peers;
        ca.x.G = e->xg_world;
        ca.x.rank = e->xg_rank;
        ca.x.W = e->xg_W;
        ca.x.timeout_ticks = (unsigned long long)(e->xg_timeout_s * 1e8);   // 100 MHz clock
        ca.x.parity = (int)(xseq & 1ull);
        ca.x.tag = (unsigned int)(xseq % 0xFFFFFFFFull) + 1u;
    }
    ca.row_splits = e->tune_combine_splits;
    ca.clamp = e->clamp ? 1 : 0;
    for (int i = 0; i < 4; ++i) ca.max_a[i] = e->max_a[i];
#ifdef MPPI_TRACE
    ca.trace = mppi::g_mppi_trace_buf;
#endif
    (void)mppi::combine_small_prepare(ca);
}

// launch the pending combine on its own (nothing to ride with)
int flush_pending(mppi_engine_t* e)
{
    if (!e->pending) return MPPI_OK;
    mppi::CombineArgs ca;
    e->u_epoch += 1;
    if (e->u_epoch == 0) e->u_epoch = 1;       // 0 is the tag of the zero-initialised buffer
    fill_own_combine(e, ca, e->pending_idx, e->u_epoch, e->pending_mode, e->pending_xseq);
    mppi::LaunchTiming tm;
    int rc = prof_pair(e, tm, 1);
    if (rc) return rc;
    e->pending = false;
#ifdef MPPI_TRACE
    ca.trace = nullptr;      // the analysis looks at the riding role only
#endif
    HIPCHK(mppi::launch_combine_small(ca, e->pending_stream, tm));
    e->n_combine_launches += 1;
    return MPPI_OK;
}

// the newest action, from the pinned words the combine kernels write
void read_action(const mppi_engine_t* e, float* next_act)
{
    for (int i = 0; i < e->A; ++i) {
        const unsigned int bits = (unsigned int)__atomic_load_n(&e->h_act[i], __ATOMIC_ACQUIRE);
        memcpy(&next_act[i], &bits, sizeof(float));
    }
}

// Report what a device-side time-out left in the watchdog word.  The fault is STICKY: the block
// that gave up published nothing, so the controls on the device are those of the last complete
// solve at best; every call that would solve or hand out results fails with MPPI_ESTATE until
// mppi_set_data starts over (it re-uploads the controls and clears the word).
int check_watchdog(mppi_engine_t* e)
{
    if (!e->fault && e->h_err && *e->h_err) {
        e->fault = *e->h_err;
        // Degrade instead of stalling again: a block of a launch waited in vain for another block
        // (a GPU shared with other work, a peer that never arrived).  From here on -- also after
        // mppi_set_data has cleared the fault -- every solve launches its rollout and its combine
        // on their own (pipeline mode 1: no block waits for a block of its own launch) until the
        // caller asks for mode 0 again with mppi_set_pipeline.
        if (e->defer) {
            e->defer = 0;
            e->degraded = true;
        }
    }
    if (!e->fault) return MPPI_OK;
    const int code = e->fault;
    const char* note = e->degraded ? "; the engine has switched to pipeline mode 1 (stand-alone "
                                     "launches) by itself" : "";
    if (code == 1)
        return fail(MPPI_ESTATE, "peer exchange timed out after %.1f s: a rank did not reach "
                    "solve %llu (call mppi_set_data to start over)%s", e->xg_timeout_s,
                    e->solve_idx, note);
    return fail(MPPI_ESTATE, "device watchdog %d: a block gave up waiting for the combine that "
                "rides in its own launch (call mppi_set_data to start over)%s", code, note);
}

// forget a reported fault: everything enqueued has drained (the caller settled), so no kernel can
// set the word again
int clear_watchdog(mppi_engine_t* e)
{
    e->fault = 0;
    if (e->h_err) *e->h_err = 0;
    if (e->d_err) HIPCHK(hipMemset(e->d_err, 0, sizeof(int)));
    return MPPI_OK;
}

// everything enqueued by this engine has run, nothing is pending
int settle(mppi_engine_t* e)
{
    int rc = flush_pending(e);
    if (rc) return rc;
    if (e->last_stream && e->last_stream != e->stream) HIPCHK(hipStreamSynchronize(e->last_stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MPPI_OK;
}

// sampling / rollout / per-block reduction; carry = let a pending combine ride in this launch
int enqueue_rollout(mppi_engine_t* e, hipStream_t st, bool carry = false)
{
    if (!e->data_set) return fail(MPPI_ESTATE, "solve before mppi_set_data");
    int rc = ensure_geometry(e);
    if (rc) return rc;
    if (e->pending) {
        // Riding pays while the launch is short: the riding variant of the kernel is a few per
        // cent slower per tile (measured 4 %) and the combine blocks take slots from the first
        // rollout blocks, against the ~8 us of a stand-alone combine launch.  With more than two
        // tiles per block the pending combine is launched on its own instead (same kernel code,
        // same bits).
        const bool short_launch =
            (long long)e->n_tileblk <= (long long)e->tune_ride_max_tiles * e->grid;
        // ... and only where EVERY block of the riding launch (rollout + combine role) holds a slot
        // at once, by the occupancy API of the riding kernel itself: the rollout blocks wait for
        // the combine blocks inside the launch, and with all of them resident no dispatch order
        // can leave a waited-for block without a slot
        bool co_resident = false;
        if (carry && short_launch && e->resident_ride > 0) {
            mppi::CombineArgs probe;
            memset(&probe, 0, sizeof probe);
            probe.TA = e->TA;
            probe.n_parts = e->grid;
            probe.row_splits = e->tune_combine_splits;
            // (an exchange that rides waits for the peers' words: the launches of all the ranks
            //  that share this GPU must then fit the chip TOGETHER)
            const long long share = e->pending_mode == 2 ? e->xg_ranks_on_my_device : 1;
            co_resident = share * ((long long)e->grid + mppi::combine_small_prepare(probe))
                          <= e->resident_ride;
        }
        // The PACKED kernel runs its first tile as a copy of its own (rollout_packed_impl.hpp): what
        // riding adds -- polling for the controls -- is outside its steady-state loop, so a combine
        // rides in a packed launch of ANY length (C3: 71 -> 68 us per solve).  A launch of more
        // blocks than the chip holds leans on index-order dispatch alone (the
        // combine-role blocks come first and hold their slots before a rollout block can wait for
        // them); the watchdog bounds the wait if that ever fails, and the engine then falls back to
        // stand-alone launches by itself (check_watchdog).
        // (an EXCHANGE that rides makes the rollout blocks wait for the peers' words too: with a
        //  peer on this very GPU -- rehearsals, GPU sharing -- a launch that fills the chip would
        //  keep the peer's launch, which has to send them, from ever getting a slot)
        const bool long_ok = e->tune_ride_long != 0 &&
                             !(e->pending_mode == 2 && e->xg_peer_on_my_device);
        const bool ride_ok = e->packed ? (long_ok || (short_launch && co_resident))
                                       : (short_launch && co_resident);
        if (!carry || e->strict || e->pending_stream != st || !ride_ok) {
            const hipStream_t was = e->pending_stream;
            if ((rc = flush_pending(e))) return rc;
            // a solve that moves to another stream must still see the controls of the last one
            if (was != st) HIPCHK(hipStreamSynchronize(was));
        }
    }
    e->last_stream = st;
    const mppi::ELayout lay = {e->packed ? 1 : 0, e->C, e->nq, e->ng, e->NGT, e->TPW};
    if (e->injected && e->inj_dirty) {
        HIPCHK(mppi::launch_import_noise(e->A, e->d_Einj, e->d_Eint, e->K, e->T, lay, st));
        e->inj_dirty = false;
    }
    float* Ecur = e->d_Eint;
    // a valid prefetch of THIS solve's noise (same stream of counters, same layout): load it
    bool use_pf = e->pf_valid && !e->injected && !e->strict && e->d_Epf && e->pf_idx == e->solve_idx &&
                  e->pf_seed == e->seed && memcmp(&e->pf_lay, &lay, sizeof lay) == 0 &&
                  e->epf_floats == e->eint_floats;
    for (int i = 0; i < e->A; ++i) use_pf = use_pf && e->pf_sigma[i] == e->sigma[i];
    e->pf_valid = false;
    if (use_pf) {
        // (the prefetch ran on the stream of the last blocking call: the same one, normally)
        if (e->pf_on_stream && e->pf_on_stream != st) HIPCHK(hipStreamSynchronize(e->pf_on_stream));
        Ecur = e->d_Epf;
    }

    mppi::RolloutArgs ra;
    memset(&ra, 0, sizeof ra);
    fill_rollout_args(e, ra);
    ra.Eint = Ecur;
    ra.dev_copy = e->d_args;
    if (!e->args_valid) {
        // The device copy holds what does not change from solve to solve; every setter that touches
        // one of its fields clears args_valid (solve index, noise pointer and x0 travel by value).
        mppi::RolloutArgs cmp = ra;
        cmp.solve_idx = 0;
        cmp.Eint = nullptr;
        for (int i = 0; i < 8; ++i) cmp.x0[i] = 0.f;
        HIPCHK(hipMemcpyAsync(e->d_args, &cmp, sizeof cmp, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        e->h_args_last = cmp;
        e->args_valid = true;
    }
    mppi::DeferredCombine dc;
    memset(&dc, 0, sizeof dc);
    if (e->pending) {       // same stream, fused kernel: the pending combine rides in this launch
        e->u_epoch += 1;
        if (e->u_epoch == 0) e->u_epoch = 1;
        fill_own_combine(e, dc.c, e->pending_idx, e->u_epoch, e->pending_mode, e->pending_xseq);
        dc.n_blocks = dc.c.n_cols * dc.c.RS;
        e->pending = false;
        e->n_riding_launches += 1;
    }
    e->n_rollout_launches += 1;
    {   // every n-th solve AND its successor: a stamp is clean only behind a stamped predecessor.
        // The pair sits a quarter of the way into each period, not at its head: a stamped launch
        // costs ~7 us of host time, and the first launches after a wait are the ones the device
        // is waiting for (a 20-solve run read 0.5 us per solve slow with its pair in front).
        const unsigned long long per = (unsigned long long)(e->prof > 0 ? e->prof : 1);
        const unsigned long long off = per >= 8 ? per / 4 : 0;
        const unsigned long long ph = e->prof > 0 ? e->prof_count++ % per : 2;
        e->prof_now = e->prof > 0 && (ph == off || (ph == off + 1 && e->prof > 2));
    }
    mppi::LaunchTiming tm;
    if ((rc = prof_pair(e, tm, 0))) return rc;
    if (tm.start) e->ev_clean.push_back(e->prof_prev ? 1 : 0);
    e->prof_prev = tm.start != nullptr;
    const bool sample_in_kernel = !e->injected && !use_pf;
    if (e->strict)
        HIPCHK(mppi::launch_rollout_stream(e->A, sample_in_kernel, e->grid, ra, st, tm));
    else if (e->packed)
        HIPCHK(mppi::launch_rollout_packed(e->A, e->ng, sample_in_kernel, e->grid, ra, dc, st, tm));
    else
        HIPCHK(mppi::launch_rollout_fused(e->A, e->NGt, sample_in_kernel, e->grid, ra, dc, st, tm));
    if (use_pf) {           // the buffers change roles: what this solve used is now "the" noise
        float* t = e->d_Eint;
        e->d_Eint = e->d_Epf;
        e->d_Epf = t;
        e->n_pf_used += 1;
    }
    e->last_E = Ecur;
    e->last_lay = lay;
    e->last_idx = e->solve_idx;
    e->last_stored = e->injected || e->store_noise || e->strict || use_pf;
    e->last_injected = e->injected;
    e->last_seed = e->seed;
    for (int i = 0; i < 4; ++i) e->last_sigma[i] = e->sigma[i];
    memcpy(e->x0_last, e->x0, sizeof e->x0);
    return MPPI_OK;
}

int enqueue_combine(mppi_engine_t* e, hipStream_t st, const float* m, const float* s,
                    const float* N, long long ms, long long ss, long long Ns, int n_parts,
                    int mode, float* partial_out)
{
    if (n_parts < 1 || n_parts > mppi::kMaxParts)
        return fail(MPPI_EINVAL, "n_parts %d out of range", n_parts);
    mppi::CombineArgs ca;
    memset(&ca, 0, sizeof ca);                 // (slab_tag = null: ticketed row splits)
    ca.dev = e->d_state;
    ca.m = m; ca.s = s; ca.N = N;
    ca.m_stride = ms; ca.s_stride = ss; ca.N_stride = Ns;
    ca.n_parts = n_parts;
    ca.TA = e->TA;
    ca.A = e->A;
    ca.inv_lambda = 1 / e->lambda;
    ca.U = e->d_U;
    ca.act_dev = e->d_act;
    ca.act_host = e->h_act_dev;
    if (mode != 0) ca.act_tag = next_act_tag(e);
    ca.partial_out = partial_out;
    ca.slab = e->d_slab;
    ca.tickets = e->d_tickets;
    ca.solve_idx = e->solve_idx;
    ca.final_mode = mode;
    memset(&ca.x, 0, sizeof ca.x);
    if (mode == 2) {
        if (!e->xg_connected) return fail(MPPI_ESTATE, "mppi_xchg_connect has not been called");
        ca.x.peers = e->d_xg_peers;
        ca.x.G = e->xg_world;
        ca.x.rank = e->xg_rank;
        ca.x.W = e->xg_W;
        ca.x.parity = (int)(e->xg_seq & 1ull);
        ca.x.tag = (unsigned int)(e->xg_seq % 0xFFFFFFFFull) + 1u;
        ca.x.timeout_ticks = (unsigned long long)(e->xg_timeout_s * 1e8);   // 100 MHz clock
        ca.x.err_dev = e->d_err;
        ca.x.err_host = e->h_err_dev;
    }
    ca.row_splits = e->tune_combine_splits;
    ca.clamp = e->clamp ? 1 : 0;
    for (int i = 0; i < 4; ++i) ca.max_a[i] = e->max_a[i];
    mppi::LaunchTiming tm;
    {
        int rc = prof_pair(e, tm, 1);
        if (rc) return rc;
    }
    HIPCHK(mppi::launch_combine(ca, st, tm));
    return MPPI_OK;
}

// Should the combine launch of the blocking call that is about to wait also draw the NEXT solve's
// noise?  Auto mode: always for launches of one tile per block, and for longer (VALU-bound)
// launches only when the host's think time between two calls has been long enough to hide it.
// (Drawing only the HEAD of a long launch's tiles -- what the idle chip has time for between two
//  calls back to back -- and letting the sampling kernel load those and draw the rest was built
//  and measured: the branch costs the sampling kernel 5 % per tile even when nothing was drawn
//  ahead, and a blocking C3 call got slower, 91.5 against 79.5 us: DESIGN 2.5.)
bool want_prefetch(const mppi_engine_t* e)
{
    if (e->pf_mode == 0 || e->injected || e->strict || !e->store_noise || !e->geom_ok || !e->data_set)
        return false;
    if (e->fault || !e->pending || e->pending_mode == 0) return false;
    if (4.0 * (double)e->eint_floats > 1.5e9) return false;     // (a second buffer of that size: no)
    if (e->pf_mode == 1 && e->grid != e->n_tileblk) {
        // A launch of one tile per block is a latency chain that leaves the chip mostly idle: its
        // draw (write-through stores: no write-back at the end of the launch) is over about when
        // the combine is, and pays even with calls back to back (C2: 20.5 against 20.9-22.0 us).
        // A longer launch is VALU-bound: the Philox + Box-Muller pass alone is 173 ns per
        // wave-block and SIMD, the next rollout is stream-ordered behind it, and it
        // pays only when the host's think time hides it.
        const double pf_us = (double)e->K * e->NBT / 64.0 * 0.173 / 1024.0;
        if (e->think_ema_us < 1.5 * pf_us + 20.0) return false;
    }
    return true;
}

// launch the pending combine on its own, with the next solve's noise drawn behind it
int flush_pending_with_prefetch(mppi_engine_t* e)
{
    if (!e->pending) return MPPI_OK;
    // Could this engine prefetch at all (whatever the timing rule says right now)?  Then the FIRST
    // blocking call -- a control loop's set-up call -- pays the one-off costs: the second noise
    // buffer (a 240 MB hipMalloc at C3: milliseconds) and the first launch of the fused kernel
    // (its code is loaded on first use), so that no later call of the loop stalls on them
    // (found with apps/mppi_closed_loop --rate-hz 100: one re-plan of 6.7 ms among 0.07 ms ones).
    const bool could = e->pf_mode != 0 && !e->injected && !e->strict && e->store_noise && e->geom_ok &&
                       e->data_set && !e->fault && e->pending_mode != 0 &&
                       4.0 * (double)e->eint_floats <= 1.5e9;
    if (!could) return flush_pending(e);
    const bool want = want_prefetch(e);
    if (!want && e->pf_warm && e->epf_floats == e->eint_floats) return flush_pending(e);
    const hipStream_t st = e->pending_stream;
    if (e->epf_floats != e->eint_floats) {
        if (e->pf_on_stream) HIPCHK(hipStreamSynchronize(e->pf_on_stream));
        if (e->d_Epf) HIPCHK(hipFree(e->d_Epf));
        e->d_Epf = nullptr;
        e->epf_floats = 0;
        HIPCHK(hipMalloc(&e->d_Epf, e->eint_floats * sizeof(float)));
        HIPCHK(hipMemsetAsync(e->d_Epf, 0, e->eint_floats * sizeof(float), st));
        e->epf_floats = e->eint_floats;
    }
    mppi::CombineArgs ca;
    e->u_epoch += 1;
    if (e->u_epoch == 0) e->u_epoch = 1;
    fill_own_combine(e, ca, e->pending_idx, e->u_epoch, e->pending_mode, e->pending_xseq);
    mppi::LaunchTiming tm;
    int rc = prof_pair(e, tm, 1);
    if (rc) return rc;
    e->pending = false;
#ifdef MPPI_TRACE
    ca.trace = nullptr;
#endif
    const mppi::ELayout lay = {e->packed ? 1 : 0, e->C, e->nq, e->ng, e->NGT, e->TPW};
    // (not wanted now: the same launch with no tile to draw -- the combine blocks alone, same bits)
    HIPCHK(mppi::launch_combine_small_prefetch(e->A, ca, e->d_Epf, lay, e->K, e->T,
                                               want ? (long long)e->n_tileblk * 4 : 0, e->seed,
                                               e->solve_idx, e->k_offset, e->sigma, st, tm));
    e->n_combine_launches += 1;
    e->pf_warm = true;
    if (!want) return MPPI_OK;
    e->pf_valid = true;
    e->pf_idx = e->solve_idx;
    e->pf_seed = e->seed;
    e->pf_lay = lay;
    e->pf_on_stream = st;
    for (int i = 0; i < 4; ++i) e->pf_sigma[i] = e->sigma[i];
    e->n_pf_launched += 1;
    return MPPI_OK;
}

int create_common(int K, long long k_offset, bool sharded, int T, float dt, int S, int A,
                  int verbose, mppi_engine** out)
{
    if (!out) return fail(MPPI_EINVAL, "out is null");
    *out = nullptr;
    if (K < 1 || T < 1) return fail(MPPI_EINVAL, "nb_sim and steps must be >= 1");
    if (A < 1 || A > 4) return fail(MPPI_EINVAL, "act_dim %d unsupported (1..4)", A);
    if (S != 2 * A) return fail(MPPI_EINVAL, "state_dim must be 2*act_dim (got %d, %d)", S, A);
    if (!(dt > 0.f)) return fail(MPPI_EINVAL, "dt must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MPPI_ENODEV, "no HIP device: the engine has no CPU fallback");
    {   // The library's code object must load on this device and hold the kernels this engine will
        // launch: asked for here, a mismatch (wrong --offload-arch, a partly rebuilt library) is an
        // error code; found at the first launch it is an abort() inside the HIP runtime.
        const hipError_t pe = mppi::probe_code_object(A);
        if (pe != hipSuccess) {
            (void)hipGetLastError();
            return fail(MPPI_ENODEV, "the gfx950 kernels of this library cannot be loaded on the "
                        "current device: HIP error %d (%s)", (int)pe, hipGetErrorString(pe));
        }
    }

    mppi_engine* e = new mppi_engine();
    if (const char* env = getenv("MPPI_COMBINE_SPLITS")) e->tune_combine_splits = atoi(env);
    if (const char* env = getenv("MPPI_RIDE_MAX_TILES")) e->tune_ride_max_tiles = atoi(env);
    if (const char* env = getenv("MPPI_RIDE_LONG")) e->tune_ride_long = atoi(env);
    if (const char* env = getenv("MPPI_STORE_MODE")) e->tune_store_mode = atoi(env);
    if (const char* env = getenv("MPPI_PREFETCH")) e->pf_mode = atoi(env);
    if (const char* env = getenv("MPPI_NT_RESIDENT_MB")) {
        e->tune_nt_resident_mb = atoi(env);
        e->tune_nt_resident_set = true;
    }
    e->K = K; e->T = T; e->S = S; e->A = A; e->TA = T * A;
    e->SG = mppi::rollout_group_steps(A);
    e->BPG = mppi::rollout_group_blocks(A);
    e->NGT = (T + e->SG - 1) / e->SG;
    e->NBT = (e->TA + 3) / 4;
    e->k_offset = k_offset;
    e->sharded = sharded;
    e->dt = dt;
    {
        const float dd = dt * dt;                 // reference src/point_mass.cu:46
        e->B0 = (float)((double)dd / 2.0);
    }
    e->verbose = verbose;
    *out = e;   // so that a failing HIPCHK below still lets the caller destroy it

    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&e->d_state, sizeof(mppi::DevState)));
    HIPCHK(hipMemset(e->d_state, 0, sizeof(mppi::DevState)));
    HIPCHK(hipMalloc(&e->d_U, 2 * (size_t)e->TA * sizeof(float)));
    HIPCHK(hipMemset(e->d_U, 0, 2 * (size_t)e->TA * sizeof(float)));
    HIPCHK(hipMalloc(&e->d_cost, (size_t)K * sizeof(float)));
    HIPCHK(hipMalloc(&e->d_act, 4 * sizeof(float)));
    HIPCHK(hipMalloc(&e->d_local_partial, (size_t)(e->TA + 2) * sizeof(float)));
    HIPCHK(hipMalloc(&e->d_args, sizeof(mppi::RolloutArgs)));
    HIPCHK(hipMalloc(&e->d_slab, (size_t)mppi::kMaxRowSplits * e->TA * sizeof(float)));
    {
        const size_t nt = (size_t)(e->TA + mppi::kCombineCols - 1) / mppi::kCombineCols;
        HIPCHK(hipMalloc(&e->d_tickets, nt * sizeof(unsigned int)));
        HIPCHK(hipMemset(e->d_tickets, 0, nt * sizeof(unsigned int)));
    }
    HIPCHK(hipMalloc(&e->d_err, sizeof(int)));
    HIPCHK(hipMemset(e->d_err, 0, sizeof(int)));
    HIPCHK(hipHostMalloc(&e->h_err, sizeof(int), hipHostMallocMapped));
    *e->h_err = 0;
    HIPCHK(hipHostGetDevicePointer((void**)&e->h_err_dev, e->h_err, 0));
    {
        const size_t words = (size_t)(mppi::kMaxSmallSplits + 8) * e->TA;   // (+ 8: MPPI_FIN_COPIES experiments)
        HIPCHK(hipMalloc(&e->d_slab_tag, words * sizeof(unsigned long long)));
        HIPCHK(hipMemset(e->d_slab_tag, 0, words * sizeof(unsigned long long)));
    }
    HIPCHK(hipHostMalloc(&e->h_act, 4 * sizeof(unsigned long long), hipHostMallocMapped));
    memset(e->h_act, 0, 4 * sizeof(unsigned long long));
    HIPCHK(hipHostGetDevicePointer((void**)&e->h_act_dev, e->h_act, 0));
    if (verbose)
        printf("mppi_gpu_amd: K=%d T=%d S=%d A=%d dt=%g (offset %lld)\n", K, T, S, A, dt,
               k_offset);
    return MPPI_OK;
}

int sync_all(mppi_engine_t* e)
{
    HIPCHK(hipStreamSynchronize(e->stream));
    return MPPI_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int mppi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mppi_last_error(void) { return g_last_error.c_str(); }
const char* mppi_version(void) { return "mppi_gpu_amd 0.1 (gfx950)"; }

int mppi_create(int nb_sim, int steps, float dt, int state_dim, int act_dim, int verbose,
                mppi_engine** out)
{
    return create_common(nb_sim, 0, false, steps, dt, state_dim, act_dim, verbose, out);
}

int mppi_create_shard(int nb_sim_local, long long k_offset, int steps, float dt, int state_dim,
                      int act_dim, int verbose, mppi_engine** out)
{
    if (k_offset < 0) return fail(MPPI_EINVAL, "k_offset < 0");
    return create_common(nb_sim_local, k_offset, true, steps, dt, state_dim, act_dim, verbose, out);
}

void mppi_destroy(mppi_engine* e)
{
    if (!e) return;
    e->pending = false;         // results nobody asked for
    if (e->last_stream && e->last_stream != e->stream) (void)hipStreamSynchronize(e->last_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    (void)mppi_xchg_close(e);
    (void)hipFree(e->d_xg_peers);
    (void)hipFree(e->d_err);
    if (e->h_err) (void)hipHostFree(e->h_err);
    (void)hipFree(e->d_slab_tag);
    for (auto& list : e->ev)
        for (hipEvent_t ev : list) (void)hipEventDestroy(ev);
    (void)hipFree(e->d_state);
    (void)hipFree(e->d_U);
    (void)hipFree(e->d_Eint);
    (void)hipFree(e->d_cost);
    (void)hipFree(e->d_pm);
    (void)hipFree(e->d_ps);
    (void)hipFree(e->d_pN);
    (void)hipFree(e->d_act);
    (void)hipFree(e->d_local_partial);
    (void)hipFree(e->d_args);
    (void)hipFree(e->d_slab);
    (void)hipFree(e->d_tickets);
    if (e->pf_on_stream) (void)hipStreamSynchronize(e->pf_on_stream);
    (void)hipFree(e->d_Epf);
    (void)hipFree(e->d_Einj);
    (void)hipFree(e->d_scratch);
    if (e->h_act) (void)hipHostFree(e->h_act);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int mppi_set_data(mppi_engine* e, const float* x0, const float* u, const float* goal,
                  const float* w)
{
    if (!e || !x0 || !u || !goal || !w) return fail(MPPI_EINVAL, "null argument");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
    {   // everything has drained: a reported device time-out ends here (fresh controls below)
        int rc_ = clear_watchdog(e);
        if (rc_) return rc_;
    }
    {   // the packed kernel needs weights >= 0: a change of their sign pattern re-plans the launch
        bool was = true, now = true;
        for (int i = 0; i < e->S; ++i) { was = was && e->w[i] >= 0.f; now = now && w[i] >= 0.f; }
        if (was != now) e->geom_ok = false;
    }
    for (int i = 0; i < e->S; ++i) { e->x0[i] = x0[i]; e->goal[i] = goal[i]; e->w[i] = w[i]; }
    e->args_valid = false;
    e->solve_idx = 0;   // the reference re-seeds its generators here (src/point_mass.cu:780)
    e->have_solve = false;
    HIPCHK(hipMemcpy(e->d_U, u, (size_t)e->TA * sizeof(float), hipMemcpyHostToDevice));
    e->data_set = true;
    return MPPI_OK;
}

int mppi_set_x(mppi_engine* e, const float* x0)
{
    if (!e || !x0) return fail(MPPI_EINVAL, "null argument");
    // Host-only: the state travels BY VALUE in the arguments of the next rollout launch, so solves
    // already enqueued keep the state they were enqueued with and nothing touches the GPU here
    // (the reference copies to the device and runs set_x_kernel over K rows, src/point_mass.cu:482-486)
    for (int i = 0; i < e->S; ++i) e->x0[i] = x0[i];
    return MPPI_OK;
}

int mppi_get_x(mppi_engine* e, float* x0)
{
    if (!e || !x0) return fail(MPPI_EINVAL, "null argument");
    for (int i = 0; i < e->S; ++i) x0[i] = e->x0[i];      // the engine is the only writer
    return MPPI_OK;
}

int mppi_solve_async(mppi_engine* e, void* stream)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (e->fault || (e->h_err && *e->h_err)) return check_watchdog(e);   // sticky until set_data
    if (e->t_return_valid) {     // the host's think time since the last blocking call returned
        const double gap = std::chrono::duration<double, std::micro>(
                               std::chrono::steady_clock::now() - e->t_return).count();
        e->think_ema_us = e->think_ema_us > 0.0 ? 0.5 * e->think_ema_us + 0.5 * gap : gap;
        e->t_return_valid = false;
    }
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    const bool defer = e->defer != 0;
    int rc = enqueue_rollout(e, st, defer);
    if (rc) return rc;
    if (defer && !e->strict) {
        // the combine is not launched yet: it rides at the front of the next solve's launch, or
        // is flushed by the first call that needs this solve's results
        e->pending = true;
        e->pending_mode = 1;
        e->pending_idx = e->solve_idx;
        e->pending_stream = st;
    } else {
        rc = enqueue_combine(e, st, e->d_pm, e->d_ps, e->d_pN, 1, 1, e->NBT * 4, e->grid, 1, nullptr);
        if (rc) return rc;
    }
    e->solve_idx += 1;
    e->have_solve = true;
    return MPPI_OK;
}

int mppi_flush_async(mppi_engine* e)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    return flush_pending(e);
}

int mppi_sync_act(mppi_engine* e, float* next_act)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    {
        int rc = settle(e);
        if (rc) return rc;
    }
    {
        int rc = check_watchdog(e);
        if (rc) return rc;
    }
    if (next_act) read_action(e, next_act);
    return MPPI_OK;
}

int mppi_get_act(mppi_engine* e, float* next_act)
{
    if (!e || !next_act) return fail(MPPI_EINVAL, "null argument");
    int rc = mppi_solve_async(e, nullptr);
    if (rc) return rc;
    return mppi_wait_act(e, next_act);
}

int mppi_wait_act(mppi_engine* e, float* next_act)
{
    if (!e || !next_act) return fail(MPPI_EINVAL, "null argument");
    int rc;
    // (with the next solve's noise drawn behind the combine, while the host waits / thinks)
    if ((rc = flush_pending_with_prefetch(e))) return rc;
    // The closed-loop call: poll the pinned words the combine kernel writes the action into
    // (8-byte {value, tag} stores) instead of sleeping in hipStreamSynchronize, whose wake-up
    // costs more than the solve at K = 1e4; after 300 us of polling fall back to the blocking
    // wait.  Later calls on the engine are stream-ordered behind the solve as always.
    const unsigned int want = e->act_seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (int i = 0; i < e->A; ++i) {
            const unsigned long long w = __atomic_load_n(&e->h_act[i], __ATOMIC_ACQUIRE);
            all = all && (unsigned int)(w >> 32) == want;
        }
        if (all) break;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
            if ((rc = settle(e))) return rc;
            break;
        }
    }
    if ((rc = check_watchdog(e))) return rc;
    read_action(e, next_act);
    e->t_return = std::chrono::steady_clock::now();
    e->t_return_valid = true;
    return MPPI_OK;
}

int mppi_get_u(mppi_engine* e, float* u)
{
    if (!e || !u) return fail(MPPI_EINVAL, "null argument");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
        if ((rc_ = check_watchdog(e))) return rc_;
    }
    HIPCHK(hipMemcpy(u, e->d_U + (e->solve_idx & 1ull) * e->TA, (size_t)e->TA * sizeof(float),
                     hipMemcpyDeviceToHost));
    return MPPI_OK;
}

int mppi_get_inf(mppi_engine* e, float* x_all, float* u, float* noise, float* cost, float* beta,
                 float* nabla, float* weight)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
        if ((rc_ = check_watchdog(e))) return rc_;
    }
    if ((x_all || noise || cost || beta || nabla || weight) && !e->have_solve)
        return fail(MPPI_ESTATE, "no solve has run since mppi_set_data");
    int rc;
    if (u && (rc = mppi_get_u(e, u))) return rc;
    if (cost)
        HIPCHK(hipMemcpy(cost, e->d_cost, (size_t)e->K * sizeof(float), hipMemcpyDeviceToHost));
    if (beta || nabla) {
        mppi::DevState hs;
        HIPCHK(hipMemcpy(&hs, e->d_state, sizeof hs, hipMemcpyDeviceToHost));
        if (beta) *beta = hs.beta;
        if (nabla) *nabla = hs.nabla;
    }
    if (weight) {
        if ((rc = ensure_scratch(e, (size_t)e->K))) return rc;
        HIPCHK(mppi::launch_weights(e->d_cost, e->d_state, e->lambda, e->d_scratch, e->K,
                                    e->stream));
        {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
        HIPCHK(hipMemcpy(weight, e->d_scratch, (size_t)e->K * sizeof(float),
                         hipMemcpyDeviceToHost));
    }
    if (noise) {
        const size_t n = (size_t)e->K * e->T * e->A;
        if ((rc = ensure_scratch(e, n))) return rc;
        if (e->last_stored)
            HIPCHK(mppi::launch_export_noise(e->A, e->last_E, e->d_scratch, e->K, e->T, e->last_lay,
                                             e->stream));
        else if (e->last_injected)      // (its tile buffer went with a geometry change: the caller's copy)
            HIPCHK(hipMemcpyAsync(e->d_scratch, e->d_Einj, n * sizeof(float), hipMemcpyDeviceToDevice,
                                  e->stream));
        else     // not materialised by the rollout: the same counters give the same bits again
            HIPCHK(mppi::launch_regen_noise(e->A, e->d_scratch, e->K, e->T, e->last_seed,
                                            e->last_idx, e->k_offset, e->last_sigma, e->stream));
        {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
        HIPCHK(hipMemcpy(noise, e->d_scratch, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (x_all) {
        const size_t n = (size_t)e->K * (e->T + 1) * e->S;
        const size_t ne = (e->last_stored || e->last_injected) ? 0 : (size_t)e->K * e->T * e->A;
        // controls and x0 the last rollout used: U buffer of parity last_idx, x0_last
        if ((rc = ensure_scratch(e, n + 8 + ne))) return rc;
        float* d_x0 = e->d_scratch + n;
        HIPCHK(hipMemcpy(d_x0, e->x0_last, 8 * sizeof(float), hipMemcpyHostToDevice));
        const float* Esrc = e->last_E;
        mppi::ELayout lay = e->last_lay;
        if (!e->last_stored && e->last_injected) {
            Esrc = e->d_Einj;
            lay = mppi::ELayout{2, 1, 0, 0, e->T, 0};
        } else if (!e->last_stored) {       // regenerate the noise next to the trace, in E[k][t][a] order
            float* d_e = e->d_scratch + n + 8;
            HIPCHK(mppi::launch_regen_noise(e->A, d_e, e->K, e->T, e->last_seed, e->last_idx,
                                            e->k_offset, e->last_sigma, e->stream));
            Esrc = d_e;
            lay = mppi::ELayout{2, 1, 0, 0, e->T, 0};
        }
        HIPCHK(mppi::launch_trace_states(e->A, Esrc, e->d_U + (e->last_idx & 1ull) * e->TA,
                                         d_x0, e->d_scratch, e->K, e->T, lay, e->dt, e->B0,
                                         e->stream));
        {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
        HIPCHK(hipMemcpy(x_all, e->d_scratch, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return MPPI_OK;
}

int mppi_get_data(mppi_engine* e, float* x_all, float* noise)
{
    return mppi_get_inf(e, x_all, nullptr, noise, nullptr, nullptr, nullptr, nullptr);
}

int mppi_set_params(mppi_engine* e, float lambda, const float* sigma, const float* inv_s)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (!(lambda > 0.f)) return fail(MPPI_EINVAL, "lambda must be positive");
    {   // a pending combine belongs to the old lambda
        int rc_ = flush_pending(e);
        if (rc_) return rc_;
    }
    e->lambda = lambda;
    if (sigma)
        for (int i = 0; i < e->A; ++i) e->sigma[i] = sigma[i];
    if (inv_s) for (int i = 0; i < e->A; ++i) e->inv_s[i] = inv_s[i];
    e->args_valid = false;
    return MPPI_OK;
}

int mppi_set_seed(mppi_engine* e, unsigned long long seed)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    e->seed = seed;
    e->args_valid = false;
    return MPPI_OK;
}

int mppi_set_noise(mppi_engine* e, const float* noise)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
    // The sampling and the injected-noise instantiations of a kernel differ in registers, hence in
    // the blocks a CU holds: the persistent grid and -- what the riding combine's co-residency test
    // leans on -- `resident_ride` belong to ONE of them.  A change of mode re-plans the launch.
    const bool was = e->injected;
    if (!noise) {
        e->injected = false;
        if (was) e->geom_ok = false;
        return MPPI_OK;
    }
    const size_t n = (size_t)e->K * e->T * e->A;
    if (!e->d_Einj) HIPCHK(hipMalloc(&e->d_Einj, n * sizeof(float)));
    HIPCHK(hipMemcpy(e->d_Einj, noise, n * sizeof(float), hipMemcpyHostToDevice));
    e->injected = true;
    e->inj_dirty = true;
    if (!was) e->geom_ok = false;
    return MPPI_OK;
}

int mppi_set_noise_store(mppi_engine* e, int on)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    {
        int rc_ = flush_pending(e);
        if (rc_) return rc_;
    }
    e->store_noise = on != 0;
    e->args_valid = false;
    return MPPI_OK;
}

int mppi_set_ref_compat(mppi_engine* e, int on)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (on && e->A == 1 && e->K >= 262144)
        return fail(MPPI_EINVAL, "ref_compat for act_dim 1 covers nb_sim < 262144 (beyond that the "
                    "reference's in-place multi-block passes race, SURVEY App. B.1)");
    if (on && e->A == 4)
        return fail(MPPI_EINVAL, "ref_compat: the reference has no 4-D system to be compatible with");
    if (on && e->sharded) return fail(MPPI_EINVAL, "ref_compat is single-GPU only");
    {
        int rc_ = flush_pending(e);
        if (rc_) return rc_;
    }
    e->ref_compat = on != 0;
    e->args_valid = false;
    return MPPI_OK;
}

int mppi_set_action_limit(mppi_engine* e, const float* max_a)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (max_a)
        for (int i = 0; i < e->A; ++i)
            if (!(max_a[i] >= 0.f)) return fail(MPPI_EINVAL, "max_a[%d] must be >= 0", i);
    {   // a pending combine belongs to the old setting
        int rc_ = flush_pending(e);
        if (rc_) return rc_;
    }
    e->clamp = max_a != nullptr;
    for (int i = 0; i < e->A; ++i) e->max_a[i] = max_a ? max_a[i] : 0.f;
    return MPPI_OK;
}

int mppi_set_tuning(mppi_engine* e, int chunks, int strict, int max_blocks)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (chunks < 0 || max_blocks < 0) return fail(MPPI_EINVAL, "negative tuning value");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
    e->user_chunks = chunks;
    e->user_strict = strict;
    e->user_max_blocks = max_blocks;
    e->geom_ok = false;
    int rc = ensure_geometry(e);
    if (rc) e->geom_ok = false;
    return rc;
}

int mppi_set_packing(mppi_engine* e, int groups_per_lane)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (groups_per_lane < -1) return fail(MPPI_EINVAL, "groups_per_lane must be >= -1");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
    const int was = e->user_packing;
    e->user_packing = groups_per_lane;
    e->geom_ok = false;
    int rc = ensure_geometry(e);
    if (rc) {
        e->user_packing = was;
        e->geom_ok = false;
    }
    return rc;
}

int mppi_get_layout(mppi_engine* e, int out[4])
{
    if (!e || !out) return fail(MPPI_EINVAL, "null argument");
    int rc = ensure_geometry(e);
    if (rc) return rc;
    out[0] = e->packed ? 1 : 0;
    out[1] = e->ng;
    out[2] = e->packed ? e->TPW : 64 / e->C;
    out[3] = e->n_tileblk;
    return MPPI_OK;
}

int mppi_set_pipeline(mppi_engine* e, int on)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    {
        int rc_ = settle(e);
        if (rc_) return rc_;
    }
    if (on < 0 || on > 1) return fail(MPPI_EINVAL, "pipeline mode must be 0 or 1");
    e->defer = on == 0 ? 1 : 0;
    e->degraded = false;        // an explicit choice overrides the watchdog's
    return MPPI_OK;
}

int mppi_set_noise_prefetch(mppi_engine* e, int mode)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (mode < 0 || mode > 2) return fail(MPPI_EINVAL, "prefetch mode must be 0, 1 or 2");
    e->pf_mode = mode;
    if (mode == 0) e->pf_valid = false;
    return MPPI_OK;
}

int mppi_get_prefetch_counts(mppi_engine* e, long long out[2])
{
    if (!e || !out) return fail(MPPI_EINVAL, "null argument");
    out[0] = e->n_pf_launched;
    out[1] = e->n_pf_used;
    return MPPI_OK;
}

int mppi_get_pipeline(mppi_engine* e, int* mode, int* degraded)
{
    if (!e || !mode) return fail(MPPI_EINVAL, "null argument");
    *mode = e->defer ? 0 : 1;
    if (degraded) *degraded = e->degraded ? 1 : 0;
    return MPPI_OK;
}

int mppi_partial_len(const mppi_engine* e) { return e ? e->TA + 2 : 0; }

int mppi_solve_local_async(mppi_engine* e, float* d_partial, void* stream)
{
    if (!e || !d_partial) return fail(MPPI_EINVAL, "null argument");
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    int rc = enqueue_rollout(e, st);      // (flushes a pending combine first)
    if (rc) return rc;
    if (e->strict)
        return enqueue_combine(e, st, e->d_pm, e->d_ps, e->d_pN, 1, 1, e->NBT * 4, e->grid, 0, d_partial);
    // the same 256-thread combine the direct exchange runs, so that both transports add this
    // rank's partial in the same order (equal bits)
    mppi::CombineArgs ca;
    e->u_epoch += 1;
    if (e->u_epoch == 0) e->u_epoch = 1;
    fill_own_combine(e, ca, e->solve_idx, e->u_epoch, 0, 0, d_partial);
    mppi::LaunchTiming tm;
    if ((rc = prof_pair(e, tm, 1))) return rc;
    HIPCHK(mppi::launch_combine_small(ca, st, tm));
    return MPPI_OK;
}

int mppi_solve_finish_async(mppi_engine* e, const float* d_gathered, int n_parts, void* stream)
{
    if (!e || !d_gathered) return fail(MPPI_EINVAL, "null argument");
    if (n_parts < 1 || n_parts > mppi::kMaxRanks)
        return fail(MPPI_EINVAL, "n_parts %d out of range (1..%d)", n_parts, mppi::kMaxRanks);
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    e->prof_now = false;   // kernel_ms() reports the rollout and the rank-local combine only
    mppi::CombineArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.dev = e->d_state;
    ca.TA = e->TA;
    ca.A = e->A;
    ca.inv_lambda = 1 / e->lambda;
    ca.U = e->d_U;
    ca.act_dev = e->d_act;
    ca.act_host = e->h_act_dev;
    ca.act_tag = next_act_tag(e);
    ca.solve_idx = e->solve_idx;
    ca.final_mode = 1;
    ca.clamp = e->clamp ? 1 : 0;
    for (int i = 0; i < 4; ++i) ca.max_a[i] = e->max_a[i];
    HIPCHK(mppi::launch_finish_gathered(ca, d_gathered, n_parts, st));
    e->last_stream = st;
    e->solve_idx += 1;
    e->have_solve = true;
    return MPPI_OK;
}

// ---- direct peer exchange ------------------------------------------------------------------
int mppi_xchg_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int mppi_xchg_open(mppi_engine* e, int rank, int world, void* handle_out, void** inbox_out)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (world < 1 || world > mppi::kMaxRanks || rank < 0 || rank >= world)
        return fail(MPPI_EINVAL, "rank %d / world %d out of range (world <= %d)", rank, world,
                    mppi::kMaxRanks);
    if (e->xg_inbox) return fail(MPPI_ESTATE, "exchange already open");
    e->xg_rank = rank;
    e->xg_world = world;
    e->xg_W = ((e->TA + 2 + 15) / 16) * 16;
    const size_t bytes = 2ull * world * e->xg_W * sizeof(unsigned long long);
    // uncached device memory: peer stores land in HBM and local polls read HBM (what RCCL uses
    // for its flag/LL buffers); fine-grained as the fallback
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    }
    e->xg_inbox = (unsigned long long*)p;
    HIPCHK(hipMemset(p, 0, bytes));
    HIPCHK(hipDeviceSynchronize());
    if (!e->d_xg_peers) HIPCHK(hipMalloc(&e->d_xg_peers, mppi::kMaxRanks * sizeof(void*)));
    if (handle_out) {
        hipIpcMemHandle_t h;
        HIPCHK(hipIpcGetMemHandle(&h, p));
        memcpy(handle_out, &h, sizeof h);
    }
    if (inbox_out) *inbox_out = p;
    return MPPI_OK;
}

int mppi_xchg_connect(mppi_engine* e, const void* handles, void* const* same_process)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (!e->xg_inbox) return fail(MPPI_ESTATE, "mppi_xchg_open first");
    if (e->xg_connected) return fail(MPPI_ESTATE, "exchange already connected");
    if (!handles && !same_process) return fail(MPPI_EINVAL, "no handles and no pointers");
    std::vector<unsigned long long*> tab(mppi::kMaxRanks, nullptr);
    for (int g = 0; g < e->xg_world; ++g) {
        if (g == e->xg_rank) {
            tab[g] = e->xg_inbox;
        } else if (same_process && same_process[g]) {
            tab[g] = (unsigned long long*)same_process[g];
        } else if (handles) {
            hipIpcMemHandle_t h;
            memcpy(&h, (const char*)handles + (size_t)g * sizeof h, sizeof h);
            void* p = nullptr;
            HIPCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            e->xg_opened.push_back(p);
            tab[g] = (unsigned long long*)p;
        } else {
            return fail(MPPI_EINVAL, "no inbox for rank %d", g);
        }
    }
    HIPCHK(hipMemcpy(e->d_xg_peers, tab.data(), mppi::kMaxRanks * sizeof(void*),
                     hipMemcpyHostToDevice));
    {   // does a peer live on this very GPU?  (several shard engines or processes sharing a device:
        // their riding launches wait for each other's words and must then fit the chip TOGETHER)
        int my_dev = 0;
        (void)hipGetDevice(&my_dev);
        e->xg_peer_on_my_device = false;
        e->xg_ranks_on_my_device = 1;
        for (int g = 0; g < e->xg_world; ++g) {
            if (g == e->xg_rank) continue;
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, tab[g]) == hipSuccess) {
                if (at.device == my_dev) {
                    e->xg_peer_on_my_device = true;
                    e->xg_ranks_on_my_device += 1;
                }
            } else {
                (void)hipGetLastError();
            }
        }
    }
    e->xg_connected = true;
    e->args_valid = false;      // the riding rollout's watchdog outwaits the exchange time-out
    return MPPI_OK;
}

int mppi_xchg_set_timeout(mppi_engine* e, double seconds)
{
    if (!e || !(seconds > 0.0) || seconds > 60.0) return fail(MPPI_EINVAL, "timeout in (0, 60] s");
    e->xg_timeout_s = seconds;
    e->args_valid = false;
    return MPPI_OK;
}

int mppi_xchg_close(mppi_engine* e)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    // A solve whose exchange is still held back is completed while the inboxes exist (its solve and
    // exchange counters have advanced already; the peers, having made the same calls, hold theirs);
    // a peer that is gone shows up as the exchange time-out, reported below.
    int rc = flush_pending(e);
    if (e->last_stream) (void)hipStreamSynchronize(e->last_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void* p : e->xg_opened) (void)hipIpcCloseMemHandle(p);
    e->xg_opened.clear();
    if (e->xg_inbox) (void)hipFree(e->xg_inbox);
    e->xg_inbox = nullptr;
    e->xg_connected = false;
    e->args_valid = false;
    if (rc) return rc;
    return check_watchdog(e);
}

int mppi_solve_exchange_async(mppi_engine* e, void* stream)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    if (!e->xg_connected) return fail(MPPI_ESTATE, "mppi_xchg_connect has not been called");
    if (e->fault || (e->h_err && *e->h_err)) return check_watchdog(e);   // sticky until set_data
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    const bool defer = e->defer != 0;
    int rc = enqueue_rollout(e, st, defer);
    if (rc) return rc;
    if (defer && !e->strict) {
        // like mppi_solve_async: the combine (here: rank-local combine, exchange, finish) rides in
        // the next solve's rollout launch -- the peers' words arrive while this rank draws the
        // next solve's noise -- or is flushed by the first call that needs the results
        e->pending = true;
        e->pending_mode = 2;
        e->pending_xseq = e->xg_seq;
        e->pending_idx = e->solve_idx;
        e->pending_stream = st;
    } else {
        rc = enqueue_combine(e, st, e->d_pm, e->d_ps, e->d_pN, 1, 1, e->NBT * 4, e->grid, 2, nullptr);
        if (rc) return rc;
    }
    e->xg_seq += 1;
    e->solve_idx += 1;
    e->have_solve = true;
    return MPPI_OK;
}

int mppi_set_profiling(mppi_engine* e, int on)
{
    if (!e) return fail(MPPI_EINVAL, "null engine");
    HIPCHK(hipStreamSynchronize(e->stream));
    e->prof = on > 0 ? on : 0;
    e->prof_now = false;
    e->prof_count = 0;
    e->prof_prev = false;
    e->ev_clean.clear();
    e->ev_used[0] = e->ev_used[1] = 0;
    return MPPI_OK;
}

int mppi_kernel_ms(mppi_engine* e, int which, double* avg_ms, int* n_out)
{
    if (!e || !avg_ms || !n_out) return fail(MPPI_EINVAL, "null argument");
    if (which < 0 || which > 1) return fail(MPPI_EINVAL, "which must be 0 or 1");
    HIPCHK(hipDeviceSynchronize());
    // A stamp behind an UNstamped dispatch also covers the wait for that dispatch's tail
    // (~1.5 us); rollout launches are therefore stamped in pairs and only the second of a pair --
    // what the profiler, which stamps everything, would report -- is averaged when there is one.
    bool have_clean = false;
    if (which == 0)
        for (size_t p = 0; p < e->ev_clean.size() && 2 * p + 2 <= e->ev_used[0]; ++p)
            have_clean = have_clean || e->ev_clean[p];
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 2 <= e->ev_used[which]; i += 2) {
        if (which == 0 && have_clean && !(i / 2 < e->ev_clean.size() && e->ev_clean[i / 2])) continue;
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->ev[which][i], e->ev[which][i + 1]));
        tot += ms;
        ++n;
    }
    *avg_ms = n ? tot / n : 0.0;
    *n_out = n;
    return MPPI_OK;
}

int mppi_get_launch_counts(mppi_engine* e, long long out[4])
{
    if (!e || !out) return fail(MPPI_EINVAL, "null argument");
    int rc = ensure_geometry(e);
    if (rc) return rc;
    out[0] = e->n_rollout_launches;
    out[1] = e->n_riding_launches;
    out[2] = e->n_combine_launches;
    out[3] = e->resident_ride;
    return MPPI_OK;
}

int mppi_get_geometry(mppi_engine* e, int out[5])
{
    if (!e || !out) return fail(MPPI_EINVAL, "null argument");
    int rc = ensure_geometry(e);
    if (rc) return rc;
    out[0] = e->packed ? 0 : e->C; out[1] = e->nq; out[2] = e->grid; out[3] = mppi::kRolloutThreads;
    out[4] = e->strict;
    return MPPI_OK;
}

}  // extern "C"
