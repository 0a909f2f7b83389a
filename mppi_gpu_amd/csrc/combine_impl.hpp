// combine_impl.hpp -- the combine step as a device function generic in the block size, shared by
// the stand-alone k_combine launch (1024 threads, kernels.hip), the stand-alone 256-thread launch
// and the combine ROLE inside the fused rollout launch (rollout_fused_impl.hpp), which must give
// equal bits.
#pragma once
#include "device_common.hpp"

// How a rollout block of a riding launch polls for its controls (tools/mkvariant.sh, tools/abn.sh on
// one box, C2): asking for the watchdog word WITH every batch halves the poll period and makes the
// solve SLOWER (12.67 against 12.41 us: 625 blocks looking at the same 3.2 KB more often stand in the
// way of the 50 blocks that write it); pauses of 1 / 8 / 16 / 32 between polls: 12.38 / 12.43 /
// 12.33 / 12.29 us -- flat; the short pause keeps the worst case of a late look small.  A pause
// BEFORE the first look (0.9 / 1.7 / 2.7 us): 12.0 / 12.5 / 13.4 against 11.95 us -- no gain.
#ifndef MPPI_POLL_WD_BATCH
#define MPPI_POLL_WD_BATCH 0   // 1: the watchdog word is requested with every poll batch, not after it
#endif
#ifndef MPPI_POLL_SLEEP
#define MPPI_POLL_SLEEP 1      // s_sleep between two polls of a rollout block waiting for its controls
#endif

namespace mppi {

struct CombineSmem {     // LDS of one combine block; carve from static or dynamic shared memory
    float* r;            // [kMaxParts]
    float* red;          // [THREADS]: (THREADS/64) waves x 64/kCombineCols row groups x kCombineCols
    float* scal;         // [2 * THREADS/64]
    int* flag;           // [1]
};
template <int THREADS>
constexpr int combine_smem_floats()
{
    return kMaxParts + THREADS + 2 * (THREADS / 64) + 1;
}
template <int THREADS>
__device__ __forceinline__ CombineSmem carve_combine_smem(float* base)
{
    CombineSmem s;
    s.r = base;
    s.red = s.r + kMaxParts;
    s.scal = s.red + THREADS;
    s.flag = reinterpret_cast<int*>(s.scal + 2 * (THREADS / 64));
    return s;
}

// ------------------------------------------------------------------------------------------
// Combine: beta (src/point_mass.cu:273-322), nabla (:328-377), weighted update
// (:384-480), action read-out and shift (:195-199, :805-824) in one launch.
//
// Grid = (ceil(TA/kCombineCols) column blocks) x (RS row splits).  Every block recomputes
// beta and nabla from the (<= kMaxParts) partial minima / exp-sums in a fixed order, then sums
// ITS rows of the weighted-noise partials for ITS kCombineCols columns, all row loads in flight, 64/kCombineCols
// rows per wave-instruction.  With RS > 1 the splits meet through a per-column-block ticket: each
// stores its 64 sums, releases at agent scope and takes a ticket; the block that draws the
// last ticket acquires and adds the RS slabs IN SPLIT ORDER (so the result does not depend on
// arrival order) and applies the update.  The ticket is zero at creation and reset by the
// last arriver.
// ------------------------------------------------------------------------------------------
// ---- rank-partial exchange words (XchgArgs) -------------------------------------------------
__device__ __forceinline__ void ll_store(unsigned long long* p, float v, unsigned int tag)
{
    const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#ifndef MPPI_FIN_SCOPE_AGENT
#define MPPI_FIN_SCOPE_AGENT 0     // experiment: agent-scope stores for words that stay on this GPU
#endif
#ifndef MPPI_FIN_COPIES
#define MPPI_FIN_COPIES 1          // experiment: one copy of the finished-control words per XCD
#endif
// the finished control n, for the rollout blocks of a riding launch to pick up
__device__ __forceinline__ void publish_fin(const CombineArgs& a, int n, float unew);
// a tagged word read only by blocks of THIS device (row-split sums, finished controls)
__device__ __forceinline__ void ll_store_local(unsigned long long* p, float v, unsigned int tag)
{
#if MPPI_FIN_SCOPE_AGENT
    const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    ll_store(p, v, tag);
#endif
}

__device__ __forceinline__ void publish_fin(const CombineArgs& a, int n, float unew)
{
#pragma unroll
    for (int cpy = 0; cpy < MPPI_FIN_COPIES; ++cpy)
        ll_store_local(a.slab_tag + (size_t)(kMaxSmallSplits + cpy) * a.TA + n, unew, a.tag);
}

// Has a block of some launch of this engine already given up waiting?  (the device watchdog
// word; sticky until mppi_set_data)  Later launches then neither wait nor publish anything.
__device__ __forceinline__ bool watchdog_tripped(const int* err_dev)
{
    return err_dev && __hip_atomic_load(err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Poll inbox word p1 (and p2 when non-null, both loads in flight together) until they carry
// `tag`; bounded by the exchange time-out so that every wave reaches its exit whatever the
// peers do, and cut short once the watchdog word says that somebody else has given up.
__device__ __forceinline__ void ll_poll2(const unsigned long long* p1, const unsigned long long* p2,
                                         unsigned int tag, unsigned long long limit, float& v1,
                                         float& v2, int& timed_out, const int* err_dev = nullptr)
{
    const unsigned long long t0 = wall_clock64();
    bool ok1 = (p1 == nullptr), ok2 = (p2 == nullptr);
    v1 = 0.0f;
    v2 = 0.0f;
    for (;;) {
        unsigned long long w1 = 0, w2 = 0;
        if (!ok1) w1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!ok2) w2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!ok1 && (unsigned int)(w1 >> 32) == tag) {
            v1 = __uint_as_float((unsigned int)w1);
            ok1 = true;
        }
        if (!ok2 && (unsigned int)(w2 >> 32) == tag) {
            v2 = __uint_as_float((unsigned int)w2);
            ok2 = true;
        }
        if (ok1 && ok2) return;
        if (wall_clock64() - t0 > limit || watchdog_tripped(err_dev)) {
            timed_out = 1;
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// The update of one control value (column n = t*A + axis): the same expression wherever it is
// evaluated, hence the same bits.  With a.clamp the result is limited to the axis' +-max_a
// (opt-in; the reference parses max-a and never applies it, src/main.cu:524,566-568).
__device__ __forceinline__ float updated_control(const CombineArgs& a, int n, float uin, float tot,
                                                 float nabla)
{
    float unew = uin + tot / nabla;
    if (a.clamp) {
        const float lim = a.max_a[n % a.A];
        unew = fminf(fmaxf(unew, -lim), lim);
    }
    return unew;
}

// Write column n of the updated, shifted controls (and the action).
__device__ __forceinline__ void publish_control(const CombineArgs& a, int n, float unew)
{
    float* Uout = a.U + (size_t)((a.solve_idx + 1ull) & 1ull) * a.TA;
    if (n < a.A) {
        a.act_dev[n] = unew;
        if (a.act_host)      // one 8-byte store over PCIe: the host may poll for it (mppi_get_act)
            __hip_atomic_store(a.act_host + n,
                               ((unsigned long long)a.act_tag << 32) | __float_as_uint(unew),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        Uout[n - a.A] = unew;
    }
    if (n >= a.TA - a.A) Uout[n] = unew;       // last step repeated
}

// Final combine of G rank partials held in LDS (xm[g], xs[g], xv[g][col]), rank order, one
// thread per column; shared by the direct exchange and by the gathered (RCCL) path.
__device__ __forceinline__ void finish_columns(const CombineArgs& a, int cb, int tid, int G,
                                               const float* xm, const float* xs, const float* xv,
                                               float uin)
{
    if (tid >= kCombineCols) return;
    const int n = cb * kCombineCols + tid;
    float beta = xm[0];
    for (int g = 1; g < G; ++g) beta = fminf(beta, xm[g]);
    float nabla = 0.0f, tot = 0.0f;
    for (int g = 0; g < G; ++g) {
        const float r = (xm[g] < INFINITY) ? expf(-a.inv_lambda * (xm[g] - beta)) : 0.0f;
        nabla = fmaf(r, xs[g], nabla);
        tot = fmaf(r, xv[g * kCombineCols + tid], tot);
    }
    if (n < a.TA) {
        const float unew = updated_control(a, n, uin, tot, nabla);
        // (the tagged word first: a rollout block of a riding launch is waiting for it)
        if (a.slab_tag) publish_fin(a, n, unew);
        publish_control(a, n, unew);
    }
    if (cb == 0 && tid == 0) {
        a.dev->beta = beta;
        a.dev->nabla = nabla;
    }
}

__device__ __forceinline__ void combine_apply(const CombineArgs& a, int n, float uin, float tot,
                                              float nabla)
{
    if (a.final_mode) {
        const float unew = updated_control(a, n, uin, tot, nabla);
        // (the tagged word first: a rollout block of a riding launch is waiting for it)
        if (a.slab_tag) publish_fin(a, n, unew);
        publish_control(a, n, unew);
    } else {
        a.partial_out[2 + n] = tot;
    }
}

// Rollout side of a riding combine (DeferredCombine): the combine-role blocks of this very launch
// are producing this solve's controls; take them from the tagged words the applying blocks
// publish, one value per thread and sweep and four words in flight per thread, polling what is not
// there yet (bounded; cut short when the watchdog word is already set), and stage them -- with
// lambda*inv_s*u -- into LDS as one float4 per Philox block (n_blocks blocks, zero padded).
template <int A>
__device__ __forceinline__ void ride_fetch_controls(const RolloutArgs& g, const DeferredCombine& d,
                                                    float lambda, float4* ulds, float4* uclds,
                                                    int n_blocks, int TA)
{
    const unsigned long long t0 = wall_clock64();
    const unsigned long long limit = g.ride_timeout_ticks;
    bool timed_out = false;
    float* uflat = reinterpret_cast<float*>(ulds);
    float* ucflat = reinterpret_cast<float*>(uclds);
#if MPPI_FIN_COPIES > 1
    // (XCC_ID: hardware register 20 of gfx950; blocks are dealt round robin over the 8 XCDs)
    const unsigned long long* fin_p =
        g.fin_tag + (size_t)(__builtin_amdgcn_s_getreg((31 << 11) | 20) % MPPI_FIN_COPIES) * TA;
#else
    const unsigned long long* fin_p = g.fin_tag;
#endif
    const unsigned int tag_want = d.c.tag;
    constexpr int kBatch = 4;      // words in flight per thread: one round trip, not four
    for (int base = threadIdx.x; base < n_blocks * 4; base += kBatch * kRolloutThreads) {
        float unew[kBatch];
        bool have[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            unew[j] = 0.0f;
            have[j] = base + j * kRolloutThreads >= TA;     // padding: nothing to fetch
        }
        for (;;) {
            unsigned long long w[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int idx = base + j * kRolloutThreads;
                const int n = (idx < TA - A) ? idx + A : idx;  // shift; last step repeats
                w[j] = have[j] ? 0ull
                               : __hip_atomic_load(fin_p + n, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            // (the watchdog word travels WITH the batch: asked for after the check it is a second
            //  memory round trip per poll, and the poll period is what a waiting block loses on
            //  average between the word's arrival and its own next look)
#if MPPI_POLL_WD_BATCH
            const bool tripped = watchdog_tripped(g.err_dev);
#endif
            bool all = true;
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                if (!have[j] && (unsigned int)(w[j] >> 32) == tag_want) {
                    unew[j] = __uint_as_float((unsigned int)w[j]);
                    have[j] = true;
                }
                all = all && have[j];
            }
            if (all) break;
#if !MPPI_POLL_WD_BATCH
            const bool tripped = watchdog_tripped(g.err_dev);
#endif
            if (wall_clock64() - t0 > limit || tripped) {
                timed_out = true;
                break;
            }
            __builtin_amdgcn_s_sleep(MPPI_POLL_SLEEP);
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int idx = base + j * kRolloutThreads;
            if (idx < n_blocks * 4) {
                uflat[idx] = unew[j];
                ucflat[idx] = lambda * (unew[j] * g.inv_s[idx % A]);
            }
        }
    }
    if (timed_out) {     // device watchdog word, reported by the next call on the engine
        *g.err_dev = 2;
        if (g.err_host) *g.err_host = 2;
    }
}

// One combine block: column block cb = bid % n_cols, row split rs = bid / n_cols.
template <int THREADS, int NR>    // NR row loads in flight per lane
__device__ __forceinline__ void combine_body(const CombineArgs& a, int bid, const CombineSmem& sm)
{
    constexpr int NW = THREADS / 64;                  // waves per block
    constexpr int RPW = 64 / kCombineCols;            // rows per wave-instruction
    constexpr int NRG = NW * RPW;                     // row groups per block
    float* const r_lds = sm.r;                        // [kMaxParts]
    float* const red = sm.red;                        // [NRG * kCombineCols]
    float* const scal = sm.scal;                      // [2 * NW]
    volatile int& last_flag = *sm.flag;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;   // 0..NW-1
    const int cb = bid % a.n_cols;
    const int rs = bid / a.n_cols;
    const int RS = a.RS;
    constexpr int PT = kMaxParts / THREADS;           // partials per thread

    // rows of this split
    const int per = (a.n_parts + RS - 1) / RS;
    const int p_begin = rs * per;
    const int p_end = min(a.n_parts, p_begin + per);
    const int col = lane & (kCombineCols - 1);
    const int rgrp = wave * RPW + lane / kCombineCols;   // 0..NRG-1
    const int n = cb * kCombineCols + col;

    // ---- every global load this block needs is issued up front: the partial minima and
    //      exp-sums, the first batch of weighted-noise rows and the nominal control; beta,
    //      nabla and the rescale factors are computed while they are in flight -------------
    // (measured and dropped: every WAVE working out beta, nabla and all rescale factors by itself
    //  -- no block barrier before the row sums -- takes the riding combine as long as this: the two
    //  barriers it saves cost what its 4x redundant exponentials do, 5.9 us either way)
    float beta, nabla;
    float v[NR];
    float uin = 0.0f;
#ifdef MPPI_TRACE
    if (a.n_parts > 0) MPPI_CSTAMP(8);        // (the kernel arguments have arrived)
#endif
    // The requests themselves are the first thing on the path between two solves, and a lone wave
    // issues one instruction every ~10-20 cycles (cold instruction cache, a SIMD shared with the
    // next solve's noise): as plain indexed loads -- 64-bit address arithmetic and an exec-masked
    // branch per element -- they were 890 instructions and 2.0 us before the first barrier.  Raw
    // BUFFER loads instead: the descriptors are wave-uniform (SGPRs), the lane part of the offset
    // is computed once, the sweep / row part is an SGPR offset, and a request past the end of the
    // buffer returns 0 by itself (no bounds test, no branch): two instructions per element.
    {
        float mreg[PT], sreg[PT];
        const int jmax = (a.n_parts + THREADS - 1) / THREADS;   // sweeps that hold any partial at all
        const unsigned int m_step = (unsigned int)a.m_stride * 4u, s_step = (unsigned int)a.s_stride * 4u;
        const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.m), 0, (int)((unsigned int)(a.n_parts - 1) * m_step + 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.s), 0, (int)((unsigned int)(a.n_parts - 1) * s_step + 4u), 0x00020000);
        const unsigned int m_off = (unsigned int)tid * m_step, s_off = (unsigned int)tid * s_step;
        auto sweep = [&](int j) {
            mreg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                m_rsrc, m_off, (unsigned int)(j * THREADS) * m_step, 0));
            sreg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                s_rsrc, s_off, (unsigned int)(j * THREADS) * s_step, 0));      // past the end: 0
        };
        constexpr int PT0 = PT < 4 ? PT : 4;     // the sweeps of up to 4 * THREADS partials: no test
#pragma unroll
        for (int j = 0; j < PT; ++j) { mreg[j] = INFINITY; sreg[j] = 0.0f; }
#pragma unroll
        for (int j = 0; j < PT0; ++j) sweep(j);
        if (jmax > PT0) {                                        // block-uniform
#pragma unroll
            for (int j = PT0; j < PT; ++j)
                if (j < jmax) sweep(j);
        }
        MPPI_CSTAMP(9);                       // (partial minima / exp-sums requested)
        {
            const int rows = p_end - p_begin;                    // rows of this split (may be <= 0)
            const unsigned int n_step = (unsigned int)a.N_stride * 4u;
            const __amdgpu_buffer_rsrc_t n_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.N + (size_t)(rows > 0 ? p_begin : 0) * a.N_stride), 0,
                rows > 0 ? (int)((unsigned int)(rows - 1) * n_step + (unsigned int)a.TA * 4u) : 0,
                0x00020000);
            // (a column n >= TA of the last column block reads the head of the next row, or 0
            //  past the last row: its sums are never applied)
            const unsigned int n_off = (unsigned int)rgrp * n_step + (unsigned int)n * 4u;
#pragma unroll
            for (int j = 0; j < NR; ++j)
                v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    n_rsrc, n_off, (unsigned int)(NRG * j) * n_step, 0));
        }
        if (a.final_mode != 0 && wave == 0 && n < a.TA) uin = a.U[(a.solve_idx & 1ull) * a.TA + n];

        // (sweeps past the last partial asked for nothing or got 0: their minimum is +inf)
#pragma unroll
        for (int j = 0; j < PT; ++j)
            if (tid + j * THREADS >= a.n_parts) mreg[j] = INFINITY;

        MPPI_CSTAMP(1);
        float mloc = mreg[0];
#pragma unroll
        for (int j = 1; j < PT; ++j)
            if (j < jmax) mloc = fminf(mloc, mreg[j]);
        mloc = wave_min(mloc);
        if (lane == 0) scal[wave] = mloc;
        __syncthreads();
        beta = scal[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) beta = fminf(beta, scal[i]);
        MPPI_CSTAMP(5);

        float sloc = 0.0f;
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            if (j < jmax) {      // block-uniform: no exponentials for sweeps without partials
                const int p = tid + j * THREADS;
                const float r = (mreg[j] < INFINITY) ? expf(-a.inv_lambda * (mreg[j] - beta)) : 0.0f;
                if (p < a.n_parts) r_lds[p] = r;
                sloc += r * sreg[j];
            }
        }
        sloc = wave_sum(sloc);
        if (lane == 0) scal[NW + wave] = sloc;
        __syncthreads();
        nabla = 0.0f;
#pragma unroll
        for (int i = 0; i < NW; ++i) nabla += scal[NW + i];
    }

    MPPI_CSTAMP(2);
#ifdef MPPI_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // analysis: when have ALL row loads landed?
    MPPI_CSTAMP(6);
#endif
    // row sums: four accumulators (the additions of a lane do not wait for each other); a row past
    // the split's end was loaded as 0 and takes a finite factor (index clamped), so no test per row
    float acc;
    {
        float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int p_top = a.n_parts - 1;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int p = min(p_begin + rgrp + NRG * j, p_top);
            a4[j & 3] = fmaf(r_lds[p], v[j], a4[j & 3]);
        }
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    if (n < a.TA) {      // rows beyond the first batch (only when RS hit its cap)
        for (int p0 = p_begin + rgrp + NRG * NR; p0 < p_end; p0 += NRG * NR) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int p = p0 + NRG * j;
                v[j] = (p < p_end) ? a.N[(size_t)p * a.N_stride + n] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int p = p0 + NRG * j;
                if (p < p_end) acc = fmaf(r_lds[p], v[j], acc);
            }
        }
    }
    red[rgrp * kCombineCols + col] = acc;
    MPPI_CSTAMP(7);
    __syncthreads();
    float tot = 0.0f;
    if (wave == 0) {       // both halves of the wave compute the same 32 sums: four running sums over
                           // the row groups rg = 0, 4, 8, .. / 1, 5, .. / .., added pairwise at the end
        float t4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) t4[rg & 3] += red[rg * kCombineCols + col];
        tot = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }

    MPPI_CSTAMP(3);
    // `mine`: this rank's finished sums for the block's columns (threads 0..kCombineCols-1 of the block
    // that applies them); apply_blk is block-uniform
    float mine = tot;
    bool apply_blk = true;
    int timed_out = 0;         // this thread gave up on a word: it publishes nothing
    if (a.slab_tag) {
        // Fence-free meeting of the row splits (the 256-thread shape, which may run inside a busy
        // rollout launch): splits 0 .. RS-2 publish their sums as tagged 8-byte words and are done;
        // the LAST split (highest block index, so everything it waits for was dispatched before
        // it) polls them, adds them in split order with its own sum last, and applies.
        const unsigned int tag = a.tag;
        if (rs != RS - 1) {
            if (tid < kCombineCols && n < a.TA)
                ll_store_local(a.slab_tag + (size_t)rs * a.TA + n, tot, tag);
            return;
        }
        if (RS > 1 && tid < kCombineCols && n < a.TA) {
            float t2 = 0.0f;
            for (int q0 = 0; q0 < RS - 1; q0 += 2) {
                const unsigned long long* p1 = a.slab_tag + (size_t)q0 * a.TA + n;
                const unsigned long long* p2 = (q0 + 1 < RS - 1) ? p1 + a.TA : nullptr;
                float v1, v2;
                ll_poll2(p1, p2, tag, a.x.timeout_ticks, v1, v2, timed_out, a.x.err_dev);
                t2 += v1;
                if (p2) t2 += v2;
            }
            mine = t2 + tot;
            MPPI_CSTAMP(4);
            if (timed_out) {
                *a.x.err_dev = 2;
                if (a.x.err_host) *a.x.err_host = 2;
            }
        }
    } else if (RS > 1) {
        // publish this split's 64 sums, then take a ticket (guide: agent-scope release before
        // the counter, agent-scope acquire in the last arriver, waits written out by hand)
        if (tid < kCombineCols && n < a.TA) a.slab[(size_t)rs * a.TA + n] = tot;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int old = __hip_atomic_fetch_add(&a.tickets[cb], 1u, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = (old == (unsigned int)(RS - 1));
            if (is_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&a.tickets[cb], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            last_flag = is_last;
        }
        __syncthreads();
        apply_blk = last_flag != 0;
        if (apply_blk && tid < kCombineCols && n < a.TA) {
            float t2 = 0.0f;
            for (int q = 0; q < RS; ++q)
                t2 += __hip_atomic_load(&a.slab[(size_t)q * a.TA + n], __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
            mine = t2;
        }
    }
    if (a.final_mode != 2) {
        // (a column whose sums did not arrive in time is left alone: the watchdog word is set, the
        //  host reports MPPI_ESTATE from then on and hands out nothing until mppi_set_data)
        if (apply_blk && !timed_out && tid < kCombineCols && n < a.TA)
            combine_apply(a, n, uin, mine, nabla);
        if (cb == 0 && rs == RS - 1 && tid == 0) {
            if (a.final_mode) {
                a.dev->beta = beta;
                a.dev->nabla = nabla;
            } else {
                a.partial_out[0] = beta;
                a.partial_out[1] = nabla;
            }
        }
        return;
    }
    if (!apply_blk) return;

    // ---- direct exchange: send this rank's partial to every inbox, collect all G, finish ----
    const XchgArgs& x = a.x;
    float* xv = r_lds;                                   // [G][16]   (r_lds is free by now)
    float* xm = r_lds + kMaxRanks * kCombineCols;        // [G]
    float* xs = xm + kMaxRanks;                          // [G]
    float* mine_lds = xs + kMaxRanks;                    // [16]
    __syncthreads();
    if (tid == 0) last_flag = 0;
    __syncthreads();
    // a block that missed a word -- a row split's sum above, a peer's word below -- sends and
    // publishes nothing: its peers time out as well, the controls stay as they were, and every
    // host reports MPPI_ESTATE until mppi_set_data
    if (timed_out) last_flag = 1;
    if (tid < kCombineCols) mine_lds[tid] = (n < a.TA) ? mine : 0.0f;
    __syncthreads();
    if (last_flag) return;
    const int c = tid & (kCombineCols - 1);
    const int nn = cb * kCombineCols + c;
    const size_t slot_w = (size_t)x.W;
    constexpr int GPB = THREADS / kCombineCols;          // peers served per sweep of the block
    for (int g = tid / kCombineCols; g < x.G; g += GPB) {   // g: peer this thread talks to
        unsigned long long* dst = x.peers[g] + ((size_t)x.parity * x.G + x.rank) * slot_w;
        if (nn < a.TA) ll_store(dst + 2 + nn, mine_lds[c], x.tag);
        if (cb == 0 && c == 0) {
            ll_store(dst + 0, beta, x.tag);
            ll_store(dst + 1, nabla, x.tag);
        }
    }
    for (int g = tid / kCombineCols; g < x.G; g += GPB) {
        const unsigned long long* src = x.peers[x.rank] + ((size_t)x.parity * x.G + g) * slot_w;
        float v1, v2;   // column word of rank g; columns 0 / 1 also fetch beta_g / S_g
        ll_poll2((nn < a.TA) ? src + 2 + nn : nullptr, (c < 2) ? src + c : nullptr, x.tag,
                 x.timeout_ticks, v1, v2, timed_out, x.err_dev);
        xv[g * kCombineCols + c] = v1;
        if (c == 0) xm[g] = v2;
        if (c == 1) xs[g] = v2;
    }
    if (timed_out) {
        *x.err_dev = 1;
        if (x.err_host) *x.err_host = 1;
        last_flag = 1;
    }
    __syncthreads();
    if (last_flag) return;
    finish_columns(a, cb, tid, x.G, xm, xs, xv, uin);
}

}  // namespace mppi
