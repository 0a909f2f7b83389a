// rollout_packed_impl.hpp -- the PACKED fused rollout: whole trajectories laid end to end over the
// lanes of a wavefront.
//
// The row-aligned kernel (rollout_fused_impl.hpp) gives every trajectory a power-of-two number of
// lanes: at T = 200 a 3-D trajectory is 50 groups of 4 steps, 12.5 lanes of 4 groups, and 16 lanes
// are spent on it -- 22 % of every pass of every tile (Philox, Box-Muller, dynamics, reductions)
// is work on lanes past the horizon.  Here a wavefront owns 64*NG consecutive GROUP SLOTS and fills
// them with TPW = floor(64*NG / NGT) whole trajectories (NGT groups each): slot s = j*NGT + r is
// group r of the wave's trajectory j and lives in lane s / NG.  A lane therefore holds NG
// consecutive groups of ONE trajectory, or the last groups of one trajectory followed by the first
// groups of the next (NGT >= NG: at most one boundary per lane).  T = 200, act_dim 3: 5
// trajectories in 62.5 lanes (97.7 % of the slots used instead of 78 %); act_dim 2, NG = 8: 5
// trajectories in 62.5 lanes (instead of 89 %).
//
// What changes against the row-aligned kernel:
//   * the chunk scan is a SEGMENTED scan over the whole wavefront (row_shr inside the DPP rows,
//     row_bcast:15 / :31 across them) with a "a trajectory starts in my range" flag; a lane with a
//     boundary hands on the response of its tail only, and starts its own tail from x0;
//   * a trajectory's cost is a segmented sum of lane partials; the totals go through a per-wave LDS
//     table so that every lane gets the weight(s) of the trajectories it holds;
//   * the weighted-noise sums leave the registers through LDS: every lane writes its w*e blocks to
//     its slot, and after the tile's barrier thread m adds, for Philox block m of the horizon,
//     the 4 waves x TPW trajectories in fixed order (no DPP reduce-scatter: the trajectories of a
//     wave no longer sit in aligned lane rows);
//   * dynamics and cost run on SCALED state variables d_p = sqrt(w_p)(p - g_p),
//     d_v = sqrt(w_v)(v - g_v), whose stage cost is d_p^2 + d_v^2: 7 VALU instructions per normal
//     instead of 11 (src/point_mass_gpu.cu:97-107 + src/cost.cu:42-55 are the same arithmetic up
//     to rounding; the test bar of the fused kernels applies).  Scales and gains come from the host.
//   * a horizon that is not a whole number of groups (T = 50 at act_dim 3: 12 groups of 4 steps
//     and one of 2 -- the reference's shipped config/point_mass3d.yaml) is padded to NGT =
//     ceil(T / SG) groups: the noise and the controls of the steps past T are zero, and in the
//     trajectory's last group those steps neither move the state nor add stage cost.  RAGGED is a
//     template parameter of the KERNEL: as a run-time branch inside one kernel it cost the
//     whole-groups instantiation 6 VGPRs and 90 spilled SGPRs.
// Requirements (the engine falls back to the row-aligned kernel otherwise): NG <= NGT <= 64*NG,
// cost weights >= 0.
#pragma once
#include "device_common.hpp"
#include "combine_impl.hpp"
#include <type_traits>

// experiment switches (tools/mkvariant.sh): defaults are the measured best
#ifndef MPPI_PK_FENCES
#define MPPI_PK_FENCES 1       // scheduling fences between the groups of passes 1b / 2
#endif
#ifndef MPPI_PK_PREFETCH
#define MPPI_PK_PREFETCH 1     // controls of the next group loaded a group ahead
#endif
#ifndef MPPI_PK_PRIO
#define MPPI_PK_PRIO 2         // s_setprio of the latency-bound passes (the Philox pass runs at 0)
#endif
#ifndef MPPI_PK_STORE_AUX
#define MPPI_PK_STORE_AUX 17   // cache policy of the noise stores: 17 = sc0 | sc1 (write-through)
#endif
#ifndef MPPI_PK_FIRST_STAGGER
#define MPPI_PK_FIRST_STAGGER 2    // the first half of the grid runs its FIRST tile one priority level up
#endif
#ifndef MPPI_PK_SCAN_BRANCH
#define MPPI_PK_SCAN_BRANCH 0  // experiment: the absorb test of the segmented scans as an exec-masked branch
#endif
#if MPPI_PK_SCAN_BRANCH
#define MPPI_PK_NOFLATTEN() asm volatile("")   // (a volatile asm cannot be speculated: no if-conversion)
#else
#define MPPI_PK_NOFLATTEN() do { } while (0)
#endif
#ifndef MPPI_PK_PRIO_ALT
#define MPPI_PK_PRIO_ALT 0     // experiment: the two blocks of a CU alternate a priority bonus per tile
#endif
#ifndef MPPI_PK_RACC3
#define MPPI_PK_RACC3 1        // three cost accumulators per axis (shorter dependent chains)
#endif
#ifndef MPPI_PK_QSCAN
#define MPPI_PK_QSCAN 1        // per-tile scans on tile-invariant absorb masks, responses summed at a common time
#endif
#ifndef MPPI_PK_NOMINAL
#define MPPI_PK_NOMINAL 1      // lane start states = nominal trajectory (once per block) + response to the noise
#endif
#if MPPI_PK_FENCES
#define MPPI_PK_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MPPI_PK_FENCE() do { } while (0)
#endif

namespace mppi {

constexpr int kPkRow = 65;     // float4 per row of the weighted-noise slots in LDS (64 lanes + 1 pad)

template <int A>
struct PackedLane {     // wave-uniform constants held in VGPRs (see LaneParams in the fused kernel)
    float sp[A], sv[A];         // state scales
    float k1[A], k2[A], k3[A];  // d_p' = d_p + k1 d_v + k2 a (+ cg),  d_v' = d_v + k3 a
    float cg[A];
    float sigma[A];
    float dt, B0, dt2;
};

template <int A, int NG, bool SAMPLE, bool RIDE, bool RAGGED>
__device__ __forceinline__ void packed_body(const RolloutHot& h, const DeferredCombine& d)
{
    const int bid = RIDE ? (int)blockIdx.x - d.n_blocks : (int)blockIdx.x;
    const int nblk = RIDE ? (int)gridDim.x - d.n_blocks : (int)gridDim.x;
    constexpr int SG = Dim<A>::SG;
    constexpr int BPG = Dim<A>::BPG;
    constexpr int NQ = NG * BPG;              // Philox blocks per lane
    constexpr int NE = NQ * 4;                // normals held per lane
    constexpr int L = NG * SG;                // steps per lane

    const int K = h.K, TA = h.TA, NBT = h.NBT, NGT = h.NGT, TPW = h.TPW;
    // RAGGED (T not a whole number of groups): NGT*BPG >= NBT blocks of the controls are staged in
    // LDS (zero padded), pk_nlast < SG steps of a trajectory's last group lie inside the horizon,
    // and the rows of part_N are padded to whole blocks
    const int NBTs = RAGGED ? h.NBTp : NBT;
    const int pk_nlast = RAGGED ? h.pk_nlast : SG;
    const int Nrow = RAGGED ? h.Nrow : TA;
    const int n_tileblk = h.n_tileblk;
    const long long k_offset = h.k_offset;
    const unsigned long long seed = h.seed;
    float* const Eint = h.Eint;
    const RolloutArgs& g = *h.rest;           // cold part of the descriptor (device memory)

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* ulds = reinterpret_cast<float4*>(smem_raw);          // [NBTs] U, one float4 per block
    float4* uclds = ulds + NBTs;                                 // [NBTs] lambda*inv_s*U
    float4* buf = uclds + NBTs;                                  // [4][NQ][kRow] weighted noise sums
    float* misc = reinterpret_cast<float*>(buf + 4 * NQ * kPkRow);   // [8]
    float* ctab = misc + 8;                                      // [4][TPW + 2] trajectory costs

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    MPPI_STAMP(0);

    // ---- loads of the nominal controls and the cold constants (complete under the Philox work) --
    const float lambda = g.lambda, inv_lambda = g.inv_lambda;
    if constexpr (!RIDE) stage_controls_pair<A>(g, h.U_in, lambda, ulds, uclds, NBTs, TA);
    PackedLane<A> P;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        P.sp[i] = to_vgpr(g.pk_sp[i]); P.sv[i] = to_vgpr(g.pk_sv[i]);
        P.k1[i] = to_vgpr(g.pk_k1[i]); P.k2[i] = to_vgpr(g.pk_k2[i]); P.k3[i] = to_vgpr(g.pk_k3[i]);
        P.cg[i] = to_vgpr(g.pk_cg[i]);
        P.sigma[i] = to_vgpr(g.sigma[i]);
    }
    P.dt = to_vgpr(g.dt);
    P.B0 = to_vgpr(g.B0);
    P.dt2 = P.dt * P.dt;
    const bool sigma_one = g.sigma_one != 0;  // one sigma for all axes: it sits in the radius factor
    const float noise_r2c = to_vgpr(g.noise_r2c);
    const bool has_cg = g.pk_has_cg != 0;     // wave-uniform: a velocity goal != 0 drifts d_p
    const long long k_cover = g.k_cover;
    const unsigned int cover_and = g.cover_and;
    const long long nt_from_tile = g.nt_from_tile;
    const int store_e = g.store_e;           // wave-uniform: 0 noise not materialised, 1 write-through
                                             // stores, 2 non-temporal stores (see engine.hip)
    float* const cost_out = g.cost;

    // ---- where this lane's group slots sit (the same for every tile) ---------------------------
    const int s0 = lane * NG;
    const int j0 = s0 / NGT;                        // trajectory (of the wave) of the first slot
    const int r0 = s0 - j0 * NGT;                   // its group
    const int split = min(NGT - r0, NG);            // groups of trajectory j0 in this lane
    const bool tail_slot = split < NG;              // trajectory j0 ENDS inside the lane and j0+1
                                                    // (or the idle rest of the wave) starts
    const bool starts0 = r0 == 0;                   // the lane starts with a trajectory start
    const bool ends_head = (NGT - r0) <= NG;        // trajectory j0 ends in this lane
    const int rbh = r0 * BPG;                       // block of the trajectory = rb? + gi*BPG + b
    const int rbt = -split * BPG;
    unsigned int split_mask = 0;                    // bit g: some lane has its boundary after g groups
#pragma unroll
    for (int gq = 1; gq < NG; ++gq)
        if (__ballot(tail_slot && split == gq) != 0ull) split_mask |= 1u << gq;
    // ragged horizon: group `split - 1` of a lane in which trajectory j0 ends is that trajectory's
    // LAST group; its normals with in-group index >= pk_nlast * A lie past the horizon
    const int last_gi = ends_head ? split - 1 : -1;
    int dead_from[NG];                              // first step of group gi past the horizon
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) dead_from[gq] = (RAGGED && gq == last_gi) ? pk_nlast : SG;
    const int flag0 = (tail_slot || starts0) ? 1 : 0;      // a trajectory starts in my range
    const int n_out = tail_slot ? L - split * SG : L;      // steps of the range handed on
    const float nh = (float)(split * SG), nt = (float)(L - split * SG);
    // scaled x0 (where a trajectory starts) and x0 moved freely to the lane's first step: the start
    // state of the lane's head is dps1 + sp * (what the lanes before it contributed), likewise dvs
    float dps0[A], dvs0[A], dps1[A];
    {
        const float tau0 = (float)(r0 * SG) * P.dt;        // time since the start of trajectory j0
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const float x0p = h.x0[i], x0v = h.x0[A + i];
            dps0[i] = fmaf(P.sp[i], x0p, -g.pk_gps[i]);
            dvs0[i] = fmaf(P.sv[i], x0v, -g.pk_gvs[i]);
            dps1[i] = fmaf(P.sp[i], tau0 * x0v, dps0[i]);
        }
    }
    const unsigned long long blk_base = h.solve_idx * (unsigned long long)NBT;
#if MPPI_PK_QSCAN
    // What the segmented scans of a tile decide is the same for every tile: whether this lane, at
    // level d of the scan, takes in what sits d lanes to its left ("a trajectory starts in my
    // range" flags only) -- worked out ONCE, one bit per level.  And with every range's position
    // response carried forward to ONE reference time (the end of the trajectory: Q = P + (R - E) dt V,
    // E = step at which the range ends), responses ADD: the scan needs no step counts and no
    // per-level time shift, and a lane gets its start state back as Qex - (R - S) dt Vex, S = the
    // step its head starts at.  Per tile and level: 6 DPP moves + 6 additions instead of 8 moves,
    // a conversion, a multiply, 3 FMAs, 7 additions and the flag bookkeeping.  (Used for the
    // response to the NOISE -- small numbers; the once-per-block nominal pass keeps the plain
    // form, whose sums do not cancel.)
    unsigned int absorb = 0;
    {
        int flg = flag0;
        const int cr = lane & 15;
#define MPPI_PK_ABS(BIT, COND, GETI)                                                \
        {                                                                           \
            const int fl = GETI(flg);                                               \
            if ((COND) && flg == 0) { absorb |= (BIT); flg = fl; }                  \
        }
        MPPI_PK_ABS(1u, cr >= 1, dppi<MPPI_ROW_SHR(1)>)
        MPPI_PK_ABS(2u, cr >= 2, dppi<MPPI_ROW_SHR(2)>)
        MPPI_PK_ABS(4u, cr >= 4, dppi<MPPI_ROW_SHR(4)>)
        MPPI_PK_ABS(8u, cr >= 8, dppi<MPPI_ROW_SHR(8)>)
        MPPI_PK_ABS(16u, (lane & 16) != 0, (dpp_rows_i<kRowBcast15, 0xA>))
        MPPI_PK_ABS(32u, (lane & 32) != 0, (dpp_rows_i<kRowBcast31, 0xC>))
#undef MPPI_PK_ABS
    }
    // steps from the end of the range handed on / from the start of the head to the end of the trajectory
    const float q_ref = (float)(NGT * SG - (tail_slot ? n_out : r0 * SG + L)) * P.dt;
    const float c_ref = (float)(NGT * SG - r0 * SG) * P.dt;
#endif
#if MPPI_PK_NOMINAL
    float dps_nom[A], dvs_nom[A];
#pragma unroll
    for (int i = 0; i < A; ++i) { dps_nom[i] = 0.f; dvs_nom[i] = 0.f; }
#endif

    float Mw = INFINITY, Sw = 0.0f;          // this WAVE's running minimum and exp-sum
    // this lane's slot q of the weighted noise sums is bw[q * kPkRow]: layout [wave][q][lane], so
    // that every wave-instruction of the tile loop reads / writes one contiguous 1 KiB row; rows are
    // 65 float4 apart so that the final merge, which gathers one lane of MANY rows per instruction,
    // does not find them all in one bank (rows 1 KiB apart: 16-way conflicts, 2.4 us per block)
    float4* const bw = buf + wave * NQ * kPkRow + lane;

#ifdef MPPI_TRACE
    int tile_no = 0;
#define MPPI_PK_STAMP(i) do { if (tile_no == h.trace_tile) MPPI_STAMP(i); } while (0)
#else
#define MPPI_PK_STAMP(i) do { } while (0)
#endif
    // The block's FIRST tile is a copy of the tile body of its own (FIRST = true): it alone stages
    // the controls (in a riding launch: polls for them), runs the once-per-block nominal pass and
    // writes -- instead of accumulating -- the wave's LDS slots.  Left as run-time flags inside ONE
    // loop body, the polling code of the riding variant costs the steady-state loop registers and
    // scalar spills (255 VGPRs / 81 spilled SGPRs against 250 / 43: 4 % per tile); peeled, every
    // later tile runs the very loop of the plain kernel.
    auto tile_body = [&](const int tb, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const long long tile = (long long)tb * 4 + wave;           // one wavefront = one tile
        MPPI_PK_STAMP(11);
        const long long kh = tile * TPW + j0, kt = kh + 1;         // local sample indices
        const bool valid_h = j0 < TPW && kh < K;
        const bool valid_t = tail_slot && j0 + 1 < TPW && kt < K;
        float* etile = Eint + ((size_t)tile * NQ * 64 + lane) * 4;  // + q*256 floats per block
        __amdgpu_buffer_rsrc_t e_rsrc;
        {
            const unsigned long long tbase =
                reinterpret_cast<unsigned long long>(Eint + (size_t)tile * NQ * 256);
            const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)tbase);
            const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(tbase >> 32));
            e_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, NQ * 1024, 0x00020000);
        }

        // ---- pass 1a: draw (or load) the lane's noise and store it: write-through (sc0 sc1, see
        //      the fused kernel) while the noise of a launch fits the 256 MB memory-side cache,
        //      NON-TEMPORAL beyond (K = 2e5 ... 1e6 at 3-D: 15-18 % of the launch; at 240 MB
        //      write-through is 2 % better; the engine chooses, RolloutArgs::store_e) --------------
        // Instruction priority by phase: the Philox pass can issue every cycle it is offered, the
        // passes after it are chains of dependent instructions.  Left at equal priority the OLDER
        // of the two waves of a SIMD wins every arbitration: the first block of a CU ran its tiles
        // in 66 us, the second in 90 (the kernel's time).  Low priority here, high below.
#if MPPI_PK_PRIO_ALT
        // experiment: the two blocks of a CU take turns at the upper hand, tile by tile (at equal
        // priority the OLDER wave of a SIMD wins every arbitration: the second block of every CU
        // runs its tiles 9 % slower and, with as many tiles, sets the launch's duration)
        const bool my_turn = (((tb - bid) / nblk + (bid >= (nblk >> 1) ? 1 : 0)) & 1) != 0;
        if (my_turn) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#elif MPPI_PK_PRIO
#if MPPI_PK_FIRST_STAGGER
        // On the FIRST tile the first half of the grid -- the first block of every CU -- runs one
        // priority level above the second half (noise pass 1 against 0, the passes after it 3
        // against 2): the two blocks of a CU leave the Philox pass one after the other instead of
        // together, and are out of phase -- one in its dependent passes while the other draws noise
        // -- from the first tile on instead of drifting there (C3 66.5 -> 65.6 us per solve)
        if (FIRST && bid < (nblk >> 1)) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#else
        __builtin_amdgcn_s_setprio(0);
#endif
#endif
        float e[NE];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const bool hd = gi < split;
            if constexpr (SAMPLE) {
                const unsigned long long kg = (unsigned long long)(k_offset + (hd ? kh : kt));
                const unsigned long long blkg =
                    blk_base + (unsigned long long)(long long)((hd ? rbh : rbt) + gi * BPG);
#pragma unroll
                for (int b = 0; b < BPG; ++b) {
                    const int q = gi * BPG + b;
                    const uint4 r = PhiloxAt::block(blkg + (unsigned long long)b, kg, seed);
                    scaled_normals4<A>(r, (q * 4) % A, sigma_one, noise_r2c, P.sigma, &e[q * 4]);
                }
                if constexpr (RAGGED) {     // zero the normals past the horizon
                    const int thr = dead_from[gi] * A;
#pragma unroll
                    for (int idx = A; idx < SG * A; ++idx)
                        e[gi * BPG * 4 + idx] = (idx >= thr) ? 0.0f : e[gi * BPG * 4 + idx];
                }
                // (one group at a time: left alone, hipcc interleaves the Philox chains of all
                //  groups of the lane and runs out of registers; the VALU is saturated by one)
                __builtin_amdgcn_sched_barrier(0);
                if (store_e && (hd ? valid_h : valid_t)) {   // idle slots, samples >= K: no store
                    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                    if (store_e == 2 && tile >= nt_from_tile) {   // (the cache policy is an instruction modifier)
#pragma unroll
                        for (int b = 0; b < BPG; ++b) {
                            const int q = gi * BPG + b;
                            const v4u val = {__float_as_uint(e[q * 4]), __float_as_uint(e[q * 4 + 1]),
                                             __float_as_uint(e[q * 4 + 2]), __float_as_uint(e[q * 4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(val, e_rsrc, lane * 16, q * 1024,
                                                                   2 /* nt */);
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < BPG; ++b) {
                            const int q = gi * BPG + b;
                            const v4u val = {__float_as_uint(e[q * 4]), __float_as_uint(e[q * 4 + 1]),
                                             __float_as_uint(e[q * 4 + 2]), __float_as_uint(e[q * 4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(val, e_rsrc, lane * 16, q * 1024,
                                                                   MPPI_PK_STORE_AUX);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < BPG; ++b) {
                    const int q = gi * BPG + b;
                    const float4 t = *reinterpret_cast<const float4*>(etile + (size_t)q * 256);
                    e[q * 4] = t.x; e[q * 4 + 1] = t.y; e[q * 4 + 2] = t.z; e[q * 4 + 3] = t.w;
                }
            }
        }
        MPPI_PK_STAMP(1);
#if MPPI_PK_PRIO_ALT
        if (my_turn) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(2);
#elif MPPI_PK_PRIO
#if MPPI_PK_FIRST_STAGGER >= 2
        if (FIRST && bid < (nblk >> 1)) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(MPPI_PK_PRIO);
#else
        __builtin_amdgcn_s_setprio(MPPI_PK_PRIO);
#endif
#endif

        // ---- pass 1b + scan: the scaled state every lane starts from.  Zero-state response of the
        //      lane's range(s) to accelerations a: V = dt*S1, P = B0*S1 + dt^2*((n-1)*S1 - S2),
        //      S1 = sum a_j, S2 = sum j*a_j.  A lane with a boundary needs the sums of its head
        //      [0, split) and tail separately: the running sums are snapshot after group `split`
        //      (wave-uniform test first).  Then a segmented affine scan over the wavefront:
        //      (n, P, V) o (n', P', V') = (n + n', P + n'*dt*V + P', V + V'); a range that holds a
        //      trajectory start (flag) absorbs nothing from its left.  The lane's head starts from
        //      base + scale * (what the lanes before it contributed).
        //      MODE 0: a = u + e, base = x0 moved freely to the lane's first step.
        //      The response is LINEAR in a, so (MPPI_PK_NOMINAL) the part of the nominal controls is
        //      the same for every trajectory: MODE 1 (a = u) runs once per block and gives the
        //      nominal start states; MODE 2 (a = e, base = nominal) runs per tile -- one addition
        //      and the LDS reads of u less per normal, and the scan carries small numbers only. ---
        auto lane_start = [&](auto mode_tag, const float (&bp)[A], const float (&bv)[A],
                              float (&dps)[A], float (&dvs)[A]) {
            constexpr int MODE = decltype(mode_tag)::value;
            float S1[A], S2[A], S1h[A], S2h[A];
#pragma unroll
            for (int i = 0; i < A; ++i) { S1[i] = 0.f; S2[i] = 0.f; S1h[i] = 0.f; S2h[i] = 0.f; }
            float4 unext[BPG];                      // controls of the NEXT group: loaded a group ahead
            if constexpr (MODE < 2) {
#pragma unroll
                for (int b = 0; b < BPG; ++b) unext[b] = ulds[rbh + b];   // (group 0 is always head)
            }
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                // (group by group: without the scheduling fences hipcc hoists the LDS loads of all
                //  groups to the top of the pass and holds them in registers)
                MPPI_PK_FENCE();
                float u[BPG * 4];
                if constexpr (MODE < 2) {
#if !MPPI_PK_PREFETCH
                    {
                        const int rbc = ((gi < split) ? rbh : rbt) + gi * BPG;
#pragma unroll
                        for (int b = 0; b < BPG; ++b) unext[b] = ulds[rbc + b];
                    }
#endif
#pragma unroll
                    for (int b = 0; b < BPG; ++b) {
                        u[b * 4 + 0] = unext[b].x; u[b * 4 + 1] = unext[b].y;
                        u[b * 4 + 2] = unext[b].z; u[b * 4 + 3] = unext[b].w;
                    }
#if MPPI_PK_PREFETCH
                    if (gi + 1 < NG) {
                        const int rbn = ((gi + 1 < split) ? rbh : rbt) + (gi + 1) * BPG;
#pragma unroll
                        for (int b = 0; b < BPG; ++b) unext[b] = ulds[rbn + b];
                    }
#endif
                }
#pragma unroll
                for (int s = 0; s < SG; ++s) {
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        float a;
                        if constexpr (MODE == 0) a = u[s * A + i] + e[gi * BPG * 4 + s * A + i];
                        else if constexpr (MODE == 1) a = u[s * A + i];
                        else a = e[gi * BPG * 4 + s * A + i];
                        S1[i] += a;
                        S2[i] = fmaf((float)(gi * SG + s), a, S2[i]);
                    }
                }
                if (gi + 1 < NG && (split_mask & (1u << (gi + 1)))) {
                    const bool here = tail_slot && split == gi + 1;
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        S1h[i] = here ? S1[i] : S1h[i];
                        S2h[i] = here ? S2[i] : S2h[i];
                    }
                }
            }
            float Pz[A], Vz[A];
#pragma unroll
            for (int i = 0; i < A; ++i) {
                if (!tail_slot) { S1h[i] = S1[i]; S2h[i] = S2[i]; }
                const float S1t = S1[i] - S1h[i];
                const float S2t = (S2[i] - S2h[i]) - nh * S1t;      // steps counted from the boundary
                const float S1o = tail_slot ? S1t : S1h[i];
                const float S2o = tail_slot ? S2t : S2h[i];
                const float no = tail_slot ? nt : nh;
                Vz[i] = P.dt * S1o;
                Pz[i] = fmaf(P.dt2, fmaf(no - 1.0f, S1o, -S2o), P.B0 * S1o);
            }
#if MPPI_PK_QSCAN
            if constexpr (MODE >= 2) {
                float Qz[A];
#pragma unroll
                for (int i = 0; i < A; ++i) Qz[i] = fmaf(q_ref, Vz[i], Pz[i]);
#define MPPI_PK_QSUM(BIT, GETF)                                                     \
                {                                                                   \
                    float Ql[A], Vl[A];                                             \
                    _Pragma("unroll") for (int i = 0; i < A; ++i) {                 \
                        Ql[i] = GETF(Qz[i]);                                        \
                        Vl[i] = GETF(Vz[i]);                                        \
                    }                                                               \
                    if (absorb & (BIT)) {                                           \
                        MPPI_PK_NOFLATTEN();                                        \
                        _Pragma("unroll") for (int i = 0; i < A; ++i) {             \
                            Qz[i] = Ql[i] + Qz[i];                                  \
                            Vz[i] = Vl[i] + Vz[i];                                  \
                        }                                                           \
                    }                                                               \
                }
                MPPI_PK_QSUM(1u, dpp<MPPI_ROW_SHR(1)>)
                MPPI_PK_QSUM(2u, dpp<MPPI_ROW_SHR(2)>)
                MPPI_PK_QSUM(4u, dpp<MPPI_ROW_SHR(4)>)
                MPPI_PK_QSUM(8u, dpp<MPPI_ROW_SHR(8)>)
                MPPI_PK_QSUM(16u, (dpp_rows<kRowBcast15, 0xA>))
                MPPI_PK_QSUM(32u, (dpp_rows<kRowBcast31, 0xC>))
#undef MPPI_PK_QSUM
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    float Qex = dpp<kWaveShr1>(Qz[i]);
                    float Vex = dpp<kWaveShr1>(Vz[i]);
                    if (starts0) { Qex = 0.f; Vex = 0.f; }
                    const float Pex = fmaf(-c_ref, Vex, Qex);
                    if constexpr (MODE == 3) {      // the raw response: the caller adds the base
                        dps[i] = Pex;
                        dvs[i] = Vex;
                    } else {
                        dps[i] = fmaf(P.sp[i], Pex, bp[i]);
                        dvs[i] = fmaf(P.sv[i], Vex, bv[i]);
                    }
                }
                return;
            }
#endif
            {
                int nacc = n_out, flg = flag0;
                const int cr = lane & 15;
#define MPPI_PK_COMBINE(COND, GETF, GETI)                                           \
                {                                                                   \
                    const int nl = GETI(nacc);                                      \
                    const int fl = GETI(flg);                                       \
                    float Pl[A], Vl[A];                                             \
                    _Pragma("unroll") for (int i = 0; i < A; ++i) {                 \
                        Pl[i] = GETF(Pz[i]);                                        \
                        Vl[i] = GETF(Vz[i]);                                        \
                    }                                                               \
                    if ((COND) && flg == 0) {                                       \
                        const float tau = (float)nacc * P.dt;                       \
                        _Pragma("unroll") for (int i = 0; i < A; ++i) {             \
                            Pz[i] = fmaf(tau, Vl[i], Pl[i]) + Pz[i];                \
                            Vz[i] = Vl[i] + Vz[i];                                  \
                        }                                                           \
                        nacc += nl;                                                 \
                        flg = fl;                                                   \
                    }                                                               \
                }
                MPPI_PK_COMBINE(cr >= 1, dpp<MPPI_ROW_SHR(1)>, dppi<MPPI_ROW_SHR(1)>)
                MPPI_PK_COMBINE(cr >= 2, dpp<MPPI_ROW_SHR(2)>, dppi<MPPI_ROW_SHR(2)>)
                MPPI_PK_COMBINE(cr >= 4, dpp<MPPI_ROW_SHR(4)>, dppi<MPPI_ROW_SHR(4)>)
                MPPI_PK_COMBINE(cr >= 8, dpp<MPPI_ROW_SHR(8)>, dppi<MPPI_ROW_SHR(8)>)
                MPPI_PK_COMBINE((lane & 16) != 0, (dpp_rows<kRowBcast15, 0xA>), (dpp_rows_i<kRowBcast15, 0xA>))
                MPPI_PK_COMBINE((lane & 32) != 0, (dpp_rows<kRowBcast31, 0xC>), (dpp_rows_i<kRowBcast31, 0xC>))
#undef MPPI_PK_COMBINE
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float Pex = dpp<kWaveShr1>(Pz[i]);
                float Vex = dpp<kWaveShr1>(Vz[i]);
                if (starts0) { Pex = 0.f; Vex = 0.f; }
                dps[i] = fmaf(P.sp[i], Pex, bp[i]);
                dvs[i] = fmaf(P.sv[i], Vex, bv[i]);
            }
        };
        float dps[A], dvs[A];
#if MPPI_PK_NOMINAL && MPPI_PK_QSCAN
        // MODE 3 = MODE 2 without the base: the response of the lanes before this one to the NOISE
        // (Pex, Vex) needs no controls, so on the block's first tile it runs BEFORE the controls are
        // staged -- in a riding launch: while the combine role is still working on them (C2: 1.1 us
        // of the dependent chain behind the hand-over; C3: 0.8 us of the first tile)
        lane_start(std::integral_constant<int, 3>(), dps1, dvs0, dps, dvs);
        if constexpr (FIRST) {
            if constexpr (RIDE) ride_fetch_controls<A>(g, d, lambda, ulds, uclds, NBTs, TA);
            __syncthreads();             // U and lambda*inv_s*U are in LDS from here on
            MPPI_PK_STAMP(2);
            // the nominal trajectory's state at the lane's first step, once per block
            lane_start(std::integral_constant<int, 1>(), dps1, dvs0, dps_nom, dvs_nom);
        } else {
            MPPI_PK_STAMP(2);
        }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            dps[i] = fmaf(P.sp[i], dps[i], dps_nom[i]);
            dvs[i] = fmaf(P.sv[i], dvs[i], dvs_nom[i]);
        }
#else
        if constexpr (FIRST) {
            if constexpr (RIDE) ride_fetch_controls<A>(g, d, lambda, ulds, uclds, NBTs, TA);
            __syncthreads();             // U and lambda*inv_s*U are in LDS from here on
        }
        MPPI_PK_STAMP(2);
#if MPPI_PK_NOMINAL
        if constexpr (FIRST)    // the nominal trajectory's state at the lane's first step
            lane_start(std::integral_constant<int, 1>(), dps1, dvs0, dps_nom, dvs_nom);
        lane_start(std::integral_constant<int, 2>(), dps_nom, dvs_nom, dps, dvs);
#else
        lane_start(std::integral_constant<int, 0>(), dps1, dvs0, dps, dvs);
#endif
#endif
        MPPI_PK_STAMP(3);
        float4 unext[BPG];

        // ---- pass 2: dynamics + stage cost on the scaled state (src/point_mass_gpu.cu:97-107,
        //      src/cost.cu:42-55).  One cost accumulator per axis; where a trajectory ends inside
        //      a lane, its sum (+ Cost::final_cost, src/cost.cu:57-64) is set aside and the lane
        //      goes on from x0 with the next trajectory. ----------------------------------------
        // (the drift term cg of a velocity goal != 0 is compiled as a SEPARATE copy of the pass,
        //  chosen by one wave-uniform branch: tested inside the step loop, hipcc turns the test into
        //  an addition and a select per normal -- 96 VALU instructions per tile, 5 % of them)
        float cH = 0.0f, c_last = 0.0f;
        auto pass2 = [&](auto cg_tag) {
            constexpr bool CG = decltype(cg_tag)::value;
        float racc[A];
#pragma unroll
        for (int i = 0; i < A; ++i) racc[i] = 0.0f;
#if MPPI_PK_RACC3
        float raccu[A], raccv[A];
#pragma unroll
        for (int i = 0; i < A; ++i) { raccu[i] = 0.0f; raccv[i] = 0.0f; }
#endif
        float4 cnext[BPG];
#pragma unroll
        for (int b = 0; b < BPG; ++b) { unext[b] = ulds[rbh + b]; cnext[b] = uclds[rbh + b]; }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            MPPI_PK_FENCE();
#if !MPPI_PK_PREFETCH
            {
                const int rbc = ((gi < split) ? rbh : rbt) + gi * BPG;
#pragma unroll
                for (int b = 0; b < BPG; ++b) { unext[b] = ulds[rbc + b]; cnext[b] = uclds[rbc + b]; }
            }
#endif
            float u[BPG * 4], uc[BPG * 4];
#pragma unroll
            for (int b = 0; b < BPG; ++b) {
                u[b * 4 + 0] = unext[b].x; u[b * 4 + 1] = unext[b].y;
                u[b * 4 + 2] = unext[b].z; u[b * 4 + 3] = unext[b].w;
                uc[b * 4 + 0] = cnext[b].x; uc[b * 4 + 1] = cnext[b].y;
                uc[b * 4 + 2] = cnext[b].z; uc[b * 4 + 3] = cnext[b].w;
            }
#if MPPI_PK_PREFETCH
            if (gi + 1 < NG) {
                const int rbn = ((gi + 1 < split) ? rbh : rbt) + (gi + 1) * BPG;
#pragma unroll
                for (int b = 0; b < BPG; ++b) { unext[b] = ulds[rbn + b]; cnext[b] = uclds[rbn + b]; }
            }
#endif
#pragma unroll
            for (int s = 0; s < SG; ++s) {
                const float* es = &e[gi * BPG * 4 + s * A];
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    const float a = u[s * A + i] + es[i];
                    float pn = fmaf(P.k2[i], a, fmaf(P.k1[i], dvs[i], dps[i]));
                    if constexpr (CG) pn += P.cg[i];
                    if constexpr (RAGGED && SG > 1) {
                        // a step past the horizon in a trajectory's last group leaves the state
                        // where it is and adds no stage cost (its noise is zero already): plain
                        // selects on a per-lane mask -- as scalar branches on s >= pk_nlast the
                        // pass grew 1 500 register moves at its block boundaries
                        if (s > 0) {
                            const bool dead = dead_from[gi] <= s;
                            const float vn = fmaf(P.k3[i], a, dvs[i]);
                            pn = dead ? dps[i] : pn;
                            dvs[i] = dead ? dvs[i] : vn;
                            dps[i] = pn;
                            const float pc = dead ? 0.0f : pn, vc = dead ? 0.0f : vn;
#if MPPI_PK_RACC3
                            raccu[i] = fmaf(uc[s * A + i], es[i], raccu[i]);
                            racc[i] = fmaf(pc, pc, racc[i]);
                            raccv[i] = fmaf(vc, vc, raccv[i]);
#else
                            racc[i] = fmaf(uc[s * A + i], es[i], racc[i]);
                            racc[i] = fmaf(pc, pc, racc[i]);
                            racc[i] = fmaf(vc, vc, racc[i]);
#endif
                            continue;
                        }
                    }
                    dvs[i] = fmaf(P.k3[i], a, dvs[i]);
                    dps[i] = pn;
#if MPPI_PK_RACC3
                    raccu[i] = fmaf(uc[s * A + i], es[i], raccu[i]);
                    racc[i] = fmaf(pn, pn, racc[i]);
                    raccv[i] = fmaf(dvs[i], dvs[i], raccv[i]);
#else
                    racc[i] = fmaf(uc[s * A + i], es[i], racc[i]);
                    racc[i] = fmaf(pn, pn, racc[i]);
                    racc[i] = fmaf(dvs[i], dvs[i], racc[i]);
#endif
                }
            }
            if (gi + 1 < NG && (split_mask & (1u << (gi + 1)))) {
                const bool here = tail_slot && split == gi + 1;
                float tot = 0.0f;
#if MPPI_PK_RACC3
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    tot += fmaf(dps[i], dps[i], fmaf(dvs[i], dvs[i], racc[i] + (raccu[i] + raccv[i])));
                    raccu[i] = here ? 0.0f : raccu[i];
                    raccv[i] = here ? 0.0f : raccv[i];
                }
#else
#pragma unroll
                for (int i = 0; i < A; ++i)
                    tot += fmaf(dps[i], dps[i], fmaf(dvs[i], dvs[i], racc[i]));
#endif
                cH = here ? tot : cH;
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    racc[i] = here ? 0.0f : racc[i];
                    dps[i] = here ? dps0[i] : dps[i];
                    dvs[i] = here ? dvs0[i] : dvs[i];
                }
            }
        }
        {
            const bool term = !tail_slot && ends_head;    // the trajectory ends with the lane
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float fc = fmaf(dps[i], dps[i], dvs[i] * dvs[i]);
#if MPPI_PK_RACC3
                racc[i] += raccu[i] + raccv[i];
#endif
                c_last += racc[i] + (term ? fc : 0.0f);
            }
        }
        };
        if (has_cg) pass2(std::true_type());
        else pass2(std::false_type());
        const float cA = tail_slot ? cH : c_last;         // my part of trajectory j0
        MPPI_PK_STAMP(4);

        // ---- trajectory costs: segmented sum of the lane parts over the wavefront -------------
        float ctA, ctB;
        {
            float cz = c_last;                            // the range handed on (tail, or all)
#if MPPI_PK_QSCAN
#define MPPI_PK_CSUMQ(BIT, GETF)                                                    \
            {                                                                       \
                const float cl = GETF(cz);                                          \
                if (absorb & (BIT)) { MPPI_PK_NOFLATTEN(); cz = cl + cz; }         \
            }
            MPPI_PK_CSUMQ(1u, dpp<MPPI_ROW_SHR(1)>)
            MPPI_PK_CSUMQ(2u, dpp<MPPI_ROW_SHR(2)>)
            MPPI_PK_CSUMQ(4u, dpp<MPPI_ROW_SHR(4)>)
            MPPI_PK_CSUMQ(8u, dpp<MPPI_ROW_SHR(8)>)
            MPPI_PK_CSUMQ(16u, (dpp_rows<kRowBcast15, 0xA>))
            MPPI_PK_CSUMQ(32u, (dpp_rows<kRowBcast31, 0xC>))
#undef MPPI_PK_CSUMQ
#else
            int flg = flag0;
            const int cr = lane & 15;
#define MPPI_PK_CSUM(COND, GETF, GETI)                                              \
            {                                                                       \
                const float cl = GETF(cz);                                          \
                const int fl = GETI(flg);                                           \
                if ((COND) && flg == 0) { cz = cl + cz; flg = fl; }                 \
            }
            MPPI_PK_CSUM(cr >= 1, dpp<MPPI_ROW_SHR(1)>, dppi<MPPI_ROW_SHR(1)>)
            MPPI_PK_CSUM(cr >= 2, dpp<MPPI_ROW_SHR(2)>, dppi<MPPI_ROW_SHR(2)>)
            MPPI_PK_CSUM(cr >= 4, dpp<MPPI_ROW_SHR(4)>, dppi<MPPI_ROW_SHR(4)>)
            MPPI_PK_CSUM(cr >= 8, dpp<MPPI_ROW_SHR(8)>, dppi<MPPI_ROW_SHR(8)>)
            MPPI_PK_CSUM((lane & 16) != 0, (dpp_rows<kRowBcast15, 0xA>), (dpp_rows_i<kRowBcast15, 0xA>))
            MPPI_PK_CSUM((lane & 32) != 0, (dpp_rows<kRowBcast31, 0xC>), (dpp_rows_i<kRowBcast31, 0xC>))
#undef MPPI_PK_CSUM
#endif
            float cex = dpp<kWaveShr1>(cz);               // what the lanes before me hold of j0
            if (starts0) cex = 0.0f;
            const float tot_h = cex + cA;                 // complete where trajectory j0 ends
            float* ct = ctab + wave * (TPW + 2);
            if (ends_head && j0 < TPW) ct[j0] = tot_h;
            if (ends_head && valid_h) cost_out[kh] = tot_h;
            __builtin_amdgcn_wave_barrier();              // (same wave: LDS operations stay in order)
            ctA = ct[j0];
            ctB = ct[j0 + 1];
        }

        // ---- wave tail: every WAVE keeps its own running (min, exp-sum, weighted noise sums):
        //      no block barrier in the tile loop.  The weighted noise of a lane is ACCUMULATED in
        //      the lane's own LDS slot, rescaled when the wave's running minimum drops; slots of
        //      different lanes / waves meet once, when the block has walked all its tiles. ------
        const float m_w = wave_min((ends_head && valid_h) ? ctA : INFINITY);
        if (m_w < INFINITY) {                            // (wave-uniform) a tile past K adds nothing
            const float Mn = fminf(Mw, m_w);
            const float alpha = FIRST ? 0.0f : expf(-inv_lambda * (Mw - Mn));   // 1 if Mn == Mw
            const float wA = valid_h ? expf(-inv_lambda * (ctA - Mn)) : 0.0f;
            const float wB = valid_t ? expf(-inv_lambda * (ctB - Mn)) : 0.0f;
            const float sw = wave_sum(ends_head ? wA : 0.0f);
            Sw = FIRST ? sw : fmaf(alpha, Sw, sw);
            Mw = Mn;
            MPPI_PK_STAMP(5);
            const unsigned long long kgh = (unsigned long long)(k_offset + kh);
            const float wAn =
                ((long long)kgh < k_cover && ((unsigned int)kgh & cover_and) == 0u) ? wA : 0.0f;
            const float wBn =
                ((long long)(kgh + 1) < k_cover && ((unsigned int)(kgh + 1) & cover_and) == 0u) ? wB : 0.0f;
            if constexpr (FIRST) {                       // first tile of the wave: plain write
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const float wg = (gi < split) ? wAn : wBn;
#pragma unroll
                    for (int b = 0; b < BPG; ++b) {
                        const int q = gi * BPG + b;
                        bw[q * kPkRow] = make_float4(wg * e[q * 4], wg * e[q * 4 + 1], wg * e[q * 4 + 2],
                                                 wg * e[q * 4 + 3]);
                    }
                }
            } else if (alpha == 1.0f) {                  // running minimum unchanged: accumulate
                // (measured and dropped: skipping the accumulate of a tile whose weights are all
                //  exactly zero -- same bits -- changes nothing even at lambda = 0.05, where all but
                //  one tile skip: 66.3 against 66.1 us at C3; the LDS traffic of this tail hides
                //  under the other wave's Philox pass.  tools/lambda_speed.py)
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const float wg = (gi < split) ? wAn : wBn;
#pragma unroll
                    for (int b = 0; b < BPG; ++b) {
                        const int q = gi * BPG + b;
                        const float4 o = bw[q * kPkRow];
                        bw[q * kPkRow] = make_float4(fmaf(wg, e[q * 4], o.x), fmaf(wg, e[q * 4 + 1], o.y),
                                                 fmaf(wg, e[q * 4 + 2], o.z), fmaf(wg, e[q * 4 + 3], o.w));
                    }
                }
            } else {                                     // a new minimum: rescale what is there
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const float wg = (gi < split) ? wAn : wBn;
#pragma unroll
                    for (int b = 0; b < BPG; ++b) {
                        const int q = gi * BPG + b;
                        const float4 o = bw[q * kPkRow];
                        bw[q * kPkRow] = make_float4(fmaf(wg, e[q * 4], alpha * o.x), fmaf(wg, e[q * 4 + 1], alpha * o.y),
                                                 fmaf(wg, e[q * 4 + 2], alpha * o.z), fmaf(wg, e[q * 4 + 3], alpha * o.w));
                    }
                }
            }
            MPPI_PK_STAMP(6); MPPI_PK_STAMP(7); MPPI_PK_STAMP(8);
        }
#ifdef MPPI_TRACE
        ++tile_no;
#endif
    };
    {
        int tb = bid;
        if (tb < n_tileblk) {        // (always: the grid never has more blocks than tile groups)
            tile_body(tb, std::true_type());
            tb += nblk;
        }
        for (; tb < n_tileblk; tb += nblk) tile_body(tb, std::false_type());
    }
#undef MPPI_PK_STAMP
    MPPI_STAMP(9);

    // ---- the block's partial: the four waves' running sums meet here, once ------------------------
    //      M = min M_w, r_w = exp(-(M_w - M)/lambda), S = sum r_w S_w, N = sum r_w N_w
    // First every WAVE adds up its own TPW trajectories, block by block, in trajectory order -- no
    // barrier needed (its own LDS rows, its own instruction stream), so a wave that is done early
    // does it while the others still compute, and all 256 threads share the gathers -- and leaves
    // the sum where trajectory 0's block sat.  (Round 2 let thread m gather all 4 x TPW slots of
    // block m behind the barrier: one or two waves walking 20 computed addresses each, 1.9-2.3 us
    // per block at any size -- profiles/r03_trace_regions.txt "merge + partial store".)
    if (Mw < INFINITY) {                    // (wave-uniform) a wave that saw nothing wrote nothing
        __builtin_amdgcn_wave_barrier();
        float4* const wbuf = buf + wave * NQ * kPkRow;
        for (int m = lane; m < NBT; m += 64) {
            const int r = m / BPG, b = m - r * BPG;          // group and block of the trajectory
            const int ln0 = r / NG;
            float4* const dst = wbuf + ((r - ln0 * NG) * BPG + b) * kPkRow + ln0;
            float4 acc = *dst;                                // trajectory 0
            for (int jj = 1; jj < TPW; ++jj) {
                const int sl = jj * NGT + r;
                const int ln = sl / NG;
                const float4 v = wbuf[((sl - ln * NG) * BPG + b) * kPkRow + ln];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *dst = acc;
        }
    }
    if (lane == 0) {
        misc[wave] = Mw;
        misc[4 + wave] = Sw;
    }
    __syncthreads();
    MPPI_STAMP(12);
    {
        const float M = fminf(fminf(misc[0], misc[1]), fminf(misc[2], misc[3]));
        float rw[4];
        float S = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            rw[w] = (misc[w] < INFINITY) ? expf(-inv_lambda * (misc[w] - M)) : 0.0f;
            S = fmaf(rw[w], misc[4 + w], S);
        }
        // thread m: Philox block m of the horizon over the 4 waves, in wave order
        float4* Nout = reinterpret_cast<float4*>(g.part_N + (size_t)bid * Nrow);
        for (int m = threadIdx.x; m < NBT; m += kRolloutThreads) {
            const int r = m / BPG, b = m - r * BPG;
            const int ln0 = r / NG;
            const int off = ((r - ln0 * NG) * BPG + b) * kPkRow + ln0;
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rw[w] != 0.0f) a = buf[w * NQ * kPkRow + off];      // (block-uniform)
                tot.x = fmaf(rw[w], a.x, tot.x); tot.y = fmaf(rw[w], a.y, tot.y);
                tot.z = fmaf(rw[w], a.z, tot.z); tot.w = fmaf(rw[w], a.w, tot.w);
            }
            Nout[m] = tot;
        }
        if (threadIdx.x == 0) {
            g.part_m[bid] = M;
            g.part_s[bid] = S;
        }
    }
    MPPI_STAMP(10);
}

// The noise of a lane stays in registers across all passes: two waves per SIMD where that is more
// than 32 normals (<= 256 VGPRs), else three (<= 168).
template <int A, int NG>
constexpr int packed_min_waves()
{
#ifdef MPPI_PK_WAVES
    return MPPI_PK_WAVES;
#else
    return NG * Dim<A>::BPG * 4 <= 32 ? 3 : 2;
#endif
}

template <int A, int NG, bool SAMPLE, bool RAGGED>
__global__ void __launch_bounds__(kRolloutThreads, (packed_min_waves<A, NG>()))
k_rollout_packed(const RolloutHot h)
{
    packed_body<A, NG, SAMPLE, false, RAGGED>(h, DeferredCombine());
}

// the same with the previous solve's combine riding at the front of the grid (see k_rollout_ride)
template <int A, int NG, bool SAMPLE, bool RAGGED>
__global__ void __launch_bounds__(kRolloutThreads, (packed_min_waves<A, NG>()))
k_rollout_packed_ride(const RolloutHot h, const DeferredCombine d)
{
    if ((int)blockIdx.x < d.n_blocks) {
        __builtin_amdgcn_s_setprio(3);
        MPPI_STAMP(0);
        extern __shared__ __align__(16) unsigned char smem_raw[];
        combine_body<kRolloutThreads, kSmallCombineNR>(
            d.c, (int)blockIdx.x,
            carve_combine_smem<kRolloutThreads>(reinterpret_cast<float*>(smem_raw)));
        MPPI_STAMP(10);
        return;
    }
    packed_body<A, NG, SAMPLE, true, RAGGED>(h, d);
}

template <int A, int NG>
size_t packed_lds_bytes_t(int NBT /* blocks of the controls staged: NGT*BPG */, int TPW)
{
    return (size_t)NBT * 2 * 16 + (size_t)4 * kPkRow * NG * Dim<A>::BPG * 16
           + (size_t)(8 + 4 * (TPW + 2)) * sizeof(float);
}

// the instantiation a launch geometry runs: sampled / injected noise, with / without a riding
// combine, whole-groups / ragged horizon
template <int A, int NG>
const void* packed_kernel(bool sample, bool ride, bool ragged)
{
    switch ((sample ? 1 : 0) | (ride ? 2 : 0) | (ragged ? 4 : 0)) {
        case 0: return reinterpret_cast<const void*>(&k_rollout_packed<A, NG, false, false>);
        case 1: return reinterpret_cast<const void*>(&k_rollout_packed<A, NG, true, false>);
        case 2: return reinterpret_cast<const void*>(&k_rollout_packed_ride<A, NG, false, false>);
        case 3: return reinterpret_cast<const void*>(&k_rollout_packed_ride<A, NG, true, false>);
        case 4: return reinterpret_cast<const void*>(&k_rollout_packed<A, NG, false, true>);
        case 5: return reinterpret_cast<const void*>(&k_rollout_packed<A, NG, true, true>);
        case 6: return reinterpret_cast<const void*>(&k_rollout_packed_ride<A, NG, false, true>);
        default: return reinterpret_cast<const void*>(&k_rollout_packed_ride<A, NG, true, true>);
    }
}

template <int A, int NG>
hipError_t launch_packed_t(bool sample, int grid, const RolloutArgs& a, const DeferredCombine& d,
                           hipStream_t st, LaunchTiming tm)
{
    size_t lds = packed_lds_bytes_t<A, NG>(a.NBTp, a.TPW);
    const bool ride = d.n_blocks > 0;
    if (ride && lds < combine_small_lds_bytes()) lds = combine_small_lds_bytes();
    RolloutHot h = make_hot(a);
    DeferredCombine dd = d;
    const dim3 g(grid + d.n_blocks), b(kRolloutThreads);
    const void* fn = packed_kernel<A, NG>(sample, ride, a.pk_nlast < Dim<A>::SG);
    if (lds > kDefaultLdsBytes)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* args[2] = {&h, &dd};          // (the plain kernel takes the first only)
    if (tm.start && tm.stop)
        return hipExtLaunchKernel(fn, g, b, args, lds, st, tm.start, tm.stop, 0);
    return hipLaunchKernel(fn, g, b, args, lds, st);
}

template <int A, int NG>      // ride: the riding variant (see fused_blocks_per_cu_t)
int packed_blocks_per_cu_t(bool sample, size_t lds, bool ride, bool ragged)
{
    int n = 0;
    if (ride && lds < combine_small_lds_bytes()) lds = combine_small_lds_bytes();
    const hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &n, packed_kernel<A, NG>(sample, ride, ragged), kRolloutThreads, lds);
    return rc == hipSuccess ? n : 0;
}

// instantiated groups-per-lane values: chosen so that T = 200 packs well (see packed_ng_list)
template <int A> struct PackedNG;
template <> struct PackedNG<1> { static constexpr int list[] = {4, 0}; };
template <> struct PackedNG<2> { static constexpr int list[] = {5, 8, 0}; };
template <> struct PackedNG<3> { static constexpr int list[] = {4, 0}; };
template <> struct PackedNG<4> { static constexpr int list[] = {10, 0}; };

template <int A>
hipError_t launch_packed_a(int NG, bool sample, int grid, const RolloutArgs& a,
                           const DeferredCombine& d, hipStream_t st, LaunchTiming tm)
{
    if constexpr (A == 1) {
        if (NG == 4) return launch_packed_t<1, 4>(sample, grid, a, d, st, tm);
    } else if constexpr (A == 2) {
        if (NG == 5) return launch_packed_t<2, 5>(sample, grid, a, d, st, tm);
        if (NG == 8) return launch_packed_t<2, 8>(sample, grid, a, d, st, tm);
    } else if constexpr (A == 3) {
        if (NG == 4) return launch_packed_t<3, 4>(sample, grid, a, d, st, tm);
    } else {
        if (NG == 10) return launch_packed_t<4, 10>(sample, grid, a, d, st, tm);
    }
    return hipErrorInvalidValue;
}

template <int A>
int packed_blocks_per_cu_a(int NG, bool sample, size_t lds, bool ride, bool ragged)
{
    if constexpr (A == 1) {
        if (NG == 4) return packed_blocks_per_cu_t<1, 4>(sample, lds, ride, ragged);
    } else if constexpr (A == 2) {
        if (NG == 5) return packed_blocks_per_cu_t<2, 5>(sample, lds, ride, ragged);
        if (NG == 8) return packed_blocks_per_cu_t<2, 8>(sample, lds, ride, ragged);
    } else if constexpr (A == 3) {
        if (NG == 4) return packed_blocks_per_cu_t<3, 4>(sample, lds, ride, ragged);
    } else {
        if (NG == 10) return packed_blocks_per_cu_t<4, 10>(sample, lds, ride, ragged);
    }
    return 0;
}

template <int A>
size_t packed_lds_bytes_a(int NG, int NBT, int TPW)
{
    if constexpr (A == 1) {
        if (NG == 4) return packed_lds_bytes_t<1, 4>(NBT, TPW);
    } else if constexpr (A == 2) {
        if (NG == 5) return packed_lds_bytes_t<2, 5>(NBT, TPW);
        if (NG == 8) return packed_lds_bytes_t<2, 8>(NBT, TPW);
    } else if constexpr (A == 3) {
        if (NG == 4) return packed_lds_bytes_t<3, 4>(NBT, TPW);
    } else {
        if (NG == 10) return packed_lds_bytes_t<4, 10>(NBT, TPW);
    }
    return 0;
}

}  // namespace mppi
