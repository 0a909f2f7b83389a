// rollout_fused_impl.hpp -- the fused rollout kernel template and its per-act_dim launcher.
// Instantiated once per act_dim in rollout_fused_a{1,2,3,4}.hip so the four translation units
// compile in parallel.
#pragma once
#include "device_common.hpp"
#include "combine_impl.hpp"
#include <type_traits>

#ifndef MPPI_FUSED_SCALED
#define MPPI_FUSED_SCALED 1    // pass 2 on the scaled state (7 instead of 11 VALU per normal)
#endif
#ifndef MPPI_FUSED_DIRECT_FOLD
#define MPPI_FUSED_DIRECT_FOLD 1   // the block's last tile folds straight into the global partial
#endif
#ifndef MPPI_FUSED_LATE_STORE
#define MPPI_FUSED_LATE_STORE 0    // riding launch, first tile: the noise stores are issued AFTER the
#endif                             // controls have arrived (the polls do not queue behind them)
#ifndef MPPI_FUSED_PRIO
#define MPPI_FUSED_PRIO 2      // s_setprio of the passes after the Philox pass (which runs at 0)
#endif

namespace mppi {

// ------------------------------------------------------------------------------------------
// Fused rollout: 2^LOGC lanes per trajectory, NG groups per lane, noise resident in registers.
//
// Arithmetic: unlike the strict kernel this one lets products feed additions as FMAs
// (explicit fmaf), as nvcc does by default for the reference's device code; the chunk
// hand-over and the cost tree re-associate anyway, so its results are the same few-ulp
// class either way (tests state the bound).
//
// Launch latency: at K = 1e4 the whole kernel is ~15 us, so every dependent memory round trip
// in the prologue (~1 us each) shows.  What the first instructions need travels BY VALUE in
// the kernel arguments (RolloutHot: pointers, Philox key/offset, geometry, x0); the rest of the
// descriptor and the nominal controls are fetched while the Philox blocks are computed, and
// the only barrier of the prologue sits AFTER noise generation.
// ------------------------------------------------------------------------------------------
template <int A>
struct LaneParams {     // wave-uniform problem constants, deliberately held in VGPRs: kept in
#if MPPI_FUSED_SCALED   // SGPRs they and the launch geometry exceed the 102-SGPR file, and
    float k1[A], k2[A], k3[A], cg[A];   // every spilled scalar costs a v_readlane + s_nop in
#else                                   // the hot loop
    float goal[2 * A];
    float w[2 * A];
#endif
    float sigma[A];
    float dt, B0, dt2;
};

// EXACT: the chunk is exactly NG groups long (ng == NG), so the group checks below fold away.
// Instantiated where it was measured to pay (kExactGroupsPays): hipcc's schedule of the single
// straight-line pass is 3.4 % faster for act_dim 2 (NG = 7) and 8 % SLOWER for act_dim 3 (NG = 4).
template <int A>
constexpr bool kExactGroupsPays = (A == 2);

template <int A, int NG, bool SAMPLE, int LOGC, bool RIDE, bool EXACT>
__device__ __forceinline__ void fused_body(const RolloutHot& h, const DeferredCombine& d)
{
    // RIDE: d.n_blocks combine-role blocks come first in the grid, the rollout blocks follow and
    // the controls are not final yet at kernel entry
    const int bid = RIDE ? (int)blockIdx.x - d.n_blocks : (int)blockIdx.x;
    const int nblk = RIDE ? (int)gridDim.x - d.n_blocks : (int)gridDim.x;
    constexpr bool deferred = RIDE;
    constexpr int SG = Dim<A>::SG;
    constexpr int BPG = Dim<A>::BPG;
    constexpr int NE = NG * BPG * 4;          // normals held per lane
    constexpr int C = 1 << LOGC;

    // everything the tile loop touches is read from the by-value arguments ONCE, up front
    const int ng = h.ng, nq = h.nq;
    const int K = h.K, T = h.T, TA = h.TA, NBT = h.NBT, NBTp = h.NBTp;
    const int c_last = h.c_last, n_last = h.n_last, n_tileblk = h.n_tileblk, L = h.L;
    const long long k_offset = h.k_offset;
    const unsigned long long seed = h.seed;
    float* const Eint = h.Eint;
    const RolloutArgs& g = *h.rest;           // cold part of the descriptor (device memory)

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* ulds = reinterpret_cast<float4*>(smem_raw);          // [NBTp] U in block layout
    float4* uclds = ulds + NBTp;                                 // [NBTp] lambda*inv_s*U
    const int TAp = C * nq * 4;
    float* wsum = reinterpret_cast<float*>(uclds + NBTp);        // [4][TAp]
    float* nrun = wsum + 4 * TAp;                                // [TAp]
    float* misc = nrun + TAp;                                    // [8]
    float* snap = misc + 8 + threadIdx.x;                        // [1 + 2A][256]: this thread's slot

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & (C - 1);
    MPPI_STAMP(0);

    // ---- issue the loads of the nominal controls and of the cold constants; they complete
    //      while the first tile's Philox blocks are computed ----------------------------------
    const float lambda = g.lambda, inv_lambda = g.inv_lambda;
    auto stage_controls_lds = [&]() {
        const float* Uin = h.U_in;
        for (int b = threadIdx.x; b < NBTp; b += kRolloutThreads) {
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
    };
    if constexpr (!deferred) stage_controls_lds();
    LaneParams<A> P;
    float x0p[A], x0v[A];
#if MPPI_FUSED_SCALED
#pragma unroll
    for (int i = 0; i < A; ++i) {
        P.k1[i] = to_vgpr(g.fs_k1[i]); P.k2[i] = to_vgpr(g.fs_k2[i]);
        P.k3[i] = to_vgpr(g.fs_k3[i]); P.cg[i] = to_vgpr(g.fs_cg[i]);
    }
    const bool has_cg = g.pk_has_cg != 0;     // wave-uniform: a velocity goal != 0 drifts d_p
#else
#pragma unroll
    for (int i = 0; i < 2 * A; ++i) { P.goal[i] = to_vgpr(g.goal[i]); P.w[i] = to_vgpr(g.w[i]); }
#endif
#pragma unroll
    for (int i = 0; i < A; ++i) {
        P.sigma[i] = to_vgpr(g.sigma[i]);
        x0p[i] = to_vgpr(h.x0[i]);
        x0v[i] = to_vgpr(h.x0[A + i]);
    }
    const bool sigma_one = g.sigma_one != 0;  // one sigma for all axes: it sits in the radius factor
    const float noise_r2c = to_vgpr(g.noise_r2c);
    P.dt = to_vgpr(g.dt);
    P.B0 = to_vgpr(g.B0);
    P.dt2 = P.dt * P.dt;
    const long long k_cover = g.k_cover;
    const unsigned int cover_and = g.cover_and;
    const int store_e = g.store_e;           // wave-uniform: 0 not materialised, 1 write-through, 2 non-temporal
    float* const cost_out = g.cost;

    // chunk geometry of this lane (same for every tile group)
    const int ns_own = (c < c_last) ? L : (c == c_last ? n_last : 0);
    const int nbefore = min(c * L, T);
    const unsigned long long blk0 = h.solve_idx * (unsigned long long)NBT
                                    + (unsigned long long)(c * nq);
    const float Lm1 = (float)(L - 1);

    RunState rs{INFINITY, 0.0f};
    bool first = true;
    float* const Nout = g.part_N + (size_t)bid * h.Nrow;     // this block's partial sums

    for (int tb = bid; tb < n_tileblk; tb += nblk) {
        const long long gid = (long long)tb * kRolloutThreads + threadIdx.x;
        const long long kloc = gid >> LOGC;
        const bool valid = kloc < K;
        const unsigned long long kglob = (unsigned long long)(k_offset + kloc);
        const size_t tile = (size_t)(gid >> 6);
        float* etile = Eint + ((tile * nq) * 64 + lane) * 4;      // + q*256 floats per block
        // the tile's slice of E as a raw buffer (base is wave-uniform: one tile = one wavefront)
        __amdgpu_buffer_rsrc_t e_rsrc;
        {
            const unsigned long long tb = reinterpret_cast<unsigned long long>(Eint + (tile * nq) * 256);
            const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)tb);
            const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(tb >> 32));
            e_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, nq * 1024, 0x00020000);
        }

        // ---- pass 1a: draw (or load) the chunk's noise into registers and store it ----------
        // `gi < ng` is wave-uniform.  Left to itself hipcc hoists the NG comparisons out of the tile
        // loop as lane masks and re-tests each with a v_cndmask + v_cmp pair (VALU, the bound
        // resource); an opaque scalar copy per pass keeps them s_cmp + s_cbranch.
        int ngs = EXACT ? NG : ng;
        if constexpr (!EXACT) asm volatile("" : "+s"(ngs));
        // instruction priority by phase (see rollout_packed_impl.hpp): the Philox pass yields to
        // the dependent chains of the passes after it
#if MPPI_FUSED_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        float e[NE];
        // (groups gi >= ng of a template larger than the chunk are never read: every use below
        //  sits under the same wave-uniform `gi < ng`, so e[] needs no initialisation)
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi < ngs) {
#pragma unroll
                for (int j = 0; j < BPG; ++j) {
                    const int q = gi * BPG + j;
                    float* eq = &e[q * 4];
                    if constexpr (SAMPLE) {
                        const uint4 r = PhiloxAt::block(blk0 + (unsigned long long)q, kglob, seed);
                        scaled_normals4<A>(r, (q * 4) % A, sigma_one, noise_r2c, P.sigma, eq);
                        if (store_e && c * nq + q < NBT &&   // blocks past the horizon are not stored
                            !(MPPI_FUSED_LATE_STORE && RIDE && first)) {
                            // Write-through store (sc0 sc1): E is not read again by this
                            // launch, and what a plain store leaves dirty in the XCD L2s -- all
                            // 16 MB at C2 -- is written back at the END of the kernel, where
                            // nothing overlaps it (measured: 1.5 us of a 15 us launch).
                            // (a raw buffer store with the sc0 sc1 cache policy: a compiler-
                            //  known instruction, so hazards and wait counts stay hipcc's
                            //  business -- an inline-asm global_store bypasses both and corrupted
                            //  one word of some blocks; a volatile store is followed by a full
                            //  wait (+34 % at 3-D K = 1e5); 8-byte system-scope atomic stores
                            //  write half sectors (2x slower at K = 1e6))
                            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                            const v4u val = {__float_as_uint(eq[0]), __float_as_uint(eq[1]),
                                             __float_as_uint(eq[2]), __float_as_uint(eq[3])};
                            if (store_e == 2)       // footprint beyond the memory-side cache: non-temporal
                                __builtin_amdgcn_raw_buffer_store_b128(val, e_rsrc, lane * 16, q * 1024, 2);
                            else
                                __builtin_amdgcn_raw_buffer_store_b128(val, e_rsrc, lane * 16, q * 1024,
                                                                       17 /* sc0 | sc1 */);
                        }
                    } else {
                        const float4 t = *reinterpret_cast<const float4*>(etile + (size_t)q * 256);
                        eq[0] = t.x; eq[1] = t.y; eq[2] = t.z; eq[3] = t.w;
                    }
                }
            }
        }
        if (first) MPPI_STAMP(1);
#if MPPI_FUSED_PRIO
        __builtin_amdgcn_s_setprio(MPPI_FUSED_PRIO);
#endif
        if (first) {
            if constexpr (deferred) ride_fetch_controls<A>(g, d, lambda, ulds, uclds, NBTp, TA);
            __syncthreads();             // U and lambda*inv_s*U are in LDS from here on
#if MPPI_FUSED_LATE_STORE
            if constexpr (RIDE && SAMPLE) {
                if (store_e) {
#pragma unroll
                    for (int gi = 0; gi < NG; ++gi) {
                        if (gi < ngs) {
#pragma unroll
                            for (int j = 0; j < BPG; ++j) {
                                const int q = gi * BPG + j;
                                if (c * nq + q < NBT) {
                                    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                                    const v4u val = {__float_as_uint(e[q * 4]), __float_as_uint(e[q * 4 + 1]),
                                                     __float_as_uint(e[q * 4 + 2]), __float_as_uint(e[q * 4 + 3])};
                                    __builtin_amdgcn_raw_buffer_store_b128(val, e_rsrc, lane * 16, q * 1024,
                                                                           17 /* sc0 | sc1 */);
                                }
                            }
                        }
                    }
                }
            }
#endif
        }
        if (first) MPPI_STAMP(2);

        // ---- pass 1b: the chunk's zero-state response is two weighted sums of a = u + e:
        //      V = dt*S1,  P = B0*S1 + dt^2*((L-1)*S1 - S2),  S1 = sum a_j, S2 = sum j*a_j.
        //      No masking: what a partial or empty chunk adds past the horizon is not used. ---
        float S1[A], S2[A];
        if constexpr (!EXACT) asm volatile("" : "+s"(ngs));
#pragma unroll
        for (int i = 0; i < A; ++i) { S1[i] = 0.f; S2[i] = 0.f; }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi < ngs) {
                float u[BPG * 4];
#pragma unroll
                for (int j = 0; j < BPG; ++j) {
                    const float4 u4 = ulds[c * nq + gi * BPG + j];
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
        //      (n + n', P + n'*dt*V + P', V + V').  All DPP: row_shr inside the 16-lane rows,
        //      then row_bcast:15 / row_bcast:31 carry the row totals across (C = 32, 64). ----
        {
            int nacc = ns_own;
            const int cr = (LOGC <= 4) ? c : (lane & 15);      // position inside the DPP row
#define MPPI_COMBINE(COND, GETF, GETI)                                              \
            {                                                                       \
                const int nl = GETI(nacc);                                          \
                float Pl[A], Vl[A];                                                 \
                _Pragma("unroll") for (int i = 0; i < A; ++i) {                     \
                    Pl[i] = GETF(Pz[i]);                                            \
                    Vl[i] = GETF(Vz[i]);                                            \
                }                                                                   \
                if (COND) {                                                         \
                    const float tau = (float)nacc * P.dt;                           \
                    _Pragma("unroll") for (int i = 0; i < A; ++i) {                 \
                        Pz[i] = fmaf(tau, Vl[i], Pl[i]) + Pz[i];                    \
                        Vz[i] = Vl[i] + Vz[i];                                      \
                    }                                                               \
                    nacc += nl;                                                     \
                }                                                                   \
            }
            if constexpr (C > 1) MPPI_COMBINE(cr >= 1, dpp<MPPI_ROW_SHR(1)>, dppi<MPPI_ROW_SHR(1)>)
            if constexpr (C > 2) MPPI_COMBINE(cr >= 2, dpp<MPPI_ROW_SHR(2)>, dppi<MPPI_ROW_SHR(2)>)
            if constexpr (C > 4) MPPI_COMBINE(cr >= 4, dpp<MPPI_ROW_SHR(4)>, dppi<MPPI_ROW_SHR(4)>)
            if constexpr (C > 8) MPPI_COMBINE(cr >= 8, dpp<MPPI_ROW_SHR(8)>, dppi<MPPI_ROW_SHR(8)>)
            if constexpr (C > 16)
                MPPI_COMBINE((lane & 16) != 0, (dpp_rows<kRowBcast15, 0xA>), (dpp_rows_i<kRowBcast15, 0xA>))
            if constexpr (C > 32)
                MPPI_COMBINE((lane & 32) != 0, (dpp_rows<kRowBcast31, 0xC>), (dpp_rows_i<kRowBcast31, 0xC>))
#undef MPPI_COMBINE
        }
        if (first) MPPI_STAMP(3);
        float p[A], v[A];
        {
            const float tau0 = (float)nbefore * P.dt;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float Pex = 0.f, Vex = 0.f;
                if constexpr (C > 1) {
                    if constexpr (LOGC <= 4) {
                        Pex = dpp<MPPI_ROW_SHR(1)>(Pz[i]);
                        Vex = dpp<MPPI_ROW_SHR(1)>(Vz[i]);
                    } else {
                        Pex = dpp<kWaveShr1>(Pz[i]);
                        Vex = dpp<kWaveShr1>(Vz[i]);
                    }
                    if (c == 0) { Pex = 0.f; Vex = 0.f; }
                }
                p[i] = fmaf(tau0, x0v[i], x0p[i]) + Pex;
                v[i] = x0v[i] + Vex;
            }
        }

#if MPPI_FUSED_SCALED
        // ---- pass 2: dynamics + stage cost over the own chunk (src/point_mass_gpu.cu:97-107,
        //      src/cost.cu:42-55) on the SCALED state d_p = sqrt|w_p| (p - g_p),
        //      d_v = sqrt|w_v| (v - g_v): d_p' = d_p + k1 d_v + k2 a (+ cg), d_v' = d_v + k3 a, and
        //      the state part of the stage cost is sgn(w_p) d_p'^2 + sgn(w_v) d_v'^2 -- 7 VALU per
        //      normal instead of 11 (gains from the host in double; a zero weight gets the scale
        //      2^-60).  One accumulator per axis and term; the signs are applied when they are
        //      summed.  No per-step masking: the chunk that holds step T-1 takes a snapshot (cost
        //      so far, state) at the wave-uniform step n_last and uses that. -----------------------
        float cpart = 0.0f;
        // (the drift term cg of a velocity goal != 0 is a SEPARATE copy of the pass behind one
        //  wave-uniform branch: tested inside the step loop, hipcc makes an addition and a select
        //  per normal of it)
        auto pass2_scaled = [&](auto cg_tag) {
            constexpr bool CG = decltype(cg_tag)::value;
            float dps[A], dvs[A], ru[A], rp[A], rv[A];
#pragma unroll
            for (int i = 0; i < A; ++i) {
                dps[i] = fmaf(g.fs_sp[i], p[i], -g.fs_gps[i]);
                dvs[i] = fmaf(g.fs_sv[i], v[i], -g.fs_gvs[i]);
                ru[i] = 0.0f; rp[i] = 0.0f; rv[i] = 0.0f;
            }
            auto signed_total = [&]() {
                float t = 0.0f;
#pragma unroll
                for (int i = 0; i < A; ++i) t += ru[i];
#pragma unroll
                for (int i = 0; i < A; ++i) t = fmaf(g.fs_sgp[i], rp[i], t);
#pragma unroll
                for (int i = 0; i < A; ++i) t = fmaf(g.fs_sgv[i], rv[i], t);
                return t;
            };
            if constexpr (!EXACT) asm volatile("" : "+s"(ngs));
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi < ngs) {
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
#pragma unroll
                        for (int i = 0; i < A; ++i) {
                            const float a = u[s * A + i] + es[i];
                            float pn = fmaf(P.k2[i], a, fmaf(P.k1[i], dvs[i], dps[i]));
                            if constexpr (CG) pn += P.cg[i];
                            dvs[i] = fmaf(P.k3[i], a, dvs[i]);
                            dps[i] = pn;
                            ru[i] = fmaf(uc[s * A + i], es[i], ru[i]);
                            rp[i] = fmaf(pn, pn, rp[i]);
                            rv[i] = fmaf(dvs[i], dvs[i], rv[i]);
                        }
                        if (sl + 1 == n_last) {
                            // wave-uniform scalar branch, taken at one step per tile: the snapshot
                            // goes through the thread's LDS slot (kept in registers it becomes
                            // loop-carried values that hipcc copies at EVERY step)
                            snap[0] = signed_total();
#pragma unroll
                            for (int i = 0; i < A; ++i) {
                                snap[(1 + i) * kRolloutThreads] = dps[i];
                                snap[(1 + A + i) * kRolloutThreads] = dvs[i];
                            }
                        }
                    }
                }
            }
            cpart = signed_total();
        };
        if (has_cg) pass2_scaled(std::true_type());
        else pass2_scaled(std::false_type());
        {
            const float cT = snap[0];
            float fc = 0.0f;    // Cost::final_cost (src/cost.cu:57-64) on the state after step T-1
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float d = snap[(1 + i) * kRolloutThreads];
                fc = fmaf(g.fs_sgp[i] * d, d, fc);
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float d = snap[(1 + A + i) * kRolloutThreads];
                fc = fmaf(g.fs_sgv[i] * d, d, fc);
            }
            cpart = (c < c_last) ? cpart : (c == c_last ? cT + fc : 0.0f);
        }
#else
        // ---- pass 2: dynamics + stage cost over the own chunk (src/point_mass_gpu.cu:97-107,
        //      src/cost.cu:42-55).  No per-step masking: the chunk that holds step T-1 takes a
        //      snapshot (cost so far, state) at the wave-uniform step n_last and uses that. ----
        float cpart = 0.0f;
        if constexpr (!EXACT) asm volatile("" : "+s"(ngs));
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi < ngs) {
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
                    if (sl + 1 == n_last) {
                        // wave-uniform scalar branch, taken at one step per tile: the snapshot
                        // goes through the thread's LDS slot.  (Kept in registers it becomes
                        // 2A+1 loop-carried values that hipcc copies at EVERY step: 70 v_mov per
                        // tile at act_dim 2, 6.6 % of the kernel's VALU instructions.)
                        snap[0] = cpart;
#pragma unroll
                        for (int i = 0; i < A; ++i) {
                            snap[(1 + i) * kRolloutThreads] = p[i];
                            snap[(1 + A + i) * kRolloutThreads] = v[i];
                        }
                    }
                }
            }
        }
        {
            const float cT = snap[0];
            float pT[A], vT[A];
#pragma unroll
            for (int i = 0; i < A; ++i) {
                pT[i] = snap[(1 + i) * kRolloutThreads];
                vT[i] = snap[(1 + A + i) * kRolloutThreads];
            }
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
            cpart = (c < c_last) ? cpart : (c == c_last ? cT + fc : 0.0f);
        }
#endif
        if (first) MPPI_STAMP(4);
        const float cost = group_sum<LOGC>(cpart);
        if (valid && c == 0) cost_out[kloc] = cost;

        // ---- block tail: min, exp weights, weighted noise sums ----------------------------
        const float m_t = tile_min(valid ? cost : INFINITY, misc, wave, lane);
        const float wt = valid ? expf(-inv_lambda * (cost - m_t)) : 0.0f;
        {
            const float sw = wave_sum(c == 0 ? wt : 0.0f);
            if (lane == 0) misc[4 + wave] = sw;
        }
        if (first) MPPI_STAMP(5);
        const float wtN =
            ((long long)kglob < k_cover && ((unsigned int)kglob & cover_and) == 0u) ? wt : 0.0f;
        float* wrow = wsum + wave * TAp + (c * nq) * 4;
        if constexpr (!EXACT) asm volatile("" : "+s"(ngs));
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi < ngs) {
#pragma unroll
                for (int j = 0; j < BPG; ++j) {
                    const int q = gi * BPG + j;
                    float xw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) xw[i] = wtN * e[q * 4 + i];
                    nreduce_block<LOGC>(xw, wrow + q * 4, lane);
                }
            }
        }
        if (first) MPPI_STAMP(6);
        __syncthreads();
        if (first) MPPI_STAMP(7);
        // (the block's last tile folds straight into the block partial in global memory)
        const bool last_tile = MPPI_FUSED_DIRECT_FOLD && tb + nblk >= n_tileblk;
        fold_tile(rs, m_t, misc, wsum, nrun, TAp, TA, inv_lambda, first, last_tile ? Nout : nullptr);
        if (first) MPPI_STAMP(8);
        first = false;
        if (!last_tile) __syncthreads();
    }
    MPPI_STAMP(9);

    // ---- publish the block partial ----------------------------------------------------------
#if MPPI_FUSED_DIRECT_FOLD
    if (first)          // (a block without a tile: the grid never has one, kept for safety)
        for (int n = threadIdx.x; n < TA; n += kRolloutThreads) Nout[n] = 0.0f;
#else
    for (int n = threadIdx.x; n < TA; n += kRolloutThreads) Nout[n] = first ? 0.0f : nrun[n];
#endif
    if (threadIdx.x == 0) {
        g.part_m[bid] = rs.M;
        g.part_s[bid] = rs.S;
    }
    MPPI_STAMP(10);
}

// Occupancy target: the kernel is VALU-issue / latency bound, so 4 waves per SIMD (<= 128 VGPRs)
// where the register-resident noise leaves room; forcing it on larger chunks spills
// (measured: -35 %; three waves for 48 resident normals: 23 spilled registers, -25 %), so those
// keep hipcc's own allocation.
template <int A, int NG>
constexpr int fused_min_waves()
{
    constexpr int NE = NG * Dim<A>::BPG * 4;
    return NE <= 16 ? 4 : 2;
}

template <int A, int NG, bool SAMPLE, bool RIDE, bool EXACT>
__device__ __forceinline__ void fused_kernel_body(const RolloutHot& h, const DeferredCombine& d)
{
    if constexpr (RIDE) {
        if ((int)blockIdx.x < d.n_blocks) {
            // combine role: the previous solve's beta / nabla / update / shift (combine_impl.hpp)
            // (the rollout blocks of this launch will wait for these few waves: let them win the
            //  instruction arbitration against the Philox work sharing their SIMDs)
            __builtin_amdgcn_s_setprio(3);
            MPPI_STAMP(0);
            extern __shared__ __align__(16) unsigned char smem_raw[];
            combine_body<kRolloutThreads, kSmallCombineNR>(
                d.c, (int)blockIdx.x,
                carve_combine_smem<kRolloutThreads>(reinterpret_cast<float*>(smem_raw)));
            MPPI_STAMP(10);
            return;
        }
    }
#ifdef MPPI_ONLY_LOGC          // analysis builds: a single body, for reading the ISA
    fused_body<A, NG, SAMPLE, MPPI_ONLY_LOGC, RIDE, EXACT>(h, d);
    return;
#endif
    switch (h.logC) {      // wave-uniform: one specialised body per lanes-per-trajectory
        case 0: fused_body<A, NG, SAMPLE, 0, RIDE, EXACT>(h, d); break;
        case 1: fused_body<A, NG, SAMPLE, 1, RIDE, EXACT>(h, d); break;
        case 2: fused_body<A, NG, SAMPLE, 2, RIDE, EXACT>(h, d); break;
        case 3: fused_body<A, NG, SAMPLE, 3, RIDE, EXACT>(h, d); break;
        case 4: fused_body<A, NG, SAMPLE, 4, RIDE, EXACT>(h, d); break;
        case 5: fused_body<A, NG, SAMPLE, 5, RIDE, EXACT>(h, d); break;
        default: fused_body<A, NG, SAMPLE, 6, RIDE, EXACT>(h, d); break;
    }
}

template <int A, int NG, bool SAMPLE, bool EXACT = false>
__global__ void __launch_bounds__(kRolloutThreads, (fused_min_waves<A, NG>()))
k_rollout_fused(const RolloutHot h)
{
    fused_kernel_body<A, NG, SAMPLE, false, EXACT>(h, DeferredCombine());
}

// The same rollout with the previous solve's combine riding at the front of the grid
// (DeferredCombine); a separate instantiation so that the plain kernel pays nothing for it.
template <int A, int NG, bool SAMPLE, bool EXACT = false>
__global__ void __launch_bounds__(kRolloutThreads, (fused_min_waves<A, NG>()))
k_rollout_ride(const RolloutHot h, const DeferredCombine d)
{
    fused_kernel_body<A, NG, SAMPLE, true, EXACT>(h, d);
}

template <int A, int NG>
hipError_t launch_fused_t(bool sample, int grid, const RolloutArgs& a, const DeferredCombine& d,
                          hipStream_t st, LaunchTiming tm)
{
    size_t lds = rollout_lds_bytes(a.NBTp, a.C * a.nq * 4);
    if (d.n_blocks > 0 && lds < combine_small_lds_bytes()) lds = combine_small_lds_bytes();
    const RolloutHot h = make_hot(a);
    const dim3 g(grid + d.n_blocks), b(kRolloutThreads);
    if constexpr (kExactGroupsPays<A>) {
        if (a.ng == NG) {
            if (d.n_blocks > 0) {
                if (sample) MPPI_LAUNCH((k_rollout_ride<A, NG, true, true>), g, b, lds, st, tm, h, d);
                else MPPI_LAUNCH((k_rollout_ride<A, NG, false, true>), g, b, lds, st, tm, h, d);
            } else {
                if (sample) MPPI_LAUNCH((k_rollout_fused<A, NG, true, true>), g, b, lds, st, tm, h);
                else MPPI_LAUNCH((k_rollout_fused<A, NG, false, true>), g, b, lds, st, tm, h);
            }
            return hipGetLastError();
        }
    }
    if (d.n_blocks > 0) {
        if (sample) MPPI_LAUNCH((k_rollout_ride<A, NG, true>), g, b, lds, st, tm, h, d);
        else MPPI_LAUNCH((k_rollout_ride<A, NG, false>), g, b, lds, st, tm, h, d);
    } else {
        if (sample) MPPI_LAUNCH((k_rollout_fused<A, NG, true>), g, b, lds, st, tm, h);
        else MPPI_LAUNCH((k_rollout_fused<A, NG, false>), g, b, lds, st, tm, h);
    }
    return hipGetLastError();
}

// resident blocks per CU of the instantiation (occupancy API).  ride = false: the plain kernel, used
// only to size the persistent grid; ride = true: the riding variants, whose blocks WAIT for each
// other inside the launch -- the engine lets a combine ride only in a launch whose blocks all fit
// the chip at once by this number.
template <int A, int NG>
int fused_blocks_per_cu_t(bool sample, size_t lds, bool ride)
{
    int n = 0;
    hipError_t rc;
    if (!ride) {
        rc = sample ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_rollout_fused<A, NG, true>,
                                                                   kRolloutThreads, lds)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_rollout_fused<A, NG, false>,
                                                                   kRolloutThreads, lds);
        return rc == hipSuccess ? n : 0;
    }
    if (lds < combine_small_lds_bytes()) lds = combine_small_lds_bytes();
    rc = sample ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_rollout_ride<A, NG, true>,
                                                               kRolloutThreads, lds)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_rollout_ride<A, NG, false>,
                                                               kRolloutThreads, lds);
    if (rc != hipSuccess) return 0;
    if constexpr (kExactGroupsPays<A>) {
        int m = 0;
        rc = sample ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, k_rollout_ride<A, NG, true, true>,
                                                                   kRolloutThreads, lds)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, k_rollout_ride<A, NG, false, true>,
                                                                   kRolloutThreads, lds);
        if (rc != hipSuccess) return 0;
        if (m < n) n = m;
    }
    return n;
}

template <int A>
int fused_blocks_per_cu_a(int NGt, bool sample, size_t lds, bool ride)
{
    if constexpr (A == 3) {
        switch (NGt) {
            case 1: return fused_blocks_per_cu_t<A, 1>(sample, lds, ride);
            case 2: return fused_blocks_per_cu_t<A, 2>(sample, lds, ride);
            case 4: return fused_blocks_per_cu_t<A, 4>(sample, lds, ride);
            case 7: return fused_blocks_per_cu_t<A, 7>(sample, lds, ride);
            default: return 0;
        }
    } else {
        switch (NGt) {
            case 1: return fused_blocks_per_cu_t<A, 1>(sample, lds, ride);
            case 2: return fused_blocks_per_cu_t<A, 2>(sample, lds, ride);
            case 4: return fused_blocks_per_cu_t<A, 4>(sample, lds, ride);
            case 7: return fused_blocks_per_cu_t<A, 7>(sample, lds, ride);
            case 13: return fused_blocks_per_cu_t<A, 13>(sample, lds, ride);
            case 20: return fused_blocks_per_cu_t<A, 20>(sample, lds, ride);
            default: return 0;
        }
    }
}

template <int A>
hipError_t launch_fused_a(int NGt, bool sample, int grid, const RolloutArgs& a,
                          const DeferredCombine& d, hipStream_t st, LaunchTiming tm)
{
    if constexpr (A == 3) {
        switch (NGt) {
            case 1: return launch_fused_t<A, 1>(sample, grid, a, d, st, tm);
            case 2: return launch_fused_t<A, 2>(sample, grid, a, d, st, tm);
            case 4: return launch_fused_t<A, 4>(sample, grid, a, d, st, tm);
            case 7: return launch_fused_t<A, 7>(sample, grid, a, d, st, tm);
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (NGt) {
            case 1: return launch_fused_t<A, 1>(sample, grid, a, d, st, tm);
            case 2: return launch_fused_t<A, 2>(sample, grid, a, d, st, tm);
            case 4: return launch_fused_t<A, 4>(sample, grid, a, d, st, tm);
            case 7: return launch_fused_t<A, 7>(sample, grid, a, d, st, tm);
            case 13: return launch_fused_t<A, 13>(sample, grid, a, d, st, tm);
            case 20: return launch_fused_t<A, 20>(sample, grid, a, d, st, tm);
            default: return hipErrorInvalidValue;
        }
    }
}

}  // namespace mppi
