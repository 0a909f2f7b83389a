/*
 * mppi_gpu_amd.h -- C ABI of the MI355X MPPI rollout-and-update engine.
 *
 * This is the drop-in boundary for the hot path of NicolayP/mppi_gpu: one MPPI solve
 * (PointMassModel::get_act, reference src/point_mass.cu:129-203) plus the data-in/data-out
 * calls around it.  Every entry point below replaces one public member of the reference's
 * `class PointMassModel` (reference include/point_mass.hpp:23-44); the C++ class of the same
 * name shipped in include/point_mass.hpp is a thin veneer over these functions.
 *
 * Conventions
 *   - plain pointers and sizes only; every `float*` is a HOST pointer owned by the caller
 *     unless its name starts with `d_` (device pointer, caller-owned, on the engine's GPU);
 *   - layouts are the reference's: X[k][t][s] with T+1 rows per sample, E[k][t][a], U[t][a];
 *     state = positions then velocities, S == 2*A, A in 1..4;
 *   - return value: 0 on success, a negative MPPI_E* code otherwise; mppi_last_error() holds
 *     the message.  (The reference prints "API error failed file:line" and exit(1)s,
 *     include/mppi_utils.hpp:19-25; the C++ veneer keeps that behaviour, the C ABI leaves the
 *     decision to the caller.)
 *   - an engine is bound to one GPU and one host thread at a time, like the reference.
 *   - there is NO CPU fallback: without a usable HIP device mppi_create fails with
 *     MPPI_ENODEV.
 *   - accuracy against the reference's serial fp32 arithmetic (src/point_mass_gpu.cu:82-121,
 *     src/cost.cu:42-64) on the same noise: the strict kernel (mppi_set_tuning) reproduces the
 *     path costs bit for bit; the fused kernels (default) re-associate the T-step recurrence and
 *     agree to max(3e-6, 0.35 * T * 2^-24) relative in every path cost -- 4.2e-6 at T = 200,
 *     2.1e-5 at T = 1000 (measured: <= 0.23 * T * 2^-24, tools/sweep_cost_error.py) -- and to
 *     1e-5 * max(|U|, sigma) in the controls wherever more than a handful of samples carry
 *     weight.  A cost weight of exactly 0 is carried as the scale 2^-60: a problem whose weights
 *     are ALL zero reports path costs of ~1e-36 instead of 0 (the controls are unaffected).
 *     Horizons are bounded by the LDS one block may use on gfx950 (160 KiB: about
 *     T * act_dim <= 5000; beyond 64 KiB fewer blocks share a CU) and by the 64 lanes x groups per
 *     lane a trajectory can be spread over (MPPI_EINVAL beyond).
 */
#ifndef MPPI_GPU_AMD_H_
#define MPPI_GPU_AMD_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mppi_engine mppi_engine;

enum {
    MPPI_OK = 0,
    MPPI_EINVAL = -1,   /* bad argument (dims, null pointer, unsupported A) */
    MPPI_ENODEV = -2,   /* no usable HIP device / runtime failure at start-up */
    MPPI_EHIP = -3,     /* a HIP runtime call failed; see mppi_last_error() */
    MPPI_ESTATE = -4    /* call out of protocol (e.g. get_act before set_data) */
};

/* ---- reference surface ------------------------------------------------------------- */

/* PointMassModel::PointMassModel(nb_sim, steps, dt, state_dim, act_dim, verbose)
 * reference include/point_mass.hpp:25-30, src/point_mass.cu:19-106.  Uses the current HIP
 * device.  lambda = 1, sigma = 0.025, inv_s = 1 as hard-coded in the reference
 * (src/point_mass.cu:53-54, src/point_mass_gpu.cu:58-61,86). */
int mppi_create(int nb_sim, int steps, float dt, int state_dim, int act_dim, int verbose,
                mppi_engine** out);

/* PointMassModel::~PointMassModel, src/point_mass.cu:108-127 */
void mppi_destroy(mppi_engine* e);

/* PointMassModel::memcpy_set_data(x, u, goal, w), src/point_mass.cu:205-228: uploads x0[S],
 * U[T*A], goal[S], w[S] and (re)starts the noise stream (the reference re-runs curand_init
 * there, :780). */
int mppi_set_data(mppi_engine* e, const float* x0, const float* u, const float* goal,
                  const float* w);

/* PointMassModel::set_x(x), src/point_mass.cu:482-486.  Host-only here: the state travels by
 * value with the arguments of the next rollout launch, so the call touches no GPU and solves that
 * are already enqueued keep the state they were enqueued with. */
int mppi_set_x(mppi_engine* e, const float* x0);

/* PointMassModel::get_x(x): declared at include/point_mass.hpp:34, never defined in the
 * reference; defined here as the current x0[S] (what the next solve will start from). */
int mppi_get_x(mppi_engine* e, float* x0);

/* PointMassModel::get_act(next_act), src/point_mass.cu:129-203: one full solve
 * (sample + rollout + cost, beta, nabla, weighted update), returns U[0,:] of the UPDATED
 * sequence in next_act[A], then shifts U left by one step.  Blocking, like the reference: it
 * returns when the action has arrived in host memory (the combine kernel writes it into pinned
 * words the call polls); later calls on the engine are ordered behind the solve as usual. */
int mppi_get_act(mppi_engine* e, float* next_act);

/* PointMassModel::get_u(u), src/point_mass.cu:488-491: current U[T*A] */
int mppi_get_u(mppi_engine* e, float* u);

/* PointMassModel::memcpy_get_data(x_all, e), src/point_mass.cu:230-234:
 * X[K*(T+1)*S] and E[K*T*A] of the LAST solve.  X is recomputed on demand from E (the timed
 * path does not store X). */
int mppi_get_data(mppi_engine* e, float* x_all, float* noise);

/* PointMassModel::get_inf(x, u, e, cost, beta, nabla, weight), src/point_mass.cu:236-262.
 * Any pointer may be NULL to skip that output. u is the current (updated, shifted) U. */
int mppi_get_inf(mppi_engine* e, float* x_all, float* u, float* noise, float* cost,
                 float* beta, float* nabla, float* weight);

/* ---- extensions (not in the reference) ---------------------------------------------- */

/* lambda, per-axis noise sigma[A] and control-cost inv_s[A]; NULL keeps the current value.
 * The reference parses these from YAML but never passes them on. */
int mppi_set_params(mppi_engine* e, float lambda, const float* sigma, const float* inv_s);

/* seed of the Philox stream (default 0); takes effect at the next mppi_set_data or call. */
int mppi_set_seed(mppi_engine* e, unsigned long long seed);

/* Injected-noise mode: the next solves use this E[K*T*A] instead of sampling (parity tests;
 * the reference exposes E as an I/O buffer, src/point_mass.cu:232,251). NULL = sample again. */
int mppi_set_noise(mppi_engine* e, const float* noise);

/* Whether the rollout materialises the sampled noise E in device memory (default 1: E is an
 * observable of the reference, src/point_mass.cu:232,251, and its store is 94 % of the solve's HBM
 * bytes).  With 0 the rollout writes the path costs and its partial sums only; E is a pure function
 * of (seed, solve index, global sample index, step), and mppi_get_inf / mppi_get_data regenerate it
 * -- bit for bit what would have been stored -- when asked.  Ignored by the strict kernel (which
 * re-reads its noise) and in injected-noise mode. */
int mppi_set_noise_store(mppi_engine* e, int on);

/* Noise prefetch for blocking calls (the reference's closed loop, src/main.cu:326-374: get_act ->
 * plant step -> set_x).  The stand-alone combine launch of a blocking call (mppi_get_act /
 * mppi_wait_act) carries extra low-priority blocks behind its combine blocks; while the host holds
 * the action of solve j they draw the noise of solve j+1 into a second buffer, in the rollout's own
 * layout, and the next rollout -- stream-ordered behind them, no event, no second queue -- loads it
 * instead of drawing it and computes the same bits (E is a pure function of seed, solve index,
 * sample, step).  mode 1 (default): launches of one tile per block (a latency chain that leaves the
 * chip idle: K = 1e4, the shipped configs) always prefetch -- the draw is over about when the
 * combine is --, longer VALU-bound launches when the host's measured think time between the return
 * of one blocking call and the next solve hides the whole draw (a 100 Hz loop at K = 1e5: yes; calls
 * back to back: no); 0 = never, 2 = whenever possible.  Costs a second noise buffer while in use.
 * Measured (tools/latency_probe, C ABI): K = 1e4 2-D, calls back to back 20.9-22.0 -> 20.5 us, with
 * a 5..20 us plant step 21.7-23.5 -> 19.6-19.8; the shipped 3-D config 21.0 -> 18.1; K = 1e5 3-D with
 * a 100 us plant step 84 -> 60. */
int mppi_set_noise_prefetch(mppi_engine* e, int mode);
/* out = { prefetch launches, rollouts that loaded a prefetched buffer } since mppi_create */
int mppi_get_prefetch_counts(mppi_engine* e, long long out[2]);

/* Reproduce the reference's update_act sample-coverage defects: act_dim 3 sums
 * only the first 512*(K/768+1) samples (src/point_mass.cu:387,402,839-842); act_dim 1 sums only
 * the samples k with k even and (k/512) even, because the block trees stop one fold early
 * (src/point_mass.cu:893,709; supported for nb_sim < 262144).  act_dim 2 is exact in the reference.
 * Default off = mathematically correct update. */
int mppi_set_ref_compat(mppi_engine* e, int on);

/* Opt-in action limit: the updated controls (every step of the sequence, hence the returned
 * action) are clamped to [-max_a[a], +max_a[a]] per axis inside the combine kernel; NULL switches
 * it off (default).  The reference parses `max-a` from its YAML (src/main.cu:524,566-568) and
 * never hands it to the controller: default off = the reference's effective
 * behaviour. */
int mppi_set_action_limit(mppi_engine* e, const float* max_a);

/* Kernel shape. chunks = lanes cooperating on one trajectory (power of two, 1..64; 0 = auto: for a
 * launch the chip holds at once as many lanes, up to 32, as keep a lane at >= 7 Philox blocks and
 * the launch at <= ~2.5 blocks per CU -- such a launch is a latency chain -- otherwise the width
 * that keeps the most waves per SIMD);
 * strict != 0 selects the sequential, association-faithful rollout kernel (one lane per
 * trajectory, cost bit-identical to the serial reference arithmetic), used as the parity
 * anchor.  max_blocks caps the persistent grid (0 = auto). */
int mppi_set_tuning(mppi_engine* e, int chunks, int strict, int max_blocks);

/* Trajectory packing of the fused rollout.  By default (0) the engine lays whole trajectories
 * end to end over the lanes of a wavefront (a trajectory need not fill a power-of-two number of
 * lanes: T = 200, act_dim 3 uses 97.7 % of the lane-slots instead of 78 %) whenever that wastes
 * fewer slots than `chunks` lanes per trajectory would AND the row-aligned launch would not fit
 * the chip at once (a launch that does is a latency problem, where the row-aligned kernel's
 * shorter tail wins); needs cost weights >= 0 and chunks == 0 (a horizon that is not a multiple of
 * the group length -- 4, 2, 4, 1 steps for act_dim 1..4 -- has its last group masked).  -1 = never (the
 * row-aligned kernel mppi_set_tuning describes), n > 0 = packed with n groups per lane
 * (MPPI_EINVAL if that size is not built or the problem does not qualify). */
int mppi_set_packing(mppi_engine* e, int groups_per_lane);
/* out = { packed (0/1), groups per lane, trajectories per wavefront, tile groups of 4 wavefronts } */
int mppi_get_layout(mppi_engine* e, int out[4]);

/* How consecutive solves are enqueued:
 *   0  deferred combine (default).  mppi_solve_async launches the rollout only; the combine
 *      (beta, nabla, update, shift, action) is launched by whatever comes next: if that is another
 *      mppi_solve_async on the same stream and the launch is short (at most two tiles per block;
 *      the packed kernel, whose first tile is a code path of its own: a launch of any length),
 *      it RIDES in that launch -- the first blocks of the grid play the combine role while the
 *      rollout blocks draw their Philox / Box-Muller noise (half of the kernel, and independent of
 *      the controls), and each rollout block polls tagged words for the finished controls before
 *      it stages them; otherwise, and for anything that needs the results (mppi_sync_act,
 *      mppi_get_act, mppi_get_u, mppi_get_inf, mppi_set_data, ...), it is launched on its own
 *      first -- same device function, same bits.  mppi_get_act alone is therefore two launches;
 *      back-to-back solves at K = 1e4 become ONE launch each.
 *   1  eager: every solve launches its rollout and a 1024-thread combine at once (another
 *      summation order: results agree with mode 0 to rounding, not bit for bit). */
int mppi_set_pipeline(mppi_engine* e, int mode);
/* The mode in use, and whether the engine chose it itself: after a device watchdog trip (a block
 * waited in vain for the combine riding in its own launch, or for a peer's exchange words -- a GPU
 * shared with other work, a rank that never arrived) the engine reports MPPI_ESTATE until
 * mppi_set_data and from then on runs in mode 1, where no block waits for a block of its own
 * launch; *degraded = 1 says so.  mppi_set_pipeline(0) asks for riding combines again. */
int mppi_get_pipeline(mppi_engine* e, int* mode, int* degraded);

/* ---- asynchronous and sharded use (bench, multi-GPU, closed loop) --------------------- */

/* Enqueue one full solve on `stream` (a hipStream_t; NULL = the engine's own non-blocking stream --
 * note that the legacy default stream also has the handle NULL, so pass a created stream) and
 * return without waiting.  The action lands in a pinned host word readable after
 * mppi_sync_act.  Solves on one engine are ordered on the stream. */
int mppi_solve_async(mppi_engine* e, void* stream);

/* Launch whatever this engine still holds back (pipeline mode 0 defers the combine of the last
 * enqueued solve until it knows what follows) without waiting for it.  Needed only by a host
 * thread that drives SEVERAL engines exchanging with each other: flush all of them before waiting
 * on the first, since a deferred exchange waits for the peers' words. */
int mppi_flush_async(mppi_engine* e);

/* Wait for everything enqueued by this engine and copy the last action into next_act[A]. */
int mppi_sync_act(mppi_engine* e, float* next_act);

/* Wait for the ACTION of the newest enqueued solve only, the way mppi_get_act does: launch what is
 * still held back, then poll the pinned {action bits, tag} words the combine kernel writes (no
 * hipStreamSynchronize and its wake-up unless 300 us pass).  mppi_get_act = mppi_solve_async +
 * mppi_wait_act; a host that drives several engines (include/mppi_gpu_amd_sharded.h) enqueues on
 * all of them first and waits afterwards.  Later calls are stream-ordered behind the solve. */
int mppi_wait_act(mppi_engine* e, float* next_act);

/* Sharded solve (one engine per GPU, samples k_offset .. k_offset+nb_sim-1 of a global
 * batch).  Step 1 enqueues sampling, rollout and the rank-local reduction and writes
 * mppi_partial_len() floats [beta_g, S_g, N_g[T*A]] to the DEVICE buffer d_partial.
 * The caller all-gathers the G partials (RCCL); step 2 combines them in rank order,
 * updates and shifts U identically on every rank. */
int mppi_create_shard(int nb_sim_local, long long k_offset, int steps, float dt, int state_dim,
                      int act_dim, int verbose, mppi_engine** out);
int mppi_partial_len(const mppi_engine* e);
int mppi_solve_local_async(mppi_engine* e, float* d_partial, void* stream);
int mppi_solve_finish_async(mppi_engine* e, const float* d_gathered, int n_parts, void* stream);

/* Direct peer exchange: the sharded solve WITHOUT a collective library on the data path.
 * Every rank owns an inbox in uncached device memory; mppi_xchg_open allocates it and returns its
 * hipIpc handle (mppi_xchg_handle_bytes() bytes) and/or its raw device pointer. The caller
 * distributes the handles out of band (any transport: torch.distributed, MPI, a file) and hands
 * mppi_xchg_connect the `world` handles in rank order; ranks living in the SAME process pass
 * their raw pointers in same_process[rank] instead (hipIpc cannot open a handle in the process
 * that made it); either argument may be NULL.  After that one call per solve,
 * mppi_solve_exchange_async, enqueues the rollout and ONE combine launch that stores this rank's
 * [beta_g, S_g, N_g[T*A]] into all inboxes as 8-byte {value, sequence tag} words over xGMI,
 * polls its own inbox for the other ranks' words and applies the update -- bit-identical to
 * mppi_solve_local_async + all-gather + mppi_solve_finish_async.  In pipeline mode 0 that launch
 * is deferred like the single-GPU combine: with solves enqueued back to back it rides in the next
 * solve's rollout launch, and the peers' words arrive while this rank draws the next noise.
 * All ranks must make the same sequence of exchange calls; a rank that waits longer than the
 * time-out (default 5 s) gives up WITHOUT applying or publishing anything for the columns it
 * missed, and the fault is sticky: every later solve or read-out call on the engine returns
 * MPPI_ESTATE until mppi_set_data starts over (fresh controls, solve counter 0).  The same holds
 * for the single-GPU watchdog (a block that waits 2 s for the combine riding in its own launch).
 * mppi_xchg_close completes a solve whose exchange is still held back before it frees the inbox
 * and reports a time-out of that exchange.  world <= 64. */
int mppi_xchg_handle_bytes(void);
int mppi_xchg_open(mppi_engine* e, int rank, int world, void* handle_out, void** inbox_out);
int mppi_xchg_connect(mppi_engine* e, const void* handles, void* const* same_process);
int mppi_xchg_set_timeout(mppi_engine* e, double seconds);
int mppi_solve_exchange_async(mppi_engine* e, void* stream);
int mppi_xchg_close(mppi_engine* e);

/* ---- measurement ------------------------------------------------------------------- */

/* every > 0: each `every`-th solve and the solve after it record HIP events around their kernels
 * on the launch stream (sparse, so that the timed region stays honest; in pairs, because a stamp
 * behind an unstamped dispatch also covers that dispatch's tail); 0 switches recording off. */
int mppi_set_profiling(mppi_engine* e, int every);
/* Average duration in ms of the sampled launches of kind `which` (0 = rollout launches, including
 * those that carry a combine: the second of each stamped pair; 1 = stand-alone combine launches)
 * since profiling was switched on; *n_out = number of launches averaged. Synchronises. */
int mppi_kernel_ms(mppi_engine* e, int which, double* avg_ms, int* n_out);
/* Launch geometry actually in use: chunks, blocks per chunk (nq), grid, block, strict. */
int mppi_get_geometry(mppi_engine* e, int out[5]);
/* Launches since mppi_create: out[0] rollout launches, out[1] those of them that carried the
 * previous solve's combine (pipeline mode 0; the row-aligned kernel: only launches that are short
 * AND whose blocks -- rollout and combine role -- all fit the chip at once; the packed kernel: any
 * launch), out[2] combine launches on their own (the 256-thread shape), out[3] blocks of the
 * riding kernel the chip holds at once (occupancy API x CUs). */
int mppi_get_launch_counts(mppi_engine* e, long long out[4]);

/* ---- serial CPU controller ------------------------------------------------------------ */

/* The reference's `class ControllerBase` (include/controller_base.hpp:7-42,
 * src/controller_base.cpp:4-98: ctor (k, tau, dt, sDim, aDim), next(x), setActions) made real:
 * one thread, one sample after the other, same noise stream and same pipeline as the GPU
 * engine.  BASELINE config 1 (point_mass1d K=100 T=50, no GPU).  A controller of its own, never
 * a fallback of the GPU entry points above.  C++ users include controller_base.hpp instead. */
typedef struct mppi_cpu_controller mppi_cpu_controller;
mppi_cpu_controller* mppi_cpu_create(int k, int tau, float dt, int s_dim, int a_dim);
void mppi_cpu_destroy(mppi_cpu_controller* c);
int mppi_cpu_set_data(mppi_cpu_controller* c, const float* u, const float* goal, const float* w);
int mppi_cpu_set_params(mppi_cpu_controller* c, float lambda, const float* sigma,
                        const float* inv_s);
int mppi_cpu_set_seed(mppi_cpu_controller* c, unsigned long long seed);
int mppi_cpu_set_noise(mppi_cpu_controller* c, const float* noise);
/* worker threads for the sample loops (default 1); results do not depend on the count */
int mppi_cpu_set_threads(mppi_cpu_controller* c, int n);
int mppi_cpu_next(mppi_cpu_controller* c, const float* x, float* act);
int mppi_cpu_get(mppi_cpu_controller* c, float* u, float* noise, float* cost, float* beta,
                 float* nabla, float* weight);

int mppi_device_count(void);
const char* mppi_last_error(void);
const char* mppi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_GPU_AMD_H_ */
