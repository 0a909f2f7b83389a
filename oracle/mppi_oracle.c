/*
 * mppi_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A serial, single-threaded CPU restatement of one MPPI solve of the reference
 * (NicolayP/mppi_gpu), used as the checker for the HIP path.  Nothing under
 * mppi_gpu_amd/ may include, link or call this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() do.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_step_cost / orc_final_cost : PINNED bit-for-bit against the reference's own
 *     src/cost.cu compiled unmodified here (oracle/_ref, tests/golden/cost_ref_*.npz).
 *   - orc_update                     : PINNED against the reference's own known-answer
 *     generator + CPU checker in src/test.cu:77-105.
 *   - orc_exp                        : PINNED against the vector of src/test.cu:11-59.
 *   - orc_rollout (the LTI step), orc_beta, orc_nabla, orc_weights, orc_shift:
 *     PARITY UNPINNED -- the reference holds no test vector for them and
 *     src/point_mass_gpu.cu / src/point_mass.cu cannot be compiled in this image
 *     (they need the cuRAND headers / the CUDA runtime).  They restate the cited lines.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: every float operation below
 * rounds separately, like the reference's host branch compiled by g++ for x86-64).
 *
 * Layouts are the reference's: X[k][t][s] with (T+1) rows per sample, E[k][t][a],
 * U[t][a]; state = positions then velocities (S == 2*A).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- Cost (reference src/cost.cu:42-64) -------------------------------------------- */

/* reference Cost::step_cost, src/cost.cu:42-55 */
float orc_step_cost(const float* x, const float* u, const float* e, const float* w,
                    const float* goal, float lambda, const float* inv_s, int S, int A)
{
    float res = 0.0f;
    for (int i = 0; i < A; i++) {
        res += u[i] * inv_s[i] * e[i];
    }
    res *= lambda;
    for (int i = 0; i < S; i++) {
        res += (x[i] - goal[i]) * w[i] * (x[i] - goal[i]);
    }
    return res;
}

/* reference Cost::final_cost, src/cost.cu:57-64 */
float orc_final_cost(const float* x, const float* w, const float* goal, int S)
{
    float res = 0.0f;
    for (int i = 0; i < S; i++) {
        res += (x[i] - goal[i]) * w[i] * (x[i] - goal[i]);
    }
    return res;
}

/* ---- Gains (reference src/point_mass.cu:46-51) -------------------------------------- */

/* x_gain = {1, dt, 0, 1}; u_gain = {dt*dt/2.0, dt}: the product dt*dt is a float product,
 * the division by 2.0 is done in double and the result is stored as float. */
void orc_gains(float dt, float* x_gain /*4*/, float* u_gain /*2*/)
{
    x_gain[0] = 1.0f;
    x_gain[1] = dt;
    x_gain[2] = 0.0f;
    x_gain[3] = 1.0f;
    float dd = dt * dt;
    u_gain[0] = (float)((double)dd / 2.0);
    u_gain[1] = dt;
}

/* ---- Rollout (reference src/point_mass_gpu.cu:82-121) ------------------------------- */

/* One sample: PointMassModelGpu::run (:111-121) = T x step (:82-109) + final_cost.
 * x_traj: (T+1)*S floats, row 0 is x0 on entry (written here). Returns the path cost. */
static float orc_run_one(float* x_traj, const float* x0, const float* U, const float* e,
                         int T, int S, int A, const float* xg, const float* ug,
                         const float* w, const float* goal, float lambda, const float* inv_s)
{
    const int h = S / 2;
    for (int s = 0; s < S; s++) x_traj[s] = x0[s];
    float c = 0.0f;
    for (int t = 0; t < T; t++) {
        const float* x = &x_traj[(size_t)t * S];
        float* xn = &x_traj[(size_t)(t + 1) * S];
        for (int i = 0; i < A; i++) {
            /* point_mass_gpu.cu:98-104, evaluated left to right */
            xn[i] = xg[0] * x[i] + xg[1] * x[i + h] + ug[0] * (U[t * A + i] + e[t * A + i]);
            xn[i + h] = xg[2] * x[i] + xg[3] * x[i + h] + ug[1] * (U[t * A + i] + e[t * A + i]);
        }
        /* point_mass_gpu.cu:107 */
        c += orc_step_cost(xn, &U[t * A], &e[t * A], w, goal, lambda, inv_s, S, A);
    }
    /* point_mass_gpu.cu:116: the terminal state is counted in the last stage AND here */
    c += orc_final_cost(&x_traj[(size_t)T * S], w, goal, S);
    return c;
}

/* All K samples. X may be NULL (then a scratch trajectory is used and nothing is kept). */
void orc_rollout(int K, int T, int S, int A, float dt, const float* x0, const float* U,
                 const float* E, const float* goal, const float* w, float lambda,
                 const float* inv_s, float* cost, float* X)
{
    float xg[4], ug[2];
    orc_gains(dt, xg, ug);
    float* scratch = NULL;
    if (!X) scratch = (float*)malloc(sizeof(float) * (size_t)(T + 1) * S);
    for (int k = 0; k < K; k++) {
        float* xt = X ? &X[(size_t)k * (T + 1) * S] : scratch;
        cost[k] = orc_run_one(xt, x0, U, &E[(size_t)k * T * A], T, S, A, xg, ug, w, goal,
                              lambda, inv_s);
    }
    free(scratch);
}

/* ---- beta / exp / nabla / weights (reference src/point_mass.cu:273-382,510-575,628-666,743-754) */

/* min_k cost[k]  (min is exact under any association; point_mass.cu:533-575) */
float orc_beta(const float* cost, int K)
{
    float b = INFINITY;
    for (int k = 0; k < K; k++) b = cost[k] < b ? cost[k] : b;
    return b;
}

/* exp_red, point_mass.cu:510-531: out[k] = expf(-(1/lambda) * (cost[k]-beta)), all float */
void orc_exp(const float* cost, float lambda, float beta, float* out, int K)
{
    for (int k = 0; k < K; k++) out[k] = expf(-(1 / lambda) * (cost[k] - beta));
}

/* One pass of sum_red (point_mass.cu:628-666): 256-thread blocks, 512 inputs per block,
 * first add pairs (i, i+256), then strides 128..1; one partial per block. */
static int orc_sum_red_pass(const float* v, float* vr, int n, int grid)
{
    float part[256];
    for (int b = 0; b < grid; b++) {
        for (int tx = 0; tx < 256; tx++) {
            long i = (long)b * 512 + tx;
            if (i + 256 < n) part[tx] = v[i] + v[i + 256];
            else if (i < n) part[tx] = v[i];
            else part[tx] = 0.0f;
        }
        for (int s = 128; s > 0; s >>= 1)
            for (int tx = 0; tx < s; tx++) part[tx] += part[tx + s];
        vr[b] = part[0];
    }
    return grid;
}

/* nabla with the reference's association (host loop of point_mass.cu:328-377).  The
 * reference's in-place multi-block mid pass is racy; it is emulated out of place. */
float orc_nabla_tree(const float* ex, int K)
{
    int n = K;
    int grid = n / 256 / 2 + 1;
    float* a = (float*)malloc(sizeof(float) * (size_t)(grid > 0 ? grid : 1));
    float* b = (float*)malloc(sizeof(float) * (size_t)(grid > 0 ? grid : 1));
    float r;
    if (grid == 1) {
        orc_sum_red_pass(ex, a, n, 1);
        r = a[0];
    } else {
        orc_sum_red_pass(ex, a, n, grid);
        n = grid;
        grid = n / 256 / 2 + 1;
        while (grid - 1 > 1) {
            orc_sum_red_pass(a, b, n, grid);
            float* tmp = a; a = b; b = tmp;
            n = grid;
            grid = n / 256 / 2 + 1;
        }
        orc_sum_red_pass(a, b, n, 1);
        r = b[0];
    }
    free(a);
    free(b);
    return r;
}

/* plain double-precision sum, rounded once: the association-free value of nabla */
float orc_nabla(const float* ex, int K)
{
    double s = 0.0;
    for (int k = 0; k < K; k++) s += (double)ex[k];
    return (float)s;
}

/* weights_kernel, point_mass.cu:743-754:
 * w[k] = 1.0/nabla * expf(-(1.0/lambda)*(cost[k]-beta)); the 1.0 literals make the
 * products double, expf takes the double argument rounded to float. */
void orc_weights(const float* cost, float lambda, float beta, float nabla, float* wout, int K)
{
    for (int k = 0; k < K; k++) {
        double arg = -(1.0 / (double)lambda) * (double)(cost[k] - beta);
        wout[k] = (float)(1.0 / (double)nabla * (double)expf((float)arg));
    }
}

/* ---- update / shift (reference point_mass.cu:384-480, 756-761, 195-199, 805-824) ---- */

/* Semantics of update_act: U[t][a] += sum_k w[k]*E[k][t][a].  Same loop order as the
 * reference's own CPU checker update_act_cpu (src/test.cu:97-105): k outermost, float. */
void orc_update(float* U, const float* wts, const float* E, int K, int T, int A)
{
    for (int k = 0; k < K; k++)
        for (int j = 0; j < T; j++)
            for (int i = 0; i < A; i++)
                U[j * A + i] += wts[k] * E[(size_t)k * T * A + j * A + i];
}

/* Same sum accumulated in double and rounded once: the association-free value used for
 * tolerance checks at large K, where the float k-loop above loses digits itself. */
void orc_update_f64(float* U, const float* wts, const float* E, int K, int T, int A)
{
    double* acc = (double*)calloc((size_t)T * A, sizeof(double));
    for (int k = 0; k < K; k++)
        for (int n = 0; n < T * A; n++)
            acc[n] += (double)wts[k] * (double)E[(size_t)k * T * A + n];
    for (int n = 0; n < T * A; n++) U[n] = (float)((double)U[n] + acc[n]);
    free(acc);
}

/* The reference's coverage defect for act_dim != 2: how many leading
 * samples update_act actually sums for A = 3 (exact for A = 2). Provided for ref_compat. */
int orc_ref_update_coverage_a3(int K)
{
    long cov = 512L * (K / 768 + 1);
    return (int)(cov < K ? cov : K);
}

/* next_act = U[0,:] (point_mass.cu:195); shift_act (:805-824): U[t]=U[t+1], last repeated */
void orc_shift(float* U, float* next_act, int T, int A)
{
    for (int a = 0; a < A; a++) next_act[a] = U[a];
    for (int t = 0; t < T - 1; t++)
        for (int a = 0; a < A; a++) U[t * A + a] = U[(t + 1) * A + a];
}

/* ---- One full solve on injected noise (PointMassModel::get_act, point_mass.cu:129-203) */

/* U is updated and shifted in place.  Optional outputs may be NULL.
 * f64_update != 0 selects orc_update_f64/orc_nabla (association-free) instead of the
 * float k-loop / reference tree. */
void orc_solve(int K, int T, int S, int A, float dt, const float* x0, float* U, const float* E,
               const float* goal, const float* w, float lambda, const float* inv_s,
               int f64_update, float* next_act, float* cost_out, float* beta_out,
               float* nabla_out, float* weights_out, float* X_out)
{
    float* cost = cost_out ? cost_out : (float*)malloc(sizeof(float) * (size_t)K);
    float* ex = (float*)malloc(sizeof(float) * (size_t)K);
    float* wts = weights_out ? weights_out : (float*)malloc(sizeof(float) * (size_t)K);

    orc_rollout(K, T, S, A, dt, x0, U, E, goal, w, lambda, inv_s, cost, X_out);
    float beta = orc_beta(cost, K);
    orc_exp(cost, lambda, beta, ex, K);
    float nabla = f64_update ? orc_nabla(ex, K) : orc_nabla_tree(ex, K);
    orc_weights(cost, lambda, beta, nabla, wts, K);
    if (f64_update) orc_update_f64(U, wts, E, K, T, A);
    else orc_update(U, wts, E, K, T, A);
    orc_shift(U, next_act, T, A);

    if (beta_out) *beta_out = beta;
    if (nabla_out) *nabla_out = nabla;
    if (!cost_out) free(cost);
    free(ex);
    if (!weights_out) free(wts);
}

/* ---- Known-answer generators of the reference's own test (src/test.cu) -------------- */

/* init_update_act_data, test.cu:77-95 */
void orc_kat_update_inputs(float* u, float* w, float* e, int n, int t, int a)
{
    for (int k = 0; k < n; k++)
        for (int j = 0; j < t; j++)
            for (int i = 0; i < a; i++)
                e[k * t * a + j * a + i] = 0.25 * (k * t * a + j * a + i);
    for (int k = 0; k < n; k++) w[k] = 0.5 * k;
    for (int j = 0; j < t; j++)
        for (int i = 0; i < a; i++) u[j * a + i] = 0.75 * (j * a + i);
}

/* test_exp's expectation, test.cu:43-46: exp(-lambda*(cost-beta)) in double */
double orc_kat_exp_expected(float cost, float lambda, float beta)
{
    return exp(-lambda * (cost - beta));
}
