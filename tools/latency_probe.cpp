// latency_probe.cpp -- where the time of a blocking get_act goes, from the host's side, through
// the C ABI (no Python): per-call wall time of
//   get_act                      (the reference's timed unit, src/main.cu:329-332)
//   get_act + set_x              (the closed-loop pattern, src/main.cu:326-374)
//   solve_async + sync_act       (same work, hipStreamSynchronize instead of polling)
//   solve_async x N, one sync    (pipelined: what bench.py's headline times)
//   get_act with the host busy for `think_us` between two calls (a plant step; the time of the
//   calls alone) -- what the noise prefetch of mppi_set_noise_prefetch is for
// usage: latency_probe [A K T [n [think_us]]]      build: g++ -O2 -std=c++17 -Iinclude tools/latency_probe.cpp
//        -Lmppi_gpu_amd/lib -lmppi_gpu_amd -Wl,-rpath,$PWD/mppi_gpu_amd/lib -Wl,-rpath,/opt/rocm/lib
#include "mppi_gpu_amd.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now_us()
{
    return std::chrono::duration<double, std::micro>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { int rc_ = (x); if (rc_) { printf("error %d: %s\n", rc_, mppi_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int A = argc > 3 ? atoi(argv[1]) : 2, K = argc > 3 ? atoi(argv[2]) : 10000,
              T = argc > 3 ? atoi(argv[3]) : 200, n = argc > 4 ? atoi(argv[4]) : 2000;
    const double think_us = argc > 5 ? atof(argv[5]) : 0.0;
    const int S = 2 * A;
    const float goals[4][8] = {{1, 0}, {1, 0, 0, 0}, {1, .5f, .75f, 0, 0, 0}, {1, .5f, .75f, .25f}};
    const float ws[4][8] = {{1, 5}, {1, 1, 50, 50}, {1, 1, 1, 5, 5, 5}, {1, 1, 1, 1, 5, 5, 5, 5}};
    std::vector<float> x(S, 0.05f), U((size_t)T * A, 0.f), act(A);
    mppi_engine* e = nullptr;
    CK(mppi_create(K, T, 0.1f, S, A, 0, &e));
    CK(mppi_set_data(e, x.data(), U.data(), goals[A - 1], ws[A - 1]));
    for (int i = 0; i < 300; ++i) CK(mppi_get_act(e, act.data()));      // clocks up

    double t0 = now_us();
    for (int i = 0; i < n; ++i) CK(mppi_get_act(e, act.data()));
    const double t_get = (now_us() - t0) / n;

    t0 = now_us();
    for (int i = 0; i < n; ++i) { CK(mppi_get_act(e, act.data())); CK(mppi_set_x(e, x.data())); }
    const double t_loop = (now_us() - t0) / n;

    double t_think = 0.0;        // the calls alone, with the host away for think_us in between
    if (think_us > 0.0) {
        for (int i = 0; i < n + 20; ++i) {
            const double a = now_us();
            CK(mppi_get_act(e, act.data()));
            const double b = now_us();
            if (i >= 20) t_think += b - a;
            CK(mppi_set_x(e, x.data()));
            while (now_us() - b < think_us) {}
        }
        t_think /= n;
    }

    t0 = now_us();
    for (int i = 0; i < n; ++i) { CK(mppi_solve_async(e, nullptr)); CK(mppi_sync_act(e, act.data())); }
    const double t_sync = (now_us() - t0) / n;

    CK(mppi_sync_act(e, act.data()));
    t0 = now_us();
    for (int i = 0; i < n; ++i) CK(mppi_solve_async(e, nullptr));
    const double t_enq = (now_us() - t0) / n;
    CK(mppi_sync_act(e, act.data()));
    const double t_pipe = (now_us() - t0) / n;

    printf("{\"A\": %d, \"K\": %d, \"T\": %d, \"n\": %d, \"get_act_us\": %.2f, \"get_act_set_x_us\": %.2f, "
           "\"solve_async_sync_us\": %.2f, \"pipelined_us\": %.2f, \"enqueue_only_us\": %.2f, "
           "\"think_us\": %.1f, \"get_act_with_think_us\": %.2f}\n",
           A, K, T, n, t_get, t_loop, t_sync, t_pipe, t_enq, think_us, t_think);
    mppi_destroy(e);
    return 0;
}
