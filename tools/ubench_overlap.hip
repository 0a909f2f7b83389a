// Do consecutive launches overlap?  Each launch spins ~10 us and stamps the wall clock at the start
// of its first block and at the end of its last; printed: start(j+1) - end(j) in us for (a) one
// stream, ordinary launches, (b) one stream, hipExtAnyOrderLaunch, (c) two streams in turn,
// (d) one stream, hipLaunchCooperativeKernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <chrono>
__global__ void k_spin(unsigned long long* out, int slot, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {      // per-block stamps (no shared address: 625 atomics on one word queue for 7 us)
        out[((size_t)slot * gridDim.x + blockIdx.x) * 2] = t0;
        out[((size_t)slot * gridDim.x + blockIdx.x) * 2 + 1] = wall_clock64();
    }
}
int main(int argc, char** argv)
{
    const unsigned long long ticks = argc > 1 ? atoi(argv[1]) : 1000;
    const int n = 400, grid = 625;
    unsigned long long* d;
    hipMalloc(&d, (size_t)n * grid * 16);
    std::vector<unsigned long long> hb((size_t)n * grid * 2);
    hipStream_t s[2];
    hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    std::vector<unsigned long long> h(n * 2);
    double host_us = 0;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) {
                hipStream_t st = (mode == 2) ? s[i & 1] : s[0];
                if (mode == 3) {       // cooperative launch: all blocks guaranteed co-resident
                    unsigned long long* dd = d; int slot = i; unsigned long long tk = ticks;
                    void* args[] = {&dd, &slot, &tk};
                    hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_spin), dim3(grid), dim3(256),
                                               args, 0, st);
                } else if (mode == 1)
                    hipExtLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, st, nullptr, nullptr,
                                          hipExtAnyOrderLaunch, d, i, ticks);
                else
                    hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, st, d, i, ticks);
            }
            host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            hipDeviceSynchronize();
            hipMemcpy(hb.data(), d, (size_t)n * grid * 16, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; ++i) {
                h[2 * i] = ~0ull; h[2 * i + 1] = 0;
                for (int b = 0; b < grid; ++b) {
                    h[2 * i] = std::min(h[2 * i], hb[((size_t)i * grid + b) * 2]);
                    h[2 * i + 1] = std::max(h[2 * i + 1], hb[((size_t)i * grid + b) * 2 + 1]);
                }
            }
        }
        std::vector<double> gap, per;
        for (int i = n / 2; i + 1 < n; ++i) {
            gap.push_back(((double)h[2 * i + 2] - (double)h[2 * i + 1]) * 0.01);
            per.push_back(((double)h[2 * i + 2] - (double)h[2 * i]) * 0.01);
        }
        std::sort(gap.begin(), gap.end()); std::sort(per.begin(), per.end());
        printf("mode %d (%s): start(j+1)-end(j) median %.2f us (p10 %.2f p90 %.2f), period median %.2f us; host enqueue %.2f us per launch\n",
               mode, mode == 0 ? "one stream" : mode == 1 ? "one stream, any-order flag" : mode == 2 ? "two streams in turn" : "one stream, cooperative launch",
               gap[gap.size() / 2], gap[gap.size() / 10], gap[gap.size() * 9 / 10], per[per.size() / 2], host_us);
    }
    return 0;
}
