// launch floor: duration of near-empty kernels of different shapes (dispatch start->end)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_empty(float* o) { if (threadIdx.x == 0 && blockIdx.x == 0) o[0] = 1.f; }
template <int LDSF> __global__ void k_lds(float* o)
{
    __shared__ float s[LDSF];
    s[threadIdx.x] = threadIdx.x; __syncthreads();
    if (threadIdx.x == 0) o[blockIdx.x] = s[5];
}
template <typename F> float timeit(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float tot = 0; const int n = 200;
    for (int i = 0; i < 20; ++i) launch(nullptr, nullptr);
    hipDeviceSynchronize();
    for (int i = 0; i < n; ++i) { launch(a, b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); tot += ms; }
    return tot / n * 1e3f;
}
int main()
{
    float* d; hipMalloc(&d, 1 << 20);
    int shapes[][2] = {{1, 64}, {13, 1024}, {52, 256}, {104, 128}, {625, 256}, {1250, 256}, {2500, 64}, {313, 512}, {157, 1024}};
    for (auto& s : shapes) {
        int g = s[0], b = s[1];
        float t = timeit([&](hipEvent_t a, hipEvent_t e) { if (a) hipExtLaunchKernelGGL(k_empty, dim3(g), dim3(b), 0, 0, a, e, 0, d); else hipLaunchKernelGGL(k_empty, dim3(g), dim3(b), 0, 0, d); });
        float t2 = timeit([&](hipEvent_t a, hipEvent_t e) { if (a) hipExtLaunchKernelGGL(k_lds<4096>, dim3(g), dim3(b), 0, 0, a, e, 0, d); else hipLaunchKernelGGL(k_lds<4096>, dim3(g), dim3(b), 0, 0, d); });
        printf("grid %5d x %4d : empty %.2f us, 16KB-LDS+barrier %.2f us\n", g, b, t, t2);
    }
    return 0;
}
