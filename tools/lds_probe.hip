// How much dynamic LDS may one workgroup of this device use?  (gfx950 has 160 KB per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int n) {
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[n - 1];
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor);
    int v = 0; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0); printf("attr MaxSharedMemoryPerBlock %d\n", v);
    hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, 0); printf("attr SharedMemPerBlockOptin %d\n", v);
    float* d; hipMalloc(&d, 1024);
    for (int kb : {48, 64, 65, 96, 128, 160}) {
        const size_t bytes = (size_t)kb * 1024;
        hipError_t a = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(k, dim3(4), dim3(256), bytes, 0, d, (int)(bytes / 4));
        hipError_t l = hipGetLastError(); hipError_t s = hipDeviceSynchronize();
        float h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, bytes);
        printf("%3d KB: setattr %d launch %d sync %d result %.0f (want %zu) occupancy %d\n", kb, (int)a, (int)l, (int)s, h, bytes / 4 - 1, occ);
    }
    return 0;
}
