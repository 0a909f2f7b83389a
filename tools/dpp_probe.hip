// dpp_probe.hip -- what the DPP controls used by the chunk scan move where, on the device itself
// (device_common.hpp cites this): every lane holds its lane id; prints the lane each lane reads.
//   row_shr:D      0x110+D   lane i <- lane i-D inside its 16-lane row (0 shifted in)
//   row_bcast:15   0x142     rows 1 and 3 <- lane 15 of the row below     (row_mask 0xA)
//   row_bcast:31   0x143     rows 2 and 3 <- lane 31                      (row_mask 0xC)
//   wave_shr:1     0x138     lane i <- lane i-1 across the whole wave
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROWMASK, bool BOUND>
__global__ void k(int* out)
{
    const int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xf, BOUND);
}
template <int CTRL, int ROWMASK, bool BOUND>
void show(const char* name, int* d)
{
    int h[64];
    k<CTRL, ROWMASK, BOUND><<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-14s:", name);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i] ? h[i] - 100 : -1);
    printf("\n");
}
int main()
{
    int* d;
    (void)hipMalloc(&d, 64 * sizeof(int));
    show<0x111, 0xf, true>("row_shr:1", d);
    show<0x118, 0xf, true>("row_shr:8", d);
    show<0x142, 0xA, false>("row_bcast:15", d);
    show<0x143, 0xC, false>("row_bcast:31", d);
    show<0x138, 0xf, true>("wave_shr:1", d);
    return 0;
}
