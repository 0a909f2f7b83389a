// micro-benchmark: VALU issue rate of gfx950 as a function of waves per SIMD and of the
// instruction-level parallelism inside one wave, for the instruction classes the rollout kernel
// is made of (v_fma_f32, v_mad_u64_u32, v_bitop3_b32, transcendentals, DPP moves).
// Prints SIMD-cycles per wave-instruction (2.4 GHz nominal and by s_memtime-free wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int ILP>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float x[ILP];
    unsigned u[ILP], w[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
        x[j] = 1.0f + 1e-3f * (threadIdx.x + j);
        u[j] = threadIdx.x * 2654435761u + j;
        w[j] = u[j] ^ 0x9E3779B9u;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                if (MODE == 0) {          // dependent fma chain per j
                    x[j] = fmaf(x[j], 1.0001f, 0.5f);
                } else if (MODE == 1) {   // v_mad_u64_u32 chain (hi word feeds the next)
                    const unsigned long long p = (unsigned long long)u[j] * 0xD2511F53u + w[j];
                    u[j] = (unsigned)(p >> 32);
                    w[j] = (unsigned)p;
                } else if (MODE == 2) {   // bitop3 chain
                    u[j] = __builtin_amdgcn_bitop3_b32(u[j], w[j], 0x9E3779B9u, 0x96);
                } else if (MODE == 3) {   // transcendental chain
                    x[j] = __builtin_amdgcn_sinf(x[j]);
                } else if (MODE == 4) {   // DPP mov + add
                    x[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[j]), 0x111, 0xf, 0xf, true));
                } else if (MODE == 5) {   // one Philox round: 2 mad_u64 + 2 bitop3
                    const unsigned long long p0 = (unsigned long long)0xD2511F53u * u[j];
                    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * w[j];
                    u[j] = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), (unsigned)p0, 0x9E3779B9u + r, 0x96);
                    w[j] = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), (unsigned)p1, 0xBB67AE85u + r, 0x96);
                }
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc += x[j] + __uint_as_float((u[j] ^ w[j]) & 0x007fffffu);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int ILP>
void run(float* d, const char* name, int insts_per_body)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-28s ILP=%d :", name, ILP);
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD: 256 CUs x wps blocks of 4 waves
        const int grid = 256 * wps;
        k<MODE, ILP><<<grid, 256>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE, ILP><<<grid, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // SIMD-cycles per wave-instruction at 2.4 GHz: time * f / (instructions issued per SIMD)
        const double per_simd = (double)wps * iters * 8.0 * ILP * insts_per_body;
        printf("  w%d %.2f", wps, ms * 1e-3 * 2.4e9 / per_simd);
    }
    printf("   (cyc per wave-instr per SIMD @2.4GHz)\n");
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0, 1>(d, "v_fma_f32 chain", 1);
    run<0, 2>(d, "v_fma_f32 chain", 1);
    run<0, 4>(d, "v_fma_f32 chain", 1);
    run<0, 8>(d, "v_fma_f32 chain", 1);
    run<1, 1>(d, "v_mad_u64_u32 chain", 1);
    run<1, 2>(d, "v_mad_u64_u32 chain", 1);
    run<1, 4>(d, "v_mad_u64_u32 chain", 1);
    run<2, 1>(d, "v_bitop3 chain", 1);
    run<2, 4>(d, "v_bitop3 chain", 1);
    run<3, 1>(d, "v_sin_f32 chain", 1);
    run<3, 2>(d, "v_sin_f32 chain", 1);
    run<3, 4>(d, "v_sin_f32 chain", 1);
    run<4, 1>(d, "dpp row_shr + add", 2);
    run<4, 4>(d, "dpp row_shr + add", 2);
    run<5, 1>(d, "philox round (2 mad+2 bitop3)", 4);
    run<5, 2>(d, "philox round (2 mad+2 bitop3)", 4);
    run<5, 3>(d, "philox round (2 mad+2 bitop3)", 4);
    return 0;
}
