// micro-benchmark: issue cost of the INTEGER / bit instructions around the Philox rounds on gfx950,
// by operand form (VOP2 against VOP3, SGPR against literal third operand), next to v_fma_f32 and
// the packed fp32 forms.  Same method and unit as ubench_issue.hip: SIMD-cycles per
// wave-instruction at 2.4 GHz nominal, 8 waves per SIMD, four independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(MODE_, ASM_)                                                          \
    if (MODE == MODE_) {                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                              \
            asm volatile(ASM_ : "+v"(u[j]) : "v"(w[j]), "s"(sk));                  \
    }

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned sk)
{
    unsigned u[4], w[4];
    unsigned long long q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[j] = threadIdx.x * 2654435761u + j;
        w[j] = u[j] ^ 0x9E3779B9u;
        q[j] = ((unsigned long long)u[j] << 32) | w[j];
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            BODY(0, "v_xor_b32 %0, %0, %1")
            BODY(1, "v_xor_b32 %0, %2, %0")
            BODY(4, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
            BODY(5, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
            BODY(6, "v_add_u32 %0, %0, %1")
            BODY(7, "v_cvt_f32_u32 %0, %0")
            BODY(8, "v_fma_f32 %0, %0, %1, %1")
            BODY(9, "v_mul_f32 %0, %0, %1")
            BODY(10, "v_add3_u32 %0, %0, %1, %2")
            BODY(11, "v_lshl_add_u32 %0, %0, 1, %1")
            BODY(12, "v_mov_b32 %0, %1")
            BODY(13, "v_and_or_b32 %0, %0, %1, %2")
            BODY(14, "v_mul_u32_u24 %0, %0, %1")
            BODY(15, "v_mul_lo_u32 %0, %0, %1")
            BODY(16, "v_mul_hi_u32 %0, %0, %1")
            BODY(17, "v_fmac_f32 %0, %1, %1")
            BODY(21, "v_fma_f32 %0, %0, %2, %1")
            BODY(22, "v_fma_f32 %0, %0, %1, %2")
            BODY(23, "v_fma_f32 %0, %0, 2.0, %1")
            BODY(24, "v_mul_f32 %0, %2, %0")
            BODY(25, "v_add_f32 %0, %2, %0")
            BODY(26, "v_add_f32 %0, 1.0, %0")
            BODY(27, "v_xor_b32 %0, 1, %0")
            BODY(28, "v_add3_u32 %0, %0, %1, %1")
            BODY(29, "v_fmac_f32 %0, %2, %1")
            BODY(30, "v_mul_f32 %0, 0x3fc01000, %0")
            BODY(35, "v_max_f32 %0, %0, %1")
            BODY(36, "v_min_f32 %0, %2, %0")
            BODY(37, "v_sub_f32 %0, %0, %1")
            BODY(38, "v_cndmask_b32 %0, %0, %1, vcc")
            BODY(39, "v_mov_b32 %0, %2")
            BODY(40, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
            BODY(41, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
            if (MODE == 18) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(q[j]));
            }
            if (MODE == 19) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(q[j]));
            }
            if (MODE == 32) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q[j]) : "v"(w[j]) : "vcc");
            }
            if (MODE == 33) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[j]) : "v"(u[j]), "s"(sk) : "vcc");
                    u[j] = (unsigned)(q[j] >> 32);
                }
            }
            if (MODE == 34) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[j]) : "v"(u[j]), "v"(w[j]) : "vcc");
                    u[j] = (unsigned)(q[j] >> 32);
                }
            }
            if (MODE == 20) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[j]) : "v"(w[j]), "s"(sk) : "vcc");
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc ^= u[j] ^ (unsigned)q[j] ^ (unsigned)(q[j] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(unsigned* d, const char* name)
{
    const int iters = 4000, wps = 8, grid = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, 10, 0x1234567u);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<grid, 256>>>(d, iters, 0x1234567u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double per_simd = (double)wps * iters * 8.0 * 4;
    printf("%-44s %.2f cyc per wave-instr per SIMD @2.4GHz\n", name, best * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<8>(d, "v_fma_f32 v,v,v");
    run<17>(d, "v_fmac_f32 (VOP2)");
    run<9>(d, "v_mul_f32 (VOP2)");
    run<18>(d, "v_pk_fma_f32 (2 fma per lane)");
    run<19>(d, "v_pk_mul_f32 (2 mul per lane)");
    run<0>(d, "v_xor_b32 v,v (VOP2)");
    run<1>(d, "v_xor_b32 s,v (VOP2)");
    run<4>(d, "v_bitop3_b32 v,v,s");
    run<5>(d, "v_bitop3_b32 v,v,v");
    run<6>(d, "v_add_u32 v,v (VOP2)");
    run<10>(d, "v_add3_u32 v,v,s");
    run<11>(d, "v_lshl_add_u32");
    run<13>(d, "v_and_or_b32 v,v,s");
    run<12>(d, "v_mov_b32");
    run<7>(d, "v_cvt_f32_u32");
    run<14>(d, "v_mul_u32_u24");
    run<15>(d, "v_mul_lo_u32");
    run<16>(d, "v_mul_hi_u32");
    run<20>(d, "v_mad_u64_u32 v,s,v64");
    run<32>(d, "v_mad_u64_u32 v,v,v64");
    run<33>(d, "v_mad_u64_u32 v,s,0 (hi feeds next)");
    run<34>(d, "v_mad_u64_u32 v,v,0 (hi feeds next)");
    run<21>(d, "v_fma_f32 v,s,v");
    run<22>(d, "v_fma_f32 v,v,s");
    run<23>(d, "v_fma_f32 v,2.0,v (inline const)");
    run<29>(d, "v_fmac_f32 v,s,v (VOP2)");
    run<24>(d, "v_mul_f32 s,v");
    run<30>(d, "v_mul_f32 literal,v");
    run<25>(d, "v_add_f32 s,v");
    run<26>(d, "v_add_f32 1.0,v");
    run<37>(d, "v_sub_f32 v,v");
    run<35>(d, "v_max_f32 v,v");
    run<36>(d, "v_min_f32 s,v");
    run<38>(d, "v_cndmask_b32 v,v,vcc");
    run<39>(d, "v_mov_b32 v,s");
    run<40>(d, "v_mov_b32_dpp row_shr:1");
    run<41>(d, "v_add_f32_dpp row_shr:1");
    run<27>(d, "v_xor_b32 1,v (inline const)");
    run<28>(d, "v_add3_u32 v,v,v");
    return 0;
}
