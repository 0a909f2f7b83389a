// micro-benchmark: cost of the 32x32->64 multiply forms used by Philox on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned* out, int iters)
{
    unsigned a = threadIdx.x * 2654435761u + 1, b = blockIdx.x + 7, c = 0x9E3779B9u, d = a ^ b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) {   // v_mad_u64_u32 (one instruction: hi and lo)
                unsigned long long m = (unsigned long long)a * 0xD2511F53u;
                unsigned long long n = (unsigned long long)d * 0xCD9E8D57u;
                a = (unsigned)(n >> 32) ^ b ^ c; b = (unsigned)n; d = (unsigned)(m >> 32) ^ d; c += 0x9E3779B9u;
                d ^= (unsigned)m;
            } else if (MODE == 1) {  // mul_hi + mul_lo
                unsigned h0 = __umulhi(a, 0xD2511F53u), l0 = a * 0xD2511F53u;
                unsigned h1 = __umulhi(d, 0xCD9E8D57u), l1 = d * 0xCD9E8D57u;
                a = h1 ^ b ^ c; b = l1; d = h0 ^ d ^ l0; c += 0x9E3779B9u;
            } else if (MODE == 2) {  // plain xor/add chain of the same length (baseline)
                a = (a ^ b) + c; b = (b ^ d) + a; d = (d ^ a) + b; c += 0x9E3779B9u; a ^= d; b ^= c;
            } else {                 // fma chain
                float x = __uint_as_float(a), y = __uint_as_float(b);
                x = fmaf(x, 1.0001f, y); y = fmaf(y, 0.9999f, x); x = fmaf(x, 1.0001f, y); y = fmaf(y, 0.9999f, x);
                x = fmaf(x, 1.0001f, y); y = fmaf(y, 0.9999f, x);
                a = __float_as_uint(x); b = __float_as_uint(y);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
template <int MODE> float run(unsigned* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * 8, 256>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256 * 8, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 2000;
    const double waves = 256.0 * 8 * 4, per_simd = waves / 1024.0;
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
    // per inner body cycles per SIMD-wave at 2.4 GHz nominal
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (per_simd * iters * 16.0); };
    printf("mad_u64_u32 body (2 mad + 6 alu): %.3f ms  %.1f cyc/body/wave-slot\n", t0, cyc(t0));
    printf("mul_hi+mul_lo body (4 mul + 5 alu): %.3f ms  %.1f cyc\n", t1, cyc(t1));
    printf("alu-only body (9 alu): %.3f ms  %.1f cyc\n", t2, cyc(t2));
    printf("fma body (6 fma): %.3f ms  %.1f cyc\n", t3, cyc(t3));
    return 0;
}
