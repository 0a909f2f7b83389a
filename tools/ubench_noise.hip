// micro-benchmark of pass 1a of the fused rollout in isolation: Philox4x32-10 blocks + Box-Muller
// + the 1 KiB-per-wave noise store, as a function of
//   IL     Philox blocks whose rounds are interleaved in one instruction stream (ILP),
//   STORE  0 none, 1 write-through buffer store (sc0 sc1, what the kernel does), 2 plain store,
//   waves per SIMD (grid), blocks per lane (NB).
// Prints ns per wave-block and SIMD-cycles per wave-block at 2.4 GHz nominal.
#include "../mppi_gpu_amd/csrc/device_common.hpp"

#include <cstdio>
#include <cstdlib>

using namespace mppi;

template <int IL>
__device__ __forceinline__ void philox_il(const unsigned long long (&blk)[IL], unsigned long long k,
                                          unsigned long long seed, uint4 (&out)[IL])
{
    unsigned int c0[IL], c1[IL], c2[IL], c3[IL];
#pragma unroll
    for (int b = 0; b < IL; ++b) {
        c0[b] = (unsigned int)blk[b]; c1[b] = (unsigned int)(blk[b] >> 32);
        c2[b] = (unsigned int)k; c3[b] = (unsigned int)(k >> 32);
    }
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#pragma unroll
        for (int b = 0; b < IL; ++b) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0[b];
            const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2[b];
            const unsigned int n0 = PhiloxAt::xor3((unsigned int)(p1 >> 32), c1[b], k0);
            const unsigned int n2 = PhiloxAt::xor3((unsigned int)(p0 >> 32), c3[b], k1);
            c1[b] = (unsigned int)p1; c3[b] = (unsigned int)p0; c0[b] = n0; c2[b] = n2;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int b = 0; b < IL; ++b) out[b] = make_uint4(c0[b], c1[b], c2[b], c3[b]);
}

template <int IL, int STORE, int NB>
__global__ void __launch_bounds__(256) k_noise(float* E, float* sink, unsigned long long seed, int tiles_per_block)
{
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int tb = 0; tb < tiles_per_block; ++tb) {
        const size_t tile = ((size_t)blockIdx.x * tiles_per_block + tb) * 4 + (threadIdx.x >> 6);
        const unsigned long long kglob = tile * 64 + lane;
        __amdgpu_buffer_rsrc_t rs;
        {
            const unsigned long long base = reinterpret_cast<unsigned long long>(E + tile * NB * 256);
            const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)base);
            const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(base >> 32));
            rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                   NB * 1024, 0x00020000);
        }
        float e[NB * 4];
#pragma unroll
        for (int q0 = 0; q0 < NB; q0 += IL) {
            unsigned long long blk[IL];
            uint4 r[IL];
#pragma unroll
            for (int b = 0; b < IL; ++b) blk[b] = 1000ull + q0 + b;
            philox_il<IL>(blk, kglob, seed, r);
#pragma unroll
            for (int b = 0; b < IL; ++b) {
                float z[4];
                box_muller_hw(r[b].x, r[b].y, z[0], z[1]);
                box_muller_hw(r[b].z, r[b].w, z[2], z[3]);
                const int q = q0 + b;
#pragma unroll
                for (int i = 0; i < 4; ++i) e[q * 4 + i] = 0.025f * z[i];
                typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                const v4u val = {__float_as_uint(e[q * 4]), __float_as_uint(e[q * 4 + 1]),
                                 __float_as_uint(e[q * 4 + 2]), __float_as_uint(e[q * 4 + 3])};
                if (STORE == 1) __builtin_amdgcn_raw_buffer_store_b128(val, rs, lane * 16, q * 1024, 17);
                if (STORE == 2) __builtin_amdgcn_raw_buffer_store_b128(val, rs, lane * 16, q * 1024, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NB * 4; ++i) acc += e[i];
    }
    if (acc == 123.456f) sink[threadIdx.x] = acc;
}

template <int IL, int STORE, int NB>
void run(float* E, float* sink, int wps)
{
    const int tiles_per_block = 8;
    const int grid = 256 * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k_noise<IL, STORE, NB><<<grid, 256>>>(E, sink, 1, tiles_per_block);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        k_noise<IL, STORE, NB><<<grid, 256>>>(E, sink, 1, tiles_per_block);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double wave_blocks_per_simd = (double)wps * tiles_per_block * NB;
    const double bytes = (double)grid * 4 * tiles_per_block * NB * 1024;
    printf("IL=%d store=%d NB=%2d waves/SIMD=%d : %7.1f us  %6.1f ns/wave-block/SIMD = %5.0f cyc@2.4GHz  store rate %.2f TB/s\n",
           IL, STORE, NB, wps, best * 1e3, best * 1e6 / wave_blocks_per_simd,
           best * 1e-3 * 2.4e9 / wave_blocks_per_simd, STORE ? bytes / (best * 1e-3) / 1e12 : 0.0);
}

int main()
{
    float *E, *sink;
    const size_t bytes = (size_t)256 * 8 * 4 * 8 * 12 * 1024;
    (void)hipMalloc(&E, bytes);
    (void)hipMalloc(&sink, 4096);
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<1, 0, 12>(E, sink, wps);
        run<1, 1, 12>(E, sink, wps);
        run<1, 2, 12>(E, sink, wps);
        run<2, 0, 12>(E, sink, wps);
        run<2, 1, 12>(E, sink, wps);
        run<3, 0, 12>(E, sink, wps);
        run<3, 1, 12>(E, sink, wps);
        run<4, 1, 12>(E, sink, wps);
        run<6, 1, 12>(E, sink, wps);
    }
    run<1, 1, 12>(E, sink, 3);
    run<3, 1, 12>(E, sink, 3);
    return 0;
}
