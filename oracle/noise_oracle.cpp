/*
 * noise_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see mppi_oracle.c header).
 *
 * Host-side statement of the engine's noise stream, written against rocRAND's PUBLIC
 * device API compiled for the host (rocrand_init / rocrand_normal4 are __host__ __device__).
 * The HIP kernels reach the same Philox blocks through the counter directly; this file is
 * the independent check that both agree (integer stream bit-exact, normals to libm-vs-GPU
 * transcendental accuracy).
 *
 * Stream definition (DESIGN.md "Noise"):
 *   spb  = steps per Philox block = 4 / A        (A in 1..4; A = 3 -> 1 step, 4th normal unused)
 *   NBT  = ceil(T / spb)                         blocks per sample per solve
 *   block b of global sample k in solve j: rocrand_init(seed, subsequence = k,
 *          offset = 4 * (j * NBT + b)), then ONE rocrand_normal4 -> z[0..3]
 *   E[k][t][a] = sigma[a] * z[(t % spb) * A + a],   b = t / spb
 * The reference draws cuRAND XORWOW normals scaled by 0.025 (src/point_mass_gpu.cu:85-86,
 * src/point_mass.cu:780); cuRAND cannot be reproduced here, so noise parity is
 * distributional only and every other quantity is checked on injected E.
 *
 * Build: hipcc -x hip --cuda-host-only (host code only; needs no GPU).
 */
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstddef>
#include <cstdint>

extern "C" {

int orc_noise_spb(int A) { return A <= 4 ? 4 / A : 1; }

/* raw Philox words of block b (4 x u32), for bit-exact integer checks */
void orc_noise_block_u32(unsigned long long seed, unsigned long long k_global,
                         unsigned long long block_index, unsigned int out[4])
{
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, k_global, 4ull * block_index, &st);
    uint4 r = rocrand4(&st);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

/* E[K][T][A] (reference layout) for local samples 0..K-1 = global k_offset..k_offset+K-1 */
int orc_noise_fill(unsigned long long seed, unsigned long long solve_index,
                   unsigned long long k_offset, int K, int T, int A, const float* sigma,
                   float* E)
{
    if (A < 1 || A > 4) return -1;
    const int spb = 4 / A;
    const unsigned long long NBT = (unsigned long long)((T + spb - 1) / spb);
    for (int k = 0; k < K; k++) {
        for (unsigned long long b = 0; b < NBT; b++) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, k_offset + (unsigned long long)k, 4ull * (solve_index * NBT + b),
                         &st);
            float4 z4 = rocrand_normal4(&st);
            const float z[4] = {z4.x, z4.y, z4.z, z4.w};
            for (int i = 0; i < spb * A; i++) {
                int t = (int)b * spb + i / A;
                int a = i % A;
                if (t < T) E[((size_t)k * T + t) * A + a] = sigma[a] * z[i];
            }
        }
    }
    return 0;
}

}  // extern "C"
