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
 *   the noise of one sample in one solve is the flat sequence n = t*A + a, n < T*A;
 *   NBT  = ceil(T*A / 4)                         Philox blocks per sample per solve
 *   block b of global sample k in solve j: rocrand_init(seed, subsequence = k,
 *          offset = 4 * (j * NBT + b)), then ONE rocrand4 -> words (x, y, z, w)
 *   (z[0], z[1]) = BM(x, y), (z[2], z[3]) = BM(z, w) with the Box-Muller of the kernels:
 *          u = 2^-32 + (float)x * 2^-32,  theta = 2^-32 + (float)y * 2^-32  (revolutions)
 *          s = sqrt((-2 ln 2) * log2(u)),  BM = (s * sin(2 pi theta), s * cos(2 pi theta))
 *   (rocRAND's own box_muller uses the same uniforms and logf/sincosf; the kernels use the
 *   gfx950 v_log/v_sqrt/v_sin/v_cos units, which take log2 and revolutions directly)
 *   E[k][t][a] = sigma[a] * z[n % 4],  b = n / 4
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
#include <cmath>

extern "C" {

static void box_muller_ref(unsigned int x, unsigned int y, float* z0, float* z1)
{
    const float kInv = 2.3283064e-10f;
    const float u = kInv + (float)x * kInv;
    const float th = kInv + (float)y * kInv;
    const float r2 = -1.3862943611198906f * log2f(u);
    const float s = sqrtf(r2);
    const double ang = 6.283185307179586476925 * (double)th;
    *z0 = (float)sin(ang) * s;
    *z1 = (float)cos(ang) * s;
}

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
    const int TA = T * A;
    const unsigned long long NBT = (unsigned long long)((TA + 3) / 4);
    for (int k = 0; k < K; k++) {
        for (unsigned long long b = 0; b < NBT; b++) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, k_offset + (unsigned long long)k, 4ull * (solve_index * NBT + b),
                         &st);
            const uint4 r = rocrand4(&st);
            float z[4];
            box_muller_ref(r.x, r.y, &z[0], &z[1]);
            box_muller_ref(r.z, r.w, &z[2], &z[3]);
            for (int i = 0; i < 4; i++) {
                const int n = (int)b * 4 + i;
                if (n < TA) E[(size_t)k * TA + n] = sigma[n % A] * z[i];
            }
        }
    }
    return 0;
}

/* the same block through rocRAND's own normal transform, for the rocRAND-agreement test */
void orc_noise_block_rocrand_normal(unsigned long long seed, unsigned long long k_global,
                                    unsigned long long block_index, float out[4])
{
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, k_global, 4ull * block_index, &st);
    float4 z = rocrand_normal4(&st);
    out[0] = z.x; out[1] = z.y; out[2] = z.z; out[3] = z.w;
}

}  // extern "C"
