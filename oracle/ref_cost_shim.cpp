/*
 * ref_cost_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points over the REFERENCE's own Cost class (include/cost.hpp, src/cost.cu), which
 * oracle/Makefile compiles UNMODIFIED from /root/reference with `hipcc -x hip
 * --cuda-host-only` (hipcc understands __host__ __device__ natively, so no stand-in header
 * or macro is involved) into oracle/_ref/libref_cost.so.  Used to pin orc_step_cost /
 * orc_final_cost and to generate tests/golden/cost_ref_*.npz.  Only built where
 * /root/reference exists; never shipped, never on the product path.
 */
#include "cost.hpp"  // the reference's header, found through -I/root/reference/include

extern "C" {

float ref_step_cost(float* x, float* u, float* e, float* w, float* goal, float lambda,
                    float* inv_s, int S, int A)
{
    Cost c(w, S, goal, S, lambda, inv_s, A);
    return c.step_cost(x, u, e, 0, 0);
}

float ref_final_cost(float* x, float* w, float* goal, int S)
{
    float inv_s_dummy[1] = {1.0f};
    Cost c;
    c.init(w, S, goal, S, 1.0f, inv_s_dummy, 0);
    return c.final_cost(x, 0);
}

}  // extern "C"
