// cost.hpp -- quadratic stage / terminal cost as a small value type, with the constructor,
// init(), step_cost() and final_cost() signatures of the reference's `class Cost`
// (reference include/cost.hpp:4-47, src/cost.cu:10-64), usable from host code and, when this
// header is compiled by hipcc, from device code.  The engine's kernels do not go through this
// type (they keep w/goal in registers); it exists so that code written against the reference's
// Cost keeps compiling, and it is what the serial ControllerBase evaluates.
//
//   step_cost  = lambda * sum_a u[a]*inv_s[a]*e[a] + sum_s (x[s]-goal[s]) * w[s] * (x[s]-goal[s])
//   final_cost =                                     sum_s (x[s]-goal[s]) * w[s] * (x[s]-goal[s])
// evaluated left to right in float, one rounding per operation, like the reference.
#ifndef MPPI_GPU_AMD_COST_HPP_
#define MPPI_GPU_AMD_COST_HPP_

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPPI_HD __host__ __device__
#else
#define MPPI_HD
#endif

class Cost {
public:
    MPPI_HD Cost() : weights_(nullptr), target_(nullptr), inv_sigma_(nullptr), n_state_(0),
                     n_act_(0), lambda_(1.0f) {}

    MPPI_HD Cost(float* w, int w_size, float* goal, int goal_size, float lambda, float* inv_s,
                 int u_size)
    {
        init(w, w_size, goal, goal_size, lambda, inv_s, u_size);
    }

    // non-owning: the caller keeps w, goal and inv_s alive (the reference does the same)
    MPPI_HD void init(float* w, int w_size, float* goal, int goal_size, float lambda,
                      float* inv_s, int u_size)
    {
        weights_ = w;
        target_ = goal;
        inv_sigma_ = inv_s;
        n_state_ = w_size < goal_size ? w_size : goal_size;
        n_act_ = u_size;
        lambda_ = lambda;
    }

    MPPI_HD float state_term(const float* x) const
    {
        float acc = 0.0f;
        for (int s = 0; s < n_state_; ++s) {
            const float d = x[s] - target_[s];
            acc += d * weights_[s] * d;
        }
        return acc;
    }

    // id and t are debugging tags in the reference; they are accepted and ignored
    MPPI_HD float step_cost(float* x, float* u, float* e, int /*id*/, int /*t*/) const
    {
        float acc = 0.0f;
        for (int a = 0; a < n_act_; ++a) acc += u[a] * inv_sigma_[a] * e[a];
        acc *= lambda_;
        for (int s = 0; s < n_state_; ++s) {
            const float d = x[s] - target_[s];
            acc += d * weights_[s] * d;
        }
        return acc;
    }

    MPPI_HD float final_cost(float* x, int /*id*/) const { return state_term(x); }

private:
    float* weights_;
    float* target_;
    float* inv_sigma_;
    int n_state_;
    int n_act_;
    float lambda_;
};

#endif  // MPPI_GPU_AMD_COST_HPP_
