// point_mass_gpu.hpp -- the per-sample rollout object of the reference as a header-only value
// type: same method names and argument order as `class PointMassModelGpu`
// (reference include/point_mass_gpu.hpp:19-92, src/point_mass_gpu.cu:25-155).
//
// The engine does NOT use this type on its hot path -- there the K samples are lanes of one
// kernel, not K objects with device-heap buffers -- but code that drives single rollouts
// through the reference's object (its stale src/test.cu, a user's own kernels) keeps working,
// and the serial ControllerBase is built on it.  Differences, all deliberate:
//   - nothing is allocated: x, u, e, w, goal are used in place (the reference mallocs four
//     device-heap arrays per sample in init() and never frees them, src/point_mass_gpu.cu:52-58);
//   - the RNG state is a rocRAND Philox state; step() draws sigma * normal per action axis when
//     a state is passed and uses the noise already in e[] when it is null (the reference draws
//     0.025 * curand_normal on the device and adds 0 on the host, :84-95);
//   - set_x() can be called before or after init() (the reference reads _x_size before
//     assigning it, :42 vs :46).
#ifndef MPPI_GPU_AMD_POINT_MASS_GPU_HPP_
#define MPPI_GPU_AMD_POINT_MASS_GPU_HPP_

#include "cost.hpp"

#if defined(__HIPCC__)
#include <rocrand/rocrand_kernel.h>
typedef rocrand_state_philox4x32_10 mppi_rng_state;
#else
struct mppi_rng_state;   // host-only translation units can only pass a null state
#endif

class PointMassModelGpu {
public:
    MPPI_HD PointMassModelGpu()
        : traj_(nullptr), controls_(nullptr), noise_(nullptr), gain_x_(nullptr), gain_u_(nullptr),
          horizon_(0), n_state_(0), n_act_(0), path_cost_(0.0f), sigma_(0.025f), tag_(0),
          chatty_(false)
    {
        for (int i = 0; i < 4; ++i) unit_inv_sigma_[i] = 1.0f;
    }

    // x: (steps+1)*x_size floats, row 0 receives init; u: steps*u_size; e: steps*u_size
    // x_gain = {1, dt, 0, 1}, u_gain = {dt*dt/2, dt}  (reference src/point_mass.cu:46-51)
    MPPI_HD void init(float* x, float* init, float* u, float* e, int steps, float* x_gain,
                      int x_size, float* u_gain, int u_size, float* w, float* goal, float lambda,
                      int id, bool verbose = false)
    {
        traj_ = x;
        controls_ = u;
        noise_ = e;
        horizon_ = steps;
        gain_x_ = x_gain;
        gain_u_ = u_gain;
        n_state_ = x_size;
        n_act_ = u_size;
        tag_ = id;
        chatty_ = verbose;
        path_cost_ = 0.0f;
        stage_.init(w, x_size, goal, x_size, lambda, unit_inv_sigma_, u_size < 4 ? u_size : 4);
        set_x(init);
    }

    // one step of x_{t+1} = A x_t + B (u_t + e_t), cost accumulated on x_{t+1}
    MPPI_HD void step(mppi_rng_state* state, int t)
    {
        float* e = &noise_[t * n_act_];
#if defined(__HIP_DEVICE_COMPILE__)
        if (state)
            for (int a = 0; a < n_act_; ++a) e[a] = sigma_ * rocrand_normal(state);
#else
        (void)state;
#endif
        const int h = n_state_ / 2;
        const float* xc = &traj_[t * n_state_];
        float* xn = &traj_[(t + 1) * n_state_];
        for (int a = 0; a < n_act_; ++a) {
            const float drive = controls_[t * n_act_ + a] + e[a];
            xn[a] = gain_x_[0] * xc[a] + gain_x_[1] * xc[a + h] + gain_u_[0] * drive;
            xn[a + h] = gain_x_[2] * xc[a] + gain_x_[3] * xc[a + h] + gain_u_[1] * drive;
        }
        path_cost_ += stage_.step_cost(xn, &controls_[t * n_act_], e, tag_, t);
    }

    // full rollout: T steps + terminal cost (the terminal state is counted twice, like the
    // reference: in the last stage and in final_cost, src/point_mass_gpu.cu:107,116)
    MPPI_HD float run(mppi_rng_state* state)
    {
        path_cost_ = 0.0f;
        for (int t = 0; t < horizon_; ++t) step(state, t);
        path_cost_ += stage_.final_cost(&traj_[horizon_ * n_state_], tag_);
        save_e();
        return path_cost_;
    }

    MPPI_HD void save_e() {}                       // noise is written in place; nothing to copy
    MPPI_HD void set_x(float* x)
    {
        if (traj_ && x)
            for (int s = 0; s < n_state_; ++s) traj_[s] = x[s];
    }
    MPPI_HD void set_state(float* x) { traj_ = x; }
    MPPI_HD void set_horizon(int horizon) { horizon_ = horizon; }
    MPPI_HD void set_sigma(float sigma) { sigma_ = sigma; }
    MPPI_HD float* get_state() { return traj_; }
    MPPI_HD int get_horizon() { return horizon_; }

private:
    float* traj_;
    float* controls_;
    float* noise_;
    float* gain_x_;
    float* gain_u_;
    int horizon_, n_state_, n_act_;
    float path_cost_;
    float sigma_;
    float unit_inv_sigma_[4];
    Cost stage_;
    int tag_;
    bool chatty_;
};

#endif  // MPPI_GPU_AMD_POINT_MASS_GPU_HPP_
