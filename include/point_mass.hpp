// point_mass.hpp -- C++ drop-in for the reference's MPPI controller class.
//
// Same class name, constructor and public members as the reference's
// `class PointMassModel` (reference include/point_mass.hpp:23-44), so a host loop written
// against the reference (reference src/main.cu:309-396) compiles and links unchanged against
// libmppi_gpu_amd.so.  What is different is everything behind it: the object holds one opaque
// engine handle of the C ABI (include/mppi_gpu_amd.h) and no device pointers, kernels or
// cuRAND types leak into this header -- it needs neither hipcc nor any GPU header to include.
//
// Error behaviour follows the reference (include/mppi_utils.hpp:19-25): a failing call prints
// "API error failed <file>:<line> Returned: <code>" and exits with status 1.
#ifndef MPPI_GPU_AMD_POINT_MASS_HPP_
#define MPPI_GPU_AMD_POINT_MASS_HPP_

struct mppi_engine;   // include/mppi_gpu_amd.h

class PointMassModel {
public:
    // nb_sim samples K, steps horizon T, dt, state_dim S = 2*act_dim, act_dim A in 1..4
    PointMassModel(int nb_sim, int steps, float dt, int state_dim, int act_dim,
                   bool verbose = false);
    ~PointMassModel();
    PointMassModel(const PointMassModel&) = delete;              // the reference's raw-pointer
    PointMassModel& operator=(const PointMassModel&) = delete;   // copies would double-free

    void get_act(float* next_act);                                        // one full MPPI solve
    void memcpy_set_data(float* x, float* u, float* goal, float* w);      // x0, U, goal, weights
    void get_x(float* x);                                                 // current x0
    void memcpy_get_data(float* x_all, float* e);                         // X, E of last solve
    void get_inf(float* x, float* u, float* e, float* cost, float* beta, float* nabla,
                 float* weight);                                          // any may be null
    void set_x(float* x);
    void get_u(float* u);

    // ---- additions (not in the reference; defaults reproduce its hard-coded values) --------
    void set_params(float lambda, const float* sigma, const float* inv_s);
    void set_seed(unsigned long long seed);
    void set_noise(const float* e);            // injected-noise mode, null = sample
    void set_ref_compat(bool on);
    void set_noise_store(bool on);             // false: noise is regenerated on request, not stored
    void set_action_limit(const float* max_a);   // opt-in clamp to +-max_a[axis]; null = off
    void set_tuning(int chunks, bool strict, int max_blocks);
    mppi_engine* handle() { return engine_; }

private:
    mppi_engine* engine_;
};

#endif  // MPPI_GPU_AMD_POINT_MASS_HPP_
