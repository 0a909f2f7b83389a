// point_mass_sharded.hpp -- the reference's controller class over the GPUs of one node, for a
// C++ host that is ONE process (the reference's own host, src/main.cu:309-396, is one).
//
// `ShardedPointMassModel` has the public members of the reference's `class PointMassModel`
// (reference include/point_mass.hpp:23-44) and the same call protocol: constructor ->
// memcpy_set_data once -> loop { get_u? -> get_act -> (get_inf?) -> set_x }.  Behind it sits one
// shard engine per device and one host worker thread per engine; the samples of a solve are split
// into contiguous ranges, the noise comes from the Philox subsequences of the GLOBAL sample
// indices (results do not depend on the number of GPUs), and every solve exchanges T*A+2 floats
// per shard -- by default through ncclAllGather (RCCL over xGMI), see mppi_gpu_amd_sharded.h.
// The reference has no multi-GPU path: this is new.
//
// Needs neither hipcc nor a GPU header to include; link libmppi_gpu_amd_sharded.so.
// Errors follow the reference (include/mppi_utils.hpp:19-25): print file:line:code, exit(1).
#ifndef MPPI_GPU_AMD_POINT_MASS_SHARDED_HPP_
#define MPPI_GPU_AMD_POINT_MASS_SHARDED_HPP_

struct mppi_sharded;   // include/mppi_gpu_amd_sharded.h

class ShardedPointMassModel {
public:
    // n_gpus = 0: every visible device.  transport: "collective" (RCCL all-gather, default),
    // "direct" (peer stores from inside the combine kernel), "copy" (hipMemcpyPeerAsync).
    ShardedPointMassModel(int nb_sim, int steps, float dt, int state_dim, int act_dim,
                          bool verbose = false, int n_gpus = 0,
                          const char* transport = "collective", const int* devices = nullptr);
    ~ShardedPointMassModel();
    ShardedPointMassModel(const ShardedPointMassModel&) = delete;
    ShardedPointMassModel& operator=(const ShardedPointMassModel&) = delete;

    // ---- the reference's members -----------------------------------------------------------
    void get_act(float* next_act);
    void memcpy_set_data(float* x, float* u, float* goal, float* w);
    void get_x(float* x);
    void memcpy_get_data(float* x_all, float* e);
    void get_inf(float* x, float* u, float* e, float* cost, float* beta, float* nabla,
                 float* weight);
    void set_x(float* x);
    void get_u(float* u);

    // ---- additions ---------------------------------------------------------------------------
    void solve_async();                         // enqueue one solve on every shard
    void sync_act(float* next_act);             // wait, read the action
    void set_params(float lambda, const float* sigma, const float* inv_s);
    void set_seed(unsigned long long seed);
    void set_noise(const float* e);             // [nb_sim][steps][act_dim], null = sample
    void set_action_limit(const float* max_a);
    int n_shards() const;
    const char* transport() const;
    mppi_sharded* handle() { return impl_; }

private:
    mppi_sharded* impl_;
};

#endif  // MPPI_GPU_AMD_POINT_MASS_SHARDED_HPP_
