// explicit instantiation of the fused rollout launcher for act_dim = 1
#include "rollout_fused_impl.hpp"
namespace mppi {
template hipError_t launch_fused_a<1>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
template int fused_blocks_per_cu_a<1>(int, bool, size_t, bool);
}
