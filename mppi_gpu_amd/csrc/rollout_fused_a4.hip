// explicit instantiation of the fused rollout launcher for act_dim = 4
#include "rollout_fused_impl.hpp"
namespace mppi {
template hipError_t launch_fused_a<4>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
template int fused_blocks_per_cu_a<4>(int, bool, size_t, bool);
}
