// explicit instantiation of the packed rollout launcher for act_dim = 4
#include "rollout_packed_impl.hpp"
namespace mppi {
template hipError_t launch_packed_a<4>(int, bool, int, const RolloutArgs&, const DeferredCombine&, hipStream_t, LaunchTiming);
template int packed_blocks_per_cu_a<4>(int, bool, size_t, bool, bool);
template size_t packed_lds_bytes_a<4>(int, int, int);
}
