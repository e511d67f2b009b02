// The one-step instantiations of k_continuous_rollout_fast (K1 = true: mdpp_step on the fast continuous shape, see
// mdpp_continuous_fast.hip), in their own translation unit so that they compile beside the rollout kernels.
#define MDPP_CFAST_TU_K1 1
#include "mdpp_continuous_fast.hip"
