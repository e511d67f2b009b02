// k_discrete_rollout_lean<..., PHILOX = 0, NZ>: the transition- / reward-noise instantiations on NUMPY streams, in a
// translation unit of their own (compile time).  The kernel is mdpp_discrete_lean.hip.
#define MDPP_LEAN_TU_NOISE 2
#include "mdpp_discrete_lean.hip"
