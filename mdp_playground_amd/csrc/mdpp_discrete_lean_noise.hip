// The transition- / reward-noise instantiations of k_discrete_rollout_lean (Philox streams; see NZ in
// mdpp_discrete_lean.hip), in their own translation unit so that the three parts compile in parallel.
#define MDPP_LEAN_TU_NOISE 1
#include "mdpp_discrete_lean.hip"
