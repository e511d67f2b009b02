// The next-step autoreset instantiations of k_discrete_rollout_lean (see mdpp_discrete_lean.hip), in their own
// translation unit so that the two halves compile in parallel.
#define MDPP_LEAN_TU_NEXT 1
#include "mdpp_discrete_lean.hip"
