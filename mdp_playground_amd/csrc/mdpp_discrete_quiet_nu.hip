// The non-unit-reward instantiations of k_discrete_rollout_quiet (UR = false, see mdpp_discrete_quiet.hip), in their own
// translation unit so that they compile beside the unit-reward ones.
#define MDPP_QUIET_TU_NU 1
#include "mdpp_discrete_quiet.hip"
