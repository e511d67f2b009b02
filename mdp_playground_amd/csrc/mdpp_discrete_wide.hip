// State spaces of 256 ... 65 535 states (round 6; the reference has no limit, rl_toy_env.py:1050-1151): the general step kernel and the reset
// kernel of mdpp_discrete.hip compiled a second time with 16-bit transition-table entries and 16-bit history fields
// (k_discrete_step_wide / k_discrete_reset_wide).  No specialised kernel serves such a handle.
#define MDPP_D_WIDE 1
#include "mdpp_discrete.hip"
