// sequence_length 8 ... 15 (round 6; the reference has no limit, rl_toy_env.py:1475-1506): the general step kernel and the reset kernel of
// mdpp_discrete.hip compiled a third time with a history of sixteen byte fields (k_discrete_step_long / k_discrete_reset_long).
// S <= 255, S^L < 4e9 keys; no specialised kernel serves such a handle.
#define MDPP_D_LONG 1
#include "mdpp_discrete.hip"
