// k_continuous_step<DMAX = 12, OMAX, PHILOX, NL = 8>: move_along_a_line with 5 to 8 relevant dimensions (see c_line_reward in
// mdpp_continuous.hip), in its own translation unit: the 8 x 8 float64 scatter matrix costs registers that the other
// instantiations of the general kernel should not pay for, and the two parts compile in parallel.
#define MDPP_CONT_TU_LINE8 1
#include "mdpp_continuous.hip"
