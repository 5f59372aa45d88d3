// The general instantiations (VAR != 0: tanh, "outer" residual / LayerNorm) of the feature-per-lane kernels without the
// fused edge embedding, and of the node kernel: mpn_f2.hip compiled as a second translation unit - see the note at
// launch_edge_f2_general in mpn_f2.hip.
#define AMBER_F2_GENERAL_TU 1
#include "mpn_f2.hip"
