// The general instantiations (VAR != 0) of the edge kernel WITH the fused edge embedding: mpn_f2.hip compiled as a third
// translation unit (build time) - see the note at launch_edge_f2_general in mpn_f2.hip.
#define AMBER_F2_GENERAL_TU 2
#include "mpn_f2.hip"
