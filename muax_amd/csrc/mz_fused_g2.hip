// mz_fused_g2.hip -- group 2 of the fused act() kernel instances (mz_instances.def); see mz_fused_launch.h.
#define MZ_FUSED_GROUP 2
#include "mz_fused_group.inc"
