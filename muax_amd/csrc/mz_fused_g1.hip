// mz_fused_g1.hip -- group 1 of the fused act() kernel instances (mz_instances.def); see mz_fused_launch.h.
#define MZ_FUSED_GROUP 1
#include "mz_fused_group.inc"
