// mz_fused_g3.hip -- group 3 of the fused act() kernel instances (mz_instances.def); see mz_fused_launch.h.
#define MZ_FUSED_GROUP 3
#include "mz_fused_group.inc"
