// mz_fused_g0.hip -- group 0 of the fused act() kernel instances (mz_instances.def); see mz_fused_launch.h.
#define MZ_FUSED_GROUP 0
#include "mz_fused_group.inc"
