// mz_fused_g4.hip -- group 4 of the fused act() kernel instances (mz_instances.def); see mz_fused_launch.h.
#define MZ_FUSED_GROUP 4
#include "mz_fused_group.inc"
