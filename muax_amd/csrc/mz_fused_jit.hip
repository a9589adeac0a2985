// mz_fused_jit.hip -- ONE instance of the fused act() kernel, built on demand into a side library.
//
// The reference's act() takes any num_simulations / network widths (muax/model.py:82-96, muax/nn.py:59-115); the fused
// kernel is a template over (num_actions, embedding_dim, support slots, tree size, wavefronts per workgroup) and
// libmzsearch.so carries the instances listed in mz_instances.def.  When mzs_act_mlp finds none for a shape, the host
// side (muax_amd/_jit.py) compiles THIS translation unit with
//     -DMZ_INSTANCES_FILE="<one-line .def>" -DMZ_FUSED_GROUP=<group of that line>
// into muax_amd/lib/jit/<shape>.so, loads it and hands mzs_jit_dispatch() to mzs_register_fused_dispatch(): the shape is
// then served by one launch per act() like a listed one, with the same kernel source and the same bits.
#ifndef MZ_FUSED_GROUP
#error "build through muax_amd/_jit.py"
#endif
#include "mz_fused_group.inc"

extern "C" void* mzs_jit_dispatch(void) { return reinterpret_cast<void*>(&mz::MZ_CAT(fused_dispatch_g, MZ_FUSED_GROUP)); }
extern "C" int mzs_jit_abi(void) { return MZS_ABI_VERSION * 1000 + (int)(sizeof(mz::FusedParams) % 1000); }
