// mz_conv_host.h -- host-side argument checking shared by the entry points that launch the ResNet recurrent kernel
// (mzs_resnet_tower in mz_conv.hip, mzs_resnet_search in mz_search_conv.hip).  Not part of the ABI.
#pragma once
#include <cstring>

#include "mz_host.h"
#include "mz_conv.cuh"

// validates `a` (messages say mzs_resnet_tower: the argument block is that entry point's), selects the device and fills p
// (`io_in_tree`: the fused search reads and writes the tree's own embedding rows -- x / y / action are not needed)
static inline int tower_params_from_args(const mzs_tower_args* a, mz::TowerParams& p, bool io_in_tree = false) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_tower_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->blocks < 0) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: batch / blocks");
  if ((!io_in_tree && (!a->x || !a->y)) || (a->blocks > 0 && (!a->conv_w || !a->ln)))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: null tensor pointer");
  if (a->stem_w && ((!io_in_tree && !a->action) || a->num_actions <= 0))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: the stem needs actions and num_actions");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_resnet_tower: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  memset(&p, 0, sizeof p);
  if (a->r_c1) {
    const float* const* hp = &a->r_c1;
    for (int i = 0; i < 17; ++i)
      if (!hp[i]) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: heads need all 17 weight arrays");
    if (!a->reward || !a->value || !a->prior_logits || !a->stem_w || !a->normalize)
      return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: heads need the stem, normalisation and the three outputs");
    if (a->support_size <= 0 || 2 * a->support_size + 1 > 64 || a->num_actions > 64)
      return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_tower: support / action count above 64");
    p.heads = 1; p.A = a->num_actions; p.support = a->support_size; p.F = 2 * a->support_size + 1;
    p.r_c1 = a->r_c1; p.r_c2 = a->r_c2; p.r_l1 = a->r_l1; p.r_b1 = a->r_b1; p.r_l2 = a->r_l2; p.r_b2 = a->r_b2;
    p.v_c1 = a->v_c1; p.v_c2 = a->v_c2; p.v_l1 = a->v_l1; p.v_b1 = a->v_b1; p.v_l2 = a->v_l2; p.v_b2 = a->v_b2;
    p.p_c1 = a->p_c1; p.p_l1 = a->p_l1; p.p_b1 = a->p_b1; p.p_l2 = a->p_l2; p.p_b2 = a->p_b2;
    p.reward = a->reward; p.value = a->value; p.prior_logits = a->prior_logits;
  }
  p.x = a->x; p.action = a->action; p.stem_w = a->stem_w; p.conv_w = a->conv_w; p.ln = a->ln; p.y = a->y;
  p.inv_num_actions = a->stem_w ? 1.0f / (float)a->num_actions : 0.0f;
  p.B = a->batch; p.blocks = a->blocks; p.normalize = a->normalize;
  return MZS_OK;
}
