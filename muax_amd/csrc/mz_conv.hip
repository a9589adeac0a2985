// mz_conv.hip -- translation unit of the ResNet recurrent kernel (mz_conv.cuh) and its C-ABI entry points.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mz_host.h"
#include "mz_conv.cuh"

extern "C" {

// ---------------------------------------------------------------------------
// ResNet dynamics: next-state tower
// ---------------------------------------------------------------------------
int mzs_resnet_tower(const mzs_tower_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_tower_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->blocks < 0) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: batch / blocks");
  if (!a->x || !a->y || (a->blocks > 0 && (!a->conv_w || !a->ln)))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: null tensor pointer");
  if (a->stem_w && (!a->action || a->num_actions <= 0))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: the stem needs actions and num_actions");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_resnet_tower: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::TowerParams p;
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
  const size_t lds = sizeof(float) * (2 * (size_t)mz::kBufWords + mz::kHeadWords);
  static bool tower_attr_dev[64] = {};  // per device: one process may drive several GPUs
  bool& tower_attr = tower_attr_dev[a->device & 63];
  if (!tower_attr) {
    MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_resnet_tower_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tower_attr = true;
  }
  if (a->pair_scratch) {
    const int64_t need = mzs_tower_pair_scratch_bytes(a->batch);
    if (need == 0) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_tower: pair mode needs batch <= 128");
    if (a->pair_scratch_bytes < need) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: pair_scratch too small");
    if (2 * a->blocks + 1 > mz::kPairMsgs) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_tower: too many blocks for pair mode");
    p.pair_f = static_cast<float*>(a->pair_scratch);
    p.pair_u = reinterpret_cast<unsigned*>(p.pair_f + (size_t)a->batch * 4 * mz::kPairSlot);
    static bool pair_attr_dev[64] = {};
    bool& pair_attr = pair_attr_dev[a->device & 63];
    if (!pair_attr) {
      MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_resnet_tower_pair_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      pair_attr = true;
    }
    const int groups = (a->batch + 7) / 8;  // 16 blocks = 8 roots x 2 halves
    hipLaunchKernelGGL(mz::mz_resnet_tower_pair_kernel, dim3(16 * groups), dim3(256), lds,
                       static_cast<hipStream_t>(stream_), p);
    MZS_HIPG(hipGetLastError());
    return MZS_OK;
  }
  hipLaunchKernelGGL(mz::mz_resnet_tower_kernel, dim3(a->batch), dim3(256), lds, static_cast<hipStream_t>(stream_), p);
  MZS_HIPG(hipGetLastError());
  return MZS_OK;
}

#ifdef MZ_PROFILE
// profiling builds only (tools/profile_tower.py): read and clear the per-workgroup phase counters
int mzs_debug_tower_profile(uint64_t* host_out, int32_t words) {
  static unsigned long long zero[1024 * 16];
  if (words > 1024 * 16) words = 1024 * 16;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mz::g_tower_prof), sizeof(uint64_t) * (size_t)words) != hipSuccess) return MZS_E_RUNTIME;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mz::g_tower_prof), zero, sizeof(zero)) != hipSuccess) return MZS_E_RUNTIME;
  return MZS_OK;
}
#endif

int64_t mzs_tower_pair_scratch_bytes(int32_t batch) {
  if (batch <= 0 || batch > 128) return 0;  // 2 * batch workgroups have to be resident together
  return (int64_t)batch * (4 * mz::kPairSlot * (int64_t)sizeof(float) + 4 * (int64_t)sizeof(unsigned));
}

}  // extern "C"
