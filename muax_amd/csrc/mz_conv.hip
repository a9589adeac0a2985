// mz_conv.hip -- translation unit of the ResNet recurrent kernel (mz_conv.cuh) and its C-ABI entry points.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mz_conv_host.h"

extern "C" {

// ---------------------------------------------------------------------------
// ResNet dynamics: next-state tower
// ---------------------------------------------------------------------------
int mzs_resnet_tower(const mzs_tower_args* a, void* stream_) {
  mz::TowerParams p;
  if (int rc = tower_params_from_args(a, p)) return rc;
  const size_t lds = sizeof(float) * (2 * (size_t)mz::kBufWords + mz::kHeadWords);
  static mzh::LdsGrant tower_attr;  // per device: one process may drive several GPUs
  if (!tower_attr.covers(a->device, lds)) {
    MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_resnet_tower_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tower_attr.note(a->device, lds);
  }
  if (a->pair_scratch) {
    const int64_t need = mzs_tower_pair_scratch_bytes(a->batch);
    if (need == 0) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_tower: pair mode needs batch <= 128");
    if (a->pair_scratch_bytes < need) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_tower: pair_scratch too small");
    if (2 * a->blocks + 1 > mz::kPairMsgs) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_tower: too many blocks for pair mode");
    p.pair_f = static_cast<float*>(a->pair_scratch);
    p.pair_u = reinterpret_cast<unsigned*>(p.pair_f + (size_t)a->batch * 4 * mz::kPairSlot * 2);  // (8-byte words)
    static mzh::LdsGrant pair_attr;
    if (!pair_attr.covers(a->device, lds)) {
      MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_resnet_tower_pair_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      pair_attr.note(a->device, lds);
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

// ---------------------------------------------------------------------------
// tail of root inference with the ResNet nets: last pool + min-max + prediction heads
// ---------------------------------------------------------------------------
int mzs_resnet_root_tail(const mzs_root_tail_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_root_tail_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_root_tail: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || !a->x || !a->embedding || !a->value || !a->prior_logits)
    return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_root_tail: batch / pointers");
  if ((a->height + 1) / 2 != mz::kTowerHW || (a->width + 1) / 2 != mz::kTowerHW)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_root_tail: the pooled map must be 6 x 6 (height, width in {11, 12})");
  const float* const* hp = &a->v_c1;
  for (int i = 0; i < 11; ++i)
    if (!hp[i]) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_root_tail: the heads need all 11 weight arrays");
  if (a->support_size <= 0 || 2 * a->support_size + 1 > 64 || a->num_actions <= 0 || a->num_actions > 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resnet_root_tail: support / action count above 64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_resnet_root_tail: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_resnet_root_tail: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::TowerParams p;
  memset(&p, 0, sizeof p);
  p.heads = 1; p.A = a->num_actions; p.support = a->support_size; p.F = 2 * a->support_size + 1; p.B = a->batch;
  p.v_c1 = a->v_c1; p.v_c2 = a->v_c2; p.v_l1 = a->v_l1; p.v_b1 = a->v_b1; p.v_l2 = a->v_l2; p.v_b2 = a->v_b2;
  p.p_c1 = a->p_c1; p.p_l1 = a->p_l1; p.p_b1 = a->p_b1; p.p_l2 = a->p_l2; p.p_b2 = a->p_b2;
  p.value = a->value; p.prior_logits = a->prior_logits;
  mz::RootTailParams t;
  t.x = a->x; t.embedding = a->embedding; t.H = a->height; t.W = a->width; t.normalize = a->normalize;
  const size_t lds = sizeof(float) * (2 * (size_t)mz::kBufWords + mz::kHeadWords);
  static mzh::LdsGrant tail_attr;
  if (!tail_attr.covers(a->device, lds)) {
    MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_resnet_root_tail_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tail_attr.note(a->device, lds);
  }
  hipLaunchKernelGGL(mz::mz_resnet_root_tail_kernel, dim3(a->batch), dim3(256), lds, static_cast<hipStream_t>(stream_), p, t);
  MZS_HIPG(hipGetLastError());
  return MZS_OK;
}

int64_t mzs_tower_pair_scratch_bytes(int32_t batch) {
  if (batch <= 0 || batch > 128) return 0;  // 2 * batch workgroups have to be resident together
  return (int64_t)batch * (4 * mz::kPairSlot * 2 * (int64_t)sizeof(float) + 4 * (int64_t)sizeof(unsigned));
}

}  // extern "C"
