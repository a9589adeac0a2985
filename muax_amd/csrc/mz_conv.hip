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
    p.pair_u = reinterpret_cast<unsigned*>(p.pair_f + (size_t)a->batch * 4 * mz::kPairSlot * 2);  // (8-byte words)
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
  return (int64_t)batch * (4 * mz::kPairSlot * 2 * (int64_t)sizeof(float) + 4 * (int64_t)sizeof(unsigned));
}

}  // extern "C"
