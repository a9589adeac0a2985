// mz_train_jit.hip -- ONE instance of the fused training-step kernel (mz_train.cuh), built on demand into a side library.
//
// The reference's update() (muax/model.py:181-201 through jax.value_and_grad of muax/loss.py:10-88) takes whatever
// widths its nets have; libmzsearch.so carries mz_train_kernel for the (num_actions, embedding_dim, 2 support_size + 1)
// triples listed in mzs_mlp_loss_grad.  Since round 5 act() serves other shapes of the default trio through instances
// built on demand (mz_fused_jit.hip); this is the same for the training step, so that such a model's update() does not
// drop to framework autograd (14 - 22 ms against 0.06 ms at 4096 x 10).  The host side (muax_amd/_jit.py) compiles THIS
// translation unit with -DMZ_TRAIN_A=.. -DMZ_TRAIN_E=.. -DMZ_TRAIN_F=.., loads it and hands mzs_jit_train_launch to
// mzs_register_train_dispatch(): same kernel source, same arithmetic.
#if !defined(MZ_TRAIN_A) || !defined(MZ_TRAIN_E) || !defined(MZ_TRAIN_F)
#error "build through muax_amd/_jit.py"
#endif
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/mzsearch.h"
#include "mz_host.h"
#include "mz_train.cuh"

namespace {
int put(char* err, int errlen, const char* what, hipError_t e) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s: %s", what, hipGetErrorString(e));
  return MZS_E_RUNTIME;
}
}  // namespace

// (argument block of mzs_mlp_loss_grad's own launcher; the caller has validated it and selected the device)
extern "C" int mzs_jit_train_launch(const void* params, void* stream_, char* err, int errlen) {
  using C = mz::TrainCfg<MZ_TRAIN_A, MZ_TRAIN_E, MZ_TRAIN_F>;
  const mz::TrainParams& p = *static_cast<const mz::TrainParams*>(params);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t lds = sizeof(float) * ((size_t)C::WEIGHT_WORDS + (size_t)p.L * C::CK_WORDS_PER_STEP);
  if (lds > 160 * 1024) {
    if (err && errlen > 0) snprintf(err, (size_t)errlen, "unroll_steps too large for the LDS");
    return MZS_E_UNSUPPORTED;
  }
  auto kern = mz::mz_train_kernel<C>;
  // (per device, as the built-in launcher does: the attribute call is not free and update() runs every step)
  static mzh::LdsGrant granted;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return put(err, errlen, "hipGetDevice", e);
  if (!granted.covers(dev, lds)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return put(err, errlen, "hipFuncSetAttribute", e);
    granted.note(dev, lds);
  }
  hipLaunchKernelGGL(kern, dim3(p.waves / 4), dim3(256), lds, stream, p);
  if ((e = hipGetLastError()) != hipSuccess) return put(err, errlen, "training kernel launch", e);
  hipLaunchKernelGGL(mz::mz_train_reduce_kernel, dim3((p.off[18] + 31) / 32), dim3(256), 0, stream, p);
  if ((e = hipGetLastError()) != hipSuccess) return put(err, errlen, "reduction kernel launch", e);
  return MZS_OK;
}
extern "C" void mzs_jit_train_shape(int32_t* A, int32_t* E, int32_t* F) {
  *A = MZ_TRAIN_A; *E = MZ_TRAIN_E; *F = MZ_TRAIN_F;
}
extern "C" int mzs_jit_train_abi(void) { return MZS_ABI_VERSION * 1000 + (int)(sizeof(mz::TrainParams) % 1000); }
