// mz_ez.hip -- translation unit of the EfficientZero-style nets' recurrent kernel (mz_ez.cuh) and its C-ABI entry point.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mz_host.h"
#include "mz_ez.cuh"

namespace {
template <int C>
int launch_ez(const mz::EzParams& p, int device, hipStream_t stream) {
  const size_t lds = sizeof(float) * mz::EzGeom<C>::LDS_WORDS;
  static mzh::LdsGrant attr;  // per device: one process may drive several GPUs
  if (!attr.covers(device, lds)) {
    MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_ez_recurrent_kernel<C>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr.note(device, lds);
  }
  hipLaunchKernelGGL(mz::mz_ez_recurrent_kernel<C>, dim3(p.B), dim3(256), lds, stream, p);
  MZS_HIPG(hipGetLastError());
  return MZS_OK;
}
bool head_ok(const mzs_ez_head& h) { return h.ln_in && h.c1 && h.ln_mid && h.fc && h.ln_vec && h.out_w && h.out_b; }
mz::EzHead head(const mzs_ez_head& h, int n) {
  mz::EzHead o;
  o.ln_in = h.ln_in; o.c1 = h.c1; o.ln_mid = h.ln_mid; o.fc = h.fc; o.ln_vec = h.ln_vec; o.out_w = h.out_w; o.out_b = h.out_b;
  o.n = n;
  return o;
}
}  // namespace

extern "C" int mzs_ez_recurrent(const mzs_ez_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_ez_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_ez_recurrent: null arguments or size mismatch (ABI)");
  if (a->batch <= 0) return mzh::fail_global(MZS_E_INVALID, "mzs_ez_recurrent: batch must be positive");
  if (a->channels != 32 && a->channels != 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_ez_recurrent: built for 32 or 64 channels (6x6 maps)");
  if (a->support_size <= 0 || 2 * a->support_size + 1 > 64 || a->num_actions <= 0 || a->num_actions > 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_ez_recurrent: support / action count above 64");
  if (!a->x || !a->action || !a->y || !a->reward || !a->value || !a->prior_logits)
    return mzh::fail_global(MZS_E_INVALID, "mzs_ez_recurrent: null tensor pointer");
  if (!a->d_ln_in || !a->d_conv || !a->d_ln0 || !a->d_conv0 || !a->d_ln1 || !a->d_conv1 || !a->p_ln0 || !a->p_conv0 ||
      !a->p_ln1 || !a->p_conv1 || !head_ok(a->r) || !head_ok(a->v) || !head_ok(a->p))
    return mzh::fail_global(MZS_E_INVALID, "mzs_ez_recurrent: null weight pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_ez_recurrent: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_ez_recurrent: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::EzParams p;
  memset(&p, 0, sizeof p);
  p.x = a->x; p.action = a->action; p.y = a->y; p.reward = a->reward; p.value = a->value; p.prior_logits = a->prior_logits;
  p.d_ln_in = a->d_ln_in; p.d_conv = a->d_conv; p.d_ln0 = a->d_ln0; p.d_conv0 = a->d_conv0; p.d_ln1 = a->d_ln1;
  p.d_conv1 = a->d_conv1; p.p_ln0 = a->p_ln0; p.p_conv0 = a->p_conv0; p.p_ln1 = a->p_ln1; p.p_conv1 = a->p_conv1;
  p.B = a->batch; p.A = a->num_actions; p.support = a->support_size; p.F = 2 * a->support_size + 1;
  p.hr = head(a->r, p.F); p.hv = head(a->v, p.F); p.hp = head(a->p, p.A);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  return a->channels == 32 ? launch_ez<32>(p, a->device, stream) : launch_ez<64>(p, a->device, stream);
}
