// mz_norm.hip -- translation unit of the fused LayerNorm (+ add, + relu) of the convolutional plugin nets (mz_norm.cuh).
#include <hip/hip_runtime.h>

#include <cstring>

#include "mz_host.h"
#include "mz_norm.cuh"

extern "C" {

int64_t mzs_layernorm_workspace_bytes(int32_t batch, int32_t n) {
  if (batch <= 0 || n <= 0) return 0;
  return (int64_t)2 * batch * mz::norm_chunks(n) * 2 * (int64_t)sizeof(double);
}

int mzs_layernorm_act(const mzs_layernorm_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_layernorm_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_layernorm_act: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->n <= 0 || a->channels <= 0)
    return mzh::fail_global(MZS_E_INVALID, "mzs_layernorm_act: batch, n and channels must be positive");
  if (a->n % 4 || a->channels % 4 || a->n % a->channels)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_layernorm_act: n and channels must be multiples of 4, n of channels");
  if (!a->x || !a->scale || !a->offset || !a->y || !a->workspace)
    return mzh::fail_global(MZS_E_INVALID, "mzs_layernorm_act: null tensor pointer");
  if (a->x2 && (!a->scale2 || !a->offset2))
    return mzh::fail_global(MZS_E_INVALID, "mzs_layernorm_act: the second tensor needs its scale and offset");
  if (a->workspace_bytes < mzs_layernorm_workspace_bytes(a->batch, a->n))
    return mzh::fail_global(MZS_E_INVALID, "mzs_layernorm_act: workspace too small");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_layernorm_act: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_layernorm_act: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::NormParams p;
  memset(&p, 0, sizeof p);
  p.x = a->x; p.scale = a->scale; p.offset = a->offset;
  p.x2 = a->x2; p.scale2 = a->scale2; p.offset2 = a->offset2;
  p.residual = a->residual; p.y = a->y; p.ws = static_cast<double*>(a->workspace);
  p.B = a->batch; p.n = a->n; p.C = a->channels; p.relu = a->relu; p.eps = a->eps;
  p.K = mz::norm_chunks(a->n);
  p.chunk = ((a->n / 4 + p.K - 1) / p.K) * 4;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(mz::ln_moments_kernel, dim3(a->batch, p.K, a->x2 ? 2 : 1), dim3(mz::kNormThreads), 0, stream, p);
  const int per_block = 4 * mz::kNormThreads * 4;  // four 16-byte loads per thread
  int slices = (a->n + per_block - 1) / per_block;
  if (slices > 65535) slices = 65535;  // (grid y; the kernel strides over the sample)
  hipLaunchKernelGGL(mz::ln_apply_kernel, dim3(a->batch, slices), dim3(mz::kNormThreads), 0, stream, p);
  MZS_HIPG(hipGetLastError());
  return MZS_OK;
}

}  // extern "C"
