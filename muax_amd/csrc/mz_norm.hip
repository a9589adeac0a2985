// mz_norm.hip -- translation unit of the fused LayerNorm (+ add, + relu) of the convolutional plugin nets (mz_norm.cuh)
// and of a whole residual block of the representation nets in three launches (mzs_resblock_v1: mz_repr.cuh + mz_norm.cuh).
#include <hip/hip_runtime.h>

#include <cstring>

#include "mz_host.h"
#include "mz_norm.cuh"
#include "mz_repr_host.h"

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

// ---- ResidualConvBlockV1 (muax/nn.py:118-148), stride 1, C -> C, in three launches --------------------------------
//   K1  projection and conv_0 of the input in one pass over the staged rows (mz_repr.cuh, NW = 2), each workgroup
//       leaving the (sum, sum of squares) of its outputs in fp64;
//   K2  conv_1 on relu(LayerNorm_0(conv_0)): the normalisation happens on the way into LDS, from K1's moments;
//   K3  relu(LayerNorm_1(conv_1) + LayerNorm_p(projection))  (or + x, the identity shortcut) from the moments: the
//       apply kernel of mz_norm.cuh.  The two moment passes and one apply pass of the five-call chain are gone.
static size_t resblock_moment_doubles(int batch, int blocks) { return (size_t)3 * batch * blocks * 2; }

int64_t mzs_resblock_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t channels) {
  if (batch <= 0 || height <= 0 || width <= 0 || (channels != 32 && channels != 64)) return 0;
  const mzr::Geometry g = mzr::geometry(height, width, channels);
  const size_t n = (size_t)batch * height * width * channels;
  return (int64_t)(3 * n * sizeof(float) + resblock_moment_doubles(batch, g.blocks) * sizeof(double));
}

int mzs_resblock_v1(const mzs_resblock_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_resblock_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v1: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->height <= 0 || a->width <= 0 || !a->x || !a->w0 || !a->w1 || !a->y || !a->workspace)
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v1: batch / height / width / pointers");
  if (!a->ln0_scale || !a->ln0_offset || !a->ln1_scale || !a->ln1_offset || (a->w_proj && (!a->proj_scale || !a->proj_offset)))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v1: every LayerNorm needs its scale and offset");
  if (a->channels != 32 && a->channels != 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resblock_v1: channels must be 32 or 64 (in == out)");
  if (a->workspace_bytes < mzs_resblock_workspace_bytes(a->batch, a->height, a->width, a->channels))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v1: workspace too small");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_resblock_v1: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v1: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  const int C = a->channels, n1 = a->height * a->width * C;
  const mzr::Geometry g = mzr::geometry(a->height, a->width, C);
  if (g.lds > 160 * 1024) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resblock_v1: image too wide for the LDS of a CU");
  const size_t n = (size_t)a->batch * n1, bk2 = (size_t)a->batch * g.blocks * 2;
  float* c0 = static_cast<float*>(a->workspace);
  float* out = c0 + n;
  float* cp = out + n;
  double* m_out = reinterpret_cast<double*>(cp + n);  // [out][projection][conv_0] moments, [B][K][2] each
  double* m_cp = m_out + bk2;
  double* m_c0 = m_cp + bk2;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  mz::ReprConvParams p;
  memset(&p, 0, sizeof p);
  p.B = a->batch; p.H = a->height; p.W = a->width; p.eps = a->eps;
  p.x = a->x;
  if (a->w_proj) {
    p.wp = a->w_proj; p.y = cp; p.wp2 = a->w0; p.y2 = c0; p.mom = m_cp;  // (stream 0 -> m_cp, stream 1 -> m_c0)
    if (int rc = mzr::conv<2, false, true>(p, C, g, stream)) return rc;
  } else {
    p.wp = a->w0; p.y = c0; p.mom = m_c0;
    if (int rc = mzr::conv<1, false, true>(p, C, g, stream)) return rc;
  }
  p.x = c0; p.wp = a->w1; p.y = out; p.wp2 = nullptr; p.y2 = nullptr; p.mom = m_out;
  p.in_mom = m_c0; p.in_scale = a->ln0_scale; p.in_offset = a->ln0_offset;
  if (int rc = mzr::conv<1, true, true>(p, C, g, stream)) return rc;
  mz::NormParams q;
  memset(&q, 0, sizeof q);
  q.x = out; q.scale = a->ln1_scale; q.offset = a->ln1_offset;
  if (a->w_proj) { q.x2 = cp; q.scale2 = a->proj_scale; q.offset2 = a->proj_offset; }
  else q.residual = a->x;
  q.y = a->y; q.ws = m_out; q.B = a->batch; q.n = n1; q.C = C; q.K = g.blocks; q.relu = 1; q.eps = a->eps;
  const int per_block = 4 * mz::kNormThreads * 4;
  int slices = (n1 + per_block - 1) / per_block;
  if (slices > 65535) slices = 65535;
  hipLaunchKernelGGL(mz::ln_apply_kernel, dim3(a->batch, slices), dim3(mz::kNormThreads), 0, stream, q);
  MZS_HIPG(hipGetLastError());
  return MZS_OK;
}

// ---- ResidualConvBlockV2 (muax/nn.py:151-178), the pre-activation block of the EZ encoder (:180-207), stride 1,
// identity shortcut, C -> C:   y = x + conv_1(relu(LN_1(conv_0(relu(LN_0(x))))))   in three launches --
//   K0  (sum, sum of squares) of x per chunk in fp64 (mz_norm.cuh's moments kernel);
//   K1  conv_0 with relu(LN_0(.)) applied to x on its way into LDS, leaving the moments of its raw outputs;
//   K2  conv_1 with relu(LN_1(.)) applied on the way in, the shortcut x added to the outputs in the epilogue.
// (Seven launches as single calls: moments + apply, convolution, moments + apply, convolution, add.)
int64_t mzs_resblock_v2_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t channels) {
  if (batch <= 0 || height <= 0 || width <= 0 || (channels != 16 && channels != 32 && channels != 64)) return 0;
  const mzr::Geometry g = mzr::geometry(height, width, channels);
  const size_t n1 = (size_t)height * width * channels;
  return (int64_t)((size_t)batch * n1 * sizeof(float)
                   + (size_t)batch * ((size_t)mz::norm_chunks((int)n1) + g.blocks) * 2 * sizeof(double));
}

int mzs_resblock_v2(const mzs_resblock_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_resblock_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v2: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->height <= 0 || a->width <= 0 || !a->x || !a->w0 || !a->w1 || !a->y || !a->workspace)
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v2: batch / height / width / pointers");
  if (!a->ln0_scale || !a->ln0_offset || !a->ln1_scale || !a->ln1_offset)
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v2: every LayerNorm needs its scale and offset");
  if (a->w_proj || a->proj_scale || a->proj_offset)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resblock_v2: identity shortcut only (the projection block is strided: single calls)");
  if (a->channels != 16 && a->channels != 32 && a->channels != 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resblock_v2: channels must be 16, 32 or 64 (in == out)");
  if (a->y == a->x) return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v2: y must not alias x");
  if (a->workspace_bytes < mzs_resblock_v2_workspace_bytes(a->batch, a->height, a->width, a->channels))
    return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v2: workspace too small");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_resblock_v2: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_resblock_v2: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  const int C = a->channels, n1 = a->height * a->width * C;
  const mzr::Geometry g = mzr::geometry(a->height, a->width, C);
  if (g.lds > 160 * 1024) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_resblock_v2: image too wide for the LDS of a CU");
  float* c0 = static_cast<float*>(a->workspace);
  double* m_x = reinterpret_cast<double*>(c0 + (size_t)a->batch * n1);  // [B][Kx][2]
  const int Kx = mz::norm_chunks(n1);
  double* m_c0 = m_x + (size_t)a->batch * Kx * 2;                       // [B][g.blocks][2]
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  mz::NormParams q;
  memset(&q, 0, sizeof q);
  q.x = a->x; q.ws = m_x; q.B = a->batch; q.n = n1; q.C = C; q.K = Kx; q.eps = a->eps;
  q.chunk = ((n1 / 4 + Kx - 1) / Kx) * 4;
  hipLaunchKernelGGL(mz::ln_moments_kernel, dim3(a->batch, Kx, 1), dim3(mz::kNormThreads), 0, stream, q);
  MZS_HIPG(hipGetLastError());
  mz::ReprConvParams p;
  memset(&p, 0, sizeof p);
  p.B = a->batch; p.H = a->height; p.W = a->width; p.eps = a->eps;
  p.x = a->x; p.wp = a->w0; p.y = c0; p.mom = m_c0;
  p.in_mom = m_x; p.in_K = Kx; p.in_scale = a->ln0_scale; p.in_offset = a->ln0_offset;
  if (int rc = mzr::conv<1, true, true>(p, C, g, stream)) return rc;
  p.x = c0; p.wp = a->w1; p.y = a->y; p.mom = nullptr;
  p.in_mom = m_c0; p.in_K = 0; p.in_scale = a->ln1_scale; p.in_offset = a->ln1_offset;
  p.residual = a->x;
  return mzr::conv<1, true, false>(p, C, g, stream);
}

}  // extern "C"
