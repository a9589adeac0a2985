// mz_repr.hip -- translation unit of the representation nets' 3x3 convolution (mz_repr.cuh) and its C-ABI entry point.
#include <hip/hip_runtime.h>

#include <cstring>

#define MZ_NORM_STATS_ONLY
#include "mz_repr_host.h"

extern "C" {

int mzs_conv3x3_nhwc(const mzs_conv3x3_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_conv3x3_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_nhwc: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->height <= 0 || a->width <= 0 || !a->x || !a->w_packed || !a->y)
    return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_nhwc: batch / height / width / pointers");
  if (a->channels != 16 && a->channels != 32 && a->channels != 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_conv3x3_nhwc: channels must be 16, 32 or 64 (in == out)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_conv3x3_nhwc: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_nhwc: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::ReprConvParams p;
  memset(&p, 0, sizeof p);
  p.x = a->x; p.wp = a->w_packed; p.y = a->y; p.B = a->batch; p.H = a->height; p.W = a->width; p.relu = a->relu;
  const mzr::Geometry g = mzr::geometry(a->height, a->width, a->channels);
  if (g.lds > 160 * 1024) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_conv3x3_nhwc: image too wide for the LDS of a CU");
  return mzr::conv<1, false, false>(p, a->channels, g, static_cast<hipStream_t>(stream_));
}

int mzs_conv3x3_stride2_nhwc(const mzs_conv3x3s_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_conv3x3s_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_stride2_nhwc: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->height <= 0 || a->width <= 0 || !a->x || !a->w_packed || !a->y)
    return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_stride2_nhwc: batch / height / width / pointers");
  // (4 | 16 -> 32: one instance, the frame stack's channels padded to 16; 4 -> 16: the EZ encoder's stem at
  // embedding_dim 32, muax/nn.py:189; 16 -> 32 / 32 -> 64: stems and the strided convolutions of projection blocks)
  const int ci = a->in_channels, co = a->out_channels;
  const bool to32 = (ci == 4 || ci == 16) && co == 32, to64 = ci == 32 && co == 64, to16 = ci == 4 && co == 16;
  if (!to32 && !to64 && !to16)
    return mzh::fail_global(MZS_E_UNSUPPORTED,
                            "mzs_conv3x3_stride2_nhwc: (in, out) channels must be (4, 16), (4, 32), (16, 32) or (32, 64)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_conv3x3_stride2_nhwc: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_stride2_nhwc: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::ReprConvParams p;
  memset(&p, 0, sizeof p);
  p.x = a->x; p.wp = a->w_packed; p.y = a->y; p.B = a->batch; p.H = a->height; p.W = a->width; p.relu = a->relu;
  p.cin_real = a->in_channels; p.in_div = a->in_div;
  const mzr::Geometry g = mzr::geometry_strided(a->height, a->width, to64 ? 32 : 16, 2, co);
  if (g.lds > 160 * 1024) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_conv3x3_stride2_nhwc: image too wide for the LDS of a CU");
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (to16) return mzr::conv_stride2<16, 16>(p, g, s);
  return to32 ? mzr::conv_stride2<32, 16>(p, g, s) : mzr::conv_stride2<64, 32>(p, g, s);
}

}  // extern "C"
