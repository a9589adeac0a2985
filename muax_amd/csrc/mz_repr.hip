// mz_repr.hip -- translation unit of the representation nets' 3x3 convolution (mz_repr.cuh) and its C-ABI entry point.
#include <hip/hip_runtime.h>

#include "mz_host.h"
#include "mz_repr.cuh"

namespace {
template <int C, int TPW>
int launch(const mz::ReprConvParams& p, int blocks, size_t lds, hipStream_t stream) {
  static size_t granted[64] = {};
  int dev = 0;
  MZS_HIPG(hipGetDevice(&dev));
  if (lds > granted[dev & 63]) {
    MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_repr_conv3x3_kernel<C, TPW>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    granted[dev & 63] = lds;
  }
  hipLaunchKernelGGL((mz::mz_repr_conv3x3_kernel<C, TPW>), dim3(blocks, p.B), dim3(256), lds, stream, p);
  MZS_HIPG(hipGetLastError());
  return MZS_OK;
}
}  // namespace

extern "C" {

int mzs_conv3x3_nhwc(const mzs_conv3x3_args* a, void* stream_) {
  if (!a || a->struct_size != (int32_t)sizeof(mzs_conv3x3_args))
    return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_nhwc: null arguments or size mismatch (ABI)");
  if (a->batch <= 0 || a->height <= 0 || a->width <= 0 || !a->x || !a->w_packed || !a->y)
    return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_nhwc: batch / height / width / pointers");
  if (a->channels != 32 && a->channels != 64)
    return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_conv3x3_nhwc: channels must be 32 or 64 (in == out)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mzh::fail_global(MZS_E_NODEVICE, "mzs_conv3x3_nhwc: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return mzh::fail_global(MZS_E_INVALID, "mzs_conv3x3_nhwc: bad device ordinal");
  MZS_HIPG(hipSetDevice(a->device));
  mz::ReprConvParams p;
  p.x = a->x; p.wp = a->w_packed; p.y = a->y; p.B = a->batch; p.H = a->height; p.W = a->width; p.relu = a->relu;
  const int C = a->channels, npix = a->height * a->width, tiles = (npix + 15) / 16;
  // tiles per workgroup: 14 / 8 / 4 (the block sizes compiled); small maps take the small block so that a batch of 128
  // images still covers the chip
  const int bt = tiles > 16 ? 14 : (tiles > 8 ? 8 : 4);
  const int blocks = (tiles + bt - 1) / bt;
  const size_t lds = sizeof(float) * (size_t)mz::repr_conv_rows(16 * bt, a->width) * (a->width + 2) * (C + 4);
  if (lds > 160 * 1024) return mzh::fail_global(MZS_E_UNSUPPORTED, "mzs_conv3x3_nhwc: image too wide for the LDS of a CU");
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (C == 64) {
    if (bt == 14) return launch<64, 14>(p, blocks, lds, s);
    if (bt == 8) return launch<64, 8>(p, blocks, lds, s);
    return launch<64, 4>(p, blocks, lds, s);
  }
  if (bt == 14) return launch<32, 7>(p, blocks, lds, s);
  if (bt == 8) return launch<32, 4>(p, blocks, lds, s);
  return launch<32, 2>(p, blocks, lds, s);
}

}  // extern "C"
