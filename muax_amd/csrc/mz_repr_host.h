// mz_repr_host.h -- host side of the representation nets' 3x3 convolution kernel (mz_repr.cuh): the launch geometry and
// the dispatch over the compiled instances, shared by mzs_conv3x3_nhwc (mz_repr.hip) and mzs_resblock_v1 (mz_norm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "mz_host.h"
#include "mz_repr.cuh"

namespace mzr {

struct Geometry {
  int tiles_per_block, blocks;
  size_t lds;
};
// tiles per workgroup: 14 / 8 / 4 (the block sizes compiled; 16 / 8 / 4 for 16 channels, where the four waves are four
// tile groups); small maps take the small block so that a batch of 128 images still covers the chip
inline Geometry geometry(int height, int width, int C) {
  Geometry g;
  const int tiles = (height * width + 15) / 16;
  g.tiles_per_block = tiles > 16 ? (C == 16 ? 16 : 14) : (tiles > 8 ? 8 : 4);
  g.blocks = (tiles + g.tiles_per_block - 1) / g.tiles_per_block;
  g.lds = sizeof(float) * (size_t)mz::repr_conv_rows(16 * g.tiles_per_block, width) * (width + 2) * (C + 4);
  return g;
}
// the strided / channel-changing variant: `height`, `width` of the INPUT, cin = the kernel's (padded) input channels;
// the longest run whose rows fit the LDS
inline Geometry geometry_strided(int height, int width, int cin, int stride, int cout = 32) {
  Geometry g;
  const int ho = (height + stride - 1) / stride, wo = (width + stride - 1) / stride;
  const int tiles = (ho * wo + 15) / 16;
  int bt = tiles > 16 ? (cout == 16 ? 16 : 14) : (tiles > 8 ? 8 : 4);
  for (;;) {
    g.lds = sizeof(float) * (size_t)mz::repr_conv_rows(16 * bt, wo, stride) * ((wo - 1) * stride + 3) * (cin + 4);
    // (the frame stem, 16 padded input channels: its matrix work per staged byte is small -- two workgroups per CU, each
    // staging while the other multiplies, beat one long run: 49 against 60 us for 128 frames; the 32 -> 64 stem: 32 / 34)
    const size_t cap = cin <= 16 ? 80 * 1024 : 160 * 1024;
    if (g.lds <= cap || bt == 4) break;
    bt = bt >= 14 ? 8 : 4;
  }
  g.tiles_per_block = bt;
  g.blocks = (tiles + bt - 1) / bt;
  return g;
}

template <int C, int TPW, int NW, bool LNIN, bool MOM, int CIN = C, int STRIDE = 1>
int launch(const mz::ReprConvParams& p, int blocks, size_t lds, hipStream_t stream) {
  static mzh::LdsGrant granted;
  int dev = 0;
  MZS_HIPG(hipGetDevice(&dev));
  if (!granted.covers(dev, lds)) {
    MZS_HIPG(hipFuncSetAttribute(reinterpret_cast<const void*>(mz::mz_repr_conv3x3_kernel<C, TPW, NW, LNIN, MOM, CIN, STRIDE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    granted.note(dev, lds);
  }
  // images in grid y: at most 65535 per launch (larger batches go in slices; the moment arrays are [tensor][B][K][2], so
  // a slice keeps the full batch as its stride and only shifts the image index)
  for (int b0 = 0; b0 < p.B; b0 += 65535) {
    mz::ReprConvParams q = p;
    const int nb = p.B - b0 < 65535 ? p.B - b0 : 65535;
    const int ho = (p.H + STRIDE - 1) / STRIDE, wo = (p.W + STRIDE - 1) / STRIDE;
    const size_t off = (size_t)b0 * ho * wo * C;
    q.x = p.x + (size_t)b0 * p.H * p.W * ((CIN == C && STRIDE == 1) ? C : p.cin_real);
    q.y = p.y + off;
    if (p.y2) q.y2 = p.y2 + off;
    q.b0 = b0;
    hipLaunchKernelGGL((mz::mz_repr_conv3x3_kernel<C, TPW, NW, LNIN, MOM, CIN, STRIDE>), dim3(blocks, nb), dim3(256), lds, stream, q);
    MZS_HIPG(hipGetLastError());
  }
  return MZS_OK;
}

// one variant (NW convolutions of the input, LayerNorm + relu on the way in, moments of the outputs) on the instance
// the geometry picks
template <int NW, bool LNIN, bool MOM>
int conv(const mz::ReprConvParams& p, int C, const Geometry& g, hipStream_t s) {
  const int bt = g.tiles_per_block;
  if constexpr (NW == 1) {
    if (C == 16) {  // (the EZ encoder's first block: 42 x 42 x 16 at embedding_dim 32; one channel block, four tile groups)
      if (bt == 16) return launch<16, 4, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
      if (bt == 8) return launch<16, 2, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
      return launch<16, 1, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
    }
  }
  if (C == 64) {
    if (bt == 14) return launch<64, 14, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
    if (bt == 8) return launch<64, 8, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
    return launch<64, 4, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
  }
  if (bt == 14) return launch<32, 7, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
  if (bt == 8) return launch<32, 4, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
  return launch<32, 2, NW, LNIN, MOM>(p, g.blocks, g.lds, s);
}

// the stems and the strided convolutions of the projection blocks: stride 2, (4 ->) 16 -> 16 / 32 and 32 -> 64 channels
template <int C, int CIN>
int conv_stride2(const mz::ReprConvParams& p, const Geometry& g, hipStream_t s) {
  constexpr int D = 64 / C;  // tiles per wave = tiles per block / tile groups
  if (g.tiles_per_block >= 14) return launch<C, (C == 16 ? 16 : 14) / D, 1, false, false, CIN, 2>(p, g.blocks, g.lds, s);
  if (g.tiles_per_block == 8) return launch<C, 8 / D, 1, false, false, CIN, 2>(p, g.blocks, g.lds, s);
  return launch<C, 4 / D, 1, false, false, CIN, 2>(p, g.blocks, g.lds, s);
}

}  // namespace mzr
