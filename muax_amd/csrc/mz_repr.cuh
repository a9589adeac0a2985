// mz_repr.cuh -- hk.Conv2D(C, kernel_shape=3, padding='SAME', with_bias=False) on NHWC maps for the reference's
// REPRESENTATION nets (muax/nn.py:118-178 ResidualConvBlockV1 / V2 inside ResNetRepresentation :291-310 and
// EZStateEncoder :180-207): all 26 convolutions of config 4's root inference (muax/model.py:251-263) --
//   * stride 1, C -> C channels with C = 32 or 64 (the 24 layers inside the residual blocks: 42 x 42 x 32, 21 x 21 x 64,
//     11 x 11 x 64; 6 x 6 x 64 in the EZ encoder), which the library's implicit GEMM ran at 31 .. 75 TFLOP/s;
//   * stride 2 with the channels changing (the stems: 4 raw frames -> 32, padded to 16 input channels, and 32 -> 64),
//     haiku's SAME geometry, observations / 255 in front and relu behind fused;
//   * variants for a whole residual block in three launches (mz_norm.hip, mzs_resblock_v1): two convolutions of one
//     staged input (NW = 2), LayerNorm + relu applied to the input on its way into LDS (LNIN), fp64 moments of the outputs
//     left for the LayerNorm that follows (MOM).
//
// Implicit GEMM on v_mfma_f32_16x16x4_f32, the recurrent kernel's tile (mz_conv.cuh):
//     out[pixel][co] = sum_{tap, ci} in[pixel + tap][ci] W[tap][ci][co],   M = pixels in tiles of 16, N = C, K = 9 C.
// A workgroup owns a run of 16 TPW NTG consecutive pixels (row-major) of ONE image and stages the rows that run
// touches, plus a zero halo, in LDS (pixel stride C + 4 words: 16-byte aligned rows that spread over the banks).  Wave
// w owns output channels 16 (w % NCB) .. + 15 (NCB = C / 16 channel blocks) and the TPW pixel tiles of its tile group
// w / NCB; K is walked in 9 C / 16 groups of 16 input channels: per group a lane reads ONE ds_read_b128 of activations
// per tile and ONE global_load_dwordx4 of weights from the host-packed array Wp[tap][c][g][co][i] = W[tap][16 c + 4 g + i][co]
// (the recurrent kernel's layout), fetched two groups ahead (activations one): an explicit software pipeline, see the
// loop.  fp32 throughout, one accumulator chain per output.
//
// Floating-point kernel: checked against the torch module it replaces (MIOpen) and an fp64 evaluation, tolerance in
// tests/test_gpu_cfg4.py.
#pragma once
#include "mz_norm.cuh"
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

struct ReprConvParams {
  const float* x;      // [B][H][W][C]
  const float* wp;     // packed weights, 9 * C * C floats
  float* y;            // [B][H][W][C]
  int B, H, W, relu;
  // --- a residual block in three launches (mzs_resblock_v1, mz_repr.hip) ---
  const float* wp2;    // NW = 2: a second convolution of the SAME input (the projection beside conv_0) ...
  float* y2;           // ... and its output
  // MOM: (sum, sum of squares) of this workgroup's outputs in fp64, the layout mz_norm.cuh's apply kernel adds up:
  double* mom;         // [NW][B][K = gridDim.x][2]
  // LNIN: the input is a raw convolution output; its hk.LayerNorm + relu happen on the way into LDS
  const double* in_mom;  // [B][K][2] of the input tensor (K = gridDim.x: same geometry)
  const float* in_scale; // [C]
  const float* in_offset;
  float eps;
  int b0;              // first image of this launch (batches above 65535 images go in slices of grid y)
  // --- the stems (stride 2, channels changing): H, W are the INPUT's; the input tensor has cin_real channels per pixel
  // (4 raw frames: padded to the kernel's 16 with zeros in LDS and in the packed weights); in_div != 0: x / in_div on
  // the way in (the reference's observations / 255)
  int cin_real;
  float in_div;
  // --- a pre-activation block (mzs_resblock_v2): the input's moments may come from mz_norm.cuh's moments kernel, whose
  // chunk count differs from this launch's grid; the block's shortcut is added to the outputs (before their moments)
  int in_K;               // (sum, sum of squares) pairs per image in in_mom; 0: gridDim.x (a convolution's own MOM output)
  const float* residual;  // [B][H][W][C] added to stream 0's outputs, or null
};

typedef float rc_f32x4 __attribute__((ext_vector_type(4)));

// C = output channels; CIN = input channels as the kernel sees them (a multiple of 16), STRIDE = 1 or 2 with haiku's
// 'SAME' geometry (output ceil(H / STRIDE); total padding max((out - 1) STRIDE + 3 - H, 0), the smaller half first)
template <int C, int TPW, int NW = 1, bool LNIN = false, bool MOM = false, int CIN = C, int STRIDE = 1>
__global__ __launch_bounds__(256) void mz_repr_conv3x3_kernel(const ReprConvParams p) {
  constexpr int NCB = C / 16, NTG = 4 / NCB, NC = CIN / 16, PS = CIN + 4, G = 9 * NC;
  constexpr int BLOCK_PX = 16 * TPW * NTG;
  static_assert(STRIDE == 1 || (NW == 1 && !LNIN), "the strided variant is the plain convolution");
  extern __shared__ __attribute__((aligned(16))) float rc_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, H = p.H, W = p.W;
  const int Ho = (H + STRIDE - 1) / STRIDE, Wo = (W + STRIDE - 1) / STRIDE, npix = Ho * Wo;  // (npix: OUTPUT pixels)
  const int pt = max((Ho - 1) * STRIDE + 3 - H, 0) / 2, pl = max((Wo - 1) * STRIDE + 3 - W, 0) / 2;
  const int W2 = (Wo - 1) * STRIDE + 3;  // staged columns: -pl .. -pl + W2 - 1
  const int p0 = blockIdx.x * BLOCK_PX, p1 = min(p0 + BLOCK_PX, npix);
  // staged input rows, halo rows included (may be -1 / H)
  const int row0 = STRIDE * (p0 / Wo) - pt, row1 = STRIDE * ((p1 - 1) / Wo) - pt + 2;
  const int nrows = row1 - row0 + 1;
  // ---- stage rows [row0, row1] x columns [-1, W] (zero outside the image): a thread keeps its channel quad and walks
  // the pixels 256 / (C / 4) apart, eight loads in flight before the first LDS write (one load per trip is one L2 / HBM
  // round trip per trip: 20 trips, a third of the 21 x 21 layers' time)
  {
    constexpr int QPP = CIN / 4, PSTEP = 256 / QPP, U = 8;
    const int cin = (CIN == C && STRIDE == 1) ? CIN : p.cin_real;  // channels per pixel of the tensor in memory
    const float* img = p.x + (size_t)b * H * W * cin;
    const int c4 = tid % QPP, npx = nrows * W2;
    const bool real_quad = 4 * c4 < cin;
    float in_mean = 0.0f, in_rstd = 1.0f;
    rc_f32x4 in_g = (rc_f32x4){1.0f, 1.0f, 1.0f, 1.0f}, in_o = (rc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (LNIN) {
      const int in_k = p.in_K > 0 ? p.in_K : (int)gridDim.x;
      ln_stats(p.in_mom + (size_t)(p.b0 + b) * in_k * 2, in_k, H * W * CIN, p.eps, in_mean, in_rstd);
      in_g = *reinterpret_cast<const rc_f32x4*>(p.in_scale + 4 * c4);
      in_o = *reinterpret_cast<const rc_f32x4*>(p.in_offset + 4 * c4);
    }
    int px = tid / QPP;
    int ry = px / W2, cx = px - ry * W2;
    while (px < npx) {
      rc_f32x4 v[U];
      int dst[U];
      bool inside_mask[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int y = row0 + ry, x = cx - pl;
        inside_mask[u] = false;
        v[u] = (rc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        dst[u] = px < npx ? px : -1;
        bool inside = px < npx && y >= 0 && y < H && x >= 0 && x < W && real_quad;
        if (inside) v[u] = *reinterpret_cast<const rc_f32x4*>(img + ((size_t)y * W + x) * cin + 4 * c4);
        if constexpr (LNIN) inside_mask[u] = inside;
        px += PSTEP;
        cx += PSTEP;
        while (cx >= W2) {
          cx -= W2;
          ++ry;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (LNIN) {
          if (inside_mask[u]) {  // (the zero padding of the convolution is applied to the ACTIVATED map: halo stays 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[u][i] = fmaxf((v[u][i] - in_mean) * in_rstd * in_g[i] + in_o[i], 0.0f);
          }
        }
        if constexpr (STRIDE != 1 || CIN != C) {
          if (p.in_div != 0.0f && real_quad) {
            // x / d for a small positive integer d (255): q0 = RN(x y), y = RN(1 / d); r = x - q0 d (exact, fma);
            // RN(q0 + r y) is the correctly rounded quotient (Markstein; tests/test_oracle_kat.py checks it against the
            // division for every mantissa and d <= 300) -- three operations instead of the IEEE expansion's eleven
            const float d = p.in_div, y = 1.0f / d;
            if (d >= 1.0f && d <= 300.0f && d == floorf(d)) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float q0 = v[u][i] * y;
                v[u][i] = __builtin_fmaf(__builtin_fmaf(-q0, d, v[u][i]), y, q0);
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[u][i] = v[u][i] / d;
            }
          }
        }
        if (dst[u] >= 0) *reinterpret_cast<rc_f32x4*>(rc_lds + (size_t)dst[u] * PS + 4 * c4) = v[u];
      }
    }
  }
  __syncthreads();
  const int cb = wave % NCB, tg = wave / NCB;
  const int g = lane >> 4, m = lane & 15;
  const int ch = 16 * cb + m;
  // A operand: lane (m, g) reads pixel tile_base + m, input channels 16 c + 4 g + {0..3}; LDS word of tap (0, 0)
  int abase[TPW];
#pragma unroll
  for (int mt = 0; mt < TPW; ++mt) {
    const int px = p0 + 16 * (tg * TPW + mt) + m;
    const int pc = px < p1 ? px : p0;  // (rows past the run shadow its first pixel: computed, never stored)
    const int py = pc / Wo, pxx = pc - py * Wo;
    abase[mt] = ((STRIDE * py - pt - row0) * W2 + STRIDE * pxx) * PS + 4 * g;
  }
  rc_f32x4 acc[NW][TPW];
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int mt = 0; mt < TPW; ++mt) acc[s][mt] = (rc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  // Software pipeline, stated explicitly (left to itself the scheduler sinks the weight load to its first use -- one L2
  // round trip per group, 20 of the 51 us of the 21 x 21 layers -- and reuses accumulator registers as load targets,
  // which serialises ds_read -> 4 dependent MFMAs): group grp's MFMAs run on registers filled during group grp - 1
  // (activations: TPW ds_read_b128) and grp - 2 (weights: one global_load_dwordx4 per stream), one LDS read issued per
  // 4 NW MFMAs.
  const rc_f32x4* wq[NW];
  wq[0] = reinterpret_cast<const rc_f32x4*>(p.wp) + g * C + ch;  // quad [g][co] of a packed group
  if constexpr (NW == 2) wq[1] = reinterpret_cast<const rc_f32x4*>(p.wp2) + g * C + ch;
  rc_f32x4 wcur[NW], wn1[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    wcur[s] = wq[s][0];
    wn1[s] = wq[s][(size_t)4 * C];
  }
  rc_f32x4 acur[TPW];
#pragma unroll
  for (int mt = 0; mt < TPW; ++mt) acur[mt] = *reinterpret_cast<const rc_f32x4*>(rc_lds + abase[mt]);
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int toff = ((tap / 3) * W2 + (tap % 3)) * PS;
    const int tn = tap < 8 ? tap + 1 : 8;
    const int toff_next = ((tn / 3) * W2 + (tn % 3)) * PS;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int grp = tap * NC + c;
      rc_f32x4 wn2[NW];
#pragma unroll
      for (int s = 0; s < NW; ++s) wn2[s] = wq[s][(size_t)(grp + 2 < G ? grp + 2 : G - 1) * 4 * C];
      const int off_next = c + 1 < NC ? toff + 16 * (c + 1) : toff_next;
      rc_f32x4 anext[TPW];
#pragma unroll
      for (int mt = 0; mt < TPW; ++mt) anext[mt] = *reinterpret_cast<const rc_f32x4*>(rc_lds + abase[mt] + off_next);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
          for (int s = 0; s < NW; ++s)
            acc[s][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[mt][i], wcur[s][i], acc[s][mt], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, NW, 0);  // the weight loads first
#pragma unroll
      for (int mt = 0; mt < TPW; ++mt) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one ds_read
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NW, 0);  // its share of the MFMAs
      }
#pragma unroll
      for (int mt = 0; mt < TPW; ++mt) acur[mt] = anext[mt];
#pragma unroll
      for (int s = 0; s < NW; ++s) {
        wcur[s] = wn1[s];
        wn1[s] = wn2[s];
      }
    }
  }
  if constexpr (NW == 1 && STRIDE == 1 && CIN == C) {
    if (p.residual != nullptr) {  // ResidualConvBlockV2's identity shortcut (muax/nn.py:166-178): y = x + conv_1(.)
      const float* res = p.residual + (size_t)b * npix * C;
#pragma unroll
      for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int px = p0 + 16 * (tg * TPW + mt) + 4 * g + v;
          if (px < p1) acc[0][mt][v] = res[(size_t)px * C + ch] + acc[0][mt][v];
        }
    }
  }
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    float* out = (s == 0 ? p.y : p.y2) + (size_t)b * npix * C;
#pragma unroll
    for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = p0 + 16 * (tg * TPW + mt) + 4 * g + v;
        if (px < p1) {
          const float o = acc[s][mt][v];
          out[(size_t)px * C + ch] = p.relu ? fmaxf(o, 0.0f) : o;
        }
      }
  }
  if constexpr (MOM) {
    // (sum, sum of squares) of the RAW outputs of this workgroup, fp64 like mz_norm.cuh's moments kernel: the LayerNorm
    // that follows needs no pass of its own over the tensor
    double sq[NW][2];
#pragma unroll
    for (int s = 0; s < NW; ++s) {
      double su = 0.0, qu = 0.0;
#pragma unroll
      for (int mt = 0; mt < TPW; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int px = p0 + 16 * (tg * TPW + mt) + 4 * g + v;
          const double x = px < p1 ? (double)acc[s][mt][v] : 0.0;
          su += x;
          qu += x * x;
        }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        su += __shfl_down(su, d);
        qu += __shfl_down(qu, d);
      }
      sq[s][0] = su;
      sq[s][1] = qu;
    }
    __syncthreads();  // every wave is past its reads of the staged rows
    double* red = reinterpret_cast<double*>(rc_lds);
    if (lane == 0)
#pragma unroll
      for (int s = 0; s < NW; ++s) {
        red[(2 * s) * 4 + wave] = sq[s][0];
        red[(2 * s + 1) * 4 + wave] = sq[s][1];
      }
    __syncthreads();
    if (tid < 2 * NW) {
      const double t = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
      p.mom[(((size_t)(tid >> 1) * p.B + p.b0 + b) * gridDim.x + blockIdx.x) * 2 + (tid & 1)] = t;
    }
  }
}

// rows of LDS a block of BLOCK_PX consecutive OUTPUT pixels of an output of width Wo needs (halo included)
inline int repr_conv_rows(int block_px, int Wo, int stride = 1) { return stride * ((block_px + Wo - 1) / Wo) + 3; }

}  // namespace mz
