// mz_conv.cuh -- recurrent_fn of the reference's ResNet nets (muax/model.py:265-282 on muax/nn.py:313-378,
// residual block muax/nn.py:118-148) as ONE kernel: reward head r_func on [s, a / num_actions]; next-state
// tower = conv1x1 stem + `blocks` x ResidualConvBlockV1(64, projection) = 3 x (conv3x3 + LayerNorm) each +
// min_max_normalize2d; ResNetPrediction on the next state; both support decodes.  SURVEY.md 8(f) n3: the
// tower is 98 % of the flops of BASELINE config 4's recurrent_fn (24 3x3 convolutions on 6x6x64 maps).
//
// One workgroup owns one root's 6x6x64 map, which never leaves the CU: two zero-haloed 8x8-pixel buffers in
// LDS (pixel stride 68 words: 16-byte aligned rows that still spread over the banks).  A 3x3 convolution is
// an implicit GEMM  out[36 px (padded to 48)][64 co] = sum_{tap, ci} in[px + tap][ci] W[tap][ci][co]  on
// v_mfma_f32_16x16x4_f32: wave w owns output channels 16 w .. 16 w + 15 for the two full pixel tiles; the
// last four pixels run on v_mfma_f32_4x4x1 with K split across the waves (see conv3x3_tiles).  K is walked
// in 36 groups of 16 input channels; per group a lane issues ONE ds_read_b128 of activations per tile and
// ONE global_load_dwordx4 of weights (host-packed so that the four k-steps of a lane are contiguous), six
// groups ahead and across convolution boundaries; the projection and conv_0 of a block share one pass.  LayerNorm over the
// whole map (two-pass mean / variance, as jnp.var) is two workgroup reductions per convolution; the
// projection shortcut stays in registers until the block's final add.  The heads are small: 1x1 convolutions
// on the same MFMA tiles, flatten -> Linear layers as VALU dot products with the weights streamed from L2.
// fp32 throughout (the search's parity bar is 1e-5 on values): 65 MFLOP per root and simulation.
//
// Floating-point kernel: checked against the torch modules of muax_amd/nn.py (tests), tolerance there.
#pragma once
#include "mz_train.cuh"  // f32x4

#pragma clang fp contract(off)

namespace mz {

struct TowerParams {
  const float* x;          // [B][36][64]  NHWC hidden state s
  const int32_t* action;   // [B] (stem only)
  const float* stem_w;     // [65][64] 1x1 conv on [s, a / num_actions] (HWIO), or nullptr: no stem
  const float* conv_w;     // [blocks][3] x packed conv: Wp[tap 9][c 4][g 4][co 64][i 4] = W[tap][16 c + 4 g + i][co]
  const float* ln;         // [blocks][3][2][64]      (scale, offset) of proj_ln, ln_0, ln_1
  float* y;                // [B][36][64]
  float inv_num_actions;
  int B, blocks, normalize;
  // optional heads (all nullptr / 0: tower only).  ResNetDynamic.r_func on [s, a / num_actions] and
  // ResNetPrediction on the normalised next state (muax/nn.py:313-341,347-357), haiku layouts:
  const float* r_c1;   // [65][64]   conv1x1
  const float* r_c2;   // [64][64]   conv1x1
  const float* r_l1;   // [2304][64] Linear on the NHWC-flattened map
  const float* r_b1;   // [64]
  const float* r_l2;   // [64][F]
  const float* r_b2;   // [F]
  const float* v_c1;   // [64][16]
  const float* v_c2;   // [16][16]
  const float* v_l1;   // [576][16]
  const float* v_b1;   // [16]
  const float* v_l2;   // [16][F]
  const float* v_b2;   // [F]
  const float* p_c1;   // [64][16]
  const float* p_l1;   // [576][16]
  const float* p_b1;   // [16]
  const float* p_l2;   // [16][A]
  const float* p_b2;   // [A]
  float* reward;       // [B]   support_to_scalar(softmax(r_logits))
  float* value;        // [B]
  float* prior_logits; // [B][A]
  int heads, A, F, support;
};

constexpr int kTowerC = 64, kTowerHW = 6, kTowerPix = 36, kHalo = 8, kPixStride = 68;  // 16-byte aligned pixels
// one haloed map + an always-zero tail of 19 pixels: the rows that pad a 36-pixel map to three 16-row MFMA
// tiles read their 3x3 windows from the tail
constexpr int kTailPix = 2 * kHalo + 2 + 1;
constexpr int kBufWords = (kHalo * kHalo + kTailPix) * kPixStride;
constexpr int kHeadWords = 8 + 3 * 768 + 2 * 256 + 64 + 64 + 2 * 4 * 4 * 64;  // ... + remainder-row partial sums  // reduction slots + scratch of the heads

template <int NV>
MZ_DEV void wg_sum(float (&v)[NV], float* red, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v[i] = v[i] + __shfl_xor(v[i], m);
  __syncthreads();  // previous use of red[] is over
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) red[NV * wave + i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i]);
}

// acc[mt] = conv3x3 of the haloed map `in`, this wave's 16 output channels.
// K is walked in 36 groups of 16 input channels (9 taps x 4): lane (m, g) reads channels 16 c + 4 g + {0..3}
// of its pixel with ONE ds_read_b128 and uses element i in k-step i; the matching weights
// W[tap][16 c + 4 g + i][co] are one global_load_dwordx4 from the host-packed array
//   Wp[tap][c][g][co][i]            (any bijection of K is a valid order for the sum).
// Weight quads are fetched kConvAhead groups ahead, activation quads one group ahead.
typedef float f32x4u __attribute__((ext_vector_type(4)));
constexpr int kConvAhead = 6;
struct ConvPrefetch {
  f32x4u q[2][kConvAhead];  // weight quads of groups 0 .. kConvAhead-1 of the NEXT call's stream(s)
};
MZ_DEV void conv_prefetch(const float* __restrict__ Wp, int wlane, f32x4u (&q)[kConvAhead]) {
  const f32x4u* wq = reinterpret_cast<const f32x4u*>(Wp) + wlane;
#pragma unroll
  for (int i = 0; i < kConvAhead; ++i) q[i] = wq[i * 256];
}
// NW convolutions of the SAME input in one pass over K (the projection and conv_0 of a residual block share
// their activation reads).  `pf` holds the first weight quads of this call's stream(s), fetched while the
// previous LayerNorm ran; on return it holds those of the next call's (Wnext[0 .. nnext)), so the L2
// latency at the head of a convolution is never exposed.
// Rows 0..31 of the 36-pixel map are two 16x16x4 tiles per wave (its 16 output channels).  The last four
// rows would waste 3/4 of a third tile, so they run on v_mfma_f32_4x4x1_16b_f32 instead: its 16 blocks are
// the 16 groups of four output channels (lane = channel), its four rows the four pixels, k = 1 per
// instruction -- and the four waves SPLIT K: wave w takes the channels 16 c + 4 w + {0..3} of every group, the
// same packed weight quads [grp][g = w][co = lane], 144 quarter-cost MFMAs per wave.  The partial sums meet
// in LDS (`part`, NW x 4 waves x 64 lanes x 4 rows) and land in acc[.][2] of the lanes that own those pixels.
template <int NW>
MZ_DEV void conv3x3_tiles(const float* in, const float* const (&Wp)[NW], const float* const (&Wnext)[2], int nnext,
                          const int (&abase)[3], int wlane, int lane, int wave, float* part, ConvPrefetch& pf,
                          f32x4 (&acc)[NW][3]) {
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[s][mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 rem[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) rem[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  constexpr int G = 36, AHEAD = kConvAhead;
  f32x4u wbuf[NW][AHEAD + 1];
  f32x4u abuf[2][2];
  auto a_off = [](int grp) { return (((grp >> 2) / 3) * kHalo + ((grp >> 2) % 3)) * kPixStride + 16 * (grp & 3); };
  // remainder rows: pixel 32 + (lane & 3), input channels 16 c + 4 wave + {0..3}; weights [grp][g = wave][co = lane]
  const int rbase = ((32 + (lane & 3)) / kTowerHW * kHalo + (32 + (lane & 3)) % kTowerHW) * kPixStride + 4 * wave;
  const int rlane = wave * kTowerC + lane;
  f32x4u rw[NW][2], ra[2];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) wbuf[s][q] = pf.q[s][q];
    rw[s][0] = (reinterpret_cast<const f32x4u*>(Wp[s]) + rlane)[0];
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) abuf[0][mt] = *reinterpret_cast<const f32x4u*>(in + abase[mt] + a_off(0));
  ra[0] = *reinterpret_cast<const f32x4u*>(in + rbase + a_off(0));
  StaticFor<0, G>::run([&](auto gc) {
    constexpr int grp = decltype(gc)::value;
    if constexpr (grp + AHEAD < G) {
#pragma unroll
      for (int s = 0; s < NW; ++s)
        wbuf[s][(grp + AHEAD) % (AHEAD + 1)] = (reinterpret_cast<const f32x4u*>(Wp[s]) + wlane)[(grp + AHEAD) * 256];
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < nnext)
          pf.q[s][grp + AHEAD - G] = (reinterpret_cast<const f32x4u*>(Wnext[s]) + wlane)[(grp + AHEAD - G) * 256];
    }
    if constexpr (grp + 1 < G) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        abuf[(grp + 1) & 1][mt] = *reinterpret_cast<const f32x4u*>(in + abase[mt] + a_off(grp + 1));
      ra[(grp + 1) & 1] = *reinterpret_cast<const f32x4u*>(in + rbase + a_off(grp + 1));
#pragma unroll
      for (int s = 0; s < NW; ++s) rw[s][(grp + 1) & 1] = (reinterpret_cast<const f32x4u*>(Wp[s]) + rlane)[(grp + 1) * 256];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetches up here: the scheduler otherwise sinks them
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < NW; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[s][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[grp & 1][mt][i], wbuf[s][grp % (AHEAD + 1)][i],
                                                            acc[s][mt], 0, 0, 0);
        rem[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(ra[grp & 1][i], rw[s][grp & 1][i], rem[s], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  });
  // partial sums of the remainder rows: [s][wave][row v][channel = lane]
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int v = 0; v < 4; ++v) part[((s * 4 + wave) * 4 + v) * 64 + lane] = rem[s][v];
  __syncthreads();
  if (lane < 16) {
    const int ch = 16 * wave + lane;
#pragma unroll
    for (int s = 0; s < NW; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        acc[s][2][v] = (part[((s * 4 + 0) * 4 + v) * 64 + ch] + part[((s * 4 + 1) * 4 + v) * 64 + ch]) +
                       (part[((s * 4 + 2) * 4 + v) * 64 + ch] + part[((s * 4 + 3) * 4 + v) * 64 + ch]);
  }
}
template <int NW>
MZ_DEV void layer_norm_tiles(f32x4 (&acc)[NW][3], const float* const (&so)[NW], const bool (&relu)[NW], int ch, int lane,
                             int wave, float* red) {
  const int g = lane >> 4;
  float mean[NW], var[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    mean[s] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) mean[s] = mean[s] + ((16 * mt + 4 * g + v < kTowerPix) ? acc[s][mt][v] : 0.0f);
  }
  wg_sum<NW>(mean, red, wave, lane);
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    mean[s] = mean[s] * (1.0f / (kTowerPix * kTowerC));
    var[s] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float d = acc[s][mt][v] - mean[s];
        var[s] = var[s] + ((16 * mt + 4 * g + v < kTowerPix) ? d * d : 0.0f);
      }
  }
  wg_sum<NW>(var, red, wave, lane);
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    const float rstd = 1.0f / __builtin_sqrtf(var[s] * (1.0f / (kTowerPix * kTowerC)) + 1e-5f);
    const float sc = so[s][ch], of = so[s][kTowerC + ch];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float o = (acc[s][mt][v] - mean[s]) * rstd * sc + of;
        acc[s][mt][v] = relu[s] ? fmaxf(o, 0.0f) : o;
      }
  }
}

MZ_DEV void store_map(const f32x4 (&acc)[3], float* buf, int ch, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int p = 16 * mt + 4 * g + v;
      if (p < kTowerPix) buf[((p / kTowerHW + 1) * kHalo + p % kTowerHW + 1) * kPixStride + ch] = acc[mt][v];
    }
}


// ---- small pieces of the heads ----
// conv1x1 as MFMA tiles: acc[mt] (mt = 0..2) = in[px][0..16 KC) . W[k][ncol], `in` rows addressed through
// rowbase[mt] (word offsets of this lane's pixel row + 4 * (lane >> 4)), W row-major [K][ldw]
MZ_DEV void conv1x1_tiles(const float* in, const int (&rowbase)[3], const float* __restrict__ W, int ldw, int ncol,
                          int KC, int g, f32x4 (&acc)[3]) {
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) acc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  for (int c = 0; c < KC; ++c) {
    f32x4u a[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const f32x4u*>(in + rowbase[mt] + 16 * c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float b = W[(16 * c + 4 * g + i) * ldw + ncol];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i], b, acc[mt], 0, 0, 0);
    }
  }
}
// one M-tile, K = 16 (the value head's second 1x1 convolution)
MZ_DEV f32x4 conv1x1_k16(const float* row, const float* __restrict__ W, int n, int g) {
  f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const f32x4u a = *reinterpret_cast<const f32x4u*>(row);
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], W[(4 * g + i) * 16 + n], acc, 0, 0, 0);
  return acc;
}
// support_to_scalar(softmax(logits[0..F))) by the first wave (F <= 64), result in every lane of that wave
MZ_DEV float decode_support(const float* logits, int F, int support, int lane) {
  float x = lane < F ? logits[lane] : -INFINITY;
  float m = x;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, __shfl_xor(m, k));
  float e = lane < F ? exp_neg(x - m) : 0.0f;
  float s = e, t = e * (float)(lane - support);
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    s = s + __shfl_xor(s, k);
    t = t + __shfl_xor(t, k);
  }
  return inv_scaling(t / s);
}

__global__ __launch_bounds__(256) void mz_resnet_tower_kernel(const TowerParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;
  float* bufB = lds + kBufWords;
  float* red = lds + 2 * kBufWords;
  float* hv = red + 8;            // [48][16] value head map (rows >= 36 stay zero)
  float* hv2 = hv + 768;          // [48][16]
  float* hp = hv2 + 768;          // [48][16] policy head map
  float* part = hp + 768;         // [256] partial sums of the flatten -> Linear layers
  float* part2 = part + 256;      // [256]
  float* vec = part2 + 256;       // [64] hidden vectors
  float* lgt = vec + 64;          // [64] logits
  float* part3 = lgt + 64;        // [2][4 waves][4 rows][64] partial sums of the 4x4x1 remainder tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  for (int i = tid; i < 2 * kBufWords + kHeadWords; i += 256) lds[i] = 0.0f;
  __syncthreads();
  const float* xin = p.x + (size_t)r * kTowerPix * kTowerC;
  for (int i = tid; i < kTowerPix * kTowerC; i += 256) {
    const int px = i >> 6, c = i & 63;
    bufA[((px / kTowerHW + 1) * kHalo + px % kTowerHW + 1) * kPixStride + c] = xin[i];
  }
  __syncthreads();

  // A operand: lane (m = lane & 15, kk = lane >> 4) reads pixel 16 mt + m, input channel 4 c4 + kk
  int abase[3];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int px = 16 * mt + (lane & 15);
    abase[mt] = (px < kTowerPix ? ((px / kTowerHW) * kHalo + px % kTowerHW) * kPixStride
                                : kHalo * kHalo * kPixStride) + 4 * (lane >> 4);
  }
  const int ch = 16 * wave + (lane & 15);            // this lane's output channel
  const int wcol = (lane >> 4) * kTowerC + ch;       // B operand: quad [g = lane >> 4][co = ch] of a packed group
  f32x4 acc[3];

  const int g4 = lane >> 4, n16 = lane & 15;
  int rowc[3];  // centre-tap rows for 1x1 convolutions on a haloed map
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) rowc[mt] = abase[mt] + (kHalo + 1) * kPixStride;
  if (p.heads) {
    // ---- reward head on [s, a / num_actions] (muax/nn.py:347-357): two 1x1 convs, flatten, two Linears ----
    conv1x1_tiles(bufA, rowc, p.r_c1, kTowerC, ch, 4, g4, acc);
    const float pl = (float)p.action[r] * p.inv_num_actions * p.r_c1[kTowerC * kTowerC + ch];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(acc[mt][v] + pl, 0.0f);
    store_map(acc, bufB, ch, lane);
    __syncthreads();
    conv1x1_tiles(bufB, rowc, p.r_c2, kTowerC, ch, 4, g4, acc);
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(acc[mt][v], 0.0f);
    __syncthreads();  // every wave has read bufB
    store_map(acc, bufB, ch, lane);
    __syncthreads();
    {
      // Linear(2304 -> 64): wave = 9 pixels of the map, lane = output unit; weights stream from L2
      float sacc = 0.0f;
      for (int px = 9 * wave; px < 9 * wave + 9; ++px) {
        const float* row = bufB + ((px / kTowerHW + 1) * kHalo + px % kTowerHW + 1) * kPixStride;
        const float* wr = p.r_l1 + (size_t)px * kTowerC * kTowerC + lane;
#pragma unroll 8
        for (int c = 0; c < kTowerC; ++c) sacc = __builtin_fmaf(row[c], wr[c * kTowerC], sacc);
      }
      part[tid] = sacc;
    }
    __syncthreads();
    if (tid < 64) vec[tid] = fmaxf(((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) + p.r_b1[tid], 0.0f);
    __syncthreads();
    if (tid < p.F) {
      float a = 0.0f;
      for (int k = 0; k < 64; ++k) a = __builtin_fmaf(vec[k], p.r_l2[k * p.F + tid], a);
      lgt[tid] = a + p.r_b2[tid];
    }
    __syncthreads();
    if (wave == 0) {
      const float rw = decode_support(lgt, p.F, p.support, lane);
      if (lane == 0) p.reward[r] = rw;
    }
    __syncthreads();
  }

  float* cur = bufA;
  float* oth = bufB;
  if (p.stem_w != nullptr) {
    // conv1x1 on [s, a / num_actions] + relu: the action plane is constant over the map
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int ctr = (kHalo + 1) * kPixStride;  // centre tap
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4u a[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const f32x4u*>(cur + abase[mt] + ctr + 16 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float b = p.stem_w[(16 * c + 4 * (lane >> 4) + i) * kTowerC + ch];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i], b, acc[mt], 0, 0, 0);
      }
    }
    const float plane = (float)p.action[r] * p.inv_num_actions * p.stem_w[kTowerC * kTowerC + ch];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(acc[mt][v] + plane, 0.0f);
    store_map(acc, oth, ch, lane);
    __syncthreads();
    float* t = cur; cur = oth; oth = t;
  }

  ConvPrefetch pf;
  constexpr size_t CW = 9 * kTowerC * kTowerC;
  if (p.blocks > 0) {
    conv_prefetch(p.conv_w, wcol, pf.q[0]);
    conv_prefetch(p.conv_w + CW, wcol, pf.q[1]);
  }
  for (int blk = 0; blk < p.blocks; ++blk) {
    const float* W = p.conv_w + (size_t)blk * 3 * CW;
    const float* LN = p.ln + (size_t)blk * 3 * 2 * kTowerC;
    const bool last = blk + 1 == p.blocks;
    // projection and conv_0 read the same map: one pass over K, one pair of LayerNorm reductions
    f32x4 pr[2][3];
    {
      const float* const w2[2] = {W, W + CW};
      const float* const nx[2] = {W + 2 * CW, nullptr};
      conv3x3_tiles<2>(cur, w2, nx, 1, abase, wcol, lane, wave, part3, pf, pr);
      const float* const so[2] = {LN, LN + 2 * kTowerC};
      const bool rl[2] = {false, true};
      layer_norm_tiles<2>(pr, so, rl, ch, lane, wave, red);
    }
    store_map(pr[1], oth, ch, lane);
    __syncthreads();
    f32x4 out[1][3];
    {
      const float* const w1[1] = {W + 2 * CW};
      const float* const nx[2] = {W + 3 * CW, W + 4 * CW};
      conv3x3_tiles<1>(oth, w1, nx, last ? 0 : 2, abase, wcol, lane, wave, part3, pf, out);
      const float* const so[1] = {LN + 4 * kTowerC};
      const bool rl[1] = {false};
      layer_norm_tiles<1>(out, so, rl, ch, lane, wave, red);
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(pr[0][mt][v] + out[0][mt][v], 0.0f);
    store_map(acc, cur, ch, lane);  // every wave is past its reads of `cur` (the LayerNorm barriers)
    __syncthreads();
  }
  if (p.blocks == 0) {
    // (stem only) bring the map back into registers
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g + v;
        acc[mt][v] = px < kTowerPix ? cur[((px / kTowerHW + 1) * kHalo + px % kTowerHW + 1) * kPixStride + ch] : 0.0f;
      }
  }

  if (p.normalize) {
    // min_max_normalize2d (muax/nn.py:47-56): per channel over the 36 pixels
    const int g = lane >> 4;
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bool ok = 16 * mt + 4 * g + v < kTowerPix;
        mn = ok ? fminf(mn, acc[mt][v]) : mn;
        mx = ok ? fmaxf(mx, acc[mt][v]) : mx;
      }
    mn = fminf(mn, __shfl_xor(mn, 16)); mn = fminf(mn, __shfl_xor(mn, 32));
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    float scale = mx - mn;
    scale = scale < 1e-5f ? scale + 1e-5f : scale;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = (acc[mt][v] - mn) / scale;
  }
  float* yout = p.y + (size_t)r * kTowerPix * kTowerC;
  {
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g + v;
        if (px < kTowerPix) yout[px * kTowerC + ch] = acc[mt][v];
      }
  }
  if (p.heads) {
    // ---- prediction heads on the normalised next state (muax/nn.py:313-341) ----
    store_map(acc, cur, ch, lane);
    __syncthreads();
    if (wave < 2) {  // wave 0: value head, wave 1: policy head -- first 1x1 conv (64 -> 16) + relu
      f32x4 h[3];
      conv1x1_tiles(cur, rowc, wave == 0 ? p.v_c1 : p.p_c1, 16, n16, 4, g4, h);
      float* dst = wave == 0 ? hv : hp;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int px = 16 * mt + 4 * g4 + v;
          if (px < kTowerPix) dst[px * 16 + n16] = fmaxf(h[mt][v], 0.0f);
        }
    }
    __syncthreads();
    if (wave == 0) {  // value head: second 1x1 conv (16 -> 16) + relu
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const f32x4 h = conv1x1_k16(hv + (16 * mt + n16) * 16 + 4 * g4, p.v_c2, n16, g4);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int px = 16 * mt + 4 * g4 + v;
          if (px < kTowerPix) hv2[px * 16 + n16] = fmaxf(h[v], 0.0f);
        }
      }
    }
    __syncthreads();
    {
      // Linear(576 -> 16) of both heads: thread = (output unit n, one of 16 slices of 36 inputs)
      const int n = tid & 15, sl = tid >> 4;
      float sv = 0.0f, sp = 0.0f;
      for (int i = 36 * sl; i < 36 * sl + 36; ++i) {
        sv = __builtin_fmaf(hv2[i], p.v_l1[i * 16 + n], sv);
        sp = __builtin_fmaf(hp[i], p.p_l1[i * 16 + n], sp);
      }
      __syncthreads();
      part[tid] = sv;
      part2[tid] = sp;
    }
    __syncthreads();
    if (tid < 32) {
      const int n = tid & 15;
      const float* src = tid < 16 ? part : part2;
      float a = 0.0f;
      for (int sl = 0; sl < 16; ++sl) a = a + src[sl * 16 + n];
      vec[tid] = fmaxf(a + (tid < 16 ? p.v_b1[n] : p.p_b1[n]), 0.0f);
    }
    __syncthreads();
    if (tid < p.F) {
      float a = 0.0f;
      for (int k = 0; k < 16; ++k) a = __builtin_fmaf(vec[k], p.v_l2[k * p.F + tid], a);
      lgt[tid] = a + p.v_b2[tid];
    } else if (tid >= 64 && tid < 64 + p.A) {
      const int j = tid - 64;
      float a = 0.0f;
      for (int k = 0; k < 16; ++k) a = __builtin_fmaf(vec[16 + k], p.p_l2[k * p.A + j], a);
      p.prior_logits[(size_t)r * p.A + j] = a + p.p_b2[j];
    }
    __syncthreads();
    if (wave == 0) {
      const float vl = decode_support(lgt, p.F, p.support, lane);
      if (lane == 0) p.value[r] = vl;
    }
  }
}

}  // namespace mz
