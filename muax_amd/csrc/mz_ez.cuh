// mz_ez.cuh -- recurrent_fn of the reference's EfficientZero-style nets as ONE kernel (muax/model.py:265-282 on
// EZDynamic muax/nn.py:267-309 and EZPrediction muax/nn.py:221-264, pre-activation blocks ResidualConvBlockV2
// muax/nn.py:151-178, use_v2 = True):
//   dynamics:    t = relu(LN(s));  out = conv3x3([t, a]) + s;  ns = out + conv3x3(relu(LN(conv3x3(relu(LN(out))))))
//   reward head: relu(LN(ns)) - conv1x1(16) - LN - relu - flatten - Linear(32, no bias) - LN - relu - Linear(F)
//   prediction:  o = ns + conv3x3(relu(LN(conv3x3(relu(LN(ns))))));  value head and policy head like the reward head
//   reward, value = support_to_scalar(softmax(logits))                                   (muax/utils.py:94-102)
// (the action enters as ONE extra plane holding the raw action index, muax/nn.py:291-296 -- no division by
// num_actions here, unlike ResNetDynamic).  Before this kernel an EZ search ran ~40 framework launches per simulation
// between the tree kernels.
//
// One 256-thread workgroup owns one root's 6x6xC map, which never leaves the CU: two zero-haloed 8x8-pixel buffers in
// LDS, pixel stride C + 20 words (C + 16 input channels: the action plane's group of 16 is zero but for its first
// channel, so the 33-channel convolution is the same code with one more K group).  A 3x3 convolution is an implicit
// GEMM on v_mfma_f32_16x16x4_f32: wave w owns output channels 16 (w % CT) .. + 15 (CT = C / 16) of the pixel tiles of
// its group; per group of 16 input channels a lane reads one ds_read_b128 of activations per tile and holds one
// 16-byte quad of host-packed weights Wp[tap][group][g][co][i] = W[tap][16 group + 4 g + i][co] (the layout of
// mz_conv.cuh) -- ALL quads of a convolution are requested before the LayerNorm that precedes it, so their L2 latency
// is hidden.  The residual stream stays in registers (MFMA C layout); LayerNorm is two workgroup reductions.  The heads
// are small: conv1x1 on the same tiles, flatten -> Linear as dot products with the weights streamed from L2.
// fp32 throughout.  Floating-point kernel: checked against the torch modules of muax_amd/nn.py (tests/test_gpu_cfg4.py).
#pragma once
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

struct EzHead {
  const float* ln_in;   // [2][C]   scale, offset
  const float* c1;      // [C][16]  conv1x1 (HWIO)
  const float* ln_mid;  // [2][16]
  const float* fc;      // [576][32] Linear on the NHWC-flattened 6x6x16 map, no bias
  const float* ln_vec;  // [2][32]
  const float* out_w;   // [32][n]
  const float* out_b;   // [n]
  int n;
};

struct EzParams {
  const float* x;         // [B][36][C]
  const int32_t* action;  // [B]
  float* y;               // [B][36][C] next state
  float* reward;          // [B]
  float* value;           // [B]
  float* prior_logits;    // [B][A]
  const float* d_ln_in;   // [2][C]
  const float* d_conv;    // packed, C + 16 input channels (C + 1 real)
  const float* d_ln0; const float* d_conv0; const float* d_ln1; const float* d_conv1;
  const float* p_ln0; const float* p_conv0; const float* p_ln1; const float* p_conv1;
  EzHead hr, hv, hp;
  int B, A, F, support;
};

constexpr int kEzHW = 6, kEzPix = 36, kEzHalo = 8, kEzTail = 2 * kEzHalo + 2 + 1;
typedef float ez4 __attribute__((ext_vector_type(4)));

template <int C>
struct EzGeom {
  static_assert(C == 32 || C == 64, "channel tiles of 16 over four wavefronts");
  static constexpr int CT = C / 16;        // channel tiles
  static constexpr int PG = 4 / CT;        // pixel-tile groups of waves
  static constexpr int TPW = (3 + PG - 1) / PG;  // pixel tiles per wave (a tile index >= 3 is skipped)
  static constexpr int STRIDE = C + 20;    // words per pixel: C + 16 input channels + 4 (bank spread, 16-byte aligned)
  static constexpr int BUF = (kEzHalo * kEzHalo + kEzTail) * STRIDE;
  static constexpr int KG = C / 16, KG1 = KG + 1;
  static constexpr int SCRATCH = 16 + 576 + 8 * 32 + 64 + 64;  // reductions | flat 6x6x16 | fc partials | vector | logits
  static constexpr int LDS_WORDS = 2 * BUF + SCRATCH;
};

MZ_DEV int ez_map_word(int p, int stride) { return ((p / kEzHW + 1) * kEzHalo + p % kEzHW + 1) * stride; }

// sum over the 64 lanes of a wavefront, in every lane: DPP butterfly inside the 16-lane rows, the four row sums
// through v_readlane (a shuffle through LDS per step costs ~10x as much, and a LayerNorm makes twelve)
MZ_DEV float ez_wave_sum(float x) {
  x = row_sum(x);
  const int xi = __float_as_int(x);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(xi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(xi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(xi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
  return (r0 + r1) + (r2 + r3);
}

template <int NV>
MZ_DEV void ez_wg_sum(float (&v)[NV], float* red, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = ez_wave_sum(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) red[NV * wave + i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i]);
}

// all weight quads of one convolution with G groups of 16 input channels, for this lane's (g, output channel)
template <int C, int G>
MZ_DEV void ez_load_weights(const float* __restrict__ Wp, int g, int co, ez4 (&wq)[9 * G]) {
  const ez4* q = reinterpret_cast<const ez4*>(Wp);
#pragma unroll
  for (int t = 0; t < 9 * G; ++t) wq[t] = q[((size_t)t * 4 + g) * C + co];
#ifdef MZ_EZ_PIN_LOADS
  __builtin_amdgcn_sched_barrier(0);  // keep the requests HERE: the scheduler otherwise sinks them next to their use
#endif
}

// acc[t] = conv3x3 of the haloed map `in` for this wave's pixel tiles t = 0 .. NT - 1, its 16 output channels.
// NT is a COMPILE-TIME count: a run-time "is this tile on" test around every group of four MFMAs turned the unrolled
// loop into ~300 basic blocks with the accumulator hazards padded out at each (45 us per launch; 25 of them here).
template <int C, int G, int TPW, int NT>
MZ_DEV void ez_conv_nt(const float* in, const ez4 (&wq)[9 * G], const int (&abase)[TPW], int g, ez4 (&acc)[TPW]) {
  constexpr int STRIDE = EzGeom<C>::STRIDE;
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = (ez4){0.0f, 0.0f, 0.0f, 0.0f};
#ifdef MZ_EZ_NO_CONV
  return;
#endif
  StaticFor<0, 9 * G>::run([&](auto ic) {
    constexpr int idx = decltype(ic)::value, tap = idx / G, c = idx % G;
    constexpr int off = ((tap / 3) * kEzHalo + tap % 3) * STRIDE + 16 * c;
    ez4 a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const ez4*>(in + abase[t] + off + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][i], wq[idx][i], acc[t], 0, 0, 0);
  });
}
// the waves of pixel group 0 hold TPW tiles, those of the last group what is left of the three (one wave-uniform branch
// per convolution)
template <int C, int G, int TPW>
MZ_DEV void ez_conv(const float* in, const ez4 (&wq)[9 * G], const int (&abase)[TPW], const bool (&on)[TPW], int g,
                    ez4 (&acc)[TPW]) {
  constexpr int LAST = 3 - (EzGeom<C>::PG - 1) * TPW;  // tiles of the last pixel group
  if constexpr (LAST == TPW) {
    ez_conv_nt<C, G, TPW, TPW>(in, wq, abase, g, acc);
  } else {
    if (on[TPW - 1]) ez_conv_nt<C, G, TPW, TPW>(in, wq, abase, g, acc);
    else ez_conv_nt<C, G, TPW, LAST>(in, wq, abase, g, acc);
  }
}

// hk.LayerNorm over the whole map (biased variance, eps 1e-5) of values held in the MFMA C layout:
// v[t][k] <-> pixel 16 (t0 + t) + 4 g + k, channel ch.  Returns relu(LN(v)) if RELU.
template <int TPW, bool RELU>
MZ_DEV void ez_layer_norm(const ez4 (&v)[TPW], const bool (&ok)[TPW][4], const float* __restrict__ so, int nch, int ch,
                          float inv_n, int lane, int wave, float* red, ez4 (&out)[TPW]) {
  const float sc = so[ch], of = so[nch + ch];  // (requested before the reductions: their L2 latency hides behind them)
  float s[1] = {0.0f};
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[0] = s[0] + (ok[t][k] ? v[t][k] : 0.0f);
  ez_wg_sum<1>(s, red, wave, lane);
  const float mean = s[0] * inv_n;
  float q[1] = {0.0f};
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = v[t][k] - mean;
      q[0] = q[0] + (ok[t][k] ? d * d : 0.0f);
    }
  ez_wg_sum<1>(q, red, wave, lane);
  const float rstd = 1.0f / __builtin_sqrtf(q[0] * inv_n + 1e-5f);
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float o = (v[t][k] - mean) * rstd * sc + of;
      out[t][k] = RELU ? fmaxf(o, 0.0f) : o;
    }
}

template <int C, int TPW>
MZ_DEV void ez_store_map(const ez4 (&v)[TPW], const bool (&ok)[TPW][4], const int (&pix)[TPW][4], float* buf, int ch) {
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ok[t][k]) buf[ez_map_word(pix[t][k], EzGeom<C>::STRIDE) + ch] = v[t][k];
}

// support_to_scalar(softmax(logits[0..F))) by the first wave (F <= 64)
MZ_DEV float ez_decode_support(const float* logits, int F, int support, int lane) {
  const float x = lane < F ? logits[lane] : -INFINITY;
  float m = row_max<4>(x);  // DPP inside the rows, v_readlane across them (no shuffles through LDS)
  {
    const int mi = __float_as_int(m);
    m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 0)), __int_as_float(__builtin_amdgcn_readlane(mi, 16))),
              fmaxf(__int_as_float(__builtin_amdgcn_readlane(mi, 32)), __int_as_float(__builtin_amdgcn_readlane(mi, 48))));
  }
  const float e = lane < F ? exp_neg(x - m) : 0.0f;
  const float s = ez_wave_sum(e), t = ez_wave_sum(e * (float)(lane - support));
  return inv_scaling(t / s);
}

// One head on the map `res` (C layout): logits[0..n) in `lgt` (LDS) when it returns (after a barrier).
template <int C>
MZ_DEV void ez_head(const EzHead& H, const ez4 (&res)[EzGeom<C>::TPW], const bool (&ok)[EzGeom<C>::TPW][4],
                    const int (&pix)[EzGeom<C>::TPW][4], float* buf, float* scratch, int ch, int tid, int lane, int wave) {
  using G = EzGeom<C>;
  constexpr int TPW = G::TPW, STRIDE = G::STRIDE;
  float* red = scratch;
  float* flat = scratch + 16;
  float* part = flat + 576;
  float* vec = part + 8 * 32;
  float* lgt = vec + 64;
#ifdef MZ_EZ_NO_HEADS
  return;
#endif
  const int g = lane >> 4, n16 = lane & 15;
  // the flatten -> Linear weights of this thread (72 inputs x its output), the output layer's column and the vector
  // LayerNorm's parameters are requested NOW: they are bound by L2 latency, and three reductions and a convolution
  // pass before their first use
  const int fo = tid & 31, fsl = tid >> 5;
  float fw[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) fw[i] = H.fc[(72 * fsl + i) * 32 + fo];
  float ow[32];
  const int on_ = tid < H.n ? tid : 0;
#pragma unroll
  for (int k = 0; k < 32; ++k) ow[k] = H.out_w[k * H.n + on_];
  const float ob = H.out_b[on_];
  const float vs = H.ln_vec[lane & 31], vo = H.ln_vec[32 + (lane & 31)];
  float c1w[C / 4];
#pragma unroll
  for (int c = 0; c < C / 16; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) c1w[4 * c + i] = H.c1[(16 * c + 4 * g + i) * 16 + n16];
  ez4 h[TPW];
  ez_layer_norm<TPW, true>(res, ok, H.ln_in, C, ch, 1.0f / (kEzPix * C), lane, wave, red, h);
  ez_store_map<C, TPW>(h, ok, pix, buf, ch);
  __syncthreads();
  // conv1x1 C -> 16: wave w < 3 takes pixel tile w, all 16 output channels
  ez4 c16 = (ez4){0.0f, 0.0f, 0.0f, 0.0f};
  bool ok16[1][4];
  int pix16[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pix16[k] = 16 * wave + 4 * g + k;
    ok16[0][k] = wave < 3 && pix16[k] < kEzPix;
  }
  if (wave < 3) {
    const int m = 16 * wave + n16;  // this lane's A row
    const int rowbase = m < kEzPix ? ez_map_word(m, STRIDE) : kEzHalo * kEzHalo * STRIDE;  // (zero tail)
#pragma unroll
    for (int c = 0; c < C / 16; ++c) {
      const ez4 a = *reinterpret_cast<const ez4*>(buf + rowbase + 16 * c + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        c16 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], c1w[4 * c + i], c16, 0, 0, 0);
    }
  }
  ez4 v16[1] = {c16}, h16[1];
  ez_layer_norm<1, true>(v16, ok16, H.ln_mid, 16, n16, 1.0f / (kEzPix * 16), lane, wave, red, h16);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (ok16[0][k]) flat[pix16[k] * 16 + n16] = h16[0][k];
  __syncthreads();
  // Linear(576 -> 32, no bias): thread = (slice of 72 inputs, output)
  {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 72; ++i) acc = __builtin_fmaf(flat[72 * fsl + i], fw[i], acc);
    part[fsl * 32 + fo] = acc;
  }
  __syncthreads();
  if (wave == 0) {
    float z = 0.0f;
    if (lane < 32) {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) z = z + part[sl * 32 + lane];
    }
    // LayerNorm over the 32-vector (lanes 0..31), relu
    const float s = ez_wave_sum(lane < 32 ? z : 0.0f);
    const float mean = s * (1.0f / 32.0f);
    const float d = lane < 32 ? z - mean : 0.0f;
    const float q = ez_wave_sum(d * d);
    const float rstd = 1.0f / __builtin_sqrtf(q * (1.0f / 32.0f) + 1e-5f);
    if (lane < 32) vec[lane] = fmaxf((z - mean) * rstd * vs + vo, 0.0f);
  }
  __syncthreads();
  if (tid < H.n) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(vec[k], ow[k], acc);
    lgt[tid] = acc + ob;
  }
  __syncthreads();
}

template <int C>
__global__ __launch_bounds__(256) void mz_ez_recurrent_kernel(const EzParams p) {
  using G = EzGeom<C>;
  constexpr int TPW = G::TPW, STRIDE = G::STRIDE, CT = G::CT;
  extern __shared__ __attribute__((aligned(16))) float ez_lds[];
  float* bufA = ez_lds;
  float* bufB = ez_lds + G::BUF;
  float* scratch = ez_lds + 2 * G::BUF;
  float* red = scratch;
  float* lgt = scratch + 16 + 576 + 8 * 32 + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n16 = lane & 15;
  const int r = blockIdx.x;
  const int ct = wave % CT, pg = wave / CT;
  const int ch = 16 * ct + n16;    // this lane's output channel
  const int co = ch;
  // pixel tiles of this wave; element k of tile t <-> pixel 16 tile + 4 g + k (MFMA C layout)
  bool on[TPW], ok[TPW][4];
  int abase[TPW], pix[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = pg * TPW + t;
    on[t] = tile < 3;
    const int m = 16 * tile + n16;  // this lane's A row (pixel)
    abase[t] = (on[t] && m < kEzPix) ? ((m / kEzHW) * kEzHalo + m % kEzHW) * STRIDE : kEzHalo * kEzHalo * STRIDE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pix[t][k] = 16 * tile + 4 * g + k;
      ok[t][k] = on[t] && pix[t][k] < kEzPix;
    }
  }
  for (int i = tid; i < 2 * G::BUF; i += 256) ez_lds[i] = 0.0f;  // halos, tails and the action group stay zero
  ez4 wq1[9 * G::KG1];
  ez_load_weights<C, G::KG1>(p.d_conv, g, co, wq1);
  // the root's map in the C layout (the residual stream)
  ez4 res[TPW];
  const float* xs = p.x + (size_t)r * kEzPix * C;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) res[t][k] = ok[t][k] ? xs[pix[t][k] * C + ch] : 0.0f;
  const float aplane = (float)p.action[r];
  const float inv_n = 1.0f / (kEzPix * C);
  __syncthreads();
  // ---- EZDynamic ----
  ez4 tmp[TPW], acc[TPW];
  ez_layer_norm<TPW, true>(res, ok, p.d_ln_in, C, ch, inv_n, lane, wave, red, tmp);
  ez_store_map<C, TPW>(tmp, ok, pix, bufA, ch);
  if (tid < kEzPix) bufA[ez_map_word(tid, STRIDE) + C] = aplane;  // the action plane: channel C of every map pixel
  __syncthreads();
  ez4 wq[9 * G::KG];
  ez_load_weights<C, G::KG>(p.d_conv0, g, co, wq);
  ez_conv<C, G::KG1, TPW>(bufA, wq1, abase, on, g, acc);
#pragma unroll
  for (int t = 0; t < TPW; ++t) res[t] = acc[t] + res[t];  // out = conv([t, a]) + s
  // block V2 on `res`
  ez_layer_norm<TPW, true>(res, ok, p.d_ln0, C, ch, inv_n, lane, wave, red, tmp);
  ez_store_map<C, TPW>(tmp, ok, pix, bufB, ch);
  __syncthreads();
  ez_conv<C, G::KG, TPW>(bufB, wq, abase, on, g, acc);
  ez_load_weights<C, G::KG>(p.d_conv1, g, co, wq);
  ez_layer_norm<TPW, true>(acc, ok, p.d_ln1, C, ch, inv_n, lane, wave, red, tmp);
  ez_store_map<C, TPW>(tmp, ok, pix, bufA, ch);  // (bufA's channel C still holds the action plane: K groups < KG1 ignore it)
  __syncthreads();
  ez_conv<C, G::KG, TPW>(bufA, wq, abase, on, g, acc);
  ez_load_weights<C, G::KG>(p.p_conv0, g, co, wq);
#pragma unroll
  for (int t = 0; t < TPW; ++t) res[t] = res[t] + acc[t];  // next state
  {
    float* ys = p.y + (size_t)r * kEzPix * C;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[t][k]) ys[pix[t][k] * C + ch] = res[t][k];
  }
  // ---- reward head on the next state ----
  ez_head<C>(p.hr, res, ok, pix, bufB, scratch, ch, tid, lane, wave);
  if (wave == 0) {
    const float rew = ez_decode_support(lgt, p.F, p.support, lane);
    if (lane == 0) p.reward[r] = rew;
  }
  // ---- EZPrediction on the next state: block V2, then the two heads ----
  ez_layer_norm<TPW, true>(res, ok, p.p_ln0, C, ch, inv_n, lane, wave, red, tmp);
  ez_store_map<C, TPW>(tmp, ok, pix, bufB, ch);
  __syncthreads();
  ez_conv<C, G::KG, TPW>(bufB, wq, abase, on, g, acc);
  ez_load_weights<C, G::KG>(p.p_conv1, g, co, wq);
  ez_layer_norm<TPW, true>(acc, ok, p.p_ln1, C, ch, inv_n, lane, wave, red, tmp);
  ez_store_map<C, TPW>(tmp, ok, pix, bufA, ch);
  __syncthreads();
  ez_conv<C, G::KG, TPW>(bufA, wq, abase, on, g, acc);
#pragma unroll
  for (int t = 0; t < TPW; ++t) res[t] = res[t] + acc[t];
  ez_head<C>(p.hv, res, ok, pix, bufB, scratch, ch, tid, lane, wave);
  if (wave == 0) {
    const float val = ez_decode_support(lgt, p.F, p.support, lane);
    if (lane == 0) p.value[r] = val;
  }
  ez_head<C>(p.hp, res, ok, pix, bufB, scratch, ch, tid, lane, wave);
  if (tid < p.A) p.prior_logits[(size_t)r * p.A + tid] = lgt[tid];
}

}  // namespace mz
