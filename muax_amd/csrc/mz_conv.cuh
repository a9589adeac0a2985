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
// last four pixels run on v_mfma_f32_4x4x1 on the same weight registers (see conv3x3_tiles).  K is walked
// in 36 groups of 16 input channels; per group a lane issues ONE ds_read_b128 of activations per tile and
// ONE global_load_dwordx4 of weights (host-packed so that the four k-steps of a lane are contiguous), six
// groups ahead and across convolution boundaries; the projection and conv_0 of a block share one pass.  LayerNorm over the
// whole map (two-pass mean / variance, as jnp.var) is two workgroup reductions per convolution; the
// projection shortcut stays in registers until the block's final add.  The heads are small: 1x1 convolutions
// on the same MFMA tiles, flatten -> Linear layers as VALU dot products with the weights streamed from L2.
// fp32 throughout (the search's parity bar is 1e-5 on values): 65 MFLOP per root and simulation.
// With <= 128 roots a second kernel gives every root two workgroups ("pair mode", below).
//
// Floating-point kernel: checked against the torch modules of muax_amd/nn.py (tests), tolerance there.
#pragma once
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

// opt-in phase timers of the recurrent kernel (tools/profile_tower.py builds with -DMZ_PROFILE); no code otherwise
#ifdef MZ_PROFILE
__device__ unsigned long long g_tower_prof[1024 * 16];
#define MZ_TT(k)                                              \
  {                                                           \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    pt[k] += t_ - tlast;                                      \
    tlast = t_;                                               \
  }
#else
#define MZ_TT(k)
#endif
// -DMZ_PROF_HEADS (with -DMZ_PROFILE): slots 3..9 time the pieces of "heads after the tower" instead of the passes
#if defined(MZ_PROFILE) && defined(MZ_PROF_HEADS)
#define MZ_TP(k)
#define MZ_TH(k) MZ_TT(k)
#define MZ_TH_PARAMS , unsigned long long* pt, unsigned long long& tlast
#define MZ_TH_ARGS , pt, tlast
#else
#define MZ_TP(k) MZ_TT(k)
#define MZ_TH(k)
#define MZ_TH_PARAMS
#define MZ_TH_ARGS
#endif

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
  // pair mode (mz_resnet_tower_pair_kernel): exchange slots and flags of the two workgroups of a root
  float* pair_f;       // [B][2 halves][2 parity][kPairSlot] (value, message number) pairs
  unsigned* pair_u;    // [B][4]: -, -, launch epoch, status
};

// where ONE root's pass reads and writes (the step-wise launches index the batch arrays of TowerParams by the root;
// the fused search, mz_search_conv.hip, points into the tree's own embedding rows)
struct TowerIO {
  const float* x;      // [36][64] hidden state
  float* y;            // [36][64] next state
  int action;
  float* reward;       // [1]
  float* value;        // [1]
  float* prior_logits; // [A]
  // the workgroup's LDS is as a previous pass of THIS kernel left it (the one-launch search): the halos, the zero tails
  // and the head maps' padding rows are still zero -- no pass writes them -- and every word a pass reads besides those it
  // writes first, so the 66 KB zero fill (2 of the 2.5 us of "LDS init + state load") is needed once per launch only
  bool lds_clean = false;
  // round 6 (the one-launch search): the weights of the prediction heads' first 1x1 convolutions in the workgroup's LDS
  // for the whole launch -- v_c1 [64][16] then p_c1 [64][16] -- when head_w_lds is set.  They are 8 KB that every
  // simulation needs at the very start of the heads, and by then the passes' 5.7 MB of convolution weights have
  // streamed through the L2 (and the TLBs) since their last use: the heads' first phase was 2.6 us of waiting for 0.64 us
  // of MFMAs
  const float* head_w = nullptr;
  bool head_w_lds = false;
};

constexpr int kTowerC = 64, kTowerHW = 6, kTowerPix = 36, kHalo = 8, kPixStride = 68;  // 16-byte aligned pixels
// one haloed map + an always-zero tail of 19 pixels: the rows that pad a 36-pixel map to three 16-row MFMA
// tiles read their 3x3 windows from the tail
constexpr int kTailPix = 2 * kHalo + 2 + 1;
constexpr int kBufWords = (kHalo * kHalo + kTailPix) * kPixStride;
constexpr int kRhWords = kTowerPix * kTowerC;  // pair mode: the reward head's second feature map, [36][64], kept through the tower
constexpr int kHeadWords = 32 + 3 * 768 + 2 * 256 + 64 + 64 + kRhWords;  // reduction slots (2 sets x 4 values x 4 waves) + scratch of the heads

// sum over the 64 lanes of a wavefront, in every lane: the DPP butterfly inside the 16-lane rows, then the four row
// sums through v_readlane (round 3; the six shuffles through LDS this replaces cost ~0.3 us per reduction, and a launch
// makes ~70 of them).  One fixed order for both launch shapes (they must agree bit for bit).
MZ_DEV float wave_sum64(float x) {
  x = row_sum(x);
  const int xi = __float_as_int(x);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(xi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(xi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(xi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
  return (r0 + r1) + (r2 + r3);
}
// value of the lane 16 / 32 lanes away (same column n) through v_permlane16_swap / v_permlane32_swap (gfx950; no LDS):
// with the same register as both operands the swap exchanges rows (0, 1), (2, 3) / halves of the wavefront
MZ_DEV float lane_xor16(float x) {
  auto s = __builtin_amdgcn_permlane16_swap(f2u(x), f2u(x), false, false);
  return u2f((threadIdx.x & 16) ? s[0] : s[1]);
}
MZ_DEV float lane_xor32(float x) {
  auto s = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
  return u2f((threadIdx.x & 32) ? s[0] : s[1]);
}
// sum over the workgroup, in every lane.  Two sets of slots used in turn: a wave that writes set k for reduction n + 2
// has passed the barrier of reduction n + 1, which every wave reaches only after its reads of reduction n -- so ONE
// barrier per reduction is enough (round 4; the second one, "previous use of red[] is over", cost 32 barriers a pass).
struct RedSlots {
  float* base;  // [2][16]
  int k;
};
template <int NV>
MZ_DEV void wg_sum(float (&v)[NV], RedSlots& R, int wave, int lane) {
  static_assert(NV <= 4, "16 words per set");
  float* red = R.base + 16 * R.k;
  R.k ^= 1;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum64(v[i]);
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) red[NV * wave + i] = v[i];
  // the barrier orders LDS only: __syncthreads() would also drain vmcnt -- the message stores of the previous layer and
  // the early loads of the partner's message (pair_peek) must stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i]);
}

// a workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): loads requested from global memory
// before it stay in flight across it -- a __syncthreads() drains vmcnt as well, i.e. WAITS for them
MZ_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// acc[mt] = conv3x3 of the haloed map `in`, this wave's 16 output channels.
// K is walked in 36 groups of 16 input channels (9 taps x 4): lane (m, g) reads channels 16 c + 4 g + {0..3}
// of its pixel with ONE ds_read_b128 and uses element i in k-step i; the matching weights
// W[tap][16 c + 4 g + i][co] are one global_load_dwordx4 from the host-packed array
//   Wp[tap][c][g][co][i]            (any bijection of K is a valid order for the sum).
// Weight quads are fetched kConvAhead groups ahead, activation quads one group ahead.
typedef float f32x4u __attribute__((ext_vector_type(4)));
#ifndef MZ_CONV_AHEAD
#define MZ_CONV_AHEAD 6
#endif
constexpr int kConvAhead = MZ_CONV_AHEAD;
#ifndef MZ_CONV_FOLD_TAPS
#define MZ_CONV_FOLD_TAPS 1
#endif
constexpr int kConvFoldTaps = MZ_CONV_FOLD_TAPS;  // taps per partial sum: 1 = every tap on its own, 3 = per kernel row, 9 = one chain
template <int AH>
struct ConvPrefetch {
  f32x4u q[2][AH];  // weight quads of groups 0 .. AH-1 of the NEXT call's stream(s)
};
template <int AH>
MZ_DEV void conv_prefetch(const float* __restrict__ Wp, int wlane, f32x4u (&q)[AH]) {
  const f32x4u* wq = reinterpret_cast<const f32x4u*>(Wp) + wlane;
#pragma unroll
  for (int i = 0; i < AH; ++i) q[i] = wq[i * 256];
}
// NW convolutions of the SAME input in one pass over K (the projection and conv_0 of a residual block share
// their activation reads).  `pf` holds the first weight quads of this call's stream(s), fetched while the
// previous LayerNorm ran; on return it holds those of the next call's (Wnext[0], Wnext[1]), so the L2
// latency at the head of a convolution is never exposed.
// Rows 0..31 of the 36-pixel map are two 16x16x4 tiles per wave (its 16 output channels).  The last four
// rows would waste 3/4 of a third tile, so they run on v_mfma_f32_4x4x1_16b_f32 instead, on the SAME B
// registers: lane l = (g = l >> 4, n = l & 15) holds W[4 g + i][16 wave + n] for the k-step i, and in the
// 4x4x1 layout lane l is (block l >> 2, column l & 3), i.e. block = (g, n >> 2): the 16 blocks are 4 groups
// of four of the wave's channels x the 4 k-quads of the group.  With A = in[pixel 32 + (l & 3)][16 c + 4 g + i]
// one instruction adds k = 4 g + i for every block; the four k-quads of a channel sit in lanes 16 apart and
// meet in two xor-shuffles at the end.  No extra weight load, no LDS, no barrier: 144 quarter-cost MFMAs.
// TSEL selects the pixel tiles a workgroup computes: 0 = the whole map (tiles 0, 1 and the remainder rows),
// 1 = tile 0 only (pixels 0..15), 2 = tile 1 + the remainder rows (pixels 16..35): the two halves of a
// root in pair mode.
template <int TSEL>
MZ_DEV constexpr bool tile_on(int mt) { return TSEL == 0 || (TSEL == 1 ? mt == 0 : mt >= 1); }
template <int NW, int TSEL = 0, int AHEAD = kConvAhead>
MZ_DEV void conv3x3_tiles(const float* in, const float* const (&Wp)[NW], const float* const (&Wnext)[2],
                          const int (&abase)[3], int wlane, int lane, ConvPrefetch<AHEAD>& pf,
                          f32x4 (&acc)[NW][3]) {
  // Accuracy: one fp32 accumulator over all K = 576 terms is a 576-long sequential sum.  Every TAP (64 terms)
  // is accumulated on its own, from zero, and the nine tap sums are added in tap order: tap t runs on `acc`
  // (even t) or `alt` (odd t), and a set is folded into `tot` when it is needed again two taps later (its MFMAs
  // have long retired: no accumulator-read stall in the loop).  Against an fp64 evaluation of the whole
  // recurrent_fn the next state's mean error goes from 1.7 x MIOpen fp32's (one chain) to 0.8 x; the folds cost
  // 3.5-4.5 % of the launch (profiles/r02_tower_accuracy.txt; MZ_CONV_FOLD_TAPS = 3 / 9 rebuild the other orders).
  f32x4 alt[NW][3], tot[NW][3];
  f32x4 rem[NW], ralt[NW], rtot[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[s][mt] = alt[s][mt] = tot[s][mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    rem[s] = ralt[s] = rtot[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  }
  constexpr int G = 36;
  f32x4u wbuf[NW][AHEAD + 1];
  f32x4u abuf[2][2];
  auto a_off = [](int grp) { return (((grp >> 2) / 3) * kHalo + ((grp >> 2) % 3)) * kPixStride + 16 * (grp & 3); };
  // remainder rows: lane reads pixel 32 + (lane & 3), input channels 16 c + 4 (lane >> 4) + {0..3}
  const int rbase = ((32 + (lane & 3)) / kTowerHW * kHalo + (32 + (lane & 3)) % kTowerHW) * kPixStride + 4 * (lane >> 4);
  f32x4u ra[2];
  constexpr bool REM = TSEL != 1;
#pragma unroll
  for (int s = 0; s < NW; ++s) {
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) wbuf[s][q] = pf.q[s][q];
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    if (tile_on<TSEL>(mt)) abuf[0][mt] = *reinterpret_cast<const f32x4u*>(in + abase[mt] + a_off(0));
  if constexpr (REM) ra[0] = *reinterpret_cast<const f32x4u*>(in + rbase + a_off(0));
  StaticFor<0, G>::run([&](auto gc) {
    constexpr int grp = decltype(gc)::value;
    constexpr int tap = grp >> 2, seg = tap / kConvFoldTaps, P = seg & 1;
    if constexpr (grp + AHEAD < G) {
#pragma unroll
      for (int s = 0; s < NW; ++s)
        wbuf[s][(grp + AHEAD) % (AHEAD + 1)] = (reinterpret_cast<const f32x4u*>(Wp[s]) + wlane)[(grp + AHEAD) * 256];
    } else {
      // unconditional (the caller passes valid pointers even when fewer streams follow): a load under a
      // branch makes the compiler's vmcnt bookkeeping conservative for every wait that follows it
#pragma unroll
      for (int s = 0; s < 2; ++s)
        pf.q[s][grp + AHEAD - G] = (reinterpret_cast<const f32x4u*>(Wnext[s]) + wlane)[(grp + AHEAD - G) * 256];
    }
    if constexpr (grp + 1 < G) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        if (tile_on<TSEL>(mt))
          abuf[(grp + 1) & 1][mt] = *reinterpret_cast<const f32x4u*>(in + abase[mt] + a_off(grp + 1));
      if constexpr (REM) ra[(grp + 1) & 1] = *reinterpret_cast<const f32x4u*>(in + rbase + a_off(grp + 1));
    }
    if constexpr ((grp & 3) == 0 && tap % kConvFoldTaps == 0 && seg >= 2) {
      // tap t starts: fold the sum of tap t - 2 (this accumulator set) into the total, restart from zero
#pragma unroll
      for (int s = 0; s < NW; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          if (tile_on<TSEL>(mt)) {
            f32x4& cur = P ? alt[s][mt] : acc[s][mt];
            tot[s][mt] = seg == 2 ? cur : tot[s][mt] + cur;
            cur = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
          }
        if constexpr (REM) {
          f32x4& cur = P ? ralt[s] : rem[s];
          rtot[s] = seg == 2 ? cur : rtot[s] + cur;
          cur = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
          // the 4x4x1 chain of tap t - 2 ends HERE: left alone the compiler moves all 144 remainder MFMAs of a convolution
          // into ONE block behind its 36 groups, parks their operands in accumulation registers on the way (~120 moves)
          // and pays a hazard wait state between every two of the dependent chain: 26.9 -> 26.55 ms per act
          asm volatile("" : "+v"(rtot[s]));
        }
      }
    }
#ifdef MZ_CONV_NO_INTERLEAVE
    __builtin_amdgcn_sched_barrier(0);  // (rounds 1-4: everything ahead of the MFMA block)
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < NW; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          if (tile_on<TSEL>(mt)) {
            f32x4& cur = P ? alt[s][mt] : acc[s][mt];
            cur = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[grp & 1][mt][i], wbuf[s][grp % (AHEAD + 1)][i], cur, 0, 0, 0);
          }
        if constexpr (REM) {
          f32x4& cur = P ? ralt[s] : rem[s];
          cur = __builtin_amdgcn_mfma_f32_4x4x1f32(ra[grp & 1][i], wbuf[s][grp % (AHEAD + 1)][i], cur, 0, 0, 0);
          if (i == 3 && (grp & 3) == 3 && tap >= 7) asm volatile("" : "+v"(cur));  // (taps 7, 8: never folded in the loop)
        }
      }
#ifndef MZ_CONV_NO_INTERLEAVE
    // the group's loads and arithmetic BETWEEN its first matrix instructions (each 16x16x4 covers eight issue slots):
    // with everything ahead of the MFMA block (a hard scheduling barrier, until round 4) only the last MFMA of the
    // previous group covered the ~16 other instructions of a group: 27.45 -> 26.9 ms per act of config 4
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    // (... then one MFMA and up to seven VALU / SALU slots, twice, then the rest of the MFMAs: six-slot groups of VALU
    // only measured 0.25 ms per act worse, four groups of four no better; the order of the MFMAs themselves -- tile and
    // remainder alternating, or the tile chain first -- makes no measurable difference)
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x006, 7, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x006, 7, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  });
  // taps 7 (alt) and 8 (acc) are still in their accumulators: total = ((sum of taps 0..6) + tap 7) + tap 8
  if constexpr (kConvFoldTaps != 9) {
#pragma unroll
    for (int s = 0; s < NW; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        if (tile_on<TSEL>(mt)) acc[s][mt] = (tot[s][mt] + alt[s][mt]) + acc[s][mt];
  }
  if constexpr (REM) {
    // the four k-quads of a channel: lanes n, n + 16, n + 32, n + 48 -> every lane gets the sum; the lanes
    // g = 0 own pixels 32..35 in the tile layout (acc[.][2][v] <-> pixel 32 + 4 g + v), the others are masked
#pragma unroll
    for (int s = 0; s < NW; ++s) {
      const f32x4 rsum = kConvFoldTaps == 9 ? rem[s] : (rtot[s] + ralt[s]) + rem[s];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float t = rsum[v];
        t = t + lane_xor16(t);
        t = t + lane_xor32(t);
        acc[s][2][v] = t;
      }
    }
  }
}
// moments of half 0 (16 x 64 elements) and half 1 (20 x 64) -> mean and 1 / sqrt(var + eps) of the whole map
MZ_DEV void merge_moments(float m0, float q0, float m1, float q1, float& mean, float& rstd) {
  constexpr float n0 = 16.0f * kTowerC, n1 = 20.0f * kTowerC, n = n0 + n1;
  const float d = m1 - m0;
  mean = m0 + d * (n1 / n);
  const float M2 = (q0 + q1) + (d * d) * (n0 * n1 / n);
  rstd = 1.0f / __builtin_sqrtf(M2 * (1.0f / n) + 1e-5f);
}
// LayerNorm over the whole map (hk.LayerNorm(axis=(-3,-2,-1)), biased variance, eps 1e-5), one workgroup per
// root.  The moments are taken EXACTLY as the two workgroups of a root take them in pair mode (below): (mean,
// M2) of pixels 0..15 and of pixels 16..35 separately -- same per-lane order, same butterfly, same wave order --
// then Chan's merge (merge_moments).  The two launch shapes therefore give the same bits for every element, and
// a search does not depend on how many roots share a launch (<= 128: pair mode, above: this one; or how a batch
// is sharded over GPUs).
template <int NW>
MZ_DEV void layer_norm_tiles(f32x4 (&acc)[NW][3], const float* const (&so)[NW], const bool (&relu)[NW], int ch, int lane,
                             int wave, RedSlots& red) {
  const int g = lane >> 4;
  float mh[2 * NW], qh[2 * NW];
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      mh[2 * s + h] = 0.0f;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          mh[2 * s + h] = mh[2 * s + h] + (((h == 0 ? mt == 0 : mt >= 1) && 16 * mt + 4 * g + v < kTowerPix) ? acc[s][mt][v] : 0.0f);
    }
  wg_sum<2 * NW>(mh, red, wave, lane);
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      mh[2 * s + h] = mh[2 * s + h] * (h == 0 ? 1.0f / (16 * kTowerC) : 1.0f / (20 * kTowerC));
      qh[2 * s + h] = 0.0f;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float d = acc[s][mt][v] - mh[2 * s + h];
          qh[2 * s + h] = qh[2 * s + h] + (((h == 0 ? mt == 0 : mt >= 1) && 16 * mt + 4 * g + v < kTowerPix) ? d * d : 0.0f);
        }
    }
  wg_sum<2 * NW>(qh, red, wave, lane);
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    float mean, rstd;
    merge_moments(mh[2 * s], qh[2 * s], mh[2 * s + 1], qh[2 * s + 1], mean, rstd);
    const float sc = so[s][ch], of = so[s][kTowerC + ch];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float o = (acc[s][mt][v] - mean) * rstd * sc + of;
        acc[s][mt][v] = relu[s] ? fmaxf(o, 0.0f) : o;
      }
  }
}

template <int TSEL = 0>
MZ_DEV void store_map(const f32x4 (&acc)[3], float* buf, int ch, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int p = 16 * mt + 4 * g + v;
      if (tile_on<TSEL>(mt) && p < kTowerPix) buf[((p / kTowerHW + 1) * kHalo + p % kTowerHW + 1) * kPixStride + ch] = acc[mt][v];
    }
}


// ---- small pieces of the heads ----
// conv1x1 as MFMA tiles: acc[mt] (mt = 0..2) = in[px][0..16 KC) . W[k][ncol], `in` rows addressed through
// rowbase[mt] (word offsets of this lane's pixel row + 4 * (lane >> 4)), W row-major [K][ldw]
MZ_DEV void conv1x1_tiles(const float* in, const int (&rowbase)[3], const float* __restrict__ W, int ldw, int ncol,
                          int KC, int g, f32x4 (&acc)[3]) {
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) acc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  // four groups of 16 input channels at a time, their 16 weights requested before the first MFMA (one group per trip
  // was one L2 round trip per group: 2.4 of the 3 us a 64-channel 1x1 convolution took)
  for (int c0 = 0; c0 < KC; c0 += 4) {
    float b[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) b[c][i] = c0 + c < KC ? W[(16 * (c0 + c) + 4 * g + i) * ldw + ncol] : 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c0 + c < KC) {
        f32x4u a[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const f32x4u*>(in + rowbase[mt] + 16 * (c0 + c));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i], b[c][i], acc[mt], 0, 0, 0);
      }
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
  const float x = lane < F ? logits[lane] : -INFINITY;
  float m = row_max<4>(x);  // DPP inside the rows, the swaps across them (no shuffles through LDS)
  m = fmaxf(m, lane_xor16(m));
  m = fmaxf(m, lane_xor32(m));
  const float e = lane < F ? exp_neg(x - m) : 0.0f;
  const float s = wave_sum64(e), t = wave_sum64(e * (float)(lane - support));
  return inv_scaling(t / s);
}

// ---- pair mode: the two workgroups of one root ---------------------------------------------------------
// With <= 128 roots a launch of one workgroup per root leaves half of the 256 CUs idle.  Pair mode gives a
// root TWO workgroups that split the PIXELS of its map: half 0 owns pixels 0..15 (tile 0), half 1 pixels
// 16..35 (tile 1 + the remainder rows); both keep all 64 channels, so a residual shortcut never leaves its
// workgroup.  A 3x3 window of an owned pixel reaches at most 7 pixels into the other half (pixels 9..15 /
// 16..22), and LayerNorm needs the other half's moments: after every convolution pass the halves swap ONE
// message through L2 -- their local (mean, M2) and the RAW boundary pixels -- and each normalises the
// partner's boundary pixels itself (Chan's merge of the moments, written so that both halves compute the
// same bits).  Messages go through two slots per half (message k in slot k & 1: the partner cannot post
// k + 2 before it has read k); a per-root launch epoch in device memory keeps the numbering going across launches,
// so a captured hipGraph can replay the kernel.  A spin that runs out sets the root's status word instead of
// hanging.  Needs both workgroups resident at once: the host uses it only while 2 B workgroups fit the chip in one
// wave of dispatch.
//
// The two halves of a root sit on the SAME XCD (see mz_resnet_tower_pair_kernel) and meet in that XCD's L2: a CU's L1
// is write-through, so a plain store is in L2 when it completes; the reader's loads bypass its L1 (volatile: sc0 sc1).
// Round 4: every word of a message travels as an 8-byte (value, message number) pair -- one aligned 64-bit store,
// one aligned 64-bit load, never torn -- and the reader polls the very words it needs until they carry the number it
// waits for.  There is no flag any more: rounds 1-3 stored the payload, drained vmcnt, met at a barrier, stored a flag;
// the reader polled the flag, met at a barrier, then fetched the payload -- two more L2 round trips and two more
// barriers per message, 17 messages per pass of the tower (~0.8 us each).  What did NOT work (round 1): agent-scope
// fences (their L2 write-back / invalidate threw the convolution weights out of L2 for every workgroup of the XCD:
// 725 us at 128 roots), group-scope atomics (sc0 loads may hit L1 when a workgroup is not split over CUs), and
// device-scope atomics for every word (each message a trip to the memory side: 4 us per message).  Each message
// carries the sender's XCC id; a half that sees another id than its own reports status 2 -- the L2 rendezvous is only
// valid inside one XCD.  A stamped word says nothing about OTHER stores of its sender: whoever needs those (the fused
// search's hand-over of embedding rows) drains vmcnt and meets at a barrier before it stamps (pair_begin_strong).
constexpr int kPairSlot = 1536;   // VALUES per message slot (8 bytes each): 8 header + 128 (min, max) + 1280 or 2 x 448 payload
constexpr int kPairBnd = 7;       // boundary pixels each half sends
constexpr int kPairMsgs = 64;     // message numbers per tower pass, upper bound (epoch stride)
constexpr unsigned kPairSpin = 1u << 19;
template <int TSEL>
struct PairGeom {
  static constexpr int first_own = TSEL == 1 ? 0 : 16, n_own = TSEL == 1 ? 16 : 20;
  static constexpr int send_first = TSEL == 1 ? 9 : 16, recv_first = TSEL == 1 ? 16 : 9;
};
typedef float pair_word __attribute__((ext_vector_type(2)));  // (value, message number)
struct PairLink {
  pair_word* mine;            // [2][kPairSlot]
  const pair_word* theirs;    // [2][kPairSlot]
  unsigned* status;
  unsigned seq;               // number of the message being written / read
  unsigned xcc;               // 1 + id of the XCD this workgroup runs on
};
// start message L.seq + 1: the slot its words go to (every pair_put after this carries the new number)
MZ_DEV pair_word* pair_begin(PairLink& L) {
  L.seq += 1;
  return L.mine + (L.seq & 1) * kPairSlot;
}
// the same after every earlier store of this workgroup has completed (a stamped word then vouches for them)
MZ_DEV pair_word* pair_begin_strong(PairLink& L) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return pair_begin(L);
}
MZ_DEV const pair_word* pair_in(const PairLink& L) { return L.theirs + (L.seq & 1) * kPairSlot; }  // the partner's message L.seq
MZ_DEV void pair_put(const PairLink& L, pair_word* out, int i, float v) {
  pair_word w;
  w.x = v;
  w.y = __uint_as_float(L.seq);
  out[i] = w;  // one global_store_dwordx2
}
MZ_DEV float pair_get(const PairLink& L, const pair_word* in, int i) {
  const volatile unsigned long long* q = reinterpret_cast<const volatile unsigned long long*>(in + i);
  unsigned long long w = *q;  // one load, sc0 sc1
  unsigned n = 0;
  while ((unsigned)(w >> 32) != L.seq) {
    n += 1;
    if ((n & 255u) == 0u && *reinterpret_cast<const volatile unsigned*>(L.status) != 0u) break;  // once lost, never wait again
    if (n > kPairSpin) {
      *reinterpret_cast<volatile unsigned*>(L.status) = 1u;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    w = *q;
  }
  return __uint_as_float((unsigned)w);
}
// Early copies (round 4): the slower half of a pair (half 1) finds the partner's message already in L2 when it
// finishes its own convolution, yet paid one L2 round trip (~0.65 us, 16 times a pass) to look at it AFTER posting its
// own.  It now issues the loads of the words it will need between its two moment reductions -- in flight while it
// reduces and posts -- and takes a copy whose stamp is the awaited message number; any other copy falls back to polling.
// (a relaxed system-scope atomic load: the same `sc0 sc1` access as the polling load, but the compiler waits for it at
// its first use -- a volatile load gets `s_waitcnt vmcnt(0)` right behind it, six serial round trips)
MZ_DEV unsigned long long pair_peek(const pair_word* in, int i) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(in + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
MZ_DEV float pair_take(const PairLink& L, const pair_word* in, int i, unsigned long long early) {
  if ((unsigned)(early >> 32) == L.seq) return __uint_as_float((unsigned)early);
  return pair_get(L, in, i);
}
// header word 4 of every message: the sender's XCC id
MZ_DEV void pair_check_xcc(const PairLink& L, const pair_word* in, int tid) {
  if (tid == 0 && __float_as_uint(pair_get(L, in, 4)) != L.xcc && *reinterpret_cast<const volatile unsigned*>(L.status) == 0u)
    *reinterpret_cast<volatile unsigned*>(L.status) = 2u;
}
MZ_DEV int map_word(int px) { return ((px / kTowerHW + 1) * kHalo + px % kTowerHW + 1) * kPixStride; }

// moments of the OWN pixels of NW maps: mean over n_own * 64 elements and M2 = sum (x - mean)^2
struct NoMidWork {
  MZ_DEV void operator()() const {}
};
// (`mid` runs between the two reductions: the early loads of the partner's message, below)
template <int NW, int TSEL, class Mid = NoMidWork>
MZ_DEV void own_moments(const f32x4 (&acc)[NW][3], float (&mean)[NW], float (&m2)[NW], int lane, int wave, RedSlots& red,
                        Mid&& mid = Mid()) {
  const int g = lane >> 4;
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    mean[s] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        mean[s] = mean[s] + ((tile_on<TSEL>(mt) && 16 * mt + 4 * g + v < kTowerPix) ? acc[s][mt][v] : 0.0f);
  }
  wg_sum<NW>(mean, red, wave, lane);
  mid();
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    mean[s] = mean[s] * (1.0f / (PairGeom<TSEL>::n_own * kTowerC));
    m2[s] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float d = acc[s][mt][v] - mean[s];
        m2[s] = m2[s] + ((tile_on<TSEL>(mt) && 16 * mt + 4 * g + v < kTowerPix) ? d * d : 0.0f);
      }
  }
  wg_sum<NW>(m2, red, wave, lane);
}
template <int TSEL>
MZ_DEV void put_boundary(const PairLink& L, const f32x4 (&acc)[3], pair_word* out, int first, int ch, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int b = 16 * mt + 4 * g + v - PairGeom<TSEL>::send_first;
      if (tile_on<TSEL>(mt) && b >= 0 && b < kPairBnd) pair_put(L, out, first + b * kTowerC + ch, acc[mt][v]);
    }
}
MZ_DEV void norm_tiles(f32x4 (&acc)[3], float mean, float rstd, const float* so, bool relu, int ch) {
  const float sc = so[ch], of = so[kTowerC + ch];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float o = (acc[mt][v] - mean) * rstd * sc + of;
      acc[mt][v] = relu ? fmaxf(o, 0.0f) : o;
    }
}

// ---- the heads (muax/nn.py:313-357), on full maps ----
struct HeadLds {
  float *hv, *hv2, *hp, *part, *part2, *vec, *lgt;
};
// reward head on [s, a / num_actions] held in `in` (haloed map); `tmp` is a second haloed map.  Three pieces so that
// pair mode can spread it over the idle time of the 16-pixel half (below); one workgroup per root runs them back to back.
// (1) the two 1x1 convolutions + relu -> feature map `fmap` (haloed layout when fmap == tmp, else compact [36][64])
template <bool COMPACT>
MZ_DEV void reward_front(const TowerParams& p, const TowerIO& io, const float* in, float* tmp, float* fmap,
                         const int (&rowc)[3], int ch, int lane) {
  const int g4 = lane >> 4;
  f32x4 acc[3];
  conv1x1_tiles(in, rowc, p.r_c1, kTowerC, ch, 4, g4, acc);
  const float pl = (float)io.action * p.inv_num_actions * p.r_c1[kTowerC * kTowerC + ch];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(acc[mt][v] + pl, 0.0f);
  store_map(acc, tmp, ch, lane);
  __syncthreads();
  conv1x1_tiles(tmp, rowc, p.r_c2, kTowerC, ch, 4, g4, acc);
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(acc[mt][v], 0.0f);
  __syncthreads();  // every wave has read tmp
  if constexpr (COMPACT) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g4 + v;
        if (px < kTowerPix) fmap[px * kTowerC + ch] = acc[mt][v];
      }
  } else {
    store_map(acc, fmap, ch, lane);
  }
  __syncthreads();
}
// (2) Linear(2304 -> 64), one pixel: wave = 9 pixels of the map (9 wave + k, k = 0..8), lane = output unit; `row` = the
// pixel's 64 features in LDS.  576 weights per thread in all, streamed from L2: the loop is bound by load latency, so a
// whole pixel's 64 loads are put in flight before the first fma (8 at a time cost 3 x the time).  The partial sum of a
// lane runs over its wave's pixels in order, channels in order -- whoever calls the pieces, the bits are the same.
MZ_DEV void reward_linear_pixel(const TowerParams& p, const float* row, int px, int lane, float& sacc) {
  const float* wr = p.r_l1 + (size_t)px * kTowerC * kTowerC + lane;
  float w[kTowerC];
#pragma unroll
  for (int c = 0; c < kTowerC; ++c) w[c] = wr[c * kTowerC];
#pragma unroll
  for (int c = 0; c < kTowerC; ++c) sacc = __builtin_fmaf(row[c], w[c], sacc);
}
// (3) the four waves' partial sums -> hidden vector -> logits -> support_to_scalar
MZ_DEV void reward_finish(const TowerParams& p, const TowerIO& io, const HeadLds& H, float sacc, int tid, int lane, int wave) {
  H.part[tid] = sacc;
  __syncthreads();
  if (tid < 64)
    H.vec[tid] = fmaxf(((H.part[tid] + H.part[64 + tid]) + (H.part[128 + tid] + H.part[192 + tid])) + p.r_b1[tid], 0.0f);
  __syncthreads();
  if (tid < p.F) {
    float a = 0.0f;
    for (int k = 0; k < 64; ++k) a = __builtin_fmaf(H.vec[k], p.r_l2[k * p.F + tid], a);
    H.lgt[tid] = a + p.r_b2[tid];
  }
  __syncthreads();
  if (wave == 0) {
    const float rw = decode_support(H.lgt, p.F, p.support, lane);
    if (lane == 0) *io.reward = rw;
  }
  __syncthreads();
}
MZ_DEV void reward_head(const TowerParams& p, const TowerIO& io, const float* in, float* tmp, const HeadLds& H,
                        const int (&rowc)[3], int ch, int tid, int lane, int wave) {
  reward_front<false>(p, io, in, tmp, tmp, rowc, ch, lane);
  float sacc = 0.0f;
  for (int px = 9 * wave; px < 9 * wave + 9; ++px) reward_linear_pixel(p, tmp + map_word(px), px, lane, sacc);
  reward_finish(p, io, H, sacc, tid, lane, wave);
}
// prediction heads on the normalised next state held in `cur` (haloed map)
// `side(k)`, k = 0, 1: work of the CALLER for wave 3, run while waves 0 / 1 are in the first 1x1 convolutions (k = 0: the
// longest stretch of the heads, during which waves 2 and 3 have nothing to do) and while wave 0 is in the value head's
// second convolution (k = 1) -- pair mode hangs the reward head's tail there (round 6)
struct NoSideWork {
  MZ_DEV void operator()(int) const {}
};
template <class Side = NoSideWork>
MZ_DEV void prediction_heads(const TowerParams& p, const TowerIO& io, const float* cur, const HeadLds& H, const int (&rowc)[3],
                             int tid, int lane, int wave MZ_TH_PARAMS, Side&& side = Side()) {
  const int g4 = lane >> 4, n16 = lane & 15;
  // the weights of the last layers do not depend on the map: requested here, before the 1x1 convolutions and their
  // barriers (a barrier is a fence: the compiler cannot hoist them itself).  Round 6: the barriers of this function
  // order LDS only -- every hand-over between its phases goes through LDS -- so these loads (L2 misses: the passes' 5.7 MB
  // of convolution weights have streamed through the L2 since their last use) stay in flight until their use four
  // phases later.  The LDS copy of the flatten -> Linear matrices for the whole search was tried in round 4 (the layer
  // went from 0.4 to 1.9 us).
  const int n = tid & 15, sl = tid >> 4;
  float wl2[16];
  float bl2, bl1;
  {
    // UNCONDITIONAL loads at clamped indices (a thread that is neither a value logit nor a policy logit reads column 0 of
    // the value matrix and never uses it).  Until round 6 they sat under per-thread conditions with a constant 0 in the
    // other lanes: the constant's move into the register a load is still to write forces an s_waitcnt vmcnt(0) on the
    // spot -- a full L2-miss latency (~1.5 us) in front of the heads' first convolutions, every simulation.
    const bool isp = tid >= 64 && tid < 64 + p.A;
    const float* w2 = isp ? p.p_l2 : p.v_l2;
    const float* b2 = isp ? p.p_b2 : p.v_b2;
    const int ld2 = isp ? p.A : p.F;
    const int j = isp ? tid - 64 : (tid < p.F ? tid : 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) wl2[k] = w2[k * ld2 + j];
    bl2 = b2[j];
    bl1 = ((tid & 16) ? p.p_b1 : p.v_b1)[n];  // (used by tid < 32: value units 0..15, policy units 16..31)
  }
  if (wave < 2) {  // wave 0: value head, wave 1: policy head -- first 1x1 conv (64 -> 16) + relu
    f32x4 h[3];
    const float* vw = io.head_w_lds ? io.head_w : p.v_c1;
    const float* pw = io.head_w_lds ? io.head_w + kTowerC * 16 : p.p_c1;
    conv1x1_tiles(cur, rowc, wave == 0 ? vw : pw, 16, n16, 4, g4, h);
    float* dst = wave == 0 ? H.hv : H.hp;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g4 + v;
        if (px < kTowerPix) dst[px * 16 + n16] = fmaxf(h[mt][v], 0.0f);
      }
  } else if (wave == 3) {
    side(0);
  }
  lds_barrier();
  MZ_TH(6)
  if (wave == 3) side(1);
  if (wave == 0) {  // value head: second 1x1 conv (16 -> 16) + relu
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const f32x4 h = conv1x1_k16(H.hv + (16 * mt + n16) * 16 + 4 * g4, p.v_c2, n16, g4);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g4 + v;
        if (px < kTowerPix) H.hv2[px * 16 + n16] = fmaxf(h[v], 0.0f);
      }
    }
  }
  lds_barrier();
  MZ_TH(7)
  {
    // Linear(576 -> 16) of both heads: thread = (output unit n, one of 16 slices of 36 inputs)
    float sv = 0.0f, sp = 0.0f;
    // (requested at the top of the heads instead -- 72 more registers in flight across the convolution phases -- the
    // layer drops from 1.0 to 0.4 us and the act does not get faster, round 5 and again round 6: 24.4-24.7 against
    // 24.4-24.5 ms, same box)
    float wv[36], wp[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) {  // all 72 loads in flight before the first fma
      wv[i] = p.v_l1[(36 * sl + i) * 16 + n];
      wp[i] = p.p_l1[(36 * sl + i) * 16 + n];
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      sv = __builtin_fmaf(H.hv2[36 * sl + i], wv[i], sv);
      sp = __builtin_fmaf(H.hp[36 * sl + i], wp[i], sp);
    }
    lds_barrier();
    H.part[tid] = sv;
    H.part2[tid] = sp;
  }
  lds_barrier();
  MZ_TH(8)
  if (tid < 32) {
    const float* src = tid < 16 ? H.part : H.part2;
    float a = 0.0f;
    for (int q = 0; q < 16; ++q) a = a + src[q * 16 + n];
    H.vec[tid] = fmaxf(a + bl1, 0.0f);
  }
  lds_barrier();
  if (tid < p.F) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a = __builtin_fmaf(H.vec[k], wl2[k], a);
    H.lgt[tid] = a + bl2;
  } else if (tid >= 64 && tid < 64 + p.A) {
    const int j = tid - 64;
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a = __builtin_fmaf(H.vec[16 + k], wl2[k], a);
    io.prior_logits[j] = a + bl2;
  }
  lds_barrier();
  if (wave == 0) {
    const float vl = decode_support(H.lgt, p.F, p.support, lane);
    if (lane == 0) *io.value = vl;
  }
  MZ_TH(9)
}

MZ_DEV void load_state(const float* xin, float* buf, int tid) {
  for (int i = tid; i < kTowerPix * kTowerC; i += 256) buf[map_word(i >> 6) + (i & 63)] = xin[i];
}

MZ_DEV TowerIO tower_io(const TowerParams& p, int r) {
  TowerIO io;
  io.x = p.x + (size_t)r * kTowerPix * kTowerC;
  io.y = p.y + (size_t)r * kTowerPix * kTowerC;
  io.action = p.action ? p.action[r] : 0;
  io.reward = p.reward ? p.reward + r : nullptr;
  io.value = p.value ? p.value + r : nullptr;
  io.prior_logits = p.prior_logits ? p.prior_logits + (size_t)r * p.A : nullptr;
  return io;
}
// the link of one half of root r to its partner; message numbers continue from the root's epoch word
template <int TSEL>
MZ_DEV void pair_link_init(const TowerParams& p, int r, PairLink& L) {
  constexpr int h = TSEL - 1;
  pair_word* base = reinterpret_cast<pair_word*>(p.pair_f) + (size_t)r * 4 * kPairSlot;
  unsigned* u = p.pair_u + (size_t)r * 4;
  L.mine = base + h * 2 * kPairSlot;
  L.theirs = base + (1 - h) * 2 * kPairSlot;
  L.status = u + 3;
  L.seq = u[2] * kPairMsgs;  // the launch epoch: written only at the very end of a launch, by half 0
  L.xcc = 1u + (unsigned)__builtin_amdgcn_s_getreg(6164);  // hwreg(HW_REG_XCC_ID, 0, 4)
}
// One pass of recurrent_fn for one root (TSEL = 0) or one half of a root (TSEL = 1: pixels 0..15, TSEL = 2: pixels
// 16..35; `L` = the half's link, initialised once per launch).
// `idle(k)`, k = 0, 1, ...: pair mode, 16-pixel half only -- called once per convolution pass after the reward head's
// pixels are done (passes 9 .. 15 of an 8-block tower), between posting a message and waiting for the partner's: work
// the caller wants done in that half's idle time (the fused search loads the tree path of the coming backup there).
// A barrier follows every call.
struct NoIdleWork {
  MZ_DEV void operator()(int) const {}
};
template <int TSEL, class Idle = NoIdleWork>
MZ_DEV void tower_body(const TowerParams& p, const TowerIO& io, float* lds, PairLink& L, Idle&& idle = Idle()) {
  constexpr bool PAIR = TSEL != 0;
  using Geo = PairGeom<TSEL>;
  float* bufA = lds;
  float* bufB = lds + kBufWords;
  RedSlots red = {lds + 2 * kBufWords, 0};
  HeadLds H;
  H.hv = red.base + 32;            // [48][16] value head map (rows >= 36 stay zero)
  H.hv2 = H.hv + 768;         // [48][16]
  H.hp = H.hv2 + 768;         // [48][16] policy head map
  H.part = H.hp + 768;        // [256] partial sums of the flatten -> Linear layers
  H.part2 = H.part + 256;     // [256]
  H.vec = H.part2 + 256;      // [64] hidden vectors
  H.lgt = H.vec + 64;         // [64] logits
  float* rhmap = H.lgt + 64;  // [36][64] pair mode: second feature map of the reward head
  const int tid = opaque_tid(), lane = tid & 63, wave = tid >> 6;
#ifdef MZ_PROFILE
  unsigned long long pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tlast;
#endif
  if (!io.lds_clean) {
    for (int i = tid; i < 2 * kBufWords + kHeadWords - kRhWords; i += 256) lds[i] = 0.0f;  // (rhmap is written before it is read)
    __syncthreads();
  }
  load_state(io.x, bufA, tid);
  __syncthreads();
  MZ_TT(0)

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
  int rowc[3];  // centre-tap rows for 1x1 convolutions on a haloed map
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) rowc[mt] = abase[mt] + (kHalo + 1) * kPixStride;
  if constexpr (!PAIR)
    if (p.heads) reward_head(p, io, bufA, bufB, H, rowc, ch, tid, lane, wave);
  // Pair mode: the reward head (it needs only (s, a)) belongs to the 16-PIXEL half, which idles ~2 us in every one of
  // the 16 convolution passes waiting for the 20-pixel half: its two 1x1 convolutions run here, up front; the
  // flatten -> Linear(2304 -> 64) layer is cut into one pixel per wave and pass (9 of the 16 passes), issued between
  // posting a message and waiting for the partner's; the rest after the tower.  (Until round 4 the 20-pixel half ran
  // the whole head AFTER the tower: 11 - 24 us on the critical path of every simulation.)
  float rh_acc = 0.0f;
  int rh_k = 0;
  if constexpr (TSEL == 1)
    if (p.heads) reward_front<true>(p, io, bufA, bufB, rhmap, rowc, ch, lane);
  MZ_TT(1)

  float* cur = bufA;
  float* oth = bufB;
  if (p.stem_w != nullptr) {
    // conv1x1 on [s, a / num_actions] + relu: the action plane is constant over the map (pair mode: both
    // halves compute the whole stem, it is 1 % of a block)
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
    const float plane = (float)io.action * p.inv_num_actions * p.stem_w[kTowerC * kTowerC + ch];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(acc[mt][v] + plane, 0.0f);
    store_map(acc, oth, ch, lane);
    __syncthreads();
    float* t = cur; cur = oth; oth = t;
  }
  MZ_TT(2)

  constexpr int AH = kConvAhead;  // (10 groups ahead in pair mode measured the same)
  ConvPrefetch<AH> pf;
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
      const float* const nx[2] = {W + 2 * CW, W + 2 * CW};  // one stream follows; the second fetch is a dummy
      conv3x3_tiles<2, TSEL, AH>(cur, w2, nx, abase, wcol, lane, pf, pr);
      MZ_TP(3)
      const float* const so[2] = {LN, LN + 2 * kTowerC};
      if constexpr (!PAIR) {
        const bool rl[2] = {false, true};
        layer_norm_tiles<2>(pr, so, rl, ch, lane, wave, red);
      } else {
        // message A: the raw boundary pixels of conv_0's map (posted first: the stores travel while the moments are
        // reduced) + the moments of both maps
        pair_word* out = pair_begin(L);
        put_boundary<TSEL>(L, pr[1], out, 8, ch, lane);
        const pair_word* ein = pair_in(L);
        float m[2], q[2];
        unsigned long long ea[6] = {0, 0, 0, 0, 0, 0};
        own_moments<2, TSEL>(pr, m, q, lane, wave, red, [&]() {
          if constexpr (TSEL == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ea[k] = pair_peek(ein, k);
            ea[4] = pair_peek(ein, 8 + tid);
            if (tid + 256 < kPairBnd * kTowerC) ea[5] = pair_peek(ein, 8 + tid + 256);
          }
        });
        MZ_TP(4)
        if (tid == 0) {
          pair_put(L, out, 0, m[0]); pair_put(L, out, 1, q[0]); pair_put(L, out, 2, m[1]); pair_put(L, out, 3, q[1]);
          pair_put(L, out, 4, __uint_as_float(L.xcc));
        }
        MZ_TP(5)
        if constexpr (TSEL == 1) {
          if (p.heads && rh_k < 9) reward_linear_pixel(p, rhmap + (9 * wave + rh_k) * kTowerC, 9 * wave + rh_k, lane, rh_acc);
          else if (rh_k >= 9) idle(rh_k - 9);
          rh_k += 1;
          MZ_TT(1)
        }
        const pair_word* in = pair_in(L);
        float mean[2], rstd[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float mo = pair_take(L, in, 2 * s, ea[2 * s]), qo = pair_take(L, in, 2 * s + 1, ea[2 * s + 1]);
          if (TSEL == 1) merge_moments(m[s], q[s], mo, qo, mean[s], rstd[s]);
          else merge_moments(mo, qo, m[s], q[s], mean[s], rstd[s]);
        }
        // (the sender's XCC id is checked once a pass, in message C: a workgroup does not move, and the check is one more
        // L2 round trip of wave 0 -- 0.45 us, 16 times a pass, in the half that does not wait for anything else)
        MZ_TP(6)
        norm_tiles(pr[0], mean[0], rstd[0], so[0], false, ch);
        norm_tiles(pr[1], mean[1], rstd[1], so[1], true, ch);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = tid + 256 * k;
          if (i < kPairBnd * kTowerC) {
            const int c = i & 63;
            const float o = (pair_take(L, in, 8 + i, ea[4 + k]) - mean[1]) * rstd[1] * so[1][c] + so[1][kTowerC + c];
            oth[map_word(Geo::recv_first + (i >> 6)) + c] = fmaxf(o, 0.0f);
          }
        }
      }
    }
    store_map<TSEL>(pr[1], oth, ch, lane);
    __syncthreads();
    MZ_TP(7)
    f32x4 out[1][3];
    {
      const float* const w1[1] = {W + 2 * CW};
      const float* const nx[2] = {last ? W : W + 3 * CW, last ? W : W + 4 * CW};  // (last block: dummies)
      conv3x3_tiles<1, TSEL, AH>(oth, w1, nx, abase, wcol, lane, pf, out);
      MZ_TP(8)
      const float* const so[1] = {LN + 4 * kTowerC};
      if constexpr (!PAIR) {
        const bool rl[1] = {false};
        layer_norm_tiles<1>(out, so, rl, ch, lane, wave, red);
      } else {
        // message B: raw boundary pixels of conv_1's map + the normalised shortcut at those pixels (posted first) + moments
        pair_word* msg = pair_begin(L);
        put_boundary<TSEL>(L, out[0], msg, 8, ch, lane);
        put_boundary<TSEL>(L, pr[0], msg, 8 + kPairBnd * kTowerC, ch, lane);
        const pair_word* ein = pair_in(L);
        float m[1], q[1];
        unsigned long long eb[6] = {0, 0, 0, 0, 0, 0};
        own_moments<1, TSEL>(out, m, q, lane, wave, red, [&]() {
          if constexpr (TSEL == 2) {
            eb[2] = pair_peek(ein, 8 + tid);
            eb[4] = pair_peek(ein, 8 + kPairBnd * kTowerC + tid);
            if (tid + 256 < kPairBnd * kTowerC) {
              eb[3] = pair_peek(ein, 8 + tid + 256);
              eb[5] = pair_peek(ein, 8 + kPairBnd * kTowerC + tid + 256);
            }
          }
        });
        if constexpr (TSEL == 2) {
          // (pass B is nearly balanced: the partner's moments are posted about now, later than its boundary pixels)
          eb[0] = pair_peek(ein, 0);
          eb[1] = pair_peek(ein, 1);
        }
        MZ_TP(4)
        if (tid == 0) {
          pair_put(L, msg, 0, m[0]); pair_put(L, msg, 1, q[0]);
          pair_put(L, msg, 4, __uint_as_float(L.xcc));
        }
        MZ_TP(5)
        if constexpr (TSEL == 1) {
          if (p.heads && rh_k < 9) reward_linear_pixel(p, rhmap + (9 * wave + rh_k) * kTowerC, 9 * wave + rh_k, lane, rh_acc);
          else if (rh_k >= 9) idle(rh_k - 9);
          rh_k += 1;
          MZ_TT(1)
        }
        const pair_word* in = pair_in(L);
        float mean, rstd;
        const float mo = pair_take(L, in, 0, eb[0]), qo = pair_take(L, in, 1, eb[1]);
        MZ_TP(6)
        if (TSEL == 1) merge_moments(m[0], q[0], mo, qo, mean, rstd);
        else merge_moments(mo, qo, m[0], q[0], mean, rstd);
        norm_tiles(out[0], mean, rstd, so[0], false, ch);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = tid + 256 * k;
          if (i < kPairBnd * kTowerC) {
            const int c = i & 63;
            const float o = (pair_take(L, in, 8 + i, eb[2 + k]) - mean) * rstd * so[0][c] + so[0][kTowerC + c];
            cur[map_word(Geo::recv_first + (i >> 6)) + c] =
                fmaxf(pair_take(L, in, 8 + kPairBnd * kTowerC + i, eb[4 + k]) + o, 0.0f);
          }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = fmaxf(pr[0][mt][v] + out[0][mt][v], 0.0f);
    store_map<TSEL>(acc, cur, ch, lane);  // every wave is past its reads of `cur` (the LayerNorm barriers)
    __syncthreads();
    MZ_TP(9)
  }
  if (p.blocks == 0) {
    // (stem only) bring the map back into registers
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g + v;
        acc[mt][v] = px < kTowerPix ? cur[map_word(px) + ch] : 0.0f;
      }
  }

  // min_max_normalize2d (muax/nn.py:47-56): per channel over the 36 pixels
  float mn = INFINITY, mx = -INFINITY;
  {
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bool ok = tile_on<TSEL>(mt) && 16 * mt + 4 * g + v < kTowerPix;
        mn = ok ? fminf(mn, acc[mt][v]) : mn;
        mx = ok ? fmaxf(mx, acc[mt][v]) : mx;
      }
    mn = fminf(mn, lane_xor16(mn)); mn = fminf(mn, lane_xor32(mn));
    mx = fmaxf(mx, lane_xor16(mx)); mx = fmaxf(mx, lane_xor32(mx));
  }
  const pair_word* fin = nullptr;
  if constexpr (PAIR) {
    // message C: per-channel (min, max) of the own pixels; half 1 adds its 20 raw pixels for the heads
    pair_word* msg = pair_begin(L);
    if (tid == 0) pair_put(L, msg, 4, __uint_as_float(L.xcc));
    if (lane < 16) {
      pair_put(L, msg, 8 + ch, mn);
      pair_put(L, msg, 8 + kTowerC + ch, mx);
    }
    if constexpr (TSEL == 2) {
      const int g = lane >> 4;
#pragma unroll
      for (int mt = 1; mt < 3; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int px = 16 * mt + 4 * g + v;
          if (px < kTowerPix) pair_put(L, msg, 8 + 2 * kTowerC + (px - 16) * kTowerC + ch, acc[mt][v]);
        }
    }
    fin = pair_in(L);
    mn = fminf(mn, pair_get(L, fin, 8 + ch));
    mx = fmaxf(mx, pair_get(L, fin, 8 + kTowerC + ch));
    pair_check_xcc(L, fin, tid);
  }
  float scale = mx - mn;
  scale = scale < 1e-5f ? scale + 1e-5f : scale;
  if (p.normalize) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[mt][v] = (acc[mt][v] - mn) / scale;
  }
  float* yout = io.y;
  {
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int px = 16 * mt + 4 * g + v;
        if (tile_on<TSEL>(mt) && px < kTowerPix) yout[px * kTowerC + ch] = acc[mt][v];
      }
  }
  MZ_TT(10)
  if (p.heads) {
    if constexpr (TSEL == 2) {
      // half 1: nothing left (the reward head runs on half 0, spread over its idle time)
    } else {
      if constexpr (TSEL == 1) {
        // (fewer than 5 blocks: the pixels the passes did not get to)
        // (the head's tail -- 4 barriers, ~1.5 us -- was also tried inside idle pass 9: it costs the pass what it saves
        // here, and the stand-alone pass kernel 8 us: profiles/r04_search_phases.txt)
        for (; rh_k < 9; ++rh_k) reward_linear_pixel(p, rhmap + (9 * wave + rh_k) * kTowerC, 9 * wave + rh_k, lane, rh_acc);
        __syncthreads();  // (every wave is done with rhmap: its words become the scratch of the reward head's tail)
      }
      // Round 6: the reward head's tail (four waves' partial sums -> hidden vector -> logits -> support decode:
      // reward_finish, four barriers, 1.6 us in front of the prediction heads) runs on WAVE 3 inside the prediction
      // heads' own barrier intervals -- the vector while the channel (min, scale) pairs are published, the logits while
      // waves 0 / 1 are in the heads' first 1x1 convolutions, the decode during the value head's second one.  Same
      // operations in the same order on other lanes: the same reward, bit for bit.
      float* rpart = rhmap;        // [256] the four waves' partial sums (rhmap is dead by now)
      float* rvec = rhmap + 256;   // [64]
      float* rlgt = rhmap + 320;   // [64]
      if constexpr (TSEL == 1) rpart[tid] = rh_acc;
      MZ_TH(3)
      // ---- prediction heads on the normalised next state ----
      store_map<TSEL>(acc, cur, ch, lane);
      if constexpr (TSEL == 1) {
        // the other 20 pixels arrive raw: normalise them with their channel's (min, scale)
        __syncthreads();
        float* cmn = H.part;   // [64] min, [64] scale per channel
        if (lane < 16) {
          cmn[ch] = mn;
          cmn[kTowerC + ch] = scale;
        }
        if (wave == 3)
          rvec[lane] = fmaxf(((rpart[lane] + rpart[64 + lane]) + (rpart[128 + lane] + rpart[192 + lane])) + p.r_b1[lane], 0.0f);
        __syncthreads();
        {
          // (five words a thread: all five loads in flight at once; a copy that is not message C yet is polled)
          unsigned long long ew[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) ew[k] = pair_peek(fin, 8 + 2 * kTowerC + tid + 256 * k);
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const int i = tid + 256 * k, c = i & 63;
            const float raw = pair_take(L, fin, 8 + 2 * kTowerC + i, ew[k]);
            cur[map_word(16 + (i >> 6)) + c] = p.normalize ? (raw - cmn[c]) / cmn[kTowerC + c] : raw;
          }
        }
      }
      __syncthreads();
      MZ_TH(4)
      if constexpr (TSEL == 1) {
        auto reward_tail = [&](int k) {  // (wave 3 only)
          if (k == 0) {
            if (lane < p.F) {
              float a = 0.0f;
              for (int q = 0; q < 64; ++q) a = __builtin_fmaf(rvec[q], p.r_l2[q * p.F + lane], a);
              rlgt[lane] = a + p.r_b2[lane];
            }
          } else {
            const float rw = decode_support(rlgt, p.F, p.support, lane);  // (the lanes' own writes: same wave, LDS in order)
            if (lane == 0) *io.reward = rw;
          }
        };
        prediction_heads(p, io, cur, H, rowc, tid, lane, wave MZ_TH_ARGS, reward_tail);
      } else {
        prediction_heads(p, io, cur, H, rowc, tid, lane, wave MZ_TH_ARGS);
      }
    }
  }
  MZ_TT(11)
#ifdef MZ_PROFILE
  if (tid == 0) {
    pt[15] = tlast - tstart;
    for (int k = 0; k < 16; ++k) g_tower_prof[(size_t)blockIdx.x * 16 + k] += pt[k];
  }
#endif
}

#ifndef MZ_NO_TOWER_KERNELS
// The tail of root inference with the ResNet nets (muax/model.py:251-263): ResNetRepresentation's last
// hk.AvgPool(3, 2, 'SAME') (the mean of the VALID elements under a window) of its [H][W][64] map (H, W in {11, 12}: 6 x 6
// out) + min_max_normalize2d (muax/nn.py:47-56, :308-310), then ResNetPrediction on that embedding and the value's
// support decode (muax/nn.py:313-341, muax/model.py:254) -- one workgroup per root; in the torch mirror these were ~35
// small framework kernels (two library GEMMs per head among them), 0.17 of the 1.1 ms root inference.
struct RootTailParams {
  const float* x;     // [B][H][W][64]
  float* embedding;   // [B][36][64]
  int H, W, normalize;
};
__global__ __launch_bounds__(256) void mz_resnet_root_tail_kernel(const TowerParams p, const RootTailParams t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = blockIdx.x;
  float* cur = lds;
  RedSlots red = {lds + 2 * kBufWords, 0};
  HeadLds H;
  H.hv = red.base + 32; H.hv2 = H.hv + 768; H.hp = H.hv2 + 768; H.part = H.hp + 768; H.part2 = H.part + 256;
  H.vec = H.part2 + 256; H.lgt = H.vec + 64;
  for (int i = tid; i < 2 * kBufWords + kHeadWords - kRhWords; i += 256) lds[i] = 0.0f;
  __syncthreads();
  // pooling: thread = (channel c, nine of the 36 output pixels); SAME geometry: window rows 2 oy - pt .. + 2
  const int c = tid & 63, q = tid >> 6;
  const int pt = max(5 * 2 + 3 - t.H, 0) / 2, pl = max(5 * 2 + 3 - t.W, 0) / 2;
  const float* img = t.x + (size_t)r * t.H * t.W * kTowerC;
  float v[9];
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int px = 9 * q + k, oy = px / kTowerHW, ox = px - oy * kTowerHW;
    float sum = 0.0f;
    int cnt = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int y = 2 * oy - pt + ky, x = 2 * ox - pl + kx;
        if (y >= 0 && y < t.H && x >= 0 && x < t.W) {
          sum = sum + img[((size_t)y * t.W + x) * kTowerC + c];
          cnt += 1;
        }
      }
    v[k] = sum / (float)cnt;
    mn = fminf(mn, v[k]);
    mx = fmaxf(mx, v[k]);
  }
  // per-channel min / max over the 36 pixels: four partial pairs per channel through LDS
  H.part[tid] = mn;
  H.part2[tid] = mx;
  __syncthreads();
  mn = fminf(fminf(H.part[c], H.part[64 + c]), fminf(H.part[128 + c], H.part[192 + c]));
  mx = fmaxf(fmaxf(H.part2[c], H.part2[64 + c]), fmaxf(H.part2[128 + c], H.part2[192 + c]));
  float scale = mx - mn;
  scale = scale < 1e-5f ? scale + 1e-5f : scale;
  float* emb = t.embedding + (size_t)r * kTowerPix * kTowerC;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int px = 9 * q + k;
    const float o = t.normalize ? (v[k] - mn) / scale : v[k];
    cur[map_word(px) + c] = o;
    emb[px * kTowerC + c] = o;
  }
  __syncthreads();
  int rowc[3];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int px = 16 * mt + (lane & 15);
    rowc[mt] = (px < kTowerPix ? ((px / kTowerHW) * kHalo + px % kTowerHW) * kPixStride : kHalo * kHalo * kPixStride) +
               4 * (lane >> 4) + (kHalo + 1) * kPixStride;
  }
  TowerIO io;
  io.x = nullptr; io.y = nullptr; io.action = 0; io.reward = nullptr;
  io.value = p.value + r;
  io.prior_logits = p.prior_logits + (size_t)r * p.A;
#if defined(MZ_PROFILE) && defined(MZ_PROF_HEADS)
  unsigned long long pt_[16] = {0}, tlast_ = 0;
  prediction_heads(p, io, cur, H, rowc, tid, lane, wave, pt_, tlast_);
#else
  prediction_heads(p, io, cur, H, rowc, tid, lane, wave);
#endif
}

__global__ __launch_bounds__(256) void mz_resnet_tower_kernel(const TowerParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  PairLink L;  // (unused: one workgroup per root)
  tower_body<0>(p, tower_io(p, blockIdx.x), lds, L);
}
// two workgroups per root, 16 blocks = 8 roots x 2 halves laid out so that the halves of a root are 8
// blocks apart: with the round-robin dispatch over the 8 XCDs they land on the same XCD and share its L2
__global__ __launch_bounds__(256) void mz_resnet_tower_pair_kernel(const TowerParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7), h = (blockIdx.x >> 3) & 1;
  if (r >= p.B) return;
  const TowerIO io = tower_io(p, r);
  PairLink L;
  if (h == 0) {
    pair_link_init<1>(p, r, L);
    tower_body<1>(p, io, lds, L);
  } else {
    pair_link_init<2>(p, r, L);
    tower_body<2>(p, io, lds, L);
  }
  // next launch's epoch, by half 0 at its end: both halves read the word before their first message, and half 0 has
  // received the last message of this launch by now
  if (h == 0 && threadIdx.x == 0) p.pair_u[(size_t)r * 4 + 2] += 1;
}
#endif  // MZ_NO_TOWER_KERNELS

}  // namespace mz
