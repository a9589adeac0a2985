// mz_spec.cuh -- device-side arithmetic spec "MZ-F32" and CDNA4 row primitives.
//
// One search root is owned by one DPP row (16 lanes of a 64-wide wavefront), so
// every cross-lane exchange on the path is a DPP modifier on a VALU op -- no LDS
// round trip, no ds_bpermute.  All float arithmetic is IEEE binary32 RN
// (+,-,*,/,sqrt,fma; -ffp-contract=off, fused multiply-adds only where written
// as __builtin_fmaf); exp/log are spelled out so that results do not depend on
// a device math library.  The reduction order of every float sum is the
// "canonical 16-wide sum": 16 partials p_l = x_l + x_{l+16} + ... then an xor
// butterfly over 1,2,4,8 -- which is exactly what the DPP butterfly below does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#pragma clang fp contract(off)

namespace mz {

#define MZ_DEV __device__ __forceinline__

constexpr float kFltTiny = 1.17549435e-38f;
constexpr float kFltLowest = -3.40282347e+38f;

template <int I, int N>
struct StaticFor {
  template <class Fn>
  static MZ_DEV void run(Fn&& f) {
    f(std::integral_constant<int, I>{});
    StaticFor<I + 1, N>::run(f);
  }
};
template <int N>
struct StaticFor<N, N> {
  template <class Fn>
  static MZ_DEV void run(Fn&&) {}
};

// threadIdx.x behind an empty asm statement: a device function that is called once per iteration of a long loop (the
// one-launch ResNet search) must not have its lane-dependent address arithmetic hoisted out of that loop -- the
// hundreds of pre-computed offsets then live across the whole body and spill (measured: 2.2 KB of scratch per lane,
// the heads of the recurrent pass 6x slower)
MZ_DEV int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

MZ_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
MZ_DEV uint32_t f2u(float f) { return __float_as_uint(f); }

// ---------------------------------------------------------------------------
// DPP row primitives (a row = 16 consecutive lanes)
// ---------------------------------------------------------------------------
constexpr int kDppXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror (== xor 4 once quads are uniform)
constexpr int kDppMirror = 0x140;     // row_mirror      (== xor 8 once halves are uniform)
constexpr int kDppBcast0 = 0x150;     // row_newbcast:0 (gfx90a+)

// Every control used through these two reads a valid lane in every lane (quad_perm, row mirrors,
// row_newbcast), so the "old" operand is dead: bound_ctrl:1 tells the compiler so and saves the
// zero-initialising v_mov in front of each DPP move.
template <int CTRL>
MZ_DEV int dpp_i(int x) {
  return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
MZ_DEV float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// value of lane I of this row, in every lane of the row
template <int I>
MZ_DEV float bcast(float x) { return dpp_f<kDppBcast0 + I>(x); }
template <int I>
MZ_DEV int bcast_i(int x) { return dpp_i<kDppBcast0 + I>(x); }
template <int I>
MZ_DEV uint32_t bcast_u(uint32_t x) { return (uint32_t)dpp_i<kDppBcast0 + I>((int)x); }

// acc += (lane I of this row of x) * w as ONE instruction: v_fmac_f32 with a row_newbcast DPP source (a fused
// multiply-add, the same rounding as __builtin_fmaf(bcast<I>(x), w, acc)) -- no separate broadcast move.  FIRST opens
// a chain: x may have been written by the instruction just before, and a DPP read of a fresh VGPR needs two wait
// states the assembler does not insert inside inline assembly; the later links of the chain depend on the first.
template <int I, bool FIRST>
MZ_DEV void fmac_bcast(float& acc, float x, float w) {
  if constexpr (FIRST)
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(I));
  else
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(I));
}

// Eight links of TWO such chains that share their input (acc0 += x[lane B + k] * w[k].x, acc1 += ... * w[k].y,
// k = 0..7, B = 0 or 8) as one statement: between separate asm statements on the same accumulator hipcc pads a wait
// state each time.  The two chains alternate, so no link waits for the one just before it.
#define MZ_FB_LINE(k, b) "v_fmac_f32_dpp %0, %2, %" #k " row_newbcast:" #b " row_mask:0xf bank_mask:0xf\n\t"
#define MZ_FB_LINE2(k, b) "v_fmac_f32_dpp %1, %2, %" #k " row_newbcast:" #b " row_mask:0xf bank_mask:0xf\n\t"
#define MZ_FB_OPS(w)                                                                                        \
  "v"(w[0].x), "v"(w[0].y), "v"(w[1].x), "v"(w[1].y), "v"(w[2].x), "v"(w[2].y), "v"(w[3].x), "v"(w[3].y), \
      "v"(w[4].x), "v"(w[4].y), "v"(w[5].x), "v"(w[5].y), "v"(w[6].x), "v"(w[6].y), "v"(w[7].x), "v"(w[7].y)
template <int B, bool FIRST, class W>
MZ_DEV void fmac_bcast_pair8(float& acc0, float& acc1, float x, const W* w) {
  static_assert(B == 0 || B == 8, "");
  static_assert(!FIRST || B == 0, "");
  // FIRST: the first eight lanes of a slot register, i.e. a register this chain has not read yet.  It may have been
  // written by the instruction just before this statement (a select that masks the lanes past E, a move), and the
  // compiler's hazard recogniser does not look inside an asm statement: the two wait states of a DPP read are spelled
  // out.  (Until round 4 only the chain's very first block had them; an on-demand instance with E = 24 -- whose third
  // block starts the second slot register right behind that select -- read a stale lane in the first link of chain 0.)
  if constexpr (FIRST)
    asm("s_nop 1\n\t" MZ_FB_LINE(3, 0) MZ_FB_LINE2(4, 0) MZ_FB_LINE(5, 1) MZ_FB_LINE2(6, 1) MZ_FB_LINE(7, 2) MZ_FB_LINE2(8, 2)
        MZ_FB_LINE(9, 3) MZ_FB_LINE2(10, 3) MZ_FB_LINE(11, 4) MZ_FB_LINE2(12, 4) MZ_FB_LINE(13, 5) MZ_FB_LINE2(14, 5)
        MZ_FB_LINE(15, 6) MZ_FB_LINE2(16, 6) MZ_FB_LINE(17, 7) MZ_FB_LINE2(18, 7)
        : "+v"(acc0), "+v"(acc1) : "v"(x), MZ_FB_OPS(w));
  else if constexpr (B == 0)
    asm(MZ_FB_LINE(3, 0) MZ_FB_LINE2(4, 0) MZ_FB_LINE(5, 1) MZ_FB_LINE2(6, 1) MZ_FB_LINE(7, 2) MZ_FB_LINE2(8, 2)
        MZ_FB_LINE(9, 3) MZ_FB_LINE2(10, 3) MZ_FB_LINE(11, 4) MZ_FB_LINE2(12, 4) MZ_FB_LINE(13, 5) MZ_FB_LINE2(14, 5)
        MZ_FB_LINE(15, 6) MZ_FB_LINE2(16, 6) MZ_FB_LINE(17, 7) MZ_FB_LINE2(18, 7)
        : "+v"(acc0), "+v"(acc1) : "v"(x), MZ_FB_OPS(w));
  else
    asm(MZ_FB_LINE(3, 8) MZ_FB_LINE2(4, 8) MZ_FB_LINE(5, 9) MZ_FB_LINE2(6, 9) MZ_FB_LINE(7, 10) MZ_FB_LINE2(8, 10)
        MZ_FB_LINE(9, 11) MZ_FB_LINE2(10, 11) MZ_FB_LINE(11, 12) MZ_FB_LINE2(12, 12) MZ_FB_LINE(13, 13) MZ_FB_LINE2(14, 13)
        MZ_FB_LINE(15, 14) MZ_FB_LINE2(16, 14) MZ_FB_LINE(17, 15) MZ_FB_LINE2(18, 15)
        : "+v"(acc0), "+v"(acc1) : "v"(x), MZ_FB_OPS(w));
}
#undef MZ_FB_LINE
#undef MZ_FB_LINE2
#undef MZ_FB_OPS

// Four links (lanes B .. B + 3 of the inputs) of the second-layer chains of a network pass in one statement:
// accumulators a0, a1 (+ a2, a3) fed from xa, accumulators fed from xb; w?[k] = this lane's weights for input B + k.
#define MZ_F(acc, x, w, b) "v_fmac_f32_dpp %" #acc ", %" #x ", %" #w " row_newbcast:" #b " row_mask:0xf bank_mask:0xf\n\t"
// 2 + 2 chains: %0 %1 <- %4 (xa), %2 %3 <- %5 (xb); weights %6.. in (k, chain) order
#define MZ_F4(b0, b1, b2, b3)                                                                          \
  MZ_F(0, 4, 6, b0) MZ_F(1, 4, 7, b0) MZ_F(2, 5, 8, b0) MZ_F(3, 5, 9, b0)                \
      MZ_F(0, 4, 10, b1) MZ_F(1, 4, 11, b1) MZ_F(2, 5, 12, b1) MZ_F(3, 5, 13, b1)                      \
      MZ_F(0, 4, 14, b2) MZ_F(1, 4, 15, b2) MZ_F(2, 5, 16, b2) MZ_F(3, 5, 17, b2)                      \
      MZ_F(0, 4, 18, b3) MZ_F(1, 4, 19, b3) MZ_F(2, 5, 20, b3) MZ_F(3, 5, 21, b3)
template <int B>
MZ_DEV void fmac_bcast_2x2(float& a0, float& a1, float& a2, float& a3, float xa, float xb,
                           const float (*wa)[2], const float (*wb)[2]) {
  static_assert(B % 4 == 0 && B < 16, "");
#define MZ_OPS                                                                                         \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                             \
  : "v"(xa), "v"(xb), "v"(wa[0][0]), "v"(wa[0][1]), "v"(wb[0][0]), "v"(wb[0][1]), "v"(wa[1][0]), "v"(wa[1][1]),     \
    "v"(wb[1][0]), "v"(wb[1][1]), "v"(wa[2][0]), "v"(wa[2][1]), "v"(wb[2][0]), "v"(wb[2][1]), "v"(wa[3][0]),        \
    "v"(wa[3][1]), "v"(wb[3][0]), "v"(wb[3][1])
  if constexpr (B == 0) asm("s_nop 1\n\t" MZ_F4(0, 1, 2, 3) MZ_OPS);  // (xa, xb may be fresh)
  else if constexpr (B == 4) asm(MZ_F4(4, 5, 6, 7) MZ_OPS);
  else if constexpr (B == 8) asm(MZ_F4(8, 9, 10, 11) MZ_OPS);
  else asm(MZ_F4(12, 13, 14, 15) MZ_OPS);
#undef MZ_OPS
}
#undef MZ_F4
// 2 + 1 chains: %0 %1 <- %3 (xa), %2 <- %4 (xb); weights %5.. in (k, chain) order
#define MZ_F3(b0, b1, b2, b3)                                                                          \
  MZ_F(0, 3, 5, b0) MZ_F(1, 3, 6, b0) MZ_F(2, 4, 7, b0) MZ_F(0, 3, 8, b1) MZ_F(1, 3, 9, b1)  \
      MZ_F(2, 4, 10, b1) MZ_F(0, 3, 11, b2) MZ_F(1, 3, 12, b2) MZ_F(2, 4, 13, b2) MZ_F(0, 3, 14, b3)    \
      MZ_F(1, 3, 15, b3) MZ_F(2, 4, 16, b3)
template <int B>
MZ_DEV void fmac_bcast_2x1(float& a0, float& a1, float& a2, float xa, float xb, const float (*wa)[2],
                           const float (*wb)[1]) {
  static_assert(B % 4 == 0 && B < 16, "");
#define MZ_OPS                                                                                         \
  : "+v"(a0), "+v"(a1), "+v"(a2)                                                                       \
  : "v"(xa), "v"(xb), "v"(wa[0][0]), "v"(wa[0][1]), "v"(wb[0][0]), "v"(wa[1][0]), "v"(wa[1][1]), "v"(wb[1][0]),   \
    "v"(wa[2][0]), "v"(wa[2][1]), "v"(wb[2][0]), "v"(wa[3][0]), "v"(wa[3][1]), "v"(wb[3][0])
  if constexpr (B == 0) asm("s_nop 1\n\t" MZ_F3(0, 1, 2, 3) MZ_OPS);
  else if constexpr (B == 4) asm(MZ_F3(4, 5, 6, 7) MZ_OPS);
  else if constexpr (B == 8) asm(MZ_F3(8, 9, 10, 11) MZ_OPS);
  else asm(MZ_F3(12, 13, 14, 15) MZ_OPS);
#undef MZ_OPS
}
#undef MZ_F3
#undef MZ_F

template <int STEP>
struct Bfly;
template <> struct Bfly<0> { static constexpr int ctrl = kDppXor1; };
template <> struct Bfly<1> { static constexpr int ctrl = kDppXor2; };
template <> struct Bfly<2> { static constexpr int ctrl = kDppHalfMirror; };
template <> struct Bfly<3> { static constexpr int ctrl = kDppMirror; };

constexpr int ceil_log2(int n) { return n <= 1 ? 0 : 1 + ceil_log2((n + 1) / 2); }

// canonical 16-wide sum of the per-lane partials (all 4 butterfly steps)
MZ_DEV float row_sum(float p) {
  p = p + dpp_f<kDppXor1>(p);
  p = p + dpp_f<kDppXor2>(p);
  p = p + dpp_f<kDppHalfMirror>(p);
  p = p + dpp_f<kDppMirror>(p);
  return p;
}
// first STEPS butterfly steps only: exact when the lanes beyond 2^STEPS hold +0 and only
// lanes < 2^STEPS use the result
template <int STEPS>
MZ_DEV float row_sum_steps(float p) {
  StaticFor<0, STEPS>::run([&](auto ic) {
    constexpr int s = decltype(ic)::value;
    p = p + dpp_f<Bfly<s>::ctrl>(p);
  });
  return p;
}
// max / min over the first 2^STEPS-aligned group of lanes (order independent)
template <int STEPS>
MZ_DEV float row_max(float x) {
  StaticFor<0, STEPS>::run([&](auto ic) {
    constexpr int s = decltype(ic)::value;
    x = fmaxf(x, dpp_f<Bfly<s>::ctrl>(x));
  });
  return x;
}
template <int STEPS>
MZ_DEV float row_min(float x) {
  StaticFor<0, STEPS>::run([&](auto ic) {
    constexpr int s = decltype(ic)::value;
    x = fminf(x, dpp_f<Bfly<s>::ctrl>(x));
  });
  return x;
}
MZ_DEV int row_sum_i(int p) {
  p = p + dpp_i<kDppXor1>(p);
  p = p + dpp_i<kDppXor2>(p);
  p = p + dpp_i<kDppHalfMirror>(p);
  p = p + dpp_i<kDppMirror>(p);
  return p;
}
MZ_DEV int row_max_i(int x) {
  x = max(x, dpp_i<kDppXor1>(x));
  x = max(x, dpp_i<kDppXor2>(x));
  x = max(x, dpp_i<kDppHalfMirror>(x));
  x = max(x, dpp_i<kDppMirror>(x));
  return x;
}
// first-max argmax over (score, index) with a payload riding along
template <int STEPS>
MZ_DEV void row_argmax(float& score, int& idx, int& payload) {
  StaticFor<0, STEPS>::run([&](auto ic) {
    constexpr int s = decltype(ic)::value;
    float os = dpp_f<Bfly<s>::ctrl>(score);
    int oi = dpp_i<Bfly<s>::ctrl>(idx);
    int op = dpp_i<Bfly<s>::ctrl>(payload);
    // (bitwise, not short-circuit: the compiler turned the || / && form into exec-masked branches around three moves)
    const bool take = (os > score) | ((os == score) & (oi < idx));
    score = take ? os : score;
    idx = take ? oi : idx;
    payload = take ? op : payload;
  });
}

// ---------------------------------------------------------------------------
// MZ-F32 scalar math
// ---------------------------------------------------------------------------
MZ_DEV float exp_core(float x, int& k) {
  const float LOG2E = 1.44269504088896341f;
  const float LN2_HI = 6.93145752e-1f;
  const float LN2_LO = 1.42860677e-6f;
  float kf = __builtin_rintf(x * LOG2E);
  float r = __builtin_fmaf(kf, -LN2_HI, x);
  r = __builtin_fmaf(kf, -LN2_LO, r);
  float p = 1.0f / 5040.0f;
  p = __builtin_fmaf(p, r, 1.0f / 720.0f);
  p = __builtin_fmaf(p, r, 1.0f / 120.0f);
  p = __builtin_fmaf(p, r, 1.0f / 24.0f);
  p = __builtin_fmaf(p, r, 1.0f / 6.0f);
  p = __builtin_fmaf(p, r, 0.5f);
  float rr = r * r;
  k = (int)kf;
  return __builtin_fmaf(p, rr, r);
}
MZ_DEV float pow2i(int k) { return u2f((uint32_t)(k + 127) << 23); }
constexpr float kRintShift = 12582912.0f;  // 1.5 * 2^23
// 2^k from the shifted value's bits (bits(1.5 * 2^23) + k): the constant's bits leave through the top of the shift
MZ_DEV float pow2_shifted(int bits) { return u2f(((uint32_t)bits << 23) + 0x3f800000u); }

MZ_DEV float exp_neg(float x) {  // x <= 88; exact 0 below -87
  int k;
  float q = exp_core(fmaxf(x, -87.0f), k);
  float e = (1.0f + q) * pow2i(k);
  return x < -87.0f ? 0.0f : e;
}
MZ_DEV float elu(float x) {  // jax.nn.elu, alpha = 1
  float xn = fminf(x, 0.0f);
  int k;
  float q = exp_core(fmaxf(xn, -87.0f), k);
  // below -87 the argument is clamped: exp(-87) - 1 = 1.6e-38 - 1 rounds to exactly -1, the value the spec names there
  float em1 = (k == 0) ? q : (1.0f + q) * pow2i(k) - 1.0f;
  return x > 0.0f ? x : em1;
}
MZ_DEV float log_pos(float x) {  // x > 0, normal
  const float LN2_HI = 6.9313812256e-01f;
  const float LN2_LO = 9.0580006145e-06f;
  const float LG1 = 0.66666662693f, LG2 = 0.40000972152f;
  const float LG3 = 0.28498786688f, LG4 = 0.24279078841f;
  uint32_t ix = f2u(x);
  ix += 0x3f800000u - 0x3f3504f3u;
  int e = (int)(ix >> 23) - 127;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  float m = u2f(ix);
  float f = m - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  float w = z * z;
  float t1 = w * (LG2 + w * LG4);
  float t2 = z * (LG1 + w * LG3);
  float R = t2 + t1;
  float hfsq = (0.5f * f) * f;
  float dk = (float)e;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}
// x / 0.002f without the IEEE division sequence: with y = RN(1 / 0.002f), q0 = RN(x y), r = x - q0 * 0.002f (exact,
// fma), q = RN(q0 + r y) is the correctly rounded quotient (Markstein) -- three dependent VALU ops instead of ~11.
// Equality with x / 0.002f is checked for EVERY binary32 x with exponent 2^-27 .. 2^13 (all of them that _inv_scaling
// can produce, and far beyond) by tests/test_oracle_kat.py (mzo_div2eps_mismatches): 344 M cases, none differs.
constexpr float kTwoEps = 0.002f, kRcpTwoEps = 0x1.f3fffep+8f;
MZ_DEV float div_two_eps(float x) {
  const float q0 = x * kRcpTwoEps;
  const float r = __builtin_fmaf(-q0, kTwoEps, x);
  return __builtin_fmaf(r, kRcpTwoEps, q0);
}
// Correctly rounded sqrt for arguments that need no range scaling (x >= 2^-96, finite): v_sqrt_f32 is within one ulp,
// so the answer is s - 1ulp, s or s + 1ulp, decided by the signs of two exact fma residuals -- the core of the
// compiler's own IEEE expansion without its denormal scaling and class checks (9 instructions instead of 17).
// mzs_selftest compares it with sqrtf on the device for every binary32 in [1, 4) (tests/test_gpu_train.py).
MZ_DEV float sqrt_normal(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  float r = rm <= 0.0f ? sm : s;
  r = rp > 0.0f ? sp : r;
  return r;
}
MZ_DEV float inv_scaling(float x) {  // muax/utils.py:70-76, eps = 1e-3
  float ax = fabsf(x);
  float a = (ax + 1.0f) + 0.001f;
  float b = 0.004f * a;
  float c = 1.0f + b;
  float d = sqrt_normal(c);  // c >= 1
  float e = div_two_eps(d - 1.0f);
  float g = e * e - 1.0f;
  float sgn = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
  return sgn * g;
}
// ---- packed pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): component-wise the very same
// operation sequences as the scalar routines above, two independent values per instruction ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // MFMA accumulator tile (mz_train.cuh, mz_conv.cuh)
MZ_DEV f32x2 splat2(float x) { return (f32x2){x, x}; }
MZ_DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// n / d for quotients that share their denominators (the support decode: every e_i / sum): the hardware's own IEEE
// division expansion -- v_rcp_f32, one Newton step on the reciprocal, the quotient and two residual corrections --
// with the reciprocal refined ONCE per denominator, the rest in packed form, and without v_div_scale / v_div_fixup,
// which are identities where this is used: d in [1, 64], n = 0 or 2^-100 <= n <= d (no operand is rescaled, every
// residual is exact).  The caller guards the range; mzs_selftest compares it with n / d on the device.
MZ_DEV f32x2 rcp_newton2(f32x2 d) {
  const f32x2 y0 = (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f32x2 e = fma2(-d, y0, splat2(1.0f));
  return fma2(e, y0, y0);
}
MZ_DEV f32x2 div_newton2(f32x2 n, f32x2 d, f32x2 y) {
  f32x2 q = n * y;
  f32x2 r = fma2(-d, q, n);
  q = fma2(r, y, q);
  r = fma2(-d, q, n);
  return fma2(r, y, q);
}
// exp_neg of an argument >= this is 0 or >= 2^-100 (exp(-69) = 1.08e-30 = 2^-99.5)
constexpr float kDivNewtonMinArg = -69.0f;

MZ_DEV f32x2 exp_core2(f32x2 x, int& k0, int& k1) {
  const float LOG2E = 1.44269504088896341f;
  const float LN2_HI = 6.93145752e-1f;
  const float LN2_LO = 1.42860677e-6f;
  f32x2 t = x * splat2(LOG2E);
  // rint through the 1.5 * 2^23 shift (|t| < 2^22: the same ties-to-even integer as rintf), both components in
  // packed adds; the integer is the low mantissa field of the shifted value, so pow2i needs no conversion
  const f32x2 sh = t + splat2(kRintShift);
  f32x2 kf = sh - splat2(kRintShift);
  f32x2 r = fma2(kf, splat2(-LN2_HI), x);
  r = fma2(kf, splat2(-LN2_LO), r);
  f32x2 p = splat2(1.0f / 5040.0f);
  p = fma2(p, r, splat2(1.0f / 720.0f));
  p = fma2(p, r, splat2(1.0f / 120.0f));
  p = fma2(p, r, splat2(1.0f / 24.0f));
  p = fma2(p, r, splat2(1.0f / 6.0f));
  p = fma2(p, r, splat2(0.5f));
  f32x2 rr = r * r;
  k0 = (int)f2u(sh.x);  // = bits(1.5 * 2^23) + k: callers shift it left by 23 or compare kf with 0
  k1 = (int)f2u(sh.y);
  return fma2(p, rr, r);
}
MZ_DEV f32x2 exp_neg2(f32x2 x) {
  int k0, k1;
  f32x2 xc = (f32x2){fmaxf(x.x, -87.0f), fmaxf(x.y, -87.0f)};
  f32x2 q = exp_core2(xc, k0, k1);
  f32x2 e = (splat2(1.0f) + q) * (f32x2){pow2_shifted(k0), pow2_shifted(k1)};
  return (f32x2){x.x < -87.0f ? 0.0f : e.x, x.y < -87.0f ? 0.0f : e.y};
}
MZ_DEV f32x2 elu2(f32x2 x) {
  int k0, k1;
  // max(min(x, 0), -87) as one v_med3_f32 per component
  f32x2 xc = (f32x2){__builtin_amdgcn_fmed3f(x.x, -87.0f, 0.0f), __builtin_amdgcn_fmed3f(x.y, -87.0f, 0.0f)};
  f32x2 q = exp_core2(xc, k0, k1);
  f32x2 big = (splat2(1.0f) + q) * (f32x2){pow2_shifted(k0), pow2_shifted(k1)} - splat2(1.0f);
  constexpr int kZero = 0x4B400000;  // bits(1.5 * 2^23): k == 0
  float e0 = (k0 == kZero) ? q.x : big.x, e1 = (k1 == kZero) ? q.y : big.y;  // (the clamp at -87 already yields exactly -1 below it)
  return (f32x2){x.x > 0.0f ? x.x : e0, x.y > 0.0f ? x.y : e1};
}
MZ_DEV f32x2 inv_scaling2(f32x2 x) {
  f32x2 ax = (f32x2){fabsf(x.x), fabsf(x.y)};
  f32x2 a = (ax + splat2(1.0f)) + splat2(0.001f);
  f32x2 b = splat2(0.004f) * a;
  f32x2 c = splat2(1.0f) + b;
  f32x2 d = (f32x2){sqrt_normal(c.x), sqrt_normal(c.y)};  // c >= 1
  f32x2 dm = d - splat2(1.0f);
  const f32x2 q0 = dm * splat2(kRcpTwoEps);  // div_two_eps on both components
  const f32x2 e = fma2(fma2(-q0, splat2(kTwoEps), dm), splat2(kRcpTwoEps), q0);
  f32x2 g = e * e - splat2(1.0f);
  float s0 = x.x > 0.0f ? 1.0f : (x.x < 0.0f ? -1.0f : 0.0f);
  float s1 = x.y > 0.0f ? 1.0f : (x.y < 0.0f ? -1.0f : 0.0f);
  return (f32x2){s0, s1} * g;
}

// sqrt(n) * (pb_c_init + log((n + base + 1) / base)), mctx muzero_action_selection
MZ_DEV float puct_scale(int n, float pb_c_init, float pb_c_base) {
  float num = ((float)n + pb_c_base) + 1.0f;
  float pb_c = pb_c_init + log_pos(num / pb_c_base);
  return sqrtf((float)n) * pb_c;
}

// ---------------------------------------------------------------------------
// JAX threefry2x32 (non-partitionable stream)
// ---------------------------------------------------------------------------
MZ_DEV void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
#define MZ_TF_ROUND(r)                       \
  x0 += x1;                                  \
  x1 = __builtin_rotateleft32(x1, r) ^ x0;
  x0 += k0; x1 += k1;
  MZ_TF_ROUND(13) MZ_TF_ROUND(15) MZ_TF_ROUND(26) MZ_TF_ROUND(6)
  x0 += k1; x1 += k2 + 1u;
  MZ_TF_ROUND(17) MZ_TF_ROUND(29) MZ_TF_ROUND(16) MZ_TF_ROUND(24)
  x0 += k2; x1 += k0 + 2u;
  MZ_TF_ROUND(13) MZ_TF_ROUND(15) MZ_TF_ROUND(26) MZ_TF_ROUND(6)
  x0 += k0; x1 += k1 + 3u;
  MZ_TF_ROUND(17) MZ_TF_ROUND(29) MZ_TF_ROUND(16) MZ_TF_ROUND(24)
  x0 += k1; x1 += k2 + 4u;
  MZ_TF_ROUND(13) MZ_TF_ROUND(15) MZ_TF_ROUND(26) MZ_TF_ROUND(6)
  x0 += k2; x1 += k0 + 5u;
#undef MZ_TF_ROUND
}
// counters of the block that produces flat[i] of threefry_2x32(key, iota(size))
MZ_DEV void bits_block(uint64_t size, uint64_t i, uint32_t& x0, uint32_t& x1, bool& second) {
  uint64_t half = (size + 1) >> 1;
  second = i >= half;
  uint64_t blk = second ? i - half : i;
  uint64_t c1 = half + blk;
  x0 = (uint32_t)blk;
  x1 = c1 < size ? (uint32_t)c1 : 0u;
}
MZ_DEV float uniform_from_bits(uint32_t bits) { return u2f((bits >> 9) | 0x3f800000u) - 1.0f; }
MZ_DEV float gumbel_from_bits(uint32_t bits) {
  float u = uniform_from_bits(bits);
  u = u * (1.0f - kFltTiny) + kFltTiny;
  u = fmaxf(u, kFltTiny);
  return -log_pos(-log_pos(u));
}

}  // namespace mz
