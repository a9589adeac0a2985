// mz_train.cuh -- fused forward + backward of the k-step unrolled MuZero loss for the default MLP trio
// (muax/loss.py:10-88 through jax.value_and_grad at muax/model.py:245-249).
//
// Mapping (the search kernel's): one sample owns one DPP row of 16 lanes, feature k of a vector lives in
// lane k & 15 (slot k >> 4), four samples per wavefront.  That layout is ALSO the operand layout of
// v_mfma_f32_16x16x4_f32 with the sample index as its k dimension:
//     A[i][k] : lane 16 k + i holds x_i of sample k        B[k][j] : lane 16 k + j holds dy_j of sample k
// so one MFMA adds X^T dY of the wave's four samples into a 16 x 16 tile of a weight gradient, which stays
// in accumulator registers across samples and unroll steps.  Activations move through the layers as
// row-distributed fma chains (row_newbcast DPP), the weights sit in LDS with odd row strides so that both
// W[k][lane] (forward) and W[lane][n] (backward) are bank-conflict free.
//
// Sweep 1 walks s_0 -> s_1 -> ... -> s_{L-1} through the dynamics' next-state branch only and keeps the
// L hidden states in LDS (64 ES bytes per sample and step).  Sweep 2 goes back from step L-1 to 0: it
// re-evaluates the step's four heads from the kept state, adds the three cross entropies to the loss and
// back-propagates, halving the state gradient that comes back through the dynamics (Appendix G,
// muax/loss.py:60-61).  Each wavefront writes its partial gradient to a workspace row; a second kernel
// sums the rows in a fixed order (deterministic, no float atomics) and adds the L2 term.
//
// This is a floating-point kernel: its check is torch autograd on the same formula (tests), tolerance
// in the test, not the bit-exact oracle of the search path.
#pragma once
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {


struct TrainParams {
  const float* obs;     // [B][obs_dim]   (batch.obs[:, 0])
  const int32_t* act;   // [B][L]
  const float* rew;     // [B][L]
  const float* ret;     // [B][L]         (n-step returns Rn)
  const float* pi;      // [B][L][A]
  const float* w[18];   // MLP_WEIGHT order: repr_w, repr_b, pv_w1, pv_b1, pv_w2, pv_b2, pp_*, dr_*, dn_*
  int off[19];          // flat offsets of the 18 arrays in the gradient vector; off[18] = total
  int B, L, obs_dim, support;
  float loss_scale;     // 1/B (muax/loss.py) or 1/(B L) (coax variant)
  float l2;             // 1e-4
  float* ws;            // [waves][off[18] + 1] partial gradients + partial loss
  float* grads;         // [off[18]]
  float* loss;          // [1]
  int waves;
};

template <int A_, int E_, int F_>
struct TrainCfg {
  static constexpr int A = A_, E = E_, F = F_, H = 16, X = E_ + A_;
  static constexpr int ES = (E + 15) / 16, FS = (F + 15) / 16, XS = (X + 15) / 16;
  static_assert(A <= 16, "policy head in one slot");
  static constexpr int ld(int n) { return 16 * ((n + 15) / 16) + 1; }  // odd, zero padded
  // LDS weight blocks: [K][ld(N)] then the bias padded to a multiple of 16
  static constexpr int blk(int k, int n) { return k * ld(n) + 16 * ((n + 15) / 16); }
  static constexpr int PV1 = 0;
  static constexpr int PV2 = PV1 + blk(E, H);
  static constexpr int PP1 = PV2 + blk(H, F);
  static constexpr int PP2 = PP1 + blk(E, H);
  static constexpr int DR1 = PP2 + blk(H, A);
  static constexpr int DR2 = DR1 + blk(X, H);
  static constexpr int DN1 = DR2 + blk(H, F);
  static constexpr int DN2 = DN1 + blk(X, H);
  static constexpr int WEIGHT_WORDS = ((DN2 + blk(H, E) + 3) / 4) * 4;
  static constexpr int CK_WORDS_PER_STEP = 16 * 16 * ES;  // 16 samples per workgroup
};

// ---- row-distributed linear layers on LDS weights ----
// y[n] = sum_k x[k] W[k][n] + b[n]
template <int K, int N>
MZ_DEV void lin_fwd(const float (&x)[(K + 15) / 16], const float* W, int j, float (&y)[(N + 15) / 16]) {
  constexpr int NS = (N + 15) / 16, LD = 16 * NS + 1;
#pragma unroll
  for (int t = 0; t < NS; ++t) y[t] = 0.0f;
  StaticFor<0, K>::run([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const float xb = bcast<(k & 15)>(x[k >> 4]);
#pragma unroll
    for (int t = 0; t < NS; ++t) y[t] = __builtin_fmaf(xb, W[k * LD + j + 16 * t], y[t]);
  });
  const float* b = W + K * LD;
#pragma unroll
  for (int t = 0; t < NS; ++t) y[t] = y[t] + b[j + 16 * t];
}
// dx[k] = sum_n dy[n] W[k][n]
template <int K, int N>
MZ_DEV void lin_bwd(const float (&dy)[(N + 15) / 16], const float* W, int j, float (&dx)[(K + 15) / 16]) {
  constexpr int KS = (K + 15) / 16, LD = 16 * ((N + 15) / 16) + 1;
#pragma unroll
  for (int t = 0; t < KS; ++t) dx[t] = 0.0f;
  StaticFor<0, N>::run([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    const float db = bcast<(n & 15)>(dy[n >> 4]);
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const int k = j + 16 * t;
      dx[t] = __builtin_fmaf(db, W[(k < K ? k : 0) * LD + n], dx[t]);
    }
  });
#pragma unroll
  for (int t = 0; t < KS; ++t) dx[t] = (j + 16 * t < K) ? dx[t] : 0.0f;
}
// weight-gradient tiles: acc[kt][nt] += X^T dY over the wave's four samples
template <int KS, int NS>
MZ_DEV void grad_tiles(const float (&x)[KS], const float (&dy)[NS], f32x4 (&acc)[KS][NS]) {
#pragma unroll
  for (int kt = 0; kt < KS; ++kt)
#pragma unroll
    for (int nt = 0; nt < NS; ++nt)
      acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[kt], dy[nt], acc[kt][nt], 0, 0, 0);
}
template <int KS, int NS>
MZ_DEV void zero_tiles(f32x4 (&acc)[KS][NS]) {
#pragma unroll
  for (int kt = 0; kt < KS; ++kt)
#pragma unroll
    for (int nt = 0; nt < NS; ++nt) acc[kt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
}
// D tile layout of v_mfma_f32_16x16x4_f32: register v of lane l holds D[4 (l >> 4) + v][l & 15]
template <int KS, int NS>
MZ_DEV void store_tiles(const f32x4 (&acc)[KS][NS], float* dst, int K, int N, int lane) {
#pragma unroll
  for (int kt = 0; kt < KS; ++kt)
#pragma unroll
    for (int nt = 0; nt < NS; ++nt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = 16 * kt + 4 * (lane >> 4) + v, n = 16 * nt + (lane & 15);
        if (k < K && n < N) dst[k * N + n] = acc[kt][nt][v];
      }
}
// sum of a per-row value over the wave's four rows (lanes of row 0 hold the result afterwards)
MZ_DEV float rows_sum(float v) {
  v = v + __shfl_xor(v, 16);
  v = v + __shfl_xor(v, 32);
  return v;
}
template <int NS>
MZ_DEV void store_bias(const float (&db)[NS], float* dst, int N, int lane) {
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const float s = rows_sum(db[t]);
    if (lane < 16 && lane + 16 * t < N) dst[lane + 16 * t] = s;
  }
}

template <int N>
MZ_DEV void elu_vec(float (&h)[N]) {
#pragma unroll
  for (int t = 0; t < N; ++t) h[t] = elu(h[t]);
}
// d elu / dh from a = elu(h): 1 for h > 0 (a > 0), exp(h) = a + 1 otherwise
MZ_DEV float elu_grad(float a) { return a > 0.0f ? 1.0f : a + 1.0f; }

// softmax probabilities and log-sum-exp of a row-distributed logit vector (lanes >= N: p = 0)
template <int N>
MZ_DEV void softmax_lse(const float (&x)[(N + 15) / 16], int j, float (&p)[(N + 15) / 16], float& lse) {
  constexpr int NS = (N + 15) / 16;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NS; ++t) m = (j + 16 * t < N) ? fmaxf(m, x[t]) : m;
  m = row_max<4>(m);
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    p[t] = (j + 16 * t < N) ? exp_neg(x[t] - m) : 0.0f;
    part = part + p[t];
  }
  const float s = row_sum(part);
#pragma unroll
  for (int t = 0; t < NS; ++t) p[t] = p[t] / s;
  lse = m + log_pos(s);
}
// muax/utils.py:65-67,79-91: two-hot target of a scalar on the 2 S + 1 bins
template <int F>
MZ_DEV void support_target(float x, int support, int j, float (&t)[(F + 15) / 16]) {
  const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
  float xs = sg * (__builtin_sqrtf(__builtin_fabsf(x) + 1.0f) - 1.0f) + 1e-3f * x;
  xs = fminf(fmaxf(xs, -(float)support), (float)support);
  const float lo = __builtin_floorf(xs), hi = __builtin_ceilf(xs);
  const float ph = xs - lo, pl = 1.0f - ph;
  const int ilo = (int)lo + support, ihi = (int)hi + support;
#pragma unroll
  for (int s = 0; s < (F + 15) / 16; ++s) {
    const int k = j + 16 * s;
    t[s] = (k == ilo ? pl : 0.0f) + (k == ihi ? ph : 0.0f);
  }
}
// cross entropy -sum t log_softmax(l) and its logit gradient (p sum(t) - t) * scale
template <int N>
MZ_DEV float ce_and_grad(const float (&l)[(N + 15) / 16], const float (&t)[(N + 15) / 16], int j, float scale,
                         float (&dl)[(N + 15) / 16]) {
  constexpr int NS = (N + 15) / 16;
  float p[NS], lse;
  softmax_lse<N>(l, j, p, lse);
  float pt = 0.0f, ptl = 0.0f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const bool ok = j + 16 * s < N;
    pt = pt + (ok ? t[s] : 0.0f);
    ptl = ptl + (ok ? t[s] * l[s] : 0.0f);
  }
  const float T = row_sum(pt), TL = row_sum(ptl);
#pragma unroll
  for (int s = 0; s < NS; ++s) dl[s] = (j + 16 * s < N) ? (p[s] * T - t[s]) * scale : 0.0f;
  return T * lse - TL;
}

// muax/nn.py:37-44 forward, keeping what the backward needs
template <int E>
MZ_DEV void minmax_fwd(const float (&u)[(E + 15) / 16], int j, float (&s)[(E + 15) / 16], float& mn, float& mx,
                       float& c) {
  constexpr int ES = (E + 15) / 16;
  mn = INFINITY;
  mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < ES; ++t) {
    const bool ok = j + 16 * t < E;
    mn = ok ? fminf(mn, u[t]) : mn;
    mx = ok ? fmaxf(mx, u[t]) : mx;
  }
  mn = row_min<4>(mn);
  mx = row_max<4>(mx);
  c = mx - mn;
  c = c < 1e-5f ? c + 1e-5f : c;
#pragma unroll
  for (int t = 0; t < ES; ++t) s[t] = (j + 16 * t < E) ? (u[t] - mn) / c : 0.0f;
}
// gradient of s = (u - min u) / c, c = max u - min u (+1e-5 when tiny), w.r.t. u; ties of the min / max
// share their gradient evenly (jax's reduce_min / reduce_max rule)
template <int E>
MZ_DEV void minmax_bwd(const float (&g)[(E + 15) / 16], const float (&u)[(E + 15) / 16],
                       const float (&s)[(E + 15) / 16], float mn, float mx, float c, int j,
                       float (&du)[(E + 15) / 16]) {
  constexpr int ES = (E + 15) / 16;
  float pg = 0.0f, pgs = 0.0f, pmin = 0.0f, pmax = 0.0f;
#pragma unroll
  for (int t = 0; t < ES; ++t) {
    const bool ok = j + 16 * t < E;
    pg = pg + (ok ? g[t] : 0.0f);
    pgs = pgs + (ok ? g[t] * s[t] : 0.0f);
    pmin = pmin + ((ok && u[t] == mn) ? 1.0f : 0.0f);
    pmax = pmax + ((ok && u[t] == mx) ? 1.0f : 0.0f);
  }
  const float sg = row_sum(pg), sgs = row_sum(pgs), nmin = row_sum(pmin), nmax = row_sum(pmax);
  const float dmax = -sgs / c;          // through c
  const float dmin = (sgs - sg) / c;    // through the numerator and through c
#pragma unroll
  for (int t = 0; t < ES; ++t) {
    const bool ok = j + 16 * t < E;
    float d = g[t] / c;
    d = d + ((u[t] == mn) ? dmin / nmin : 0.0f);
    d = d + ((u[t] == mx) ? dmax / nmax : 0.0f);
    du[t] = ok ? d : 0.0f;
  }
}

template <class C>
__global__ __launch_bounds__(256) void mz_train_kernel(const TrainParams p) {
  constexpr int A = C::A, E = C::E, F = C::F, H = C::H, X = C::X;
  constexpr int ES = C::ES, FS = C::FS, XS = C::XS;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid & 15;
  const int row = tid >> 4;

  // ---- weights -> LDS, [K][ld(N)] blocks with zero padding, bias behind each block ----
  for (int i = tid; i < C::WEIGHT_WORDS; i += 256) lds[i] = 0.0f;
  __syncthreads();
  {
    const int Ks[8] = {E, H, E, H, X, H, X, H};
    const int Ns[8] = {H, F, H, A, H, F, H, E};
    const int Os[8] = {C::PV1, C::PV2, C::PP1, C::PP2, C::DR1, C::DR2, C::DN1, C::DN2};
    for (int l = 0; l < 8; ++l) {
      const int K = Ks[l], N = Ns[l], LD = 16 * ((N + 15) / 16) + 1;
      const float* W = p.w[2 + 2 * l];
      const float* Bv = p.w[3 + 2 * l];
      for (int i = tid; i < K * N; i += 256) lds[Os[l] + (i / N) * LD + (i % N)] = W[i];
      for (int i = tid; i < N; i += 256) lds[Os[l] + K * LD + i] = Bv[i];
    }
  }
  __syncthreads();
  const float* Wpv1 = lds + C::PV1; const float* Wpv2 = lds + C::PV2;
  const float* Wpp1 = lds + C::PP1; const float* Wpp2 = lds + C::PP2;
  const float* Wdr1 = lds + C::DR1; const float* Wdr2 = lds + C::DR2;
  const float* Wdn1 = lds + C::DN1; const float* Wdn2 = lds + C::DN2;
  float* ck = lds + C::WEIGHT_WORDS + row * 16 * ES;  // + step * CK_WORDS_PER_STEP

  const int L = p.L;
  const int r_raw = blockIdx.x * 16 + row;
  const bool live = r_raw < p.B;
  const int r = live ? r_raw : p.B - 1;
  const float scale = live ? p.loss_scale : 0.0f;  // rows past the batch contribute exact zeros

  // x = [s, onehot(a)] as a row-distributed vector of X elements
  auto make_x = [&](const float (&s)[ES], int a, float (&x)[XS]) {
#pragma unroll
    for (int t = 0; t < XS; ++t) {
      const int k = j + 16 * t;
      float v = 0.0f;
      if (t < ES) v = (k < E) ? s[t] : 0.0f;
      x[t] = (k >= E && k < X) ? ((k - E == a) ? 1.0f : 0.0f) : v;
    }
  };

  // ---- sweep 1: hidden states s_0 .. s_{L-1} ----
  float u0[ES], s0mn, s0mx, s0c;  // representation pre-activation, kept for its backward
  {
#pragma unroll
    for (int t = 0; t < ES; ++t) {
      const int k = j + 16 * t;
      float acc = 0.0f;
      if (k < E) {
        for (int i = 0; i < p.obs_dim; ++i) acc = __builtin_fmaf(p.obs[(size_t)r * p.obs_dim + i], p.w[0][i * E + k], acc);
        acc = acc + p.w[1][k];
      }
      u0[t] = acc;
    }
    float s[ES];
    minmax_fwd<E>(u0, j, s, s0mn, s0mx, s0c);
#pragma unroll
    for (int t = 0; t < ES; ++t) ck[j + 16 * t] = s[t];
    for (int i = 0; i + 1 < L; ++i) {
      float x[XS], hn[1], u[ES], mn, mx, c;
      make_x(s, p.act[(size_t)r * L + i], x);
      lin_fwd<X, H>(x, Wdn1, j, hn);
      elu_vec(hn);
      lin_fwd<H, E>(hn, Wdn2, j, u);
      minmax_fwd<E>(u, j, s, mn, mx, c);
#pragma unroll
      for (int t = 0; t < ES; ++t) ck[(i + 1) * C::CK_WORDS_PER_STEP + j + 16 * t] = s[t];
    }
  }

  // ---- sweep 2: heads, loss and gradients, last step first ----
  f32x4 g_pv1[ES][1], g_pv2[1][FS], g_pp1[ES][1], g_pp2[1][1], g_dr1[XS][1], g_dr2[1][FS], g_dn1[XS][1], g_dn2[1][ES];
  zero_tiles(g_pv1); zero_tiles(g_pv2); zero_tiles(g_pp1); zero_tiles(g_pp2);
  zero_tiles(g_dr1); zero_tiles(g_dr2); zero_tiles(g_dn1); zero_tiles(g_dn2);
  float b_pv1[1] = {0.0f}, b_pp1[1] = {0.0f}, b_pp2[1] = {0.0f}, b_dr1[1] = {0.0f}, b_dn1[1] = {0.0f};
  float b_pv2[FS], b_dr2[FS], b_dn2[ES];
#pragma unroll
  for (int t = 0; t < FS; ++t) b_pv2[t] = b_dr2[t] = 0.0f;
#pragma unroll
  for (int t = 0; t < ES; ++t) b_dn2[t] = 0.0f;
  float loss = 0.0f;
  float ds_next[ES];
#pragma unroll
  for (int t = 0; t < ES; ++t) ds_next[t] = 0.0f;

  for (int i = L - 1; i >= 0; --i) {
    float s[ES], x[XS];
#pragma unroll
    for (int t = 0; t < ES; ++t) s[t] = ck[i * C::CK_WORDS_PER_STEP + j + 16 * t];
    const int a = p.act[(size_t)r * L + i];
    make_x(s, a, x);
    float ds[ES];
#pragma unroll
    for (int t = 0; t < ES; ++t) ds[t] = 0.0f;

    // -- dynamics: next-state branch (gradient arrives from step i + 1) and reward head --
    float dx[XS];
#pragma unroll
    for (int t = 0; t < XS; ++t) dx[t] = 0.0f;
    if (i + 1 < L) {
      float an[1], u[ES], ns[ES], mn, mx, c, du[ES], dan[1], dxn[XS];
      lin_fwd<X, H>(x, Wdn1, j, an);
      elu_vec(an);
      lin_fwd<H, E>(an, Wdn2, j, u);
      minmax_fwd<E>(u, j, ns, mn, mx, c);
      minmax_bwd<E>(ds_next, u, ns, mn, mx, c, j, du);
      grad_tiles(an, du, g_dn2);
#pragma unroll
      for (int t = 0; t < ES; ++t) b_dn2[t] = b_dn2[t] + du[t];
      lin_bwd<H, E>(du, Wdn2, j, dan);
      dan[0] = dan[0] * elu_grad(an[0]);
      grad_tiles(x, dan, g_dn1);
      b_dn1[0] = b_dn1[0] + dan[0];
      lin_bwd<X, H>(dan, Wdn1, j, dxn);
#pragma unroll
      for (int t = 0; t < XS; ++t) dx[t] = dx[t] + dxn[t];
    }
    {
      float ar[1], lr[FS], tr[FS], dlr[FS], dar[1], dxr[XS];
      lin_fwd<X, H>(x, Wdr1, j, ar);
      elu_vec(ar);
      lin_fwd<H, F>(ar, Wdr2, j, lr);
      support_target<F>(p.rew[(size_t)r * L + i], p.support, j, tr);
      loss = loss + ce_and_grad<F>(lr, tr, j, scale, dlr);
      grad_tiles(ar, dlr, g_dr2);
#pragma unroll
      for (int t = 0; t < FS; ++t) b_dr2[t] = b_dr2[t] + dlr[t];
      lin_bwd<H, F>(dlr, Wdr2, j, dar);
      dar[0] = dar[0] * elu_grad(ar[0]);
      grad_tiles(x, dar, g_dr1);
      b_dr1[0] = b_dr1[0] + dar[0];
      lin_bwd<X, H>(dar, Wdr1, j, dxr);
#pragma unroll
      for (int t = 0; t < XS; ++t) dx[t] = dx[t] + dxr[t];
    }
    // scale_gradient(s, 0.5) in front of the dynamics (muax/loss.py:60-61)
#pragma unroll
    for (int t = 0; t < ES; ++t) ds[t] = (j + 16 * t < E) ? 0.5f * dx[t] : 0.0f;

    // -- prediction heads on s_i --
    {
      float av[1], lv[FS], tv[FS], dlv[FS], dav[1], dsv[ES];
      lin_fwd<E, H>(s, Wpv1, j, av);
      elu_vec(av);
      lin_fwd<H, F>(av, Wpv2, j, lv);
      support_target<F>(p.ret[(size_t)r * L + i], p.support, j, tv);
      loss = loss + ce_and_grad<F>(lv, tv, j, scale, dlv);
      grad_tiles(av, dlv, g_pv2);
#pragma unroll
      for (int t = 0; t < FS; ++t) b_pv2[t] = b_pv2[t] + dlv[t];
      lin_bwd<H, F>(dlv, Wpv2, j, dav);
      dav[0] = dav[0] * elu_grad(av[0]);
      grad_tiles(s, dav, g_pv1);
      b_pv1[0] = b_pv1[0] + dav[0];
      lin_bwd<E, H>(dav, Wpv1, j, dsv);
#pragma unroll
      for (int t = 0; t < ES; ++t) ds[t] = ds[t] + dsv[t];
    }
    {
      float ap[1], lp[1], tp[1], dlp[1], dap[1], dsp[ES];
      lin_fwd<E, H>(s, Wpp1, j, ap);
      elu_vec(ap);
      lin_fwd<H, A>(ap, Wpp2, j, lp);
      tp[0] = j < A ? p.pi[((size_t)r * L + i) * A + j] : 0.0f;
      loss = loss + ce_and_grad<A>(lp, tp, j, scale, dlp);
      grad_tiles(ap, dlp, g_pp2);
      b_pp2[0] = b_pp2[0] + dlp[0];
      lin_bwd<H, A>(dlp, Wpp2, j, dap);
      dap[0] = dap[0] * elu_grad(ap[0]);
      grad_tiles(s, dap, g_pp1);
      b_pp1[0] = b_pp1[0] + dap[0];
      lin_bwd<E, H>(dap, Wpp1, j, dsp);
#pragma unroll
      for (int t = 0; t < ES; ++t) ds[t] = ds[t] + dsp[t];
    }
#pragma unroll
    for (int t = 0; t < ES; ++t) ds_next[t] = ds[t];
  }

  // ---- representation: s_0 = minmax(obs W + b) ----
  float du0[ES];
  {
    float s[ES];
#pragma unroll
    for (int t = 0; t < ES; ++t) s[t] = ck[j + 16 * t];
    minmax_bwd<E>(ds_next, u0, s, s0mn, s0mx, s0c, j, du0);
  }
  f32x4 g_rep[1][ES];
  zero_tiles(g_rep);
  {
    float ob[1];
    ob[0] = j < p.obs_dim ? p.obs[(size_t)r * p.obs_dim + j] : 0.0f;
    grad_tiles(ob, du0, g_rep);
  }

  // ---- this wavefront's partial gradient -> its workspace row ----
  float* dst = p.ws + (size_t)(blockIdx.x * 4 + (tid >> 6)) * (p.off[18] + 1);
  store_tiles(g_rep, dst + p.off[0], p.obs_dim, E, lane);
  store_bias(du0, dst + p.off[1], E, lane);
  store_tiles(g_pv1, dst + p.off[2], E, H, lane);  store_bias(b_pv1, dst + p.off[3], H, lane);
  store_tiles(g_pv2, dst + p.off[4], H, F, lane);  store_bias(b_pv2, dst + p.off[5], F, lane);
  store_tiles(g_pp1, dst + p.off[6], E, H, lane);  store_bias(b_pp1, dst + p.off[7], H, lane);
  store_tiles(g_pp2, dst + p.off[8], H, A, lane);  store_bias(b_pp2, dst + p.off[9], A, lane);
  store_tiles(g_dr1, dst + p.off[10], X, H, lane); store_bias(b_dr1, dst + p.off[11], H, lane);
  store_tiles(g_dr2, dst + p.off[12], H, F, lane); store_bias(b_dr2, dst + p.off[13], F, lane);
  store_tiles(g_dn1, dst + p.off[14], X, H, lane); store_bias(b_dn1, dst + p.off[15], H, lane);
  store_tiles(g_dn2, dst + p.off[16], H, E, lane); store_bias(b_dn2, dst + p.off[17], E, lane);
  const float wl = rows_sum(loss * scale);
  if (lane == 0) dst[p.off[18]] = wl;
}

// Fixed-order sum of the per-wavefront partials, plus the L2 term 1e-4 * 0.5 * sum w^2 (muax/loss.py:84-87).
__global__ __launch_bounds__(256) void mz_train_reduce_kernel(const TrainParams p) {
  // block = 32 gradient entries x 8 groups of workspace rows; partials meet in LDS in a fixed order
  const int NP = p.off[18];
  __shared__ float red[256];
  const int pl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + pl;
  const int per = (p.waves + 7) / 8;
  const int w0 = q * per, w1 = min(p.waves, w0 + per);
  float acc = 0.0f;
  if (idx < NP) {
    const float* src = p.ws + idx;
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(w + u) * (NP + 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; w < w1; ++w) acc = acc + src[(size_t)w * (NP + 1)];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (q == 0 && idx < NP) {
    float t = red[pl];
#pragma unroll
    for (int g = 1; g < 8; ++g) t = t + red[32 * g + pl];
    int a = 0;
    while (idx >= p.off[a + 1]) ++a;
    p.grads[idx] = t + p.l2 * p.w[a][idx - p.off[a]];
  }
  if (blockIdx.x == gridDim.x - 1) {  // loss: partial sums of every wavefront + 0.5 l2 sum w^2
    __syncthreads();
    float la = 0.0f;
    for (int w = threadIdx.x; w < p.waves; w += 256) la = la + p.ws[(size_t)w * (NP + 1) + NP];
    float sq = 0.0f;
    for (int a = 0; a < 18; ++a)
      for (int i = threadIdx.x; i < p.off[a + 1] - p.off[a]; i += 256) sq = __builtin_fmaf(p.w[a][i], p.w[a][i], sq);
    red[threadIdx.x] = la + 0.5f * p.l2 * sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) p.loss[0] = red[0];
  }
}

}  // namespace mz
