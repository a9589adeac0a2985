/*
 * mz_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See mz_oracle.h for scope, provenance and the "parity unpinned" statement.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off, explicit fmaf only where
 * the arithmetic spec says so).  Every float op below is an IEEE-754 binary32
 * round-to-nearest-even +,-,*,/,sqrt or fma, so results are reproducible on any
 * conforming machine; transcendental functions are spelled out (mzo_exp,
 * mzo_log) instead of calling libm so that no libm version enters the result.
 */
#include "mz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== */
/* Arithmetic spec "MZ-F32"                                               */
/* ===================================================================== */

static inline float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp core: x = k*ln2 + r, |r| <= ln2/2; returns q = exp(r)-1 (degree-7 Taylor,
 * Horner in fmaf) and k.  Valid for x in [-87, 88]. */
static inline float exp_core(float x, int *k_out) {
  const float LOG2E = 1.44269504088896341f;
  const float LN2_HI = 6.93145752e-1f;   /* 0x3f317200 */
  const float LN2_LO = 1.42860677e-6f;   /* 0x35bfbe8e */
  float kf = rintf(x * LOG2E);
  float r = fmaf(kf, -LN2_HI, x);
  r = fmaf(kf, -LN2_LO, r);
  float p = 1.0f / 5040.0f;
  p = fmaf(p, r, 1.0f / 720.0f);
  p = fmaf(p, r, 1.0f / 120.0f);
  p = fmaf(p, r, 1.0f / 24.0f);
  p = fmaf(p, r, 1.0f / 6.0f);
  p = fmaf(p, r, 0.5f);
  float rr = r * r;
  *k_out = (int)kf;
  return fmaf(p, rr, r);
}

static inline float pow2i(int k) { return f32_from_bits((uint32_t)(k + 127) << 23); }

/* exp(x) for x <= 88; exact 0 below -87 (the softmax/ELU callers only pass x <= 0). */
float mzo_exp(float x) {
  if (x < -87.0f) return 0.0f;
  int k;
  float q = exp_core(x, &k);
  return (1.0f + q) * pow2i(k);
}

/* expm1(x) for x <= 0 (the ELU branch, jax.nn.elu; muax/nn.py:78,82,98,102). */
float mzo_expm1_neg(float x) {
  if (x < -87.0f) return -1.0f;
  int k;
  float q = exp_core(x, &k);
  if (k == 0) return q;
  return (1.0f + q) * pow2i(k) - 1.0f;
}

float mzo_elu(float x) { return x > 0.0f ? x : mzo_expm1_neg(x); }

/* log(x), x > 0 finite normal (callers clamp to FLT_MIN).  Classic argument
 * reduction x = 2^e * m, m in [sqrt(.5), sqrt(2)), s = f/(2+f). */
float mzo_log(float x) {
  const float LN2_HI = 6.9313812256e-01f; /* 0x3f317180 */
  const float LN2_LO = 9.0580006145e-06f; /* 0x3717f7d1 */
  const float LG1 = 0.66666662693f;       /* 0xaaaaaa.0p-24 */
  const float LG2 = 0.40000972152f;       /* 0xccce13.0p-25 */
  const float LG3 = 0.28498786688f;       /* 0x91e9ee.0p-25 */
  const float LG4 = 0.24279078841f;       /* 0xf89e26.0p-26 */
  uint32_t ix = f32_bits(x);
  ix += 0x3f800000u - 0x3f3504f3u;
  int e = (int)(ix >> 23) - 127;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  float m = f32_from_bits(ix);
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

/* Canonical sum: 16 partials p_l = x_l + x_{l+16} + x_{l+32} + ... (ascending),
 * then an xor butterfly over 1,2,4,8.  This is the one reduction order every
 * float sum on the path uses (softmax denominators, support decode). */
float mzo_sum16(const float *x, int n) {
  float p[16];
  for (int l = 0; l < 16; ++l) {
    float acc = 0.0f;
    int first = 1;
    for (int i = l; i < n; i += 16) {
      if (first) { acc = x[i]; first = 0; } else { acc = acc + x[i]; }
    }
    p[l] = acc;
  }
  for (int m = 1; m < 16; m <<= 1) {
    float q[16];
    for (int l = 0; l < 16; ++l) q[l] = p[l] + p[l ^ m];
    memcpy(p, q, sizeof p);
  }
  return p[0];
}

/* jax.nn.softmax: exp(x - max) / sum. */
void mzo_softmax(const float *x, int n, float *p) {
  float mx = x[0];
  for (int i = 1; i < n; ++i) mx = x[i] > mx ? x[i] : mx;
  float e[256];
  for (int i = 0; i < n; ++i) e[i] = mzo_exp(x[i] - mx);
  float s = mzo_sum16(e, n);
  for (int i = 0; i < n; ++i) p[i] = e[i] / s;
}

/* muax/utils.py:70-76 _inv_scaling, eps = 1e-3. */
float mzo_inv_scaling(float x) {
  const float EPS = 0.001f, FOUR_EPS = 0.004f, TWO_EPS = 0.002f;
  float ax = fabsf(x);
  float a = (ax + 1.0f) + EPS;
  float b = FOUR_EPS * a;
  float c = 1.0f + b;
  float d = sqrtf(c);
  float e = (d - 1.0f) / TWO_EPS;
  float g = e * e - 1.0f;
  float sgn = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
  return sgn * g;
}

/* muax/utils.py:94-102 support_to_scalar. */
float mzo_support_to_scalar(const float *probs, int support_size) {
  int F = 2 * support_size + 1;
  float t[256];
  for (int k = 0; k < F; ++k) t[k] = (float)(k - support_size) * probs[k];
  return mzo_inv_scaling(mzo_sum16(t, F));
}

/* muax/nn.py:37-44 min_max_normalize over the feature axis. */
void mzo_min_max_normalize(float *s, int n) {
  float mn = s[0], mx = s[0];
  for (int i = 1; i < n; ++i) {
    mn = s[i] < mn ? s[i] : mn;
    mx = s[i] > mx ? s[i] : mx;
  }
  float scale = mx - mn;
  if (scale < 1e-5f) scale = scale + 1e-5f;
  for (int i = 0; i < n; ++i) s[i] = (s[i] - mn) / scale;
}

/* ===================================================================== */
/* JAX PRNG (threefry2x32; jax/_src/prng.py, non-partitionable stream)    */
/* ===================================================================== */

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

void mzo_threefry2x32(const uint32_t key[2], uint32_t x0, uint32_t x1, uint32_t out[2]) {
  static const int R0[4] = {13, 15, 26, 6}, R1[4] = {17, 29, 16, 24};
  uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    const int *R = (g & 1) ? R1 : R0;
    for (int i = 0; i < 4; ++i) {
      x0 += x1;
      x1 = rotl32(x1, R[i]);
      x1 ^= x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0;
  out[1] = x1;
}

/* flat[i] of threefry_2x32(key, iota(size)): size padded to even, first half
 * of the counters hashed against the second half. */
uint32_t mzo_random_bits(const uint32_t key[2], int64_t size, int64_t i) {
  int64_t half = (size + 1) / 2;
  int64_t blk = i < half ? i : i - half;
  int64_t c1 = half + blk;
  uint32_t x1 = (c1 < size) ? (uint32_t)c1 : 0u; /* odd size: zero pad */
  uint32_t out[2];
  mzo_threefry2x32(key, (uint32_t)blk, x1, out);
  return i < half ? out[0] : out[1];
}

/* jax.random.split(key, n)[row] */
void mzo_split(const uint32_t key[2], int64_t n, int64_t row, uint32_t out[2]) {
  out[0] = mzo_random_bits(key, 2 * n, 2 * row);
  out[1] = mzo_random_bits(key, 2 * n, 2 * row + 1);
}

/* jax.random.uniform float32 in [0,1): mantissa fill. */
float mzo_uniform_from_bits(uint32_t bits) {
  return f32_from_bits((bits >> 9) | 0x3f800000u) - 1.0f;
}

/* jax.random.gumbel: -log(-log(uniform(minval=tiny, maxval=1))) */
float mzo_gumbel_from_bits(uint32_t bits) {
  const float TINY = 1.17549435e-38f;
  float u = mzo_uniform_from_bits(bits);
  u = u * (1.0f - TINY) + TINY; /* (1 - tiny) rounds to 1 */
  u = u > TINY ? u : TINY;
  return -mzo_log(-mzo_log(u));
}

/* ===================================================================== */
/* jax.random.dirichlet restated                                          */
/* ===================================================================== */
/* mctx.muzero_policy draws its root noise with
 *   jax.random.dirichlet(dirichlet_rng_key, alpha = full([A], dirichlet_alpha), shape = (B,))
 * (call site of the policy: muax/policy.py:18-30; defaults muax/model.py:92-93).  Restated from the
 * PUBLISHED algorithm of jax/_src/random.py (jax 0.4.x): _dirichlet = softmax(loggamma(key, alpha, shape + [A])),
 * _gamma_impl: one key per element = split(key, B * A)[i], _gamma_one(key, alpha, log_space = True):
 * Marsaglia-Tsang rejection with per-iteration key splits, alpha < 1 boosted to alpha + 1 and corrected in log
 * space with log1p(-uniform(subkey)) / alpha; normal = sqrt(2) erf_inv(uniform(lo = nextafter(-1, 0), 1)) with
 * XLA's single-precision erf_inv (Giles' polynomials).
 * SPEC-TO-CONFIRM: jax is not installable here, so neither the op order nor the last bits (XLA's own log /
 * log1p / erf_inv polynomials) can be checked against a real jax; the integer key walk is exact threefry.  The
 * tested, bit-pinned path for "same noise as the reference" remains act(dirichlet_noise = <array from jax>). */
float mzo_log1p(float x) { /* x > -1: log(1 + x) with the rounding error of (1 + x) compensated */
  float u = 1.0f + x;
  if (u == 1.0f) return x;
  return mzo_log(u) * (x / (u - 1.0f));
}

float mzo_erf_inv(float x) { /* |x| < 1; M. Giles, "Approximating the erfinv function", single precision */
  float w = -mzo_log1p(-(x * x));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = 3.43273939e-07f + p * w;
    p = -3.5233877e-06f + p * w;
    p = -4.39150654e-06f + p * w;
    p = 0.00021858087f + p * w;
    p = -0.00125372503f + p * w;
    p = -0.00417768164f + p * w;
    p = 0.246640727f + p * w;
    p = 1.50140941f + p * w;
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = 0.000100950558f + p * w;
    p = 0.00134934322f + p * w;
    p = -0.00367342844f + p * w;
    p = 0.00573950773f + p * w;
    p = -0.0076224613f + p * w;
    p = 0.00943887047f + p * w;
    p = 1.00167406f + p * w;
    p = 2.83297682f + p * w;
  }
  return p * x;
}

/* jax.random.uniform(key, (), float32, minval, maxval): max(minval, floats * (maxval - minval) + minval) */
static float jax_uniform(const uint32_t key[2], float minval, float maxval) {
  float f = mzo_uniform_from_bits(mzo_random_bits(key, 1, 0));
  float u = f * (maxval - minval) + minval;
  return u > minval ? u : minval;
}
/* jax.random.normal(key, ()) */
static float jax_normal(const uint32_t key[2]) {
  const float LO = -0.99999994f; /* nextafter(-1, 0) */
  return 1.41421354f * mzo_erf_inv(jax_uniform(key, LO, 1.0f));
}

/* exported for the known-answer tests: jax.random.normal(key, ()) and jax.random.normal(key, (n,)) */
float mzo_normal(const uint32_t key[2]) { return jax_normal(key); }
void mzo_normal_vec(const uint32_t key[2], int64_t n, float *out) {
  const float LO = -0.99999994f;
  for (int64_t i = 0; i < n; ++i) {
    const float f = mzo_uniform_from_bits(mzo_random_bits(key, n, i));
    const float u = f * (1.0f - LO) + LO;
    out[i] = 1.41421354f * mzo_erf_inv(u > LO ? u : LO);
  }
}

/* jax _gamma_one(key, alpha, log_space = True): log of a Gamma(alpha, 1) draw */
float mzo_loggamma_one(const uint32_t key_in[2], float alpha_orig) {
  const float THIRD = 0.333333343f, SQUEEZE = 0.0331f;
  const int boost_mask = alpha_orig >= 1.0f;
  const float alpha = boost_mask ? alpha_orig : alpha_orig + 1.0f;
  const float d = alpha - THIRD;
  const float c = THIRD / sqrtf(d);
  uint32_t key[2], subkey[2];
  mzo_split(key_in, 2, 0, key);
  mzo_split(key_in, 2, 1, subkey);
  float X = 0.0f, V = 1.0f, U = 2.0f;
  for (int guard = 0; guard < 1000; ++guard) {
    /* _cond_fn: keep looping while the candidate is REJECTED (the initial state always is) */
    const float logU = U > 0.0f ? mzo_log(U) : -INFINITY;
    const int cond = (U >= 1.0f - SQUEEZE * (X * X)) && (logU >= X * 0.5f + d * ((1.0f - V) + mzo_log(V)));
    if (!cond) break;
    uint32_t nk[2], x_key[2], u_key[2];
    mzo_split(key, 3, 0, nk);
    mzo_split(key, 3, 1, x_key);
    mzo_split(key, 3, 2, u_key);
    key[0] = nk[0]; key[1] = nk[1];
    float x = 0.0f, v = -1.0f;
    for (int g2 = 0; g2 < 1000 && v <= 0.0f; ++g2) {
      uint32_t k2[2], sub[2];
      mzo_split(x_key, 2, 0, k2);
      mzo_split(x_key, 2, 1, sub);
      x_key[0] = k2[0]; x_key[1] = k2[1];
      x = jax_normal(sub);
      v = 1.0f + x * c;
    }
    X = x * x;
    V = (v * v) * v;
    U = jax_uniform(u_key, 0.0f, 1.0f);
  }
  const float log_samples = mzo_log1p(-jax_uniform(subkey, 0.0f, 1.0f)); /* -exponential(subkey) */
  const float log_boost = (boost_mask || log_samples == 0.0f) ? 0.0f : log_samples * (1.0f / alpha_orig);
  return (mzo_log(d) + mzo_log(V)) + log_boost;
}

/* rows [root_offset, root_offset + B) of jax.random.dirichlet(key, full([A], alpha), (global_batch,)) */
void mzo_dirichlet(const uint32_t key[2], float alpha, int B, int A, int64_t global_batch,
                   int64_t root_offset, float *out) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    float lg[256], mx = 0.0f, sum = 0.0f;
    for (int a = 0; a < A; ++a) {
      uint32_t ek[2];
      mzo_split(key, global_batch * A, (root_offset + b) * A + a, ek);
      lg[a] = mzo_loggamma_one(ek, alpha);
      mx = (a == 0 || lg[a] > mx) ? lg[a] : mx;
    }
    /* jax.nn.softmax: exp(x - max) / sum, summed in action order */
    for (int a = 0; a < A; ++a) {
      lg[a] = mzo_exp(lg[a] - mx);
      sum = a == 0 ? lg[a] : sum + lg[a];
    }
    for (int a = 0; a < A; ++a) out[(int64_t)b * A + a] = lg[a] / sum;
  }
}

/* ===================================================================== */
/* Default MLP trio (muax/nn.py:59-115)                                   */
/* ===================================================================== */

/* haiku Linear: dot (k-ordered fma chain from 0) then + bias. */
static void linear(const float *x, int n_in, const float *w, const float *b, int n_out,
                   float *y) {
  for (int j = 0; j < n_out; ++j) {
    float acc = 0.0f;
    for (int i = 0; i < n_in; ++i) acc = fmaf(x[i], w[i * n_out + j], acc);
    y[j] = acc + b[j];
  }
}

static void mlp2(const float *x, int n_in, const float *w1, const float *b1, int H,
                 const float *w2, const float *b2, int n_out, float *y) {
  float h[64];
  linear(x, n_in, w1, b1, H, h);
  for (int j = 0; j < H; ++j) h[j] = mzo_elu(h[j]);
  linear(h, H, w2, b2, n_out, y);
}

static void prediction(const mzo_mlp *m, const float *s, float *prior_logits, float *value) {
  float v_logits[256], v_probs[256];
  mlp2(s, m->E, m->pv_w1, m->pv_b1, m->H, m->pv_w2, m->pv_b2, m->F, v_logits);
  mlp2(s, m->E, m->pp_w1, m->pp_b1, m->H, m->pp_w2, m->pp_b2, m->A, prior_logits);
  mzo_softmax(v_logits, m->F, v_probs);
  *value = mzo_support_to_scalar(v_probs, m->support_size);
}

/* muax/model.py:251-263 _root_inference */
void mzo_root_inference(const mzo_mlp *m, const float *obs, float *embedding,
                        float *prior_logits, float *value) {
  linear(obs, m->obs_dim, m->repr_w, m->repr_b, m->E, embedding);
  mzo_min_max_normalize(embedding, m->E);
  prediction(m, embedding, prior_logits, value);
}

/* muax/model.py:265-282 _recurrent_inference; Dynamic at muax/nn.py:93-115. */
void mzo_recurrent_inference(const mzo_mlp *m, int action, const float *embedding,
                             float *reward, float *discount, float *prior_logits,
                             float *value, float *next_embedding) {
  float sa[512], r_logits[256], r_probs[256];
  int n_in = m->E + m->A;
  for (int i = 0; i < m->E; ++i) sa[i] = embedding[i];
  for (int k = 0; k < m->A; ++k) sa[m->E + k] = (k == action) ? 1.0f : 0.0f;
  mlp2(sa, n_in, m->dr_w1, m->dr_b1, m->H, m->dr_w2, m->dr_b2, m->F, r_logits);
  mlp2(sa, n_in, m->dn_w1, m->dn_b1, m->H, m->dn_w2, m->dn_b2, m->E, next_embedding);
  mzo_min_max_normalize(next_embedding, m->E);
  prediction(m, m->recurrent_pred_on == 1 ? embedding : next_embedding, prior_logits, value);
  mzo_softmax(r_logits, m->F, r_probs);
  *reward = mzo_support_to_scalar(r_probs, m->support_size);
  *discount = m->discount;
}

/* ===================================================================== */
/* mctx search restated                                                   */
/* ===================================================================== */

#define FLT_TINY 1.17549435e-38f
#define FLT_LOWEST (-3.40282347e+38f)

/* mctx policies.muzero_policy prelude: _add_dirichlet_noise,
 * _get_logits_from_probs, _mask_invalid_actions. */
void mzo_root_prior(const float *prior_logits, int A, const float *dirichlet_noise,
                    float dirichlet_fraction, const uint8_t *invalid, float *out_logits) {
  float p[256];
  mzo_softmax(prior_logits, A, p);
  float keep = 1.0f - dirichlet_fraction;
  for (int a = 0; a < A; ++a) {
    float nz = dirichlet_noise ? dirichlet_noise[a] : 0.0f;
    float noisy = keep * p[a] + dirichlet_fraction * nz;
    float cl = noisy > FLT_TINY ? noisy : FLT_TINY;
    out_logits[a] = mzo_log(cl);
  }
  if (invalid) {
    float mx = out_logits[0];
    for (int a = 1; a < A; ++a) mx = out_logits[a] > mx ? out_logits[a] : mx;
    for (int a = 0; a < A; ++a)
      out_logits[a] = invalid[a] ? FLT_LOWEST : out_logits[a] - mx;
  }
}

/* mctx search.instantiate_tree_from_root + update_tree_node for the root. */
void mzo_tree_init(mzo_tree *t, const float *prior_logits, const float *value,
                   const float *embedding, const uint8_t *invalid) {
  int B = t->B, N = t->N, A = t->A, E = t->E;
  for (int64_t i = 0; i < (int64_t)B * N; ++i) {
    t->node_visits[i] = 0;
    t->raw_values[i] = 0.0f;
    t->node_values[i] = 0.0f;
    t->parents[i] = MZO_NO_PARENT;
    t->action_from_parent[i] = MZO_NO_PARENT;
  }
  for (int64_t i = 0; i < (int64_t)B * N * A; ++i) {
    t->children_index[i] = MZO_UNVISITED;
    t->children_prior_logits[i] = 0.0f;
    t->children_values[i] = 0.0f;
    t->children_visits[i] = 0;
    t->children_rewards[i] = 0.0f;
    t->children_discounts[i] = 0.0f;
  }
  memset(t->embeddings, 0, sizeof(float) * (size_t)B * N * E);
  for (int b = 0; b < B; ++b) {
    int64_t n0 = (int64_t)b * N;
    for (int a = 0; a < A; ++a) {
      t->children_prior_logits[n0 * A + a] = prior_logits[(int64_t)b * A + a];
      t->root_invalid_actions[(int64_t)b * A + a] = invalid ? invalid[(int64_t)b * A + a] : 0;
    }
    t->raw_values[n0] = value[b];
    t->node_values[n0] = value[b];
    t->node_visits[n0] = 1;
    memcpy(t->embeddings + n0 * E, embedding + (int64_t)b * E, sizeof(float) * E);
  }
}

/* mctx action_selection.muzero_action_selection with
 * qtransforms.qtransform_by_parent_and_siblings; noise[A] already scaled by
 * 1e-7 (or NULL). */
/* value_score[a], policy_score[a] of muzero_action_selection at `node` (the two terms mzo_select_action adds): exposed
   so that tools/triage_capture.py can print the oracle's own arithmetic for a disputed decision. */
void mzo_action_scores(const mzo_tree *t, int b, int node, const mzo_search_cfg *cfg,
                       float *value_score, float *policy_score) {
  int A = t->A;
  int64_t n = (int64_t)b * t->N + node;
  const int32_t *vc = t->children_visits + n * A;
  const float *cr = t->children_rewards + n * A;
  const float *cd = t->children_discounts + n * A;
  const float *cv = t->children_values + n * A;
  int32_t nv = t->node_visits[n];
  float nval = t->node_values[n];

  float num = ((float)nv + cfg->pb_c_base) + 1.0f;
  float pb_c = cfg->pb_c_init + mzo_log(num / cfg->pb_c_base);
  float tn = sqrtf((float)nv) * pb_c;
  float prior[256];
  mzo_softmax(t->children_prior_logits + n * A, A, prior);

  float q[256];
  float lo = nval, hi = nval;
  for (int a = 0; a < A; ++a) {
    q[a] = cr[a] + cd[a] * cv[a];
    float safe = vc[a] > 0 ? q[a] : nval;
    lo = safe < lo ? safe : lo;
    hi = safe > hi ? safe : hi;
  }
  float span = hi - lo;
  span = span > 1e-8f ? span : 1e-8f;
  for (int a = 0; a < A; ++a) {
    float cbm = vc[a] > 0 ? q[a] : lo;
    value_score[a] = (cbm - lo) / span;
    policy_score[a] = (tn * prior[a]) / (float)(vc[a] + 1);
  }
}

int mzo_select_action(const mzo_tree *t, int b, int node, int depth,
                      const mzo_search_cfg *cfg, const float *noise) {
  int A = t->A;
  float value_score[256], policy_score[256];
  mzo_action_scores(t, b, node, cfg, value_score, policy_score);
  int best = 0;
  float best_score = 0.0f;
  for (int a = 0; a < A; ++a) {
    float score = value_score[a] + policy_score[a];
    if (noise) score = score + noise[a];
    if (depth == 0 && t->root_invalid_actions[(int64_t)b * A + a]) score = -INFINITY;
    if (a == 0 || score > best_score) { best = a; best_score = score; }
  }
  return best;
}

/* mctx search.simulate for root b.  root_key = simulate_keys[b].  `uniforms` ([D][A], may be NULL): the tie-break
   uniforms of levels d < D INJECTED instead of drawn (a capture's rng_tiebreak[s][b]): the search pinned independently
   of the threefry walk.  The key walk advances at every level either way; levels >= D draw from it. */
void mzo_simulate_injected(const mzo_tree *t, int b, const mzo_search_cfg *cfg,
                           const uint32_t root_key[2], const float *uniforms, int D,
                           int32_t *parent_out, int32_t *action_out, int32_t *depth_out) {
  int A = t->A;
  int max_depth = cfg->max_depth > 0 ? cfg->max_depth : cfg->num_simulations;
  uint32_t key[2] = {0, 0};
  if (cfg->tiebreak) { key[0] = root_key[0]; key[1] = root_key[1]; }
  int node = 0, depth = 0, action = 0, parent = 0;
  for (;;) {
    float noise_buf[256];
    const float *noise = NULL;
    if (cfg->tiebreak) {
      uint32_t nk[2], sel[2];
      mzo_split(key, 2, 0, nk);
      mzo_split(key, 2, 1, sel);
      key[0] = nk[0];
      key[1] = nk[1];
      for (int a = 0; a < A; ++a) {
        float u = (uniforms && depth < D) ? uniforms[(int64_t)depth * A + a]
                                          : mzo_uniform_from_bits(mzo_random_bits(sel, A, a));
        noise_buf[a] = 1e-7f * u;
      }
      noise = noise_buf;
    }
    action = mzo_select_action(t, b, node, depth, cfg, noise);
    parent = node;
    int next = t->children_index[((int64_t)b * t->N + node) * A + action];
    depth += 1;
    if (next == MZO_UNVISITED || depth >= max_depth) break;
    node = next;
  }
  *parent_out = parent;
  *action_out = action;
  if (depth_out) *depth_out = depth;
}

void mzo_simulate(const mzo_tree *t, int b, const mzo_search_cfg *cfg,
                  const uint32_t root_key[2], int32_t *parent_out, int32_t *action_out,
                  int32_t *depth_out) {
  mzo_simulate_injected(t, b, cfg, root_key, NULL, 0, parent_out, action_out, depth_out);
}

/* mctx search.expand (+ update_tree_node) for root b. */
void mzo_expand(mzo_tree *t, int b, int parent, int action, int next, float reward,
                float discount, const float *prior_logits, float value,
                const float *next_embedding) {
  int N = t->N, A = t->A, E = t->E;
  int64_t nn = (int64_t)b * N + next, pn = (int64_t)b * N + parent;
  for (int a = 0; a < A; ++a) t->children_prior_logits[nn * A + a] = prior_logits[a];
  t->raw_values[nn] = value;
  t->node_values[nn] = value;
  t->node_visits[nn] = t->node_visits[nn] + 1;
  memcpy(t->embeddings + nn * E, next_embedding, sizeof(float) * E);
  t->children_index[pn * A + action] = next;
  t->children_rewards[pn * A + action] = reward;
  t->children_discounts[pn * A + action] = discount;
  t->parents[nn] = parent;
  t->action_from_parent[nn] = action;
}

/* mctx search.backward for root b. */
void mzo_backward(mzo_tree *t, int b, int leaf) {
  int N = t->N, A = t->A;
  int64_t base = (int64_t)b * N;
  float leaf_value = t->node_values[base + leaf];
  int idx = leaf;
  while (idx != 0) {
    int parent = t->parents[base + idx];
    int32_t count = t->node_visits[base + parent];
    int action = t->action_from_parent[base + idx];
    int64_t e = (base + parent) * A + action;
    leaf_value = t->children_rewards[e] + t->children_discounts[e] * leaf_value;
    float parent_value =
        (t->node_values[base + parent] * (float)count + leaf_value) / ((float)count + 1.0f);
    t->node_values[base + parent] = parent_value;
    t->node_visits[base + parent] = count + 1;
    t->children_values[e] = t->node_values[base + idx];
    t->children_visits[e] = t->children_visits[e] + 1;
    idx = parent;
  }
}

/* mctx Tree.summary + policies._apply_temperature + jax.random.categorical. */
void mzo_summary_sample(const mzo_tree *t, int b, float temperature, const float *gumbel,
                        int32_t *action_out, float *action_weights_out) {
  int A = t->A;
  const int32_t *vc = t->children_visits + (int64_t)b * t->N * A;
  float total = 0.0f;
  for (int a = 0; a < A; ++a) total = total + (float)vc[a];
  float denom = total > 1.0f ? total : 1.0f;
  float logits[256];
  float mx = 0.0f;
  for (int a = 0; a < A; ++a) {
    float p = (float)vc[a] / denom;
    if (!(total > 0.0f)) p = 1.0f / (float)A;
    action_weights_out[a] = p;
    float cl = p > FLT_TINY ? p : FLT_TINY;
    logits[a] = mzo_log(cl);
    if (a == 0 || logits[a] > mx) mx = logits[a];
  }
  float tden = temperature > FLT_TINY ? temperature : FLT_TINY;
  int best = 0;
  float best_score = 0.0f;
  for (int a = 0; a < A; ++a) {
    float l = (logits[a] - mx) / tden;
    float score = l + gumbel[a];
    if (a == 0 || score > best_score) { best = a; best_score = score; }
  }
  *action_out = best;
}

/* ===================================================================== */
/* Gumbel MuZero (mctx gumbel_muzero_policy)                              */
/* ===================================================================== */

/* mctx qtransforms: per-child transformed Q of `node`. */
void mzo_qtransform(const mzo_tree *t, int b, int node, int qtransform, float *out) {
  int A = t->A;
  int64_t n = (int64_t)b * t->N + node;
  const int32_t *vc = t->children_visits + n * A;
  float q[256];
  for (int a = 0; a < A; ++a)
    q[a] = t->children_rewards[n * A + a] + t->children_discounts[n * A + a] * t->children_values[n * A + a];
  if (qtransform == 0) {
    /* qtransform_by_parent_and_siblings */
    float nval = t->node_values[n];
    float lo = nval, hi = nval;
    for (int a = 0; a < A; ++a) {
      float safe = vc[a] > 0 ? q[a] : nval;
      lo = safe < lo ? safe : lo;
      hi = safe > hi ? safe : hi;
    }
    float span = hi - lo;
    span = span > 1e-8f ? span : 1e-8f;
    for (int a = 0; a < A; ++a) out[a] = ((vc[a] > 0 ? q[a] : lo) - lo) / span;
    return;
  }
  /* qtransform_completed_by_mix_value(value_scale=0.1, maxvisit_init=50, rescale_values, use_mixed_value) */
  float prior[256], tmp[256];
  mzo_softmax(t->children_prior_logits + n * A, A, prior);
  int32_t sum_visits = 0, maxvisit = 0;
  for (int a = 0; a < A; ++a) {
    sum_visits += vc[a];
    maxvisit = vc[a] > maxvisit ? vc[a] : maxvisit;
    prior[a] = prior[a] > FLT_TINY ? prior[a] : FLT_TINY;
  }
  for (int a = 0; a < A; ++a) tmp[a] = vc[a] > 0 ? prior[a] : 0.0f;
  float sum_probs = mzo_sum16(tmp, A);
  for (int a = 0; a < A; ++a)
    tmp[a] = vc[a] > 0 ? (prior[a] * q[a]) / (vc[a] > 0 ? sum_probs : 1.0f) : 0.0f;
  float weighted_q = mzo_sum16(tmp, A);
  float value = (t->raw_values[n] + (float)sum_visits * weighted_q) / (float)(sum_visits + 1);
  float lo = 0.0f, hi = 0.0f;
  for (int a = 0; a < A; ++a) {
    out[a] = vc[a] > 0 ? q[a] : value;
    if (a == 0 || out[a] < lo) lo = out[a];
    if (a == 0 || out[a] > hi) hi = out[a];
  }
  float span = hi - lo;
  span = span > 1e-8f ? span : 1e-8f;
  float scale = (50.0f + (float)maxvisit) * 0.1f;
  for (int a = 0; a < A; ++a) out[a] = scale * ((out[a] - lo) / span);
}

/* mctx seq_halving.get_sequence_of_considered_visits */
void mzo_considered_visits(int m, int num_simulations, int32_t *seq) {
  if (m <= 1) {
    for (int i = 0; i < num_simulations; ++i) seq[i] = i;
    return;
  }
  int log2max = 0;
  while ((1 << log2max) < m) ++log2max;
  int32_t visits[256];
  for (int i = 0; i < m; ++i) visits[i] = 0;
  int n = 0, num_considered = m;
  while (n < num_simulations) {
    int extra = num_simulations / (log2max * num_considered);
    if (extra < 1) extra = 1;
    for (int e = 0; e < extra; ++e) {
      for (int i = 0; i < num_considered && n < num_simulations; ++i) seq[n++] = visits[i];
      for (int i = 0; i < num_considered; ++i) visits[i] += 1;
    }
    num_considered = num_considered / 2 > 2 ? num_considered / 2 : 2;
  }
}

/* seq_halving.score_considered + masked_argmax */
static int gumbel_argmax(int A, int considered_visit, const float *gumbel, const float *logits,
                         const float *qv, const int32_t *vc, const uint8_t *invalid) {
  float mx = logits[0];
  for (int a = 1; a < A; ++a) mx = logits[a] > mx ? logits[a] : mx;
  int best = 0;
  float bs = 0.0f;
  for (int a = 0; a < A; ++a) {
    float s = (gumbel[a] + (logits[a] - mx)) + qv[a];
    s = s > -1e9f ? s : -1e9f;
    s = s + (vc[a] == considered_visit ? 0.0f : -INFINITY);
    if (invalid && invalid[a]) s = -INFINITY;
    if (a == 0 || s > bs) { best = a; bs = s; }
  }
  return best;
}

int mzo_gumbel_select_action(const mzo_tree *t, int b, int node, int depth, int qtransform,
                             const float *root_gumbel, int num_simulations,
                             int max_num_considered_actions) {
  int A = t->A;
  int64_t n = (int64_t)b * t->N + node;
  const int32_t *vc = t->children_visits + n * A;
  float qv[256];
  mzo_qtransform(t, b, node, qtransform, qv);
  if (depth == 0) {
    /* gumbel_muzero_root_action_selection */
    const uint8_t *inv = t->root_invalid_actions + (int64_t)b * A;
    int num_valid = 0, sim_index = 0;
    for (int a = 0; a < A; ++a) { num_valid += inv[a] ? 0 : 1; sim_index += vc[a]; }
    int num_considered = max_num_considered_actions < num_valid ? max_num_considered_actions : num_valid;
    int32_t *seq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_simulations > 0 ? num_simulations : 1));
    mzo_considered_visits(num_considered, num_simulations, seq);
    int considered_visit = seq[sim_index < num_simulations ? sim_index : num_simulations - 1];
    free(seq);
    return gumbel_argmax(A, considered_visit, root_gumbel + (int64_t)b * A,
                         t->children_prior_logits + n * A, qv, vc, inv);
  }
  /* gumbel_muzero_interior_action_selection: argmax(softmax(logits + q) - visits / (1 + sum visits)) */
  float x[256], p[256];
  int32_t sum = 0;
  for (int a = 0; a < A; ++a) { x[a] = t->children_prior_logits[n * A + a] + qv[a]; sum += vc[a]; }
  mzo_softmax(x, A, p);
  int best = 0;
  float bs = 0.0f;
  for (int a = 0; a < A; ++a) {
    float s = p[a] - (float)vc[a] / (float)(1 + sum);
    if (a == 0 || s > bs) { best = a; bs = s; }
  }
  return best;
}

void mzo_gumbel_step_select(const mzo_tree *t, const mzo_search_cfg *cfg, int qtransform,
                            const float *root_gumbel, int max_num_considered_actions,
                            int32_t *parent_out, int32_t *action_out, int32_t *depth_out) {
  int A = t->A;
  int max_depth = cfg->max_depth > 0 ? cfg->max_depth : cfg->num_simulations;
  for (int b = 0; b < t->B; ++b) {
    int node = 0, depth = 0, action = 0, parent = 0;
    for (;;) {
      action = mzo_gumbel_select_action(t, b, node, depth, qtransform, root_gumbel,
                                        cfg->num_simulations, max_num_considered_actions);
      parent = node;
      int next = t->children_index[((int64_t)b * t->N + node) * A + action];
      depth += 1;
      if (next == MZO_UNVISITED || depth >= max_depth) break;
      node = next;
    }
    parent_out[b] = parent;
    action_out[b] = action;
    if (depth_out) depth_out[b] = depth;
  }
}

/* tail of mctx gumbel_muzero_policy: best considered action, completed-Q policy target */
void mzo_gumbel_finish(const mzo_tree *t, int b, int qtransform, const float *root_gumbel,
                       const float *root_logits, int32_t *action_out, float *action_weights_out) {
  int A = t->A;
  const int32_t *vc = t->children_visits + (int64_t)b * t->N * A;
  const uint8_t *inv = t->root_invalid_actions + (int64_t)b * A;
  (void)root_logits;
  const float *logits = t->children_prior_logits + (int64_t)b * t->N * A;
  int considered_visit = 0, any_invalid = 0;
  for (int a = 0; a < A; ++a) {
    considered_visit = vc[a] > considered_visit ? vc[a] : considered_visit;
    any_invalid |= inv[a];
  }
  float qv[256], x[256];
  mzo_qtransform(t, b, 0, qtransform, qv);
  *action_out = gumbel_argmax(A, considered_visit, root_gumbel + (int64_t)b * A, logits, qv, vc, inv);
  /* completed_search_logits = _mask_invalid_actions(prior_logits + completed_qvalues, invalid_actions) */
  float mx = 0.0f;
  for (int a = 0; a < A; ++a) {
    x[a] = logits[a] + qv[a];
    if (a == 0 || x[a] > mx) mx = x[a];
  }
  if (any_invalid)
    for (int a = 0; a < A; ++a) x[a] = inv[a] ? FLT_LOWEST : x[a] - mx;
  mzo_softmax(x, A, action_weights_out);
}

/* ---- stepwise driver ---- */

void mzo_step_select(const mzo_tree *t, const mzo_search_cfg *cfg, int sim,
                     const uint32_t sim_key[2], int32_t *parent_out,
                     int32_t *action_out, int32_t *depth_out) {
  (void)sim;
  for (int b = 0; b < t->B; ++b) {
    uint32_t rk[2] = {0, 0};
    if (cfg->tiebreak) mzo_split(sim_key, cfg->global_batch, cfg->root_offset + b, rk);
    int32_t d;
    mzo_simulate(t, b, cfg, rk, parent_out + b, action_out + b, &d);
    if (depth_out) depth_out[b] = d;
  }
}

/* mzo_step_select with the tie-break uniforms of this simulation injected: uniforms [B][D][A] (may be NULL). */
void mzo_step_select_injected(const mzo_tree *t, const mzo_search_cfg *cfg, int sim,
                              const uint32_t sim_key[2], const float *uniforms, int D,
                              int32_t *parent_out, int32_t *action_out, int32_t *depth_out) {
  (void)sim;
  for (int b = 0; b < t->B; ++b) {
    uint32_t rk[2] = {0, 0};
    if (cfg->tiebreak) mzo_split(sim_key, cfg->global_batch, cfg->root_offset + b, rk);
    int32_t d;
    mzo_simulate_injected(t, b, cfg, rk, uniforms ? uniforms + (int64_t)b * D * t->A : NULL, D,
                          parent_out + b, action_out + b, &d);
    if (depth_out) depth_out[b] = d;
  }
}

void mzo_step_expand_backup(mzo_tree *t, int sim, const int32_t *parent,
                            const int32_t *action, const float *reward,
                            const float *discount, const float *prior_logits,
                            const float *value, const float *next_embedding) {
  int A = t->A, E = t->E;
  for (int b = 0; b < t->B; ++b) {
    int next = t->children_index[((int64_t)b * t->N + parent[b]) * A + action[b]];
    if (next == MZO_UNVISITED) next = sim + 1;
    mzo_expand(t, b, parent[b], action[b], next, reward[b], discount[b],
               prior_logits + (int64_t)b * A, value[b], next_embedding + (int64_t)b * E);
    mzo_backward(t, b, next);
  }
}

/* ---- whole act() ---- */

void mzo_act_mlp(const mzo_mlp *m, const mzo_search_cfg *cfg, mzo_tree *t,
                 const float *obs, const uint32_t key[2],
                 const float *dirichlet_noise, float dirichlet_fraction,
                 const uint8_t *invalid_actions, float temperature,
                 const float *gumbel, int32_t *action_out, float *action_weights_out,
                 float *root_value_out, int64_t *depth_sum_out, int nthreads) {
  int B = t->B, A = t->A, E = t->E, S = cfg->num_simulations;
  /* mctx muzero_policy: rng_key, dirichlet_rng_key, search_rng_key = split(rng_key, 3) */
  uint32_t k_sample[2], k_search[2];
  mzo_split(key, 3, 0, k_sample);
  mzo_split(key, 3, 2, k_search);
  /* search body_fun: rng_key, simulate_key, expand_key = split(rng_key, 3) */
  uint32_t *sim_keys = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)(S > 0 ? S : 1));
  {
    uint32_t rk[2] = {k_search[0], k_search[1]};
    for (int s = 0; s < S; ++s) {
      uint32_t nk[2];
      mzo_split(rk, 3, 1, sim_keys + 2 * s);
      mzo_split(rk, 3, 0, nk);
      rk[0] = nk[0];
      rk[1] = nk[1];
    }
  }

  const int N = t->N;
  /* Roots never interact, so the simulation loop is run
   * root-major here; mctx runs it simulation-major over the whole batch. */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 1 ? nthreads : 1)
#endif
  for (int b = 0; b < B; ++b) {
    /* root inference + muzero_policy prelude + instantiate_tree_from_root, one root */
    {
      float raw_logits[256], pl0[256], emb0[1024], v0;
      mzo_root_inference(m, obs + (int64_t)b * m->obs_dim, emb0, raw_logits, &v0);
      root_value_out[b] = v0;
      mzo_root_prior(raw_logits, A, dirichlet_noise ? dirichlet_noise + (int64_t)b * A : NULL,
                     dirichlet_fraction, invalid_actions ? invalid_actions + (int64_t)b * A : NULL,
                     pl0);
      mzo_tree t1 = *t;
      int64_t n0 = (int64_t)b * N;
      t1.B = 1;
      t1.node_visits += n0; t1.raw_values += n0; t1.node_values += n0;
      t1.parents += n0; t1.action_from_parent += n0;
      t1.children_index += n0 * A; t1.children_prior_logits += n0 * A;
      t1.children_values += n0 * A; t1.children_visits += n0 * A;
      t1.children_rewards += n0 * A; t1.children_discounts += n0 * A;
      t1.embeddings += n0 * E; t1.root_invalid_actions += (int64_t)b * A;
      mzo_tree_init(&t1, pl0, &v0, emb0, invalid_actions ? invalid_actions + (int64_t)b * A : NULL);
    }
    int64_t dsum = 0;
    for (int s = 0; s < S; ++s) {
      uint32_t rk[2] = {0, 0};
      if (cfg->tiebreak) mzo_split(sim_keys + 2 * s, cfg->global_batch, cfg->root_offset + b, rk);
      int32_t parent, action, depth;
      mzo_simulate(t, b, cfg, rk, &parent, &action, &depth);
      dsum += depth;
      int next = t->children_index[((int64_t)b * t->N + parent) * A + action];
      if (next == MZO_UNVISITED) next = s + 1;
      float reward, discount, value, pl[256], ne[1024];
      mzo_recurrent_inference(m, action, t->embeddings + ((int64_t)b * t->N + parent) * E,
                              &reward, &discount, pl, &value, ne);
      mzo_expand(t, b, parent, action, next, reward, discount, pl, value, ne);
      mzo_backward(t, b, next);
    }
    if (depth_sum_out) depth_sum_out[b] = dsum;
    float g[256];
    if (gumbel) {
      for (int a = 0; a < A; ++a) g[a] = gumbel[(int64_t)b * A + a];
    } else {
      int64_t gb = cfg->root_offset + b;
      for (int a = 0; a < A; ++a)
        g[a] = mzo_gumbel_from_bits(
            mzo_random_bits(k_sample, cfg->global_batch * A, gb * A + a));
    }
    mzo_summary_sample(t, b, temperature, g, action_out + b,
                       action_weights_out + (int64_t)b * A);
  }
  free(sim_keys);
}

/* see mz_oracle.h: the HIP kernels divide by 2 eps = 0.002f with y = RN(1 / 0.002f), q0 = x y, r = fma(-q0, 0.002f, x),
 * q = fma(r, y, q0); count the binary32 x (every mantissa, exponents e_lo .. e_hi biased) for which q != x / 0.002f */
int64_t mzo_div2eps_mismatches(int e_lo, int e_hi) {
  const float c = 0.002f, y = 1.0f / c;
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic)
  for (int e = e_lo; e <= e_hi; ++e)
    for (uint32_t m = 0; m < (1u << 23); ++m) {
      const float x = f32_from_bits(((uint32_t)e << 23) | m);
      const float q0 = x * y;
      const float r = fmaf(-q0, c, x);
      if (fmaf(r, y, q0) != x / c) ++bad;
    }
  return bad;
}
/* elu(x) for x <= -87 through the clamped path only (no select): must be exactly -1 */
float mzo_elu_clamped(float x) {
  float xn = x < 0.0f ? x : 0.0f;
  int k;
  float q = exp_core(xn > -87.0f ? xn : -87.0f, &k);
  float em1 = (k == 0) ? q : (1.0f + q) * pow2i(k) - 1.0f;
  return x > 0.0f ? x : em1;
}

/* see mz_oracle.h: exhaustive check of the small-integer Markstein division used by the HIP kernel */
int64_t mzo_markstein_mismatches(int dmax, int exponent) {
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic)
  for (int d = 1; d <= dmax; ++d) {
    const float fd = (float)d;
    volatile float yv = 1.0f / fd;
    const float y = yv;
    for (uint32_t m = 0; m < (1u << 23); ++m) {
      uint32_t bits = ((uint32_t)(127 + exponent) << 23) | m;
      for (int sgn = 0; sgn < 2; ++sgn) {
        uint32_t b2 = bits | ((uint32_t)sgn << 31);
        float x;
        memcpy(&x, &b2, 4);
        const float q0 = x * y;
        const float r = fmaf(-q0, fd, x);
        const float q = fmaf(r, y, q0);
        bad += (q != x / fd);
      }
    }
  }
  return bad;
}
