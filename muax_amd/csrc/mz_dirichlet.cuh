// mz_dirichlet.cuh -- mctx.muzero_policy's root exploration noise on the device:
//   jax.random.dirichlet(dirichlet_rng_key, alpha = full([A], dirichlet_alpha), shape = (B,))
// (the policy call site muax/policy.py:18-30, defaults muax/model.py:92-93), restated from the published algorithm
// of jax/_src/random.py (jax 0.4.x): softmax over loggamma draws, one key per element = split(key, B A)[i],
// _gamma_one(log_space = True) = Marsaglia-Tsang rejection with per-iteration key splits (oracle/mz_oracle.c holds
// the same restatement line by line and the caveat: SPEC-TO-CONFIRM against a real jax; act(dirichlet_noise = ...)
// stays the bit-pinned route).  Four lanes per (root, action) element walk the rejection loop speculatively
// (loggamma_group), then one thread per root does the softmax.  Rows are indexed by the GLOBAL root, so a shard draws exactly its rows of the whole batch.
#pragma once
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

MZ_DEV uint32_t jax_bits(uint32_t k0, uint32_t k1, uint64_t size, uint64_t i) {
  uint32_t x0, x1;
  bool second;
  bits_block(size, i, x0, x1, second);
  threefry2x32(k0, k1, x0, x1);
  return second ? x1 : x0;
}
MZ_DEV void jax_split(uint32_t k0, uint32_t k1, uint64_t n, uint64_t row, uint32_t& o0, uint32_t& o1) {
  o0 = jax_bits(k0, k1, 2 * n, 2 * row);
  o1 = jax_bits(k0, k1, 2 * n, 2 * row + 1);
}
MZ_DEV float log1p_f(float x) {  // x > -1
  const float u = 1.0f + x;
  return u == 1.0f ? x : log_pos(u) * (x / (u - 1.0f));
}
MZ_DEV float erf_inv_f(float x) {  // Giles, single precision (XLA's ErfInv32)
  float w = -log1p_f(-(x * x));
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
MZ_DEV float jax_uniform(uint32_t k0, uint32_t k1, float minval, float maxval) {
  const float f = uniform_from_bits(jax_bits(k0, k1, 1, 0));
  const float u = f * (maxval - minval) + minval;
  return u > minval ? u : minval;
}
MZ_DEV float jax_normal(uint32_t k0, uint32_t k1) {
  return 1.41421354f * erf_inv_f(jax_uniform(k0, k1, -0.99999994f, 1.0f));
}
// One candidate of the rejection loop from its keys: x from the first normal with 1 + x c > 0 (inner loop of
// _gamma_one), X = x^2, V = (1 + x c)^3, U = uniform(u-key)
MZ_DEV void gamma_candidate(uint32_t x0, uint32_t x1, uint32_t u0, uint32_t u1, float c, float& X, float& V, float& U) {
  float x = 0.0f, v = -1.0f;
  for (int g2 = 0; g2 < 1000 && v <= 0.0f; ++g2) {
    uint32_t a0, a1, b0, b1;
    jax_split(x0, x1, 2, 0, a0, a1);
    jax_split(x0, x1, 2, 1, b0, b1);
    x0 = a0; x1 = a1;
    x = jax_normal(b0, b1);
    v = 1.0f + x * c;
  }
  X = x * x;
  V = (v * v) * v;
  U = jax_uniform(u0, u1, 0.0f, 1.0f);
}
// the loop condition of _gamma_one: true = reject the candidate and draw another one
MZ_DEV bool gamma_rejects(float X, float V, float U, float d) {
  const float SQUEEZE = 0.0331f;
  const float logU = U > 0.0f ? log_pos(U) : -INFINITY;
  return (U >= 1.0f - SQUEEZE * (X * X)) && (logU >= X * 0.5f + d * ((1.0f - V) + log_pos(V)));
}
// jax's loggamma draw of ONE element by the kSpec = 4 lanes of its group (consecutive lanes of a wavefront).  The
// rejection loop is a key chain k_0 -> k_1 -> ... (one split per iteration) with iteration t's candidate hanging off
// k_t; a lone thread walks it until a candidate is accepted, and a wavefront of such threads waits for its slowest
// lane.  Here lane t of the group walks t links and evaluates candidate t at once with the others; the first accepted
// one (lowest t) is what the serial loop would have returned, value for value.  If all four are rejected (probability
// ~6e-6 per element) the group continues the serial loop from k_4 with candidate 3 as the loop state.
constexpr int kSpec = 4;
MZ_DEV float loggamma_group(uint32_t ki0, uint32_t ki1, float alpha_orig, int t, int lane) {
  const float THIRD = 0.333333343f;
  const bool boost_mask = alpha_orig >= 1.0f;
  const float alpha = boost_mask ? alpha_orig : alpha_orig + 1.0f;
  const float d = alpha - THIRD;
  const float c = THIRD / sqrtf(d);
  uint32_t k0, k1, s0, s1;
  jax_split(ki0, ki1, 2, 0, k0, k1);
  jax_split(ki0, ki1, 2, 1, s0, s1);
  uint32_t x0 = 0, x1 = 0, u0 = 0, u1 = 0;
  for (int lvl = 0; lvl < kSpec; ++lvl) {  // (the first iteration always runs: the loop starts from X = 0, V = 1, U = 2)
    if (lvl <= t) {
      uint32_t n0, n1;
      jax_split(k0, k1, 3, 0, n0, n1);
      jax_split(k0, k1, 3, 1, x0, x1);
      jax_split(k0, k1, 3, 2, u0, u1);
      k0 = n0; k1 = n1;  // lane t ends with k_{t+1}
    }
  }
  float X, V, U;
  gamma_candidate(x0, x1, u0, u1, c, X, V, U);
  const bool accepted = !gamma_rejects(X, V, U, d);
  const int base = lane & ~(kSpec - 1);
  const unsigned hits = (unsigned)((__ballot(accepted) >> base) & ((1u << kSpec) - 1u));
  if (hits != 0) {
    V = __shfl(V, base + __builtin_ctz(hits));
  } else {
    // the serial loop from the last lane's state (k_4 and the rejected candidate 3), run by the whole group
    const int last = base + kSpec - 1;
    k0 = (uint32_t)__shfl((int)k0, last); k1 = (uint32_t)__shfl((int)k1, last);
    X = __shfl(X, last); V = __shfl(V, last); U = __shfl(U, last);
    for (int guard = kSpec; guard < 1000 && gamma_rejects(X, V, U, d); ++guard) {
      uint32_t n0, n1;
      jax_split(k0, k1, 3, 0, n0, n1);
      jax_split(k0, k1, 3, 1, x0, x1);
      jax_split(k0, k1, 3, 2, u0, u1);
      k0 = n0; k1 = n1;
      gamma_candidate(x0, x1, u0, u1, c, X, V, U);
    }
  }
  const float log_samples = log1p_f(-jax_uniform(s0, s1, 0.0f, 1.0f));
  const float log_boost = (boost_mask || log_samples == 0.0f) ? 0.0f : log_samples * (1.0f / alpha_orig);
  return (log_pos(d) + log_pos(V)) + log_boost;
}

// block = 256 threads = R = 64 / A roots x A actions x kSpec lanes (A <= 64); dynamic LDS: R * A floats
__global__ __launch_bounds__(256) void dirichlet_kernel(uint32_t k0, uint32_t k1, float alpha, int B, int A,
                                                        uint64_t global_batch, uint64_t root_offset, float* out) {
  extern __shared__ float lg[];
  const int R = (256 / kSpec) / A;
  const int tid = threadIdx.x;
  const int el = tid / kSpec, t = tid % kSpec;  // element (root, action) of this block and the lane's iteration
  const int rb = el / A, a = el - rb * A;
  const int b = blockIdx.x * R + rb;
  // (idle groups shadow element 0 of the block: every lane of a wavefront takes part in the ballot / shuffles)
  const bool live = rb < R && b < B;
  const uint64_t row = live ? (root_offset + (uint64_t)b) * (uint64_t)A + (uint64_t)a
                            : (root_offset + (uint64_t)(blockIdx.x * R)) * (uint64_t)A;
  uint32_t e0, e1;
  jax_split(k0, k1, global_batch * (uint64_t)A, row, e0, e1);
  const float v = loggamma_group(e0, e1, alpha, t, tid & 63);
  if (live && t == 0) lg[rb * A + a] = v;
  __syncthreads();
  const int b2 = blockIdx.x * R + tid;
  if (tid < R && b2 < B) {
    float* rowp = lg + tid * A;
    float mx = rowp[0];
    for (int i = 1; i < A; ++i) mx = rowp[i] > mx ? rowp[i] : mx;
    float sum = 0.0f;
    for (int i = 0; i < A; ++i) {
      const float e = exp_neg(rowp[i] - mx);
      rowp[i] = e;
      sum = i == 0 ? e : sum + e;
    }
    for (int i = 0; i < A; ++i) out[(size_t)b2 * A + i] = rowp[i] / sum;
  }
}

}  // namespace mz
