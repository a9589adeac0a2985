// mz_step.cuh -- step-wise search for ARBITRARY plugin nets (the repr_fn /
// pred_fn / dy_fn surface of muax/model.py:52-54): the caller runs its own
// networks between mzs_select and mzs_expand_backup, exactly where mctx calls
// recurrent_fn (mctx search.expand).  The tree lives in HBM in mctx's own
// layout so that PolicyOutput.search_tree is a plain copy.  Same row mapping
// and arithmetic spec as the fused kernel; A and E are run-time (A <= 64).
#pragma once
#include "../../include/mzsearch.h"
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

constexpr int kMaxAS = 4;  // action slots per lane (A <= 64)

struct StepArgs {
  int32_t B, N, A, E, S, max_depth, tiebreak;
  float pb_c_init, pb_c_base;
  uint64_t global_batch, root_offset;
  int32_t* node_visits; float* raw_values; float* node_values;
  int32_t* parents; int32_t* action_from_parent;
  int32_t* children_index; float* children_prior_logits; float* children_prior_probs;
  float* children_values; int32_t* children_visits; float* children_rewards;
  float* children_discounts; float* embeddings;
  uint8_t* root_invalid;
  int32_t *sel_parent, *sel_action, *sel_depth, *depth_sum, *path;
  // wide embeddings (E >= kWideEmb): the tree kernels only note WHICH node's row moves (xfer_node); the
  // rows themselves are moved by emb_gather / emb_scatter with a whole workgroup per KiB instead of the
  // 16 lanes that own the root
  int32_t* xfer_node;
  int32_t wide;
  const uint32_t* sim_keys;
  // gumbel policy
  int32_t qtransform, max_considered;
  float gumbel_scale;
  float* root_gumbel;           // [B, A]
  const int32_t* visit_table;   // [(max_considered + 1), S] seq_halving.get_table_of_considered_visits
};

constexpr int kWideEmb = 256;

struct StepState {
  bool allocated = false, rooted = false;
  int32_t* node_visits = nullptr; float* raw_values = nullptr; float* node_values = nullptr;
  int32_t* parents = nullptr; int32_t* action_from_parent = nullptr;
  int32_t* children_index = nullptr; float* children_prior_logits = nullptr;
  float* children_prior_probs = nullptr; float* children_values = nullptr;
  int32_t* children_visits = nullptr; float* children_rewards = nullptr;
  float* children_discounts = nullptr; float* embeddings = nullptr;
  uint8_t* root_invalid = nullptr;
  int32_t *sel_parent = nullptr, *sel_action = nullptr, *sel_depth = nullptr, *depth_sum = nullptr,
          *path = nullptr;
  uint32_t* sim_keys = nullptr;
  int32_t* xfer_node = nullptr;
  float* root_gumbel = nullptr;
  int32_t* visit_table = nullptr;
  void* slab = nullptr;

  // one slab, carved: the only allocation the handle ever makes
  hipError_t allocate(int B, int N, int A, int E, int table_words) {
    size_t BN = (size_t)B * N;
    size_t words = 5 * BN + 7 * BN * A + BN * E + 4 * (size_t)B + BN + 2 * (size_t)N + (size_t)B * A +
                   (size_t)table_words;
    words += (size_t)B;  // xfer_node
    size_t bytes = words * 4 + (size_t)B * A + 256;
    hipError_t e = hipMalloc(&slab, bytes);
    if (e != hipSuccess) return e;
    uint32_t* w = static_cast<uint32_t*>(slab);
    auto take = [&](size_t n) { uint32_t* p = w; w += n; return p; };
    node_visits = (int32_t*)take(BN); raw_values = (float*)take(BN); node_values = (float*)take(BN);
    parents = (int32_t*)take(BN); action_from_parent = (int32_t*)take(BN);
    children_index = (int32_t*)take(BN * A); children_prior_logits = (float*)take(BN * A);
    children_prior_probs = (float*)take(BN * A); children_values = (float*)take(BN * A);
    children_visits = (int32_t*)take(BN * A); children_rewards = (float*)take(BN * A);
    children_discounts = (float*)take(BN * A); embeddings = (float*)take(BN * E);
    sel_parent = (int32_t*)take(B); sel_action = (int32_t*)take(B); sel_depth = (int32_t*)take(B);
    depth_sum = (int32_t*)take(B); path = (int32_t*)take(BN); sim_keys = take(2 * (size_t)N);
    xfer_node = (int32_t*)take(B);
    root_gumbel = (float*)take((size_t)B * A); visit_table = (int32_t*)take(table_words);
    root_invalid = reinterpret_cast<uint8_t*>(w);
    allocated = true;
    return hipSuccess;
  }
  void release() {
    if (slab) hipFree(slab);
    slab = nullptr;
    allocated = rooted = false;
  }
  StepArgs args(const mzs_config& c) const {
    StepArgs a;
    a.B = c.batch; a.N = c.num_simulations + 1; a.A = c.num_actions; a.E = c.embed_dim;
    a.S = c.num_simulations; a.max_depth = c.max_depth > 0 ? c.max_depth : c.num_simulations;
    a.tiebreak = c.tiebreak; a.pb_c_init = c.pb_c_init; a.pb_c_base = c.pb_c_base;
    a.global_batch = (uint64_t)c.global_batch; a.root_offset = (uint64_t)c.root_offset;
    a.node_visits = node_visits; a.raw_values = raw_values; a.node_values = node_values;
    a.parents = parents; a.action_from_parent = action_from_parent;
    a.children_index = children_index; a.children_prior_logits = children_prior_logits;
    a.children_prior_probs = children_prior_probs; a.children_values = children_values;
    a.children_visits = children_visits; a.children_rewards = children_rewards;
    a.children_discounts = children_discounts; a.embeddings = embeddings;
    a.root_invalid = root_invalid;
    a.sel_parent = sel_parent; a.sel_action = sel_action; a.sel_depth = sel_depth;
    a.depth_sum = depth_sum; a.path = path; a.sim_keys = sim_keys;
    a.xfer_node = xfer_node; a.wide = c.embed_dim >= kWideEmb ? 1 : 0;
    a.qtransform = c.qtransform; a.max_considered = c.max_num_considered_actions;
    a.gumbel_scale = c.gumbel_scale; a.root_gumbel = root_gumbel; a.visit_table = visit_table;
    return a;
  }
};

// canonical softmax over a row-distributed vector with run-time length A
MZ_DEV void row_softmax_rt(const float (&x)[kMaxAS], int A, int j, float (&p)[kMaxAS]) {
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) m = (j + 16 * t < A) ? fmaxf(m, x[t]) : m;
  m = row_max<4>(m);
  float e[kMaxAS];
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    bool ok = j + 16 * t < A;
    e[t] = ok ? exp_neg(x[t] - m) : 0.0f;
    part = (t == 0) ? e[0] : (ok ? part + e[t] : part);
  }
  float s = row_sum(part);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) p[t] = e[t] / s;
}

#define MZ_ROW_SETUP                                              \
  const int lane = threadIdx.x & 63;                              \
  const int j = lane & 15;                                        \
  const int r = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);  \
  if (r >= s.B) return;                                           \
  const int N = s.N, A = s.A, E = s.E;                            \
  const size_t rb = (size_t)r * N;

// mctx instantiate_tree_from_root + muzero_policy prelude (dirichlet, mask)
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void step_root_kernel(StepArgs s, const float* prior_logits,
                                                         const float* value, const float* embedding,
                                                         const uint8_t* invalid, const float* noise,
                                                         float fraction, int gumbel_policy,
                                                         const float* gumbel_in, uint32_t gk0, uint32_t gk1) {
  MZ_ROW_SETUP
  for (int n = j; n < N; n += 16) {
    s.node_visits[rb + n] = 0;
    s.raw_values[rb + n] = 0.0f;
    s.node_values[rb + n] = 0.0f;
    s.parents[rb + n] = -1;
    s.action_from_parent[rb + n] = -1;
  }
  for (int i = j; i < N * A; i += 16) {
    size_t o = rb * A + i;
    s.children_index[o] = -1;
    s.children_prior_logits[o] = 0.0f;
    s.children_prior_probs[o] = 0.0f;
    s.children_values[o] = 0.0f;
    s.children_visits[o] = 0;
    s.children_rewards[o] = 0.0f;
    s.children_discounts[o] = 0.0f;
  }
  if (!s.wide)  // (wide: zeroed by a memset on the stream)
    for (size_t i = j; i < (size_t)N * E; i += 16) s.embeddings[rb * E + i] = 0.0f;
  float x[kMaxAS], pr[kMaxAS], lg[kMaxAS];
  bool inv[kMaxAS];
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    x[t] = a < A ? prior_logits[(size_t)r * A + a] : 0.0f;
    inv[t] = (invalid != nullptr && a < A) ? invalid[(size_t)r * A + a] != 0 : false;
    if (a < A) s.root_invalid[(size_t)r * A + a] = inv[t] ? 1 : 0;
  }
  float mx = -INFINITY;
  bool any_invalid = false;
  if (!gumbel_policy) {
    // mctx muzero_policy prelude: Dirichlet mix, log
    row_softmax_rt(x, A, j, pr);
    float keep = 1.0f - fraction;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      int a = j + 16 * t;
      float nz = (noise != nullptr && a < A) ? noise[(size_t)r * A + a] : 0.0f;
      float noisy = keep * pr[t] + fraction * nz;
      lg[t] = log_pos(fmaxf(noisy, kFltTiny));
      mx = a < A ? fmaxf(mx, lg[t]) : mx;
    }
    any_invalid = invalid != nullptr;
  } else {
    // mctx gumbel_muzero_policy: the logits only pass through _mask_invalid_actions; root Gumbel noise
    const uint64_t rg = s.root_offset + (uint64_t)r;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      int a = j + 16 * t;
      lg[t] = x[t];
      mx = a < A ? fmaxf(mx, lg[t]) : mx;
      any_invalid = any_invalid || inv[t];
      if (a < A) {
        float g;
        if (gumbel_in != nullptr) {
          g = gumbel_in[(size_t)r * A + a];
        } else {
          uint32_t x0, x1;
          bool second;
          bits_block(s.global_batch * (uint64_t)A, rg * (uint64_t)A + (uint64_t)a, x0, x1, second);
          threefry2x32(gk0, gk1, x0, x1);
          g = s.gumbel_scale * gumbel_from_bits(second ? x1 : x0);
        }
        s.root_gumbel[(size_t)r * A + a] = g;
      }
    }
    any_invalid = ((__builtin_amdgcn_ballot_w64(any_invalid) >> (lane & 48)) & 0xffffull) != 0;  // any lane of the row
  }
  if (any_invalid) {
    mx = row_max<4>(mx);
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) lg[t] = inv[t] ? kFltLowest : lg[t] - mx;
  }
  float pq[kMaxAS];
  row_softmax_rt(lg, A, j, pq);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    if (a < A) {
      s.children_prior_logits[rb * A + a] = lg[t];
      s.children_prior_probs[rb * A + a] = pq[t];
    }
  }
  if (s.wide) {
    if (j == 0) s.xfer_node[r] = 0;
  } else {
    for (int i = j; i < E; i += 16) s.embeddings[rb * E + i] = embedding[(size_t)r * E + i];
  }
  if (j == 0) {
    s.node_visits[rb] = 1;
    s.raw_values[rb] = value[r];
    s.node_values[rb] = value[r];
    s.depth_sum[r] = 0;
  }
}
#endif  // MZ_NO_STEP_KERNELS

// mctx search.simulate (+ the parent-embedding gather of search.expand)
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void step_select_kernel(StepArgs s, int sim, int32_t* action_out,
                                                           float* parent_embedding_out) {
  MZ_ROW_SETUP
  const uint64_t rg = s.root_offset + (uint64_t)r;
  uint32_t k0 = 0, k1 = 0;
  if (s.tiebreak) {
    uint32_t x0, x1;
    bool second;
    bits_block(2 * s.global_batch, 2 * rg + (uint64_t)(j & 1), x0, x1, second);
    threefry2x32(s.sim_keys[2 * sim], s.sim_keys[2 * sim + 1], x0, x1);
    uint32_t word = second ? x1 : x0;
    k0 = bcast_u<0>(word);
    k1 = bcast_u<1>(word);
  }
  const int NB = (A + 1) / 2;
  int node = 0, depth = 0, parent = 0, action = 0;
  for (;;) {
    const size_t nb = (rb + node) * A;
    int nvis = s.node_visits[rb + node];
    float nval = s.node_values[rb + node];
    float tn = puct_scale(nvis, s.pb_c_init, s.pb_c_base);
    uint32_t s0 = 0, s1 = 0;
    if (s.tiebreak) {
      // rng_key, action_selection_key = split(rng_key)
      uint32_t x0 = (uint32_t)(j & 1), x1 = 2u + (uint32_t)(j & 1);
      threefry2x32(k0, k1, x0, x1);
      k0 = bcast_u<0>(x0); k1 = bcast_u<1>(x0);
      s0 = bcast_u<0>(x1); s1 = bcast_u<1>(x1);
    }
    float q[kMaxAS];
    int cidx[kMaxAS], cvis[kMaxAS];
    float prob[kMaxAS];
    float lo = nval, hi = nval;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      int a = j + 16 * t;
      bool ok = a < A;
      cidx[t] = -1; cvis[t] = 0; prob[t] = 0.0f; q[t] = 0.0f;
      if (16 * t < A) {  // (wave-uniform: slots past the action count cost nothing)
        size_t o = nb + (ok ? a : 0);
        cidx[t] = s.children_index[o];
        cvis[t] = s.children_visits[o];
        prob[t] = s.children_prior_probs[o];
        q[t] = s.children_rewards[o] + s.children_discounts[o] * s.children_values[o];
        float safe = (ok && cvis[t] > 0) ? q[t] : nval;
        lo = fminf(lo, safe);
        hi = fmaxf(hi, safe);
      }
    }
    lo = row_min<4>(lo);
    hi = row_max<4>(hi);
    float span = fmaxf(hi - lo, 1e-8f);
    float sc[kMaxAS];
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      int a = j + 16 * t;
      bool ok = a < A;
      sc[t] = -INFINITY;
      if (16 * t < A) {
        float value_score = ((cvis[t] > 0 ? q[t] : lo) - lo) / span;
        float policy_score = (tn * prob[t]) / (float)(cvis[t] + 1);
        sc[t] = value_score + policy_score;
        if (depth == 0 && ok && s.root_invalid[(size_t)r * A + a]) sc[t] = -INFINITY;
        if (!ok) sc[t] = -INFINITY;
      }
    }
    // mctx adds 1e-7 * uniform[0,1) to every score.  If fl(score_a + 1e-7) < best for every other action
    // the argmax cannot depend on the draw (rounding is monotone): only waves that hold a near tie pay for
    // the threefry blocks.  (-inf scores stay -inf with or without noise.)
    float bscore = -INFINITY;
    int best = 1 << 20, bnext = -1;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      int a = j + 16 * t;
      bool take = (t == 0) || (sc[t] > bscore);  // first max wins inside the lane
      if (take) { bscore = sc[t]; best = (a < A) ? a : (1 << 20); bnext = cidx[t]; }
    }
    row_argmax<4>(bscore, best, bnext);
    bool need_noise = false;
    if (s.tiebreak) {
      bool unsafe = false;
#pragma unroll
      for (int t = 0; t < kMaxAS; ++t) {
        int a = j + 16 * t;
        unsafe = unsafe || (a < A && a != best && !((sc[t] + 1e-7f) < bscore));
      }
      need_noise = __any(unsafe);
    }
    if (need_noise) {
      bscore = -INFINITY; best = 1 << 20; bnext = -1;
#pragma unroll
      for (int t = 0; t < kMaxAS; ++t) {
        int a = j + 16 * t;
        float score = sc[t];
        if (16 * t < A) {
          int jb = a < NB ? a : a - NB;
          uint32_t x0 = (uint32_t)jb, x1 = (NB + jb < A) ? (uint32_t)(NB + jb) : 0u;
          threefry2x32(s0, s1, x0, x1);
          score = score + 1e-7f * uniform_from_bits(a < NB ? x0 : x1);
        }
        bool take = (t == 0) || (score > bscore);
        if (take) { bscore = score; best = (a < A) ? a : (1 << 20); bnext = cidx[t]; }
      }
      row_argmax<4>(bscore, best, bnext);
    }
    row_argmax<4>(bscore, best, bnext);
    if (j == 0) s.path[rb + depth] = node | (best << 16);
    parent = node;
    action = best;
    depth += 1;
    if (bnext == -1 || depth >= s.max_depth) break;
    node = bnext;
  }
  if (j == 0) {
    s.sel_parent[r] = parent;
    s.sel_action[r] = action;
    s.sel_depth[r] = depth;
    s.depth_sum[r] += depth;
    action_out[r] = action;
  }
  if (s.wide) {
    if (j == 0) s.xfer_node[r] = parent;
  } else {
    const float* src = s.embeddings + (rb + parent) * E;
    for (int i = j; i < E; i += 16) parent_embedding_out[(size_t)r * E + i] = src[i];
  }
}
#endif  // MZ_NO_STEP_KERNELS

// mctx search.expand (update_tree_node + edge) and search.backward
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void step_expand_backup_kernel(StepArgs s, int sim, const float* reward,
                                                                  const float* discount,
                                                                  const float* prior_logits,
                                                                  const float* value,
                                                                  const float* next_embedding) {
  MZ_ROW_SETUP
  const int parent = s.sel_parent[r], action = s.sel_action[r], depth = s.sel_depth[r];
  const size_t eo = (rb + parent) * A + action;
  int next = s.children_index[eo];
  const int newn = next == -1 ? sim + 1 : next;
  const float v = value[r];
  float x[kMaxAS], pr[kMaxAS];
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    x[t] = a < A ? prior_logits[(size_t)r * A + a] : 0.0f;
  }
  row_softmax_rt(x, A, j, pr);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    if (a < A) {
      s.children_prior_logits[(rb + newn) * A + a] = x[t];
      s.children_prior_probs[(rb + newn) * A + a] = pr[t];
    }
  }
  if (s.wide) {
    if (j == 0) s.xfer_node[r] = newn;
  } else {
    for (int i = j; i < E; i += 16) s.embeddings[(rb + newn) * E + i] = next_embedding[(size_t)r * E + i];
  }
  const float rew_new = reward[r], dis_new = discount[r];
  if (j == 0) {
    s.raw_values[rb + newn] = v;
    s.node_values[rb + newn] = v;
    s.node_visits[rb + newn] = s.node_visits[rb + newn] + 1;
    s.children_index[eo] = newn;
    s.children_rewards[eo] = rew_new;
    s.children_discounts[eo] = dis_new;
    s.parents[rb + newn] = parent;
    s.action_from_parent[rb + newn] = action;
    // backward along the staged path (only this lane touches these words)
    float leaf = v, childv = v;
    for (int d = depth - 1; d >= 0; --d) {
      int pk = s.path[rb + d];
      int pn = pk & 0xffff, pa = pk >> 16;
      size_t e2 = (rb + pn) * A + pa;
      int cnt = s.node_visits[rb + pn];
      float rw = (d == depth - 1) ? rew_new : s.children_rewards[e2];
      float ds = (d == depth - 1) ? dis_new : s.children_discounts[e2];
      leaf = rw + ds * leaf;
      float newv = (s.node_values[rb + pn] * (float)cnt + leaf) / ((float)cnt + 1.0f);
      s.node_values[rb + pn] = newv;
      s.node_visits[rb + pn] = cnt + 1;
      s.children_values[e2] = childv;
      s.children_visits[e2] = s.children_visits[e2] + 1;
      childv = newv;
    }
  }
}
#endif  // MZ_NO_STEP_KERNELS

// mctx Tree.summary + _apply_temperature + jax.random.categorical
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void step_finish_kernel(StepArgs s, float temperature, const float* gumbel,
                                                           uint32_t ks0, uint32_t ks1, int32_t* action_out,
                                                           float* action_weights_out, float* search_value_out,
                                                           int32_t* depth_sum_out) {
  MZ_ROW_SETUP
  const uint64_t rg = s.root_offset + (uint64_t)r;
  int vc[kMaxAS];
  int part = 0;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    vc[t] = a < A ? s.children_visits[rb * A + a] : 0;
    part += vc[t];
  }
  float total = (float)row_sum_i(part);
  float denom = fmaxf(total, 1.0f);
  float lg[kMaxAS], prob[kMaxAS];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    float p = (float)vc[t] / denom;
    p = total > 0.0f ? p : 1.0f / (float)A;
    prob[t] = p;
    lg[t] = log_pos(fmaxf(p, kFltTiny));
    mx = a < A ? fmaxf(mx, lg[t]) : mx;
  }
  mx = row_max<4>(mx);
  float tden = fmaxf(temperature, kFltTiny);
  float bscore = -INFINITY;
  int best = 1 << 20, dummy = 0;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    bool ok = a < A;
    float g;
    if (gumbel != nullptr) {
      g = ok ? gumbel[(size_t)r * A + a] : 0.0f;
    } else {
      uint32_t x0, x1;
      bool second;
      bits_block(s.global_batch * (uint64_t)A, rg * (uint64_t)A + (uint64_t)(ok ? a : 0), x0, x1, second);
      threefry2x32(ks0, ks1, x0, x1);
      g = gumbel_from_bits(second ? x1 : x0);
    }
    float score = ok ? (lg[t] - mx) / tden + g : -INFINITY;
    if (ok) action_weights_out[(size_t)r * A + a] = prob[t];
    bool take = (t == 0) || (ok && score > bscore);
    if (take) { bscore = score; best = ok ? a : (1 << 20); }
  }
  row_argmax<4>(bscore, best, dummy);
  if (j == 0) {
    action_out[r] = best;
    if (search_value_out) search_value_out[r] = s.node_values[rb];
    if (depth_sum_out) depth_sum_out[r] = s.depth_sum[r];
  }
}
#endif  // MZ_NO_STEP_KERNELS

// canonical 16-wide sum of a row-distributed vector with run-time length
MZ_DEV float row_sum_rt(const float (&x)[kMaxAS], int A, int j) {
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    bool ok = j + 16 * t < A;
    part = (t == 0) ? (ok ? x[0] : 0.0f) : (ok ? part + x[t] : part);
  }
  return row_sum(part);
}

// mctx qtransforms for one node, row-distributed (kind 0: by_parent_and_siblings, 1: completed_by_mix_value)
MZ_DEV void row_qtransform(const StepArgs& s, size_t rb, int node, int A, int j, float (&out)[kMaxAS],
                           int (&vc)[kMaxAS], float (&logits)[kMaxAS], int& sum_visits) {
  const size_t nb = (rb + node) * A;
  float q[kMaxAS];
  int part = 0, mxv = 0;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    bool ok = a < A;
    size_t o = nb + (ok ? a : 0);
    vc[t] = ok ? s.children_visits[o] : 0;
    logits[t] = ok ? s.children_prior_logits[o] : 0.0f;
    q[t] = s.children_rewards[o] + s.children_discounts[o] * s.children_values[o];
    part += vc[t];
    mxv = max(mxv, vc[t]);
  }
  sum_visits = row_sum_i(part);
  if (s.qtransform == 0) {
    float nval = s.node_values[rb + node];
    float lo = nval, hi = nval;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      float safe = (j + 16 * t < A && vc[t] > 0) ? q[t] : nval;
      lo = fminf(lo, safe);
      hi = fmaxf(hi, safe);
    }
    lo = row_min<4>(lo);
    hi = row_max<4>(hi);
    float span = fmaxf(hi - lo, 1e-8f);
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) out[t] = ((vc[t] > 0 ? q[t] : lo) - lo) / span;
    return;
  }
  float maxvisit = (float)row_max_i(mxv);
  float prior[kMaxAS], tmp[kMaxAS];
  row_softmax_rt(logits, A, j, prior);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    prior[t] = fmaxf(prior[t], kFltTiny);
    tmp[t] = vc[t] > 0 ? prior[t] : 0.0f;
  }
  float sum_probs = row_sum_rt(tmp, A, j);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) tmp[t] = vc[t] > 0 ? (prior[t] * q[t]) / sum_probs : 0.0f;
  float weighted_q = row_sum_rt(tmp, A, j);
  float value = (s.raw_values[rb + node] + (float)sum_visits * weighted_q) / (float)(sum_visits + 1);
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    out[t] = vc[t] > 0 ? q[t] : value;
    bool ok = j + 16 * t < A;
    lo = ok ? fminf(lo, out[t]) : lo;
    hi = ok ? fmaxf(hi, out[t]) : hi;
  }
  lo = row_min<4>(lo);
  hi = row_max<4>(hi);
  float span = fmaxf(hi - lo, 1e-8f);
  float scale = (50.0f + maxvisit) * 0.1f;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) out[t] = scale * ((out[t] - lo) / span);
}

// seq_halving.score_considered + masked_argmax over a row-distributed action set
MZ_DEV int row_gumbel_argmax(const StepArgs& s, int r, int A, int j, int considered_visit,
                             const float (&logits)[kMaxAS], const float (&qv)[kMaxAS], const int (&vc)[kMaxAS],
                             int (&cidx)[kMaxAS], int& next) {
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) mx = (j + 16 * t < A) ? fmaxf(mx, logits[t]) : mx;
  mx = row_max<4>(mx);
  float bscore = -INFINITY;
  int best = 1 << 20, bnext = -1;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    int a = j + 16 * t;
    bool ok = a < A;
    float g = ok ? s.root_gumbel[(size_t)r * A + a] : 0.0f;
    float sc = (g + (logits[t] - mx)) + qv[t];
    sc = fmaxf(sc, -1e9f);
    sc = sc + (vc[t] == considered_visit ? 0.0f : -INFINITY);
    if (ok && s.root_invalid[(size_t)r * A + a]) sc = -INFINITY;
    if (!ok) sc = -INFINITY;
    bool take = (t == 0) || (sc > bscore);
    if (take) { bscore = sc; best = ok ? a : (1 << 20); bnext = cidx[t]; }
  }
  row_argmax<4>(bscore, best, bnext);
  next = bnext;
  return best;
}

// mctx search.simulate with gumbel_muzero_{root,interior}_action_selection
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void step_select_gumbel_kernel(StepArgs s, int sim, int32_t* action_out,
                                                                  float* parent_embedding_out) {
  MZ_ROW_SETUP
  int node = 0, depth = 0, parent = 0, action = 0;
  for (;;) {
    const size_t nb = (rb + node) * A;
    float qv[kMaxAS], logits[kMaxAS];
    int vc[kMaxAS], cidx[kMaxAS], sum_visits;
    row_qtransform(s, rb, node, A, j, qv, vc, logits, sum_visits);
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) cidx[t] = s.children_index[nb + (j + 16 * t < A ? j + 16 * t : 0)];
    int best, next;
    if (depth == 0) {
      int ninv = 0;
#pragma unroll
      for (int t = 0; t < kMaxAS; ++t)
        ninv += (j + 16 * t < A && s.root_invalid[(size_t)r * A + j + 16 * t]) ? 1 : 0;
      const int num_valid = A - row_sum_i(ninv);
      const int num_considered = min(s.max_considered, num_valid);
      const int si = min(sum_visits, s.S - 1);
      const int considered_visit = s.visit_table[(size_t)num_considered * s.S + si];
      best = row_gumbel_argmax(s, r, A, j, considered_visit, logits, qv, vc, cidx, next);
    } else {
      float x[kMaxAS], p[kMaxAS];
#pragma unroll
      for (int t = 0; t < kMaxAS; ++t) x[t] = logits[t] + qv[t];
      row_softmax_rt(x, A, j, p);
      float bscore = -INFINITY;
      int b2 = 1 << 20, bnext = -1;
#pragma unroll
      for (int t = 0; t < kMaxAS; ++t) {
        int a = j + 16 * t;
        bool ok = a < A;
        float sc = ok ? p[t] - (float)vc[t] / (float)(1 + sum_visits) : -INFINITY;
        bool take = (t == 0) || (sc > bscore);
        if (take) { bscore = sc; b2 = ok ? a : (1 << 20); bnext = cidx[t]; }
      }
      row_argmax<4>(bscore, b2, bnext);
      best = b2;
      next = bnext;
    }
    if (j == 0) s.path[rb + depth] = node | (best << 16);
    parent = node;
    action = best;
    depth += 1;
    if (next == -1 || depth >= s.max_depth) break;
    node = next;
  }
  if (j == 0) {
    s.sel_parent[r] = parent;
    s.sel_action[r] = action;
    s.sel_depth[r] = depth;
    s.depth_sum[r] += depth;
    action_out[r] = action;
  }
  if (s.wide) {
    if (j == 0) s.xfer_node[r] = parent;
  } else {
    const float* src = s.embeddings + (rb + parent) * E;
    for (int i = j; i < E; i += 16) parent_embedding_out[(size_t)r * E + i] = src[i];
  }
}
#endif  // MZ_NO_STEP_KERNELS

// tail of mctx gumbel_muzero_policy: best considered action + completed-Q policy target
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void step_finish_gumbel_kernel(StepArgs s, int32_t* action_out,
                                                                  float* action_weights_out,
                                                                  float* search_value_out, int32_t* depth_sum_out) {
  MZ_ROW_SETUP
  float qv[kMaxAS], logits[kMaxAS];
  int vc[kMaxAS], cidx[kMaxAS] = {0, 0, 0, 0}, sum_visits, next;
  row_qtransform(s, rb, 0, A, j, qv, vc, logits, sum_visits);
  int mxv = 0;
  bool any_inv = false;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    mxv = max(mxv, vc[t]);
    any_inv = any_inv || (j + 16 * t < A && s.root_invalid[(size_t)r * A + j + 16 * t]);
  }
  const int considered_visit = row_max_i(mxv);
  const int best = row_gumbel_argmax(s, r, A, j, considered_visit, logits, qv, vc, cidx, next);
  any_inv = ((__builtin_amdgcn_ballot_w64(any_inv) >> (lane & 48)) & 0xffffull) != 0;
  float x[kMaxAS], w[kMaxAS];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    x[t] = logits[t] + qv[t];
    mx = (j + 16 * t < A) ? fmaxf(mx, x[t]) : mx;
  }
  if (any_inv) {
    mx = row_max<4>(mx);
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t)
      x[t] = (j + 16 * t < A && s.root_invalid[(size_t)r * A + j + 16 * t]) ? kFltLowest : x[t] - mx;
  }
  row_softmax_rt(x, A, j, w);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t)
    if (j + 16 * t < A) action_weights_out[(size_t)r * A + j + 16 * t] = w[t];
  if (j == 0) {
    action_out[r] = best;
    if (search_value_out) search_value_out[r] = s.node_values[rb];
    if (depth_sum_out) depth_sum_out[r] = s.depth_sum[r];
  }
}
#endif  // MZ_NO_STEP_KERNELS

#undef MZ_ROW_SETUP

// ---- wide embedding rows: one workgroup per (root, KiB of the row) ----
// dir 0: rows[b] <- tree.embeddings[b][xfer_node[b]]   (parent embedding for recurrent_fn)
// dir 1: tree.embeddings[b][xfer_node[b]] <- rows[b]   (root / next embedding)
#ifndef MZ_NO_STEP_KERNELS
__global__ __launch_bounds__(256) void emb_xfer_kernel(const StepArgs s, float* rows, int dir) {
  const int b = blockIdx.x;
  const size_t E = (size_t)s.E;
  float* node = s.embeddings + ((size_t)b * s.N + s.xfer_node[b]) * E;
  float* row = rows + (size_t)b * E;
  const size_t i0 = (size_t)blockIdx.y * 1024, i1 = i0 + 1024 < E ? i0 + 1024 : E;
  if (dir == 0)
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) row[i] = node[i];
  else
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) node[i] = row[i];
}
#endif  // MZ_NO_STEP_KERNELS

}  // namespace mz
