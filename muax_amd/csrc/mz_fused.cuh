// mz_fused.cuh -- the whole MuZero.act() search for the default MLP trio in ONE
// launch (reference path: muax/model.py:222-282 -> mctx.muzero_policy; nets
// muax/nn.py:59-115; codec muax/utils.py:70-102).
//
// Mapping (MI355X-first, not a translation of mctx's vmapped XLA program):
//   * one search root  = one DPP row (16 lanes); 4 roots per wavefront,
//     16 roots per 256-thread workgroup, no barrier after the prologue;
//   * the root's whole tree lives in LDS for the duration of the act
//     (struct-per-node records, NS words each), HBM is touched only for the
//     observation, the weights (once, into VGPRs) and the outputs;
//   * the MLPs run as row-distributed fma chains: input element i lives in lane
//     i&15 (slot i>>4) and is fetched with a row_newbcast DPP modifier; each
//     lane keeps its own column of every weight matrix in VGPRs;
//   * pUCT selection: lane a scores action a, first-max argmax is a DPP
//     butterfly; tie-break noise is JAX's threefry stream, generated one level
//     ahead by otherwise idle lanes (14/15 split the key, lanes < A draw bits);
//   * the selected path is staged in LDS so that backup never chases parents.
#pragma once
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

constexpr int kMaxSims = 256;
constexpr int kHidden = 16;  // hk.Linear(16) everywhere in muax/nn.py:73-115

struct FusedParams {
  // inputs
  const float* obs;              // [B, obs_dim]
  const float* dirichlet_noise;  // [B, A] or null
  const uint8_t* invalid;        // [B, A] or null
  const float* gumbel;           // [B, A] or null (null -> threefry from k_sample)
  // weights, haiku layout w[in][out]
  const float *repr_w, *repr_b;
  const float *pv_w1, *pv_b1, *pv_w2, *pv_b2;
  const float *pp_w1, *pp_b1, *pp_w2, *pp_b2;
  const float *dr_w1, *dr_b1, *dr_w2, *dr_b2;
  const float *dn_w1, *dn_b1, *dn_w2, *dn_b2;
  // outputs
  int32_t* action;        // [B]
  float* action_weights;  // [B, A]
  float* root_value;      // [B]   network value of the root (muax/model.py:243)
  float* search_value;    // [B]   node_values[:,0] after search, or null
  int32_t* depth_sum;     // [B]   sum over simulations of selection depth, or null
  // optional tree export, mctx layout ([B,N], [B,N,A], [B,N,E]); all or none
  int32_t* t_node_visits; float* t_raw_values; float* t_node_values;
  int32_t* t_parents; int32_t* t_action_from_parent;
  int32_t* t_children_index; float* t_children_prior_logits; float* t_children_values;
  int32_t* t_children_visits; float* t_children_rewards; float* t_children_discounts;
  float* t_embeddings;
  // scalars
  int32_t B, obs_dim, S, max_depth, support, pred_on_parent, export_tree;
  float pb_c_init, pb_c_base, dirichlet_fraction, discount, temperature;
  uint64_t global_batch, root_offset;
  uint32_t k_sample[2];
  uint32_t sim_keys[kMaxSims][2];
};

template <int A_, int E_, int F_, int NMAX_, bool TB_, int WAVES_ = 4>
struct FusedCfg {
  static constexpr int WAVES = WAVES_, THREADS = 64 * WAVES_;
  static constexpr int A = A_, E = E_, F = F_, NMAX = NMAX_;
  static constexpr bool TB = TB_;
  static constexpr int H = kHidden;
  static constexpr int ES = (E + 15) / 16, FS = (F + 15) / 16;
  static constexpr int ASTEPS = ceil_log2(A);
  // node record (32-bit words): visits, value, puct scale, pad, A x {index, prob,
  // value, visits, reward, discount}, embedding
  static constexpr int CH0 = 4, CHW = 6;
  static constexpr int EMB0 = CH0 + CHW * A;
  static constexpr int NS = EMB0 + E;
  static constexpr int TREE_WORDS = NS * NMAX;
  static constexpr int PATH_WORDS = NMAX;
  static constexpr int ROOT_WORDS = TREE_WORDS + PATH_WORDS;
  static constexpr int ROOTS_PER_WG = 4 * WAVES;
  static constexpr int TBL_WORDS = ((NMAX + 2 + 3) / 4) * 4;
  static constexpr int LDS_BYTES = 4 * (TBL_WORDS + ROOTS_PER_WG * ROOT_WORDS);
  static_assert(A <= 14, "lanes 14/15 of the row split the PRNG key");
  static_assert(F <= 32 && E <= 32 * 16, "row-distributed vectors");
};

// y = x . W + b for row-distributed vectors; W column(s) of this lane in VGPRs.
template <int NIN, int NOUT>
struct RowLinear {
  static constexpr int IS = (NIN + 15) / 16, OS = (NOUT + 15) / 16;
  float w[NIN][OS];
  float b[OS];
  MZ_DEV void load(const float* __restrict__ W, const float* __restrict__ Bv, int j) {
#pragma unroll
    for (int t = 0; t < OS; ++t) {
      int k = j + 16 * t;
      b[t] = k < NOUT ? Bv[k] : 0.0f;
#pragma unroll
      for (int i = 0; i < NIN; ++i) w[i][t] = k < NOUT ? W[i * NOUT + k] : 0.0f;
    }
  }
  // k-ordered fma chain from 0, bias added last (haiku Linear: dot then + b)
  MZ_DEV void dot(const float (&x)[IS], float (&acc)[OS]) const {
#pragma unroll
    for (int t = 0; t < OS; ++t) acc[t] = 0.0f;
    StaticFor<0, NIN>::run([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float xb = bcast<(i & 15)>(x[i >> 4]);
#pragma unroll
      for (int t = 0; t < OS; ++t) acc[t] = __builtin_fmaf(xb, w[i][t], acc[t]);
    });
  }
  MZ_DEV void apply(const float (&x)[IS], float (&y)[OS]) const {
    dot(x, y);
#pragma unroll
    for (int t = 0; t < OS; ++t) y[t] = y[t] + b[t];
  }
};

// first layer of Dynamic: input [s, onehot(a)] (muax/nn.py:104-110).  The
// one-hot rows are E..E+A-1 of W; zero terms of the chain are exact no-ops, so
// the chain is "s terms, then + W[E+a]".
template <int E, int A>
struct RowLinearOneHot {
  static constexpr int IS = (E + 15) / 16;
  float w[E];
  float wa[A];
  float b;
  MZ_DEV void load(const float* __restrict__ W, const float* __restrict__ Bv, int j) {
    b = Bv[j];
#pragma unroll
    for (int i = 0; i < E; ++i) w[i] = W[i * kHidden + j];
#pragma unroll
    for (int a = 0; a < A; ++a) wa[a] = W[(E + a) * kHidden + j];
  }
  MZ_DEV float apply(const float (&x)[IS], int action) const {
    float acc = 0.0f;
    StaticFor<0, E>::run([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      acc = __builtin_fmaf(bcast<(i & 15)>(x[i >> 4]), w[i], acc);
    });
    float wsel = wa[0];
#pragma unroll
    for (int a = 1; a < A; ++a) wsel = (action == a) ? wa[a] : wsel;
    acc = acc + wsel;
    return acc + b;
  }
};

// jax.nn.softmax over a row-distributed vector of N elements (N <= 32)
template <int N>
MZ_DEV void row_softmax(const float (&x)[(N + 15) / 16], int j, float (&p)[(N + 15) / 16]) {
  constexpr int NSLOT = (N + 15) / 16;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) m = (j + 16 * t < N) ? fmaxf(m, x[t]) : m;
  m = row_max<4>(m);
  float e[NSLOT];
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    bool ok = j + 16 * t < N;
    e[t] = ok ? exp_neg(x[t] - m) : 0.0f;
    part = (t == 0) ? e[0] : (ok ? part + e[t] : part);
  }
  float s = row_sum(part);
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) p[t] = e[t] / s;
}

// support_to_scalar(softmax(logits)) (muax/utils.py:94-102, muax/model.py:254,273-274)
template <int F>
MZ_DEV float row_decode(const float (&logits)[(F + 15) / 16], int j, int support) {
  constexpr int NSLOT = (F + 15) / 16;
  float p[NSLOT];
  row_softmax<F>(logits, j, p);
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    bool ok = j + 16 * t < F;
    float term = (float)(j + 16 * t - support) * p[t];
    part = (t == 0) ? (ok ? term : 0.0f) : (ok ? part + term : part);
  }
  return inv_scaling(row_sum(part));
}

// muax/nn.py:37-44 over a row-distributed vector
template <int E>
MZ_DEV void row_min_max_normalize(float (&s)[(E + 15) / 16], int j) {
  constexpr int NSLOT = (E + 15) / 16;
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    bool ok = j + 16 * t < E;
    mn = ok ? fminf(mn, s[t]) : mn;
    mx = ok ? fmaxf(mx, s[t]) : mx;
  }
  mn = row_min<4>(mn);
  mx = row_max<4>(mx);
  float scale = mx - mn;
  scale = scale < 1e-5f ? scale + 1e-5f : scale;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) s[t] = (s[t] - mn) / scale;
}

template <class C>
struct Nets {
  RowLinear<C::E, kHidden> pv1, pp1;
  RowLinear<kHidden, C::F> pv2;
  RowLinear<kHidden, C::A> pp2;
  RowLinearOneHot<C::E, C::A> dr1, dn1;
  RowLinear<kHidden, C::F> dr2;
  RowLinear<kHidden, C::E> dn2;

  MZ_DEV void load(const FusedParams& p, int j) {
    pv1.load(p.pv_w1, p.pv_b1, j); pv2.load(p.pv_w2, p.pv_b2, j);
    pp1.load(p.pp_w1, p.pp_b1, j); pp2.load(p.pp_w2, p.pp_b2, j);
    dr1.load(p.dr_w1, p.dr_b1, j); dr2.load(p.dr_w2, p.dr_b2, j);
    dn1.load(p.dn_w1, p.dn_b1, j); dn2.load(p.dn_w2, p.dn_b2, j);
  }
  // Prediction (muax/nn.py:73-90) + value decode
  MZ_DEV void predict(const float (&s)[C::ES], int j, int support, float& value,
                      float& pi_logit) const {
    float h[1], v_logits[C::FS], pl[1];
    pv1.apply(s, h);
    h[0] = elu(h[0]);
    pv2.apply(h, v_logits);
    float g[1];
    pp1.apply(s, g);
    g[0] = elu(g[0]);
    pp2.apply(g, pl);
    pi_logit = pl[0];
    value = row_decode<C::F>(v_logits, j, support);
  }
  // Dynamic (muax/nn.py:93-115) + reward decode
  MZ_DEV void dynamics(const float (&s)[C::ES], int action, int j, int support, float& reward,
                       float (&ns)[C::ES]) const {
    float h[1], r_logits[C::FS];
    h[0] = elu(dr1.apply(s, action));
    dr2.apply(h, r_logits);
    float g[1];
    g[0] = elu(dn1.apply(s, action));
    dn2.apply(g, ns);
    row_min_max_normalize<C::E>(ns, j);
    reward = row_decode<C::F>(r_logits, j, support);
  }
};

// One threefry pass for the row: lanes 14/15 split `key`, lanes < A draw the
// tie-break bits from `sel`.  Returns this lane's noise bits (for the level
// AFTER the one `sel` belonged to -- see the pipeline in the kernel).
template <int A>
struct RowRng {
  uint32_t k0, k1;  // walking key (mctx simulate: rng_key)
  uint32_t s0, s1;  // action_selection_key of the next level
  MZ_DEV uint32_t pass(int j) {
    constexpr int NB = (A + 1) / 2;
    bool splitter = j >= 14;
    int jb = j < NB ? j : j - NB;  // noise block of action j
    uint32_t x0 = splitter ? (uint32_t)(j - 14) : (uint32_t)jb;
    uint32_t x1 = splitter ? (uint32_t)(j - 12) : ((NB + jb < A) ? (uint32_t)(NB + jb) : 0u);
    uint32_t kk0 = splitter ? k0 : s0, kk1 = splitter ? k1 : s1;
    threefry2x32(kk0, kk1, x0, x1);
    // split(key) -> flat [y0(blk0), y0(blk1), y1(blk0), y1(blk1)]
    k0 = bcast_u<14>(x0); k1 = bcast_u<15>(x0);
    s0 = bcast_u<14>(x1); s1 = bcast_u<15>(x1);
    return j < NB ? x0 : x1;
  }
};

template <class C>
__global__ __launch_bounds__(C::THREADS, 1) void mz_act_fused_kernel(const FusedParams p) {
  constexpr int A = C::A, E = C::E, NS = C::NS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int j = lane & 15;
  const int root_in_wg = (tid >> 6) * 4 + (lane >> 4);
  const int r = blockIdx.x * C::ROOTS_PER_WG + root_in_wg;

  float* tbl = lds;  // puct scale by visit count
  for (int i = tid; i < C::TBL_WORDS; i += C::THREADS) tbl[i] = puct_scale(i, p.pb_c_init, p.pb_c_base);
  __syncthreads();
  if (r >= p.B) return;  // whole row leaves together; no barrier below

  float* tree = lds + C::TBL_WORDS + root_in_wg * C::ROOT_WORDS;
  int* itree = reinterpret_cast<int*>(tree);
  int* path = itree + C::TREE_WORDS;
  const uint64_t rg = p.root_offset + (uint64_t)r;
  const int S = p.S;
  const int max_depth = p.max_depth > 0 ? p.max_depth : S;
  const int N = S + 1;
  const int support = p.support;
  const bool ex = p.export_tree != 0;

  Nets<C> nets;
  nets.load(p, j);

  // ---- tree init (mctx instantiate_tree_from_root) ----
  for (int n = 0; n < N; ++n) {
    for (int wq = j; wq < NS; wq += 16) {
      bool is_index = wq >= C::CH0 && wq < C::EMB0 && ((wq - C::CH0) % C::CHW == 0);
      itree[n * NS + wq] = is_index ? -1 : 0;
    }
  }
  if (ex) {
    for (int n = j; n < N; n += 16) {
      size_t o = (size_t)r * N + n;
      p.t_raw_values[o] = 0.0f;
      p.t_parents[o] = -1;
      p.t_action_from_parent[o] = -1;
    }
    for (int i = j; i < N * A; i += 16) p.t_children_prior_logits[(size_t)r * N * A + i] = 0.0f;
  }

  // ---- root inference (muax/model.py:251-263) ----
  float s[C::ES];
  {
    const float* ob = p.obs + (size_t)r * p.obs_dim;
#pragma unroll
    for (int t = 0; t < C::ES; ++t) {
      int k = j + 16 * t;
      float acc = 0.0f;
      if (k < E) {
        for (int i = 0; i < p.obs_dim; ++i) acc = __builtin_fmaf(ob[i], p.repr_w[i * E + k], acc);
        acc = acc + p.repr_b[k];
      }
      s[t] = acc;
    }
    row_min_max_normalize<E>(s, j);
  }
  float v0, pl0;
  nets.predict(s, j, support, v0, pl0);
  const bool inv_lane = (p.invalid != nullptr) && (j < A) && p.invalid[(size_t)r * A + (j < A ? j : 0)];
  {
    // mctx muzero_policy prelude: dirichlet mix, log, invalid-action mask
    float x[1] = {pl0}, pr[1];
    row_softmax<A>(x, j, pr);
    float nz = (p.dirichlet_noise != nullptr && j < A) ? p.dirichlet_noise[(size_t)r * A + j] : 0.0f;
    float keep = 1.0f - p.dirichlet_fraction;
    float noisy = keep * pr[0] + p.dirichlet_fraction * nz;
    float lg = log_pos(fmaxf(noisy, kFltTiny));
    if (p.invalid != nullptr) {
      float mx = row_max<4>(j < A ? lg : -INFINITY);
      lg = inv_lane ? kFltLowest : lg - mx;
    }
    float lx[1] = {lg}, pq[1];
    row_softmax<A>(lx, j, pq);
    if (j < A) tree[C::CH0 + C::CHW * j + 1] = pq[0];
    if (ex && j < A) p.t_children_prior_logits[(size_t)r * N * A + j] = lg;
    if (j == 0) {
      itree[0] = 1;
      tree[1] = v0;
      tree[2] = tbl[1];
      p.root_value[r] = v0;
      if (ex) p.t_raw_values[(size_t)r * N] = v0;
    }
#pragma unroll
    for (int t = 0; t < C::ES; ++t)
      if (j + 16 * t < E) tree[C::EMB0 + j + 16 * t] = s[t];
  }

  int depth_total = 0;
  const int ja = j < A ? j : A - 1;  // lanes >= A shadow the last action (masked later)

  // ---- simulations (mctx search.search body_fun) ----
  for (int sim = 0; sim < S; ++sim) {
    RowRng<A> rng;
    uint32_t nbits = 0;
    if constexpr (C::TB) {
      // simulate_keys[b] = split(simulate_key, B)[b]: words 2b, 2b+1 of the flat stream
      uint32_t x0, x1;
      bool second;
      bits_block(2 * p.global_batch, 2 * rg + (uint64_t)(j & 1), x0, x1, second);
      threefry2x32(p.sim_keys[sim][0], p.sim_keys[sim][1], x0, x1);
      uint32_t word = second ? x1 : x0;
      rng.k0 = bcast_u<0>(word);
      rng.k1 = bcast_u<1>(word);
      rng.s0 = 0; rng.s1 = 0;
      rng.pass(j);          // -> key_1, sel_0
      nbits = rng.pass(j);  // -> key_2, sel_1, bits of level 0
    }

    // -- simulate (mctx search.simulate) --
    int node = 0, depth = 0, parent = 0, action = 0, next = -1;
    for (;;) {
      const float* nd = tree + node * NS;
      const int* ndi = itree + node * NS;
      int nvis = ndi[0];
      float nval = nd[1];
      float tn = nd[2];
      const int co = C::CH0 + C::CHW * ja;
      int cidx = ndi[co + 0];
      float prob = nd[co + 1];
      float cval = nd[co + 2];
      int cvis = ndi[co + 3];
      float crew = nd[co + 4];
      float cdis = nd[co + 5];
      float noise = 0.0f;
      if constexpr (C::TB) {
        noise = 1e-7f * uniform_from_bits(nbits);
        nbits = rng.pass(j);  // bits for the next level, off the critical path
      }
      // qtransform_by_parent_and_siblings
      float q = crew + cdis * cval;
      bool has = cvis > 0;
      float safe = (has && j < A) ? q : nval;
      float lo = fminf(nval, row_min<C::ASTEPS>(safe));
      float hi = fmaxf(nval, row_max<C::ASTEPS>(safe));
      float span = fmaxf(hi - lo, 1e-8f);
      float value_score = ((has ? q : lo) - lo) / span;
      // muzero_action_selection
      float policy_score = (tn * prob) / (float)(cvis + 1);
      float score = value_score + policy_score;
      if constexpr (C::TB) score = score + noise;
      score = (depth == 0 && inv_lane) ? -INFINITY : score;
      score = j < A ? score : -INFINITY;
      int best = j, nxt = cidx;
      row_argmax<C::ASTEPS>(score, best, nxt);
      best = bcast_i<0>(best);  // lanes >= 2^ASTEPS did not take part: keep the row uniform
      nxt = bcast_i<0>(nxt);
      (void)nvis;
      if (j == 0) path[depth] = node | (best << 16);
      parent = node;
      action = best;
      next = nxt;
      depth += 1;
      if (next == -1 || depth >= max_depth) break;
      node = next;
    }
    depth_total += depth;
    const bool fresh = next == -1;
    const int newn = fresh ? sim + 1 : next;

    // -- expand (mctx search.expand, recurrent_fn = muax/model.py:265-282) --
    float sp[C::ES];
#pragma unroll
    for (int t = 0; t < C::ES; ++t)
      sp[t] = (j + 16 * t < E) ? tree[parent * NS + C::EMB0 + j + 16 * t] : 0.0f;
    float reward, value, pil;
    float ns[C::ES];
    nets.dynamics(sp, action, j, support, reward, ns);
    if (p.pred_on_parent) nets.predict(sp, j, support, value, pil);
    else nets.predict(ns, j, support, value, pil);
    float px[1] = {pil}, pp[1];
    row_softmax<A>(px, j, pp);
    {
      float* nn = tree + newn * NS;
      int* nni = itree + newn * NS;
      int vis = nni[0] + 1;
      if (j < A) nn[C::CH0 + C::CHW * j + 1] = pp[0];
#pragma unroll
      for (int t = 0; t < C::ES; ++t)
        if (j + 16 * t < E) nn[C::EMB0 + j + 16 * t] = ns[t];
      if (j == 0) {
        nni[0] = vis;
        nn[1] = value;
        nn[2] = tbl[vis];
        int eo = parent * NS + C::CH0 + C::CHW * action;
        itree[eo + 0] = newn;
        tree[eo + 4] = reward;
        tree[eo + 5] = p.discount;
      }
      if (ex) {
        size_t o = (size_t)r * N + newn;
        if (j < A) p.t_children_prior_logits[o * A + j] = pil;
        if (j == 0) {
          p.t_raw_values[o] = value;
          p.t_parents[o] = parent;
          p.t_action_from_parent[o] = action;
        }
      }
    }

    // -- backward (mctx search.backward), walking the staged path --
    {
      float leaf = value;
      float childv = value;
      for (int d = depth - 1; d >= 0; --d) {
        int pk = path[d];
        int pn = pk & 0xffff, pa = pk >> 16;
        float* nd = tree + pn * NS;
        int* ndi = itree + pn * NS;
        int cnt = ndi[0];
        float pv = nd[1];
        int eo = C::CH0 + C::CHW * pa;
        int cv = ndi[eo + 3];
        float rew = nd[eo + 4];
        float dis = nd[eo + 5];
        leaf = rew + dis * leaf;
        float newv = (pv * (float)cnt + leaf) / ((float)cnt + 1.0f);
        if (j == 0) {
          nd[1] = newv;
          ndi[0] = cnt + 1;
          nd[2] = tbl[cnt + 1];
          nd[eo + 2] = childv;
          ndi[eo + 3] = cv + 1;
        }
        childv = newv;
      }
    }
  }

  // ---- summary + sample (mctx Tree.summary, _apply_temperature, categorical) ----
  {
    int vc = itree[C::CH0 + C::CHW * ja + 3];
    vc = j < A ? vc : 0;
    float total = (float)row_sum_i(vc);
    float denom = fmaxf(total, 1.0f);
    float prob = (float)vc / denom;
    prob = total > 0.0f ? prob : 1.0f / (float)A;
    float lg = log_pos(fmaxf(prob, kFltTiny));
    float mx = row_max<4>(j < A ? lg : -INFINITY);
    float tden = fmaxf(p.temperature, kFltTiny);
    float al = (lg - mx) / tden;
    float g;
    if (p.gumbel != nullptr) {
      g = j < A ? p.gumbel[(size_t)r * A + j] : 0.0f;
    } else {
      uint32_t x0, x1;
      bool second;
      bits_block(p.global_batch * (uint64_t)A, rg * (uint64_t)A + (uint64_t)ja, x0, x1, second);
      threefry2x32(p.k_sample[0], p.k_sample[1], x0, x1);
      g = gumbel_from_bits(second ? x1 : x0);
    }
    float score = j < A ? al + g : -INFINITY;
    int best = j, dummy = 0;
    row_argmax<C::ASTEPS>(score, best, dummy);
    if (j < A) p.action_weights[(size_t)r * A + j] = prob;
    if (j == 0) {
      p.action[r] = best;
      if (p.search_value) p.search_value[r] = tree[1];
      if (p.depth_sum) p.depth_sum[r] = depth_total;
    }
  }

  if (ex) {
    for (int n = 0; n < N; ++n) {
      size_t o = (size_t)r * N + n;
      const float* nd = tree + n * NS;
      const int* ndi = itree + n * NS;
      if (j == 0) {
        p.t_node_visits[o] = ndi[0];
        p.t_node_values[o] = nd[1];
      }
      if (j < A) {
        int co = C::CH0 + C::CHW * j;
        p.t_children_index[o * A + j] = ndi[co + 0];
        p.t_children_values[o * A + j] = nd[co + 2];
        p.t_children_visits[o * A + j] = ndi[co + 3];
        p.t_children_rewards[o * A + j] = nd[co + 4];
        p.t_children_discounts[o * A + j] = nd[co + 5];
      }
      for (int i = j; i < E; i += 16) p.t_embeddings[o * E + i] = nd[C::EMB0 + i];
    }
  }
}

}  // namespace mz
