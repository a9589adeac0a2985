// mz_step_jump.cuh -- step-wise MuZero search with cached decisions (the fused kernel's scheme) on the
// HBM-resident tree of mz_step.cuh.  Same C-ABI entry points (mzs_root / mzs_select / mzs_expand_backup),
// same arithmetic per node, same results bit for bit; what changes is WHERE the per-level work happens:
//
//   * mctx's simulate() walks the tree level by level, a chain of dependent decisions.  Here every node
//     carries a JUMP record: the end point (parent, action, level) of the greedy, noise-free descent below
//     it, stopped early at the first node whose argmax is a near tie (fl(score + 1e-7) !< best for some
//     other action: only there can mctx's 1e-7 * uniform tie-break noise matter).  Sub-trees off a
//     backed-up path never change, so their records stay valid.  mzs_select reads the root's record: O(1),
//     plus one exact noisy evaluation per near tie (JAX's key chain is walked lazily down to that level);
//   * the decisions of the nodes ON the backed-up path are refreshed by mzs_expand_backup, where the path
//     is known up front: one workgroup per root, one 16-lane row per LEVEL (lanes over the actions), 16
//     levels in flight, all their loads independent.  The discounted-return chain is the only sequential
//     part and runs over LDS.
//
// Every node stores its own root path (written once at expansion: parent's path + one entry), so the
// backup finds its levels without a pointer chase.  Used (both policies) when N <= kJumpMaxNodes and B N^2
// words fit the budget; otherwise mz_step.cuh's walking kernels run.
#pragma once
#include "mz_step.cuh"

#pragma clang fp contract(off)

namespace mz {

constexpr int kJumpMaxNodes = 1024;

// opt-in phase timers of the expand / backward / refresh body (tools/profile_search.py builds with -DMZ_PROFILE)
#ifdef MZ_PROFILE
__device__ unsigned long long g_jump_prof[1024 * 8];
#define MZ_JT_BEGIN unsigned long long jt_last = __builtin_amdgcn_s_memtime();
#define MZ_JT(k)                                                             \
  if (threadIdx.x == 0) {                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();              \
    g_jump_prof[(size_t)(blockIdx.x & 1023) * 8 + (k)] += t_ - jt_last;      \
    jt_last = t_;                                                            \
  }
#else
#define MZ_JT_BEGIN
#define MZ_JT(k)
#endif
constexpr int kLevelsInFlight = 3;  // levels of a backed-up path a 16-lane row refreshes at once (mzs_expand_backup)

#define MZ_JROW_SETUP                                                 \
  const int lane = threadIdx.x & 63;                                  \
  const int j = lane & 15;                                            \
  const int r = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);  \
  if (r >= s.B) return;                                               \
  const int N = s.N, A = s.A, E = s.E;                                \
  const size_t rb = (size_t)r * N;                                    \
  (void)lane; (void)N; (void)A; (void)E;

struct JumpArgs {
  int32_t* jump_pa;     // [B][N]  parent | action << 16 | near tie << 31
  int32_t* jump_lv;     // [B][N]  level of `parent` (edges from the root)
  int32_t* node_depth;  // [B][N]  length of the node's own root path
  uint32_t* node_path;  // [B][N][N]  entry e = node at level e | action taken there << 16
};

// The statistics of ONE root's tree that the per-simulation step both reads and rewrites (the MuZero policy's decisions:
// children_{index, visits, prior_probs, rewards, values}, node_{visits, values}, the JUMP records), as pointers to the
// root's own rows.  The step-wise kernels point them at the HBM tree (tree_view_global).  The one-launch ResNet search
// (mz_search_conv.hip) keeps them in the CU's LDS for the whole search when they fit (round 6): between two tree steps
// of a root its XCD streams 5.7 MB of convolution weights through a 4 MB L2, so every tree step used to find its ~260
// cache lines evicted -- 4.5 of the 7 us of a 44-level decision refresh were memory round trips, not arithmetic.
struct TreeView {
  int* cidx; int* cvis; float* prob; float* rew; float* val;  // [N][A], element (node, a) at [(node * A + a) * cs]
  float* dis;    // [N][A] children_discounts, or nullptr: `disc` on every edge (an unexpanded child's value is +0, so
  float disc;    // rew + disc * val is the same +0 as with the array's 0 there)
  int* nvis; float* nval;  // [N], element `node` at [node * ns]
  int* jpa; int* jlv;      // [N]
  // word strides: 1 / 1 for the HBM tree's separate arrays; the LDS copy interleaves the five words of a child and the
  // four of a node (cs = 5, ns = 4): one address per (node, action), its fields at immediate offsets (ds_read2_b32)
  int cs, ns;
  // root_invalid_actions of this lane's actions (bit t: action j + 16 t), read once by a caller that lives for a whole
  // search; -1: level_load reads s.root_invalid whenever it meets the root (a global round trip in front of the scores)
  int inv_bits;
};
MZ_DEV TreeView tree_view_global(const StepArgs& s, const JumpArgs& g, size_t rb) {
  const size_t o = rb * (size_t)s.A;
  TreeView T;
  T.cidx = s.children_index + o; T.cvis = s.children_visits + o; T.prob = s.children_prior_probs + o;
  T.rew = s.children_rewards + o; T.val = s.children_values + o; T.dis = s.children_discounts + o; T.disc = 0.0f;
  T.nvis = s.node_visits + rb; T.nval = s.node_values + rb; T.jpa = g.jump_pa + rb; T.jlv = g.jump_lv + rb;
  T.cs = 1; T.ns = 1; T.inv_bits = -1;
  return T;
}

// pUCT scores of all children of `node` (muzero_action_selection with qtransform_by_parent_and_siblings),
// noise-free; first-max argmax; `near` = some other action is within the reach of the tie-break noise.
// Split into the loads and the arithmetic so that a row can put the loads of several levels in flight before it
// computes on the first (mzs_expand_backup refreshes up to kLevelsInFlight levels per row at once).
struct LevelIn {
  int cidx[kMaxAS], cvis[kMaxAS];
  float prob[kMaxAS], rew[kMaxAS], dis[kMaxAS], val[kMaxAS];
  int nvis, inv;  // inv: bit t = action j + 16 t is invalid at the root
  float nval;
};
// AS: 0 = any action count up to 16 kMaxAS (every 16-lane slot behind a run-time test), else the number of slots the
// caller's action count fills (16 (AS - 1) < A <= 16 AS): the same operations without the dead slots' branches
template <int AS = 0>
MZ_DEV void level_load(const StepArgs& s, const TreeView& T, int r, int node, int j, LevelIn& L) {
  const int A = s.A;
  const int nb = node * A;
  L.nvis = T.nvis[node * T.ns];
  L.nval = T.nval[node * T.ns];
  L.inv = 0;
#pragma unroll
  for (int t = 0; t < (AS ? AS : kMaxAS); ++t) {
    const int a = j + 16 * t;
    const bool ok = a < A;
    L.cidx[t] = -1; L.cvis[t] = 0; L.prob[t] = 0.0f; L.rew[t] = 0.0f; L.dis[t] = 0.0f; L.val[t] = 0.0f;
    if (AS || 16 * t < A) {
      const int o = nb + (ok ? a : 0);
      L.cidx[t] = T.cidx[o * T.cs];
      L.cvis[t] = T.cvis[o * T.cs];
      L.prob[t] = T.prob[o * T.cs];
      L.rew[t] = T.rew[o * T.cs];
      L.dis[t] = T.dis ? T.dis[o * T.cs] : T.disc;
      L.val[t] = T.val[o * T.cs];
      if (T.inv_bits < 0 && node == 0 && ok && s.root_invalid[(size_t)r * A + a]) L.inv |= 1 << t;  // the root is level 0 only
    }
  }
  if (T.inv_bits >= 0 && node == 0) L.inv = T.inv_bits;
}
// `tbl` (optional, LDS): {sqrt(n) pb_c(n), RN(1 / n)} for n = 0 .. S + 1, built once per launch by a kernel that lives
// for a whole search (mz_search_conv.hip) -- the same puct_scale() values, and x / n by Markstein's exact sequence for
// n <= 1030 (div_small, mz_fused.cuh; tests/test_oracle_kat.py): the refresh of a path's decisions is arithmetic-bound,
// a log, a sqrt and three of its four IEEE divisions per level go
// TBL: `tbl` is there (the caller's launch built it): no code for the other case
template <int AS = 0, bool TBL = false>
MZ_DEV void level_compute(const StepArgs& s, int j, const LevelIn& L, float (&sc)[kMaxAS], int& best, int& child,
                          bool& near, const float* tbl = nullptr) {
  constexpr int NSLOT = AS ? AS : kMaxAS;
  const int A = s.A;
  const float nval = L.nval;
  const float tn = (TBL || tbl) ? tbl[2 * L.nvis] : puct_scale(L.nvis, s.pb_c_init, s.pb_c_base);
  float q[kMaxAS];
  float lo = nval, hi = nval;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    const bool ok = j + 16 * t < A;
    q[t] = 0.0f;
    if (AS || 16 * t < A) {
      q[t] = L.rew[t] + L.dis[t] * L.val[t];
      const float safe = (ok && L.cvis[t] > 0) ? q[t] : nval;
      lo = fminf(lo, safe);
      hi = fmaxf(hi, safe);
    }
  }
  lo = row_min<4>(lo);
  hi = row_max<4>(hi);
  const float span = fmaxf(hi - lo, 1e-8f);
  float bscore = -INFINITY;
  best = 1 << 20;
  child = -1;
  // two slots of actions: the lane's two value scores share their denominator -- ONE refined reciprocal and the
  // hardware's own quotient correction on the packed pair (div_newton2, mz_spec.cuh: the IEEE quotient while every
  // numerator is 0 or >= 2^-100 and the span is below 2^100; a wave-uniform test, the IEEE divisions otherwise) as in
  // the fused kernel's puct_scores: 12 instructions instead of 26
  [[maybe_unused]] f32x2 vs2 = splat2(0.0f);
  [[maybe_unused]] bool have_vs2 = false;
  if constexpr (AS == 2) {
    const float n0 = (L.cvis[0] > 0 ? q[0] : lo) - lo, n1 = (L.cvis[1] > 0 ? q[1] : lo) - lo;
    const uint32_t low = min(f2u(n0) - 1u, f2u(n1) - 1u);  // 0 wraps to the top: only (0, 2^-100) fails the test
    const bool risky = (low < f2u(0x1p-100f) - 1u) | !(span < 0x1p100f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(risky) == 0, 1)) {
      const float y0 = __builtin_amdgcn_rcpf(span);
      const float y = __builtin_fmaf(__builtin_fmaf(-span, y0, 1.0f), y0, y0);
      vs2 = div_newton2((f32x2){n0, n1}, splat2(span), splat2(y));
      have_vs2 = true;
    }
  }
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    const int a = j + 16 * t;
    const bool ok = a < A;
    sc[t] = -INFINITY;
    if (AS || 16 * t < A) {
      float value_score;
      if (AS == 2 && have_vs2) value_score = t == 0 ? vs2.x : vs2.y;
      else value_score = ((L.cvis[t] > 0 ? q[t] : lo) - lo) / span;
      float policy_score;
      if (TBL || tbl) {
        const float x = tn * L.prob[t], d = (float)(L.cvis[t] + 1), y = tbl[2 * (L.cvis[t] + 1) + 1];
        const float q0 = x * y;
        policy_score = __builtin_fmaf(__builtin_fmaf(-q0, d, x), y, q0);
      } else {
        policy_score = (tn * L.prob[t]) / (float)(L.cvis[t] + 1);
      }
      sc[t] = value_score + policy_score;
      if ((L.inv >> t) & 1) sc[t] = -INFINITY;
      if (!ok) sc[t] = -INFINITY;
    }
    const bool take = (t == 0) || (sc[t] > bscore);  // first max wins inside the lane
    if (take) { bscore = sc[t]; best = ok ? a : (1 << 20); child = L.cidx[t]; }
  }
  row_argmax<4>(bscore, best, child);
  bool unsafe = false;
  if (s.tiebreak) {
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) {
      const int a = j + 16 * t;
      unsafe = unsafe | ((a < A) & (a != best) & !((sc[t] + 1e-7f) < bscore));
    }
  }
  near = ((__builtin_amdgcn_ballot_w64(unsafe) >> (threadIdx.x & 48)) & 0xffffull) != 0;  // any lane of the row
}
MZ_DEV void level_decide(const StepArgs& s, const TreeView& T, int r, int node, int j, float (&sc)[kMaxAS],
                         int (&cidx)[kMaxAS], int& best, int& child, bool& near) {
  LevelIn L;
  level_load(s, T, r, node, j, L);
  level_compute(s, j, L, sc, best, child, near);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) cidx[t] = L.cidx[t];
}

// mctx gumbel_muzero_{root,interior}_action_selection at `node` (the root is only ever selected at level 0):
// deterministic in the node's statistics -- the root's sequential-halving table entry is indexed by its
// own visit sum, i.e. by the number of the NEXT simulation once the backup has run.
MZ_DEV void gumbel_decide(const StepArgs& s, size_t rb, int r, int node, int j, int& best, int& child) {
  const int A = s.A;
  const size_t nb = (rb + node) * A;
  float qv[kMaxAS], logits[kMaxAS];
  int vc[kMaxAS], cidx[kMaxAS], sum_visits;
  row_qtransform(s, rb, node, A, j, qv, vc, logits, sum_visits);
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) cidx[t] = s.children_index[nb + (j + 16 * t < A ? j + 16 * t : 0)];
  if (node == 0) {
    int ninv = 0;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) ninv += (j + 16 * t < A && s.root_invalid[(size_t)r * A + j + 16 * t]) ? 1 : 0;
    const int num_valid = A - row_sum_i(ninv);
    const int num_considered = min(s.max_considered, num_valid);
    const int si = min(sum_visits, s.S - 1);
    const int considered_visit = s.visit_table[(size_t)num_considered * s.S + si];
    best = row_gumbel_argmax(s, r, A, j, considered_visit, logits, qv, vc, cidx, child);
  } else {
    float x[kMaxAS], p[kMaxAS];
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) x[t] = logits[t] + qv[t];
    row_softmax_rt(x, A, j, p);
    float bscore = -INFINITY;
    best = 1 << 20;
    child = -1;
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      const int a = j + 16 * t;
      const bool ok = a < A;
      const float sc = ok ? p[t] - (float)vc[t] / (float)(1 + sum_visits) : -INFINITY;
      const bool take = (t == 0) || (sc > bscore);
      if (take) { bscore = sc; best = ok ? a : (1 << 20); child = cidx[t]; }
    }
    row_argmax<4>(bscore, best, child);
  }
}
template <bool GUMBEL>
MZ_DEV void decide_any(const StepArgs& s, const TreeView& T, size_t rb, int r, int node, int j, int& best, int& child, bool& near) {
  if constexpr (GUMBEL) {
    gumbel_decide(s, rb, r, node, j, best, child);  // (the Gumbel policy's statistics stay in the HBM tree)
    near = false;
  } else {
    float sc[kMaxAS];
    int cidx[kMaxAS];
    level_decide(s, T, r, node, j, sc, cidx, best, child, near);
  }
}

// root decision (after step_root_kernel): one row per root
template <bool GUMBEL>
__global__ __launch_bounds__(256) void jump_root_kernel(StepArgs s, JumpArgs g) {
  MZ_JROW_SETUP
  int best, child;
  bool near;
  decide_any<GUMBEL>(s, tree_view_global(s, g, rb), rb, r, 0, j, best, child, near);
  if (j == 0) {
    g.jump_pa[rb] = best << 16 | (near ? (int)0x80000000 : 0);
    g.jump_lv[rb] = 0;
    g.node_depth[rb] = 0;
  }
}

// mctx search.simulate through the JUMP records, for root r.  WG: the calling workgroup belongs to this root alone
// -- every row repeats the (O(1)) selection, so every thread knows the parent and the whole workgroup gathers its
// embedding row (a separate transfer kernel costs its own 4.7 us minimum); otherwise one 16-lane row per root.
// jump_select_core: the decision alone -- every 16-lane row that calls it gets the same (parent, action, depth).
MZ_DEV void jump_select_core(const StepArgs& s, const JumpArgs& g, const TreeView& T, int sim, int r, int& parent, int& action,
                             int& depth) {
  const int lane = opaque_tid() & 63;
  const int j = lane & 15;
  const int N = s.N, A = s.A;
  const size_t rb = (size_t)r * N;
  const uint64_t rg = s.root_offset + (uint64_t)r;
  const int NB = (A + 1) / 2;
  uint32_t k0 = 0, k1 = 0, s0 = 0, s1 = 0;
  int klevel = -1;  // key walk not started
  int jw = T.jpa[0], level = T.jlv[0];
  for (;;) {
    parent = jw & 0xffff;
    action = (jw >> 16) & 0xff;
    if (jw < 0 && level + 1 <= s.max_depth) {
      // near tie at `parent`: score + 1e-7 * uniform(action_selection_key of this level), as mctx
      if (klevel < 0) {
        uint32_t x0, x1;
        bool second;
        bits_block(2 * s.global_batch, 2 * rg + (uint64_t)(j & 1), x0, x1, second);
        threefry2x32(s.sim_keys[2 * sim], s.sim_keys[2 * sim + 1], x0, x1);
        const uint32_t word = second ? x1 : x0;
        k0 = bcast_u<0>(word);
        k1 = bcast_u<1>(word);
        klevel = 0;
      }
      while (klevel <= level) {  // rng_key, action_selection_key = split(rng_key), level by level
        uint32_t x0 = (uint32_t)(j & 1), x1 = 2u + (uint32_t)(j & 1);
        threefry2x32(k0, k1, x0, x1);
        k0 = bcast_u<0>(x0); k1 = bcast_u<1>(x0);
        s0 = bcast_u<0>(x1); s1 = bcast_u<1>(x1);
        klevel += 1;
      }
      float sc[kMaxAS];
      int cidx[kMaxAS], best, child;
      bool near;
      level_decide(s, T, r, parent, j, sc, cidx, best, child, near);
      float bscore = -INFINITY;
      best = 1 << 20;
      child = -1;
#pragma unroll
      for (int t = 0; t < kMaxAS; ++t) {
        const int a = j + 16 * t;
        float score = sc[t];
        if (16 * t < A) {
          const int jb = a < NB ? a : a - NB;
          uint32_t x0 = (uint32_t)jb, x1 = (NB + jb < A) ? (uint32_t)(NB + jb) : 0u;
          threefry2x32(s0, s1, x0, x1);
          score = score + 1e-7f * uniform_from_bits(a < NB ? x0 : x1);
        }
        const bool take = (t == 0) || (score > bscore);
        if (take) { bscore = score; best = (a < A) ? a : (1 << 20); child = cidx[t]; }
      }
      row_argmax<4>(bscore, best, child);
      action = best;
      if (child >= 0 && level + 1 < s.max_depth) {
        jw = T.jpa[child * T.ns];
        level = T.jlv[child * T.ns];
        continue;
      }
    }
    break;
  }
  depth = level + 1;
  if (depth > s.max_depth) {
    // the cached descent overshoots max_depth: stop at level max_depth - 1 of the same path
    depth = s.max_depth;
    const uint32_t ent = g.node_path[(rb + parent) * N + depth - 1];
    parent = (int)(ent & 0xffffu);
    action = (int)(ent >> 16);
  }
}
// `parent_embedding_out` == nullptr: no gather (the caller reads the tree's embedding row in place)
template <bool WG>
MZ_DEV void jump_select_body(const StepArgs& s, const JumpArgs& g, const TreeView& T, int sim, int r, int32_t* action_out,
                             float* parent_embedding_out, int* sel_out = nullptr, int* depth_acc = nullptr) {
  const int j = threadIdx.x & 15;
  const int N = s.N, E = s.E;
  const size_t rb = (size_t)r * N;
  int parent, action, depth;
  jump_select_core(s, g, T, sim, r, parent, action, depth);
  if (sel_out) { sel_out[0] = parent; sel_out[1] = action; sel_out[2] = depth; }
  if (WG ? threadIdx.x == 0 : j == 0) {
    s.sel_parent[r] = parent;
    s.sel_action[r] = action;
    s.sel_depth[r] = depth;
    if (depth_acc) *depth_acc += depth;
    else s.depth_sum[r] += depth;
    if (action_out) action_out[r] = action;
    if (WG) s.xfer_node[r] = parent;
  }
  if (parent_embedding_out == nullptr) return;
  const float* src = s.embeddings + (rb + parent) * E;
  if (WG) {
    for (int i = threadIdx.x; i < E; i += blockDim.x) parent_embedding_out[(size_t)r * E + i] = src[i];
  } else {
    for (int i = j; i < E; i += 16) parent_embedding_out[(size_t)r * E + i] = src[i];
  }
}

// WIDE: one 256-thread workgroup per root (wide embedding rows), else one row per root
template <bool WIDE>
__global__ __launch_bounds__(256) void jump_select_kernel(StepArgs s, JumpArgs g, int sim, int32_t* action_out,
                                                           float* parent_embedding_out) {
  const int r = WIDE ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4));
  if (r >= s.B) return;
  jump_select_body<WIDE>(s, g, tree_view_global(s, g, (size_t)r * s.N), sim, r, action_out, parent_embedding_out);
}

// mctx search.expand + search.backward + refresh of the decisions on the path: one workgroup per root.
// `next_action_out` != null: the NEXT simulation's selection (simulate() of sim + 1: the root's fresh JUMP record is
// in this workgroup's hands) and the gather of its parent's embedding row run as the tail of this launch -- one launch
// and one kernel boundary fewer per simulation (mzs_expand_backup_select).
// The body is a device function of the calling workgroup (any multiple of 64 threads): the step-wise kernel below runs
// it on a workgroup of its own, the fused ResNet search (mz_search_conv.hip) as the tail of the recurrent_fn pass of
// the same root.  `prior_logits_row`: the root's A logits; `next_embedding_row` == nullptr: the caller has written the
// new node's embedding row in place; `select_next`: also run simulate() of sim + 1 (sel_out[0..2] = its parent / action / depth
// in every thread; `next_parent_embedding_out` == nullptr: no gather).
// LDS of the body: per level e in [0, depth] (entry `depth` is the leaf), 15 arrays of N + 1 words
struct JumpLds {
  int *pn, *pa, *cnt;       // node at level e; action taken there (e < depth); node_visits before this backup
  float *val, *rw, *ds;     // node_values before this backup; reward / discount of edge e
  float *Gs, *nv;           // leaf_value arriving at level e; node value after this backup
  int *bst, *chd, *flg;     // refreshed decision, its child index, near tie
  int *cjp, *cjl;           // stored JUMP record of that child
  int *njp, *njl;           // new JUMP record of the level's node
  int *nxa, *nxb;           // pointer jumping (two buffers, in the backward pass's arrays, dead by then)
};
MZ_DEV JumpLds jump_lds(int* lds_i, int N) {
  const int D1 = N + 1;
  JumpLds L;
  L.pn = lds_i; L.pa = L.pn + D1; L.cnt = L.pa + D1;
  L.val = reinterpret_cast<float*>(L.cnt + D1); L.rw = L.val + D1; L.ds = L.rw + D1; L.Gs = L.ds + D1; L.nv = L.Gs + D1;
  L.bst = reinterpret_cast<int*>(L.nv + D1); L.chd = L.bst + D1; L.flg = L.chd + D1; L.cjp = L.flg + D1; L.cjl = L.cjp + D1;
  L.njp = L.cjl + D1; L.njl = L.njp + D1;
  L.nxa = L.cnt; L.nxb = reinterpret_cast<int*>(L.val);
  return L;
}
// The two pieces of the body that need nothing from recurrent_fn -- the fused search runs them while the root's
// convolution passes wait for their partner (mz_search_conv.hip); a barrier must separate them and follow the second.
// (1) path of the leaf: the parent's own root path + (parent, action); a fresh node gets its copy
MZ_DEV void jump_prefetch_path(const StepArgs& s, const JumpArgs& g, int r, const JumpLds& L, int tid, int nthr, int parent,
                               int action, int depth, int newn, bool fresh) {
  const int N = s.N;
  const size_t rb = (size_t)r * N;
  for (int e = tid; e < depth; e += nthr) {
    const uint32_t ent = (e < depth - 1) ? g.node_path[(rb + parent) * N + e] : ((uint32_t)parent | ((uint32_t)action << 16));
    L.pn[e] = (int)(ent & 0xffffu);
    L.pa[e] = (int)(ent >> 16);
    if (fresh) g.node_path[(rb + newn) * N + e] = ent;
  }
  if (tid == 0) {
    L.pn[depth] = newn;
    L.pa[depth] = 0;
    if (fresh) g.node_depth[rb + newn] = depth;
  }
}
// (2) per-level inputs of the backward pass (the last edge's reward / discount come from recurrent_fn: filled in later)
MZ_DEV void jump_prefetch_levels(const StepArgs& s, const TreeView& T, const JumpLds& L, int tid, int nthr, int depth) {
  for (int e = tid; e < depth; e += nthr) {
    const int e2 = L.pn[e] * s.A + L.pa[e];
    L.cnt[e] = T.nvis[(L.pn[e]) * T.ns];
    L.val[e] = T.nval[(L.pn[e]) * T.ns];
    L.rw[e] = (e == depth - 1) ? 0.0f : T.rew[e2 * T.cs];
    L.ds[e] = (e == depth - 1) ? 0.0f : (T.dis ? T.dis[e2 * T.cs] : T.disc);  // (levels above the last edge are expanded edges)
  }
}

// leaf_value = reward + discount * leaf_value, leaf to root (mctx search.backward): the one sequential chain of the step.
// One wavefront (`lane` = 0 .. 63 of the calling wavefront), 63 levels per chunk, every lane its own level: one step is
// X[e] = r[e] + d[e] * X[e + 1] on all lanes at once -- v_mul_f32_dpp wave_shl:1 + v_add_f32 -- and level e is final
// after (levels of the chunk - e) steps: ~25 cycles per level (round 3: two v_readlane + mul + add + select per level on
// the wave-uniform G, ~140 cycles as compiled; until round 6 a select per step, ~100).  The paths of a long search on few
// roots are 50 .. 150 levels deep and this chain paces the root.  Same operations in the same order for every level.
// `patch`: the last edge's (reward, discount) come from the caller's registers (rew_new, dis_new), not from rw / ds.
MZ_DEV void return_chain(const float* rw, const float* ds, float* Gs, float v, int depth, int lane, bool patch, float rew_new,
                         float dis_new) {
  float G = v;
  for (int c = (depth - 1) / 63; c >= 0; --c) {
    const int e = 63 * c + lane;
    const int cn = min(63, depth - 63 * c);
    const bool mine = lane < cn;
    // lanes past the chunk's levels run identity steps on the value arriving from below: (-0) + 1 * x == x for every x
    float rr = mine ? rw[e] : -0.0f, dd = mine ? ds[e] : 1.0f;
    if (patch && mine && e == depth - 1) {  // (`mine`: lane 63 of the chunk BELOW the last edge's has the same e)
      rr = rew_new;
      dd = dis_new;
    }
    float X = G, Xt = dd * G;  // (lane 63 has no lane above it: its Xt stays dd * G = G through every step)
    // four steps per statement: the DPP source X is the register the add before it has just written -- two wait
    // states (s_nop 1), which the hazard recogniser does not insert inside asm.  A level is final after (levels of
    // the chunk - e) steps and a further step maps it onto itself, so the count is rounded up to a multiple of four.
#define MZ_CSTEP "s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %1, %0, %3\n\t"
    for (int k = 0; k < cn; k += 4)
      asm volatile(MZ_CSTEP MZ_CSTEP MZ_CSTEP MZ_CSTEP : "+v"(Xt), "+v"(X) : "v"(dd), "v"(rr));
#undef MZ_CSTEP
    if (mine) Gs[e] = X;
    G = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(X)));  // arrival at the chunk's first level
  }
}

// `prefetched`: bit 0 = jump_prefetch_path, bit 1 = jump_prefetch_levels have run for this simulation (and a barrier since)
// AS: see level_load (0: any action count); `depth_acc` != nullptr: the next selection's depth is added there (a register
// of the caller, who owns depth_sum[r] for the launch) instead of to the HBM word -- a load-add-store round trip per
// simulation on the caller's critical path otherwise
// LDSB: every array the phases of this step hand to one another is in LDS (the LDS tree of the one-launch search): its
// barriers wait for the LDS queue only (s_waitcnt lgkmcnt(0); s_barrier) -- a __syncthreads() also drains vmcnt, i.e.
// waits for the acknowledgement of the write-only HBM stores (prior logits, parents, the selection ...) issued just before
template <bool LDSB>
MZ_DEV void step_barrier() {
  if constexpr (LDSB) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}
template <bool GUMBEL, int LIF = kLevelsInFlight, int AS = 0, bool TBL = false, bool LDSB = false>
MZ_DEV void jump_expand_backup_body(const StepArgs& s, const JumpArgs& g, const TreeView& T, int sim, int r, int* lds_i, float rew_new,
                                    float dis_new, const float* prior_logits_row, float v, const float* next_embedding_row,
                                    bool select_next, int32_t* next_action_out, float* next_parent_embedding_out,
                                    int* sel_out = nullptr, const int* known = nullptr, int prefetched = 0,
                                    const float* score_tbl = nullptr, int* depth_acc = nullptr) {
  const int tid = opaque_tid(), j = tid & 15, row = tid >> 4;
  MZ_JT_BEGIN
  const int nthr = blockDim.x, nrows = blockDim.x >> 4;  // 1024 / 256 threads (64 / 16 levels in flight) or one wavefront (4)
  const int N = s.N, A = s.A, E = s.E;
  const size_t rb = (size_t)r * N;
  // `known` = {parent, action, depth, new node} of this simulation when the caller has them in registers (the fused
  // search: two dependent memory round trips fewer on its critical path); an existing node has an index <= sim
  const int parent = known ? known[0] : s.sel_parent[r], action = known ? known[1] : s.sel_action[r];
  const int depth = known ? known[2] : s.sel_depth[r];
  const size_t eo = (rb + parent) * A + action;
  const int eol = parent * A + action;
  const int next = known ? (known[3] == sim + 1 ? -1 : known[3]) : T.cidx[eol * T.cs];
  __syncthreads();  // every thread has read the edge before row 0 rewrites it
  const bool fresh = next == -1;
  const int newn = fresh ? sim + 1 : next;
  const JumpLds JL = jump_lds(lds_i, N);
  int* const pn = JL.pn; int* const pa = JL.pa; int* const cnt = JL.cnt;
  float* const val = JL.val; float* const rw = JL.rw; float* const ds = JL.ds; float* const Gs = JL.Gs; float* const nv = JL.nv;
  int* const bst = JL.bst; int* const chd = JL.chd; int* const flg = JL.flg; int* const cjp = JL.cjp; int* const cjl = JL.cjl;
  int* const njp = JL.njp; int* const njl = JL.njl; int* const nxa = JL.nxa; int* const nxb = JL.nxb;

  // -- path of the leaf: the parent's own root path + (parent, action) --
  if (!(prefetched & 1)) jump_prefetch_path(s, g, r, JL, tid, nthr, parent, action, depth, newn, fresh);
  // the caller has prefetched the path AND the per-level inputs (the fused search, in idle convolution passes): the
  // discounted-return chain needs nothing else but (rew_new, dis_new, v), which are in registers -- it runs on the SECOND
  // wavefront right away, beside the expansion on the first (the new node's softmax, ~0.9 us), and the barrier between
  // the two phases goes
  const bool early_chain = prefetched == 3 && nthr >= 128;
  if (early_chain && (tid >> 6) == 1) return_chain(rw, ds, Gs, v, depth, tid & 63, true, rew_new, dis_new);
  // -- expand (row 0): prior of the new node, node and edge records --
  if (row == 0) {
    float x[kMaxAS], pr[kMaxAS];
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      const int a = j + 16 * t;
      x[t] = a < A ? prior_logits_row[a] : 0.0f;
    }
    row_softmax_rt(x, A, j, pr);
#pragma unroll
    for (int t = 0; t < kMaxAS; ++t) {
      const int a = j + 16 * t;
      if (a < A) {
        s.children_prior_logits[(rb + newn) * A + a] = x[t];
        T.prob[(newn * A + a) * T.cs] = pr[t];
      }
    }
    if (j == 0) {
      s.raw_values[rb + newn] = v;
      T.nval[newn * T.ns] = v;
      T.nvis[newn * T.ns] = T.nvis[newn * T.ns] + 1;
      T.cidx[eol * T.cs] = newn;
      T.rew[eol * T.cs] = rew_new;
      s.children_discounts[eo] = dis_new;  // (== T.dis[eol * T.cs] when the view is the HBM tree)
      s.parents[rb + newn] = parent;
      s.action_from_parent[rb + newn] = action;
      s.xfer_node[r] = newn;
    }
  }
  // (a whole workgroup per root: wide rows need no separate transfer kernel here)
  if (next_embedding_row != nullptr)
    for (int i = tid; i < E; i += nthr) s.embeddings[(rb + newn) * E + i] = next_embedding_row[i];
  step_barrier<LDSB>();
  MZ_JT(0)
  if (!early_chain) {
    // -- per-level inputs of the backward pass --
    if (!(prefetched & 2)) jump_prefetch_levels(s, T, JL, tid, nthr, depth);
    if (tid == (depth - 1) % nthr && depth > 0) {  // (the thread that wrote the entry, if it was written just now)
      rw[depth - 1] = rew_new;
      ds[depth - 1] = dis_new;
    }
    step_barrier<LDSB>();
    MZ_JT(1)
    if (tid < 64) return_chain(rw, ds, Gs, v, depth, tid, false, rew_new, dis_new);
    step_barrier<LDSB>();
  }
  MZ_JT(2)
  for (int e = tid; e <= depth; e += nthr)
    nv[e] = (e == depth) ? v : (val[e] * (float)cnt[e] + Gs[e]) / ((float)cnt[e] + 1.0f);
  step_barrier<LDSB>();
  for (int e = tid; e < depth; e += nthr) {
    const int e2 = pn[e] * A + pa[e];
    T.nval[(pn[e]) * T.ns] = nv[e];
    T.nvis[(pn[e]) * T.ns] = cnt[e] + 1;
    T.val[e2 * T.cs] = nv[e + 1];
    T.cvis[e2 * T.cs] = T.cvis[e2 * T.cs] + 1;
  }
  step_barrier<LDSB>();  // (workgroup-scope: the refreshed statistics are visible to every row below)
  MZ_JT(3)
  // -- decisions of the path nodes and the leaf: one row per level, LIF (kLevelsInFlight) levels per row at once (all
  // their loads are issued before the first is used: a deep path costs one memory round trip, not one per 64 levels) --
  for (int base = 0; base <= depth; base += nrows * LIF) {
    if constexpr (GUMBEL) {
      for (int u = 0; u < LIF; ++u) {
        const int e = base + u * nrows + row;
        if (e <= depth) {
          int best, child;
          bool near;
          decide_any<GUMBEL>(s, T, rb, r, pn[e], j, best, child, near);
          if (j == 0) {
            bst[e] = best;
            chd[e] = child;
            flg[e] = near ? 1 : 0;
            const bool off_path = child >= 0 && !(e < depth && child == pn[e + 1]);
            cjp[e] = off_path ? T.jpa[child * T.ns] : 0;
            cjl[e] = off_path ? T.jlv[child * T.ns] : 0;
          }
        }
      }
    } else {
      LevelIn L[LIF];
      int bestu[LIF], childu[LIF], cj[LIF], cl[LIF];
      bool nearu[LIF], offu[LIF];
#pragma unroll
      for (int u = 0; u < LIF; ++u) {
        const int e = base + u * nrows + row;
        level_load<AS>(s, T, r, pn[e <= depth ? e : depth], j, L[u]);
      }
#ifdef MZ_PROF_DECIDE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      MZ_JT(7)
#endif
#pragma unroll
      for (int u = 0; u < LIF; ++u) {
        const int e = base + u * nrows + row;
        float sc[kMaxAS];
        level_compute<AS, TBL>(s, j, L[u], sc, bestu[u], childu[u], nearu[u], score_tbl);
        offu[u] = e <= depth && childu[u] >= 0 && !(e < depth && childu[u] == pn[e + 1]);
      }
#pragma unroll
      for (int u = 0; u < LIF; ++u) {  // the off-path children's stored records, all requested together
        cj[u] = offu[u] ? T.jpa[(childu[u]) * T.ns] : 0;
        cl[u] = offu[u] ? T.jlv[(childu[u]) * T.ns] : 0;
      }
#pragma unroll
      for (int u = 0; u < LIF; ++u) {
        const int e = base + u * nrows + row;
        if (e <= depth && j == 0) {
          bst[e] = bestu[u];
          chd[e] = childu[u];
          flg[e] = nearu[u] ? 1 : 0;
          cjp[e] = cj[u];
          cjl[e] = cl[u];
        }
      }
    }
  }
  step_barrier<LDSB>();
  MZ_JT(4)
  // -- JUMP records: a level takes its own end point (near tie, or an unexpanded best child), the stored record of
  // its off-path best child, or -- when its best child is the next level of this very path -- whatever that level
  // resolves to.  The bottom-up chain of the last case is resolved by pointer jumping (log2(depth) rounds over all
  // levels at once) instead of a serial walk by one thread; the leaf level never inherits, so every chain ends. --
  for (int e = tid; e <= depth; e += nthr) {
    const bool own = flg[e] || chd[e] < 0;
    const bool inherit = !own && e < depth && chd[e] == pn[e + 1];
    njp[e] = own ? (pn[e] | (bst[e] << 16) | (flg[e] ? (int)0x80000000 : 0)) : cjp[e];
    njl[e] = own ? e : cjl[e];
    nxa[e] = inherit ? e + 1 : e;
  }
  step_barrier<LDSB>();
  if (depth < 64) {
    // the whole path in one wavefront: lane e chases its pointer through ds_bpermute (a register exchange: no LDS
    // array, no barrier per round -- six rounds of ~100 cycles instead of six workgroup barriers)
    if (tid < 64) {
      const int e = tid <= depth ? tid : depth;
      int from = nxa[e];
      for (int span = 1; span <= depth; span <<= 1) from = __builtin_amdgcn_ds_bpermute(4 * from, from);
      if (tid <= depth) {
        T.jpa[(pn[e]) * T.ns] = njp[from];
        T.jlv[(pn[e]) * T.ns] = njl[from];
      }
    }
  } else {
    int* src = nxa;
    int* dst = nxb;
    for (int span = 1; span <= depth; span <<= 1) {
      for (int e = tid; e <= depth; e += nthr) dst[e] = src[src[e]];
      step_barrier<LDSB>();
      int* t = src; src = dst; dst = t;
    }
    for (int e = tid; e <= depth; e += nthr) {
      const int from = src[e];
      T.jpa[(pn[e]) * T.ns] = njp[from];
      T.jlv[(pn[e]) * T.ns] = njl[from];
    }
  }
  MZ_JT(5)
  if (select_next && sim + 1 < s.S) {
    step_barrier<LDSB>();  // the refreshed records (and, above, the statistics a near-tie evaluation reads) are visible
    jump_select_body<true>(s, g, T, sim + 1, r, next_action_out, next_parent_embedding_out, sel_out, depth_acc);
  }
  MZ_JT(6)
}
template <bool GUMBEL>
__global__ __launch_bounds__(1024) void jump_expand_backup_kernel(StepArgs s, JumpArgs g, int sim, const float* reward,
                                                                  const float* discount, const float* prior_logits,
                                                                  const float* value, const float* next_embedding,
                                                                  int32_t* next_action_out,
                                                                  float* next_parent_embedding_out) {
  extern __shared__ int lds_i[];
  const int r = blockIdx.x;
  jump_expand_backup_body<GUMBEL>(s, g, tree_view_global(s, g, (size_t)r * s.N), sim, r, lds_i, reward[r], discount[r], prior_logits + (size_t)r * s.A, value[r],
                                  next_embedding + (size_t)r * s.E, next_action_out != nullptr, next_action_out,
                                  next_parent_embedding_out);
}

#undef MZ_JROW_SETUP

}  // namespace mz
