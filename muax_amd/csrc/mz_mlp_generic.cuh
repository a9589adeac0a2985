// mz_mlp_generic.cuh -- the default MLP trio (muax/nn.py:59-115) with RUN-TIME shapes, and a whole search on it in one
// launch, for the shapes the fused kernel (mz_fused.cuh: tree in LDS, weights in registers, templates over every
// width) has no instance for and cannot be instantiated for: more than 8 actions, more than 127 simulations,
// embeddings wider than 64.  The reference's act() takes any of them (muax/model.py:82-96).
//
//   root:    obs -> Representation -> Prediction -> (prior_logits, value, embedding) arrays  (muax/model.py:251-263);
//            the tree kernels of the step-wise path take it from there (mzs_root / mzs_select);
//   search:  ONE launch for all simulations, one wavefront per root:  gather the parent's embedding row from the tree ->
//            Dynamic + Prediction (muax/model.py:265-282) -> next state written into the new node's row -> mctx's expand
//            / backward / next simulate with the cached decisions of mz_step_jump.cuh (tree in HBM / L2, A <= 64,
//            num_simulations <= 1023).  4096 roots are 4 wavefronts per SIMD: the memory round trips of one root's tree
//            step hide behind the other roots'.
//
// Arithmetic: the project's one spec ("MZ-F32") exactly as the oracle states it (oracle/mz_oracle.c:300-398):
// a linear layer is a k-ordered fma chain from 0 with the bias added last, ELU / exp / log / inv_scaling from mz_spec.cuh,
// every float sum 16 partials + xor butterfly (row_softmax_rt, row_sum) -- the same bits as the fused kernel and the
// oracle for any shape, which is what the tests check.
#pragma once
#include "mz_step_jump.cuh"

#pragma clang fp contract(off)

namespace mz {

struct MlpGen {
  const float *repr_w, *repr_b;
  const float *pv_w1, *pv_b1, *pv_w2, *pv_b2, *pp_w1, *pp_b1, *pp_w2, *pp_b2;
  const float *dr_w1, *dr_b1, *dr_w2, *dr_b2, *dn_w1, *dn_b1, *dn_w2, *dn_b2;
  int obs_dim, E, A, F, support, pred_on_parent;
  float discount;
};
constexpr int kGenHidden = 16;  // hk.Linear(16) everywhere in muax/nn.py:73-115

// haiku Linear, output j: dot (k-ordered fma chain from 0) then + bias; x in LDS (every lane reads the same word)
// U32: the weights indexed from the UNIFORM base with a 32-bit per-lane offset (global_load saddr + voffset) instead of
// through the per-lane pointer `w + j`.  That pointer is loop-invariant for the whole search and the compiler keeps one
// register pair per weight array alive across every simulation -- ~40 of the search kernel's 215 VGPRs; the 32-bit form
// costs an address add per load (1 - 3 % at equal occupancy) and is what lets the four-wavefront build of the search
// kernel (mz_mlp_search_kernel_occ4) get by with two dozen spilled loop invariants instead of 66 registers
template <bool U32 = false>
MZ_DEV float gen_linear(const float* x, int n_in, const float* __restrict__ w, const float* __restrict__ b, int n_out, int j) {
  // the weights of eight links are requested before the first of their fmas issues (one link at a time is one L1 / L2
  // round trip per link: ~0.5 us each on a lone chain); the chain itself stays k-ordered
  float acc = 0.0f;
  const float* wj = w + j;
  auto wt = [&](int i) { return U32 ? w[(unsigned)(i * n_out + j)] : wj[(size_t)i * n_out]; };
  int i = 0;
  for (; i + 8 <= n_in; i += 8) {
    float wv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wv[k] = wt(i + k);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = __builtin_fmaf(x[i + k], wv[k], acc);
  }
  for (; i < n_in; ++i) acc = __builtin_fmaf(x[i], wt(i), acc);
  return acc + b[j];
}
// muax/nn.py:37-44 over a vector of n floats in LDS, by one wavefront (min / max do not depend on the order)
MZ_DEV void gen_min_max_normalize(float* v, int n, int tid) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = tid; i < n; i += 64) {
    mn = fminf(mn, v[i]);
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, d));
    mx = fmaxf(mx, __shfl_xor(mx, d));
  }
  float scale = mx - mn;
  scale = scale < 1e-5f ? scale + 1e-5f : scale;
  __syncthreads();
  for (int i = tid; i < n; i += 64) v[i] = (v[i] - mn) / scale;
}
// support_to_scalar(softmax(logits[0..F))) (muax/utils.py:70-102) by the 16 lanes of one row; F in (16, 64]
MZ_DEV float gen_decode(const float* logits, int F, int support, int j) {
  float x[kMaxAS], p[kMaxAS];
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) x[t] = (j + 16 * t < F) ? logits[j + 16 * t] : 0.0f;
  row_softmax_rt(x, F, j, p);
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < kMaxAS; ++t) {
    const float term = (float)(j + 16 * t - support) * p[t];
    part = (t == 0) ? term : ((j + 16 * t < F) ? part + term : part);
  }
  return inv_scaling(row_sum(part));
}
// LDS scratch of the trio (floats): sa [E + A] | hid [32] | rl [64] | vl [64] | pl [64] | ns [E] | scal [4]
inline __host__ __device__ int gen_scratch_words(int E, int A) { return ((E + A + 3) / 4) * 4 + 32 + 3 * 64 + ((E + 3) / 4) * 4 + 4; }
struct GenLds {
  float *sa, *hid, *rl, *vl, *pl, *ns, *scal;
};
MZ_DEV GenLds gen_lds(float* f, int E, int A) {
  GenLds G;
  G.sa = f; G.hid = G.sa + ((E + A + 3) / 4) * 4; G.rl = G.hid + 32; G.vl = G.rl + 64; G.pl = G.vl + 64;
  G.ns = G.pl + 64; G.scal = G.ns + ((E + 3) / 4) * 4;
  return G;
}
// Prediction on the embedding `s` (LDS): value logits -> G.vl, prior logits -> G.pl, value -> G.scal[1]
template <bool U32 = false>
MZ_DEV void gen_prediction(const MlpGen& w, const GenLds& G, const float* s, int tid) {
  const int E = w.E, A = w.A, F = w.F;
  if (tid < 32) {
    const int u = tid & 15;
    const float a = (tid < 16) ? gen_linear<U32>(s, E, w.pv_w1, w.pv_b1, kGenHidden, u) : gen_linear<U32>(s, E, w.pp_w1, w.pp_b1, kGenHidden, u);
    G.hid[tid] = elu(a);
  }
  __syncthreads();
  for (int j = tid; j < F; j += 64) G.vl[j] = gen_linear<U32>(G.hid, kGenHidden, w.pv_w2, w.pv_b2, F, j);
  for (int j = tid; j < A; j += 64) G.pl[j] = gen_linear<U32>(G.hid + 16, kGenHidden, w.pp_w2, w.pp_b2, A, j);
  __syncthreads();
  if (tid < 16) {
    const float v = gen_decode(G.vl, F, w.support, tid);
    if (tid == 0) G.scal[1] = v;
  }
  __syncthreads();
}

// muax/model.py:251-263 for every root: one wavefront per root
__global__ __launch_bounds__(64) void mz_mlp_root_kernel(const MlpGen w, int B, const float* obs, float* prior_logits, float* value,
                                                         float* embedding) {
  extern __shared__ float gen_f[];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r >= B) return;
  const GenLds G = gen_lds(gen_f, w.E > w.obs_dim ? w.E : w.obs_dim, w.A);
  for (int i = tid; i < w.obs_dim; i += 64) G.sa[i] = obs[(size_t)r * w.obs_dim + i];
  __syncthreads();
  for (int e = tid; e < w.E; e += 64) G.ns[e] = gen_linear(G.sa, w.obs_dim, w.repr_w, w.repr_b, w.E, e);
  __syncthreads();
  gen_min_max_normalize(G.ns, w.E, tid);
  __syncthreads();
  gen_prediction(w, G, G.ns, tid);
  for (int e = tid; e < w.E; e += 64) embedding[(size_t)r * w.E + e] = G.ns[e];
  for (int j = tid; j < w.A; j += 64) prior_logits[(size_t)r * w.A + j] = G.pl[j];
  if (tid == 0) value[r] = G.scal[1];
}

// all simulations [sim_begin, sim_end) of every root (simulate() of sim_begin has run: mzs_select)
// AS (round 6, MuZero policy): the 16-lane slots the action count fills (1: A <= 16, 2: A <= 32; 0: any A <= 64) -- the tree
// step's decision refresh without the run-time tests of the general code's four slots (mz_step_jump.cuh, level_load); TBL:
// the {sqrt(n) pb_c(n), RN(1 / n)} table behind the trio's scratch (visit counts up to 1030: Markstein's checked range)
template <bool GUMBEL, int AS, bool TBL, bool U32>
MZ_DEV void mlp_search_body(const StepArgs& s, const JumpArgs& g, const MlpGen& w, int sim_begin, int sim_end) {
  extern __shared__ int gen_i[];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r >= s.B) return;
  const int N = s.N, A = s.A, E = s.E;
  const size_t rb = (size_t)r * N;
  int* tree_lds = gen_i;
  const GenLds G = gen_lds(reinterpret_cast<float*>(gen_i + 15 * (N + 1)), E, A);
  float* score_tbl = nullptr;
  if constexpr (TBL) {
    score_tbl = reinterpret_cast<float*>(gen_i + 15 * (N + 1)) + gen_scratch_words(E, A);
    for (int n = tid; n < N + 1; n += 64) {
      score_tbl[2 * n] = puct_scale(n, s.pb_c_init, s.pb_c_base);
      score_tbl[2 * n + 1] = n > 0 ? 1.0f / (float)n : 0.0f;
    }
    __syncthreads();
  }
  TreeView T = tree_view_global(s, g, rb);
  if constexpr (AS != 0) {  // the root's invalid-action mask once per launch, not a load in front of level 0's scores every time
    T.inv_bits = 0;
#pragma unroll
    for (int t = 0; t < AS; ++t) {
      const int a = (tid & 15) + 16 * t;
      if (a < A && s.root_invalid[(size_t)r * A + a]) T.inv_bits |= 1 << t;
    }
  }
  int parent = s.sel_parent[r], action = s.sel_action[r], depth = s.sel_depth[r];
  int newn;
  {
    const int next = s.children_index[(rb + parent) * A + action];
    newn = next == -1 ? sim_begin + 1 : next;
  }
  for (int sim = sim_begin; sim < sim_end; ++sim) {
    // recurrent_fn: Dynamic on [s, onehot(a)], Prediction on the next state (or the parent's, pip-release quirk)
    const float* prow = s.embeddings + (rb + parent) * E;
    for (int i = tid; i < E; i += 64) G.sa[i] = prow[i];
    for (int k = tid; k < A; k += 64) G.sa[E + k] = (k == action) ? 1.0f : 0.0f;
    __syncthreads();
    if (tid < 32) {
      const int u = tid & 15;
      const float a = (tid < 16) ? gen_linear<U32>(G.sa, E + A, w.dr_w1, w.dr_b1, kGenHidden, u)
                                 : gen_linear<U32>(G.sa, E + A, w.dn_w1, w.dn_b1, kGenHidden, u);
      G.hid[tid] = elu(a);
    }
    __syncthreads();
    for (int j = tid; j < w.F; j += 64) G.rl[j] = gen_linear<U32>(G.hid, kGenHidden, w.dr_w2, w.dr_b2, w.F, j);
    for (int e = tid; e < E; e += 64) G.ns[e] = gen_linear<U32>(G.hid + 16, kGenHidden, w.dn_w2, w.dn_b2, E, e);
    __syncthreads();
    gen_min_max_normalize(G.ns, E, tid);
    __syncthreads();
    float* nrow = s.embeddings + (rb + newn) * E;
    for (int e = tid; e < E; e += 64) nrow[e] = G.ns[e];
    if (tid >= 48) {  // the reward decode on the last row while rows 0 / 1 start the prediction net
      const float rw = gen_decode(G.rl, w.F, w.support, tid & 15);
      if (tid == 48) G.scal[0] = rw;
    }
    gen_prediction<U32>(w, G, w.pred_on_parent ? G.sa : G.ns, tid);
    const float rew = G.scal[0], val = G.scal[1];
    const int known[4] = {parent, action, depth, newn};
    int sel[3] = {0, 0, 0};
    jump_expand_backup_body<GUMBEL, kLevelsInFlight, AS, TBL>(s, g, T, sim, r, tree_lds, rew, w.discount, G.pl, val, nullptr, true, nullptr,
                                                              nullptr, sel, known, 0, score_tbl);
    if (sim + 1 < sim_end && sim + 1 < s.S) {
      parent = sel[0];
      action = sel[1];
      depth = sel[2];
      const int next = s.children_index[(rb + parent) * A + action];
      newn = next == -1 ? sim + 2 : next;
    }
    __syncthreads();
  }
}
// As compiled the search needs 215 .. 225 VGPRs = TWO wavefronts per SIMD: 2048 roots resident, 4096 roots two rounds (per
// simulation 18.6 us at 1024 roots, 21.7 at 2048, 41.9 at 4096).  _occ4: the same body held to 128 registers (four
// wavefronts per SIMD; ~25 loop invariants spilled to scratch) for launches of more roots than two wavefronts per SIMD
// hold whose workgroups fit sixteen to a CU's LDS -- 1.2 - 1.3x there, slower everywhere else (the host chooses:
// mz_api.hip, launch_mlp_search; both measured in profiles/r06_generic_notes.txt).
template <bool GUMBEL, int AS = 0, bool TBL = false>
__global__ __launch_bounds__(64) void mz_mlp_search_kernel(const StepArgs s, const JumpArgs g, const MlpGen w, int sim_begin,
                                                           int sim_end) {
  mlp_search_body<GUMBEL, AS, TBL, false>(s, g, w, sim_begin, sim_end);
}
template <bool GUMBEL, int AS = 0, bool TBL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void mz_mlp_search_kernel_occ4(
    const StepArgs s, const JumpArgs g, const MlpGen w, int sim_begin, int sim_end) {
  mlp_search_body<GUMBEL, AS, TBL, true>(s, g, w, sim_begin, sim_end);
}

}  // namespace mz
