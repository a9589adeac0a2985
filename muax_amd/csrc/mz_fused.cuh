// mz_fused.cuh -- the whole MuZero.act() search for the default MLP trio
// (reference path: muax/model.py:222-282 -> mctx.muzero_policy; nets
// muax/nn.py:59-115; codec muax/utils.py:70-102) in ONE launch:
// root inference, S simulations, summary + sampling.
//
// Mapping of the search kernel (MI355X-first, not a translation of mctx's
// vmapped XLA program).  Measured on gfx950 (tools/ubench_lat.hip): a lone wavefront
// issues ONE instruction of any kind per ~4.8 cycles, 8.3 when it depends on the one
// before, ~20 per dependent DPP step and ~64 per dependent LDS read -- and 4096 roots
// are exactly one wavefront per SIMD -- so the design minimises the number of
// instructions of the per-root chain and how many of them wait:
//   * one search root = one DPP row (16 lanes); 4 roots per wavefront, one
//     wavefront per SIMD, no barrier after the prologue;
//   * the root's whole tree lives in LDS for the duration of the act; HBM is
//     touched for the observation, the weights (once, into VGPRs) and the outputs;
//   * pUCT decisions are CACHED per node: a node's scores only change when the
//     node lies on a backed-up path, so they are recomputed in the backup
//     phase, where lane e owns path entry e (all levels in parallel).  mctx's
//     tie-break noise is < 1e-7, so a decision whose runner-up satisfies
//     fl(score + 1e-7) < best is provably independent of the noise (rounding is
//     monotone).  Only near-tie nodes evaluate score + noise, drawing JAX's
//     threefry stream on demand (lazy key walk per simulation) -- bit-identical
//     to drawing it at every level;
//   * selection does not walk level by level: every node carries a JUMP word,
//     the end point (parent, action, depth) of the greedy descent below it,
//     valid because sub-trees off the backed-up path never change; the path
//     nodes' words are refreshed with a log-step DPP scan.  Every node also
//     stores its own root path as packed bytes (written once at expansion), so
//     the backup lanes find their entries without a walk.  The root's word lives
//     in a register (the backup ends by producing it): a simulation's selection
//     reads NOTHING unless a row meets a near tie or the depth limit, and those
//     cases sit behind one wave-uniform branch;
//   * backup is lane-parallel: the discounted-return chain advances all path
//     entries at once (G[e] = r[e] + g G[e+1] through a row_shl:1 DPP operand),
//     running means and prior / (visits + 1) divide by small integers with a
//     correctly rounded reciprocal from an LDS table (Markstein, 3 ops);
//   * the node record has an odd word stride (16 records of a row in 16 banks),
//     embeddings move to HBM when E > 16 so that 16 roots still fit a CU;
//   * the MLPs run as row-distributed fma chains: input element i lives in lane
//     i&15 (slot i>>4) and is fetched with a row_newbcast DPP modifier -- fused
//     into v_fmac_f32_dpp where the chain is scalar, a mov feeding v_pk_fma_f32
//     where two chains share the input; each lane keeps its own column of every
//     weight matrix in VGPRs;
//   * quotients that share a denominator (softmax terms, value scores) divide
//     through ONE refined reciprocal in packed form, behind a wave-uniform range
//     test with the IEEE division as the other branch (mz_spec.cuh).
#pragma once
#include "mz_spec.cuh"

#pragma clang fp contract(off)

namespace mz {

// opt-in phase timers (tools/profile_phases.py builds with -DMZ_PROFILE); no code otherwise
#ifdef MZ_PROFILE
#define MZ_TICK(slot)                                  \
  do {                                                 \
    uint64_t now_ = __builtin_amdgcn_s_memtime();      \
    prof_acc[slot] += now_ - prof_t;                   \
    prof_t = now_;                                     \
  } while (0)
// sub-phase tick: drains outstanding LDS/scalar traffic first so the time lands in the right slot
#define MZ_TICKW(slot)                                 \
  do {                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    MZ_TICK(slot);                                     \
  } while (0)
#else
#define MZ_TICK(slot) do {} while (0)
#define MZ_TICKW(slot) do {} while (0)
#endif

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
  float* emb_scratch;  // [B][S+1][E], instances with the embeddings out of LDS and no export
  int32_t* path_scratch;  // [B][S+1][PATHW], instances with the root paths out of LDS (FusedCfg::PH)
  // scalars
  int32_t B, obs_dim, S, max_depth, support, F, pred_on_parent, export_tree;  // F = 2 support + 1
  float pb_c_init, pb_c_base, dirichlet_fraction, discount, temperature;
  uint64_t global_batch, root_offset;
  uint32_t k_sample[2];
  // gumbel policy (MODE >= 2): root Gumbel noise comes from `gumbel` or from k_gumbel
  const int32_t* visit_table;      // [(max_considered + 1), S] seq_halving table
  int32_t max_considered;
  float gumbel_scale;
  uint32_t k_gumbel[2];
  uint32_t sim_keys[kMaxSims][2];  // simulate_key of every simulation (mctx search body_fun)
  uint64_t* prof;  // MZ_PROFILE builds only: [waves][8] cycle counters
  int32_t path_words;  // words per node `path_scratch` was allocated with (an instance needs FusedCfg::PATHW)
};

// MODE: 0 muzero policy without tie-break noise, 1 muzero policy with mctx's tie-break noise,
//       2 gumbel policy + qtransform_by_parent_and_siblings, 3 gumbel policy + completed_by_mix_value
// FS_: 16-lane slots of the support logits (2: F = 2 support + 1 in 17..32, 4: 33..64); F itself is a run-time
// parameter (support_size is a constructor argument of the reference, muax/model.py:48-49)
// PH_ ("paths in HBM"): the compact record for launches with MORE workgroups than CUs.  A node's root path -- a
// third of the record -- moves to global memory ([root][node][PATHW] words: read once per simulation, of the parent,
// which is known when the simulation starts, so the load runs behind the network pass; written once per node) and the
// raw value leaves the header where no decision reads it: 16 roots then need < 80 KiB of LDS, TWO workgroups share a
// CU and every SIMD has two wavefronts whose issue latencies and LDS round trips overlap.
// LONG (REC_ = 2): long searches / wide action sets.  A node's root path is up to 128 words (255 simulations)
// -- 256 of them per root cannot live in LDS, and from ~100 simulations on they are most of the record -- so the paths
// take the compact record's place in HBM (up to eight words per lane of the node's row instead of one), while the
// launch shape stays the plain one (as many roots per workgroup as the LDS holds records for, any policy); the
// dispatcher takes such an instance whatever the batch size.
// REC_: 0 the plain record (paths in LDS), 1 (`true` in an instance list) the compact record, 2 LONG
template <int A_, int E_, int FS_, int NMAX_, int MODE_, int WAVES_ = 4, int REC_ = 0>
struct FusedCfg {
  static_assert(REC_ >= 0 && REC_ <= 2, "record kind");
  static constexpr bool PH_ = REC_ != 0;
  static constexpr bool PH = PH_;
  static constexpr bool LONG = REC_ == 2;
  static constexpr int WAVES = WAVES_, THREADS = 64 * WAVES_;
  static constexpr int A = A_, E = E_, FS = FS_, NMAX = NMAX_, MODE = MODE_;
  static_assert(FS_ == 2 || FS_ == 4, "support logits are handled as one or two packed pairs of lane slots");
  static constexpr bool TB = MODE_ == 1;
  static constexpr bool GUMBEL = MODE_ >= 2;
  static constexpr int QT = MODE_ == 3 ? 1 : 0;
  static constexpr int H = kHidden;
  static constexpr int ES = (E + 15) / 16;
  // first layers over a long embedding: two v_fmac_f32_dpp chains per input instead of broadcast move + packed fma
  // (under register pressure the compiler funnels every broadcast through one temporary and pads each term)
  static constexpr bool L1_DPP = E_ > 16 && E_ % 8 == 0;  // (eight inputs per asm statement; other widths take the packed-fma chains)
  // ... and the second layers as blocks of v_fmac_f32_dpp chains (two logit slots, two state slots, one policy slot)
  static constexpr bool L2_BLOCK = L1_DPP && FS_ == 2 && (E_ + 15) / 16 == 2 && A_ <= 16;
  // ---- node record in LDS (32-bit words, 16-byte aligned) ----
  //   [SEL0  ..) A x {child index, cached pUCT score}   (4-byte aligned only: the stride is odd)
  //   [HDR0  ..) visits, value, JUMP word, raw value
  //   [ST0   ..) A x {prob, value, visits, reward[, prior logit (Gumbel)]}; children_discounts is the
  //              constant discount on expanded edges and multiplies a zero value on the others
  //   [EMB0  ..) embedding
  //   [PATH0 ..) this node's own root path, one packed (node, action) entry per level
  // JUMP word: end point of the greedy descent below this node:
  //   parent[0:12) | action[12:16) | depth of parent[16:24) | bit 31: the end point is a near tie
  // PK ("packed", the compact record of the instances whose embeddings are in HBM as well, E > 16): child indices
  // and child visit counts are BYTES (one word per node each: nodes and counts are below 256), the two first-layer
  // weight matrices live in LDS, shared by the workgroup, instead of in 136 registers per lane -- 21 words per node
  // for four actions, 78 KB per 16-root workgroup with the weights, at most 256 registers per lane: TWO workgroups
  // share a CU (8192 LunarLander roots are then ONE round of workgroups instead of two):
  //   [SEL0 ..) child index bytes | A cached pUCT scores   [HDR0 ..) visits, value, JUMP
  //   [ST0  ..) A probs | A values | A rewards | child visit bytes
  static constexpr bool PK = PH_ && E_ > 16 && !LONG;
  static_assert(!PK || (A_ <= 4 && NMAX_ <= 128 && MODE_ < 2), "packed record: four byte-sized children, MuZero policy");
  static constexpr int SEL0 = 0, SELW = PK ? 1 + A : ((2 * A + 3) / 4) * 4;
  static constexpr int HDR0 = SELW, JUMP = HDR0 + 2;
  // raw_values are read inside the kernel only by qtransform_completed_by_mix_value (MODE 3); the compact record
  // drops the word elsewhere (an export writes raw values straight to the caller's array)
  static constexpr bool RAW_OK = !(PH_ && MODE_ != 3);
  static constexpr int HDRW = RAW_OK ? 4 : 3;
  static constexpr int ST0 = HDR0 + HDRW, STW = MODE_ >= 2 ? 5 : 4, ST_LOGIT = 4;
  // embeddings: in the LDS record while 16 roots per workgroup still fit the CU's LDS with them, else in
  // HBM ([root][node][E], one coalesced E*4-byte row per access, L2-resident while the root is active)
  // (LONG instances keep them in HBM whatever their width: the record without the embedding lets twice as many roots
  // of a 128 .. 255-simulation search share a CU -- 23 -> 15 words per node for CartPole's shape)
  static constexpr bool EMB_LDS = E_ <= 16 && !LONG;
  static constexpr int EMB0 = PK ? ST0 + 3 * A + 1 : ST0 + STW * A;
  // field offsets inside a record (a: child)
  static constexpr int off_score(int a) { return PK ? SEL0 + 1 + a : SEL0 + 2 * a + 1; }
  static constexpr int off_prob(int a) { return PK ? ST0 + a : ST0 + STW * a; }
  static constexpr int off_val(int a) { return PK ? ST0 + A + a : ST0 + STW * a + 1; }
  static constexpr int off_rew(int a) { return PK ? ST0 + 2 * A + a : ST0 + STW * a + 3; }
  static constexpr int off_logit(int a) { return ST0 + STW * a + ST_LOGIT; }  // Gumbel modes (never packed)
  static constexpr int VIS0 = ST0 + 3 * A;  // packed record: the word of child visit bytes
  // children_index / children_visits of a node (ndi = the node's record)
  static MZ_DEV void load_cidx(const int* ndi, int (&cidx)[A]) {
    if constexpr (PK) {
      const int w = ndi[SEL0];
#pragma unroll
      for (int a = 0; a < A; ++a) cidx[a] = (w << (24 - 8 * a)) >> 24;  // sign-extending byte: 0xff = -1 (unvisited)
    } else {
#pragma unroll
      for (int a = 0; a < A; ++a) cidx[a] = ndi[SEL0 + 2 * a];
    }
  }
  static MZ_DEV int load_cidx1(const int* ndi, int a) {
    if constexpr (PK) return reinterpret_cast<const int8_t*>(ndi + SEL0)[a];
    else return ndi[SEL0 + 2 * a];
  }
  static MZ_DEV void store_cidx(int* ndi, int a, int v) {
    if constexpr (PK) reinterpret_cast<int8_t*>(ndi + SEL0)[a] = (int8_t)v;
    else ndi[SEL0 + 2 * a] = v;
  }
  static MZ_DEV void load_vis(const int* ndi, int (&vis)[A]) {
    if constexpr (PK) {
      const unsigned w = (unsigned)ndi[VIS0];
#pragma unroll
      for (int a = 0; a < A; ++a) vis[a] = (int)((w >> (8 * a)) & 0xffu);
    } else {
#pragma unroll
      for (int a = 0; a < A; ++a) vis[a] = ndi[ST0 + STW * a + 2];
    }
  }
  static MZ_DEV void store_vis(int* ndi, int a, int v) {
    if constexpr (PK) reinterpret_cast<uint8_t*>(ndi + VIS0)[a] = (uint8_t)v;
    else ndi[ST0 + STW * a + 2] = v;
  }
  // first-layer weights in LDS (packed record): per matrix [inputs / 2][16 lanes][2 inputs x (net 0, net 1)]
  static constexpr int W1P_WORDS = PK ? (E / 2) * 16 * 4 : 0;          // Prediction: (value net, policy net)
  static constexpr int W1D_WORDS = PK ? ((E + A + 1) / 2) * 16 * 4 : 0;  // Dynamic: (reward net, state net), E + A rows
  static constexpr int WLDS_WORDS = W1P_WORDS + W1D_WORDS;
  // a path entry packs (node, action): one byte while ceil(log2 NMAX) + ceil(log2 A) <= 8, else 16 bits
  static constexpr int NODE_BITS = ceil_log2(NMAX), ACT_BITS = ceil_log2(A) < 1 ? 1 : ceil_log2(A);
  static constexpr int ENTRY_BITS = (NODE_BITS + ACT_BITS <= 8) ? 8 : 16;
  static constexpr int ENTRY_ACT_SHIFT = ENTRY_BITS == 8 ? NODE_BITS : 12;
  static constexpr int PATH0 = EMB0 + (EMB_LDS ? E : 0);
  static constexpr int PATHW = (NMAX * ENTRY_BITS + 31) / 32;
  // odd record stride: lane e of the backup reads node(e)'s record, and with an odd stride the 16 records
  // of a row start in 16 different LDS banks (a stride of 40 words put them in 4)
  static constexpr int NS = (PATH0 + (PH_ ? 0 : PATHW)) | 1;
  static constexpr int TREE_WORDS = ((NS * NMAX + 3) / 4) * 4;
  static constexpr int PATH_WORDS = 0;
  // root Gumbel noise (gumbel policy): in the root's own, empty, path slot -- or behind the tree when paths are in HBM
  static constexpr int NOISE_WORDS = (PH_ && MODE_ >= 2) ? ((A + 3) / 4) * 4 : 0;
  static constexpr int GUM0 = PH_ ? TREE_WORDS : PATH0;
  static_assert(NMAX <= 4096 && A <= 16, "JUMP word fields");
  static_assert(PH_ || A <= PATHW, "the root's (empty) path slot holds its Gumbel noise");
  static constexpr int PATHS = (PATHW + 15) / 16;  // path words per lane when a node's path is copied
  static_assert(PATHS <= (LONG ? 8 : 4), "a node's path is copied by the 16 lanes of its row");
  // the four roots of a wave start 8 banks apart: row-uniform reads of the same field of four trees
  // (selection, expansion) then hit four different banks
  static constexpr int pad_root(int w) { return w + ((8 - w % 32 + 32) % 32); }
  static constexpr int ROOT_WORDS = pad_root(TREE_WORDS + PATH_WORDS + NOISE_WORDS);
  static constexpr int ROOTS_PER_WG = 4 * WAVES;
  static constexpr int TBL_WORDS = 2 * (((NMAX + 2 + 3) / 4) * 4);  // {sqrt(n) pb_c(n), 1/n} pairs
  static constexpr int LDS_BYTES = 4 * (TBL_WORDS + WLDS_WORDS + ROOTS_PER_WG * ROOT_WORDS);
  static_assert(LDS_BYTES <= 160 * 1024, "tree does not fit the 160 KiB LDS of a CU: lower WAVES");
  static_assert(!PH_ || LONG || (2 * LDS_BYTES <= 160 * 1024 && WAVES_ == 4 && (PATHW + 15) / 16 == 1),
                "compact record: two 16-root workgroups per CU, one path word per lane");
  static_assert(NMAX <= 256, "JUMP word: the depth of an end point is a byte; the argument block holds 256 simulation keys");
  static_assert(A <= 16, "selection keeps all A scores in registers; a child's action is four bits of a JUMP word");
  static_assert(E <= 32 * 16, "row-distributed vectors");
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
  MZ_DEV void apply(const float (&x)[IS], float (&y)[OS]) const {
#pragma unroll
    for (int t = 0; t < OS; ++t) y[t] = 0.0f;
    StaticFor<0, NIN>::run([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float xb = bcast<(i & 15)>(x[i >> 4]);
#pragma unroll
      for (int t = 0; t < OS; ++t) y[t] = __builtin_fmaf(xb, w[i][t], y[t]);
    });
#pragma unroll
    for (int t = 0; t < OS; ++t) y[t] = y[t] + b[t];
  }
};

// the same with a run-time number of outputs (the support logits): OS lane slots of capacity
template <int NIN, int OS>
struct RowLinearRT {
  float w[NIN][OS];
  float b[OS];
  MZ_DEV void load(const float* __restrict__ W, const float* __restrict__ Bv, int j, int nout) {
#pragma unroll
    for (int t = 0; t < OS; ++t) {
      int k = j + 16 * t;
      b[t] = k < nout ? Bv[k] : 0.0f;
#pragma unroll
      for (int i = 0; i < NIN; ++i) w[i][t] = k < nout ? W[i * nout + k] : 0.0f;
    }
  }
};

// first layer of Dynamic: input [s, onehot(a)] (muax/nn.py:104-110).  The one-hot rows are E..E+A-1 of W;
// zero terms of the chain are exact no-ops, so the chain is "s terms, then + W[E+a]".  The two first layers
// (reward net, next-state net) share their input: their weights are kept as (reward, state) pairs from the
// start, the operand form of v_pk_fma_f32
template <int E, int A>
struct RowLinearOneHot2 {
  f32x2 w[E];
  f32x2 wa[A];
  f32x2 b;
  MZ_DEV void load(const float* __restrict__ W0, const float* __restrict__ B0, const float* __restrict__ W1,
                   const float* __restrict__ B1, int j) {
    b = (f32x2){B0[j], B1[j]};
    StaticFor<0, E>::run([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      w[i] = (f32x2){W0[i * kHidden + j], W1[i * kHidden + j]};
    });
    StaticFor<0, A>::run([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      wa[a] = (f32x2){W0[(E + a) * kHidden + j], W1[(E + a) * kHidden + j]};
    });
  }
};

// first layers of Prediction (value net, policy net): (value, policy) weight pairs
template <int E>
struct RowLinearPair {
  f32x2 w[E];
  f32x2 b;
  MZ_DEV void load(const float* __restrict__ W0, const float* __restrict__ B0, const float* __restrict__ W1,
                   const float* __restrict__ B1, int j) {
    b = (f32x2){B0[j], B1[j]};
    StaticFor<0, E>::run([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      w[i] = (f32x2){W0[i * kHidden + j], W1[i * kHidden + j]};
    });
  }
};

// jax.nn.softmax over a row-distributed vector of N elements (N <= 32).  For
// N < 16 the result is only valid in lanes < 2^ceil(log2 N) (all that is used).
template <int N>
MZ_DEV void row_softmax(const float (&x)[(N + 15) / 16], int j, float (&p)[(N + 15) / 16]) {
  constexpr int NSLOT = (N + 15) / 16;
  constexpr int STEPS = N >= 16 ? 4 : ceil_log2(N);
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) m = (j + 16 * t < N) ? fmaxf(m, x[t]) : m;
  m = row_max<STEPS>(m);
  float e[NSLOT];
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    bool ok = j + 16 * t < N;
    e[t] = ok ? exp_neg(x[t] - m) : 0.0f;
    part = (t == 0) ? e[0] : (ok ? part + e[t] : part);
  }
  // lanes >= N hold +0: the butterfly steps that would only add those zeros are exact no-ops
  float s = row_sum_steps<STEPS>(part);
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) p[t] = e[t] / s;
}

// support_to_scalar(softmax(logits)) (muax/utils.py:94-102, muax/model.py:254,273-274), F = 2 support + 1 > 16
// logits in FS lane slots
template <int FS>
MZ_DEV float row_decode(const float (&logits)[FS], int j, int support, int F) {
  bool ok[FS];
#pragma unroll
  for (int t = 0; t < FS; ++t) ok[t] = j + 16 * t < F;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < FS; ++t) m = ok[t] ? fmaxf(m, logits[t]) : m;
  m = row_max<4>(m);
  float e[FS];
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < FS; ++t) {
    e[t] = ok[t] ? exp_neg(logits[t] - m) : 0.0f;
    part = (t == 0) ? e[0] : (ok[t] ? part + e[t] : part);
  }
  const float s = row_sum(part);
  float tpart = 0.0f;
#pragma unroll
  for (int t = 0; t < FS; ++t) {
    const float term = (float)(j + 16 * t - support) * (e[t] / s);
    tpart = (t == 0) ? (ok[0] ? term : 0.0f) : (ok[t] ? tpart + term : tpart);
  }
  return inv_scaling(row_sum(tpart));
}

// muax/nn.py:37-44 over a row-distributed vector
template <int E>
MZ_DEV void row_min_max_normalize(float (&s)[(E + 15) / 16], int j) {
  constexpr int NSLOT = (E + 15) / 16;
  constexpr int STEPS = E >= 16 ? 4 : ceil_log2(E);
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    bool ok = j + 16 * t < E;
    mn = ok ? fminf(mn, s[t]) : mn;
    mx = ok ? fmaxf(mx, s[t]) : mx;
  }
  // lanes >= E hold the neutral element; lanes < 2^STEPS agree after STEPS steps
  mn = row_min<STEPS>(mn);  // (lanes >= 2^STEPS see garbage; their slots are never used)
  mx = row_max<STEPS>(mx);
  float scale = mx - mn;
  scale = scale < 1e-5f ? scale + 1e-5f : scale;
  if constexpr (NSLOT == 2) {
    // the two quotients share their denominator (>= 1e-5): one refined reciprocal, the pair in packed form (div_newton2, as
    // the value scores of puct_scores) -- valid while every numerator is 0 or >= 2^-100 and the scale is below 2^41
    // (mzs_selftest's range); a wave-uniform test, the IEEE divisions as the other branch.  4384 -> 3936 cycles per pass
    // of the E = 32 instance, same bits (tools/ubench_netpass_e32.hip).  (A SINGLE quotient behind such a test is slower
    // than its IEEE expansion: one-slot embeddings and the policy softmax keep the division, 1958 against 2244 cycles
    // per CartPole pass.)
    const float n0 = s[0] - mn, n1 = s[1] - mn;
    const uint32_t low = min(f2u(n0) - 1u, f2u(n1) - 1u);  // 0 wraps to the top: only (0, 2^-100) fails the test
    const bool risky = low < f2u(0x1p-100f) - 1u || !(scale < 0x1p41f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(risky) == 0, 1)) {
      const float y0 = __builtin_amdgcn_rcpf(scale);
      const float y = __builtin_fmaf(__builtin_fmaf(-scale, y0, 1.0f), y0, y0);
      const f32x2 q = div_newton2((f32x2){n0, n1}, splat2(scale), splat2(y));
      s[0] = q.x;
      s[1] = q.y;
      return;
    }
  }
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) s[t] = (s[t] - mn) / scale;
}

// first layers whose matrices are in LDS (packed record): only the (net 0, net 1) bias pair stays in registers
struct RowBias2 {
  f32x2 b;
  MZ_DEV void load(const float* __restrict__, const float* __restrict__ B0, const float* __restrict__,
                   const float* __restrict__ B1, int j) {
    b = (f32x2){B0[j], B1[j]};
  }
};

template <bool COND, class T, class F> struct PickType { using type = T; };
template <class T, class F> struct PickType<false, T, F> { using type = F; };

template <class C>
struct Nets {
  typename PickType<C::PK, RowBias2, RowLinearPair<C::E>>::type p1;  // (pv1, pp1)
  RowLinearRT<kHidden, C::FS> pv2;
  RowLinear<kHidden, C::A> pp2;
  typename PickType<C::PK, RowBias2, RowLinearOneHot2<C::E, C::A>>::type d1;  // (dr1, dn1)
  RowLinearRT<kHidden, C::FS> dr2;
  RowLinear<kHidden, C::E> dn2;
  const float* wlds = nullptr;  // packed record: this lane's column of the two LDS matrices (Prediction, then Dynamic)

  // packed record: the workgroup's copy of the first-layer matrices, [input pair][lane][2 inputs x (net 0, net 1)]
  // (one ds_read_b128 per lane fetches two links of the packed chain; the rows of a wave read the same addresses)
  static MZ_DEV void fill_lds(const FusedParams& p, float* wl, int tid) {
    if constexpr (C::PK) {
      constexpr int E = C::E, RD = E + C::A;
      for (int idx = tid; idx < (E / 2) * 16; idx += C::THREADS) {
        const int i = 2 * (idx >> 4), jj = idx & 15;
        *reinterpret_cast<float4*>(wl + 4 * idx) = make_float4(p.pv_w1[i * kHidden + jj], p.pp_w1[i * kHidden + jj],
                                                               p.pv_w1[(i + 1) * kHidden + jj], p.pp_w1[(i + 1) * kHidden + jj]);
      }
      float* wd = wl + C::W1P_WORDS;
      for (int idx = tid; idx < ((RD + 1) / 2) * 16; idx += C::THREADS) {
        const int i = 2 * (idx >> 4), jj = idx & 15;
        const bool two = i + 1 < RD;
        *reinterpret_cast<float4*>(wd + 4 * idx) = make_float4(p.dr_w1[i * kHidden + jj], p.dn_w1[i * kHidden + jj],
                                                               two ? p.dr_w1[(i + 1) * kHidden + jj] : 0.0f,
                                                               two ? p.dn_w1[(i + 1) * kHidden + jj] : 0.0f);
      }
    }
  }
  template <int WHICH>  // 0: Prediction's pair of first layers, 1: Dynamic's (state rows)
  MZ_DEV f32x2 first_layers_lds(const float (&x)[C::ES]) const {
    static_assert(C::E % 8 == 0, "eight inputs per statement");
    const float* base = wlds + (WHICH ? C::W1P_WORDS : 0);
    float h0 = 0.0f, h1 = 0.0f;
    StaticFor<0, C::E / 8>::run([&](auto ic) {
      constexpr int i = 8 * decltype(ic)::value;
      f32x2 w[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(base + (i / 2 + q) * 64);
        w[2 * q] = (f32x2){v.x, v.y};
        w[2 * q + 1] = (f32x2){v.z, v.w};
      }
      fmac_bcast_pair8<(i & 15), (i & 15) == 0>(h0, h1, x[i >> 4], w);
    });
    return (f32x2){h0, h1};
  }
  MZ_DEV f32x2 first_p(const float (&x)[C::ES]) const {
    if constexpr (C::PK) return first_layers_lds<0>(x);
    else return first_layers(x, p1.w);
  }
  MZ_DEV f32x2 first_d(const float (&x)[C::ES]) const {
    if constexpr (C::PK) return first_layers_lds<1>(x);
    else return first_layers(x, d1.w);
  }

  MZ_DEV void load(const FusedParams& p, int j) {
    p1.load(p.pv_w1, p.pv_b1, p.pp_w1, p.pp_b1, j);
    pv2.load(p.pv_w2, p.pv_b2, j, p.F); pp2.load(p.pp_w2, p.pp_b2, j);
    d1.load(p.dr_w1, p.dr_b1, p.dn_w1, p.dn_b1, j);
    dr2.load(p.dr_w2, p.dr_b2, j, p.F); dn2.load(p.dn_w2, p.dn_b2, j);
  }
  // x . (W0, W1) for a row-distributed x: the two first layers that share an input, as (net 0, net 1) pairs
  template <class W>
  MZ_DEV f32x2 first_layers(const float (&x)[C::ES], const W& w) const {
    if constexpr (C::L1_DPP) {
      static_assert(C::E % 8 == 0, "eight inputs per statement");
      float h0 = 0.0f, h1 = 0.0f;
      StaticFor<0, C::E / 8>::run([&](auto ic) {
        constexpr int i = 8 * decltype(ic)::value;
        fmac_bcast_pair8<(i & 15), (i & 15) == 0>(h0, h1, x[i >> 4], &w[i]);
      });
      return (f32x2){h0, h1};
    } else {
      f32x2 h = splat2(0.0f);
      StaticFor<0, C::E>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        h = fma2(splat2(bcast<(i & 15)>(x[i >> 4])), w[i], h);
      });
      return h;
    }
  }
  // Prediction (muax/nn.py:73-90) + value decode
  MZ_DEV void predict(const float (&s)[C::ES], int j, int support, int F, float& value,
                      float& pi_logit) const {
    // same packed chains as forward() below: every weight register then has ONE pairing in the whole
    // kernel (a second, scalar use made the compiler re-pair them through scratch memory)
    constexpr int NP = C::FS / 2;
    f32x2 g = first_p(s);
    g = elu2(g + p1.b);
    f32x2 vl[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) vl[q] = splat2(0.0f);
    float pl = 0.0f;
    StaticFor<0, kHidden>::run([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const f32x2 gv = splat2(bcast<i>(g.x));
#pragma unroll
      for (int q = 0; q < NP; ++q) vl[q] = fma2(gv, (f32x2){pv2.w[i][2 * q], pv2.w[i][2 * q + 1]}, vl[q]);
      fmac_bcast<i, i == 0>(pl, g.y, pp2.w[i][0]);
    });
    pi_logit = pl + pp2.b[0];
    float v_logits[C::FS];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      vl[q] = vl[q] + (f32x2){pv2.b[2 * q], pv2.b[2 * q + 1]};
      v_logits[2 * q] = vl[q].x;
      v_logits[2 * q + 1] = vl[q].y;
    }
    value = row_decode<C::FS>(v_logits, j, support, F);
  }
  // ---- the per-simulation pass: Dynamic (muax/nn.py:93-115) on (s, action), Prediction
  // (muax/nn.py:73-90) on the child (or parent) embedding, both support decodes.  Same arithmetic
  // as predict() above (and a plain layer-by-layer Dynamic), organised for the machine: the two hidden layers that share an
  // input run as ONE packed chain ((reward-net, state-net), (value-net, policy-net)), 2-slot logits
  // are packed, and the reward and value decodes advance together as (reward, value) pairs. ----
  MZ_DEV void forward(const float (&s)[C::ES], int action, int j, int support, int F, bool pred_on_parent,
                      float& reward, float& value, float& pi_logit, float& pi_prob,
                      float (&ns)[C::ES]) const {
    constexpr int E = C::E, A = C::A, FS = C::FS, NP = C::FS / 2;
    // Dynamic, first layer: [s, onehot(a)] -> 16 hidden units of the reward net and of the state net
    f32x2 h = first_d(s);
    {
      f32x2 wsel;
      if constexpr (C::PK) {
        // row E + action of the LDS matrix (one 8-byte read at a row-uniform address)
        const int row = E + action;
        const float2 v = *reinterpret_cast<const float2*>(wlds + C::W1P_WORDS + (row >> 1) * 64 + 2 * (row & 1));
        wsel = (f32x2){v.x, v.y};
      } else {
        wsel = d1.wa[0];
        StaticFor<1, A>::run([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          wsel = (action == a) ? d1.wa[a] : wsel;
        });
      }
      h = (h + wsel) + d1.b;
    }
    h = elu2(h);
    // second layer: reward logits (FS slots, packed in pairs) and next state
    f32x2 rl[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) rl[q] = splat2(0.0f);
#pragma unroll
    for (int t = 0; t < C::ES; ++t) ns[t] = 0.0f;
    if constexpr (C::L2_BLOCK) {
      // (two logit slots + two state slots: four v_fmac_f32_dpp chains, four inputs per statement)
      float r0 = 0.0f, r1 = 0.0f;
      StaticFor<0, kHidden / 4>::run([&](auto ic) {
        constexpr int i = 4 * decltype(ic)::value;
        fmac_bcast_2x2<i>(r0, r1, ns[0], ns[1], h.x, h.y, &dr2.w[i], &dn2.w[i]);
      });
      rl[0] = (f32x2){r0, r1};
    } else {
      StaticFor<0, kHidden>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const f32x2 hr = splat2(bcast<i>(h.x));
#pragma unroll
        for (int q = 0; q < NP; ++q) rl[q] = fma2(hr, (f32x2){dr2.w[i][2 * q], dr2.w[i][2 * q + 1]}, rl[q]);
#pragma unroll
        for (int t = 0; t < C::ES; ++t) fmac_bcast<i, i == 0>(ns[t], h.y, dn2.w[i][t]);
      });
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) rl[q] = rl[q] + (f32x2){dr2.b[2 * q], dr2.b[2 * q + 1]};
#pragma unroll
    for (int t = 0; t < C::ES; ++t) ns[t] = ns[t] + dn2.b[t];
    row_min_max_normalize<E>(ns, j);
    // Prediction, first layer, on the child embedding (muax/model.py:272) or the parent (coax quirk)
    float x[C::ES];
#pragma unroll
    for (int t = 0; t < C::ES; ++t) x[t] = pred_on_parent ? s[t] : ns[t];
    f32x2 g = first_p(x);
    g = elu2(g + p1.b);
    f32x2 vl[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) vl[q] = splat2(0.0f);
    float pl = 0.0f;
    if constexpr (C::L2_BLOCK) {
      float v0 = 0.0f, v1 = 0.0f;
      StaticFor<0, kHidden / 4>::run([&](auto ic) {
        constexpr int i = 4 * decltype(ic)::value;
        fmac_bcast_2x1<i>(v0, v1, pl, g.x, g.y, &pv2.w[i], &pp2.w[i]);
      });
      vl[0] = (f32x2){v0, v1};
    } else {
      StaticFor<0, kHidden>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const f32x2 gv = splat2(bcast<i>(g.x));
#pragma unroll
        for (int q = 0; q < NP; ++q) vl[q] = fma2(gv, (f32x2){pv2.w[i][2 * q], pv2.w[i][2 * q + 1]}, vl[q]);
        fmac_bcast<i, i == 0>(pl, g.y, pp2.w[i][0]);
      });
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) vl[q] = vl[q] + (f32x2){pv2.b[2 * q], pv2.b[2 * q + 1]};
    pi_logit = pl + pp2.b[0];
    {
      // children_prior of the new node: softmax over the A policy logits (an independent chain the
      // scheduler interleaves with the decodes below)
      float px[1] = {pi_logit}, pp[1];
      row_softmax<A>(px, j, pp);
      pi_prob = pp[0];
    }
    // support_to_scalar(softmax(.)) of the reward logits and the value logits, as (reward, value) pairs per
    // lane slot; slot 0 is always inside the support (F > 16), the others are masked by ok[t]
    f32x2 xs[FS];
    bool ok[FS];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      xs[2 * q] = (f32x2){rl[q].x, vl[q].x};
      xs[2 * q + 1] = (f32x2){rl[q].y, vl[q].y};
    }
#pragma unroll
    for (int t = 0; t < FS; ++t) ok[t] = t == 0 || j + 16 * t < F;
    f32x2 m = xs[0];
#pragma unroll
    for (int t = 1; t < FS; ++t) m = ok[t] ? (f32x2){fmaxf(m.x, xs[t].x), fmaxf(m.y, xs[t].y)} : m;
    m = (f32x2){row_max<4>(m.x), row_max<4>(m.y)};
    f32x2 e[FS], dl[FS];
    dl[0] = xs[0] - m;
    e[0] = exp_neg2(dl[0]);
    f32x2 part = e[0];
    float lowest = fminf(dl[0].x, dl[0].y);
#pragma unroll
    for (int t = 1; t < FS; ++t) {
      dl[t] = xs[t] - m;
      lowest = fminf(lowest, fminf(dl[t].x, dl[t].y));
      e[t] = exp_neg2(dl[t]);
      e[t] = ok[t] ? e[t] : splat2(0.0f);
      part = part + e[t];  // (a masked slot adds +0: exact, no select needed)
    }
    const f32x2 sum = (f32x2){row_sum(part.x), row_sum(part.y)};
    // e / sum: the sums are in [1, F]; unless some lane of the wavefront holds a logit more than 69 below its
    // row's maximum (an exp in (0, 2^-100): wave-uniform test, never seen with real networks) the quotients take the
    // shared-reciprocal form of the division
    f32x2 w[FS];
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(lowest < kDivNewtonMinArg) == 0, 1)) {
      const f32x2 y = rcp_newton2(sum);
#pragma unroll
      for (int t = 0; t < FS; ++t) w[t] = div_newton2(e[t], sum, y);
    } else {
#pragma unroll
      for (int t = 0; t < FS; ++t) w[t] = (f32x2){e[t].x / sum.x, e[t].y / sum.y};
    }
    f32x2 tp = splat2((float)(j - support)) * w[0];
#pragma unroll
    for (int t = 1; t < FS; ++t) {
      const f32x2 tt = splat2((float)(j + 16 * t - support)) * w[t];
      tp = tp + tt;  // (a masked slot's term is (+-n) * (+0 / sum) = +-0: adding it is exact)
    }
    const f32x2 dec = inv_scaling2((f32x2){row_sum(tp.x), row_sum(tp.y)});
    reward = dec.x;
    value = dec.y;
  }
};

// mctx muzero_action_selection (value_score + policy_score) for every child of
// one node, with qtransform_by_parent_and_siblings; tie-break noise and the
// root mask are applied at selection time.
// x / d for a small positive integer d, given y = RN(1/d): q0 = RN(x y), r = x - q0 d (exact, fma),
// q = RN(q0 + r y) is the correctly rounded quotient (Markstein); three dependent VALU ops instead of the
// ~11 of the IEEE expansion.  tests/test_oracle_kat.py checks q == x / d for every mantissa and d <= 300.
MZ_DEV float div_small(float x, float d, float y) {
  const float q0 = x * y;
  const float r = __builtin_fmaf(-q0, d, x);
  return __builtin_fmaf(r, y, q0);
}
// rcp1[a] = RN(1 / (vis[a] + 1)) from the LDS table
// PRE: the policy scores come precomputed in `score` (the backup computes them in the wait slots of its discounted-return
// chain: they need nothing the chain produces)
template <int A, bool SHARED_RCP = false, bool PRE = false>
MZ_DEV void puct_scores(float nval, float tn, const float (&prob)[A], const float (&val)[A],
                        const int (&vis)[A], const float (&rew)[A], const float (&dis)[A],
                        const float (&rcp1)[A], float (&score)[A]) {
  float q[A];
  float lo = nval, hi = nval;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    q[a] = rew[a] + dis[a] * val[a];
    float safe = vis[a] > 0 ? q[a] : nval;
    lo = fminf(lo, safe);
    hi = fmaxf(hi, safe);
  }
  float span = fmaxf(hi - lo, 1e-8f);
  float num[A], vs[A];
#pragma unroll
  for (int a = 0; a < A; ++a) num[a] = (vis[a] > 0 ? q[a] : lo) - lo;
  bool plain = true;
  if constexpr (SHARED_RCP && A >= 2) {
    // the A quotients share their denominator: one refined reciprocal, numerators in packed pairs (div_newton2) --
    // valid while every numerator is 0 or >= 2^-100 and the span is below 2^100 (wave-uniform test; else IEEE division)
    uint32_t low = f2u(num[0]) - 1u;  // 0 wraps to the top: only (0, 2^-100) fails the test
#pragma unroll
    for (int a = 1; a < A; ++a) low = min(low, f2u(num[a]) - 1u);
    const bool risky = low < f2u(0x1p-100f) - 1u || span >= 0x1p100f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(risky) == 0, 1)) {
      plain = false;
      const float y0 = __builtin_amdgcn_rcpf(span);
      const float y = __builtin_fmaf(__builtin_fmaf(-span, y0, 1.0f), y0, y0);
#pragma unroll
      for (int a = 0; a < A; a += 2) {
        const int b = a + 1 < A ? a + 1 : a;
        const f32x2 qq = div_newton2((f32x2){num[a], num[b]}, splat2(span), splat2(y));
        vs[a] = qq.x;
        vs[b] = qq.y;
      }
    }
  }
  if (plain) {
#pragma unroll
    for (int a = 0; a < A; ++a) vs[a] = num[a] / span;
  }
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const float policy_score = PRE ? score[a] : div_small(tn * prob[a], (float)(vis[a] + 1), rcp1[a]);
    score[a] = vs[a] + policy_score;
  }
}

// First-max argmax of the cached scores and the noise-independence test: mctx adds
// 1e-7 * uniform[0,1) to every score; if fl(score_a + 1e-7) < score_best for every other
// action, then for ANY draw fl(score_a + n_a) <= fl(score_a + 1e-7) < score_best <=
// fl(score_best + n_best) (rounding is monotone), so the argmax cannot change.
template <int A, bool TB>
MZ_DEV void decide(const float (&sc)[A], const int (&cidx)[A], int& best, int& child, bool& safe) {
  best = 0;
  float bs = sc[0];
  child = cidx[0];
#pragma unroll
  for (int a = 1; a < A; ++a) {
    bool take = sc[a] > bs;  // first max wins
    bs = take ? sc[a] : bs;
    best = take ? a : best;
    child = take ? cidx[a] : child;
  }
  safe = true;
  if constexpr (TB) {
#pragma unroll
    for (int a = 0; a < A; ++a) safe = safe && (a == best || (sc[a] + 1e-7f) < bs);
  }
}
MZ_DEV int jump_word(int node, int action, int depth, bool near_tie) {
  return node | (action << 12) | (depth << 16) | (near_tie ? (int)0x80000000 : 0);
}

// ---- Gumbel MuZero decisions (mctx gumbel_muzero_{root,interior}_action_selection), all A children of
// one node inside a lane.  Sums follow the canonical 16-wide butterfly on the zero-padded vector. ----
template <int A>
MZ_DEV float sum16_inlane(const float (&x)[A]) {
  static_assert(A <= 16, "");
  float p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = i < A ? x[i] : 0.0f;
  float q0 = p[0] + p[1], q1 = p[2] + p[3], q2 = p[4] + p[5], q3 = p[6] + p[7];
  float r0 = q0 + q1, r1 = q2 + q3;
  if constexpr (A <= 8) {
    return (r0 + r1) + 0.0f;  // last butterfly step adds the (all-zero) upper half row
  } else {
    float q4 = p[8] + p[9], q5 = p[10] + p[11], q6 = p[12] + p[13], q7 = p[14] + p[15];
    float r2 = q4 + q5, r3 = q6 + q7;
    return (r0 + r1) + (r2 + r3);
  }
}
template <int A>
MZ_DEV void softmax_inlane(const float (&x)[A], float (&p)[A]) {
  float m = x[0];
#pragma unroll
  for (int a = 1; a < A; ++a) m = fmaxf(m, x[a]);
  float e[A];
#pragma unroll
  for (int a = 0; a < A; ++a) e[a] = exp_neg(x[a] - m);
  const float s = sum16_inlane<A>(e);
#pragma unroll
  for (int a = 0; a < A; ++a) p[a] = e[a] / s;
}
// mctx qtransforms (QT 0: by_parent_and_siblings, 1: completed_by_mix_value)
template <int A, int QT>
MZ_DEV void qtransform_inlane(float nval, float raw, const float (&logit)[A], const float (&val)[A],
                              const int (&vis)[A], const float (&rew)[A], const float (&dis)[A],
                              float (&out)[A], int& sum_visits) {
  float q[A];
  sum_visits = 0;
  int maxvisit = 0;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    q[a] = rew[a] + dis[a] * val[a];
    sum_visits += vis[a];
    maxvisit = max(maxvisit, vis[a]);
  }
  if constexpr (QT == 0) {
    float lo = nval, hi = nval;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      float safe = vis[a] > 0 ? q[a] : nval;
      lo = fminf(lo, safe);
      hi = fmaxf(hi, safe);
    }
    const float span = fmaxf(hi - lo, 1e-8f);
#pragma unroll
    for (int a = 0; a < A; ++a) out[a] = ((vis[a] > 0 ? q[a] : lo) - lo) / span;
  } else {
    float prior[A], tmp[A];
    softmax_inlane<A>(logit, prior);
#pragma unroll
    for (int a = 0; a < A; ++a) {
      prior[a] = fmaxf(prior[a], kFltTiny);
      tmp[a] = vis[a] > 0 ? prior[a] : 0.0f;
    }
    const float sum_probs = sum16_inlane<A>(tmp);
#pragma unroll
    for (int a = 0; a < A; ++a) tmp[a] = vis[a] > 0 ? (prior[a] * q[a]) / sum_probs : 0.0f;
    const float weighted_q = sum16_inlane<A>(tmp);
    const float value = (raw + (float)sum_visits * weighted_q) / (float)(sum_visits + 1);
    float lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      out[a] = vis[a] > 0 ? q[a] : value;
      lo = fminf(lo, out[a]);
      hi = fmaxf(hi, out[a]);
    }
    const float span = fmaxf(hi - lo, 1e-8f);
    const float scale = (50.0f + (float)maxvisit) * 0.1f;
#pragma unroll
    for (int a = 0; a < A; ++a) out[a] = scale * ((out[a] - lo) / span);
  }
}
// root: seq_halving.score_considered + invalid mask; interior: softmax(logits + q) - visits / (1 + sum)
template <int A, int QT>
MZ_DEV void gumbel_scores(bool is_root, float nval, float raw, const float (&logit)[A], const float (&val)[A],
                          const int (&vis)[A], const float (&rew)[A], const float (&dis)[A],
                          const float (&gum)[A], int considered_visit, uint32_t inv_bits, float (&sc)[A]) {
  float qv[A];
  int sum_visits;
  qtransform_inlane<A, QT>(nval, raw, logit, val, vis, rew, dis, qv, sum_visits);
  float x[A], pr[A];
  float mxl = logit[0];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    x[a] = logit[a] + qv[a];
    mxl = fmaxf(mxl, logit[a]);
  }
  softmax_inlane<A>(x, pr);
#pragma unroll
  for (int a = 0; a < A; ++a) {
    float r = fmaxf((gum[a] + (logit[a] - mxl)) + qv[a], -1e9f);
    r = r + (vis[a] == considered_visit ? 0.0f : -INFINITY);
    r = ((inv_bits >> a) & 1u) ? -INFINITY : r;
    const float in = pr[a] - (float)vis[a] / (float)(1 + sum_visits);
    sc[a] = is_root ? r : in;
  }
}

template <class C>
__global__ __launch_bounds__(C::THREADS, (C::PH && !C::LONG) ? 2 : 1) void mz_act_fused_kernel(const FusedParams p) {
  constexpr int A = C::A, E = C::E, NS = C::NS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // One wavefront per SIMD owns the whole 512-entry unified register file.  LLVM infers "no AGPRs"
  // for a kernel without MFMA and would spill to scratch beyond 256 VGPRs; naming an AGPR keeps the
  // accumulator half allocatable so that spills (the E=32 weight columns) stay in registers.
  // (the packed instances run two wavefronts per SIMD: their 256 registers are all VGPRs, no AGPR half to keep)
  if constexpr (!C::PK) asm volatile("; keep AGPRs allocatable" ::: "a0");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int j = lane & 15;
  const int root_in_wg = (tid >> 6) * 4 + (lane >> 4);
  // rows past the end of the batch shadow the last root (identical values, identical stores): every
  // lane stays live, so wave-uniform control flow below can read any row with v_readlane
  const int r_raw = blockIdx.x * C::ROOTS_PER_WG + root_in_wg;
  const int r = r_raw < p.B ? r_raw : p.B - 1;

#ifdef MZ_PROFILE
  uint64_t prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t prof_t = __builtin_amdgcn_s_memtime();
#endif
  // packed instances (two wavefronts per SIMD, E = 32): the weights first -- their round trip runs behind the table's
  // arithmetic, the LDS fill and the barrier (LunarLander's 8192 roots: 189.5 -> 186.2 us; the plain instances lose 0.7 %
  // with the same order and keep theirs: same-box A/B, round 6)
  Nets<C> nets;
  if constexpr (C::PK) {
    nets.load(p, j);
    nets.wlds = lds + C::TBL_WORDS + 4 * j;
  }
  float* tbl = lds;  // sqrt(n) * pb_c(n) by visit count
  for (int i = tid; i < C::TBL_WORDS / 2; i += C::THREADS) {
    tbl[2 * i] = puct_scale(i, p.pb_c_init, p.pb_c_base);
    tbl[2 * i + 1] = i > 0 ? 1.0f / (float)i : 0.0f;  // correctly rounded reciprocal for div_small
  }
  Nets<C>::fill_lds(p, lds + C::TBL_WORDS, tid);
  __syncthreads();  // the only barrier

  float* tree = lds + C::TBL_WORDS + C::WLDS_WORDS + root_in_wg * C::ROOT_WORDS;
  int* itree = reinterpret_cast<int*>(tree);
  const uint64_t rg = p.root_offset + (uint64_t)r;
  const int S = p.S;
  const int max_depth = p.max_depth > 0 ? p.max_depth : S;
  const int N = S + 1;
  const int support = p.support;
  const bool ex = p.export_tree != 0;

  // embeddings of this root in HBM (instances that keep them out of LDS): the caller's export buffer when
  // a tree export is requested, else the handle's scratch
  float* gemb = C::EMB_LDS ? nullptr : (ex ? p.t_embeddings : p.emb_scratch) + (size_t)r * N * E;

  // root paths of this root's nodes in HBM (compact record)
  int32_t* gpath = C::PH ? p.path_scratch + (size_t)r * N * C::PATHW : nullptr;

  if constexpr (!C::PK) nets.load(p, j);
  // ---- tree init (mctx instantiate_tree_from_root) ----
  // all-zero records written 16 bytes per lane, then children_index = -1 (same wave: LDS keeps the order)
  {
    int4* q4 = reinterpret_cast<int4*>(itree);
    const int nq = (N * NS + 3) / 4;
    for (int q = j; q < nq; q += 16) q4[q] = make_int4(0, 0, 0, 0);
    for (int n = j; n < N; n += 16) {
      if constexpr (C::PK) itree[n * NS + C::SEL0] = -1;  // four unvisited children: 0xff bytes
      else {
#pragma unroll
        for (int a = 0; a < A; ++a) itree[n * NS + C::SEL0 + 2 * a] = -1;
      }
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
    if constexpr (!C::EMB_LDS)
      for (int i = j; i < N * E; i += 16) gemb[i] = 0.0f;
  }

  // ---- root inference (muax/model.py:251-263) ----
  float s[C::ES];
  {
    const float* ob = p.obs + (size_t)r * p.obs_dim;
#pragma unroll
    for (int t = 0; t < C::ES; ++t) {
      int k = j + 16 * t;
      float acc = 0.0f;
      {
        // four links' operands requested before the first of their fmas issues (obs_dim is a run-time count: one link at a
        // time was one memory round trip per link -- eight in a row for LunarLander); lanes past E compute a value nobody
        // reads from a clamped column instead of branching around the loads
        const int kc = k < E ? k : 0;
        int i = 0;
        for (; i + 4 <= p.obs_dim; i += 4) {
          float o4[4], w4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o4[q] = ob[i + q];
            w4[q] = p.repr_w[(i + q) * E + kc];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_fmaf(o4[q], w4[q], acc);
        }
        for (; i < p.obs_dim; ++i) acc = __builtin_fmaf(ob[i], p.repr_w[i * E + kc], acc);
        acc = acc + p.repr_b[kc];
      }
      s[t] = k < E ? acc : 0.0f;
    }
    row_min_max_normalize<E>(s, j);
  }
  float v0, pl0;
  nets.predict(s, j, support, p.F, v0, pl0);
  uint32_t inv_bits = 0;  // root_invalid_actions as a bit mask (row uniform)
  if (p.invalid != nullptr) {
#pragma unroll
    for (int a = 0; a < A; ++a) inv_bits |= p.invalid[(size_t)r * A + a] ? (1u << a) : 0u;
  }
  int ncons = 0;  // gumbel policy: number of root actions sequential halving considers
  // the root's JUMP word (row uniform): every selection starts from it, and the backup ends by producing it -- it
  // stays in a register, and a selection that meets no near tie and no depth cut reads nothing from LDS
  int root_jw;
  {
    float lg, pq[1];
    if constexpr (!C::GUMBEL) {
      // mctx muzero_policy prelude: dirichlet mix, log, invalid-action mask
      float x[1] = {pl0}, pr[1];
      row_softmax<A>(x, j, pr);
      float nz = (p.dirichlet_noise != nullptr && j < A) ? p.dirichlet_noise[(size_t)r * A + j] : 0.0f;
      float keep = 1.0f - p.dirichlet_fraction;
      float noisy = keep * pr[0] + p.dirichlet_fraction * nz;
      lg = log_pos(fmaxf(noisy, kFltTiny));
      if (p.invalid != nullptr) {
        float mx = row_max<4>(j < A ? lg : -INFINITY);
        lg = ((inv_bits >> j) & 1u) ? kFltLowest : lg - mx;
      }
      float lx[1] = {lg};
      row_softmax<A>(lx, j, pq);
    } else {
      // mctx gumbel_muzero_policy prelude: invalid-action mask only; root Gumbel noise
      lg = pl0;
      if (inv_bits != 0) {
        float mx = row_max<4>(j < A ? lg : -INFINITY);
        lg = ((inv_bits >> j) & 1u) ? kFltLowest : lg - mx;
      }
      pq[0] = 0.0f;
      float g;
      if (p.gumbel != nullptr) {
        g = j < A ? p.gumbel[(size_t)r * A + j] : 0.0f;
      } else {
        uint32_t x0, x1;
        bool second;
        bits_block(p.global_batch * (uint64_t)A, rg * (uint64_t)A + (uint64_t)(j < A ? j : 0), x0, x1, second);
        threefry2x32(p.k_gumbel[0], p.k_gumbel[1], x0, x1);
        g = p.gumbel_scale * gumbel_from_bits(second ? x1 : x0);
      }
      if (j < A) {
        tree[C::GUM0 + j] = g;  // root_gumbel: in the root's own (empty) path slot, or behind the tree
        tree[C::off_logit(j)] = lg;
      }
      ncons = min(p.max_considered, A - __builtin_popcount(inv_bits));
    }
    if (j < A) tree[C::off_prob(j)] = pq[0];
    if (ex && j < A) p.t_children_prior_logits[(size_t)r * N * A + j] = lg;
    if (j == 0) {
      itree[C::HDR0] = 1;
      tree[C::HDR0 + 1] = v0;
      if constexpr (C::RAW_OK) tree[C::HDR0 + 3] = v0;
      p.root_value[r] = v0;
      if (ex) p.t_raw_values[(size_t)r * N] = v0;
    }
#pragma unroll
    for (int t = 0; t < C::ES; ++t)
      if (j + 16 * t < E) {
        if constexpr (C::EMB_LDS) tree[C::EMB0 + j + 16 * t] = s[t];
        else gemb[j + 16 * t] = s[t];
      }
    // cached scores of the root's children (all unvisited)
    float prob[A], val[A], rew[A], dis[A], sc[A];
    int vis[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      prob[a] = tree[C::off_prob(a)];
      val[a] = 0.0f; vis[a] = 0; rew[a] = 0.0f; dis[a] = p.discount;
    }
    int cidx[A], best, child;
    bool safe;
    if constexpr (!C::GUMBEL) {
      float rcp1[A];
#pragma unroll
      for (int a = 0; a < A; ++a) rcp1[a] = 1.0f;  // no child visited yet
      puct_scores<A>(v0, tbl[2], prob, val, vis, rew, dis, rcp1, sc);
    } else {
      float logit[A], gum[A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        logit[a] = tree[C::off_logit(a)];
        gum[a] = tree[C::GUM0 + a];
      }
      gumbel_scores<A, C::QT>(true, v0, v0, logit, val, vis, rew, dis, gum, p.visit_table[(size_t)ncons * S],
                              inv_bits, sc);
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
      cidx[a] = -1;
      sc[a] = ((inv_bits >> a) & 1u) ? -INFINITY : sc[a];  // the root is only ever selected at depth 0
    }
    decide<A, C::TB>(sc, cidx, best, child, safe);
    root_jw = jump_word(0, best, 0, !safe);
    if (j == 0) {
#pragma unroll
      for (int a = 0; a < A; ++a) tree[C::off_score(a)] = sc[a];
      itree[C::JUMP] = root_jw;
    }
  }

  int depth_total = 0;
  // embeddings in HBM: the parent row of the NEXT simulation is prefetched behind the backup's stores
  int pref_parent = 0;
  float pre[C::ES];
#pragma unroll
  for (int t = 0; t < C::ES; ++t) pre[t] = 0.0f;
  if constexpr (!C::EMB_LDS) {
#pragma unroll
    for (int t = 0; t < C::ES; ++t) pre[t] = s[t];  // simulation 0 expands below the root
  }
  MZ_TICKW(12);  // prologue: weights, tree init, root inference

  // ---- simulations (mctx search.search body_fun) ----
  for (int sim = 0; sim < S; ++sim) {
    MZ_TICK(0);
    int cv_next = 0;  // gumbel: visits the root's considered actions must have at the NEXT simulation
    if constexpr (C::GUMBEL) cv_next = p.visit_table[(size_t)ncons * S + (sim + 1 < S ? sim + 1 : S - 1)];
    // -- simulate (mctx search.simulate) through the JUMP words: one iteration per near tie --
    int parent, action, dP;
    // children_index[parent, action] of the end point: a cached descent ends at an unexpanded edge (-1) unless it
    // ends at a near tie (the noisy evaluation below names the child) or is cut at max_depth (read there)
    int next = -1;
    parent = root_jw & 0xfff;
    action = (root_jw >> 12) & 0xf;
    dP = (root_jw >> 16) & 0xff;
    int depth = dP + 1;
    // The common case -- no row of the wavefront meets a near tie or the depth limit -- is decided by the root's word
    // alone; everything else (per-row control flow, key walk, noisy evaluations, the max_depth cut) sits behind ONE
    // wave-uniform branch, so the common case pays no exec-mask bookkeeping for it.
    bool rare = depth > max_depth;
    if constexpr (C::TB) rare = rare || root_jw < 0;
    if (__builtin_amdgcn_ballot_w64(rare) != 0) {
      uint32_t fk0 = 0, fk1 = 0, fs0 = 0, fs1 = 0;  // lazy key walk: rng_key / action_selection_key
      int fk_level = -1;                             // levels already split off (-1: not started)
      int jw = root_jw;
      for (;;) {
        parent = jw & 0xfff;
        action = (jw >> 12) & 0xf;
        dP = (jw >> 16) & 0xff;
        if constexpr (C::TB) {
          if (jw < 0 && dP + 1 <= max_depth) {
            // near tie at `parent` (level dP): score + 1e-7 * uniform(action_selection_key_dP), as mctx
            const int* ndi = itree + __umul24((unsigned)parent, (unsigned)NS);
            if (fk_level < 0) {
              // simulate_keys[b] = split(simulate_key, B)[b]: words 2b, 2b+1 of the flat stream
              uint32_t x0, x1;
              bool second;
              bits_block(2 * p.global_batch, 2 * rg + (uint64_t)(j & 1), x0, x1, second);
              threefry2x32(p.sim_keys[sim][0], p.sim_keys[sim][1], x0, x1);
              uint32_t word = second ? x1 : x0;
              fk0 = bcast_u<0>(word);
              fk1 = bcast_u<1>(word);
              fk_level = 0;
            }
#ifdef MZ_PROFILE
            prof_acc[14] += 1;                               // near-tie evaluations
            prof_acc[15] += (uint64_t)(dP + 1 - fk_level);   // key-walk levels hashed for them
#endif
            while (fk_level <= dP) {
              // rng_key, action_selection_key = split(rng_key): lanes 0/1 hash one block each
              uint32_t x0 = (uint32_t)(j & 1), x1 = 2u + (uint32_t)(j & 1);
              threefry2x32(fk0, fk1, x0, x1);
              fk0 = bcast_u<0>(x0); fk1 = bcast_u<1>(x0);
              fs0 = bcast_u<0>(x1); fs1 = bcast_u<1>(x1);
              fk_level += 1;
            }
            constexpr int NB = (A + 1) / 2;
            const int jb = j % NB;
            uint32_t x0 = (uint32_t)jb, x1 = (NB + jb < A) ? (uint32_t)(NB + jb) : 0u;
            threefry2x32(fs0, fs1, x0, x1);
            int cidx[A];
            float score[A];
            C::load_cidx(ndi, cidx);
            StaticFor<0, A>::run([&](auto ic) {
              constexpr int a = decltype(ic)::value;
              const uint32_t bits = bcast_u<(a % NB)>(a < NB ? x0 : x1);
              score[a] = __int_as_float(ndi[C::off_score(a)]) + 1e-7f * uniform_from_bits(bits);
            });
            int best = 0, bn = cidx[0];
            float bs = score[0];
#pragma unroll
            for (int a = 1; a < A; ++a) {
              bool take = score[a] > bs;  // first max wins
              bs = take ? score[a] : bs;
              best = take ? a : best;
              bn = take ? cidx[a] : bn;
            }
            action = best;
            next = bn;
            if (bn >= 0 && dP + 1 < max_depth) {
              // the noisy choice is an expanded child within reach: keep descending from it
              jw = itree[__umul24((unsigned)bn, (unsigned)NS) + C::JUMP];
              next = -1;
              continue;
            }
          }
        }
        break;
      }
      depth = dP + 1;
      if (depth > max_depth) {
        // the cached descent overshoots max_depth: stop at level max_depth - 1 of the same path
        depth = max_depth;
        const int* pb = C::PH ? gpath + (size_t)parent * C::PATHW : itree + __umul24((unsigned)parent, (unsigned)NS) + C::PATH0;
        const int e = depth - 1;
        const int ent = (pb[(e * C::ENTRY_BITS) >> 5] >> ((e * C::ENTRY_BITS) & 31)) & ((1 << C::ENTRY_BITS) - 1);
        parent = ent & ((1 << C::ENTRY_ACT_SHIFT) - 1);
        action = ent >> C::ENTRY_ACT_SHIFT;
        next = C::load_cidx1(itree + __umul24((unsigned)parent, (unsigned)NS), action);
      }
    }
    depth_total += depth;
    MZ_TICK(1);  // select
    const bool fresh = next < 0;
    const int newn = fresh ? sim + 1 : next;  // (an expanded child is only re-expanded at the max_depth cut)

    // -- expand (mctx search.expand, recurrent_fn = muax/model.py:265-282) --
    float sp[C::ES];
#pragma unroll
    for (int t = 0; t < C::ES; ++t)
      if constexpr (C::EMB_LDS) {
        // (an unconditional read at a clamped index + a select: no exec-mask region)
        const int jc = j + 16 * t < E ? j + 16 * t : E - 1;
        const float v = tree[__umul24((unsigned)parent, (unsigned)NS) + C::EMB0 + jc];
        sp[t] = (j + 16 * t < E) ? v : 0.0f;
      } else {
        // the row was requested at the end of the previous simulation (from the root's fresh JUMP word);
        // only a near-tie redraw or a max_depth cut can have picked another parent
        float v = pre[t];
        if (parent != pref_parent && j + 16 * t < E) v = gemb[(size_t)parent * E + j + 16 * t];
        sp[t] = (j + 16 * t < E) ? v : 0.0f;
      }
    // LDS reads the expansion needs are issued before the network pass so their latency hides behind it
    float* nn = tree + __umul24((unsigned)newn, (unsigned)NS);
    int* nni = reinterpret_cast<int*>(nn);
    const unsigned po = __umul24((unsigned)parent, (unsigned)NS);
    const int vis_old = nni[C::HDR0];
    int ppw[C::PATHS];
#pragma unroll
    for (int t = 0; t < C::PATHS; ++t) {
      const int wi = j + 16 * t < C::PATHW ? j + 16 * t : 0;
      if constexpr (C::PH) ppw[t] = gpath[(size_t)parent * C::PATHW + wi];  // (consumed after the network pass)
      else ppw[t] = itree[po + C::PATH0 + wi];
    }
    float reward, value, pil, pprob;
    float ns[C::ES];
    nets.forward(sp, action, j, support, p.F, p.pred_on_parent != 0, reward, value, pil, pprob, ns);
    MZ_TICK(2);  // network pass (+ parent embedding gather)
    {
      const int vis = vis_old + 1;
      if (j < A) {
        nn[C::off_prob(j)] = pprob;
        if constexpr (C::GUMBEL) nn[C::off_logit(j)] = pil;
      }
#pragma unroll
      for (int t = 0; t < C::ES; ++t)
        if (j + 16 * t < E) {
          if constexpr (C::EMB_LDS) nn[C::EMB0 + j + 16 * t] = ns[t];
          else gemb[(size_t)newn * E + j + 16 * t] = ns[t];
        }
      if (j == 0) {
        nni[C::HDR0] = vis;
        nn[C::HDR0 + 1] = value;
        if constexpr (C::RAW_OK) nn[C::HDR0 + 3] = value;  // raw_values[new] (mctx update_tree_node)
        C::store_cidx(itree + po, action, newn);
        tree[po + C::off_rew(action)] = reward;
      }
      if (fresh) {
        // the new node's root path = its parent's path + (parent, action); written once
        const int e = depth - 1;
        const int sh = (e * C::ENTRY_BITS) & 31;
        const int ent = parent | (action << C::ENTRY_ACT_SHIFT);
#pragma unroll
        for (int t = 0; t < C::PATHS; ++t) {
          const int wi = j + 16 * t;
          int w = ppw[t];
          w = (wi == ((e * C::ENTRY_BITS) >> 5))
                  ? (int)(((unsigned)w & ~(((1u << C::ENTRY_BITS) - 1u) << sh)) | ((unsigned)ent << sh))
                  : w;
          if (wi < C::PATHW) {
            if constexpr (C::PH) gpath[(size_t)newn * C::PATHW + wi] = w;
            else nni[C::PATH0 + wi] = w;
          }
        }
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

    MZ_TICK(4);  // expand stores
    // -- backward (mctx search.backward) + decision refresh, lane e <-> path entry e --
    // entries 0..depth-1 are the (parent, action) edges of the path, entry `depth` is the leaf.
    {
      // wave-uniform trip counts: the deepest of the wave's four rows
      int wmax;
      {
        const int d0 = __builtin_amdgcn_readlane(depth, 0), d1 = __builtin_amdgcn_readlane(depth, 16);
        const int d2 = __builtin_amdgcn_readlane(depth, 32), d3 = __builtin_amdgcn_readlane(depth, 48);
        int m01, m23;  // (scalar max: the compiler otherwise moves two of the four back into VGPRs for a v_max3)
        asm("s_max_i32 %0, %1, %2" : "=s"(m01) : "s"(d0), "s"(d1) : "scc");
        asm("s_max_i32 %0, %1, %2" : "=s"(m23) : "s"(d2), "s"(d3) : "scc");
        asm("s_max_i32 %0, %1, %2" : "=s"(wmax) : "s"(m01), "s"(m23) : "scc");
      }
      float G = value;        // leaf_value walking up (row uniform)
      float carry_v = value;  // node value of the entry just below this chunk
      int carry_n = -1;       // node index of the entry just below this chunk
      int carry_j = 0;        // resolved JUMP word of the entry just below this chunk
      for (int c = wmax >> 4; c >= 0; --c) {
        const int e = 16 * c + j;
        const bool valid = e <= depth;
        const bool isleaf = e == depth;
        const bool edge = e < depth;
        int pn, pa;
        {
          const int ec = e < depth - 1 ? e : 0;
          // word (ec * ENTRY_BITS) / 32 of the parent's path sits in that lane (and slot) of this row's ppw, loaded
          // before the network pass: a cross-lane fetch, not a second LDS read behind the expansion's stores
          const int widx = (ec * C::ENTRY_BITS) >> 5;
          // the entries of chunk c lie in ONE of the row's path registers (16 entries are 4 or 8 words; a register slot is
          // 16 words): picked with a wave-uniform select, then ONE cross-lane fetch -- round 5: a fetch per slot and a
          // per-lane select cost a 255-simulation instance (eight slots) 1700 cycles per simulation in this phase.  (The
          // lanes at and past the leaf, whose clamped entry 0 sits in slot 0, fetch a word they never use: their (pn, pa)
          // are overridden below.)
          const int slot_c = ((16 * c * C::ENTRY_BITS) >> 5) >> 4;
          int psrc = ppw[0];
#pragma unroll
          for (int t = 1; t < C::PATHS; ++t) psrc = (slot_c == t) ? ppw[t] : psrc;
          const int pword = __builtin_amdgcn_ds_bpermute(4 * ((lane & ~15) + (widx & 15)), psrc);
          const int ent = (pword >> ((ec * C::ENTRY_BITS) & 31)) & ((1 << C::ENTRY_BITS) - 1);
          pn = ent & ((1 << C::ENTRY_ACT_SHIFT) - 1);
          pa = ent >> C::ENTRY_ACT_SHIFT;
          pn = e == depth - 1 ? parent : pn;
          pa = e == depth - 1 ? action : pa;
          pn = isleaf ? newn : pn;
          pa = isleaf ? 0 : pa;
          pn = valid ? pn : 0;
        }
        MZ_TICKW(6);  // path entry fetch + decode
        float* nd = tree + __umul24((unsigned)pn, (unsigned)NS);
        int* ndi = reinterpret_cast<int*>(nd);
        const int cnt = ndi[C::HDR0];
        const float pv = nd[C::HDR0 + 1];
        float prob[A], val[A], rew[A], dis[A];
        int vis[A], cidx[A];
        C::load_cidx(ndi, cidx);
        C::load_vis(ndi, vis);
#pragma unroll
        for (int a = 0; a < A; ++a) {
          prob[a] = nd[C::off_prob(a)];
          val[a] = nd[C::off_val(a)];
          rew[a] = nd[C::off_rew(a)];
          dis[a] = p.discount;
        }
        // cached JUMP words of all children (clamped addresses), fetched now so that the one the
        // refreshed decision picks is already here
        int jch[A];
#pragma unroll
        for (int a = 0; a < A; ++a)
          jch[a] = itree[__umul24((unsigned)(cidx[a] < 0 ? 0 : cidx[a]), (unsigned)NS) + C::JUMP];
        // visit counts after this backup and the table entries they select ({sqrt(n) pb_c(n), 1/n}
        // pairs): fetched together with the JUMP words, behind the G chain
        const int nvis = edge ? cnt + 1 : cnt;
        const float2 tn_rc = *reinterpret_cast<const float2*>(tbl + 2 * (valid ? nvis : 0));
        float rcp1[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
          vis[a] = (edge && pa == a) ? vis[a] + 1 : vis[a];
          rcp1[a] = tbl[2 * (vis[a] + 1) + 1];
        }
        MZ_TICKW(7);  // node + child JUMP loads
        float re = rew[0];
#pragma unroll
        for (int a = 1; a < A; ++a) re = (pa == a) ? rew[a] : re;
        re = edge ? re : 0.0f;  // identity step for the leaf entry and for idle lanes
        const float ge = edge ? p.discount : 1.0f;
        // leaf_value = reward + discount * leaf_value, deepest entry first.  Every lane keeps its OWN
        // entry's value: one step is G[e] = re[e] + ge[e] * G[e + 1] on all lanes at once (row_shl:1;
        // lane 15 keeps the product with the value carried in from the chunk below), so entry e is final
        // after (deepest entry - e + 1) steps and stays put afterwards; identity lanes hold the carry.
        // Steps above the wave's deepest entry are identities for every row and are jumped over.
        const int kstart = min(15, wmax - 16 * c);
        float Gt = ge * G;
#define MZ_GSTEP_TXT "s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %1, %0, %3\n\t"
#define MZ_GSTEP4 \
  asm volatile(MZ_GSTEP_TXT MZ_GSTEP_TXT MZ_GSTEP_TXT MZ_GSTEP_TXT : "+v"(Gt), "+v"(G) : "v"(ge), "v"(re));
        if (kstart >= 12) { MZ_GSTEP4 }
        if (kstart >= 8) { MZ_GSTEP4 }
        if (kstart >= 4) { MZ_GSTEP4 }
        float sc[A];
        // the last four steps always run.  Two actions, MuZero policy: the policy scores div_small(tn prob[a], vis[a] + 1,
        // rcp1[a]) -- twelve instructions that need nothing from the chain -- take the place of the steps' wait states
        // (three independent instructions between the add and the DPP read of its result), so they issue for free and the
        // chain's own dependent-issue stalls overlap them.  Same operations on the same operands as div_small.
        constexpr bool kFill = !C::GUMBEL && A == 2;
        if constexpr (kFill) {
          float d0, d1, x0, x1, q0, q1, r0, r1;
#define MZ_GSTEP_BODY "v_mul_f32_dpp %[Gt], %[G], %[ge] row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %[G], %[Gt], %[re]\n\t"
          asm volatile(
              "v_add_u32 %[d0], 1, %[v0]\n\tv_add_u32 %[d1], 1, %[v1]\n\tv_mul_f32 %[x0], %[tn], %[p0]\n\t" MZ_GSTEP_BODY
              "v_cvt_f32_i32 %[d0], %[d0]\n\tv_cvt_f32_i32 %[d1], %[d1]\n\tv_mul_f32 %[x1], %[tn], %[p1]\n\t" MZ_GSTEP_BODY
              "v_mul_f32 %[q0], %[x0], %[y0]\n\tv_mul_f32 %[q1], %[x1], %[y1]\n\tv_fma_f32 %[r0], -%[q0], %[d0], %[x0]\n\t" MZ_GSTEP_BODY
              "v_fma_f32 %[r1], -%[q1], %[d1], %[x1]\n\tv_fma_f32 %[s0], %[r0], %[y0], %[q0]\n\tv_fma_f32 %[s1], %[r1], %[y1], %[q1]\n\t"
              MZ_GSTEP_BODY
              : [Gt] "+v"(Gt), [G] "+v"(G), [d0] "=&v"(d0), [d1] "=&v"(d1), [x0] "=&v"(x0), [x1] "=&v"(x1), [q0] "=&v"(q0),
                [q1] "=&v"(q1), [r0] "=&v"(r0), [r1] "=&v"(r1), [s0] "=&v"(sc[0]), [s1] "=&v"(sc[A - 1])
              : [ge] "v"(ge), [re] "v"(re), [tn] "v"(tn_rc.x), [p0] "v"(prob[0]), [p1] "v"(prob[A - 1]), [v0] "v"(vis[0]),
                [v1] "v"(vis[A - 1]), [y0] "v"(rcp1[0]), [y1] "v"(rcp1[A - 1]));
#undef MZ_GSTEP_BODY
        } else {
          MZ_GSTEP4
        }
#undef MZ_GSTEP4
#undef MZ_GSTEP_TXT
        const float Gown = G;
        G = bcast<0>(G);  // carried into the chunk above
        MZ_TICKW(8);  // G chain
        const float newv = div_small(pv * (float)cnt + Gown, (float)cnt + 1.0f, tn_rc.y);
        // children_values[parent, action] = node_values[child]: the child is the next entry
        float childv = __int_as_float(__builtin_amdgcn_update_dpp(
            __float_as_int(carry_v), __float_as_int(newv), 0x101 /* row_shl:1 */, 0xf, 0xf, false));
        childv = (e == depth - 1) ? value : childv;
        carry_v = bcast<0>(newv);
        const int next_pn = __builtin_amdgcn_update_dpp(carry_n, pn, 0x101, 0xf, 0xf, false);
        carry_n = bcast_i<0>(pn);
#pragma unroll
        for (int a = 0; a < A; ++a) val[a] = (edge && pa == a) ? childv : val[a];
        const float nval = edge ? newv : pv;
        MZ_TICKW(9);  // value update
        if constexpr (!C::GUMBEL) {
          puct_scores<A, true, kFill>(nval, tn_rc.x, prob, val, vis, rew, dis, rcp1, sc);
#pragma unroll
          for (int a = 0; a < A; ++a)  // root_invalid_actions: the root is only ever selected at depth 0
            sc[a] = (pn == 0 && ((inv_bits >> a) & 1u)) ? -INFINITY : sc[a];
        } else {
          float logit[A], gum[A];
#pragma unroll
          for (int a = 0; a < A; ++a) {
            logit[a] = nd[C::off_logit(a)];
            gum[a] = tree[C::GUM0 + a];
          }
          gumbel_scores<A, C::QT>(pn == 0, nval, C::RAW_OK ? nd[C::HDR0 + C::HDRW - 1] : 0.0f, logit, val, vis, rew, dis, gum, cv_next,
                                  pn == 0 ? inv_bits : 0u, sc);
        }
        MZ_TICKW(10);  // scores
        int best, child;
        bool safe;
        decide<A, C::TB>(sc, cidx, best, child, safe);
        // JUMP word: own end point, the off-path best child's cached word, or (when the best child
        // is the next entry of this very path) whatever that entry resolves to
        const bool inherit0 = valid && safe && child >= 0 && !isleaf && child == next_pn;
        int jchild = jch[0];
#pragma unroll
        for (int a = 1; a < A; ++a) jchild = (best == a) ? jch[a] : jchild;
        const int jwd0 = jump_word(pn, best, e, !safe);
        int jwd = (safe && child >= 0 && !inherit0) ? jchild : jwd0;
        // log-step resolution of "inherit from the next entry": done = all ones once a lane's word is final.  A lane
        // reading past the row end sees done = 0 (bound_ctrl zero fill) and stays open; what is still open after the
        // 15 hops inherits from the chunk below (carry_j)
        int done = inherit0 ? 0 : -1;
#define MZ_JSCAN(d)                                                                              \
  {                                                                                              \
    const int jn = __builtin_amdgcn_update_dpp(0, jwd, 0x100 + d, 0xf, 0xf, true);               \
    const int dn = __builtin_amdgcn_update_dpp(0, done, 0x100 + d, 0xf, 0xf, true);              \
    jwd = (jwd & done) | (jn & ~done);                                                           \
    done = done | dn;                                                                            \
  }
        MZ_JSCAN(1) MZ_JSCAN(2) MZ_JSCAN(4) MZ_JSCAN(8)
#undef MZ_JSCAN
        jwd = (jwd & done) | (carry_j & ~done);
        carry_j = bcast_i<0>(jwd);
        MZ_TICKW(11);  // decide + JUMP scan
        if (valid) {
          // one masked region; for the leaf entry the header / edge stores rewrite what was loaded
#pragma unroll
          for (int a = 0; a < A; ++a) nd[C::off_score(a)] = sc[a];
          ndi[C::JUMP] = jwd;
          ndi[C::HDR0] = nvis;
          nd[C::HDR0 + 1] = nval;
          float cvn = val[0];
          int cin = vis[0];
#pragma unroll
          for (int a = 1; a < A; ++a) {
            cvn = (pa == a) ? val[a] : cvn;
            cin = (pa == a) ? vis[a] : cin;
          }
          nd[C::off_val(pa)] = cvn;
          C::store_vis(ndi, pa, cin);
        }
      }
      root_jw = carry_j;  // entry 0 is the root: its refreshed JUMP word
      if constexpr (!C::EMB_LDS) {
        pref_parent = carry_j & 0xfff;  // the root's refreshed JUMP word: where the next descent ends
#pragma unroll
        for (int t = 0; t < C::ES; ++t)
          pre[t] = (j + 16 * t < E) ? gemb[(size_t)pref_parent * E + j + 16 * t] : 0.0f;
      }
    }
    MZ_TICK(5);  // backward + score refresh
  }

  if constexpr (C::GUMBEL) {
    // ---- tail of mctx gumbel_muzero_policy: best action among the most visited, completed-Q target ----
    float logit[A], val[A], rew[A], dis[A], gum[A], qv[A], sc[A], x[A], w[A];
    int vis[A], sumv, cv = 0;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      val[a] = tree[C::off_val(a)];
      vis[a] = itree[C::ST0 + C::STW * a + 2];
      rew[a] = tree[C::off_rew(a)];
      dis[a] = p.discount;
      logit[a] = tree[C::off_logit(a)];
      gum[a] = tree[C::GUM0 + a];
      cv = max(cv, vis[a]);
    }
    const float nval = tree[C::HDR0 + 1], raw = C::RAW_OK ? tree[C::HDR0 + C::HDRW - 1] : 0.0f;
    gumbel_scores<A, C::QT>(true, nval, raw, logit, val, vis, rew, dis, gum, cv, inv_bits, sc);
    qtransform_inlane<A, C::QT>(nval, raw, logit, val, vis, rew, dis, qv, sumv);
    int best = 0;
    float bs = sc[0], mx = logit[0] + qv[0];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      x[a] = logit[a] + qv[a];
      mx = fmaxf(mx, x[a]);
      bool take = a > 0 && sc[a] > bs;  // first max wins
      bs = take ? sc[a] : bs;
      best = take ? a : best;
    }
    if (inv_bits != 0) {
#pragma unroll
      for (int a = 0; a < A; ++a) x[a] = ((inv_bits >> a) & 1u) ? kFltLowest : x[a] - mx;
    }
    softmax_inlane<A>(x, w);
    if (j < A) {
      float mine = w[0];
#pragma unroll
      for (int a = 1; a < A; ++a) mine = (j == a) ? w[a] : mine;
      p.action_weights[(size_t)r * A + j] = mine;
    }
    if (j == 0) {
      p.action[r] = best;
      if (p.search_value) p.search_value[r] = nval;
      if (p.depth_sum) p.depth_sum[r] = depth_total;
    }
  } else {
  // ---- summary + sample (mctx Tree.summary, _apply_temperature, categorical) ----
  {
    const int ja = j < A ? j : A - 1;
    int vc;
    if constexpr (C::PK) vc = reinterpret_cast<const uint8_t*>(itree + C::VIS0)[ja];
    else vc = itree[C::ST0 + C::STW * ja + 2];
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
    row_argmax<4>(score, best, dummy);
    if (j < A) p.action_weights[(size_t)r * A + j] = prob;
    if (j == 0) {
      p.action[r] = best;
      if (p.search_value) p.search_value[r] = tree[C::HDR0 + 1];
      if (p.depth_sum) p.depth_sum[r] = depth_total;
    }
  }
  }

  if (ex) {
    for (int n = 0; n < N; ++n) {
      size_t o = (size_t)r * N + n;
      const float* nd = tree + n * NS;
      const int* ndi = itree + n * NS;
      if (j == 0) {
        p.t_node_visits[o] = ndi[C::HDR0];
        p.t_node_values[o] = nd[C::HDR0 + 1];
      }
      if (j < A) {
        int ci, cv;
        if constexpr (C::PK) {
          ci = reinterpret_cast<const int8_t*>(ndi + C::SEL0)[j];
          cv = reinterpret_cast<const uint8_t*>(ndi + C::VIS0)[j];
        } else {
          ci = ndi[C::SEL0 + 2 * j];
          cv = ndi[C::ST0 + C::STW * j + 2];
        }
        p.t_children_index[o * A + j] = ci;
        p.t_children_values[o * A + j] = nd[C::off_val(j)];
        p.t_children_visits[o * A + j] = cv;
        p.t_children_rewards[o * A + j] = nd[C::off_rew(j)];
        p.t_children_discounts[o * A + j] = ci >= 0 ? p.discount : 0.0f;
      }
      if constexpr (C::EMB_LDS)  // (else: the search ran on t_embeddings itself)
        for (int i = j; i < E; i += 16) p.t_embeddings[o * E + i] = nd[C::EMB0 + i];
    }
  }
  MZ_TICKW(13);  // epilogue: summary, sample, outputs
#ifdef MZ_PROFILE
  if (p.prof != nullptr && lane == 0) {
    uint64_t* dst = p.prof + ((size_t)blockIdx.x * C::WAVES + (tid >> 6)) * 16;
    for (int q = 0; q < 16; ++q) dst[q] = prof_acc[q];
  }
#endif
}

}  // namespace mz
