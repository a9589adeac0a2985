/*
 * mz_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the batched MuZero search that sits behind
 * muax.MuZero.act(): reference call sites muax/model.py:82-179,222-282,
 * muax/policy.py:13-30, muax/nn.py:37-44,59-115, muax/utils.py:65-102.  The
 * search arithmetic itself lives in third-party `mctx` (un-vendored, unpinned:
 * reference setup.py:14), so rows a7-a9 of SURVEY.md section 8(a) are restated
 * from mctx's published algorithm (mctx 0.0.5: policies.muzero_policy,
 * search.{search,simulate,expand,backward}, action_selection.
 * muzero_action_selection, qtransforms.qtransform_by_parent_and_siblings,
 * tree.Tree.summary).
 *
 * PARITY UNPINNED: neither jax nor mctx can be imported in the build container
 * and the reference holds no golden vectors for this path.
 * The SEARCH is pinned only by hand-derived known-answer tests
 * (tests/test_oracle_kat.py) and by an independent NumPy restatement
 * (oracle/mz_numpy.py); the PRNG layer (threefry, split, the vector layout of
 * random_bits, uniform, the erf_inv normal) additionally reproduces every value
 * JAX's own documentation prints (Random123 vectors; split / uniform / normal of
 * PRNGKey(0) and PRNGKey(42); the quickstart's ten normals), to the bit.
 * tests/golden/capture_from_mctx.py captures real reference outputs on a machine
 * that has jax + mctx; tests/test_mctx_pin_cpu.py compares this oracle with them
 * when they exist.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (muax_amd/) never does.
 *
 * Arrays use mctx's own layout: batch-major, row-major ([B,N], [B,N,A],
 * [B,N,E]); float32 / int32 throughout.
 */
#ifndef MZ_ORACLE_H
#define MZ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZO_UNVISITED (-1)
#define MZO_NO_PARENT (-1)

/* mctx.Tree restated (search/tree.py): every pointer is caller-owned. */
typedef struct {
  int32_t B, N, A, E;
  int32_t *node_visits;            /* [B,N]   */
  float *raw_values;               /* [B,N]   */
  float *node_values;              /* [B,N]   */
  int32_t *parents;                /* [B,N]   */
  int32_t *action_from_parent;     /* [B,N]   */
  int32_t *children_index;         /* [B,N,A] */
  float *children_prior_logits;    /* [B,N,A] */
  float *children_values;          /* [B,N,A] */
  int32_t *children_visits;        /* [B,N,A] */
  float *children_rewards;         /* [B,N,A] */
  float *children_discounts;       /* [B,N,A] */
  float *embeddings;               /* [B,N,E] */
  uint8_t *root_invalid_actions;   /* [B,A], 1 = invalid */
} mzo_tree;

/* Search hyper-parameters (muax/model.py:86-95 defaults). */
typedef struct {
  int32_t num_simulations;
  int32_t max_depth;        /* <=0 -> num_simulations (mctx search.py) */
  float pb_c_init;
  float pb_c_base;
  int32_t tiebreak;         /* 0: no tie-break noise; 1: JAX threefry stream */
  int64_t global_batch;     /* B of the un-sharded batch (RNG stream layout) */
  int64_t root_offset;      /* global index of tree root 0 */
} mzo_search_cfg;

/* Default MLP trio, muax/nn.py:59-115; haiku Linear layout w[in][out]. */
typedef struct {
  int32_t obs_dim, E, A, F, H;  /* H = 16 in the reference */
  const float *repr_w, *repr_b;                 /* [obs,E],[E]       */
  const float *pv_w1, *pv_b1, *pv_w2, *pv_b2;   /* [E,H],[H],[H,F],[F] */
  const float *pp_w1, *pp_b1, *pp_w2, *pp_b2;   /* [E,H],[H],[H,A],[A] */
  const float *dr_w1, *dr_b1, *dr_w2, *dr_b2;   /* [E+A,H],[H],[H,F],[F] */
  const float *dn_w1, *dn_b1, *dn_w2, *dn_b2;   /* [E+A,H],[H],[H,E],[E] */
  float discount;
  int32_t support_size;         /* F == 2*support_size+1 */
  int32_t recurrent_pred_on;    /* 0 child (muax/model.py:272), 1 parent (coax :448) */
} mzo_mlp;

/* ---- arithmetic spec (DESIGN.md "MZ-F32") ---- */
float mzo_exp(float x);
float mzo_expm1_neg(float x);
float mzo_elu(float x);
float mzo_log(float x);
float mzo_sum16(const float *x, int n);
void mzo_softmax(const float *x, int n, float *p);
float mzo_inv_scaling(float x);
float mzo_support_to_scalar(const float *probs, int support_size);
void mzo_min_max_normalize(float *s, int n);

/* ---- JAX PRNG restated (threefry2x32, non-partitionable stream) ---- */
void mzo_threefry2x32(const uint32_t key[2], uint32_t x0, uint32_t x1, uint32_t out[2]);
void mzo_split(const uint32_t key[2], int64_t n, int64_t row, uint32_t out[2]);
uint32_t mzo_random_bits(const uint32_t key[2], int64_t size, int64_t i);
float mzo_uniform_from_bits(uint32_t bits);
float mzo_gumbel_from_bits(uint32_t bits);
float mzo_normal(const uint32_t key[2]);   /* jax.random.normal(key, ()) as the gamma sampler draws it */
void mzo_normal_vec(const uint32_t key[2], int64_t n, float *out);   /* jax.random.normal(key, (n,)) */
/* jax.random.dirichlet restated (spec-to-confirm, see mz_oracle.c) */
float mzo_log1p(float x);
float mzo_erf_inv(float x);
float mzo_loggamma_one(const uint32_t key[2], float alpha);
void mzo_dirichlet(const uint32_t key[2], float alpha, int B, int A, int64_t global_batch,
                   int64_t root_offset, float *out);

/* ---- nets ---- */
void mzo_root_inference(const mzo_mlp *m, const float *obs, float *embedding,
                        float *prior_logits, float *value);
void mzo_recurrent_inference(const mzo_mlp *m, int action, const float *embedding,
                             float *reward, float *discount, float *prior_logits,
                             float *value, float *next_embedding);

/* ---- search pieces, one root b at a time ---- */
void mzo_root_prior(const float *prior_logits, int A, const float *dirichlet_noise,
                    float dirichlet_fraction, const uint8_t *invalid, float *out_logits);
void mzo_tree_init(mzo_tree *t, const float *prior_logits, const float *value,
                   const float *embedding, const uint8_t *invalid);
void mzo_action_scores(const mzo_tree *t, int b, int node, const mzo_search_cfg *cfg,
                       float *value_score, float *policy_score);
void mzo_simulate_injected(const mzo_tree *t, int b, const mzo_search_cfg *cfg,
                           const uint32_t root_key[2], const float *uniforms, int D,
                           int32_t *parent_out, int32_t *action_out, int32_t *depth_out);
void mzo_step_select_injected(const mzo_tree *t, const mzo_search_cfg *cfg, int sim,
                              const uint32_t sim_key[2], const float *uniforms, int D,
                              int32_t *parent_out, int32_t *action_out, int32_t *depth_out);
int mzo_select_action(const mzo_tree *t, int b, int node, int depth,
                      const mzo_search_cfg *cfg, const float *noise);
void mzo_simulate(const mzo_tree *t, int b, const mzo_search_cfg *cfg,
                  const uint32_t root_key[2], int32_t *parent_out, int32_t *action_out,
                  int32_t *depth_out);
void mzo_expand(mzo_tree *t, int b, int parent, int action, int next,
                float reward, float discount, const float *prior_logits, float value,
                const float *next_embedding);
void mzo_backward(mzo_tree *t, int b, int leaf);
void mzo_summary_sample(const mzo_tree *t, int b, float temperature, const float *gumbel,
                        int32_t *action_out, float *action_weights_out);

/* ---- Gumbel MuZero (muax/policy.py:33-47 -> mctx.gumbel_muzero_policy, restated from mctx 0.0.5:
 * policies.gumbel_muzero_policy, action_selection.gumbel_muzero_{root,interior}_action_selection,
 * seq_halving.{score_considered,get_sequence_of_considered_visits}, qtransforms.
 * qtransform_completed_by_mix_value).  qtransform: 0 = by_parent_and_siblings (what muax/model.py:230-231
 * forces onto every policy), 1 = completed_by_mix_value (mctx's own default for this policy). ---- */
void mzo_qtransform(const mzo_tree *t, int b, int node, int qtransform, float *out);
void mzo_considered_visits(int max_num_considered_actions, int num_simulations, int32_t *seq);
int mzo_gumbel_select_action(const mzo_tree *t, int b, int node, int depth, int qtransform,
                             const float *root_gumbel, int num_simulations,
                             int max_num_considered_actions);
void mzo_gumbel_step_select(const mzo_tree *t, const mzo_search_cfg *cfg, int qtransform,
                            const float *root_gumbel, int max_num_considered_actions,
                            int32_t *parent_out, int32_t *action_out, int32_t *depth_out);
void mzo_gumbel_finish(const mzo_tree *t, int b, int qtransform, const float *root_gumbel,
                       const float *root_logits, int32_t *action_out, float *action_weights_out);

/* ---- stepwise driver (any recurrent_fn supplied by the caller) ---- */
void mzo_step_select(const mzo_tree *t, const mzo_search_cfg *cfg, int sim,
                     const uint32_t sim_key[2], int32_t *parent_out,
                     int32_t *action_out, int32_t *depth_out);
void mzo_step_expand_backup(mzo_tree *t, int sim, const int32_t *parent,
                            const int32_t *action, const float *reward,
                            const float *discount, const float *prior_logits,
                            const float *value, const float *next_embedding);

/* ---- whole act() for the default MLP trio ----
 * key: the rng_key handed to MuZero.act (uint32[2]).
 * dirichlet_noise [B,A] or NULL (NULL => fraction must be 0).
 * gumbel [B,A] or NULL (NULL => drawn from the key as jax.random.categorical).
 * depth_sum_out [B] or NULL: sum over simulations of selection depth (D of
 * SURVEY.md 8(d)).  nthreads <= 1 -> serial. */
void mzo_act_mlp(const mzo_mlp *m, const mzo_search_cfg *cfg, mzo_tree *t,
                 const float *obs, const uint32_t key[2],
                 const float *dirichlet_noise, float dirichlet_fraction,
                 const uint8_t *invalid_actions, float temperature,
                 const float *gumbel, int32_t *action_out, float *action_weights_out,
                 float *root_value_out, int64_t *depth_sum_out, int nthreads);

/* ---- checker for a kernel-side identity (test infrastructure, not part of the restatement) ----
 * muax_amd/csrc/mz_fused.cuh divides by small integers d (visit counts) as
 *   q0 = x * y;  r = fma(-q0, d, x);  q = fma(r, y, q0)     with y = RN(1 / d)   (Markstein)
 * where this oracle writes x / d.  Returns how many of the 2^24 binary32 values of the given exponent
 * (every mantissa, both signs) give q != x / d, summed over d = 1 .. dmax.  0 means the two agree for
 * every normal x (scaling by a power of two changes neither side). */
int64_t mzo_markstein_mismatches(int dmax, int exponent);
int64_t mzo_div2eps_mismatches(int e_lo, int e_hi);
float mzo_elu_clamped(float x);

#ifdef __cplusplus
}
#endif
#endif
