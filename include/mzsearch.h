/*
 * mzsearch.h -- C-ABI of the MI355X-native batched MuZero search.
 *
 * This is the drop-in boundary for the one path this repository accelerates:
 * what muax.MuZero._plan hands to mctx.muzero_policy (reference
 * muax/model.py:222-243, muax/policy.py:13-30) and the two callbacks mctx makes
 * into muax (root inference muax/model.py:251-263, recurrent inference
 * muax/model.py:265-282).  The reference has no FFI of its own (it is pure
 * Python on JAX); INTEGRATION.md shows the ctypes stub a muax maintainer would
 * add.  Plain pointers and sizes only; every pointer is a DEVICE pointer owned
 * by the caller (PyTorch) unless stated, borrowed for the duration of the call.
 * All kernels are enqueued on the caller's HIP stream (`stream` is a
 * hipStream_t passed as void*); no entry point synchronises.
 *
 * There is NO CPU fallback behind this ABI: mzs_create fails without a gfx950
 * device.
 *
 * Layout of batched arrays follows mctx: row-major [B], [B,A], [B,E], and for
 * tree views [B,N], [B,N,A], [B,N,E] with N = num_simulations + 1.
 */
#ifndef MZSEARCH_H
#define MZSEARCH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZS_ABI_VERSION 1

enum {
  MZS_OK = 0,
  MZS_E_INVALID = -1,     /* bad argument (shape, null, range) -> ValueError */
  MZS_E_UNSUPPORTED = -2, /* configuration has no kernel instance             */
  MZS_E_RUNTIME = -3,     /* HIP runtime error                                 */
  MZS_E_NODEVICE = -4     /* no gfx950 device                                  */
};

typedef struct mzs_handle mzs_handle;

/* Search configuration: the keyword arguments of MuZero.act that reach
 * mctx.muzero_policy (muax/model.py:82-96), plus the batch geometry. */
typedef struct {
  int32_t struct_size;     /* sizeof(mzs_config), for ABI checking */
  int32_t device;          /* HIP device ordinal */
  int32_t batch;           /* B: roots held by this handle (this GPU's shard) */
  int32_t num_actions;     /* A */
  int32_t num_simulations; /* S */
  int32_t embed_dim;       /* E: flattened embedding elements per node */
  int32_t max_depth;       /* <= 0: num_simulations (mctx default) */
  int32_t qtransform;      /* 0: qtransform_by_parent_and_siblings; 1: qtransform_completed_by_mix_value (gumbel policy only) */
  int32_t tiebreak;        /* 0: none; 1: JAX threefry stream (1e-7 * uniform); muzero policy only */
  int32_t policy;          /* 0: mctx.muzero_policy (muax/policy.py:13-30); 1: mctx.gumbel_muzero_policy (muax/policy.py:33-47) */
  float pb_c_init;         /* 1.25  */
  float pb_c_base;         /* 19652 */
  int64_t global_batch;    /* B of the un-sharded batch (PRNG stream layout); 0 -> batch */
  int64_t root_offset;     /* global index of local root 0 */
  int32_t max_num_considered_actions; /* gumbel policy: 16 (muax/policy.py:45) */
  float gumbel_scale;                 /* gumbel policy: 1.0 (muax/policy.py:46) */
} mzs_config;

/* Weights of the default MLP trio (muax/nn.py:59-115), haiku layout w[in][out],
 * float32 device pointers. hidden width is 16 as in the reference. */
typedef struct {
  int32_t struct_size;
  int32_t obs_dim;
  int32_t support_size;       /* F = 2*support_size+1 */
  int32_t recurrent_pred_on;  /* 0: child embedding (muax/model.py:272); 1: parent (frameworks/coax/model.py:448) */
  float discount;
  float reserved0;
  const float *repr_w, *repr_b;               /* [obs,E] [E]   */
  const float *pv_w1, *pv_b1, *pv_w2, *pv_b2; /* [E,16] [16] [16,F] [F] */
  const float *pp_w1, *pp_b1, *pp_w2, *pp_b2; /* [E,16] [16] [16,A] [A] */
  const float *dr_w1, *dr_b1, *dr_w2, *dr_b2; /* [E+A,16] [16] [16,F] [F] */
  const float *dn_w1, *dn_b1, *dn_w2, *dn_b2; /* [E+A,16] [16] [16,E] [E] */
} mzs_mlp_weights;

/* Caller-owned output arrays in mctx.Tree layout (search_tree of PolicyOutput).
 * Either every pointer is set or the struct pointer is NULL. */
typedef struct {
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
} mzs_tree_view;

/* Per-call arguments of the fused act path. */
typedef struct {
  int32_t struct_size;
  int32_t reserved0;
  const float *obs;               /* [B,obs_dim] */
  const float *dirichlet_noise;   /* [B,A] or NULL (then dirichlet_fraction must be 0) */
  const uint8_t *invalid_actions; /* [B,A] 1 = invalid, or NULL */
  const float *gumbel;            /* [B,A] or NULL: drawn from `key` (muzero policy: the categorical's Gumbel;
                                     gumbel policy: the ROOT Gumbel noise, gumbel_scale * gumbel(split(key)[1])) */
  uint32_t key[2];                /* the rng_key given to MuZero.act (HOST values) */
  float dirichlet_fraction;       /* 0.25 */
  float temperature;              /* 1.0  */
  int32_t *action;                /* out [B] */
  float *action_weights;          /* out [B,A] */
  float *root_value;              /* out [B]: network value of the root (muax/model.py:243) */
  float *search_value;            /* out [B] or NULL: node_values[:,0] after search */
  int32_t *depth_sum;             /* out [B] or NULL: sum over simulations of selection depth */
  const mzs_tree_view *tree;      /* or NULL */
} mzs_act_args;

int mzs_abi_version(void);
const char *mzs_last_error(const mzs_handle *h); /* h may be NULL: last create error */

int mzs_create(const mzs_config *cfg, mzs_handle **out);
int mzs_destroy(mzs_handle *h);

/* ---- fused path: whole act() for the default MLP trio in one launch ---- */
int mzs_mlp_set_weights(mzs_handle *h, const mzs_mlp_weights *w);
int mzs_act_mlp(mzs_handle *h, const mzs_act_args *args, void *stream);

/* ---- the same from HOST memory: what the reference's act() does around its jitted _plan ----
 * muax.MuZero.act takes NumPy observations and returns NumPy / Python values, synchronising on the way out
 * (np.asarray / .item(), muax/model.py:160-179).  mzs_act_mlp_host is that whole round trip in one call: the
 * observations (and optional masks / noise) are copied into pinned memory owned by the handle, which the search kernel
 * reads -- and whose output half it writes -- directly (mapped host memory: no upload / download command), the root
 * noise is drawn on the device from split(key, 3)[1] unless given (mctx.muzero_policy's jax.random.dirichlet, see
 * mzs_dirichlet), and the call returns after synchronising `stream`.  All pointers here are HOST pointers.  (The handle allocates its staging buffers at the
 * first call; this is the only entry point that synchronises.) */
typedef struct {
  int32_t struct_size;
  int32_t draw_dirichlet;              /* muzero policy: 1 = draw the root noise from `key` with dirichlet_alpha */
  const float *obs;                    /* [B, obs_dim] */
  const float *dirichlet_noise;        /* [B, A] or NULL (NULL and draw_dirichlet == 0: no noise, fraction forced to 0) */
  const uint8_t *invalid_actions;      /* [B, A] or NULL */
  uint32_t key[2];
  float dirichlet_fraction, dirichlet_alpha, temperature, reserved0;
  int32_t *action;                     /* out [B] */
  float *action_weights;               /* out [B, A] */
  float *root_value;                   /* out [B] */
} mzs_act_host_args;
int mzs_act_mlp_host(mzs_handle *h, const mzs_act_host_args *args, void *stream);

/* ---- step-wise path: any repr/pred/dyn plugin nets run by the caller ----
 * mzs_root        <- RootFnOutput(prior_logits, value, embedding)  (muax/model.py:258-262)
 * mzs_select      -> (parent embedding, action) for recurrent_fn   (mctx simulate + expand gather)
 * mzs_expand_backup <- RecurrentFnOutput + next embedding            (muax/model.py:276-282)
 * mzs_finish      -> PolicyOutput(action, action_weights)           (mctx summary + categorical)
 * `key` is the act rng_key (host values); used only when tiebreak != 0 or gumbel == NULL. */
int mzs_root(mzs_handle *h, const float *prior_logits, const float *value,
             const float *embedding, const uint8_t *invalid_actions,
             const float *dirichlet_noise, float dirichlet_fraction,
             const uint32_t key[2], void *stream);
/* gumbel policy root: only mctx's invalid-action mask is applied to the logits; `gumbel` [B,A] is
 * the root Gumbel noise or NULL to draw gumbel_scale * jax.random.gumbel(split(key)[1], [B,A]). */
int mzs_root_gumbel(mzs_handle *h, const float *prior_logits, const float *value,
                    const float *embedding, const uint8_t *invalid_actions,
                    const float *gumbel, const uint32_t key[2], void *stream);
int mzs_select(mzs_handle *h, int32_t sim, int32_t *action_out,
               float *parent_embedding_out, void *stream);
int mzs_expand_backup(mzs_handle *h, int32_t sim, const float *reward,
                      const float *discount, const float *prior_logits,
                      const float *value, const float *next_embedding, void *stream);
/* mzs_expand_backup(sim) and mzs_select(sim + 1) in ONE launch (the workgroup that refreshed the root's cached
 * decision performs the next simulate() and gathers its parent's embedding row): one launch and one kernel boundary
 * fewer per simulation of a launch-bound loop.  After the last simulation only the expand + backward half runs and the
 * outputs are left untouched.  The caller must not call mzs_select for sim + 1 afterwards.  Same results as the two
 * separate calls, bit for bit. */
int mzs_expand_backup_select(mzs_handle *h, int32_t sim, const float *reward,
                             const float *discount, const float *prior_logits,
                             const float *value, const float *next_embedding,
                             int32_t *next_action_out, float *next_parent_embedding_out,
                             void *stream);
int mzs_finish(mzs_handle *h, float temperature, const float *gumbel,
               int32_t *action_out, float *action_weights_out,
               float *search_value_out, int32_t *depth_sum_out, void *stream);
int mzs_tree_export(mzs_handle *h, const mzs_tree_view *out, void *stream);

/* ---- recurrent_fn of the EfficientZero-style nets ----
 * mzs_ez_recurrent evaluates, for a batch of 6x6xC hidden states (NHWC, C = 32 or 64) and actions, the whole
 * MuZero._recurrent_inference (muax/model.py:265-282) of EZDynamic + EZPrediction with use_v2 = True
 * (muax/nn.py:221-309, pre-activation blocks muax/nn.py:151-178) in ONE launch: next state, reward and value as
 * support_to_scalar(softmax(logits)), prior logits.  Weights, caller-owned device arrays:
 *   LayerNorm       [2][channels]                  scale row, offset row
 *   3x3 convolution [9][Cin / 16][4][C][4]         Wp[tap][c][g][co][i] = W[tap][16 c + 4 g + i][co] from haiku's
 *                                                  HWIO w[kh][kw][in][out] (the layout of mzs_resnet_tower); the
 *                                                  dynamics' first convolution has C + 1 inputs (last: the action
 *                                                  plane), zero-padded to Cin = C + 16
 *   head            ln_in [2][C], c1 [C][16] (1x1 conv), ln_mid [2][16], fc [576][32] (no bias), ln_vec [2][32],
 *                   out_w [32][n], out_b [n] with n = 2 support_size + 1 (reward, value) or num_actions (policy) */
typedef struct mzs_ez_head {
  const float *ln_in, *c1, *ln_mid, *fc, *ln_vec, *out_w, *out_b;
} mzs_ez_head;
typedef struct mzs_ez_args {
  int32_t struct_size;     /* = sizeof(mzs_ez_args) */
  int32_t device;
  int32_t batch;
  int32_t channels;        /* C: 32 or 64 */
  int32_t num_actions;     /* <= 64 */
  int32_t support_size;    /* 2 support_size + 1 <= 64 */
  const float *x;          /* [B, 6, 6, C] */
  const int32_t *action;   /* [B]; the plane holds the raw index (muax/nn.py:291-296) */
  float *y;                /* [B, 6, 6, C] next state out */
  float *reward;           /* [B] out */
  float *value;            /* [B] out */
  float *prior_logits;     /* [B, A] out */
  const float *d_ln_in, *d_conv;                       /* EZDynamic: LN before the action conv, the conv */
  const float *d_ln0, *d_conv0, *d_ln1, *d_conv1;      /* ... its residual block */
  const float *p_ln0, *p_conv0, *p_ln1, *p_conv1;      /* EZPrediction's residual block */
  mzs_ez_head r, v, p;                                 /* reward / value / policy heads */
} mzs_ez_args;
int mzs_ez_recurrent(const mzs_ez_args *a, void *stream);

/* ---- fused LayerNorm of the convolutional plugin nets ----
 * y = [relu]( LN(x) [+ LN2(x2)] [+ residual] ) with LN(x) = (x - mean) * rsqrt(var + eps) * scale[c] + offset[c],
 * statistics over ALL n elements of a sample (biased variance), scale / offset indexed by (element % channels):
 * hk.LayerNorm(axis=(-3,-2,-1), create_scale=True, create_offset=True) of NHWC tensors followed by the shortcut
 * addition and relu of ResidualConvBlockV1/V2 and of the EZ heads (muax/nn.py:118-178, 232-288) -- the chains between
 * the convolutions of the plugin nets' root inference, two launches instead of ~10 framework kernels each.
 * Inference only (no gradients).  All tensors fp32, contiguous, caller-owned; `workspace` >=
 * mzs_layernorm_workspace_bytes(batch, n) bytes of device memory, overwritten. */
typedef struct mzs_layernorm_args {
  int32_t struct_size;   /* = sizeof(mzs_layernorm_args) */
  int32_t device;
  int32_t batch;         /* samples */
  int32_t n;             /* elements per sample (H*W*C), multiple of 4 and of channels */
  int32_t channels;      /* C, multiple of 4 */
  int32_t relu;          /* 1: relu at the end */
  float eps;             /* haiku: 1e-5 */
  const float *x, *scale, *offset;      /* [batch, n], [C], [C] */
  const float *x2, *scale2, *offset2;   /* optional second normalised tensor (projected shortcut), or NULL */
  const float *residual;                /* optional plain tensor added (identity shortcut), or NULL */
  float *y;                             /* [batch, n]; may alias x, x2 or residual */
  void *workspace;
  int64_t workspace_bytes;
} mzs_layernorm_args;
int mzs_layernorm_act(const mzs_layernorm_args *a, void *stream);
int64_t mzs_layernorm_workspace_bytes(int32_t batch, int32_t n);

/* ---- device self-test ----
 * Three places of the kernels replace a library sequence by a shorter one that is only valid for this hardware's
 * v_sqrt_f32 / v_rcp_f32 / fma: sqrt on arguments that need no range scaling, division by 2 eps = 0.002f
 * (muax/utils.py:70-76), and the support decode's e_i / sum with one refined reciprocal per sum.  mzs_selftest compares
 * them with the IEEE operations on the device -- exhaustively over [1, 4) resp. 2^-9 .. 2^-2, over 2^24
 * denominators in [1, 64) with twelve numerators each, and over 2^24 denominators in 2^-27 .. 2^41 with eight -- and
 * returns the mismatch counts (all must be 0) in mismatches[0..3].  Synchronous.  errors: mzs_last_error(NULL) */
int mzs_selftest(int32_t device, int64_t *mismatches);

/* ---- root exploration noise ----
 * rows [root_offset, root_offset + batch) of what mctx.muzero_policy draws for a `global_batch`-root act:
 *   jax.random.dirichlet(split(rng_key, 3)[1], alpha = full([num_actions], dirichlet_alpha), shape = (global_batch,))
 * (policy call site muax/policy.py:18-30, defaults muax/model.py:92-93), into out [batch, num_actions] on the
 * device.  `key` is the DIRICHLET sub-key (host values).  Restated from jax's published sampler on the exact
 * threefry key walk; float bits are spec-to-confirm against a real jax (oracle/mz_oracle.c): inject an array
 * through mzs_act_args.dirichlet_noise / mzs_root for bit-pinned noise.  errors: mzs_last_error(NULL) */
int mzs_dirichlet(int32_t device, const uint32_t key[2], float alpha, int32_t batch, int32_t num_actions,
                  int64_t global_batch, int64_t root_offset, float *out, void *stream);

/* ---- training step of the default MLP trio ----
 * mzs_mlp_loss_grad replaces jax.value_and_grad(loss_fn) at muax/model.py:245-249 with the default
 * loss (muax/loss.py:10-88): for a batch of k-step trajectories it returns the scalar loss and the
 * gradient of every one of the 18 weight arrays, concatenated in the member order of mzs_mlp_weights
 * (repr_w, repr_b, pv_w1, ... dn_b2), each in its own haiku layout.  One fused forward+backward kernel
 * and a fixed-order reduction: bit-reproducible run to run.  The optimiser (muax/optimizers.py) and the
 * optional data-parallel gradient mean (one flat all-reduce over `grads`) stay with the caller. */
typedef struct mzs_train_args {
  int32_t struct_size;      /* = sizeof(mzs_train_args) */
  int32_t device;
  int32_t batch;            /* B trajectories */
  int32_t unroll_steps;     /* L = k_steps (muax/replay_buffer.py:192-240 batch layout [B, L, ...]) */
  int32_t num_actions;
  int32_t embed_dim;
  const float *obs;         /* [B, obs_dim]  batch.obs[:, 0] */
  const int32_t *actions;   /* [B, L] */
  const float *rewards;     /* [B, L]   batch.r  */
  const float *returns;     /* [B, L]   batch.Rn */
  const float *policy;      /* [B, L, A] batch.pi */
  float loss_scale;         /* 1/B (muax/loss.py) or 1/(B L) (frameworks/coax/loss.py:70-71) */
  float l2_coeff;           /* 1e-4 (muax/loss.py:84-87) */
  float *loss;              /* [1] out */
  float *grads;             /* [mzs_mlp_num_params] out */
  void *workspace;          /* >= mzs_mlp_train_workspace_bytes(...) */
  int64_t workspace_bytes;
} mzs_train_args;
int64_t mzs_mlp_num_params(int32_t obs_dim, int32_t embed_dim, int32_t num_actions, int32_t support_size);
int64_t mzs_mlp_train_workspace_bytes(int32_t batch, int32_t obs_dim, int32_t embed_dim,
                                      int32_t num_actions, int32_t support_size);
/* errors: mzs_last_error(NULL) */
int mzs_mlp_loss_grad(const mzs_mlp_weights *w, const mzs_train_args *a, void *stream);

/* ---- next-state tower of the ResNet dynamics net ----
 * mzs_resnet_tower evaluates, for a batch of 6x6x64 hidden states (NHWC, the reference's embedding
 * layout), what muax/nn.py:344-378 calls ns_func followed by min_max_normalize2d: conv1x1 on
 * [s, a / num_actions] + relu, `blocks` x ResidualConvBlockV1(64, stride 1, projection)
 * (muax/nn.py:118-148: 3 x conv3x3 + LayerNorm over (H, W, C)), per-channel min-max normalisation -- one
 * kernel, one workgroup per root, fp32 MFMA.  Weights stay in haiku's layouts:
 *   stem_w  [65][64]                 (hk.Conv2D 1x1, HWIO)  or NULL to skip the stem
 *   conv_w  [blocks][3] convolutions (projection conv, conv_0, conv_1), each haiku HWIO w[3][3][64][64]
 *           re-ordered once by the caller to Wp[tap 9][c 4][g 4][co 64][i 4] = w[tap][16 c + 4 g + i][co]
 *           (one 16-byte load per lane and 4 k-steps)
 *   ln      [blocks][3][2][64]        (scale, offset) of the projection's, ln_0's, ln_1's LayerNorm */
typedef struct mzs_tower_args {
  int32_t struct_size;     /* = sizeof(mzs_tower_args) */
  int32_t device;
  int32_t batch;
  int32_t blocks;
  int32_t normalize;       /* != 0: apply min_max_normalize2d at the end */
  int32_t num_actions;     /* the action plane is a / num_actions */
  const float *x;          /* [B, 6, 6, 64] */
  const int32_t *action;   /* [B] (needed with stem_w) */
  const float *stem_w;
  const float *conv_w;
  const float *ln;
  float *y;                /* [B, 6, 6, 64] out */
  /* Optional heads, fused into the same launch when r_c1 != NULL (then all of them must be given): the
   * whole recurrent_fn of muax/model.py:265-282 for the ResNet nets -- reward head r_func on
   * [s, a / num_actions] (muax/nn.py:347-357) and ResNetPrediction on the normalised next state
   * (muax/nn.py:313-341); reward / value come out as support_to_scalar(softmax(logits)).  haiku layouts:
   * conv1x1 w[in][out], Linear w[in][out] on the NHWC-flattened map, biases [out]. */
  const float *r_c1, *r_c2, *r_l1, *r_b1, *r_l2, *r_b2;   /* [65,64] [64,64] [2304,64] [64] [64,F] [F] */
  const float *v_c1, *v_c2, *v_l1, *v_b1, *v_l2, *v_b2;   /* [64,16] [16,16] [576,16] [16] [16,F] [F] */
  const float *p_c1, *p_l1, *p_b1, *p_l2, *p_b2;          /* [64,16] [576,16] [16] [16,A] [A]        */
  float *reward;           /* [B] out */
  float *value;            /* [B] out */
  float *prior_logits;     /* [B, A] out */
  int32_t support_size;    /* F = 2 * support_size + 1 <= 64 */
  int32_t reserved0;
  /* Optional pair mode for small batches (2 * batch workgroups must be resident at once: batch <= 128):
   * two workgroups per root split the pixels of its map and swap boundary pixels + LayerNorm moments
   * through `pair_scratch` (caller-owned device memory of mzs_tower_pair_scratch_bytes(batch) bytes,
   * ZEROED once when allocated and then left to the library; one scratch per stream).  NULL: one
   * workgroup per root.  Word 4 r + 3 of the trailing uint32 region turns non-zero if root r's halves ever
   * lost each other (bounded spin ran out): results of that launch are then invalid. */
  void *pair_scratch;
  int64_t pair_scratch_bytes;
} mzs_tower_args;
int mzs_resnet_tower(const mzs_tower_args *a, void *stream);
/* bytes of pair_scratch for `batch` roots (0 if pair mode cannot run that batch) */
int64_t mzs_tower_pair_scratch_bytes(int32_t batch);

/* The tail of root inference with those nets (muax/model.py:251-263), one launch: the last hk.AvgPool(3, 2, 'SAME') of
 * ResNetRepresentation (muax/nn.py:308; the mean of the VALID elements under each window) on its [B, H, W, 64] map with
 * H, W in {11, 12} -> 6 x 6, min_max_normalize2d (:47-56, :310; normalize != 0), ResNetPrediction on that embedding
 * (:313-341; weights as in mzs_tower_args) and support_to_scalar(softmax(value logits)) (muax/model.py:254). */
typedef struct mzs_root_tail_args {
  int32_t struct_size;     /* = sizeof(mzs_root_tail_args) */
  int32_t device;
  int32_t batch, height, width;
  int32_t num_actions, support_size, normalize;
  const float *x;          /* [B, H, W, 64] */
  const float *v_c1, *v_c2, *v_l1, *v_b1, *v_l2, *v_b2, *p_c1, *p_l1, *p_b1, *p_l2, *p_b2;
  float *embedding;        /* [B, 6, 6, 64] out */
  float *value;            /* [B] out */
  float *prior_logits;     /* [B, num_actions] out */
} mzs_root_tail_args;
int mzs_resnet_root_tail(const mzs_root_tail_args *a, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * hk.Conv2D(C, kernel_shape=3, stride=1, padding='SAME', with_bias=False) on NHWC maps with C -> C channels, C = 16, 32
 * or 64: the convolutions inside the residual blocks of the representation nets at their 21 x 21 / 11 x 11 / 6 x 6 stages
 * (muax/nn.py:118-178 inside ResNetRepresentation :291-310 and EZStateEncoder :180-207; root inference,
 * muax/model.py:251-263).  fp32 MFMA implicit GEMM (mz_repr.cuh); any height / width whose rows fit a CU's LDS.
 *   w_packed: the HWIO kernel w[3][3][C][C] re-ordered once by the caller to Wp[tap][c][g][co][i] = w[tap][16 c + 4 g + i][co]
 *   (the layout of mzs_resnet_tower's conv_w).  relu != 0: max(., 0) on the way out. */
typedef struct mzs_conv3x3_args {
  int32_t struct_size;     /* = sizeof(mzs_conv3x3_args) */
  int32_t device;
  int32_t batch, height, width, channels;
  int32_t relu, reserved0;
  const float *x;          /* [B, H, W, C] */
  const float *w_packed;   /* 9 * C * C floats */
  float *y;                /* [B, H, W, C] out */
} mzs_conv3x3_args;
int mzs_conv3x3_nhwc(const mzs_conv3x3_args *a, void *stream);

/* The stems of those nets: hk.Conv2D(out, kernel_shape=3, stride=2, padding='SAME', with_bias=False) with (in, out)
 * channels (4, 32) / (4, 16) -- raw frame stacks, muax/nn.py:299 / :189 -- or (32, 64) / (16, 32) (muax/nn.py:303, and
 * the strided convolutions of the EZ encoder's projection block, :151-178 inside :180-207); output
 * [B, ceil(H / 2), ceil(W / 2), out].  w_packed: Wp[tap][c][g][co][i] = w[tap][16 c + 4 g + i][co] with the 4 frame
 * channels padded to 16 by zero rows (9 * 16 * 32 floats).  in_div != 0: the input is divided by it on the way in (the
 * reference's observations / 255); relu != 0: max(., 0) on the way out. */
typedef struct mzs_conv3x3s_args {
  int32_t struct_size;     /* = sizeof(mzs_conv3x3s_args) */
  int32_t device;
  int32_t batch, height, width;   /* of the input */
  int32_t in_channels, out_channels;
  int32_t relu;
  float in_div;
  int32_t reserved0;
  const float *x;          /* [B, H, W, in] */
  const float *w_packed;
  float *y;                /* [B, ceil(H/2), ceil(W/2), out] */
} mzs_conv3x3s_args;
int mzs_conv3x3_stride2_nhwc(const mzs_conv3x3s_args *a, void *stream);

/* A whole ResidualConvBlockV1 (muax/nn.py:118-148: conv_0 - LN - relu - conv_1 - LN, + LN(projection conv) or + x,
 * relu) of those nets, stride 1, C -> C with C = 32 or 64, inference, in three launches: the projection and conv_0
 * share one pass over the input and leave the moments of their outputs, conv_1 normalises its input on the way into
 * LDS, one streaming pass applies the last LayerNorm(s), the shortcut and the relu (mz_repr.cuh, mz_norm.cuh).
 *   w_proj / w0 / w1: packed kernels as for mzs_conv3x3_nhwc; w_proj NULL = identity shortcut (y = relu(LN1(.) + x)).
 *   workspace >= mzs_resblock_workspace_bytes(...) bytes of device memory, overwritten; y must not alias x. */
typedef struct mzs_resblock_args {
  int32_t struct_size;     /* = sizeof(mzs_resblock_args) */
  int32_t device;
  int32_t batch, height, width, channels;
  float eps;               /* haiku: 1e-5 */
  int32_t reserved0;
  const float *x;          /* [B, H, W, C] */
  const float *w_proj, *w0, *w1;
  const float *proj_scale, *proj_offset;   /* [C] each; NULL with w_proj NULL */
  const float *ln0_scale, *ln0_offset, *ln1_scale, *ln1_offset;
  float *y;                /* [B, H, W, C] out */
  void *workspace;
  int64_t workspace_bytes;
} mzs_resblock_args;
int mzs_resblock_v1(const mzs_resblock_args *a, void *stream);
int64_t mzs_resblock_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t channels);

/* A whole ResidualConvBlockV2 (muax/nn.py:151-178: the pre-activation block of the EfficientZero-style encoder,
 * muax/nn.py:180-207) with the identity shortcut, stride 1, C -> C with C = 16, 32 or 64, inference, in three launches:
 *     y = x + conv_1(relu(LN_1(conv_0(relu(LN_0(x))))))
 * fp64 moments of x; conv_0 normalising x on its way into LDS and leaving the moments of its outputs; conv_1
 * normalising those on the way in and adding x in its epilogue.  Same argument block as mzs_resblock_v1 with
 * w_proj / proj_scale / proj_offset NULL (the reference's projection block of this kind is strided: single calls);
 * workspace >= mzs_resblock_v2_workspace_bytes(...); y must not alias x. */
int mzs_resblock_v2(const mzs_resblock_args *a, void *stream);
int64_t mzs_resblock_v2_workspace_bytes(int32_t batch, int32_t height, int32_t width, int32_t channels);

/* ------------------------------------------------------------------------------------------------------------
 * Fused-kernel instances built on demand.
 *
 * mzs_act_mlp serves the (num_actions, embedding_dim, support_size, num_simulations) shapes compiled into the library
 * (muax_amd/csrc/mz_instances.def) and returns MZS_E_UNSUPPORTED for others, although the reference's act() takes any
 * (muax/model.py:82-96).  A host that has hipcc can close the gap at run time: compile muax_amd/csrc/mz_fused_jit.hip for
 * the missing shape into a side library (muax_amd/_jit.py does), dlopen it and pass its
 * two entry points (the values of `mzs_jit_dispatch` and `mzs_jit_abi`) here; later mzs_act_mlp calls (any handle) try the registered instances after the
 * built-in ones.  `jit_abi` must equal mzs_fused_jit_abi() (same kernel-argument layout). */
int mzs_register_fused_dispatch(void *dispatch, int32_t jit_abi);
/* Round 6: an instance compiled with -DMZ_FUSED_MUZERO_ONLY=1 serves the MuZero policy's modes alone (its record has four
 * words per child instead of the Gumbel modes' five: more roots per workgroup, muax_amd/_jit.py::plan(gumbel=False)) and
 * declines a Gumbel handle.  Registered here it is tried BEFORE the instances registered with the call above, so that a
 * MuZero-policy handle takes it even when an all-modes instance of the same shape is present. */
int mzs_register_fused_dispatch_muzero(void *dispatch, int32_t jit_abi);
int mzs_fused_jit_abi(void);
/* The same for the training step (round 5): mzs_mlp_loss_grad carries mz_train_kernel for a list of (num_actions,
 * embedding_dim, 2 support_size + 1) triples and returns MZS_E_UNSUPPORTED for others, although the reference's
 * update() takes whatever widths its nets have (muax/model.py:181-201).  muax_amd/csrc/mz_train_jit.hip compiled for
 * the missing triple gives a side library whose `mzs_jit_train_launch` is passed here together with its
 * `mzs_jit_train_shape` values and the value of its `mzs_jit_train_abi` (must equal mzs_train_jit_abi(): same argument
 * layout); later
 * mzs_mlp_loss_grad calls of that triple take it. */
int mzs_register_train_dispatch(void *launch, int32_t num_actions, int32_t embed_dim, int32_t full_support_size, int32_t jit_abi);
int mzs_train_jit_abi(void);
/* Shapes the fused kernel cannot be instantiated for at all (more than 16 actions, more than 255 simulations, embeddings
 * wider than 64): allow != 0 lets mzs_act_mlp / mzs_act_mlp_host serve them through the generic route instead of
 * returning MZS_E_UNSUPPORTED -- the trio with run-time shapes (num_actions <= 64, support_size 8..31), tree in HBM with
 * cached decisions, ONE launch for all simulations plus root / select / finish launches (mz_mlp_generic.cuh).  Same
 * arithmetic spec, same results bit for bit as an instance would give; several times slower per simulation than a
 * tuned instance, an order of magnitude faster than per-simulation launches with the caller's own nets. */
int mzs_mlp_allow_generic(mzs_handle *h, int32_t allow);

/* ------------------------------------------------------------------------------------------------------------
 * The simulation loop of a search with the ResNet nets in ONE launch (mz_search_conv.hip).
 *
 * Replaces, for simulations [sim_begin, sim_end) of a search on handle `h` (rooted with mzs_root / mzs_root_gumbel and
 * with simulate() of `sim_begin` already run: mzs_select(h, sim_begin, ...), or the tail of a previous call), the loop
 *     recurrent_fn (mzs_resnet_tower with heads)  ->  mzs_expand_backup_select                (2 launches per simulation)
 * that mirrors mctx's search loop calling muax's recurrent_fn (muax/model.py:265-282 through muax/policy.py:13-30).
 * Every root is advanced by its own workgroup(s) through all the simulations: the next state is written into the tree's
 * embedding row of the new node, the next pass reads the parent's row in place, reward / value / prior logits stay on
 * the CU.  Same device code as the step-wise entry points, same results bit for bit.
 *
 * `a`: the weights, `num_actions`, `support_size`, `blocks`, `normalize` (must be set), all 17 head arrays and the three
 * per-root output arrays `reward` [B], `value` [B], `prior_logits` [B, A] (used as scratch; they hold the last
 * simulation's values afterwards) of mzs_tower_args; `x`, `y`, `action` are ignored.  `pair_scratch` != NULL: two
 * workgroups per root (batch <= 128), as for mzs_resnet_tower -- check the status words afterwards and repeat the
 * search without it if any is set.  `discount`: the constant muax's recurrent_fn returns (muax/model.py:274).
 * The handle's embed_dim must be 2304 (6 x 6 x 64) and its tree must use cached decisions (the default whenever
 * batch * (num_simulations + 1)^2 words fit 1 GiB).  Errors: mzs_last_error(h). */
int mzs_resnet_search(mzs_handle *h, const mzs_tower_args *a, float discount, int32_t sim_begin, int32_t sim_end,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif
