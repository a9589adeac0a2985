// mz_api.hip -- host side of the C-ABI declared in include/mzsearch.h.
// Pure HIP runtime: no torch types, no CPU compute fallback.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mzsearch.h"
#include "mz_host.h"
#include "mz_fused_launch.h"
#include "mz_step.cuh"
#include "mz_step_jump.cuh"
#include "mz_mlp_generic.cuh"
#include "mz_train.cuh"
#include "mz_dirichlet.cuh"

namespace mzh {
thread_local std::string g_create_error;
}

namespace {

using mzh::g_create_error;

// fused-kernel instances built on demand and registered at run time (mzs_register_fused_dispatch; muax_amd/_jit.py)
std::mutex g_jit_mutex;
std::vector<mz::FusedDispatch> g_jit_dispatch;
std::vector<mz::FusedDispatch> g_jit_dispatch_muzero;  // MuZero-policy-only instances: tried before the all-modes ones
// training-step instances built on demand (mz_train_jit.hip): launcher of one (A, E, F = 2 support + 1) each
using JitTrainLaunch = int (*)(const void* train_params, void* stream, char* err, int errlen);
struct JitTrain {
  int A, E, F;
  JitTrainLaunch launch;
};
std::vector<JitTrain> g_jit_train;

// ---- host-side JAX threefry (key bookkeeping only: 3 blocks per simulation) ----
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
void h_threefry(const uint32_t key[2], uint32_t x0, uint32_t x1, uint32_t out[2]) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  const uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int i = 0; i < 4; ++i) {
      x0 += x1;
      x1 = rotl32(x1, R[g & 1][i]) ^ x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0;
  out[1] = x1;
}
uint32_t h_bits(const uint32_t key[2], uint64_t size, uint64_t i) {
  uint64_t half = (size + 1) / 2;
  uint64_t blk = i < half ? i : i - half;
  uint64_t c1 = half + blk;
  uint32_t out[2];
  h_threefry(key, (uint32_t)blk, c1 < size ? (uint32_t)c1 : 0u, out);
  return i < half ? out[0] : out[1];
}
void h_split(const uint32_t key[2], uint64_t n, uint64_t row, uint32_t out[2]) {
  out[0] = h_bits(key, 2 * n, 2 * row);
  out[1] = h_bits(key, 2 * n, 2 * row + 1);
}

}  // namespace

struct mzs_handle {
  mzs_config cfg;
  std::string err;
  bool have_weights = false;
  mzs_mlp_weights w;
  mz::StepState step;  // device buffers of the step-wise path (lazily allocated)
  uint32_t k_sample[2] = {0, 0};
  std::vector<uint32_t> sim_keys;  // [num_simulations][2], sized at create: simulate_key of every simulation
  uint64_t* prof = nullptr;        // MZ_PROFILE builds only
  int32_t* fused_table = nullptr;  // gumbel policy, fused path: seq_halving table on the device
  float* fused_emb = nullptr;      // fused path, embed_dim > 16: [B][S+1][E] embeddings in HBM
  int32_t* fused_path = nullptr;   // fused path, instances with the root paths in HBM: [B][S+1][fused_path_words]
  int fused_path_words = 0;
  int cu_count = 0;
  // mzs_act_mlp_host: pinned staging (in: obs | noise | invalid, out: action | weights | value) and their device twins
  void* host_in = nullptr; void* host_out = nullptr; void* dev_noise = nullptr;  // dev_noise: [B, A] drawn root noise
  size_t host_in_bytes = 0;
  mz::JumpArgs jump = {nullptr, nullptr, nullptr, nullptr};  // step-wise path with cached decisions
  void* jump_slab = nullptr;
  bool use_jump = false;
  int jump_roots = 0;              // roots the cached-decision slab holds: the batch (use_jump), or -- generic route of trees whose
                                   // B N^2 path words exceed the budget -- the chunk of roots act() searches at a time
  bool allow_generic = false;      // mzs_mlp_allow_generic: shapes without a fused instance take the generic one-launch search
  float* gen_scratch = nullptr;    // generic route: prior logits [B, A] | embeddings [B, E] | actions [B]
};

namespace {

int fail(mzs_handle* h, int code, const char* fmt, const char* a = "") {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, a);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}
#define MZS_HIP(h, call)                                                   \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) return fail(h, MZS_E_RUNTIME, #call ": %s", hipGetErrorString(e_)); \
  } while (0)

// mctx seq_halving.get_table_of_considered_visits (host integers, uploaded once per handle)
void considered_visits(int m, int S, int32_t* seq) {
  if (m <= 1) {
    for (int i = 0; i < S; ++i) seq[i] = i;
    return;
  }
  int log2max = 0;
  while ((1 << log2max) < m) ++log2max;
  std::vector<int32_t> visits(m, 0);
  int n = 0, nc = m;
  while (n < S) {
    int extra = S / (log2max * nc);
    if (extra < 1) extra = 1;
    for (int e = 0; e < extra; ++e) {
      for (int i = 0; i < nc && n < S; ++i) seq[n++] = visits[i];
      for (int i = 0; i < nc; ++i) visits[i] += 1;
    }
    nc = nc / 2 > 2 ? nc / 2 : 2;
  }
}

// (clang -O3 turns the four scalar threefry blocks of a simulation into ~32 ns: 1.6 us per 50-simulation act, measured;
// a hand-vectorised split3 was no faster -- round 6)
// mctx muzero_policy / search key walk: (k_sample, k_dirichlet, k_search) = split(key, 3);
// per simulation (rng, simulate_key, expand_key) = split(rng, 3).
void derive_keys(mzs_handle* h, const uint32_t key[2]) {
  uint32_t rk[2];
  h_split(key, 3, 0, h->k_sample);
  h_split(key, 3, 2, rk);
  const int S = h->cfg.num_simulations;
  for (int s = 0; s < S; ++s) {  // every simulation: the step-wise path takes up to 65534 of them
    uint32_t nk[2];
    h_split(rk, 3, 1, &h->sim_keys[2 * (size_t)s]);
    h_split(rk, 3, 0, nk);
    rk[0] = nk[0];
    rk[1] = nk[1];
  }
}

}  // namespace

namespace mzh {
int fail_handle(mzs_handle* h, int code, const char* msg) { return fail(h, code, "%s", msg); }
int step_view(mzs_handle* h, mz::StepArgs* sa, mz::JumpArgs* ja, int* policy, const char* who, int* device) {
  if (!h) return MZS_E_INVALID;
  if (!h->step.rooted) return fail(h, MZS_E_INVALID, "%s: call mzs_root first", who);
  if (!h->use_jump) return fail(h, MZS_E_UNSUPPORTED, "%s: this handle's tree has no cached decisions (too large, or MZS_STEP_WALK=1)", who);
  *sa = h->step.args(h->cfg);
  *ja = h->jump;
  *policy = h->cfg.policy;
  if (device) *device = h->cfg.device;
  return MZS_OK;
}
}  // namespace mzh

static void mlp_offsets(int obs_dim, int E, int A, int F, int off[19]) {
  const int H = mz::kHidden, X = E + A;
  const int sizes[18] = {obs_dim * E, E, E * H, H, H * F, F, E * H, H, H * A, A, X * H, H, H * F, F, X * H, H, H * E, E};
  off[0] = 0;
  for (int i = 0; i < 18; ++i) off[i + 1] = off[i] + sizes[i];
}

template <class C>
static int launch_train(const mz::TrainParams& p, hipStream_t stream) {
  const size_t lds = sizeof(float) * ((size_t)C::WEIGHT_WORDS + (size_t)p.L * C::CK_WORDS_PER_STEP);
  if (lds > 160 * 1024) return fail(nullptr, MZS_E_UNSUPPORTED, "mzs_mlp_loss_grad: unroll_steps too large for the LDS");
  auto kern = mz::mz_train_kernel<C>;
  static mzh::LdsGrant granted;  // (per device and instance: the attribute call is not free, update() runs every step)
  int dev = 0;
  MZS_HIP(nullptr, hipGetDevice(&dev));
  if (!granted.covers(dev, lds)) {
    MZS_HIP(nullptr, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    granted.note(dev, lds);
  }
  hipLaunchKernelGGL(kern, dim3(p.waves / 4), dim3(256), lds, stream, p);
  MZS_HIP(nullptr, hipGetLastError());
  hipLaunchKernelGGL(mz::mz_train_reduce_kernel, dim3((p.off[18] + 31) / 32), dim3(256), 0, stream, p);
  MZS_HIP(nullptr, hipGetLastError());
  return MZS_OK;
}

extern "C" {

int mzs_abi_version(void) { return MZS_ABI_VERSION; }
int mzs_fused_jit_abi(void) { return MZS_ABI_VERSION * 1000 + (int)(sizeof(mz::FusedParams) % 1000); }
int mzs_train_jit_abi(void) { return MZS_ABI_VERSION * 1000 + (int)(sizeof(mz::TrainParams) % 1000); }

int mzs_register_train_dispatch(void* launch, int32_t num_actions, int32_t embed_dim, int32_t full_support, int32_t jit_abi) {
  if (!launch) return fail(nullptr, MZS_E_INVALID, "mzs_register_train_dispatch: null");
  if (jit_abi != mzs_train_jit_abi())
    return fail(nullptr, MZS_E_INVALID, "mzs_register_train_dispatch: the side library was built from other sources (ABI)");
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  for (const JitTrain& t : g_jit_train)
    if (t.A == num_actions && t.E == embed_dim && t.F == full_support) return MZS_OK;
  g_jit_train.push_back({num_actions, embed_dim, full_support, reinterpret_cast<JitTrainLaunch>(launch)});
  return MZS_OK;
}

const char* mzs_last_error(const mzs_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mzs_create(const mzs_config* cfg, mzs_handle** out) {
  if (!cfg || !out) return fail(nullptr, MZS_E_INVALID, "mzs_create: null argument");
  if (cfg->struct_size != (int32_t)sizeof(mzs_config))
    return fail(nullptr, MZS_E_INVALID, "mzs_create: mzs_config size mismatch (ABI)");
  if (cfg->batch <= 0 || cfg->num_actions <= 0 || cfg->num_simulations <= 0 || cfg->embed_dim <= 0)
    return fail(nullptr, MZS_E_INVALID, "mzs_create: batch, num_actions, num_simulations, embed_dim must be positive");
  if (cfg->num_actions > 64) return fail(nullptr, MZS_E_UNSUPPORTED, "mzs_create: num_actions > 64");
  if (cfg->num_simulations >= 65535) return fail(nullptr, MZS_E_UNSUPPORTED, "mzs_create: num_simulations too large");
  if (cfg->policy != 0 && cfg->policy != 1) return fail(nullptr, MZS_E_INVALID, "mzs_create: policy must be 0 (muzero) or 1 (gumbel)");
  if (cfg->qtransform != 0 && !(cfg->qtransform == 1 && cfg->policy == 1))
    return fail(nullptr, MZS_E_UNSUPPORTED, "mzs_create: qtransform_completed_by_mix_value is built for the gumbel policy only");
  if (cfg->policy == 1 && cfg->max_num_considered_actions < 0)
    return fail(nullptr, MZS_E_INVALID, "mzs_create: max_num_considered_actions");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MZS_E_NODEVICE, "mzs_create: no HIP device (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MZS_E_INVALID, "mzs_create: bad device ordinal");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
    return fail(nullptr, MZS_E_RUNTIME, "mzs_create: hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, MZS_E_NODEVICE, "mzs_create: device is %s, kernels are built for gfx950 only", prop.gcnArchName);
  mzs_handle* h = new mzs_handle();
  h->cfg = *cfg;
  h->cu_count = prop.multiProcessorCount;
  if (h->cfg.global_batch <= 0) h->cfg.global_batch = cfg->batch;
  if (h->cfg.root_offset < 0 || h->cfg.root_offset + cfg->batch > h->cfg.global_batch) {
    delete h;
    return fail(nullptr, MZS_E_INVALID, "mzs_create: root_offset + batch exceeds global_batch");
  }
  memset(&h->w, 0, sizeof h->w);
  h->sim_keys.assign(2 * (size_t)cfg->num_simulations, 0u);
  *out = h;
  return MZS_OK;
}

int mzs_mlp_allow_generic(mzs_handle* h, int32_t allow) {
  if (!h) return MZS_E_INVALID;
  h->allow_generic = allow != 0;
  return MZS_OK;
}

int mzs_destroy(mzs_handle* h) {
  if (!h) return MZS_OK;
  hipSetDevice(h->cfg.device);
  h->step.release();
  if (h->fused_table) hipFree(h->fused_table);
  if (h->fused_emb) hipFree(h->fused_emb);
  if (h->fused_path) hipFree(h->fused_path);
  if (h->jump_slab) hipFree(h->jump_slab);
  if (h->host_in) hipHostFree(h->host_in);
  if (h->host_out) hipHostFree(h->host_out);
  if (h->dev_noise) hipFree(h->dev_noise);
  if (h->gen_scratch) hipFree(h->gen_scratch);
  delete h;
  return MZS_OK;
}

int mzs_mlp_set_weights(mzs_handle* h, const mzs_mlp_weights* w) {
  if (!h) return MZS_E_INVALID;
  if (!w || w->struct_size != (int32_t)sizeof(mzs_mlp_weights))
    return fail(h, MZS_E_INVALID, "mzs_mlp_set_weights: null or size mismatch (ABI)");
  const float* const* ptrs = &w->repr_w;
  for (int i = 0; i < 18; ++i)
    if (!ptrs[i]) return fail(h, MZS_E_INVALID, "mzs_mlp_set_weights: null weight pointer");
  if (w->obs_dim <= 0 || w->support_size <= 0) return fail(h, MZS_E_INVALID, "mzs_mlp_set_weights: obs_dim/support_size");
  h->w = *w;
  h->have_weights = true;
  return MZS_OK;
}

static int ensure_step_state(mzs_handle* h);
static int step_block(int batch);
static int step_grid(int batch);
// the generic route's ONE search launch over `n` roots (round 6: MuZero-policy instances specialised on the 16-lane slots
// the action count fills, with the pUCT table in LDS while it fits the workgroup's 64 KB)
static void launch_mlp_search(const mzs_config& c, const mz::StepArgs& sa, const mz::JumpArgs& ja, const mz::MlpGen& g, int n,
                              size_t lds_search, hipStream_t stream) {
  const size_t lds_tbl = lds_search + sizeof(float) * 2 * ((size_t)sa.S + 2);
  const bool tbl = sa.S + 2 <= 1030 && lds_tbl <= 64 * 1024;  // (Markstein's sequence is checked for every divisor up to 1030)
  // the 128-register build (four wavefronts per SIMD, mz_mlp_generic.cuh) where it puts MORE roots on the chip: more roots
  // than two wavefronts per SIMD hold, and workgroups small enough that sixteen share a CU's LDS
  const size_t lds = (c.policy != 1 && tbl && sa.A <= 32) ? lds_tbl : lds_search;
  const bool occ4 = n > 2 * 4 * 256 && 16 * lds <= 160 * 1024;
#define MZ_GEN_LAUNCH(...)                                                                                            \
  do {                                                                                                                \
    if (occ4) hipLaunchKernelGGL((mz::mz_mlp_search_kernel_occ4<__VA_ARGS__>), dim3(n), dim3(64), lds, stream, sa, ja, g, 0, sa.S); \
    else hipLaunchKernelGGL((mz::mz_mlp_search_kernel<__VA_ARGS__>), dim3(n), dim3(64), lds, stream, sa, ja, g, 0, sa.S);           \
  } while (0)
  if (c.policy == 1) MZ_GEN_LAUNCH(true);
  else if (tbl && sa.A <= 16) MZ_GEN_LAUNCH(false, 1, true);
  else if (tbl && sa.A <= 32) MZ_GEN_LAUNCH(false, 2, true);
  else MZ_GEN_LAUNCH(false);
#undef MZ_GEN_LAUNCH
}
// rows [rb, rb + n) of the step-wise tree as a batch of their own: every per-root array starts at row rb, the PRNG streams
// stay those of the global root index (root_offset + rb)
static mz::StepArgs slice_rows(mz::StepArgs s, size_t rb, int n) {
  const size_t N = (size_t)s.N, A = (size_t)s.A, E = (size_t)s.E;
  s.B = n;
  s.root_offset += rb;
  s.node_visits += rb * N; s.raw_values += rb * N; s.node_values += rb * N; s.parents += rb * N;
  s.action_from_parent += rb * N; s.path += rb * N;
  s.children_index += rb * N * A; s.children_prior_logits += rb * N * A; s.children_prior_probs += rb * N * A;
  s.children_values += rb * N * A; s.children_visits += rb * N * A; s.children_rewards += rb * N * A;
  s.children_discounts += rb * N * A; s.embeddings += rb * N * E;
  s.root_invalid += rb * A; s.root_gumbel += rb * A;
  s.sel_parent += rb; s.sel_action += rb; s.sel_depth += rb; s.depth_sum += rb; s.xfer_node += rb;
  return s;
}
// The generic route for a tree whose B N^2 cached path words exceed the slab budget (4096 roots x 1000 simulations would
// be 16 GB): the handle's slab holds `jump_roots` roots and the batch is searched in chunks of that many -- root /
// select(0) / ONE search launch / finish per chunk on the caller's stream, the slab reused chunk after chunk (stream order),
// the tree arrays those of the whole batch (an export copies them as ever).  Same kernels, same per-root PRNG streams
// (root_offset + row), hence the same bits as the undivided launch.
static int act_mlp_generic_chunks(mzs_handle* h, const mzs_act_args* a, const mz::MlpGen& g, float* pl, float* emb,
                                  int32_t* act0, size_t lds_search, hipStream_t stream) {
  const mzs_config& c = h->cfg;
  const size_t A = (size_t)c.num_actions, E = (size_t)c.embed_dim;
  uint32_t gk[2] = {0, 0};
  if (c.policy == 1) {
    h_split(a->key, 2, 1, gk);  // mctx gumbel_muzero_policy: rng_key, gumbel_rng = split(rng_key)
  } else {
    derive_keys(h, a->key);
    if (c.tiebreak)
      MZS_HIP(h, hipMemcpyAsync(h->step.sim_keys, h->sim_keys.data(), sizeof(uint32_t) * 2 * (size_t)c.num_simulations,
                                hipMemcpyHostToDevice, stream));
  }
  const mz::StepArgs whole = h->step.args(c);
  for (size_t rb = 0; rb < (size_t)c.batch; rb += (size_t)h->jump_roots) {
    const int n = (int)std::min((size_t)h->jump_roots, (size_t)c.batch - rb);
    const mz::StepArgs sa = slice_rows(whole, rb, n);
    const dim3 grid(step_grid(n)), blk(step_block(n));
    const uint8_t* inv = a->invalid_actions ? a->invalid_actions + rb * A : nullptr;
    if (c.policy == 1) {
      hipLaunchKernelGGL(mz::step_root_kernel, grid, blk, 0, stream, sa, pl + rb * A, a->root_value + rb, emb + rb * E, inv,
                         static_cast<const float*>(nullptr), 0.0f, 1, a->gumbel ? a->gumbel + rb * A : nullptr, gk[0], gk[1]);
      hipLaunchKernelGGL(mz::jump_root_kernel<true>, grid, blk, 0, stream, sa, h->jump);
    } else {
      hipLaunchKernelGGL(mz::step_root_kernel, grid, blk, 0, stream, sa, pl + rb * A, a->root_value + rb, emb + rb * E, inv,
                         a->dirichlet_noise ? a->dirichlet_noise + rb * A : nullptr, a->dirichlet_fraction, 0,
                         static_cast<const float*>(nullptr), 0u, 0u);
      hipLaunchKernelGGL(mz::jump_root_kernel<false>, grid, blk, 0, stream, sa, h->jump);
    }
    hipLaunchKernelGGL(mz::jump_select_kernel<false>, grid, blk, 0, stream, sa, h->jump, 0, act0 + rb, emb + rb * E);
    launch_mlp_search(c, sa, h->jump, g, n, lds_search, stream);
    if (c.policy == 1) {
      hipLaunchKernelGGL(mz::step_finish_gumbel_kernel, grid, blk, 0, stream, sa, a->action + rb, a->action_weights + rb * A,
                         a->search_value ? a->search_value + rb : nullptr, a->depth_sum ? a->depth_sum + rb : nullptr);
    } else {
      hipLaunchKernelGGL(mz::step_finish_kernel, grid, blk, 0, stream, sa, a->temperature, a->gumbel ? a->gumbel + rb * A : nullptr,
                         h->k_sample[0], h->k_sample[1], a->action + rb, a->action_weights + rb * A,
                         a->search_value ? a->search_value + rb : nullptr, a->depth_sum ? a->depth_sum + rb : nullptr);
    }
    MZS_HIP(h, hipGetLastError());
  }
  h->step.rooted = true;
  if (a->tree) return mzs_tree_export(h, a->tree, stream);
  return MZS_OK;
}
// act() of the default MLP trio for shapes the fused kernel has no instance for (mz_mlp_generic.cuh): root inference,
// mzs_root, mzs_select(0), ONE launch for all simulations, mzs_finish -- five launches per act instead of two per
// simulation, the nets evaluated by the library to the project's arithmetic spec (== the oracle for any shape).
static int act_mlp_generic(mzs_handle* h, const mzs_act_args* a, void* stream_) {
  const mzs_config& c = h->cfg;
  const mzs_mlp_weights& w = h->w;
  const int A = c.num_actions, E = c.embed_dim, F = 2 * w.support_size + 1, S = c.num_simulations;
  if (F < 17 || F > 64 || A > 64)
    return fail(h, MZS_E_UNSUPPORTED, "mzs_act_mlp (generic route): support_size must be 8..31 and num_actions <= 64");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));  // (reached before mzs_act_mlp's own hipSetDevice when num_simulations > kMaxSims)
  if (int rc = ensure_step_state(h)) return rc;
  if (h->jump_roots < 1 || (!h->use_jump && E >= mz::kWideEmb))
    return fail(h, MZS_E_UNSUPPORTED, "mzs_act_mlp (generic route): no cached-decision slab for this tree (more than 1023 "
                                      "simulations, MZS_STEP_WALK=1, or out of device memory); use the step-wise path");
  const size_t B = (size_t)c.batch;
  if (!h->gen_scratch) MZS_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->gen_scratch), (B * A + B * E + B) * sizeof(float)));
  float* pl = h->gen_scratch;
  float* emb = pl + B * A;
  int32_t* act0 = reinterpret_cast<int32_t*>(emb + B * E);
  mz::MlpGen g;
  g.repr_w = w.repr_w; g.repr_b = w.repr_b;
  g.pv_w1 = w.pv_w1; g.pv_b1 = w.pv_b1; g.pv_w2 = w.pv_w2; g.pv_b2 = w.pv_b2;
  g.pp_w1 = w.pp_w1; g.pp_b1 = w.pp_b1; g.pp_w2 = w.pp_w2; g.pp_b2 = w.pp_b2;
  g.dr_w1 = w.dr_w1; g.dr_b1 = w.dr_b1; g.dr_w2 = w.dr_w2; g.dr_b2 = w.dr_b2;
  g.dn_w1 = w.dn_w1; g.dn_b1 = w.dn_b1; g.dn_w2 = w.dn_w2; g.dn_b2 = w.dn_b2;
  g.obs_dim = w.obs_dim; g.E = E; g.A = A; g.F = F; g.support = w.support_size; g.pred_on_parent = w.recurrent_pred_on;
  g.discount = w.discount;
  const int ew = E > w.obs_dim ? E : w.obs_dim;
  const size_t lds_root = sizeof(float) * (size_t)mz::gen_scratch_words(ew, A);
  const size_t lds_search = sizeof(int32_t) * 15 * ((size_t)S + 2) + sizeof(float) * (size_t)mz::gen_scratch_words(E, A);
  if (lds_root > 64 * 1024 || lds_search > 64 * 1024)
    return fail(h, MZS_E_UNSUPPORTED, "mzs_act_mlp (generic route): num_simulations / embedding too large for the LDS of a workgroup");
  hipLaunchKernelGGL(mz::mz_mlp_root_kernel, dim3(c.batch), dim3(64), lds_root, stream, g, c.batch, a->obs, pl, a->root_value, emb);
  MZS_HIP(h, hipGetLastError());
  if (!h->use_jump) return act_mlp_generic_chunks(h, a, g, pl, emb, act0, lds_search, stream);
  int rc = c.policy == 1 ? mzs_root_gumbel(h, pl, a->root_value, emb, a->invalid_actions, a->gumbel, a->key, stream_)
                         : mzs_root(h, pl, a->root_value, emb, a->invalid_actions, a->dirichlet_noise, a->dirichlet_fraction,
                                    a->key, stream_);
  if (rc) return rc;
  if ((rc = mzs_select(h, 0, act0, emb, stream_))) return rc;  // simulate() of simulation 0 (emb: consumed by mzs_root, reused)
  mz::StepArgs sa = h->step.args(c);
  launch_mlp_search(c, sa, h->jump, g, c.batch, lds_search, stream);
  MZS_HIP(h, hipGetLastError());
  if ((rc = mzs_finish(h, a->temperature, c.policy == 1 ? nullptr : a->gumbel, a->action, a->action_weights, a->search_value,
                       a->depth_sum, stream_)))
    return rc;
  if (a->tree) return mzs_tree_export(h, a->tree, stream_);
  return MZS_OK;
}

static int register_dispatch(std::vector<mz::FusedDispatch>& list, void* dispatch, int32_t jit_abi) {
  if (!dispatch) return fail(nullptr, MZS_E_INVALID, "mzs_register_fused_dispatch: null");
  if (jit_abi != mzs_fused_jit_abi()) return fail(nullptr, MZS_E_INVALID, "mzs_register_fused_dispatch: the side library was built from other sources (ABI)");
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  for (auto f : list)
    if (reinterpret_cast<void*>(f) == dispatch) return MZS_OK;
  list.push_back(reinterpret_cast<mz::FusedDispatch>(dispatch));
  return MZS_OK;
}
int mzs_register_fused_dispatch(void* dispatch, int32_t jit_abi) { return register_dispatch(g_jit_dispatch, dispatch, jit_abi); }
int mzs_register_fused_dispatch_muzero(void* dispatch, int32_t jit_abi) {
  return register_dispatch(g_jit_dispatch_muzero, dispatch, jit_abi);
}

int mzs_act_mlp(mzs_handle* h, const mzs_act_args* a, void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!a || a->struct_size != (int32_t)sizeof(mzs_act_args))
    return fail(h, MZS_E_INVALID, "mzs_act_mlp: null or size mismatch (ABI)");
  if (!h->have_weights) return fail(h, MZS_E_INVALID, "mzs_act_mlp: call mzs_mlp_set_weights first");
  if (!a->obs || !a->action || !a->action_weights || !a->root_value)
    return fail(h, MZS_E_INVALID, "mzs_act_mlp: obs/action/action_weights/root_value must be set");
  const mzs_config& c = h->cfg;
  if (c.policy == 0 && !a->dirichlet_noise && a->dirichlet_fraction != 0.0f)
    return fail(h, MZS_E_INVALID, "mzs_act_mlp: dirichlet_fraction != 0 needs dirichlet_noise");
  if (c.num_simulations > mz::kMaxSims) {  // (the fused kernel's argument block holds 256 simulation keys)
    if (h->allow_generic) return act_mlp_generic(h, a, stream_);
    return fail(h, MZS_E_UNSUPPORTED, "mzs_act_mlp: no fused kernel instance for num_simulations > 256; use the generic route "
                                      "(mzs_mlp_allow_generic) or the step-wise path");
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));

  mz::FusedParams p;
  memset(&p, 0, sizeof p);
  p.obs = a->obs; p.dirichlet_noise = a->dirichlet_noise; p.invalid = a->invalid_actions; p.gumbel = a->gumbel;
  const mzs_mlp_weights& w = h->w;
  p.repr_w = w.repr_w; p.repr_b = w.repr_b;
  p.pv_w1 = w.pv_w1; p.pv_b1 = w.pv_b1; p.pv_w2 = w.pv_w2; p.pv_b2 = w.pv_b2;
  p.pp_w1 = w.pp_w1; p.pp_b1 = w.pp_b1; p.pp_w2 = w.pp_w2; p.pp_b2 = w.pp_b2;
  p.dr_w1 = w.dr_w1; p.dr_b1 = w.dr_b1; p.dr_w2 = w.dr_w2; p.dr_b2 = w.dr_b2;
  p.dn_w1 = w.dn_w1; p.dn_b1 = w.dn_b1; p.dn_w2 = w.dn_w2; p.dn_b2 = w.dn_b2;
  p.action = a->action; p.action_weights = a->action_weights; p.root_value = a->root_value;
  p.search_value = a->search_value; p.depth_sum = a->depth_sum;
  if (a->tree) {
    const mzs_tree_view& t = *a->tree;
    const void* const* tp = reinterpret_cast<const void* const*>(&t);
    for (int i = 0; i < 12; ++i)
      if (!tp[i]) return fail(h, MZS_E_INVALID, "mzs_act_mlp: tree view has a null array");
    p.t_node_visits = t.node_visits; p.t_raw_values = t.raw_values; p.t_node_values = t.node_values;
    p.t_parents = t.parents; p.t_action_from_parent = t.action_from_parent;
    p.t_children_index = t.children_index; p.t_children_prior_logits = t.children_prior_logits;
    p.t_children_values = t.children_values; p.t_children_visits = t.children_visits;
    p.t_children_rewards = t.children_rewards; p.t_children_discounts = t.children_discounts;
    p.t_embeddings = t.embeddings;
    p.export_tree = 1;
  }
  p.B = c.batch; p.obs_dim = w.obs_dim; p.S = c.num_simulations; p.max_depth = c.max_depth;
  p.support = w.support_size; p.pred_on_parent = w.recurrent_pred_on;
  p.pb_c_init = c.pb_c_init; p.pb_c_base = c.pb_c_base;
  p.dirichlet_fraction = a->dirichlet_fraction; p.discount = w.discount; p.temperature = a->temperature;
  p.global_batch = (uint64_t)c.global_batch; p.root_offset = (uint64_t)c.root_offset;
  p.prof = h->prof;
  // (embeddings wider than 16 -- and those of every FusedCfg::LONG instance: long searches, wide action sets -- live in
  // HBM: the caller's export buffer when a tree is exported, else this scratch, allocated when a dispatcher asks for it)
  p.emb_scratch = p.export_tree ? nullptr : h->fused_emb;
  if (c.policy == 1) {
    // gumbel policy: seq_halving table on the device (once), root Gumbel key = split(key)[1]
    const int rows = c.max_num_considered_actions + 1;
    if (!h->fused_table) {
      std::vector<int32_t> table((size_t)rows * c.num_simulations);
      for (int m = 0; m < rows; ++m) considered_visits(m, c.num_simulations, table.data() + (size_t)m * c.num_simulations);
      MZS_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->fused_table), table.size() * sizeof(int32_t)));
      MZS_HIP(h, hipMemcpy(h->fused_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    p.visit_table = h->fused_table;
    p.max_considered = c.max_num_considered_actions;
    p.gumbel_scale = c.gumbel_scale;
    h_split(a->key, 2, 1, p.k_gumbel);
  }
  derive_keys(h, a->key);
  p.k_sample[0] = h->k_sample[0]; p.k_sample[1] = h->k_sample[1];
  memcpy(p.sim_keys, h->sim_keys.data(), sizeof(uint32_t) * 2 * (size_t)c.num_simulations);  // <= kMaxSims (checked above)

  const int A = c.num_actions, E = c.embed_dim, F = 2 * w.support_size + 1, N = c.num_simulations + 1;
  p.F = F;
  const int mode = c.policy == 1 ? (c.qtransform == 1 ? 3 : 2) : (c.tiebreak ? 1 : 0);
  std::vector<mz::FusedDispatch> groups = {mz::fused_dispatch_g0, mz::fused_dispatch_g1, mz::fused_dispatch_g2,
                                           mz::fused_dispatch_g3, mz::fused_dispatch_g4};
  {
    std::lock_guard<std::mutex> lock(g_jit_mutex);  // instances built on demand (mzs_register_fused_dispatch[_muzero])
    if (mode < 2) groups.insert(groups.end(), g_jit_dispatch_muzero.begin(), g_jit_dispatch_muzero.end());
    groups.insert(groups.end(), g_jit_dispatch.begin(), g_jit_dispatch.end());
  }
  // (tools/bench_generic.py: MZS_FORCE_GENERIC=1 sends a shape that HAS an instance through the generic route, for A/B timing)
  if (h->allow_generic && getenv("MZS_FORCE_GENERIC") != nullptr) return act_mlp_generic(h, a, stream_);
  // more 16-root workgroups than CUs: prefer a compact-record instance (two workgroups per CU), if the shape has one
  for (int compact = (c.batch > 16 * h->cu_count) ? 1 : 0; compact >= 0; --compact) {
    // (handed over whenever it exists: an instance for 128..255 simulations keeps its root paths there at any batch size)
    p.path_scratch = h->fused_path;
    p.path_words = h->fused_path_words;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      std::string err;
      int rc = groups[gi](mode, c.device, p, stream, A, E, F, N, compact != 0, &err);
      for (int tries = 0; tries < 2 && (rc == mz::kNeedEmbScratch || rc >= mz::kNeedPathScratch); ++tries) {
        if (rc == mz::kNeedEmbScratch) {  // first launch (without a tree export) of an instance with its embeddings in HBM
          MZS_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->fused_emb),
                               (size_t)c.batch * (c.num_simulations + 1) * c.embed_dim * sizeof(float)));
          p.emb_scratch = h->fused_emb;
        } else {  // first launch of an instance with its root paths in HBM: their array
          const int words = rc - mz::kNeedPathScratch;
          if (h->fused_path) MZS_HIP(h, hipFree(h->fused_path));
          h->fused_path = nullptr;
          h->fused_path_words = 0;
          MZS_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->fused_path), (size_t)c.batch * N * words * sizeof(int32_t)));
          h->fused_path_words = words;
          p.path_scratch = h->fused_path;
          p.path_words = words;
        }
        rc = groups[gi](mode, c.device, p, stream, A, E, F, N, compact != 0, &err);
      }
      if (rc == mz::kNoFusedInstance) continue;
      if (rc != MZS_OK) return fail(h, rc, "mzs_act_mlp: %s", err.c_str());
      return MZS_OK;
    }
  }
  if (h->allow_generic) return act_mlp_generic(h, a, stream_);
  return fail(h, MZS_E_UNSUPPORTED,
              "mzs_act_mlp: no fused kernel instance for this (A, E, F, S) (muax_amd/csrc/mz_instances.def); use the step-wise path");
}

int mzs_act_mlp_host(mzs_handle* h, const mzs_act_host_args* a, void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!a || a->struct_size != (int32_t)sizeof(mzs_act_host_args))
    return fail(h, MZS_E_INVALID, "mzs_act_mlp_host: null or size mismatch (ABI)");
  if (!h->have_weights) return fail(h, MZS_E_INVALID, "mzs_act_mlp_host: call mzs_mlp_set_weights first");
  if (!a->obs || !a->action || !a->action_weights || !a->root_value)
    return fail(h, MZS_E_INVALID, "mzs_act_mlp_host: obs/action/action_weights/root_value must be set");
  const mzs_config& c = h->cfg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  const size_t B = (size_t)c.batch, A = (size_t)c.num_actions, OD = (size_t)h->w.obs_dim;
  // staging layout (4-byte words): obs [B, OD] | noise [B, A] | invalid [B, A] bytes
  const size_t obs_b = B * OD * 4, noise_b = B * A * 4, inv_b = (B * A + 3) / 4 * 4, in_b = obs_b + noise_b + inv_b;
  const size_t out_b = B * (2 + A) * 4;
  if (h->host_in_bytes < in_b) {
    if (h->host_in) { hipHostFree(h->host_in); h->host_in = nullptr; }
    MZS_HIP(h, hipHostMalloc(&h->host_in, in_b, hipHostMallocDefault));
    h->host_in_bytes = in_b;
  }
  if (!h->host_out) MZS_HIP(h, hipHostMalloc(&h->host_out, out_b, hipHostMallocDefault));
  if (!h->dev_noise) MZS_HIP(h, hipMalloc(&h->dev_noise, noise_b));
  // The kernels read the host's inputs and write its outputs THROUGH THE PINNED STAGING BUFFERS themselves (hipHostMalloc
  // memory is mapped into the device's address space, coherent): an act moves 16..32 bytes per root each way, read once
  // at the kernel's start and written once at its end, and a copy command costs more in launch and engine latency than
  // those bytes cost over the host link.  Only the drawn root noise lives in device memory (its producer is a kernel).
  char* hin = static_cast<char*>(h->host_in);
  char* hin_dev = nullptr;
  float* hout_dev = nullptr;
  MZS_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&hin_dev), h->host_in, 0));
  MZS_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&hout_dev), h->host_out, 0));
  const bool muzero = c.policy == 0;
  const bool given = muzero && a->dirichlet_noise != nullptr;
  const bool draw = muzero && !given && a->draw_dirichlet != 0 && a->dirichlet_fraction != 0.0f;
  float* d_noise = static_cast<float*>(h->dev_noise);
  if (draw) {  // first: it needs nothing from the host and runs while the host fills the staging buffer
    uint32_t kd[2];
    h_split(a->key, 3, 1, kd);  // mctx: rng_key, dirichlet_rng_key, search_rng_key = split(rng_key, 3)
    if (int rc = mzs_dirichlet(c.device, kd, a->dirichlet_alpha, c.batch, c.num_actions, c.global_batch, c.root_offset,
                               d_noise, stream_))
      return fail(h, rc, "mzs_act_mlp_host: %s", mzs_last_error(nullptr));
  }
  memcpy(hin, a->obs, obs_b);
  if (given) memcpy(hin + obs_b, a->dirichlet_noise, noise_b);
  if (a->invalid_actions) memcpy(hin + obs_b + noise_b, a->invalid_actions, B * A);
  float* dout = hout_dev;
  mzs_act_args args;
  memset(&args, 0, sizeof args);
  args.struct_size = (int32_t)sizeof args;
  args.obs = reinterpret_cast<const float*>(hin_dev);
  args.dirichlet_noise = draw ? d_noise : (given ? reinterpret_cast<const float*>(hin_dev + obs_b) : nullptr);
  args.invalid_actions = a->invalid_actions ? reinterpret_cast<const uint8_t*>(hin_dev + obs_b + noise_b) : nullptr;
  args.key[0] = a->key[0]; args.key[1] = a->key[1];
  args.dirichlet_fraction = (given || draw) ? a->dirichlet_fraction : 0.0f;
  args.temperature = a->temperature;
  args.action = reinterpret_cast<int32_t*>(dout);
  args.action_weights = dout + B;
  args.root_value = dout + B + B * A;
  if (int rc = mzs_act_mlp(h, &args, stream_)) return rc;
  MZS_HIP(h, hipStreamSynchronize(stream));
  const char* hout = static_cast<const char*>(h->host_out);
  memcpy(a->action, hout, B * 4);
  memcpy(a->action_weights, hout + B * 4, B * A * 4);
  memcpy(a->root_value, hout + B * 4 + B * A * 4, B * 4);
  return MZS_OK;
}

#ifdef MZ_PROFILE
// tools-only entry point (not part of the ABI): per-wave phase cycle counters [waves][8]
int mzs_debug_profile(mzs_handle* h, uint64_t* device_buffer) {
  if (!h) return MZS_E_INVALID;
  h->prof = device_buffer;
  return MZS_OK;
}
// ... and the tree-step phase counters of THIS translation unit's kernels (the generic one-launch search, the step-wise
// launches): read and clear (tools/profile_generic.py)
int mzs_debug_generic_jump_profile(uint64_t* host_out, int32_t words) {
  static unsigned long long zero[1024 * 8];
  if (words > 1024 * 8) words = 1024 * 8;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mz::g_jump_prof), sizeof(uint64_t) * (size_t)words) != hipSuccess) return MZS_E_RUNTIME;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mz::g_jump_prof), zero, sizeof(zero)) != hipSuccess) return MZS_E_RUNTIME;
  return MZS_OK;
}
#endif

// ---------------------------------------------------------------------------
// step-wise path
// ---------------------------------------------------------------------------

// Small batches run one wavefront (4 roots) per workgroup: the tree of a root is then always walked from the
// same XCD, all 8 L2s share the trees, and the dependent per-level loads hit L2 instead of HBM.
static int step_block(int batch) { return batch >= 4096 ? 256 : 64; }
static int step_grid(int batch) { const int per = step_block(batch) / 16; return (batch + per - 1) / per; }

static void emb_xfer(const mz::StepArgs& sa, float* rows, int dir, hipStream_t stream) {
  hipLaunchKernelGGL(mz::emb_xfer_kernel, dim3(sa.B, (sa.E + 1023) / 1024), dim3(256), 0, stream, sa, rows, dir);
}

static int ensure_step_state(mzs_handle* h) {
  const mzs_config& c = h->cfg;
  if (h->step.allocated) return MZS_OK;
  const int rows = c.policy == 1 ? c.max_num_considered_actions + 1 : 0;
  const int table_words = rows * c.num_simulations;
  hipError_t e = h->step.allocate(c.batch, c.num_simulations + 1, c.num_actions, c.embed_dim, table_words);
  if (e != hipSuccess) return fail(h, MZS_E_RUNTIME, "tree allocation: %s", hipGetErrorString(e));
  // cached-decision kernels (mz_step_jump.cuh): bounded tree, B N^2 path words within the slab budget (8 GiB of the
  // 288 GB: 4096 roots x 300 simulations are 1.5 GB; MZS_JUMP_BUDGET_MB overrides); MZS_STEP_WALK=1 keeps the
  // level-by-level kernels (A/B testing).  A tree beyond the budget gets a slab for a CHUNK of roots: the step-wise
  // entry points then walk level by level (they address the whole batch), the generic one-launch search of mzs_act_mlp
  // runs the batch chunk by chunk (roots never interact; muax/model.py:222-243 takes any num_simulations).
  {
    const size_t B = (size_t)c.batch, N = (size_t)c.num_simulations + 1;
    const char* walk = getenv("MZS_STEP_WALK");
    const char* mb = getenv("MZS_JUMP_BUDGET_MB");
    const size_t budget = mb && atoll(mb) > 0 ? (size_t)atoll(mb) << 20 : (size_t)8 << 30;
    const size_t per_root = (3 * N + N * N) * 4;
    if (N <= (size_t)mz::kJumpMaxNodes && !(walk && walk[0] == '1')) {
      const bool whole = B * per_root <= budget;
      size_t roots = whole ? B : budget / per_root;
      if (roots > B) roots = B;
      if (roots >= 1 && hipMalloc(&h->jump_slab, roots * per_root) == hipSuccess) {
        int32_t* w = static_cast<int32_t*>(h->jump_slab);
        h->jump.jump_pa = w; h->jump.jump_lv = w + roots * N; h->jump.node_depth = w + 2 * roots * N;
        h->jump.node_path = reinterpret_cast<uint32_t*>(w + 3 * roots * N);
        h->use_jump = whole;
        h->jump_roots = (int)roots;
      }
    }
  }
  if (table_words) {
    std::vector<int32_t> table((size_t)table_words);
    for (int m = 0; m < rows; ++m) considered_visits(m, c.num_simulations, table.data() + (size_t)m * c.num_simulations);
    MZS_HIP(h, hipMemcpy(h->step.visit_table, table.data(), sizeof(int32_t) * (size_t)table_words, hipMemcpyHostToDevice));
  }
  return MZS_OK;
}

int mzs_root(mzs_handle* h, const float* prior_logits, const float* value, const float* embedding,
             const uint8_t* invalid_actions, const float* dirichlet_noise, float dirichlet_fraction,
             const uint32_t key[2], void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!prior_logits || !value || !embedding) return fail(h, MZS_E_INVALID, "mzs_root: null input");
  if (!dirichlet_noise && dirichlet_fraction != 0.0f)
    return fail(h, MZS_E_INVALID, "mzs_root: dirichlet_fraction != 0 needs dirichlet_noise");
  const mzs_config& c = h->cfg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  if (c.policy != 0) return fail(h, MZS_E_INVALID, "mzs_root: this handle runs the gumbel policy; use mzs_root_gumbel");
  if (int rc = ensure_step_state(h)) return rc;
  uint32_t zero[2] = {0, 0};
  derive_keys(h, key ? key : zero);
  if (c.tiebreak) {
    // pageable source: the runtime stages it before the call returns, so the next act() may rewrite sim_keys
    MZS_HIP(h, hipMemcpyAsync(h->step.sim_keys, h->sim_keys.data(), sizeof(uint32_t) * 2 * (size_t)c.num_simulations,
                              hipMemcpyHostToDevice, stream));
  }
  mz::StepArgs sa = h->step.args(c);
  if (sa.wide) MZS_HIP(h, hipMemsetAsync(sa.embeddings, 0, sizeof(float) * (size_t)sa.B * sa.N * sa.E, stream));
  hipLaunchKernelGGL(mz::step_root_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, prior_logits,
                     value, embedding, invalid_actions, dirichlet_noise, dirichlet_fraction, 0,
                     static_cast<const float*>(nullptr), 0u, 0u);
  if (h->use_jump)
    hipLaunchKernelGGL(mz::jump_root_kernel<false>, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, h->jump);
  if (sa.wide) emb_xfer(sa, const_cast<float*>(embedding), 1, stream);
  MZS_HIP(h, hipGetLastError());
  h->step.rooted = true;
  return MZS_OK;
}

int mzs_root_gumbel(mzs_handle* h, const float* prior_logits, const float* value, const float* embedding,
                    const uint8_t* invalid_actions, const float* gumbel, const uint32_t key[2], void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!prior_logits || !value || !embedding) return fail(h, MZS_E_INVALID, "mzs_root_gumbel: null input");
  const mzs_config& c = h->cfg;
  if (c.policy != 1) return fail(h, MZS_E_INVALID, "mzs_root_gumbel: handle was created with policy 0 (muzero)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  if (int rc = ensure_step_state(h)) return rc;
  // mctx gumbel_muzero_policy: rng_key, gumbel_rng = jax.random.split(rng_key)
  uint32_t zero[2] = {0, 0}, gk[2];
  h_split(key ? key : zero, 2, 1, gk);
  mz::StepArgs sa = h->step.args(c);
  if (sa.wide) MZS_HIP(h, hipMemsetAsync(sa.embeddings, 0, sizeof(float) * (size_t)sa.B * sa.N * sa.E, stream));
  hipLaunchKernelGGL(mz::step_root_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, prior_logits,
                     value, embedding, invalid_actions, static_cast<const float*>(nullptr), 0.0f, 1, gumbel, gk[0],
                     gk[1]);
  if (h->use_jump)
    hipLaunchKernelGGL(mz::jump_root_kernel<true>, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, h->jump);
  if (sa.wide) emb_xfer(sa, const_cast<float*>(embedding), 1, stream);
  MZS_HIP(h, hipGetLastError());
  h->step.rooted = true;
  return MZS_OK;
}

int mzs_select(mzs_handle* h, int32_t sim, int32_t* action_out, float* parent_embedding_out, void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!h->step.rooted) return fail(h, MZS_E_INVALID, "mzs_select: call mzs_root first");
  if (sim < 0 || sim >= h->cfg.num_simulations) return fail(h, MZS_E_INVALID, "mzs_select: sim out of range");
  if (!action_out || !parent_embedding_out) return fail(h, MZS_E_INVALID, "mzs_select: null output");
  const mzs_config& c = h->cfg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  mz::StepArgs sa = h->step.args(c);
  bool gathered = false;
  if (h->use_jump && sa.wide) {  // one workgroup per root: selection + the gather of the wide embedding row
    hipLaunchKernelGGL(mz::jump_select_kernel<true>, dim3(c.batch), dim3(256), 0, stream, sa, h->jump, sim, action_out,
                       parent_embedding_out);
    gathered = true;
  } else if (h->use_jump)
    hipLaunchKernelGGL(mz::jump_select_kernel<false>, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa,
                       h->jump, sim, action_out, parent_embedding_out);
  else if (c.policy == 1)
    hipLaunchKernelGGL(mz::step_select_gumbel_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, sim,
                       action_out, parent_embedding_out);
  else
    hipLaunchKernelGGL(mz::step_select_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, sim,
                       action_out, parent_embedding_out);
  if (sa.wide && !gathered) emb_xfer(sa, parent_embedding_out, 0, stream);
  MZS_HIP(h, hipGetLastError());
  return MZS_OK;
}

static int expand_backup_impl(mzs_handle* h, int32_t sim, const float* reward, const float* discount,
                              const float* prior_logits, const float* value, const float* next_embedding,
                              int32_t* next_action_out, float* next_parent_embedding_out, void* stream_,
                              const char* who) {
  if (!h) return MZS_E_INVALID;
  if (!h->step.rooted) return fail(h, MZS_E_INVALID, "%s: call mzs_root first", who);
  if (sim < 0 || sim >= h->cfg.num_simulations) return fail(h, MZS_E_INVALID, "%s: sim out of range", who);
  if (!reward || !discount || !prior_logits || !value || !next_embedding)
    return fail(h, MZS_E_INVALID, "%s: null input", who);
  const mzs_config& c = h->cfg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  mz::StepArgs sa = h->step.args(c);
  const bool want_next = next_action_out != nullptr && sim + 1 < c.num_simulations;
  if (h->use_jump)
    // small batches: 16 levels in flight per root; large ones: one wavefront per root keeps the launch small
  {
    // few roots and long searches (deep paths): 64 levels in flight
    const dim3 blk(c.batch <= 256 && c.num_simulations >= 64 ? 1024 : (c.batch <= 1024 ? 256 : 64));
    const size_t lds = sizeof(int32_t) * 15 * ((size_t)c.num_simulations + 2);
    if (c.policy == 1)
      hipLaunchKernelGGL(mz::jump_expand_backup_kernel<true>, dim3(c.batch), blk, lds, stream, sa, h->jump, sim, reward,
                         discount, prior_logits, value, next_embedding, next_action_out, next_parent_embedding_out);
    else
      hipLaunchKernelGGL(mz::jump_expand_backup_kernel<false>, dim3(c.batch), blk, lds, stream, sa, h->jump, sim, reward,
                         discount, prior_logits, value, next_embedding, next_action_out, next_parent_embedding_out);
  }
  else
    hipLaunchKernelGGL(mz::step_expand_backup_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, sim,
                       reward, discount, prior_logits, value, next_embedding);
  if (sa.wide && !h->use_jump) emb_xfer(sa, const_cast<float*>(next_embedding), 1, stream);
  MZS_HIP(h, hipGetLastError());
  // the walking kernels (trees beyond the cached-decision budget, MZS_STEP_WALK=1) select in a launch of their own
  if (want_next && !h->use_jump) return mzs_select(h, sim + 1, next_action_out, next_parent_embedding_out, stream_);
  return MZS_OK;
}

int mzs_expand_backup(mzs_handle* h, int32_t sim, const float* reward, const float* discount,
                      const float* prior_logits, const float* value, const float* next_embedding,
                      void* stream_) {
  return expand_backup_impl(h, sim, reward, discount, prior_logits, value, next_embedding, nullptr, nullptr, stream_,
                            "mzs_expand_backup");
}

int mzs_expand_backup_select(mzs_handle* h, int32_t sim, const float* reward, const float* discount,
                             const float* prior_logits, const float* value, const float* next_embedding,
                             int32_t* next_action_out, float* next_parent_embedding_out, void* stream_) {
  if (h && (!next_action_out || !next_parent_embedding_out))
    return fail(h, MZS_E_INVALID, "mzs_expand_backup_select: null output");
  return expand_backup_impl(h, sim, reward, discount, prior_logits, value, next_embedding, next_action_out,
                            next_parent_embedding_out, stream_, "mzs_expand_backup_select");
}

int mzs_finish(mzs_handle* h, float temperature, const float* gumbel, int32_t* action_out,
               float* action_weights_out, float* search_value_out, int32_t* depth_sum_out, void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!h->step.rooted) return fail(h, MZS_E_INVALID, "mzs_finish: call mzs_root first");
  if (!action_out || !action_weights_out) return fail(h, MZS_E_INVALID, "mzs_finish: null output");
  const mzs_config& c = h->cfg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  mz::StepArgs sa = h->step.args(c);
  if (c.policy == 1)
    hipLaunchKernelGGL(mz::step_finish_gumbel_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa,
                       action_out, action_weights_out, search_value_out, depth_sum_out);
  else
    hipLaunchKernelGGL(mz::step_finish_kernel, dim3(step_grid(c.batch)), dim3(step_block(c.batch)), 0, stream, sa, temperature,
                       gumbel, h->k_sample[0], h->k_sample[1], action_out, action_weights_out, search_value_out,
                       depth_sum_out);
  MZS_HIP(h, hipGetLastError());
  return MZS_OK;
}

int mzs_tree_export(mzs_handle* h, const mzs_tree_view* out, void* stream_) {
  if (!h) return MZS_E_INVALID;
  if (!h->step.rooted) return fail(h, MZS_E_INVALID, "mzs_tree_export: no step-wise tree (call mzs_root first)");
  if (!out) return fail(h, MZS_E_INVALID, "mzs_tree_export: null view");
  const void* const* tp = reinterpret_cast<const void* const*>(out);
  for (int i = 0; i < 12; ++i)
    if (!tp[i]) return fail(h, MZS_E_INVALID, "mzs_tree_export: tree view has a null array");
  const mzs_config& c = h->cfg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MZS_HIP(h, hipSetDevice(c.device));
  const size_t BN = (size_t)c.batch * (c.num_simulations + 1);
  const mz::StepState& s = h->step;
#define CP(dst, src, n) MZS_HIP(h, hipMemcpyAsync(dst, src, (n) * 4, hipMemcpyDeviceToDevice, stream))
  CP(out->node_visits, s.node_visits, BN); CP(out->raw_values, s.raw_values, BN);
  CP(out->node_values, s.node_values, BN); CP(out->parents, s.parents, BN);
  CP(out->action_from_parent, s.action_from_parent, BN);
  CP(out->children_index, s.children_index, BN * c.num_actions);
  CP(out->children_prior_logits, s.children_prior_logits, BN * c.num_actions);
  CP(out->children_values, s.children_values, BN * c.num_actions);
  CP(out->children_visits, s.children_visits, BN * c.num_actions);
  CP(out->children_rewards, s.children_rewards, BN * c.num_actions);
  CP(out->children_discounts, s.children_discounts, BN * c.num_actions);
  CP(out->embeddings, s.embeddings, BN * c.embed_dim);
#undef CP
  return MZS_OK;
}

// ---------------------------------------------------------------------------
// training step of the default MLP trio
// ---------------------------------------------------------------------------
int64_t mzs_mlp_num_params(int32_t obs_dim, int32_t embed_dim, int32_t num_actions, int32_t support_size) {
  int off[19];
  mlp_offsets(obs_dim, embed_dim, num_actions, 2 * support_size + 1, off);
  return off[18];
}

int64_t mzs_mlp_train_workspace_bytes(int32_t batch, int32_t obs_dim, int32_t embed_dim, int32_t num_actions,
                                      int32_t support_size) {
  const int64_t waves = 4 * (int64_t)((batch + 15) / 16);
  return waves * (mzs_mlp_num_params(obs_dim, embed_dim, num_actions, support_size) + 1) * (int64_t)sizeof(float);
}

int mzs_mlp_loss_grad(const mzs_mlp_weights* w, const mzs_train_args* a, void* stream_) {
  if (!w || w->struct_size != (int32_t)sizeof(mzs_mlp_weights))
    return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: null weights or size mismatch (ABI)");
  if (!a || a->struct_size != (int32_t)sizeof(mzs_train_args))
    return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: null arguments or size mismatch (ABI)");
  const float* const* ptrs = &w->repr_w;
  for (int i = 0; i < 18; ++i)
    if (!ptrs[i]) return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: null weight pointer");
  if (a->batch <= 0 || a->unroll_steps <= 0) return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: batch and unroll_steps must be positive");
  if (!a->obs || !a->actions || !a->rewards || !a->returns || !a->policy || !a->loss || !a->grads || !a->workspace)
    return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: null batch / output / workspace pointer");
  if (w->obs_dim <= 0 || w->obs_dim > 16) return fail(nullptr, MZS_E_UNSUPPORTED, "mzs_mlp_loss_grad: obs_dim must be 1..16");
  const int A = a->num_actions, E = a->embed_dim, F = 2 * w->support_size + 1;
  if (a->workspace_bytes < mzs_mlp_train_workspace_bytes(a->batch, w->obs_dim, E, A, w->support_size))
    return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: workspace too small");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MZS_E_NODEVICE, "mzs_mlp_loss_grad: no HIP device (this library has no CPU fallback)");
  if (a->device < 0 || a->device >= ndev) return fail(nullptr, MZS_E_INVALID, "mzs_mlp_loss_grad: bad device ordinal");
  MZS_HIP(nullptr, hipSetDevice(a->device));
  mz::TrainParams p;
  memset(&p, 0, sizeof p);
  p.obs = a->obs; p.act = a->actions; p.rew = a->rewards; p.ret = a->returns; p.pi = a->policy;
  for (int i = 0; i < 18; ++i) p.w[i] = ptrs[i];
  mlp_offsets(w->obs_dim, E, A, F, p.off);
  p.B = a->batch; p.L = a->unroll_steps; p.obs_dim = w->obs_dim; p.support = w->support_size;
  p.loss_scale = a->loss_scale; p.l2 = a->l2_coeff;
  p.ws = static_cast<float*>(a->workspace); p.grads = a->grads; p.loss = a->loss;
  p.waves = 4 * ((a->batch + 15) / 16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (A == 2 && E == 8 && F == 21) return launch_train<mz::TrainCfg<2, 8, 21>>(p, stream);
  if (A == 4 && E == 32 && F == 21) return launch_train<mz::TrainCfg<4, 32, 21>>(p, stream);
  if (A == 3 && E == 8 && F == 21) return launch_train<mz::TrainCfg<3, 8, 21>>(p, stream);
  if (A == 4 && E == 8 && F == 21) return launch_train<mz::TrainCfg<4, 8, 21>>(p, stream);
  if (A == 2 && E == 16 && F == 21) return launch_train<mz::TrainCfg<2, 16, 21>>(p, stream);
  if (A == 4 && E == 16 && F == 21) return launch_train<mz::TrainCfg<4, 16, 21>>(p, stream);
  if (A == 2 && E == 10 && F == 21) return launch_train<mz::TrainCfg<2, 10, 21>>(p, stream);  // the reference notebooks
  if (A == 4 && E == 10 && F == 21) return launch_train<mz::TrainCfg<4, 10, 21>>(p, stream);
  if (A == 6 && E == 8 && F == 21) return launch_train<mz::TrainCfg<6, 8, 21>>(p, stream);
  if (A == 8 && E == 8 && F == 21) return launch_train<mz::TrainCfg<8, 8, 21>>(p, stream);
  if (A == 2 && E == 32 && F == 21) return launch_train<mz::TrainCfg<2, 32, 21>>(p, stream);
  if (A == 2 && E == 8 && F == 31) return launch_train<mz::TrainCfg<2, 8, 31>>(p, stream);  // support_size 15, 20
  if (A == 2 && E == 8 && F == 41) return launch_train<mz::TrainCfg<2, 8, 41>>(p, stream);
  {
    JitTrainLaunch fn = nullptr;  // an instance built on demand (mzs_register_train_dispatch; muax_amd/_jit.py)
    {
      std::lock_guard<std::mutex> lock(g_jit_mutex);
      for (const JitTrain& t : g_jit_train)
        if (t.A == A && t.E == E && t.F == F) fn = t.launch;
    }
    if (fn) {
      char msg[256] = "";
      const int rc = fn(&p, stream_, msg, (int)sizeof msg);
      return rc == MZS_OK ? MZS_OK : fail(nullptr, rc, "mzs_mlp_loss_grad (on-demand instance): %s", msg);
    }
  }
  return fail(nullptr, MZS_E_UNSUPPORTED, "mzs_mlp_loss_grad: no kernel instance for this (A, E, F)");
}

// ---------------------------------------------------------------------------
// device self-test of the hardware-dependent arithmetic identities
// ---------------------------------------------------------------------------
namespace mz {
__global__ void selftest_kernel(unsigned long long* bad) {
  // every binary32 in [1, 4): sqrt_normal vs the IEEE sqrt; the same mantissas at 2^-9 .. 2^-2: div_two_eps vs x / 0.002f;
  // bad[2], bad[3]: see below
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // 2^24 threads
  const float x = __uint_as_float(0x3f800000u + i);
  unsigned long long b0 = sqrt_normal(x) != sqrtf(x);
  unsigned long long b1 = 0;
  for (int e = 118; e <= 125; ++e) {
    const float y = __uint_as_float(((uint32_t)e << 23) | (i & 0x7fffffu));
    b1 += div_two_eps(y) != y / 0.002f;
  }
  // shared-reciprocal division (rcp_newton2 / div_newton2) vs n / d: 2^24 denominators spread over [1, 64), each with
  // numerators 0, 2^-100, d itself, d's predecessor and eight pseudo-random ones in [2^-100, d]
  unsigned long long b2 = 0;
  {
    uint32_t h = i * 2654435761u + 0x9e3779b9u;
    const float d = __uint_as_float(((127u + i % 6u) << 23) | (h >> 9));
    const f32x2 dd = (f32x2){d, d};
    const f32x2 y = rcp_newton2(dd);
    float ns[12] = {0.0f, 0x1p-100f, d, __uint_as_float(__float_as_uint(d) - 1u)};
    for (int k = 4; k < 12; ++k) {
      h = h * 1664525u + 1013904223u;
      const uint32_t ex = 27u + (h >> 7) % 106u;  // 2^-100 .. 2^5
      h = h * 1664525u + 1013904223u;
      const float n = __uint_as_float((ex << 23) | (h >> 9));
      ns[k] = n <= d ? n : d * 0.37f;
    }
    for (int k = 0; k < 12; k += 2) {
      const f32x2 q = div_newton2((f32x2){ns[k], ns[k + 1]}, dd, y);
      b2 += (q.x != ns[k] / d) + (q.y != ns[k + 1] / d);
    }
  }
  // the same for the value scores' range: denominators (the span) spread over 2^-27 .. 2^41, numerators 0, the span
  // itself and pseudo-random ones in [2^-100, span]
  unsigned long long b3 = 0;
  {
    uint32_t h = i * 2246822519u + 0x85ebca6bu;
    const float d = __uint_as_float(((100u + i % 68u) << 23) | (h >> 9));
    const f32x2 dd = (f32x2){d, d};
    const f32x2 y = rcp_newton2(dd);
    float ns[8] = {0.0f, d};
    for (int k = 2; k < 8; ++k) {
      h = h * 1664525u + 1013904223u;
      const uint32_t ex = 27u + (h >> 7) % 142u;  // 2^-100 .. 2^41
      h = h * 1664525u + 1013904223u;
      const float n = __uint_as_float((ex << 23) | (h >> 9));
      ns[k] = n <= d ? n : d * 0.61f;
      ns[k] = ns[k] < 0x1p-100f ? 0x1p-100f : ns[k];
    }
    for (int k = 0; k < 8; k += 2) {
      const f32x2 q = div_newton2((f32x2){ns[k], ns[k + 1]}, dd, y);
      b3 += (q.x != ns[k] / d) + (q.y != ns[k + 1] / d);
    }
  }
  if (b0) atomicAdd(&bad[0], b0);
  if (b1) atomicAdd(&bad[1], b1);
  if (b2) atomicAdd(&bad[2], b2);
  if (b3) atomicAdd(&bad[3], b3);
}
}  // namespace mz

int mzs_selftest(int32_t device, int64_t* mismatches) {
  if (!mismatches) return fail(nullptr, MZS_E_INVALID, "mzs_selftest: null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MZS_E_NODEVICE, "mzs_selftest: no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, MZS_E_INVALID, "mzs_selftest: bad device ordinal");
  MZS_HIP(nullptr, hipSetDevice(device));
  unsigned long long* d = nullptr;
  MZS_HIP(nullptr, hipMalloc(reinterpret_cast<void**>(&d), 32));
  MZS_HIP(nullptr, hipMemset(d, 0, 32));
  hipLaunchKernelGGL(mz::selftest_kernel, dim3((1u << 24) / 256), dim3(256), 0, nullptr, d);
  unsigned long long h2[4] = {0, 0, 0, 0};
  hipError_t e = hipMemcpy(h2, d, 32, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) return fail(nullptr, MZS_E_RUNTIME, "mzs_selftest: %s", hipGetErrorString(e));
  mismatches[0] = (int64_t)h2[0];
  mismatches[1] = (int64_t)h2[1];
  mismatches[2] = (int64_t)h2[2];
  mismatches[3] = (int64_t)h2[3];
  return MZS_OK;
}

// ---------------------------------------------------------------------------
// root exploration noise
// ---------------------------------------------------------------------------
int mzs_dirichlet(int32_t device, const uint32_t key[2], float alpha, int32_t batch, int32_t num_actions,
                  int64_t global_batch, int64_t root_offset, float* out, void* stream_) {
  if (!key || !out) return fail(nullptr, MZS_E_INVALID, "mzs_dirichlet: null argument");
  if (batch <= 0 || num_actions <= 0 || num_actions > 64) return fail(nullptr, MZS_E_INVALID, "mzs_dirichlet: batch / num_actions (1..64)");
  if (!(alpha > 0.0f)) return fail(nullptr, MZS_E_INVALID, "mzs_dirichlet: alpha must be positive");
  if (global_batch <= 0) global_batch = batch;
  if (root_offset < 0 || root_offset + batch > global_batch)
    return fail(nullptr, MZS_E_INVALID, "mzs_dirichlet: root_offset + batch exceeds global_batch");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MZS_E_NODEVICE, "mzs_dirichlet: no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, MZS_E_INVALID, "mzs_dirichlet: bad device ordinal");
  MZS_HIP(nullptr, hipSetDevice(device));
  const int R = (256 / mz::kSpec) / num_actions;  // roots per workgroup
  hipLaunchKernelGGL(mz::dirichlet_kernel, dim3((batch + R - 1) / R), dim3(256), sizeof(float) * (size_t)R * num_actions,
                     static_cast<hipStream_t>(stream_), key[0], key[1], alpha, batch, num_actions, (uint64_t)global_batch,
                     (uint64_t)root_offset, out);
  MZS_HIP(nullptr, hipGetLastError());
  return MZS_OK;
}

}  // extern "C"
