// mz_search_conv.hip -- the whole simulation loop of a search with the reference's ResNet nets as ONE launch.
//
// mctx runs, per simulation, simulate -> recurrent_fn -> expand -> backward over the whole batch (muax/policy.py:13-30 ->
// mctx.muzero_policy; recurrent_fn = muax/model.py:265-282 on muax/nn.py:313-378).  The step-wise C-ABI mirrors that as
// 2 launches per simulation (mzs_resnet_tower, mzs_expand_backup_select): 400 kernel boundaries per 200-simulation act,
// every one of them paced by the slowest root (the tree kernel by the DEEPEST path of the batch), plus two 9 KB row
// copies per root and simulation between the tree's embedding array and the kernels' staging buffers.  Roots never
// interact, so none of that is needed: here the workgroup(s) that own a root run
//
//     for sim in [sim_begin, sim_end):   recurrent_fn(embeddings[parent], action) -> embeddings[new node]   (mz_conv.cuh)
//                                        expand + backward + refresh of the cached decisions + simulate() of sim + 1
//                                                                                                     (mz_step_jump.cuh)
//
// back to back: the next state is written straight into the tree's row of the new node, the next pass reads its input
// from the parent's row, reward / value / prior logits never leave the CU, and a root with a short path does not wait
// for one with a long path.  Same device functions as the step-wise kernels, same order of operations: same bits.
//
//   * one workgroup per root (any batch): no communication between workgroups at all;
//   * pair mode (<= 128 roots, two workgroups per root splitting the pixels, mz_conv.cuh): the halves already meet in
//     their XCD's L2 after every convolution pass; half 0 (which also evaluates both heads) owns the tree, and one more
//     message per simulation tells half 1 the next simulation's (parent, action, new node).
#include <hip/hip_runtime.h>

#include <cstdlib>

#define MZ_NO_STEP_KERNELS   // device functions and types of the step-wise path only: its kernels live in mz_api.hip,
#define MZ_NO_TOWER_KERNELS  // the recurrent kernel's in mz_conv.hip
#include "mz_conv_host.h"
#include "mz_step_jump.cuh"

#pragma clang fp contract(off)

namespace mz {

struct SearchLoop {
  int sim_begin, sim_end;
  float discount;
};

// has this root's pair lost a rendezvous?  One thread looks, every thread gets the same answer (the word may be
// written concurrently: a per-thread read could split the workgroup around its next barrier)
MZ_DEV bool pair_lost_uniform(const PairLink& L, int tid, int* flag_lds) {
  __syncthreads();
  if (tid == 0) *flag_lds = (int)*reinterpret_cast<const volatile unsigned*>(L.status);
  __syncthreads();
  return *flag_lds != 0;
}

// LDSTREE (round 6, MuZero policy): the statistics the tree step reads and rewrites -- children_{index, visits,
// prior_probs, rewards, values}[N][A], node_{visits, values}[N], the JUMP records [N] -- live in this CU's LDS for the
// whole launch (copied in from the handle's HBM tree at the start, written back at the end; 75.6 KB for config 4's 201
// nodes x 18 actions, next to the pass' 66 KB, the step's 14 KB and 8 KB of head weights: 163 784 of the CU's 163 840 bytes).  Between two tree steps
// of a root its XCD streams the 5.7 MB of convolution weights through its 4 MB L2: the step's ~260 tree lines were gone
// every time, and a 44-level decision refresh spent 4.5 of its 7 us on memory round trips.  Same device functions on
// the same values in the same order (a TreeView of LDS rows instead of HBM rows): same bits.
// AS (LDSTREE instances): the 16-lane slots the action count fills (1: A <= 16, 2: A <= 32) -- the decision refresh of
// a path level without the run-time tests of the general code's four slots (mz_step_jump.cuh, level_load); 0: any A
template <bool GUMBEL, bool PAIRED, bool LDSTREE, int AS = 0>
__global__ __launch_bounds__(256) void mz_resnet_search_kernel(const TowerParams p, const StepArgs s, const JumpArgs g,
                                                               const SearchLoop loop) {
  static_assert(!(GUMBEL && LDSTREE), "the Gumbel policy's decisions read the HBM tree (row_qtransform)");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int* tree_lds = reinterpret_cast<int*>(lds + 2 * kBufWords + kHeadWords);
  __shared__ int lost_flag;
  const int tid = threadIdx.x;
  // {sqrt(n) pb_c(n), RN(1 / n)} by visit count, behind the tree step's arrays (level_compute): while every count
  // stays within Markstein's tested range
  float* score_tbl = (!GUMBEL && s.S + 2 <= 1030) ? reinterpret_cast<float*>(tree_lds + 15 * (s.S + 2)) : nullptr;
  if (score_tbl) {
    for (int n = tid; n < s.S + 2; n += 256) {
      score_tbl[2 * n] = puct_scale(n, s.pb_c_init, s.pb_c_base);
      score_tbl[2 * n + 1] = n > 0 ? 1.0f / (float)n : 0.0f;
    }
  }  // (the passes' barriers come before its first use)
  int r, h = 0;
  if constexpr (PAIRED) {  // 16 blocks = 8 roots x 2 halves, the halves of a root 8 blocks apart (same XCD)
    r = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    h = (blockIdx.x >> 3) & 1;
  } else {
    r = blockIdx.x;
  }
  if (r >= p.B) return;
  const int N = s.N, A = s.A, E = s.E;
  const size_t rb = (size_t)r * N;
  PairLink L;
  if constexpr (PAIRED) {
    if (h == 0) pair_link_init<1>(p, r, L);
    else pair_link_init<2>(p, r, L);
  }
  // the tree statistics of this root: rows of the handle's HBM tree, or their copy in LDS (half 0 / the single workgroup)
  TreeView T = tree_view_global(s, g, rb);
  const bool owns_tree = !PAIRED || h == 0;
  int depth_acc = 0;  // LDSTREE: the selection depths of this launch (added to depth_sum[r] at its end)
  if constexpr (LDSTREE) {
    // child records {index, visits, prior prob, reward, value} and node records {visits, value, JUMP parent | action,
    // JUMP level}, interleaved: one address per (node, action) / node, the fields at immediate offsets
    int* tl = tree_lds + 17 * (s.S + 2);
    const int NA = N * A;
    T.cs = 5; T.ns = 4;
    T.cidx = tl; T.cvis = tl + 1; T.prob = reinterpret_cast<float*>(tl + 2); T.rew = reinterpret_cast<float*>(tl + 3);
    T.val = reinterpret_cast<float*>(tl + 4);
    T.dis = nullptr; T.disc = loop.discount;
    int* nl = tl + 5 * NA;
    T.nvis = nl; T.nval = reinterpret_cast<float*>(nl + 1); T.jpa = nl + 2; T.jlv = nl + 3;
    // root_invalid_actions of this lane's two action slots, once per launch (the root is refreshed every simulation)
    T.inv_bits = 0;
    if (owns_tree) {
      const int j = tid & 15;
#pragma unroll
      for (int t = 0; t < AS; ++t)
        if (j + 16 * t < A && s.root_invalid[(size_t)r * A + j + 16 * t]) T.inv_bits |= 1 << t;
    }
    if (owns_tree) {
      const size_t o = rb * A;
      for (int i = tid; i < NA; i += 256) {
        T.cidx[5 * i] = s.children_index[o + i];
        T.cvis[5 * i] = s.children_visits[o + i];
        T.prob[5 * i] = s.children_prior_probs[o + i];
        T.rew[5 * i] = s.children_rewards[o + i];
        T.val[5 * i] = s.children_values[o + i];
      }
      for (int i = tid; i < N; i += 256) {
        T.nvis[4 * i] = s.node_visits[rb + i];
        T.nval[4 * i] = s.node_values[rb + i];
        T.jpa[4 * i] = g.jump_pa[rb + i];
        T.jlv[4 * i] = g.jump_lv[rb + i];
      }
      __syncthreads();
    }
  }
  // simulate() of sim_begin has run (mzs_select, or the tail of the previous launch): its decision is in the handle
  int parent = s.sel_parent[r], action = s.sel_action[r], depth = s.sel_depth[r];
  int newn;
  {
    const int next = owns_tree ? T.cidx[(parent * A + action) * T.cs] : s.children_index[(rb + parent) * A + action];
    newn = next == -1 ? loop.sim_begin + 1 : next;
  }
  TowerIO io;
  io.reward = p.reward + r;
  io.value = p.value + r;
  io.prior_logits = p.prior_logits + (size_t)r * A;
  if constexpr (LDSTREE) {
    // reward / value / prior logits of a pass go from the heads to the tree step of the SAME workgroup: through LDS, not
    // through an HBM word and back (two L2 round trips per simulation) -- in words of the pass' own scratch that are
    // dead between the heads and the next pass for the prior logits (the Linear layers' partial sums: A <= 32 of 256
    // words, written after their last reader of the simulation), two words of their own for reward and value (one
    // workgroup per root evaluates the reward head BEFORE the passes: nothing of the pass' scratch survives those)
    io.prior_logits = lds + 2 * kBufWords + 32 + 3 * 768;
    // the prediction heads' first 1x1 convolutions (v_c1 | p_c1, 2 x 64 x 16 floats) behind the node records, copied once
    float* hw = reinterpret_cast<float*>(tree_lds + 17 * (s.S + 2) + 5 * N * A + 4 * N);
    io.reward = hw + 2 * kTowerC * 16;
    io.value = hw + 2 * kTowerC * 16 + 1;
    if (owns_tree) {
      for (int i = tid; i < kTowerC * 16; i += 256) {
        hw[i] = p.v_c1[i];
        hw[kTowerC * 16 + i] = p.p_c1[i];
      }
    }
    io.head_w = hw;
    io.head_w_lds = true;
  }
#ifdef MZ_PROFILE
  unsigned long long st[4] = {0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define MZ_ST(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[k] += t_ - tl; tl = t_; }
#else
#define MZ_ST(k)
#endif
  for (int sim = loop.sim_begin; sim < loop.sim_end; ++sim) {
    // pair mode: rows of the tree's embedding array hold pixels the partner's CU wrote in earlier simulations (in L2 by
    // now: messages Y / T below); this CU's L1 may hold older copies of those lines
    if constexpr (PAIRED) asm volatile("buffer_inv sc0" ::: "memory");
    io.x = s.embeddings + (rb + parent) * E;
    io.y = s.embeddings + (rb + newn) * E;
    io.action = action;
    io.lds_clean = sim != loop.sim_begin;
    const bool more = sim + 1 < loop.sim_end && sim + 1 < s.S;
    int sel[3] = {0, 0, 0};
    if (!PAIRED || h == 0) {
      int prefetched = 0;
      if constexpr (PAIRED) {
        // the half's idle time in passes 9 and 10: the path of the coming backup and its per-level inputs (they need
        // nothing from this pass) go from the tree into LDS
        const JumpLds JL = jump_lds(tree_lds, N);
        const bool fresh = newn == sim + 1;
        auto idle = [&](int k) {
          if (k == 0) {
            jump_prefetch_path(s, g, r, JL, tid, 256, parent, action, depth, newn, fresh);
            prefetched |= 1;
          } else if (k == 1) {
            jump_prefetch_levels(s, T, JL, tid, 256, depth);
            prefetched |= 2;
          }
        };
        tower_body<1>(p, io, lds, L, idle);  // ... ends with the reward head's tail and the prediction heads of this root
        MZ_ST(0)
        if (pair_lost_uniform(L, tid, &lost_flag)) return;  // the host sees the status word and repeats the search
        MZ_ST(1)
      } else {
        tower_body<0>(p, io, lds, L);
        MZ_ST(0)
      }
      __syncthreads();  // reward / value / prior logits were written by other threads of this workgroup
      const float rew = *reinterpret_cast<const volatile float*>(io.reward);
      const float val = *reinterpret_cast<const volatile float*>(io.value);
      const int known[4] = {parent, action, depth, newn};
#ifndef MZ_SEARCH_LIF
#define MZ_SEARCH_LIF kLevelsInFlight
#endif
      jump_expand_backup_body<GUMBEL, MZ_SEARCH_LIF, AS, LDSTREE, LDSTREE>(s, g, T, sim, r, tree_lds, rew, loop.discount, io.prior_logits, val, nullptr, true,
                                                         nullptr, nullptr, sel, known, prefetched, score_tbl, LDSTREE ? &depth_acc : nullptr);
      MZ_ST(2)
      if (more) {
        parent = sel[0];
        action = sel[1];
        depth = sel[2];
        const int next = T.cidx[(parent * A + action) * T.cs];
        newn = next == -1 ? sim + 2 : next;
      }
      if constexpr (PAIRED) {
        // message Y of half 1 (number k): its pixels of this simulation's new embedding row are in L2 -- the next pass
        // may read that row.  Half 0 posts nothing under that number.
        L.seq += 1;
        if (tid == 0) pair_check_xcc(L, pair_in(L), tid);
        if (more) {
          // message T (number k + 1): what the next pass works on; "strong": half 0's own pixels of the new row and the
          // tree are complete before the partner can see it (and every thread is past message Y)
          pair_word* msg = pair_begin_strong(L);
          // (one word: parent | action << 12 | new node << 20 -- the partner's next pass starts one L2 round trip after it)
          if (tid == 0) pair_put(L, msg, 0, __int_as_float(parent | (action << 12) | (newn << 20)));
        } else {
          L.seq += 1;
        }
      }
      __syncthreads();
      MZ_ST(3)
    } else {
      if constexpr (PAIRED) {
        tower_body<2>(p, io, lds, L);
        MZ_ST(0)
        {  // message Y: every store of this half (its pixels of the new row) has completed
          pair_word* msg = pair_begin_strong(L);
          if (tid == 0) pair_put(L, msg, 4, __uint_as_float(L.xcc));
        }
        L.seq += 1;  // (half 1 posts nothing under the number of message T)
        if (more) {
          const pair_word* in = pair_in(L);
          const unsigned w = (unsigned)__float_as_int(pair_get(L, in, 0));
          parent = (int)(w & 0xfffu);
          action = (int)((w >> 12) & 0xffu);
          newn = (int)(w >> 20);
          MZ_ST(2)
          if (pair_lost_uniform(L, tid, &lost_flag)) return;
          parent = min(max(parent, 0), N - 1);  // (a lost message must not turn into a wild address)
          action = min(max(action, 0), A - 1);
          newn = min(max(newn, 0), N - 1);
        }
      }
    }
  }
#ifdef MZ_PROFILE
  if (tid == 0)
    for (int k = 1; k < 4; ++k) g_tower_prof[(size_t)blockIdx.x * 16 + 11 + k] += st[k];  // slots 12..14 (0..11, 15: the pass itself)
#endif
  if constexpr (LDSTREE) {
    if (owns_tree) {  // back into the handle's tree: mzs_finish / mzs_tree_export / the next launch read it there
      __syncthreads();
      const size_t o = rb * A;
      const int NA = N * A;
      for (int i = tid; i < NA; i += 256) {
        s.children_index[o + i] = T.cidx[5 * i];
        s.children_visits[o + i] = T.cvis[5 * i];
        s.children_prior_probs[o + i] = T.prob[5 * i];
        s.children_rewards[o + i] = T.rew[5 * i];
        s.children_values[o + i] = T.val[5 * i];
      }
      for (int i = tid; i < N; i += 256) {
        s.node_visits[rb + i] = T.nvis[4 * i];
        s.node_values[rb + i] = T.nval[4 * i];
        g.jump_pa[rb + i] = T.jpa[4 * i];
        g.jump_lv[rb + i] = T.jlv[4 * i];
      }
      if (tid == 0) s.depth_sum[r] += depth_acc;
    }
  }
  if constexpr (PAIRED) {
    // the root's epoch advances by the simulations of this launch (each uses fewer than kPairMsgs message numbers);
    // both halves read it before their first message and half 0 is past the last one it waits for
    if (h == 0 && tid == 0) p.pair_u[(size_t)r * 4 + 2] += (unsigned)(loop.sim_end - loop.sim_begin);
  }
}

}  // namespace mz

extern "C" {

#ifdef MZ_PROFILE
// profiling builds only (tools/profile_search.py): read and clear this translation unit's per-workgroup phase counters
int mzs_debug_search_profile(uint64_t* host_out, int32_t words) {
  static unsigned long long zero[1024 * 16];
  if (words > 1024 * 16) words = 1024 * 16;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mz::g_tower_prof), sizeof(uint64_t) * (size_t)words) != hipSuccess) return MZS_E_RUNTIME;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mz::g_tower_prof), zero, sizeof(zero)) != hipSuccess) return MZS_E_RUNTIME;
  return MZS_OK;
}
int mzs_debug_jump_profile(uint64_t* host_out, int32_t words) {
  static unsigned long long zero[1024 * 8];
  if (words > 1024 * 8) words = 1024 * 8;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mz::g_jump_prof), sizeof(uint64_t) * (size_t)words) != hipSuccess) return MZS_E_RUNTIME;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mz::g_jump_prof), zero, sizeof(zero)) != hipSuccess) return MZS_E_RUNTIME;
  return MZS_OK;
}
#endif

int mzs_resnet_search(mzs_handle* h, const mzs_tower_args* a, float discount, int32_t sim_begin, int32_t sim_end,
                      void* stream_) {
  if (!h) return MZS_E_INVALID;
  mz::StepArgs sa;
  mz::JumpArgs ja;
  int policy = 0;
  int tree_device = -1;
  if (int rc = mzh::step_view(h, &sa, &ja, &policy, "mzs_resnet_search", &tree_device)) return rc;
  // the tree belongs to the handle's device; the nets' arrays are named by a->device -- one launch cannot serve two
  if (a && a->struct_size == (int32_t)sizeof(mzs_tower_args) && a->device != tree_device)
    return mzh::fail_handle(h, MZS_E_INVALID, "mzs_resnet_search: the nets' device differs from the handle's");
  mz::TowerParams p;
  if (int rc = tower_params_from_args(a, p, true)) return mzh::fail_handle(h, rc, mzs_last_error(nullptr));
  if (!p.heads) return mzh::fail_handle(h, MZS_E_INVALID, "mzs_resnet_search: needs the heads (the whole recurrent_fn)");
  if (a->batch != sa.B || a->num_actions != sa.A || sa.E != mz::kTowerPix * mz::kTowerC)
    return mzh::fail_handle(h, MZS_E_INVALID, "mzs_resnet_search: batch / num_actions / embedding (6x6x64) do not match the handle");
  if (sim_begin < 0 || sim_end > sa.S || sim_begin >= sim_end)
    return mzh::fail_handle(h, MZS_E_INVALID, "mzs_resnet_search: simulation range");
  if (2 * a->blocks + 3 > mz::kPairMsgs)
    return mzh::fail_handle(h, MZS_E_UNSUPPORTED, "mzs_resnet_search: too many blocks");
  size_t lds = sizeof(float) * (2 * (size_t)mz::kBufWords + mz::kHeadWords) + sizeof(int32_t) * 17 * ((size_t)sa.S + 2);  // (15 arrays of the tree step + the score table)
  if (lds > 160 * 1024) return mzh::fail_handle(h, MZS_E_UNSUPPORTED, "mzs_resnet_search: num_simulations too large for the LDS of a CU");
  // the tree's statistics in LDS as well when they fit next to that (MuZero policy; MZS_SEARCH_LDS_TREE=0: A/B, tests)
  const size_t lds_tree = lds + sizeof(int32_t) * (5 * (size_t)sa.N * sa.A + 4 * (size_t)sa.N + 2 * mz::kTowerC * 16 + 2);  // (+ the heads' 1x1 weights, reward, value)
  const char* lt = getenv("MZS_SEARCH_LDS_TREE");
  const bool ldstree = policy != 1 && sa.A <= 32 && sa.S + 2 <= 1030 && lds_tree <= 160 * 1024 && !(lt && lt[0] == '0');
  const bool wide = sa.A > 16;  // (LDS-tree instances: one or two 16-lane slots of actions)
  if (ldstree) lds = lds_tree;
  if (sa.S + 1 > 4096 || sa.A > 255)
    return mzh::fail_handle(h, MZS_E_UNSUPPORTED, "mzs_resnet_search: message T packs (node, action, node) as 12 + 8 + 12 bits");
  const mz::SearchLoop loop = {sim_begin, sim_end, discount};
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool gumbel = policy == 1;
  const void* fn;
  dim3 grid;
  if (a->pair_scratch) {
    const int64_t need = mzs_tower_pair_scratch_bytes(a->batch);
    if (need == 0) return mzh::fail_handle(h, MZS_E_UNSUPPORTED, "mzs_resnet_search: pair mode needs batch <= 128");
    if (a->pair_scratch_bytes < need) return mzh::fail_handle(h, MZS_E_INVALID, "mzs_resnet_search: pair_scratch too small");
    p.pair_f = static_cast<float*>(a->pair_scratch);
    p.pair_u = reinterpret_cast<unsigned*>(p.pair_f + (size_t)a->batch * 4 * mz::kPairSlot * 2);  // (8-byte words)
    fn = gumbel ? reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<true, true, false>)
                : (ldstree ? (wide ? reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<false, true, true, 2>)
                                   : reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<false, true, true, 1>))
                           : reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<false, true, false>));
    grid = dim3(16 * ((a->batch + 7) / 8));
  } else {
    fn = gumbel ? reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<true, false, false>)
                : (ldstree ? (wide ? reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<false, false, true, 2>)
                                   : reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<false, false, true, 1>))
                           : reinterpret_cast<const void*>(mz::mz_resnet_search_kernel<false, false, false>));
    grid = dim3(a->batch);
  }
  {
    // raise the kernel's dynamic-LDS limit once per (instance, device) and size
    static mzh::LdsGrant granted[8];
    mzh::LdsGrant& have = granted[(gumbel ? 6 : (ldstree ? (wide ? 4 : 2) : 0)) + (a->pair_scratch ? 1 : 0)];
    if (!have.covers(a->device, lds)) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return mzh::fail_handle(h, MZS_E_RUNTIME, "mzs_resnet_search: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      have.note(a->device, lds);
    }
  }
  void* args[] = {&p, &sa, &ja, const_cast<mz::SearchLoop*>(&loop)};
  if (hipLaunchKernel(fn, grid, dim3(256), args, lds, stream) != hipSuccess || hipGetLastError() != hipSuccess)
    return mzh::fail_handle(h, MZS_E_RUNTIME, "mzs_resnet_search: launch failed");
  return MZS_OK;
}

}  // extern "C"
