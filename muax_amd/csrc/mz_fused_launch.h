// mz_fused_launch.h -- the fused act() kernel is compiled as several translation units (groups of instances, built
// in parallel); each group exports one dispatcher.  Instances are listed in mz_instances.def.
#pragma once
#include <string>

#include "mz_fused.cuh"

namespace mz {
constexpr int kNoFusedInstance = 1;  // dispatcher result: this group has no instance for the shape
// kNeedPathScratch + w: an instance with its root paths in HBM fits, but p.path_scratch is not set or was allocated with
// fewer than w words per node (p.path_words): allocate [B][S + 1][w] words and call again
constexpr int kNeedPathScratch = 1000;
// an instance that keeps its embeddings in HBM fits, no tree export was asked for and p.emb_scratch is not set: allocate
// [B][S + 1][E] floats and call again
constexpr int kNeedEmbScratch = 2;
// mode: FusedCfg::MODE (0 muzero, 1 muzero + tie-break noise, 2 gumbel / parent-and-siblings, 3 gumbel / mix value).
// Returns MZS_OK after the launch, kNoFusedInstance, or a negative MZS_E_* with *err set.
// compact: take an instance with the compact tree record (FusedCfg::PH; needs p.path_scratch), else a plain one.
using FusedDispatch = int (*)(int mode, int device, const FusedParams& p, hipStream_t stream, int A, int E, int F,
                              int N, bool compact, std::string* err);
int fused_dispatch_g0(int mode, int device, const FusedParams& p, hipStream_t stream, int A, int E, int F, int N, bool compact, std::string* err);
int fused_dispatch_g1(int mode, int device, const FusedParams& p, hipStream_t stream, int A, int E, int F, int N, bool compact, std::string* err);
int fused_dispatch_g2(int mode, int device, const FusedParams& p, hipStream_t stream, int A, int E, int F, int N, bool compact, std::string* err);
int fused_dispatch_g3(int mode, int device, const FusedParams& p, hipStream_t stream, int A, int E, int F, int N, bool compact, std::string* err);
int fused_dispatch_g4(int mode, int device, const FusedParams& p, hipStream_t stream, int A, int E, int F, int N, bool compact, std::string* err);
}  // namespace mz
