// ubench_netpass_e32.hip -- the per-simulation NETWORK PASS of the packed E = 32 instance of the fused search kernel
// (LunarLander, BASELINE configs[2]: Nets<FusedCfg<4, 32, 2, 51, 1, 4, true>>::forward of muax_amd/csrc/mz_fused.cuh) in
// isolation, as the product runs it: 512 workgroups x 16 roots, TWO workgroups per CU (two wavefronts per SIMD),
// first-layer weights in LDS.  `iters` dependent passes (next state fed back), cycles by s_memtime.
//   * the instruction ledger of the pass comes from this kernel's ISA (tools/netpass_ledger.py: the loop body is straight
//     line code);
//   * A/B switches of mz_fused.cuh (-DMZ_AB_...) are timed here against the product code, every output of every root
//     compared bit for bit with the build without switches (checksums written to a file: argv[3]).
// Build: hipcc --offload-arch=gfx950 <flags of muax_amd/_build.py> tools/ubench_netpass_e32.hip -o tools/bin/ubench_netpass_e32
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../muax_amd/csrc/mz_fused.cuh"

#pragma clang fp contract(off)

using namespace mz;
#ifndef NP_WAVES_PER_SIMD
#define NP_WAVES_PER_SIMD 2
#endif
using Cfg = FusedCfg<4, 32, 2, 51, 1, 4, true>;
constexpr int A = Cfg::A, E = Cfg::E, F = 21, H = kHidden, SUPPORT = 10;

struct BenchParams {
  FusedParams fp;
  const float* s0;   // [B][E]
  uint32_t* chk;     // [B] checksum over all passes
  float* last;       // [B][4 + E]: reward, value, pi logit (lane a), pi prob of the LAST pass + next state
  uint64_t* cycles;  // [waves]
  int iters;
};

__global__ __launch_bounds__(256, NP_WAVES_PER_SIMD) void netpass_e32(const BenchParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
  const int r = blockIdx.x * 16 + (tid >> 6) * 4 + (lane >> 4);
  Nets<Cfg>::fill_lds(p.fp, lds + Cfg::TBL_WORDS, tid);
  __syncthreads();
  Nets<Cfg> nets;
  nets.load(p.fp, j);
  nets.wlds = lds + Cfg::TBL_WORDS + 4 * j;
  float s[Cfg::ES] = {p.s0[(size_t)r * E + j], p.s0[(size_t)r * E + 16 + j]};
  int action = r % A;
  uint32_t chk = 0;
  float reward = 0, value = 0, pil = 0, pprob = 0, ns[Cfg::ES] = {0, 0};
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < p.iters; ++it) {
    asm volatile("; ---- pass begin" ::: "memory");
    nets.forward(s, action, j, SUPPORT, F, false, reward, value, pil, pprob, ns);
    asm volatile("; ---- pass end" ::: "memory");
    chk = (chk << 1 | chk >> 31) ^ f2u(reward) ^ (f2u(value) * 3u) ^ f2u(bcast<0>(pil)) ^ (f2u(bcast<1>(pprob)) * 5u) ^
          (f2u(bcast<3>(ns[0])) * 7u) ^ (f2u(bcast<9>(ns[1])) * 11u);
    s[0] = ns[0];
    s[1] = ns[1];
    action = (action + 1 + (it & 1)) % A;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) p.cycles[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
  if (j == 0) {
    p.chk[r] = chk;
    p.last[(size_t)r * (4 + E)] = reward;
    p.last[(size_t)r * (4 + E) + 1] = value;
  }
  if (j < A) {
    if (j == 0) p.last[(size_t)r * (4 + E) + 2] = pil;
    if (j == 1) p.last[(size_t)r * (4 + E) + 3] = pprob;
  }
  p.last[(size_t)r * (4 + E) + 4 + j] = ns[0];
  p.last[(size_t)r * (4 + E) + 4 + 16 + j] = ns[1];
}

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50, WG = argc > 2 ? atoi(argv[2]) : 512, B = 16 * WG;
  const char* ref = argc > 3 ? argv[3] : nullptr;   // checksum file: written when absent, compared when present
  std::mt19937 rng(0);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  auto dev = [&](size_t n, float scale, bool bias) {
    std::vector<float> h(n);
    for (auto& v : h) v = nd(rng) * scale * (bias ? 0.3f : 1.0f);
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  };
  BenchParams p;
  memset(&p, 0, sizeof p);
  FusedParams& f = p.fp;
  f.F = F;
  const float se = 1.0f / sqrtf((float)E), sx = 1.0f / sqrtf((float)(E + A)), sh = 0.25f;
  f.pv_w1 = dev(E * H, se, 0); f.pv_b1 = dev(H, 1, 1); f.pv_w2 = dev(H * F, sh, 0); f.pv_b2 = dev(F, 1, 1);
  f.pp_w1 = dev(E * H, se, 0); f.pp_b1 = dev(H, 1, 1); f.pp_w2 = dev(H * A, sh, 0); f.pp_b2 = dev(A, 1, 1);
  f.dr_w1 = dev((E + A) * H, sx, 0); f.dr_b1 = dev(H, 1, 1); f.dr_w2 = dev(H * F, sh, 0); f.dr_b2 = dev(F, 1, 1);
  f.dn_w1 = dev((E + A) * H, sx, 0); f.dn_b1 = dev(H, 1, 1); f.dn_w2 = dev(H * E, sh, 0); f.dn_b2 = dev(E, 1, 1);
  {
    std::vector<float> h((size_t)B * E);
    std::uniform_real_distribution<float> ud(0.0f, 1.0f);
    for (auto& v : h) v = ud(rng);
    float* d;
    CK(hipMalloc(&d, h.size() * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    p.s0 = d;
  }
  CK(hipMalloc(&p.chk, 4 * B));
  CK(hipMalloc(&p.last, 4 * (size_t)B * (4 + E)));
  CK(hipMalloc(&p.cycles, 8 * WG * 4));
  p.iters = iters;
  // the product's LDS footprint (tables + weights + 16 trees): what makes TWO workgroups share a CU, no more
  const int lds_bytes = NP_WAVES_PER_SIMD == 2 ? Cfg::LDS_BYTES : 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&netpass_e32), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(netpass_e32, dim3(WG), dim3(256), lds_bytes, 0, p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<uint64_t> c(WG * 4);
  CK(hipMemcpy(c.data(), p.cycles, 8 * WG * 4, hipMemcpyDeviceToHost));
  double sum = 0, mxv = 0;
  for (auto x : c) { sum += (double)x; mxv = mxv > (double)x ? mxv : (double)x; }
  std::vector<uint32_t> hc(B);
  std::vector<float> hl((size_t)B * (4 + E));
  CK(hipMemcpy(hc.data(), p.chk, 4 * B, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hl.data(), p.last, 4 * hl.size(), hipMemcpyDeviceToHost));
  printf("network pass, LunarLander shapes (A=4 E=32 F=21 H=16, packed record), %d workgroups x 16 roots, %d waves/SIMD, %d dependent passes: "
         "%8.1f cycles/pass (slowest wave %8.1f), kernel %.1f us\n", WG, NP_WAVES_PER_SIMD, iters, sum / c.size() / iters, mxv / iters, ms * 1e3);
  int rc = 0;
  if (ref) {
    FILE* fh = fopen(ref, "rb");
    if (!fh) {
      fh = fopen(ref, "wb");
      fwrite(hc.data(), 4, B, fh);
      fwrite(hl.data(), 4, hl.size(), fh);
      fclose(fh);
      printf("  reference outputs written to %s\n", ref);
    } else {
      std::vector<uint32_t> rc_(B);
      std::vector<float> rl(hl.size());
      size_t n1 = fread(rc_.data(), 4, B, fh), n2 = fread(rl.data(), 4, rl.size(), fh);
      fclose(fh);
      size_t bad = 0, badl = 0;
      for (int i = 0; i < B; ++i) bad += rc_[i] != hc[i];
      badl = memcmp(rl.data(), hl.data(), 4 * hl.size()) != 0;
      printf("  against %s: %zu / %d roots differ in the all-pass checksum, last pass' outputs %s (%zu, %zu words read)\n", ref, bad, B,
             badl ? "DIFFER" : "identical", n1, n2);
      rc = (bad || badl) ? 2 : 0;
    }
  }
  printf("  sample root 0: reward %.9g value %.9g ns0 %.9g\n", hl[0], hl[1], hl[4]);
  return rc;
}
