// ubench_coissue.hip -- do the matrix pipe and the vector ALU of a SIMD run side by side when TWO wavefronts share
// the SIMD (the compact instances of the fused search kernel: two 16-root workgroups per CU)?  Decides whether the
// network pass' fma chains (v_fmac_f32_dpp, VALU) are worth moving onto v_mfma_f32_4x4x1_16b_f32.
//   role V: `iters` x 64 v_fmac_f32 in two dependent chains (the shape of a first layer of the E = 32 trio)
//   role M: `iters` x 32 dependent v_mfma_f32_4x4x1 (one chain: the same layer as 8 blocks x 4 roots)
// cases: one wave per SIMD (V | M), two waves per SIMD (V+V | M+M | V+M).  Also checks that a chain of 4x4x1 MFMAs is
// the k-ordered fmaf chain bit for bit (denormal products and results included).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench_coissue.hip -o tools/bin/ubench_coissue
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#pragma clang fp contract(off)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOP>
__global__ __launch_bounds__(512) void coissue(int mode, int iters, const float* in, float* out, unsigned long long* cyc) {
  const int tid = threadIdx.x, wave = tid >> 6;
  // waves w and w + 4 of a workgroup share a SIMD
  const int role = mode == 2 ? ((wave >> 2) & 1) : mode;  // 0: V, 1: M
  float x = in[tid], w0 = in[512 + tid], w1 = in[1024 + tid];
  float a0 = 0.0f, a1 = 0.0f;
  f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        asm volatile("v_fmac_f32_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %1, %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1)
                     : "v"(x), "v"(w0), "v"(w1));
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(x, w0, d, 0, 0, 0);
        if constexpr (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = a0 + a1 + d[0] + d[1] + d[2] + d[3];
  if ((tid & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// exactness: D[i][j] of block b after K steps against fmaf chains
__global__ void exact(const float* A, const float* B, float* D, int K) {
  const int lane = threadIdx.x;
  f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = 0; k < K; ++k) d = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k * 64 + lane], B[k * 64 + lane], d, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[i * 64 + lane] = d[i];
}

int main() {
  const int blocks = 256, iters = 200;
  std::vector<float> h(1536);
  std::mt19937 g(1);
  std::uniform_real_distribution<float> u(-1.0f, 1.0f);
  for (auto& v : h) v = u(g) * 1e-3f;
  float *in, *out;
  unsigned long long* cyc;
  hipMalloc(&in, h.size() * 4);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  struct Case { const char* name; int mode, threads; } cases[] = {
      {"one wave per SIMD, V (64 v_fmac_f32_dpp / iter)", 0, 256}, {"one wave per SIMD, M (32 dependent mfma 4x4x1 / iter)", 1, 256},
      {"two waves per SIMD, V + V", 0, 512},                        {"two waves per SIMD, M + M", 1, 512},
      {"two waves per SIMD, V + M", 2, 512}};
  for (int nop = -1; nop <= 7; ++nop)
  for (auto& c : cases) {
    if (nop >= 0 && c.mode == 0) continue;
    for (int rep = 0; rep < 2; ++rep) {
      auto L = [&](auto k) { hipLaunchKernelGGL(k, dim3(blocks), dim3(c.threads), 0, 0, c.mode, iters, in, out, cyc); };
      switch (nop) {
        case -1: L(coissue<-1>); break; case 0: L(coissue<0>); break; case 1: L(coissue<1>); break; case 2: L(coissue<2>); break;
        case 3: L(coissue<3>); break; case 4: L(coissue<4>); break; case 5: L(coissue<5>); break; case 6: L(coissue<6>); break;
        default: L(coissue<7>); break;
      }
    }
    printf("[s_nop %2d after every MFMA] ", nop);
    hipDeviceSynchronize();
    const int nw = c.threads / 64;
    std::vector<unsigned long long> hc(blocks * nw);
    hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    double s[2] = {0, 0};
    int n[2] = {0, 0};
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < nw; ++w) {
        const int role = c.mode == 2 ? ((w >> 2) & 1) : c.mode;
        s[role] += (double)hc[b * nw + w];
        n[role] += 1;
      }
    // s_memtime counts at 100 MHz on gfx950: report in its own ticks per iteration and the ratio between cases
    printf("%-56s", c.name);
    if (n[0]) printf("  V: %8.2f ticks/iter", s[0] / n[0] / iters);
    if (n[1]) printf("  M: %8.2f ticks/iter", s[1] / n[1] / iters);
    printf("\n");
  }
  // ---- exactness ----
  const int K = 36;
  std::vector<float> A(K * 64), B(K * 64), D(256), R(256);
  int bad = 0, total = 0, denorm_seen = 0;
  for (int trial = 0; trial < 200; ++trial) {
    const float scale = trial % 4 == 0 ? 1e-19f : (trial % 4 == 1 ? 1e-22f : 1.0f);  // products / sums around and below 2^-126
    for (auto& v : A) v = u(g) * scale;
    for (auto& v : B) v = u(g) * (trial % 4 == 3 ? 1e-30f : scale);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dD);
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 4; ++i) {
        const int blk = lane >> 2;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = fmaf(A[k * 64 + 4 * blk + i], B[k * 64 + lane], acc);
        uint32_t x, y;
        memcpy(&x, &acc, 4); memcpy(&y, &D[i * 64 + lane], 4);
        total += 1;
        if (acc != 0.0f && std::fabs(acc) < 1.17549435e-38f) denorm_seen += 1;
        if (x != y) {
          if (bad < 5) printf("  mismatch trial %d lane %d i %d: fmaf chain %.9g (%08x) mfma %.9g (%08x)\n", trial, lane, i, acc, x, D[i * 64 + lane], y);
          bad += 1;
        }
      }
  }
  printf("exactness: %d / %d outputs of 36-step v_mfma_f32_4x4x1 chains differ from the k-ordered fmaf chain (%d denormal results among them)\n",
         bad, total, denorm_seen);
  return 0;
}
