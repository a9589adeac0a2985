// ubench_netpass.hip -- A/B of the per-simulation NETWORK PASS of the fused search kernel, in isolation.
//
//   A  the product code: Nets<C>::forward of muax_amd/csrc/mz_fused.cuh -- one root per DPP row (16 lanes), 4 roots
//      per wavefront, row-distributed v_pk_fma_f32 chains, DPP butterflies;
//   B  the workgroup-cooperative fp32-MFMA formulation VERDICT r1 asked for: the 16 roots of a workgroup are the
//      N = 16 columns of v_mfma_f32_16x16x4_f32 tiles (weights = A operand, activations = B operand), the waves
//      take roles (reward path | state -> value path | state -> policy path), layer outputs stay in the MFMA's
//      C layout (4 elements per lane) for ELU / min-max / softmax / decode and are re-fed as B operands after a
//      4x4 (lane row, register) transpose on v_permlane{16,32}_swap.  Inputs and outputs cross between the
//      row owners and the role waves through LDS + two workgroup barriers per pass -- the honest interface cost
//      inside the real kernel, where select and backup stay row-per-root.
//
// Both variants follow the MZ-F32 spec (k-ordered fma chains, canonical 16-wide sums) and must produce the SAME
// BITS: the host compares every output of every root.  Timing: s_memtime around `iters` dependent passes (next
// state fed back, like a search path), one 256-thread workgroup per CU, 256 workgroups = the 4096-root launch.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans tools/ubench_netpass.hip -o tools/bin/ubench_netpass
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../muax_amd/csrc/mz_fused.cuh"

#pragma clang fp contract(off)

using namespace mz;
using Cfg = FusedCfg<2, 8, 2, 51, 1, 4>;  // CartPole shapes (BASELINE configs[1])
constexpr int A = Cfg::A, E = Cfg::E, F = 21, H = kHidden, SUPPORT = 10;

struct Out {
  float reward, value, pil[A], pprob[A], ns[E];
};
struct BenchParams {
  FusedParams fp;
  const float* s0;  // [B][E] initial embeddings (min-max normalised)
  Out* out;         // [B] outputs of the LAST pass
  uint32_t* chk;    // [B] xor checksum over all passes
  uint64_t* cycles; // [waves]
  int iters;
};

// ------------------------------------------------------------------------------------------------ variant A
__global__ __launch_bounds__(256, 1) void netpass_rows(const BenchParams p) {
  asm volatile("; keep AGPRs allocatable" ::: "a0");
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
  const int r = blockIdx.x * 16 + (tid >> 6) * 4 + (lane >> 4);
  Nets<Cfg> nets;
  nets.load(p.fp, j);
  float s[Cfg::ES] = {j < E ? p.s0[(size_t)r * E + j] : 0.0f};
  int action = r % A;
  uint32_t chk = 0;
  float reward = 0, value = 0, pil = 0, pprob = 0, ns[Cfg::ES] = {0};
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < p.iters; ++it) {
    nets.forward(s, action, j, SUPPORT, F, false, reward, value, pil, pprob, ns);
    chk = (chk << 1 | chk >> 31) ^ f2u(reward) ^ (f2u(value) * 3u) ^ f2u(bcast<0>(pil)) ^ (f2u(bcast<1>(pprob)) * 5u);
    s[0] = ns[0];
    action = (action + 1 + (it & 1)) % A;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) p.cycles[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
  if (j == 0) {
    p.out[r].reward = reward;
    p.out[r].value = value;
    p.chk[r] = chk;
  }
  if (j < A) {
    p.out[r].pil[j] = pil;
    p.out[r].pprob[j] = pprob;
  }
  if (j < E) p.out[r].ns[j] = ns[0];
}

// ------------------------------------------------------------------------------------------------ variant B
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4x4 transpose over (lane row g = lane >> 4, register r): out[r] of row g = in[g] of row r
MZ_DEV void transpose4(float (&v)[4]) {
  unsigned a0 = f2u(v[0]), a1 = f2u(v[1]), a2 = f2u(v[2]), a3 = f2u(v[3]);
  auto s02 = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);  // a0 rows {2,3} <-> a2 rows {0,1}
  auto s13 = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
  auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);  // odd rows of vdst <-> even rows of src
  auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
  v[0] = u2f(s01[0]); v[1] = u2f(s01[1]); v[2] = u2f(s23[0]); v[3] = u2f(s23[1]);
}
// value of the lane 16 / 32 lanes away (same column n), through the swap instructions (no LDS)
MZ_DEV float xor16(float x) {
  auto s = __builtin_amdgcn_permlane16_swap(f2u(x), f2u(x), false, false);
  return u2f((threadIdx.x & 16) ? s[0] : s[1]);  // row g odd: vdst now holds row g-1's; even: src holds row g+1's
}
MZ_DEV float xor32(float x) {
  auto s = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
  return u2f((threadIdx.x & 32) ? s[0] : s[1]);
}
// A operand of one k-step of y = x W (+ zero rows / columns outside the matrix): lane (kk, i) holds W[4 t + kk][col0 + i]
MZ_DEV float a_operand(const float* __restrict__ W, int K, int NOUT, int t, int col0, int lane) {
  const int k = 4 * t + (lane >> 4), c = col0 + (lane & 15);
  return (k < K && c < NOUT) ? W[k * NOUT + c] : 0.0f;
}
// bias of C-layout register r of this lane: row 4 g + r of tile `col0`
MZ_DEV float c_bias(const float* __restrict__ Bv, int NOUT, int col0, int r, int lane) {
  const int i = col0 + 4 * (lane >> 4) + r;
  return i < NOUT ? Bv[i] : 0.0f;
}
template <int NT>
struct Layer {  // NT k-steps, one 16-row output tile
  float a[NT];
  float b[4];
  MZ_DEV void load(const float* W, const float* Bv, int K, int NOUT, int col0, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = a_operand(W, K, NOUT, t, col0, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = c_bias(Bv, NOUT, col0, r, lane);
  }
  MZ_DEV void apply(const float (&x)[NT], float (&y)[4]) const {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NT; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], x[t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc[r] + b[r];
  }
};
// first layer of Dynamic: the chain over the E state terms, then (h + W[E + a]) + b -- as one more k-step whose B
// operand is the one-hot (fma(1, w, h) = h + w, fma(0, w, h) = h), then the bias
template <int NT>
struct LayerNoBiasFirst {
  float a[NT];
  float b[4];
  MZ_DEV void load(const float* W, const float* Bv, int K, int NOUT, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = a_operand(W, K, NOUT, t, 0, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = c_bias(Bv, NOUT, 0, r, lane);
  }
  MZ_DEV void apply(const float (&x)[NT], float (&y)[4]) const {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NT; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], x[t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc[r] + b[r];
  }
};
// support_to_scalar(softmax(.)) of F = 21 logits held as two C-layout tiles (rows 0..15, 16..31), per column
MZ_DEV float c_decode(const float (&x0)[4], const float (&x1)[4], int lane) {
  const int g = lane >> 4;
  bool ok1[4];
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ok1[r] = 16 + 4 * g + r < F;
    m = fmaxf(m, x0[r]);
    m = ok1[r] ? fmaxf(m, x1[r]) : m;
  }
  m = fmaxf(m, xor16(m));
  m = fmaxf(m, xor32(m));
  float e0[4], e1[4], part[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e0[r] = exp_neg(x0[r] - m);
    e1[r] = ok1[r] ? exp_neg(x1[r] - m) : 0.0f;
    part[r] = ok1[r] ? e0[r] + e1[r] : e0[r];
  }
  float s = (part[0] + part[1]) + (part[2] + part[3]);  // butterfly steps xor 1, xor 2 (index = 4 g + r)
  s = s + xor16(s);                                     // xor 4
  s = s + xor32(s);                                     // xor 8
  float tp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float p0 = e0[r] / s, p1 = e1[r] / s;
    const float t0 = (float)(4 * g + r - SUPPORT) * p0, t1 = (float)(16 + 4 * g + r - SUPPORT) * p1;
    tp[r] = ok1[r] ? t0 + t1 : t0;
  }
  float xs = (tp[0] + tp[1]) + (tp[2] + tp[3]);
  xs = xs + xor16(xs);
  xs = xs + xor32(xs);
  return inv_scaling(xs);
}

constexpr int XS = 13;  // words per root of the staged input (odd: 16 roots in 16 banks)
struct Stage {
  float xin[16 * XS];   // [root][k]: state (E), one-hot action (A), zero padding
  float ns[16 * 8];     // outputs back to the row owners
  float rew[16], val[16], pil[16 * A], ppr[16 * A];
};

template <bool SKEL>
__global__ __launch_bounds__(256, 1) void netpass_mfma(const BenchParams p) {
  __shared__ Stage st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int root_in_wg = wave * 4 + g;
  const int r = blockIdx.x * 16 + root_in_wg;
  const FusedParams& w = p.fp;
  // role weights (A operands), loaded once
  LayerNoBiasFirst<3> d1;   // dynamics layer 1 of this role's net (K = E + A = 10 -> 3 k-steps)
  Layer<4> l2a, l2b;        // second layer: tile 0 (+ tile 1 for the 21 support logits)
  Layer<2> p1;              // prediction layer 1 of this role's head (K = E = 8)
  Layer<4> q2a, q2b;        // prediction layer 2
  Layer<4> n2;              // next-state layer 2 (roles 1, 2)
  if (wave == 0) {
    d1.load(w.dr_w1, w.dr_b1, E + A, H, lane);
    l2a.load(w.dr_w2, w.dr_b2, H, F, 0, lane);
    l2b.load(w.dr_w2, w.dr_b2, H, F, 16, lane);
  } else {
    d1.load(w.dn_w1, w.dn_b1, E + A, H, lane);
    n2.load(w.dn_w2, w.dn_b2, H, E, 0, lane);
    if (wave == 1) {
      p1.load(w.pv_w1, w.pv_b1, E, H, 0, lane);
      q2a.load(w.pv_w2, w.pv_b2, H, F, 0, lane);
      q2b.load(w.pv_w2, w.pv_b2, H, F, 16, lane);
    } else {
      p1.load(w.pp_w1, w.pp_b1, E, H, 0, lane);
      q2a.load(w.pp_w2, w.pp_b2, H, A, 0, lane);
    }
  }
  float s = j < E ? p.s0[(size_t)r * E + j] : 0.0f;
  int action = r % A;
  uint32_t chk = 0;
  float reward = 0, value = 0, pil = 0, pprob = 0, ns = 0;
  for (int i = tid; i < (int)(sizeof(Stage) / 4); i += 256) reinterpret_cast<float*>(&st)[i] = 0.0f;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < p.iters; ++it) {
    // row owners -> stage
    if (j < E + A) st.xin[root_in_wg * XS + j] = j < E ? s : (j - E == action ? 1.0f : 0.0f);
    __syncthreads();
    if (wave < 3) {
      float x[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) x[t] = st.xin[j * XS + 4 * t + g];  // B operand: lane (kk = g, n = j)
      float h[4];
      d1.apply(x, h);
#pragma unroll
      for (int q = 0; q < 4; ++q) h[q] = SKEL ? h[q] : elu(h[q]);
      transpose4(h);
      if (wave == 0) {
        float y0[4], y1[4];
        l2a.apply(h, y0);
        l2b.apply(h, y1);
        const float rw = SKEL ? (y0[0] + y1[1]) : c_decode(y0, y1, lane);
        if (g == 0) st.rew[j] = rw;
      } else {
        float y[4];
        n2.apply(h, y);
        // min_max_normalize over rows 0..E-1 (lane rows g = 0, 1)
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = 4 * g + q < E;
          mn = ok ? fminf(mn, y[q]) : mn;
          mx = ok ? fmaxf(mx, y[q]) : mx;
        }
        mn = fminf(mn, xor16(mn));
        mx = fmaxf(mx, xor16(mx));
        float scale = mx - mn;
        scale = scale < 1e-5f ? scale + 1e-5f : scale;
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = SKEL ? y[q] - mn : (y[q] - mn) / scale;
        if (wave == 1 && g < 2) *reinterpret_cast<f32x4*>(&st.ns[j * 8 + 4 * g]) = (f32x4){y[0], y[1], y[2], y[3]};
        transpose4(y);  // y[t] of lane (kk, n) = ns_n[4 t + kk]
        float xp[2] = {y[0], y[1]};
        float hp[4];
        p1.apply(xp, hp);
#pragma unroll
        for (int q = 0; q < 4; ++q) hp[q] = SKEL ? hp[q] : elu(hp[q]);
        transpose4(hp);
        if (wave == 1) {
          float y0[4], y1[4];
          q2a.apply(hp, y0);
          q2b.apply(hp, y1);
          const float vl = SKEL ? (y0[0] + y1[1]) : c_decode(y0, y1, lane);
          if (g == 0) st.val[j] = vl;
        } else {
          float lg[4];
          q2a.apply(hp, lg);
          // softmax over the A = 2 policy logits (rows 0, 1: lane row 0, registers 0, 1)
          const float m = fmaxf(lg[0], lg[1]);
          const float e0 = SKEL ? lg[0] - m : exp_neg(lg[0] - m), e1 = SKEL ? lg[1] - m : exp_neg(lg[1] - m);
          const float sm = e0 + e1;
          if (g == 0) {
            st.pil[j * A + 0] = lg[0]; st.pil[j * A + 1] = lg[1];
            st.ppr[j * A + 0] = SKEL ? e0 + sm : e0 / sm; st.ppr[j * A + 1] = SKEL ? e1 + sm : e1 / sm;
          }
        }
      }
    }
    __syncthreads();
    // stage -> row owners
    reward = st.rew[root_in_wg];
    value = st.val[root_in_wg];
    pil = j < A ? st.pil[root_in_wg * A + j] : 0.0f;
    pprob = j < A ? st.ppr[root_in_wg * A + j] : 0.0f;
    ns = j < E ? st.ns[root_in_wg * 8 + j] : 0.0f;
    chk = (chk << 1 | chk >> 31) ^ f2u(reward) ^ (f2u(value) * 3u) ^ f2u(bcast<0>(pil)) ^ (f2u(bcast<1>(pprob)) * 5u);
    s = ns;
    action = (action + 1 + (it & 1)) % A;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) p.cycles[blockIdx.x * 4 + wave] = t1 - t0;
  if (j == 0) {
    p.out[r].reward = reward;
    p.out[r].value = value;
    p.chk[r] = chk;
  }
  if (j < A) {
    p.out[r].pil[j] = pil;
    p.out[r].pprob[j] = pprob;
  }
  if (j < E) p.out[r].ns[j] = ns;
}

// ------------------------------------------------------------------------------------------------ host
#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50, WG = argc > 2 ? atoi(argv[2]) : 256, B = 16 * WG;
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
  Out* out[2];
  uint32_t* chk[2];
  uint64_t* cyc[2];
  for (int v = 0; v < 2; ++v) {
    CK(hipMalloc(&out[v], sizeof(Out) * B));
    CK(hipMalloc(&chk[v], 4 * B));
    CK(hipMalloc(&cyc[v], 8 * WG * 4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms[3] = {0, 0, 0};
  double cycles_mean[3], cycles_max[3];
  for (int v : {2, 0, 1}) {
    p.out = out[v & 1]; p.chk = chk[v & 1]; p.cycles = cyc[v & 1]; p.iters = iters;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      if (v == 0) hipLaunchKernelGGL(netpass_rows, dim3(WG), dim3(256), 0, 0, p);
      else if (v == 2) hipLaunchKernelGGL(netpass_mfma<true>, dim3(WG), dim3(256), 0, 0, p);
      else hipLaunchKernelGGL(netpass_mfma<false>, dim3(WG), dim3(256), 0, 0, p);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      CK(hipEventElapsedTime(&ms[v], e0, e1));
    }
    std::vector<uint64_t> c(WG * 4);
    CK(hipMemcpy(c.data(), cyc[v & 1], 8 * WG * 4, hipMemcpyDeviceToHost));
    double sum = 0, mxv = 0;
    for (auto x : c) { sum += (double)x; mxv = mxv > (double)x ? mxv : (double)x; }
    cycles_mean[v] = sum / c.size() / iters;
    cycles_max[v] = mxv / iters;
  }
  std::vector<Out> ho[2];
  std::vector<uint32_t> hc[2];
  for (int v = 0; v < 2; ++v) {
    ho[v].resize(B); hc[v].resize(B);
    CK(hipMemcpy(ho[v].data(), out[v], sizeof(Out) * B, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc[v].data(), chk[v], 4 * B, hipMemcpyDeviceToHost));
  }
  size_t bad_out = 0, bad_chk = 0;
  for (int r = 0; r < B; ++r) {
    if (memcmp(&ho[0][r], &ho[1][r], sizeof(Out)) != 0) ++bad_out;
    if (hc[0][r] != hc[1][r]) ++bad_chk;
  }
  printf("network pass, CartPole shapes (A=2 E=8 F=21 H=16), %d workgroups x 16 roots, %d dependent passes\n", WG, iters);
  printf("  A row-distributed pk-fma chains (product): %8.1f cycles/pass (slowest wave %8.1f), kernel %.1f us\n",
         cycles_mean[0], cycles_max[0], ms[0] * 1e3);
  printf("  B workgroup-cooperative fp32 MFMA        : %8.1f cycles/pass (slowest wave %8.1f), kernel %.1f us\n",
         cycles_mean[1], cycles_max[1], ms[1] * 1e3);
  printf("  B' the same without its elementwise work : %8.1f cycles/pass (slowest wave %8.1f), kernel %.1f us   (MFMAs, transposes, LDS hand-over, 2 barriers only)\n",
         cycles_mean[2], cycles_max[2], ms[2] * 1e3);
  printf("  bitwise agreement: %zu / %d roots differ in the last pass' outputs, %zu / %d in the all-pass checksum\n",
         bad_out, B, bad_chk, B);
  printf("  sample root 0: reward %.9g value %.9g ns0 %.9g | B: reward %.9g value %.9g ns0 %.9g\n", ho[0][0].reward,
         ho[0][0].value, ho[0][0].ns[0], ho[1][0].reward, ho[1][0].value, ho[1][0].ns[0]);
  return (bad_out || bad_chk) ? 2 : 0;
}
