// mz_norm.cuh -- hk.LayerNorm over a whole sample + what follows it in the reference's convolutional nets, fused:
//   y = [relu]( LN(x) [+ LN2(x2)] [+ residual] ),   LN(x) = (x - mean) * rsqrt(var + eps) * scale[c] + offset[c]
// (muax/nn.py:118-148 ResidualConvBlockV1: conv - LN - relu - conv - LN, + (projected: conv - LN) shortcut, relu;
// :151-178 V2; :232-288 the EZ heads; hk.LayerNorm(axis=(-3,-2,-1), create_scale=True, create_offset=True), eps 1e-5,
// biased variance).  Between the convolutions of the plugin nets' ROOT inference these chains were ~10 small
// framework kernels each (mean, Welford variance, sub, rsqrt, mul, mul, add, add, relu ...), ~90 us per
// convolution of 48 us at config 4's shapes; here they are two launches, both bandwidth-bound:
//   1. moments: grid (samples, chunks, tensors); per-thread fp64 sum / sum of squares over 16-byte loads, wave
//      shuffle + LDS reduction, one (sum, sumsq) pair per chunk;
//   2. apply: every thread adds up its sample's few chunk pairs (fp64: the moments are exact to fp32 rounding,
//      independent of the chunking), then streams 16-byte loads / stores with the reference's op order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mz {

struct NormParams {
  const float* x; const float* scale; const float* offset;
  const float* x2; const float* scale2; const float* offset2;
  const float* residual;
  float* y;
  double* ws;  // [tensors][B][K][2]
  int B, n, C, K, chunk, relu;
  float eps;
};

constexpr int kNormThreads = 256;

// mean and 1 / sqrt(var + eps) of a sample from its K (sum, sum of squares) pairs
__device__ __forceinline__ void ln_stats(const double* ws, int K, int n, float eps, float& mean, float& rstd) {
  double s = 0.0, q = 0.0;
  for (int k = 0; k < K; ++k) { s += ws[2 * k]; q += ws[2 * k + 1]; }
  const double m = s / (double)n;
  double var = q / (double)n - m * m;
  var = var < 0.0 ? 0.0 : var;
  mean = (float)m;
  rstd = 1.0f / sqrtf((float)var + eps);
}


// (mz_repr.hip takes ln_stats only: the kernels below belong to mz_norm.hip's translation unit)
#ifndef MZ_NORM_STATS_ONLY

__global__ __launch_bounds__(kNormThreads) void ln_moments_kernel(NormParams p) {
  const int b = blockIdx.x, k = blockIdx.y, t = blockIdx.z;  // samples in grid x: no 65535 limit on the batch
  const float* src = (t == 0 ? p.x : p.x2) + (size_t)b * p.n;
  const int lo = k * p.chunk, hi = min(p.n, lo + p.chunk);
  double s = 0.0, q = 0.0;
  for (int i = lo + 4 * (int)threadIdx.x; i < hi; i += 4 * kNormThreads) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += __shfl_down(s, d);
    q += __shfl_down(q, d);
  }
  __shared__ double red[2 * (kNormThreads / 64)];
  if ((threadIdx.x & 63) == 0) {
    red[2 * (threadIdx.x >> 6)] = s;
    red[2 * (threadIdx.x >> 6) + 1] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int w = 0; w < kNormThreads / 64; ++w) { ts += red[2 * w]; tq += red[2 * w + 1]; }
    double* out = p.ws + (((size_t)t * p.B + b) * p.K + k) * 2;
    out[0] = ts;
    out[1] = tq;
  }
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(kNormThreads) void ln_apply_kernel(NormParams p) {
  const int b = blockIdx.x;  // samples in grid x (any batch), the sample's slices in grid y
  float mean, rstd, mean2 = 0.0f, rstd2 = 0.0f;
  ln_stats(p.ws + (size_t)b * p.K * 2, p.K, p.n, p.eps, mean, rstd);
  if (p.x2) ln_stats(p.ws + ((size_t)p.B + b) * p.K * 2, p.K, p.n, p.eps, mean2, rstd2);
  const size_t base = (size_t)b * p.n;
  for (int i = 4 * (int)(blockIdx.y * kNormThreads + threadIdx.x); i < p.n; i += 4 * kNormThreads * (int)gridDim.y) {
    const int c = i % p.C;
    const float4 v = *reinterpret_cast<const float4*>(p.x + base + i);
    const float4 g = *reinterpret_cast<const float4*>(p.scale + c);
    const float4 o = *reinterpret_cast<const float4*>(p.offset + c);
    float4 y;
    y.x = (v.x - mean) * rstd * g.x + o.x;
    y.y = (v.y - mean) * rstd * g.y + o.y;
    y.z = (v.z - mean) * rstd * g.z + o.z;
    y.w = (v.w - mean) * rstd * g.w + o.w;
    if (p.x2) {
      const float4 v2 = *reinterpret_cast<const float4*>(p.x2 + base + i);
      const float4 g2 = *reinterpret_cast<const float4*>(p.scale2 + c);
      const float4 o2 = *reinterpret_cast<const float4*>(p.offset2 + c);
      y.x = y.x + ((v2.x - mean2) * rstd2 * g2.x + o2.x);
      y.y = y.y + ((v2.y - mean2) * rstd2 * g2.y + o2.y);
      y.z = y.z + ((v2.z - mean2) * rstd2 * g2.z + o2.z);
      y.w = y.w + ((v2.w - mean2) * rstd2 * g2.w + o2.w);
    }
    if (p.residual) {
      const float4 r = *reinterpret_cast<const float4*>(p.residual + base + i);
      y.x = y.x + r.x; y.y = y.y + r.y; y.z = y.z + r.z; y.w = y.w + r.w;
    }
    if (p.relu) {
      y.x = fmaxf(y.x, 0.0f); y.y = fmaxf(y.y, 0.0f); y.z = fmaxf(y.z, 0.0f); y.w = fmaxf(y.w, 0.0f);
    }
    *reinterpret_cast<float4*>(p.y + base + i) = y;
  }
}

#endif  // MZ_NORM_STATS_ONLY

inline int norm_chunks(int n) {
  int k = n / 4096;
  return k < 1 ? 1 : (k > 16 ? 16 : k);
}

}  // namespace mz
