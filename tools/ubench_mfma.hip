// How fast does ONE wavefront per SIMD issue v_mfma_f32_16x16x4_f32 (three independent accumulators, no
// memory)?  The ceiling of mz_conv.cuh's inner loop.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2];
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  for (int wgs : {128, 256}) {
    const int iters = 20000;  // 240 K MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 12.0);
    printf("%d workgroups x 4 waves: %.2f ns per MFMA per wave = %.1f cycles at 2.4 GHz; %.1f TFLOP/s total\n", wgs, per,
           per * 2.4, wgs * 4 * 2048.0 / per / 1e3);
  }
  return 0;
}
