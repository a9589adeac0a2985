// ubench_issue.hip -- how fast does ONE wave per SIMD issue VALU work on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o gpurun_out/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void k_fma(float* out, int iters, float a, float b) {
  float x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ __launch_bounds__(256) void k_dpp(float* out, int iters, float a) {
  float x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i)
        x[i] = x[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0xB1, 0xf, 0xf, false)) * a;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  __shared__ int buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = (i * 37 + 11) & 4095;
  __syncthreads();
  int p = threadIdx.x;
  for (int it = 0; it < iters * 16; ++it) p = buf[p];
  out[blockIdx.x * blockDim.x + threadIdx.x] = p;
}
template <class F>
float time_it(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4 * 4);
  const int iters = 20000;
  const double ghz = 2.4;
#define RUN(NAME, KERNEL, BLOCKS, THREADS, NINSTR)                                     \
  {                                                                                    \
    float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(BLOCKS), dim3(THREADS), 0, 0, out, iters, 1.0001f, 0.5f); }); \
    printf("%-40s %8.3f ms  %6.2f cycles/instr/wave @2.4GHz\n", NAME, ms, ms * 1e-3 * ghz * 1e9 / ((double)iters * 16 * NINSTR)); \
  }
  RUN("fma dep chain, 1 wave/SIMD", (k_fma<1, 1>), 256, 256, 1)
  RUN("fma ILP2, 1 wave/SIMD", (k_fma<2, 1>), 256, 256, 2)
  RUN("fma ILP4, 1 wave/SIMD", (k_fma<4, 1>), 256, 256, 4)
  RUN("fma ILP8, 1 wave/SIMD", (k_fma<8, 1>), 256, 256, 8)
  RUN("fma dep chain, 2 waves/SIMD", (k_fma<1, 2>), 256, 512, 1)
  RUN("fma ILP4, 2 waves/SIMD", (k_fma<4, 2>), 256, 512, 4)
  RUN("fma ILP8, 2 waves/SIMD (per wave)", (k_fma<8, 2>), 256, 512, 8)
  RUN("fma ILP4, 4 waves/SIMD (per wave)", (k_fma<4, 4>), 256, 1024, 4)
#undef RUN
#define RUN2(NAME, KERNEL, NINSTR)                                                     \
  {                                                                                    \
    float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(256), dim3(256), 0, 0, out, iters, 1.0001f); }); \
    printf("%-40s %8.3f ms  %6.2f cycles/(dpp+fma pair)/wave\n", NAME, ms, ms * 1e-3 * ghz * 1e9 / ((double)iters * 16 * NINSTR)); \
  }
  RUN2("dpp mov + fma dep chain", (k_dpp<1>), 1)
  RUN2("dpp mov + fma ILP4", (k_dpp<4>), 4)
  {
    float ms = time_it([&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, 0, out, iters); });
    printf("%-40s %8.3f ms  %6.2f cycles per dependent ds_read_b32\n", "lds pointer chase", ms, ms * 1e-3 * ghz * 1e9 / ((double)iters * 16));
  }
  return 0;
}
