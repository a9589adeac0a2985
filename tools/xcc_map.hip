// Which XCD does workgroup i of a 1-D launch land on?  (pair mode of mz_conv.cuh wants the two halves of a
// root on the same XCD.)  hipcc --offload-arch=gfx950 -O2 tools/xcc_map.hip -o tools/bin/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(6164);          // hwreg(HW_REG_XCC_ID, 0, 4)
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID, all 32 bits
  }
  lds[threadIdx.x] = 1.0f;
  for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(10);  // keep every block resident for a while
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 64;
  unsigned* d;
  hipMalloc(&d, 8 * n);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65000);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 65000, 0, d);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%d:%u%s", i, h[2 * i], (i % 16 == 15) ? "\n" : " ");
  printf("\n");
  return 0;
}
