// ubench_lat.hip -- latency / issue cost of the instruction classes of the fused act() kernel's network pass for ONE
// wavefront per SIMD on gfx950 (s_memtime ticks per instruction; 256 workgroups x 256 threads).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lat.hip -o tools/bin/ubench_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

typedef float f32x2 __attribute__((ext_vector_type(2)));

// each kernel: `iters` x 64 copies of BODY; reports ticks via s_memtime
#define KERNEL(NAME, DECL, BODY, SINK)                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, uint64_t* ticks, int iters, float a, float b) { \
    DECL;                                                                           \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int it = 0; it < iters; ++it) { REP64(BODY) }                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = SINK;                              \
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                    \
  }

#define X0 float x0 = threadIdx.x * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7
#define P0 f32x2 p0 = {threadIdx.x * 0.001f, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, w = {a, b}, c = {b, a}

KERNEL(k_fma_dep, X0, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));, x0)
KERNEL(k_fma_ilp2, X0, asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));, x0 + x1)
KERNEL(k_fma_ilp4, X0, asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));, x0 + x1 + x2 + x3)
KERNEL(k_fma_ilp8, X0, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));, x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
KERNEL(k_pk_dep, P0, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(w), "v"(c));, p0.x + p0.y)
KERNEL(k_pk_ilp2, P0, asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(w), "v"(c));, p0.x + p1.y)
KERNEL(k_pk_ilp4, P0, asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w), "v"(c));, p0.x + p1.y + p2.x + p3.y)
// the compiler pads one wait state between dependent packed ops (LLVM reads op_sel_hi of src0 as a partial-dword write):
// what does the pad cost on a dependent chain, and between independent ones?
KERNEL(k_pk_dep_nop, P0, asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(p0) : "v"(w), "v"(c));, p0.x + p0.y)
KERNEL(k_pk_ilp2_nop, P0, asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n s_nop 0\n v_pk_fma_f32 %1, %1, %2, %3\n s_nop 0" : "+v"(p0), "+v"(p1) : "v"(w), "v"(c));, p0.x + p1.y)
// the discounted-return step of the backup: s_nop 1, v_mul_f32_dpp row_shl:1, v_add_f32
KERNEL(k_gstep, X0, asm volatile("s_nop 1\n v_mul_f32_dpp %1, %0, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %1, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));, x0 + x1)
// ... with the two wait states filled by independent VALU work instead of s_nop 1
KERNEL(k_gstep_fill, X0, asm volatile("v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_mul_f32_dpp %1, %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %1, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));, x0 + x1 + x2 + x3)
// one pk chain + one scalar chain (the second-layer loops of the network pass)
KERNEL(k_pk_fma_mix, P0; float x0 = a, asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_fmac_f32 %1, %4, %5" : "+v"(p0), "+v"(x0) : "v"(w), "v"(c), "v"(a), "v"(b));, p0.x + x0)
// DPP broadcast movs, independent (issue cost)
KERNEL(k_movdpp_ilp4, X0, asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4));, x0 + x1 + x2 + x3)
// mov_dpp -> fmac (acc chain), source fixed: what the kernel does today per term when the movs are not hoisted
KERNEL(k_movdpp_fmac, X0, asm volatile("v_mov_b32_dpp %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %0, %1, %3" : "+v"(x0), "+v"(x1) : "v"(x4), "v"(a));, x0 + x1)
// the fused form
KERNEL(k_fmac_dpp, X0, asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(x0) : "v"(x4), "v"(a));, x0)
KERNEL(k_fmac_dpp_ilp2, X0, asm volatile("v_fmac_f32_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1) : "v"(x4), "v"(a));, x0 + x1)
// fused scalar chain next to a pk chain fed by movs: the proposed second-layer loop
KERNEL(k_mix_fused, P0; float x0 = a; float x4 = a + b, asm volatile("v_mov_b32_dpp v100, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_pk_fma_f32 %0, v[100:101], %3, %0 op_sel_hi:[0,1,1]\n v_fmac_f32_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(p0), "+v"(x0) : "v"(x4), "v"(w), "v"(a) : "v100", "v101");, p0.x + x0)
// today's loop: two movs, pk_fma, fmac
KERNEL(k_mix_today, P0; float x0 = a; float x4 = a + b, asm volatile("v_mov_b32_dpp v100, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v102, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_pk_fma_f32 %0, v[100:101], %3, %0 op_sel_hi:[0,1,1]\n v_fmac_f32 %1, v102, %4" : "+v"(p0), "+v"(x0) : "v"(x4), "v"(w), "v"(a) : "v100", "v101", "v102");, p0.x + x0)
// dependent DPP butterfly step (v_add_f32_dpp reading its own result)
KERNEL(k_adddpp_dep, X0, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x0));, x0)
KERNEL(k_adddpp_ilp2, X0, asm volatile("s_nop 0\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1));, x0 + x1)
// compare -> select chain through VCC
KERNEL(k_cmp_sel, X0, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x0) : "v"(a), "v"(b) : "vcc");, x0)
KERNEL(k_cmp_sel_sgpr, X0, asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(x0) : "v"(a), "v"(b) : "s20", "s21");, x0)
// transcendental and conversion ops, dependent
KERNEL(k_rcp_dep, X0, asm volatile("v_rcp_f32 %0, %0" : "+v"(x0));, x0)
KERNEL(k_rcp_ilp4, X0, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));, x0 + x1 + x2 + x3)
KERNEL(k_sqrt_dep, X0, asm volatile("v_sqrt_f32 %0, %0" : "+v"(x0));, x0)
KERNEL(k_rndne_dep, X0, asm volatile("v_rndne_f32 %0, %0" : "+v"(x0));, x0)
KERNEL(k_cvt_dep, X0, asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0" : "+v"(x0));, x0)
KERNEL(k_divscale_dep, X0, asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x0) : "v"(a) : "vcc");, x0)
KERNEL(k_divfmas_dep, X0, asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b) : "vcc");, x0)
KERNEL(k_divfixup_dep, X0, asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));, x0)
KERNEL(k_pkmul_dep, P0, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(w));, p0.x + p0.y)
KERNEL(k_pkadd_dep, P0, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(w));, p0.x + p0.y)
KERNEL(k_lshladd_dep, X0; int i0 = threadIdx.x, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(i0) : "v"(7));, (float)i0)
// whole IEEE division, dependent (compiler expansion)
__global__ __launch_bounds__(256) void k_div_dep(float* out, uint64_t* ticks, int iters, float a, float b) {
  float x = threadIdx.x * 0.001f + 1.0f;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64; ++r) x = a / x;
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* out;
  uint64_t* ticks;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&ticks, 8);
  const int iters = 2000;
#define RUN(K, NINSTR, NOTE)                                                                      \
  {                                                                                               \
    hipLaunchKernelGGL(K, dim3(256), dim3(256), 0, 0, out, ticks, iters, 1.0001f, 0.5f);          \
    hipLaunchKernelGGL(K, dim3(256), dim3(256), 0, 0, out, ticks, iters, 1.0001f, 0.5f);          \
    uint64_t t;                                                                                   \
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);                                               \
    printf("%-18s %7.2f ticks per group, %6.2f per instruction   %s\n", #K, (double)t / (iters * 64.0), \
           (double)t / (iters * 64.0 * NINSTR), NOTE);                                            \
  }
  RUN(k_fma_dep, 1, "v_fma_f32 dependent")
  RUN(k_fma_ilp2, 2, "")
  RUN(k_fma_ilp4, 4, "")
  RUN(k_fma_ilp8, 8, "")
  RUN(k_pk_dep, 1, "v_pk_fma_f32 dependent")
  RUN(k_pk_ilp2, 2, "")
  RUN(k_pk_ilp4, 4, "")
  RUN(k_pk_dep_nop, 1, "dependent v_pk_fma_f32 + s_nop 0 (per pair)")
  RUN(k_pk_ilp2_nop, 2, "two chains, each op followed by s_nop 0 (per op + nop)")
  RUN(k_gstep, 1, "s_nop 1 + v_mul_f32_dpp + v_add_f32 (per step)")
  RUN(k_gstep_fill, 1, "two independent fmas instead of the s_nop (per step)")
  RUN(k_pk_fma_mix, 2, "pk chain + fmac chain")
  RUN(k_movdpp_ilp4, 4, "independent row_newbcast movs")
  RUN(k_movdpp_fmac, 2, "mov_dpp -> fmac chain")
  RUN(k_fmac_dpp, 1, "v_fmac_f32_dpp chain")
  RUN(k_fmac_dpp_ilp2, 2, "")
  RUN(k_mix_fused, 3, "mov + pk_fma + fmac_dpp per term")
  RUN(k_mix_today, 4, "2 mov + pk_fma + fmac per term")
  RUN(k_adddpp_dep, 1, "s_nop 1 + v_add_f32_dpp dependent")
  RUN(k_adddpp_ilp2, 2, "")
  RUN(k_cmp_sel, 2, "v_cmp -> vcc -> v_cndmask dependent")
  RUN(k_cmp_sel_sgpr, 2, "v_cmp -> sgpr pair -> v_cndmask")
  RUN(k_rcp_dep, 1, "")
  RUN(k_rcp_ilp4, 4, "")
  RUN(k_sqrt_dep, 1, "")
  RUN(k_rndne_dep, 1, "")
  RUN(k_cvt_dep, 2, "")
  RUN(k_divscale_dep, 1, "")
  RUN(k_divfmas_dep, 1, "")
  RUN(k_divfixup_dep, 1, "")
  RUN(k_pkmul_dep, 1, "")
  RUN(k_pkadd_dep, 1, "")
  RUN(k_lshladd_dep, 1, "")
  RUN(k_div_dep, 1, "a / x dependent (IEEE expansion)")
  return 0;
}
