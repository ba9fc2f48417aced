// Dependent-issue latency of single instructions on one lone wavefront (developer microbenchmark, gfx950):
//   hipcc --offload-arch=gfx950 -O3 tools/inst_latency.hip -o build/lat && ./build/lat
// Results of round 1 are quoted in profiles/r01_decode_bisect.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define CHAIN(NAME, ASM)                                                          \
  __global__ void k_##NAME(float* out, unsigned long long* cyc, float x0, float c) { \
    float x = x0 + threadIdx.x * 1e-6f;                                            \
    float y = c;                                                                   \
    unsigned long long t0 = __builtin_readcyclecounter();                          \
    _Pragma("unroll") for (int i = 0; i < REP; ++i) asm volatile(ASM : "+v"(x) : "v"(y)); \
    unsigned long long t1 = __builtin_readcyclecounter();                          \
    out[threadIdx.x] = x;                                                          \
    if (threadIdx.x == 0) *cyc = t1 - t0;                                          \
  }
CHAIN(fma, "v_fma_f32 %0, %0, %1, %1")
CHAIN(mul, "v_mul_f32 %0, %0, %1")
CHAIN(exp, "v_exp_f32 %0, %0")
CHAIN(rcp, "v_rcp_f32 %0, %0")
CHAIN(sqrt, "v_sqrt_f32 %0, %0")
CHAIN(rsq, "v_rsq_f32 %0, %0")
CHAIN(dpp_quad, "s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
CHAIN(dpp_bcast, "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CHAIN(readlane, "v_readlane_b32 s20, %0, 63\n s_nop 3\n v_mul_f32 %0, s20, %1")
CHAIN(fma_then_exp, "v_mul_f32 %0, %0, %1\n v_exp_f32 %0, %0")
CHAIN(sqrt_add_rcp, "v_sqrt_f32 %0, %0\n v_add_f32 %0, %0, %1\n v_rcp_f32 %0, %0")
__global__ void k_pkfma(float* out, unsigned long long* cyc, float x0, float c) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 x = {x0 + threadIdx.x * 1e-6f, x0}; v2 y = {c, c};
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < REP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x.x + x.y;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
// independent streams: 4 interleaved chains of fma (issue rate)
__global__ void k_fma4(float* out, unsigned long long* cyc, float x0, float c) {
  float a = x0, b = x0 + 1, d = x0 + 2, e = x0 + 3, y = c;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < REP; ++i) asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(d), "+v"(e) : "v"(y));
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + b + d + e;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_rcp4(float* out, unsigned long long* cyc, float x0, float c) {
  float a = x0, b = x0 + 1, d = x0 + 2, e = x0 + 3;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < REP; ++i) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(d), "+v"(e));
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + b + d + e;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
// cross-row sum + broadcast after four in-row DPP stages: (a) two row_bcast stages + v_readlane + first use,
// (b) one v_mfma_f32_16x16x4_f32 with A = 1 (sums the four 16-lane rows into every lane) + first use
__global__ void k_tail_dpp(float* out, unsigned long long* cyc, float x0, float c) {
  float x = x0 + threadIdx.x * 1e-6f, y = c;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < REP; ++i)
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 0\n"
                 "v_readlane_b32 s20, %0, 63\n s_nop 3\n v_mul_f32 %0, s20, %1" : "+v"(x) : "v"(y) : "s20");
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_tail_mfma(float* out, unsigned long long* cyc, float x0, float c) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  float x = x0 + threadIdx.x * 1e-6f, y = c;
  asm volatile("" : "+v"(x));
  unsigned long long t0 = __builtin_readcyclecounter();
  asm volatile("" : "+v"(x));
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, x, acc, 0, 0, 0);
    x = acc[0] * y;
    asm volatile("" : "+v"(x));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
#define RUN(NAME, N) do { k_##NAME<<<1, 64>>>(out, cyc, 1.0001f, 0.999f); hipDeviceSynchronize(); k_##NAME<<<1, 64>>>(out, cyc, 1.0001f, 0.999f); hipDeviceSynchronize(); \
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-14s %7.1f cycles per step (%d instr/step)\n", #NAME, (double)h / REP, N); } while (0)
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256); hipMalloc(&cyc, 8);
  RUN(fma, 1); RUN(mul, 1); RUN(pkfma, 1); RUN(exp, 1); RUN(rcp, 1); RUN(sqrt, 1); RUN(rsq, 1); RUN(dpp_quad, 1); RUN(dpp_bcast, 1);
  RUN(readlane, 2); RUN(fma_then_exp, 2); RUN(sqrt_add_rcp, 3); RUN(fma4, 4); RUN(rcp4, 4); RUN(tail_dpp, 4); RUN(tail_mfma, 2);
  return 0;
}
