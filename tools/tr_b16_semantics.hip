// Developer aid: what ds_read_b64_tr_b16 (gfx950) hands each lane.  hipcc --offload-arch=gfx950 tools/tr_b16_semantics.hip -o /tmp/tr && /tmp/tr
// Result (MI355X): per 16-lane group, lane i supplies 4 contiguous 16-bit elements = M[i / 4][4 (i % 4) .. + 3] of a [4][16] block;
// lane j receives M[0..3][j].  Used by gemm_tn_bf16_kernel (cdae_full_kernels.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out, int mode) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t lane = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = lane * 8;                                  // lane i: elements 4i .. 4i+3
  else if (mode == 1) addr = (lane >> 2) * 256 + (lane & 3) * 8;   // [row = lane/4 (stride 128 el)][chunk = lane%4]
  else addr = (lane & 15) * 256 + (lane >> 4) * 8;                 // [row = lane%16][chunk = lane/16]
  uint32_t lo, hi;
  const uint32_t base = (uint32_t)(uintptr_t)lds;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&lo) : "v"(base + addr) : "memory");
  uint64_t v; asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
  out[lane * 4 + 0] = (uint32_t)(v & 0xFFFF); out[lane * 4 + 1] = (uint32_t)((v >> 16) & 0xFFFF);
  out[lane * 4 + 2] = (uint32_t)((v >> 32) & 0xFFFF); out[lane * 4 + 3] = (uint32_t)(v >> 48);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 4); uint32_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
