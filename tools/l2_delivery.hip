// Developer aid: what rate do the L2s deliver 1 KiB wave requests at, chip-wide?  (profiles/r05_gemm1_duo_anatomy.txt reads 9.3-9.5 TB/s
// out of three different kernels.)  Every workgroup streams over a region that all workgroups of ITS XCD share (block b runs on XCD
// b % 8: MI355X_MICROARCH.md), 16 bytes per lane and load, eight loads in flight per lane; region sizes below, at and above the 4 MiB L2.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_delivery.hip -o build/l2_delivery && build/l2_delivery
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(256) stream_kernel(const uint4* __restrict__ buf, size_t region_u4, int passes, uint4* __restrict__ sink) {
  const uint4* r = buf + (size_t)(blockIdx.x & 7u) * region_u4;
  const size_t stride = (size_t)256 * 8, per_xcd_blocks = gridDim.x / 8;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)(blockIdx.x >> 3) * stride + threadIdx.x; i + 7 * 256 < region_u4; i += per_xcd_blocks * stride) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r[i + (size_t)j * 256];
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
    }
  if (acc.x == 0x12345678u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

// random 1 KiB rows (the hidden-gradient gather's pattern): every wavefront gathers rows of ITS XCD's region, UN in flight, FMA'd like the gather
template <int UN>
__global__ void __launch_bounds__(256) gather_kernel(const float4* __restrict__ buf, uint32_t rows_per_xcd, int trips, float4* __restrict__ sink) {
  const float4* r = buf + (size_t)(blockIdx.x & 7u) * rows_per_xcd * 64;
  const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  uint32_t state = wave * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int t = 0; t < trips; ++t) {
    float4 v[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      state = state * 1664525u + 1013904223u;
      const uint32_t row = __builtin_amdgcn_readfirstlane((state >> 8) % rows_per_xcd);
      v[j] = r[(size_t)row * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  }
  if (acc.x == 1234.5f) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int UN>
void run_gather(const float4* buf, float4* sink, hipEvent_t a, hipEvent_t b) {
  for (int wg_per_cu : {1, 2, 4, 8})
    for (uint32_t rows : {1325u, 4096u}) {
      const int grid = 256 * wg_per_cu, trips = 4096 / UN / wg_per_cu * 4;
      gather_kernel<UN><<<grid, 256>>>(buf, rows, 2, sink);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(a));
      gather_kernel<UN><<<grid, 256>>>(buf, rows, trips, sink);
      CHK(hipEventRecord(b));
      CHK(hipEventSynchronize(b));
      float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
      const double bytes = (double)grid * 4.0 * trips * UN * 1024.0;
      std::printf("gather: rows in flight per wavefront %2d  workgroups per CU %d  rows per XCD %5u (%4.1f MiB)  %8.3f ms  %7.2f TB/s\n", UN, wg_per_cu, rows,
                  rows / 1024.0, ms, bytes / ms * 1e-9);
    }
}

int main() {
  const size_t sizes_kb[] = {512, 1024, 2048, 3072, 4096, 8192, 32768, 262144};
  const size_t max_bytes = 8ull * 262144 * 1024;
  uint4 *buf, *sink;
  CHK(hipMalloc(&buf, max_bytes));
  CHK(hipMalloc(&sink, 4096 * 256 * sizeof(uint4)));
  CHK(hipMemset(buf, 1, max_bytes));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int wg_per_cu : {1, 2, 4})
    for (size_t kb : sizes_kb) {
      const size_t region_u4 = kb * 1024 / 16;
      const int grid = 256 * wg_per_cu;
      const int passes = (int)std::max<size_t>(2, (16ull << 30) / (kb * 1024 * 8));      // ~16 GiB delivered per measurement
      stream_kernel<<<grid, 256>>>(buf, region_u4, 2, sink);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(a));
      stream_kernel<<<grid, 256>>>(buf, region_u4, passes, sink);
      CHK(hipEventRecord(b));
      CHK(hipEventSynchronize(b));
      float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
      const double bytes = 8.0 * (double)kb * 1024.0 * passes;               // every region is read once per pass by its XCD's blocks together
      std::printf("workgroups per CU %d  region per XCD %7zu KiB  passes %5d  %8.3f ms  %7.2f TB/s delivered to the CUs\n", wg_per_cu, kb, passes, ms, bytes / ms * 1e-9);
    }
  run_gather<4>((const float4*)buf, (float4*)sink, a, b);
  run_gather<8>((const float4*)buf, (float4*)sink, a, b);
  run_gather<16>((const float4*)buf, (float4*)sink, a, b);
  return 0;
}
