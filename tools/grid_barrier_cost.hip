// Developer measurement (round 4, review item 8): what does a device-wide barrier inside ONE launch cost on MI355X, against the
// launch boundary and the cross-stream event it would replace?
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_cost.hip -o /tmp/gbc && /tmp/gbc
// (a) N dependent launches of a kernel that touches `bytes` per workgroup (same stream): per-launch time;
// (b) one launch, 256 workgroups, N phases separated by a sense-reversing barrier on one global counter, each phase touching the
//     same bytes and publishing them with an agent-scope release / acquire (what a real phase hand-off across the 8 XCDs needs);
// (c) two streams ping-ponging through events: per hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)

__global__ void phase_kernel(float* buf, size_t per_wg, float v) {
  float* p = buf + (size_t)blockIdx.x * per_wg;
  for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) p[i] = p[i] * 0.5f + v;
}

__global__ void phased_kernel(float* buf, size_t per_wg, int phases, unsigned* counter, unsigned* gen) {
  // every phase reads the NEXT workgroup's slice as the previous phase left it (a real cross-workgroup dependency) and writes its own
  const unsigned n = gridDim.x;
  for (int ph = 0; ph < phases; ++ph) {
    float* mine = buf + (size_t)blockIdx.x * per_wg;
    const float* other = buf + (size_t)((blockIdx.x + 1) % n) * per_wg;
    for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) mine[i] = __builtin_nontemporal_load(other + i) * 0.5f + 1.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);                       // agent scope: publish this workgroup's slice beyond its XCD's L2
      const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gen, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
  }
}

int main() {
  const int WG = 256, N = 200;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  unsigned* cnt; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
  for (size_t kb : {4, 64, 512}) {                                    // KiB touched per workgroup and phase
    const size_t per = kb * 1024 / 4;
    float* buf; CK(hipMalloc(&buf, per * 4 * WG)); CK(hipMemset(buf, 0, per * 4 * WG));
    float ms;
    for (int w = 0; w < 2; ++w) {                                     // (second pass is the timed one)
      CK(hipEventRecord(a, s1));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(phase_kernel, dim3(WG), dim3(256), 0, s1, buf, per, 1.f);
      CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b));
    }
    CK(hipEventElapsedTime(&ms, a, b));
    const double launches = 1e3 * ms / N;
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(a, s1));
      hipLaunchKernelGGL(phased_kernel, dim3(WG), dim3(256), 0, s1, buf, per, N, cnt, cnt + 1);
      CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b));
    }
    CK(hipEventElapsedTime(&ms, a, b));
    const double phased = 1e3 * ms / N;
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(a, s1));
      for (int i = 0; i < N; ++i) {                                   // s1 kernel -> event -> s2 kernel -> event -> s1 ...
        hipLaunchKernelGGL(phase_kernel, dim3(WG), dim3(256), 0, s1, buf, per, 1.f);
        CK(hipEventRecord(ev[2 * i], s1)); CK(hipStreamWaitEvent(s2, ev[2 * i], 0));
        hipLaunchKernelGGL(phase_kernel, dim3(WG), dim3(256), 0, s2, buf, per, 1.f);
        CK(hipEventRecord(ev[2 * i + 1], s2)); CK(hipStreamWaitEvent(s1, ev[2 * i + 1], 0));
      }
      CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b));
    }
    CK(hipEventElapsedTime(&ms, a, b));
    const double hop = 1e3 * ms / (2 * N);
    printf("%4zu KiB per workgroup and phase: dependent launches %.2f us each | one launch, grid barrier %.2f us per phase | two streams through events %.2f us per kernel\n",
           kb, launches, phased, hop);
    for (auto& e : ev) (void)hipEventDestroy(e);
    CK(hipFree(buf));
  }
  return 0;
}
