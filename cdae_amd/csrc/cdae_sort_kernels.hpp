// cdae_sort_kernels.hpp — item-major ordering of a batch's example list without a library sort.
//
// The decode is transposed (cdae_kernels.hpp): every item row walks the batch's examples on it in USER order, so the
// user-major example list that sample_kernel writes (cdae.hpp:217-220, 361-371) has to be turned item-major, stably.  A
// generic radix sort of (16-bit key, 64-bit value) pairs — rocPRIM onesweep: histogram, scan, two digit passes — cost
// ~70-85 us of mostly launch latency per batch on the prep stream, as much as the training kernels of a 256-user batch.
// The keys are item ids below 65 536 and the order wanted inside an item is the order of the (unique) example words, so a
// counting sort does it in three small launches:
//   sample_kernel         counts the examples of every item: atomicAdd(item_count[item], 1), no return value
//   count_scan_kernel     one workgroup: exclusive prefix of item_count -> seg_begin / seg_end, prefix[] (I + 1 entries), cursor[]
//   scatter_kernel        one thread per example: bucketed_val[atomicAdd(cursor[item], 1)] = val   (arrival order inside an item)
//   segment_sort_kernel   per item: order the segment by value (= example index = user order; rank by counting in LDS),
//                         mark runs of one user's examples (duplicate negatives), number them, clear item_count
// The result is bit-identical to a stable sort by item (tests/test_gpu_integer.py compares with numpy's stable argsort).
// More than 65 536 items keep the rocPRIM path (cdae_hip.hip: prep_batch).
//
// STATUS: opt-in (CDAE_SORT_COUNTING=1), not the default.  Measured on MI355X at ML-10M shape (profiles/r02_counting_sort.txt):
// the step got SLOWER — 0.117 -> 0.134 ms at 256 users per batch, 0.166 -> 0.217 ms at 512.  The prep stream runs beside the
// previous batch's training kernels; the same-address atomics of the hot items (the top item takes ~250 tickets per batch)
// serialise in one L2 channel and slow the decode that overlaps them (64 -> 101 us), while rocPRIM's onesweep passes, although
// four launches, touch memory in streams and disturb it less.  Kept for the bit-exact order test and as the starting point of
// an atomics-free variant (positives placed from a static CSC rank, only the negatives counted).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cdae_kernels.hpp"

namespace cdae {

constexpr uint32_t SCAN_THREADS = 1024;
constexpr uint32_t COUNTING_SORT_MAX_ITEMS = 65536;     // count_scan_kernel: SCAN_THREADS x 64 counters
constexpr uint32_t SEGSORT_THREADS = 256;               // small workgroups: they share the chip with the training kernels of the previous batch
constexpr uint32_t SEGSORT_ITEMS = 16;                   // consecutive items per workgroup
constexpr uint32_t SEGSORT_WINDOW = 3072;                // examples held in LDS at a time (2 x 24 KiB)

// exclusive prefix over the per-item example counts of the batch; one workgroup.  Wavefront w owns the contiguous item range
// [w, w+1) * ceil(I / 16 / 64) * 64 and walks it 64 items at a time (coalesced, all loads of the range issued up front: the
// kernel is one L2 round trip plus shuffles, not a chain of dependent loads).
constexpr uint32_t SCAN_CHUNKS_MAX = COUNTING_SORT_MAX_ITEMS / SCAN_THREADS;   // 64 chunks of 64 items per wavefront
__global__ void __launch_bounds__(SCAN_THREADS)
count_scan_kernel(const uint32_t* __restrict__ item_count, uint32_t num_items, uint32_t* __restrict__ prefix /* [I + 1] */,
                  uint32_t* __restrict__ cursor /* [I]: = prefix, bumped by scatter_kernel */,
                  uint32_t* __restrict__ seg_begin, uint32_t* __restrict__ seg_end, uint32_t* __restrict__ dup_count) {
  __shared__ uint32_t wave_tot[SCAN_THREADS / WAVE];
  constexpr uint32_t NW = SCAN_THREADS / WAVE;
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;
  const uint32_t chunks = (num_items + NW * WAVE - 1) / (NW * WAVE);        // per wavefront, <= SCAN_CHUNKS_MAX
  const uint32_t base_item = wid * chunks * WAVE;
  auto body = [&](auto tag) {
    constexpr uint32_t NC = decltype(tag)::value;                           // compile-time bound: loads unrolled and in flight together
    uint32_t c[NC];
#pragma unroll
    for (uint32_t k = 0; k < NC; ++k) {
      const uint32_t i = base_item + k * WAVE + lane;
      c[k] = (k < chunks && i < num_items) ? item_count[i] : 0u;
    }
    uint32_t carry = 0;
    uint32_t incl[NC];
#pragma unroll
    for (uint32_t k = 0; k < NC; ++k) {
      uint32_t v = c[k];
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, WAVE);
        if ((int)lane >= off) v += o;
      }
      incl[k] = v + carry;
      carry += __shfl(v, WAVE - 1, WAVE);
    }
    if (lane == 0) wave_tot[wid] = carry;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wid; ++w) before += wave_tot[w];
#pragma unroll
    for (uint32_t k = 0; k < NC; ++k) {
      const uint32_t i = base_item + k * WAVE + lane;
      if (k < chunks && i < num_items) {
        const uint32_t excl = before + incl[k] - c[k];
        prefix[i] = excl;
        cursor[i] = excl;
        seg_begin[i] = c[k] ? excl : 0u;                                    // items without examples keep (0, 0)
        seg_end[i] = c[k] ? excl + c[k] : 0u;
      }
    }
    if (threadIdx.x == SCAN_THREADS - 1) prefix[num_items] = before + carry;   // last wavefront: grand total
  };
  if (chunks <= 4) body(std::integral_constant<uint32_t, 4>{});
  else if (chunks <= 16) body(std::integral_constant<uint32_t, 16>{});
  else body(std::integral_constant<uint32_t, SCAN_CHUNKS_MAX>{});
  if (threadIdx.x == 0) *dup_count = 0u;
}

__global__ void __launch_bounds__(256)
scatter_kernel(const uint32_t* __restrict__ ex_item, const uint64_t* __restrict__ ex_val, uint32_t n_ex,
               uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted_item,
               uint64_t* __restrict__ bucketed_val /* item-major, arrival order inside an item */) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ex) return;
  const uint32_t item = ex_item[e];
  const uint32_t pos = atomicAdd(cursor + item, 1u);        // one returning atomic per thread: the latency hides behind E threads
  sorted_item[pos] = item;
  bucketed_val[pos] = ex_val[e];
}

// One workgroup per SEGSORT_ITEMS consecutive items = one contiguous range of sorted positions.  Inside an item the
// examples are ordered by value (the example index sits in the high word: user-major order) by counting the smaller values
// of the same item — out of place, bucketed_val -> sorted_val, through LDS while a window of items fits (always, at the
// batch sizes in use: an item holds at most one positive per batch user plus a few dozen negatives), straight from global
// memory for a longer segment.  Then runs of one user's examples are flagged (DUP_PREV / DUP_NEXT in the example word),
// every second-or-later example of a run gets a correction-row number (one global counter bump per workgroup), and the
// items' ticket counters are cleared for the next batch that uses this buffer set.
__global__ void __launch_bounds__(SEGSORT_THREADS)
segment_sort_kernel(uint32_t num_items, const uint32_t* __restrict__ prefix, const uint64_t* __restrict__ bucketed_val,
                    uint64_t* __restrict__ sorted_val, uint32_t* __restrict__ item_count, uint32_t* __restrict__ dup_count,
                    uint32_t dup_cap, uint32_t* __restrict__ dup_of_pos,
                    uint32_t* __restrict__ dup_of_ex /* pre-filled with DUP_NONE */) {
  __shared__ uint64_t raw[SEGSORT_WINDOW], srt[SEGSORT_WINDOW];
  __shared__ uint32_t item_off[SEGSORT_ITEMS + 1];
  __shared__ uint32_t blk_count, blk_base;
  const uint32_t i0 = blockIdx.x * SEGSORT_ITEMS, i1 = min(num_items, i0 + SEGSORT_ITEMS);
  const uint32_t nk = i1 - i0;
  if (threadIdx.x <= nk) item_off[threadIdx.x] = prefix[i0 + threadIdx.x];
  if (threadIdx.x == 0) blk_count = 0u;
  __syncthreads();
  if (threadIdx.x < nk) item_count[i0 + threadIdx.x] = 0u;                 // tickets of the next batch on this buffer set
  auto item_of = [&](uint32_t k, uint32_t pos) {                           // item (window-local index) of sorted position `pos`
    while (item_off[k + 1] <= pos) ++k;
    return k;
  };
  uint32_t k0 = 0;
  while (k0 < nk) {                                                         // windows of consecutive items
    uint32_t k1 = k0 + 1;
    while (k1 < nk && item_off[k1 + 1] - item_off[k0] <= SEGSORT_WINDOW) ++k1;
    const uint32_t p0 = item_off[k0], n = item_off[k1] - p0;
    const bool in_lds = n <= SEGSORT_WINDOW;                                // else: ONE oversized item
    if (in_lds) {
      for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) raw[q] = bucketed_val[p0 + q];
      __syncthreads();
    }
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {               // rank inside the item -> sorted place
      const uint32_t k = item_of(k0, p0 + q);
      const uint32_t a = item_off[k] - p0, b = item_off[k + 1] - p0;
      uint32_t rank = 0;
      if (in_lds) {
        const uint64_t v = raw[q];
        for (uint32_t j = a; j < b; ++j) rank += raw[j] < v ? 1u : 0u;
        srt[a + rank] = v;
      } else {
        const uint64_t v = bucketed_val[p0 + q];
        for (uint32_t j = a; j < b; ++j) rank += bucketed_val[p0 + j] < v ? 1u : 0u;
        sorted_val[p0 + a + rank] = v;
      }
    }
    __syncthreads();                                                        // (orders the workgroup's global writes too)
    uint32_t my_dups = 0;
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {               // flags of sorted position q
      const uint32_t k = item_of(k0, p0 + q);
      const uint32_t a = item_off[k] - p0, b = item_off[k + 1] - p0;
      // neighbours may be mid-update when they are read from global memory: only their slot bits are compared
      uint64_t v = in_lds ? srt[q] : sorted_val[p0 + q];
      const uint32_t slot = (uint32_t)v & SLOT_MASK;
      uint32_t flags = 0;
      if (q > a && ((uint32_t)(in_lds ? srt[q - 1] : sorted_val[p0 + q - 1]) & SLOT_MASK) == slot) flags |= DUP_PREV_BIT;
      if (q + 1 < b && ((uint32_t)(in_lds ? srt[q + 1] : sorted_val[p0 + q + 1]) & SLOT_MASK) == slot) flags |= DUP_NEXT_BIT;
      if (in_lds || flags) sorted_val[p0 + q] = v | flags;
      if (flags & DUP_PREV_BIT) ++my_dups;
    }
    uint32_t off = my_dups ? atomicAdd(&blk_count, my_dups) : 0u;           // LDS
    __syncthreads();
    if (threadIdx.x == 0) { blk_base = blk_count ? atomicAdd(dup_count, blk_count) : 0u; blk_count = 0u; }
    __syncthreads();
    if (my_dups) {
      off += blk_base;
      for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {             // this thread's own positions again
        const uint64_t v = sorted_val[p0 + q];
        if (!((uint32_t)v & DUP_PREV_BIT)) continue;
        const uint32_t idx = off < dup_cap ? off : DUP_NONE;
        ++off;
        dup_of_pos[p0 + q] = idx;
        dup_of_ex[(uint32_t)(v >> 32)] = idx;
      }
    }
    __syncthreads();
    k0 = k1;
  }
}

}  // namespace cdae
