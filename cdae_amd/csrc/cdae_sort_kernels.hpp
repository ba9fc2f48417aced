// cdae_sort_kernels.hpp — item-major ordering of a batch's example list without a library sort.
//
// The decode is transposed (cdae_kernels.hpp): every item row walks the batch's examples on it in USER order, so the
// user-major example list that sample_kernel writes (cdae.hpp:217-220, 361-371) has to be turned item-major, stably.  A
// generic radix sort of (16-bit key, 64-bit value) pairs — rocPRIM onesweep: histogram, scan, two digit passes, five fills of
// its own — is 10 launches and ~75 us of mostly launch latency per batch on the prep stream: once the training step dropped
// below ~95 us it was the prep chain that paced the loop (profiles/r02_wave_timeline_256.txt: decode waiting 12 us for `ready`).
// The keys are item ids and the order wanted inside an item is the order of the (unique) example words, so a counting sort
// does it in four small launches and without a single global atomic:
//   tile_hist_kernel      one workgroup per TILE_EX examples: per-item counts of the tile in LDS -> hist[tile][item]
//   item_tile_scan_kernel one thread per item: exclusive prefix over the tiles (in place), the item's total -> item_count, and the
//                         totals' exclusive prefix inside each block of 256 items + the block's sum
//   tile_scatter_kernel   the tiles again: prefix[item] = scan of the block sums + the in-block prefix (workgroup 0 writes prefix[] /
//                         seg_begin / seg_end), cursor[item] = prefix[item] + hist[tile][item]; every example takes a ticket from
//                         its item's cursor (LDS atomic: arrival order inside (item, tile), tiles in order) -> bucketed_val
//   segment_sort_kernel   per item: order the segment by value (= example index = user order; rank by counting in LDS),
//                         mark runs of one user's examples (duplicate negatives), number them
// The result is bit-identical to a stable sort by item (tests/test_gpu_integer.py compares with numpy's stable argsort).
// Item spaces above TILE_SORT_MAX_ITEMS (the per-tile cursors must fit LDS) always take the rocPRIM path (cdae_hip.hip: prep_batch).
//
// STATUS: opt-in (CDAE_SORT_TILE=1), not the default.  Measured on MI355X at ML-10M shape (profiles/r02_tile_sort.txt): the prep
// chain is shorter — sample 17.5 + hist 6.6 + scan 7.9 + scatter 16.5 + segment sort 23.4 = 72 us against ~100 us with rocPRIM —
// but the training step does not follow: 0.1001 vs 0.0998 ms at 256 users per batch, 0.146 vs 0.141 at 512.  The prep stream
// runs beside the training kernels of the previous batch; its kernels are slowed 2-3x by that company and slow the decode in
// turn, and the wide launches here (1325 workgroups of segment_sort_kernel, 1024-thread tiles) disturb it at least as much as
// rocPRIM's ~85-workgroup passes.  Variants tried: 2048 / 4096 examples per tile, 256 / 1024 threads per tile, the item prefix
// as a launch of its own / in every scatter workgroup / two-level, 4 / 8 / 16 items and 64 / 128 / 256 threads per segment-sort
// workgroup, LDS window 768 / 3072: all within +-4 % of rocPRIM, none consistently better.
//
// History: the first counting sort of this round took its tickets with GLOBAL atomics (one counter per item, bumped in
// sample_kernel and again in a scatter kernel): the ~130-250 same-address atomics of a hot item serialise in one L2 channel
// (step 0.117 -> 0.134 ms at 256 users; profiles/r02_counting_sort.txt).  Per-tile LDS counters have no such hot spot.  Two
// findings carried over to the default path: numbering the correction rows from ONE global counter is a chain of same-address
// returning atomics (now striped, cdae_kernels.hpp DUP_STRIPES), and a rank-by-counting loop with one LDS read in flight made
// the workgroup that holds the most popular items a 35 us serial tail (now eight reads in flight).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cdae_kernels.hpp"

namespace cdae {

#ifndef CDAE_SEGSORT_THREADS
#define CDAE_SEGSORT_THREADS 256
#endif
constexpr uint32_t SEGSORT_THREADS = CDAE_SEGSORT_THREADS;               // small workgroups: they share the chip with the training kernels of the previous batch
#ifndef CDAE_SEGSORT_ITEMS
#define CDAE_SEGSORT_ITEMS 8
#endif
constexpr uint32_t SEGSORT_ITEMS = CDAE_SEGSORT_ITEMS;    // consecutive items per workgroup
#ifndef CDAE_SEGSORT_WINDOW
#define CDAE_SEGSORT_WINDOW 3072
#endif
constexpr uint32_t SEGSORT_WINDOW = CDAE_SEGSORT_WINDOW;   // examples held in LDS at a time (2 arrays of 8-byte values)

constexpr uint32_t TILE_SORT_MAX_ITEMS = 32768;         // tile_scatter_kernel's cursors: 4 bytes per item of LDS (128 KiB at the limit)
#ifndef CDAE_TILE_EX
#define CDAE_TILE_EX 4096
#endif
constexpr uint32_t TILE_EX = CDAE_TILE_EX;                // examples per tile (<= 65535: 16-bit counters in tile_hist_kernel)
constexpr uint32_t TILE_THREADS = 1024;                 // four examples per thread: the scatter is a chain of LDS-atomic -> scattered store per example

__global__ void __launch_bounds__(TILE_THREADS)
tile_hist_kernel(const uint32_t* __restrict__ ex_item, uint32_t n_ex, uint32_t num_items, uint32_t* __restrict__ hist /* [tiles][I] */) {
  extern __shared__ uint32_t tile_cnt[];                                   // two 16-bit counters per word (a tile holds TILE_EX <= 65535 examples)
  const uint32_t words = (num_items + 1u) / 2u;
  for (uint32_t i = threadIdx.x; i < words; i += TILE_THREADS) tile_cnt[i] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * TILE_EX;
#pragma unroll
  for (uint32_t j = 0; j < TILE_EX / TILE_THREADS; ++j) {
    const uint32_t e = base + j * TILE_THREADS + threadIdx.x;
    if (e < n_ex) {
      const uint32_t it = ex_item[e];
      // (item shard, sampled decode: an example on another shard's row carries VOID = num_items — it takes no ticket and no place)
      if (it < num_items) atomicAdd(&tile_cnt[it >> 1], 1u << (16u * (it & 1u)));              // LDS
    }
  }
  __syncthreads();
  uint32_t* out = hist + (size_t)blockIdx.x * num_items;
  for (uint32_t i = threadIdx.x; i < num_items; i += TILE_THREADS) out[i] = (tile_cnt[i >> 1] >> (16u * (i & 1u))) & 0xFFFFu;
}

// hist[t][i] -> exclusive prefix over t (the tile's first ticket inside item i); item_count[i] = the item's total;
// local_prefix[i] = exclusive prefix of the totals inside the workgroup's block of 256 items, block_total[b] = the block's sum
// (tile_scatter_kernel adds the scan of the <= 128 block totals: the item prefix without a pass over all items per workgroup)
__global__ void __launch_bounds__(256)
item_tile_scan_kernel(uint32_t* __restrict__ hist, uint32_t n_tiles, uint32_t num_items, uint32_t* __restrict__ item_count,
                      uint32_t* __restrict__ local_prefix, uint32_t* __restrict__ block_total) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t run = 0;
  if (i < num_items) {
    constexpr uint32_t UN = 8;                                             // loads of eight tiles in flight
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += UN) {
      uint32_t c[UN];
#pragma unroll
      for (uint32_t k = 0; k < UN; ++k) c[k] = t0 + k < n_tiles ? hist[(size_t)(t0 + k) * num_items + i] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < UN; ++k) {
        if (t0 + k < n_tiles) hist[(size_t)(t0 + k) * num_items + i] = run;
        run += c[k];
      }
    }
    item_count[i] = run;
  }
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;
  uint32_t incl = run;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, WAVE);
    if ((int)lane >= off) incl += o;
  }
  if (lane == WAVE - 1) wave_sum[wid] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wid; ++w) before += wave_sum[w];
  if (i < num_items) local_prefix[i] = before + incl - run;
  if (threadIdx.x == 255) block_total[blockIdx.x] = before + incl;
}

// cursor[item] = (scan of the <= 128 block totals)[item / 256] + local_prefix[item] + hist[tile][item]: every workgroup scans the
// block totals itself (one wavefront, two values per lane) instead of waiting on a launch of its own; workgroup 0 writes
// prefix[] and the segment table for the kernels that follow.
__global__ void __launch_bounds__(TILE_THREADS)
tile_scatter_kernel(const uint32_t* __restrict__ ex_item, const uint64_t* __restrict__ ex_val, uint32_t n_ex, uint32_t num_items,
                    const uint32_t* __restrict__ hist /* tile prefixes */, const uint32_t* __restrict__ item_count,
                    const uint32_t* __restrict__ local_prefix, const uint32_t* __restrict__ block_total,
                    uint32_t* __restrict__ prefix /* [I + 1], written by workgroup 0 */, uint32_t* __restrict__ seg_begin,
                    uint32_t* __restrict__ seg_end, uint32_t* __restrict__ dup_count,
                    uint64_t* __restrict__ bucketed_val /* item-major, arrival order inside (item, tile) */,
                    const uint32_t* __restrict__ rank_of, uint32_t* __restrict__ segr_begin, uint32_t* __restrict__ segr_end) {
  extern __shared__ uint32_t tile_cur[];                                   // [num_items] cursors, then [128] block bases
  uint32_t* block_base = tile_cur + num_items;
  const uint32_t n_blocks = (num_items + 255u) / 256u;                     // <= 128 (TILE_SORT_MAX_ITEMS / 256)
  if (threadIdx.x < WAVE) {
    const uint32_t a0 = 2u * threadIdx.x, a1 = a0 + 1u;
    const uint32_t v0 = a0 < n_blocks ? block_total[a0] : 0u, v1 = a1 < n_blocks ? block_total[a1] : 0u;
    uint32_t incl = v0 + v1;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, WAVE);
      if ((int)threadIdx.x >= off) incl += o;
    }
    block_base[a0] = incl - v0 - v1;
    block_base[a1] = incl - v1;
  }
  __syncthreads();
  const uint32_t* mine_hist = hist + (size_t)blockIdx.x * num_items;
  for (uint32_t i = threadIdx.x; i < num_items; i += TILE_THREADS) {
    const uint32_t p = block_base[i >> 8] + local_prefix[i];
    tile_cur[i] = p + mine_hist[i];                                        // this tile's first ticket of item i
    if (blockIdx.x == 0) {                                                 // the tables the kernels after this one read
      const uint32_t c = item_count[i];
      prefix[i] = p;
      seg_begin[i] = c ? p : 0u;                                           // items without examples keep (0, 0)
      seg_end[i] = c ? p + c : 0u;
      if (rank_of) { segr_begin[rank_of[i]] = c ? p : 0u; segr_end[rank_of[i]] = c ? p + c : 0u; }
    }
  }
  if (blockIdx.x == 0) {
    // one past the last item's segment = the number of examples that have a place (n_ex unless some are VOID)
    if (threadIdx.x == 0) prefix[num_items] = block_base[(num_items - 1u) >> 8] + local_prefix[num_items - 1u] + item_count[num_items - 1u];
    if (threadIdx.x < DUP_STRIPES) dup_count[threadIdx.x] = 0u;
  }
  __syncthreads();
  const uint32_t base = blockIdx.x * TILE_EX;
#pragma unroll
  for (uint32_t j = 0; j < TILE_EX / TILE_THREADS; ++j) {
    const uint32_t e = base + j * TILE_THREADS + threadIdx.x;
    if (e < n_ex) {
      const uint32_t it = ex_item[e];
      if (it < num_items) bucketed_val[atomicAdd(&tile_cur[it], 1u)] = ex_val[e];
    }
  }
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
                    uint32_t* __restrict__ dup_of_ex /* pre-filled with DUP_NONE */, uint32_t stripes /* 1..DUP_STRIPES */) {
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
        // eight LDS reads in flight (independent counters): a popular item has 100+ examples per batch, and with one read at a
        // time the workgroup that holds the most popular items was a 35 us serial tail (113 us at 512 users per batch)
        const uint64_t v = raw[q];
        uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t j = a;
        for (; j + 8 <= b; j += 8) {
#pragma unroll
          for (int t = 0; t < 8; ++t) r[t] += raw[j + t] < v ? 1u : 0u;
        }
        for (; j < b; ++j) r[0] += raw[j] < v ? 1u : 0u;
        rank = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
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
    if (threadIdx.x == 0) { blk_base = blk_count ? atomicAdd(dup_count + blockIdx.x % stripes, blk_count) : 0u; blk_count = 0u; }
    __syncthreads();
    if (my_dups) {
      off += blk_base;
      const uint32_t stripe_cap = dup_cap / stripes, stripe0 = (blockIdx.x % stripes) * stripe_cap;
      for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {             // this thread's own positions again
        const uint64_t v = sorted_val[p0 + q];
        if (!((uint32_t)v & DUP_PREV_BIT)) continue;
        const uint32_t idx = off < stripe_cap ? stripe0 + off : DUP_NONE;
        ++off;
        dup_of_pos[p0 + q] = idx;
        dup_of_ex[(uint32_t)(v >> 32)] = idx;
      }
    }
    __syncthreads();
    k0 = k1;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// bucket_sort_kernel (round 5): the DEFAULT item-major ordering of a batch — ONE narrow launch behind sample_kernel, no library
// call, no fill, nothing cleared in front of it.
//
// Why narrow, and why it does not scan.  The prep chain runs beside the previous batch's training kernels; what counts is how little it
// takes from them (round 2's tile kernels, ~1300 workgroups wide, lost to rocPRIM for that reason; this kernel's own first two forms —
// every workgroup scanning the batch's whole key list once or twice — measured 90 and 51 us on 64 CUs' worth of LDS and lost 3 % of the
// step to the library sort).  So the routing is done where the examples are BORN: the item ids are cut, on the host, into ranges that
// expect equal numbers of examples per batch (popularity + the uniform negatives; <= BK_ITEMS items each), and sample_kernel's
// wavefront (one work unit: <= 384 examples) drops every example into a CELL per (range, unit) — 32 words in global memory, word 0 the
// count, then (example index << 12 | item - range start) — with an LDS counter per range, no global atomic.  A workgroup here OWNS a range:
//   gather   read the range's cells (one 128-byte cell per unit of the batch), prefix their counts, copy the entries into an LDS list;
//   prefix   publish the range's total and wait for the totals of the ranges in front (one word per range, written once; workgroups
//            are dispatched in index order, so what a workgroup waits for is running or done); count the list per item in LDS — the
//            counting atomic's return value is the example's ticket inside its item —, scan the counts: the global item-major position
//            of every segment -> all four segment tables, for every item of the range, also the empty ones (nothing to clear);
//   place    fetch the example words into the items' LDS buckets (offset + ticket);
//   order    inside an item the tickets are in arrival order: rank every word among its item's words (the example index sits in the
//            high half: user order), all in LDS; flag runs of one user's examples (duplicate negatives), number them (striped counters,
//            as segment_kernel does), write sorted_val / dup_of_pos / dup_of_ex.
// Output: bit-identical to a stable sort by item + segment_kernel (tests/test_gpu_integer.py: numpy's stable argsort; VOID examples of
// a sampled item shard belong to no range and are never picked up).
// SCAN fallback (the first form, kept): when there are no cells (IMF / BPR sampler), a cell overflowed (a unit put > 31 examples into
// one range: sample_kernel raises the batch's tag in `cell_flag`) or the range holds more than the LDS window, the workgroup scans the
// batch's 16-bit key list instead — once when its wave lists hold everything, once more per group of items otherwise; a single item above
// the window is ranked in global memory (correct, slow; never at the batch sizes in use).
constexpr uint32_t BK_THREADS = 512;
constexpr uint32_t BK_WINDOW = 4096;      // example words of one group held in LDS (x 2: arrival order, sorted)
constexpr uint32_t BK_ITEMS = 2048;       // most items of one range (LDS counters: 4 per thread in bk_block_scan)
constexpr uint32_t BK_WAVE_LIST = BK_WINDOW / (BK_THREADS / 64);   // scan fallback: records per wavefront list (they live in the sorted-words array until it is needed)
constexpr uint32_t BK_SPIN_CAP = 1u << 21;  // polls (~0.5 us each) of a range's total before bucket_sort_kernel gives up (error bit 1)
constexpr uint32_t BK_MAX_RANGES = 1024;  // wg_state words (sample_kernel clears them with the other per-batch counters)
constexpr uint32_t BK_CELL_UNITS_MAX = 16384;   // most units of a batch the cell path takes (their counts, 2 bytes each, fit the `raw` array)
constexpr size_t BK_LDS_BYTES = (size_t)BK_WINDOW * 8 * 2 + (size_t)BK_WINDOW * 2 + (size_t)(BK_ITEMS + 1) * 4 * 2 + 64 * 4;
static_assert(BK_ITEMS <= 4 * BK_THREADS && BK_ITEMS <= (1u << BKC_ITEM_BITS), "bk_block_scan takes 4 counters per thread; a cell entry holds the item in BKC_ITEM_BITS bits");

// exclusive scan of v[0 .. n) (n <= 4 * BK_THREADS) in place, v[n] = total; every thread of the workgroup calls it
__device__ __forceinline__ void bk_block_scan(uint32_t* v, uint32_t n, uint32_t* wave_tot) {
  const uint32_t t = threadIdx.x, lane = t % WAVE, wid = t / WAVE;
  uint32_t c[4], sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) { c[j] = 4u * t + j < n ? v[4u * t + j] : 0u; sum += c[j]; }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, WAVE);
    if ((int)lane >= off) incl += o;
  }
  if (lane == WAVE - 1) wave_tot[wid] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wid; ++w) before += wave_tot[w];
  uint32_t run = before + incl - sum;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) { if (4u * t + j < n) v[4u * t + j] = run; run += c[j]; }
  if (t == BK_THREADS - 1) v[n] = run;           // (the last thread's run ends at the total: items beyond n counted 0)
  __syncthreads();
}

// f(e, k, valid): example index e carries key k (valid = e < n_ex; f is called by ALL lanes of a wavefront together, so it may use
// wave-wide operations) — every example of the batch.  8 keys per 16-byte load, BK_UNROLL loads per thread in flight and the next
// round's requested before this round's keys are looked at.
constexpr uint32_t BK_UNROLL = 4;
template <typename F>
__device__ __forceinline__ void bk_scan_keys(const uint16_t* __restrict__ keys, uint32_t n_ex, F&& f) {
  const uint32_t n8 = n_ex & ~7u;
  constexpr uint32_t ROUND = BK_THREADS * 8u * BK_UNROLL;
  uint4 q[BK_UNROLL], nx[BK_UNROLL];
  auto request = [&](uint4 (&dst)[BK_UNROLL], uint32_t r0) {
#pragma unroll
    for (uint32_t u = 0; u < BK_UNROLL; ++u) {
      const uint32_t e0 = r0 + (u * BK_THREADS + threadIdx.x) * 8u;
      dst[u] = e0 < n8 ? *reinterpret_cast<const uint4*>(keys + e0) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
  };
  request(q, 0u);
  for (uint32_t r0 = 0; r0 < n8; r0 += ROUND) {
    const bool more = r0 + ROUND < n8;
    if (more) request(nx, r0 + ROUND);
#pragma unroll
    for (uint32_t u = 0; u < BK_UNROLL; ++u) {
      const uint32_t e0 = r0 + (u * BK_THREADS + threadIdx.x) * 8u;
      if (r0 + u * BK_THREADS * 8u >= n8) break;                              // (workgroup-uniform)
      const bool valid = e0 < n8;
      const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) { f(e0 + 2u * j, w[j] & 0xFFFFu, valid); f(e0 + 2u * j + 1u, w[j] >> 16, valid); }
    }
    if (more) {
#pragma unroll
      for (uint32_t u = 0; u < BK_UNROLL; ++u) q[u] = nx[u];
    }
  }
  if (n_ex != n8 && threadIdx.x < WAVE) {                                     // the last n_ex % 8 keys: wavefront 0, all of its lanes
    const bool valid = threadIdx.x < n_ex - n8;
    f(n8 + threadIdx.x, valid ? (uint32_t)keys[n8 + threadIdx.x] : 0u, valid);
  }
}

__global__ void __launch_bounds__(BK_THREADS)
bucket_sort_kernel(const uint16_t* __restrict__ keys, const uint64_t* __restrict__ ex_val, uint32_t n_ex,
                   const uint32_t* __restrict__ range_cut /* [ranges + 1] item ids */, uint32_t* __restrict__ wg_state /* [ranges], 0 on entry */,
                   const uint32_t* __restrict__ cells /* [ranges][cell_units][BKC_SLOTS] written by sample_kernel, or nullptr */, uint32_t cell_units,
                   const uint32_t* __restrict__ cell_flag /* == cell_tag: a cell of this batch overflowed */, uint32_t cell_tag,
                   uint32_t* __restrict__ seg_begin, uint32_t* __restrict__ seg_end, const uint32_t* __restrict__ rank_of,
                   uint32_t* __restrict__ segr_begin, uint32_t* __restrict__ segr_end, uint64_t* __restrict__ sorted_val,
                   uint64_t* __restrict__ scratch_val /* [n_ex]: only an item above the LDS window uses it */,
                   uint32_t* __restrict__ dup_count, uint32_t dup_cap, uint32_t* __restrict__ dup_of_pos,
                   uint32_t* __restrict__ dup_of_ex /* pre-filled with DUP_NONE */, uint32_t stripes,
                   uint32_t* __restrict__ err /* the handle's device error word: bit 1 is raised when the wait below gives up */) {
#ifdef CDAE_PREP_SETPRIO
  __builtin_amdgcn_s_setprio(CDAE_PREP_SETPRIO);
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char bk_lds[];
  uint64_t* raw = reinterpret_cast<uint64_t*>(bk_lds);                        // [BK_WINDOW] arrival order inside an item
  uint64_t* srt = raw + BK_WINDOW;                                            // [BK_WINDOW] sorted
  uint32_t* off = reinterpret_cast<uint32_t*>(srt + BK_WINDOW);               // [BK_ITEMS + 1] counts -> exclusive prefix (range-relative)
  uint32_t* cur = off + BK_ITEMS + 1;                                         // [BK_ITEMS + 1] scan fallback: tickets of the group being placed
  uint32_t* misc = cur + BK_ITEMS + 1;                                        // [64]: wave totals (16) | base | blk_count | blk_base | overflow | total
  uint16_t* raw_item = reinterpret_cast<uint16_t*>(misc + 64);                // [BK_WINDOW] range-local item of raw[q]
  const uint32_t t = threadIdx.x, w = blockIdx.x, lane = t % WAVE;
  const uint32_t lo = range_cut[w], n_it = range_cut[w + 1] - lo;
  for (uint32_t i = t; i <= n_it; i += BK_THREADS) off[i] = 0u;
  if (t == 0) { misc[16] = 0u; misc[17] = 0u; misc[19] = 0u; }
  __syncthreads();

  // ================= front end A: the range's cells ==========================================================================
  bool listed = false;                                                         // raw / raw_item hold the whole range, off[] is scanned
  bool scanned = false;                                                        // off[] holds the scanned counts (either front end)
  if (cells && cell_units <= BK_CELL_UNITS_MAX && *cell_flag != cell_tag) {
    uint16_t* wcnt = reinterpret_cast<uint16_t*>(raw);                         // [cell_units] counts of the range's cells (raw is free until `place`)
    uint32_t* lst = reinterpret_cast<uint32_t*>(srt);                          // [BK_WINDOW] entries; tickets behind them
    uint16_t* tk = reinterpret_cast<uint16_t*>(lst + BK_WINDOW);
    const uint32_t* mine = cells + (size_t)w * cell_units * BKC_SLOTS;
    // thread t owns the consecutive units [u0, u1): their counts, then their entries
    const uint32_t per = (cell_units + BK_THREADS - 1) / BK_THREADS, u0 = min(cell_units, t * per), u1 = min(cell_units, u0 + per);
    uint32_t sum = 0;
    for (uint32_t u = u0; u < u1; ++u) { const uint32_t c = mine[(size_t)u * BKC_SLOTS]; wcnt[u] = (uint16_t)c; sum += c; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o, WAVE);
      if ((int)lane >= o) incl += v;
    }
    if (lane == WAVE - 1) misc[t / WAVE] = incl;
    __syncthreads();
    uint32_t pos = incl - sum, total = 0;
    for (uint32_t v = 0; v < BK_THREADS / WAVE; ++v) { if (v < t / WAVE) pos += misc[v]; total += misc[v]; }
    __syncthreads();                                                           // (misc[0..8) is reused by bk_block_scan)
    if (total <= BK_WINDOW) {
      for (uint32_t u = u0; u < u1; ++u) {
        const uint32_t c = wcnt[u];
        for (uint32_t j = 0; j < c; ++j) lst[pos + j] = mine[(size_t)u * BKC_SLOTS + 1u + j];
        pos += c;
      }
      __syncthreads();
      for (uint32_t i = t; i < total; i += BK_THREADS) tk[i] = (uint16_t)atomicAdd(&off[lst[i] & BKC_ITEM_MASK], 1u);
      __syncthreads();
      bk_block_scan(off, n_it, misc);
      for (uint32_t i = t; i < total; i += BK_THREADS) {                       // place: offset of the item + the ticket inside it
        const uint32_t ent = lst[i], d = ent & BKC_ITEM_MASK, p = off[d] + tk[i];
        // (wcnt aliases raw: every count was consumed above, before the first barrier of this block)
        raw[p] = ex_val[ent >> BKC_ITEM_BITS];
        raw_item[p] = (uint16_t)d;
      }
      listed = true; scanned = true;
    } else {
      for (uint32_t i = t; i <= n_it; i += BK_THREADS) off[i] = 0u;            // (untouched so far; kept for symmetry) the scan fallback counts from zero
      __syncthreads();
    }
  }

  // ================= front end B: scan the batch's key list ==================================================================
  uint64_t* const wave_list = srt + (t / WAVE) * BK_WAVE_LIST;
  uint32_t wave_n = 0;                                                         // (wave-uniform)
  if (!scanned) {
    // pass 1: counts of the own items.  The LDS atomic's return value is the example's ticket inside its item, and every wavefront keeps
    // a list of what it found — (example index, range-local item, ticket), appended with a wave-wide ballot, no atomic — in the LDS
    // that holds the sorted words later: when everything fits, the key list is not scanned a second time.
    bk_scan_keys(keys, n_ex, [&](uint32_t e, uint32_t k, bool valid) {
      const uint32_t d = k - lo;
      const bool mine = valid && d < n_it;
      const uint64_t m = __ballot(mine);
      if (m == 0ull) return;
      if (mine) {
        const uint32_t ticket = atomicAdd(&off[d], 1u);
        const uint32_t slot = wave_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (slot < BK_WAVE_LIST) wave_list[slot] = ((uint64_t)e << 32) | (uint64_t)(d << 16) | (uint64_t)(ticket & 0xFFFFu);
      }
      wave_n += (uint32_t)__popcll(m);
    });
    if (wave_n > BK_WAVE_LIST && lane == 0) misc[19] = 1u;                     // a list ran over: the groups below scan the keys again
    __syncthreads();
    bk_block_scan(off, n_it, misc);                                            // off[i] = examples of the range's items before i; off[n_it] = total
  }

  // ---- the range's place in the item-major list: totals of the ranges in front (one word each: total + 1, 0 = not yet known)
  if (t == 0) __hip_atomic_store(&wg_state[w], off[n_it] + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // (bounded: a range in front that never publishes — its words were not cleared by the batch's sample_kernel, or workgroups are not
  // dispatched in index order on some future stack — must end in an error the host reports, not in a hung device)
  for (uint32_t r = t; r < w; r += BK_THREADS) {
    uint32_t v, spin = 0;
    while ((v = __hip_atomic_load(&wg_state[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
      if (++spin > BK_SPIN_CAP) { v = 1u; __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }     // (host memory: a plain store)
      __builtin_amdgcn_s_sleep(1);
    }
    atomicAdd(&misc[16], v - 1u);
  }
  __syncthreads();
  const uint32_t base = misc[16];
  for (uint32_t i = t; i < n_it; i += BK_THREADS) {
    const uint32_t c = off[i + 1] - off[i], b = c ? base + off[i] : 0u, e = c ? base + off[i + 1] : 0u;   // items without examples keep (0, 0)
    seg_begin[lo + i] = b; seg_end[lo + i] = e;
    if (rank_of) { const uint32_t r = rank_of[lo + i]; segr_begin[r] = b; segr_end[r] = e; }
  }
  const bool wave_listed = !listed && misc[19] == 0u && off[n_it] <= BK_WINDOW;   // scan front end, one group, every example of it in a wave list

  // ---- groups of consecutive items whose examples fit the LDS window (ONE group unless the batch is far larger than the cut expected)
  uint32_t ia = 0;
  while (ia < n_it) {
    uint32_t ib = ia + 1;
    {
      uint32_t l = ia + 1, h = n_it;                                           // largest ib with off[ib] - off[ia] <= BK_WINDOW
      while (l < h) { const uint32_t mid = (l + h + 1) >> 1; if (off[mid] - off[ia] <= BK_WINDOW) l = mid; else h = mid - 1; }
      ib = l;
    }
    const uint32_t g0 = off[ia], n = off[ib] - g0;
    const bool in_lds = n <= BK_WINDOW;                                        // else: ONE oversized item, ranked in global memory
    if (n == 0) { ia = ib; continue; }
    if (listed) {
      // (front end A placed everything: ia = 0, ib = n_it, g0 = 0)
    } else if (wave_listed) {
      for (uint32_t i = lane; i < wave_n; i += WAVE) {
        const uint64_t rec = wave_list[i];
        const uint32_t d = (uint32_t)(rec >> 16) & 0xFFFFu, p = off[d] + ((uint32_t)rec & 0xFFFFu);
        raw[p] = ex_val[(uint32_t)(rec >> 32)];
        raw_item[p] = (uint16_t)d;
      }
    } else {
      for (uint32_t i = ia + t; i < ib; i += BK_THREADS) cur[i] = off[i] - g0;
      __syncthreads();
      bk_scan_keys(keys, n_ex, [&](uint32_t e, uint32_t k, bool valid) {
        const uint32_t d = k - lo;
        if (valid && d - ia < ib - ia) {
          const uint32_t p = atomicAdd(&cur[d], 1u);
          const uint64_t v = ex_val[e];
          if (in_lds) { raw[p] = v; raw_item[p] = (uint16_t)d; } else scratch_val[base + g0 + p] = v;
        }
      });
    }
    __threadfence_block();
    __syncthreads();
    // ---- order inside the items
    for (uint32_t q = t; q < n; q += BK_THREADS) {
      if (in_lds) {
        const uint32_t d = raw_item[q], a = off[d] - g0, b = off[d + 1] - g0;
        const uint64_t v = raw[q];
        uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};                             // eight LDS reads in flight (a popular item holds 100+ examples)
        uint32_t j = a;
        for (; j + 8 <= b; j += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) r[u] += raw[j + u] < v ? 1u : 0u;
        }
        for (; j < b; ++j) r[0] += raw[j] < v ? 1u : 0u;
        srt[a + ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))] = v;
      } else {
        const uint64_t v = scratch_val[base + g0 + q];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += scratch_val[base + g0 + j] < v ? 1u : 0u;
        sorted_val[base + g0 + rank] = v;
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- runs of one user's examples inside an item: flags, correction-row numbers (as segment_kernel)
    uint32_t my_dups = 0;
    for (uint32_t q = t; q < n; q += BK_THREADS) {
      uint32_t a = 0, b = n;
      if (in_lds) {                                                            // the item of sorted position q: the last one starting at or before it
        uint32_t l = ia, h = ib - 1;
        while (l < h) { const uint32_t mid = (l + h + 1) >> 1; if (off[mid] - g0 <= q) l = mid; else h = mid - 1; }
        a = off[l] - g0; b = off[l + 1] - g0;
      }
      const uint64_t v = in_lds ? srt[q] : sorted_val[base + g0 + q];
      const uint32_t slot = (uint32_t)v & SLOT_MASK;
      uint32_t flags = 0;                                                      // neighbours may be mid-update in global memory: only their slot bits are compared
      if (q > a && ((uint32_t)(in_lds ? srt[q - 1] : sorted_val[base + g0 + q - 1]) & SLOT_MASK) == slot) flags |= DUP_PREV_BIT;
      if (q + 1 < b && ((uint32_t)(in_lds ? srt[q + 1] : sorted_val[base + g0 + q + 1]) & SLOT_MASK) == slot) flags |= DUP_NEXT_BIT;
      if (in_lds || flags) sorted_val[base + g0 + q] = v | flags;
      if (flags & DUP_PREV_BIT) ++my_dups;
    }
    uint32_t o = my_dups ? atomicAdd(&misc[17], my_dups) : 0u;
    __syncthreads();
    if (t == 0) { misc[18] = misc[17] ? atomicAdd(dup_count + w % stripes, misc[17]) : 0u; misc[17] = 0u; }
    __syncthreads();
    if (my_dups) {
      o += misc[18];
      const uint32_t stripe_cap = dup_cap / stripes, stripe0 = (w % stripes) * stripe_cap;
      for (uint32_t q = t; q < n; q += BK_THREADS) {                          // this thread's own positions again
        const uint64_t v = sorted_val[base + g0 + q];
        if (!((uint32_t)v & DUP_PREV_BIT)) continue;
        const uint32_t idx = o < stripe_cap ? stripe0 + o : DUP_NONE;
        ++o;
        dup_of_pos[base + g0 + q] = idx;
        dup_of_ex[(uint32_t)(v >> 32)] = idx;
      }
    }
    __syncthreads();
    ia = ib;
  }
}

}  // namespace cdae
