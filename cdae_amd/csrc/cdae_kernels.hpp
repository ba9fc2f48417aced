// cdae_kernels.hpp — gfx950 device kernels of the CDAE training hot path.
//
// Replaces, per batch of users, the body of CDAE::train_one_iteration
// (/root/reference/src/model/recsys/cdae.hpp:136-146) and train_one_user_corruption (cdae.hpp:198-358).
// The reference's loop nest is   for user: { encode; for output item: {dot, loss', row step}; hidden; input rows }.
// Here the decode loop nest is TRANSPOSED: for item row: for (user, target) example of the batch in
// user order: {dot, loss', row step}.  A row lives in the registers of one wavefront for the whole
// batch, every dot product sees every earlier update of that row exactly as in the reference, and W /
// W_ag cross HBM once per batch instead of once per (user, item) touch.  The hidden layer of a user
// (z_u, and the hidden gradient hg_u = sum_e g_e D[j_e]) is evaluated against the batch-start snapshot
// of the parameters plus the user's own duplicate-negative updates (DESIGN.md "Schedule"); with
// batch_users == 1 that is exactly the reference.
//
// Register layout of a K-vector: rows are padded to Kp = 64*NI floats (NI in {1,2,4,8}); lane l of a
// 64-wide wavefront holds the NI contiguous elements k = NI*l .. NI*l+NI-1, so one row is ONE
// global_load_dwordx{NI} per lane = a fully coalesced 256*NI-byte wave access, with no tail predication.
// Pad elements are 0 in every parameter / activation (accumulators: 1) and provably stay there.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cdae_rng.h"
#include "cdae_exchange_algebra.h"

namespace cdae {

constexpr int WAVE = 64;
constexpr uint32_t SLOT_MASK = 0x0FFFFFFFu;   // example word: slot | dup_prev << 28 | dup_next << 29 | target << 30 | is_input << 31
constexpr uint32_t DUP_PREV_BIT = 1u << 28;   // (set by segment_kernel) the row's previous example is the same user's
constexpr uint32_t DUP_NEXT_BIT = 1u << 29;   // the row's next example is the same user's
constexpr uint32_t DUP_NONE = 0xFFFFFFFFu;    // "no correction row" (dup_of_pos / dup_of_ex)
// Correction rows are numbered by bumping a counter once per workgroup of segment_kernel / segment_sort_kernel.  ONE counter for
// hundreds of workgroups is a chain of same-address returning atomics (~55 ns each: 37 us of a 660-workgroup launch); the index
// space is cut into up to DUP_STRIPES stripes of at least 4096 rows with a counter each (workgroup b draws from stripe b mod
// stripes; a stripe that runs out hands out DUP_NONE and decode falls back to its atomics path for those examples — small
// problems use one stripe: they must not run out where one counter would not).
constexpr uint32_t DUP_STRIPES = 64;
constexpr uint32_t TARGET_BIT = 1u << 30;
constexpr uint32_t INPUT_BIT = 1u << 31;

// ---- late rows and the fused decode + gather launch (round 6) ----
// The `late_rows` (<= LATE_MAX) most popular rows finish last — their serial example chains bound the decode launch — so their terms
// of the hidden gradient, sum_e g_e D0[j_e] (cdae.hpp:240,248,277,285), are NOT gathered by hidden_gather_kernel: the decode leaves
// g of (user slot, late row) in Ghot[slot][rank] (one entry per user and row: the sum over a run of the user's duplicate negatives)
// and hidden_finish_kernel adds  sum_r Ghot[slot][r] D0[item_order[r]]  (+ the runs' correction rows) itself, in rank order, behind
// the gathered partial rows.  That is what lets the gather of everything else run INSIDE the decode launch, beside the late rows'
// chains (decode_gather_kernel): a gather wavefront only ever waits for rows that finish early.
constexpr uint32_t LATE_MAX = 64;               // one lane of hidden_finish_kernel per late row
struct DecodeLate {
  float* Ghot;                                  // [batch users][LATE_MAX], zeroed by the batch's encode
  uint32_t* hotdup;                             // [batch users][LATE_MAX]: correction row of (user, late row)'s duplicate run (DUP_NONE: none)
  uint32_t late_rows;                           // rows [0, late_rows) of item_order; 0: no late rows (the gather takes every example)
};
struct LateFinish {                             // what hidden_finish_kernel / hg_raw_kernel need to add the late rows' terms
  const float* Ghot; const uint32_t* hotdup; const uint32_t* items /* item_order */; const float* D0; const float* dup_corr;
  uint32_t late_rows;
};
// Fused launch only: G[e] reads G_PENDING (a quiet NaN no loss' produces) until the decode has written it.  The batch's encode fills
// G with it (a launch boundary earlier); decode writes g THROUGH to memory (sc1 stores: visible to the other XCDs' L2s without a
// fence), the gather wavefronts poll with sc1 loads — a self-validating 4-byte granule, MI355X_MICROARCH.md "handoff".
constexpr uint32_t G_PENDING = 0x7FC0DEADu;
constexpr uint32_t FUSED_SPIN_CAP = 1u << 18;   // polls (>= 1 us each) before a gather wavefront gives up and raises the handle's error word

template <bool SC1>
__device__ __forceinline__ void store_f32(float* p, float v) {
  if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ float load_f32_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (see HyperParams::debug_skip) — constant false in the shipped build: the tests compile out
#ifdef CDAE_DEVELOPER
#define CDAE_SKIP_ROLE(hp, bit) (((hp).debug_skip & (bit)) != 0u)
#else
#define CDAE_SKIP_ROLE(hp, bit) false
#endif

struct HyperParams {
  float lambda, lr, beta, scale;
  uint32_t num_neg;
  uint32_t loss_type;        // 0 SQUARE, 5 CROSS_ENTROPY (loss.hpp:10-18)
  uint32_t adagrad, asymmetric, user_factor, linear, tanh_act;
  uint32_t linear_function;  // per-user elementwise gate Uu on the input sum (cdae.hpp:382-384)
  uint64_t keep_thr;         // cdae_keep_threshold(q)
  uint64_t uid_offset;       // global id of local user 0 (data-parallel shards keep global random streams)
  uint32_t num_items;
  uint32_t K, Kp;            // num_dim and row stride (floats, = 64 * NI)
  uint32_t unit_pos;         // positives per work unit (<= UNIT_POS_MAX), see "Work units" below
  uint32_t debug_rank;       // -DCDAE_DECODE_TIMING builds: the row whose timeline decode_rows_kernel records
  unsigned long long* trace; // CDAE_WAVE_TRACE (developer aid, tools/wave_trace.py): per-wavefront {tag, start, end, extra} records, or nullptr
  uint32_t trace_odd;        // (wave trace) 1 on batches with an odd sequence number
  uint32_t debug_skip;       // -DCDAE_DEVELOPER builds only (CDAE_DEBUG_SKIP_ROLES: timing experiments, WRONG results): 1 hidden-bias role, 2 input-row role, 4 decode hot rows, 8 decode four-per-wave rows, 16 / 32 fused row step; the shipped kernels do not read it (skip_role)
  // users whose private rows (Wu, Wu_ag, Uu, Uu_ag) THIS handle holds, table row 0 = user own_u0.  Everything except an item
  // shard owns every user ([0, 2^64)); an item shard owns a contiguous range (the user node is sharded by user)
  uint64_t own_u0, own_u1;
};

// ------------------------------------------------------------------------------------------------
// Developer aid: wavefront timeline.  With CDAE_WAVE_TRACE set the handle passes a buffer of four-word records and every traced
// wavefront fills its own: {tag << 32 | id, start, end, extra} in 100 MHz device time.
constexpr uint32_t TRACE_ROLES = 32, TRACE_IDS = 1u << 16;   // record slot = role * TRACE_IDS + id (no shared counter: it would serialise the wavefronts)
constexpr unsigned long long TRACE_CAP = (unsigned long long)TRACE_ROLES * TRACE_IDS;
__device__ __forceinline__ unsigned long long trace_begin(const HyperParams& hp) {
  return hp.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
}
// where a wavefront runs: HW_ID[15:0] (wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13) | XCC_ID << 16
__device__ __forceinline__ uint32_t hw_place() {
  return (__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) & 0xFFFFu) | ((__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu) << 16);
}
__device__ __forceinline__ void trace_end(const HyperParams& hp, uint32_t tag, uint32_t id, unsigned long long t0, uint32_t extra = 0) {
  if (hp.trace && threadIdx.x % 64 == 0 && id < TRACE_IDS) {
    tag += 16u * hp.trace_odd;                                   // odd batches keep their own records: two consecutive batches survive
    unsigned long long* r = hp.trace + 4ull * ((unsigned long long)tag * TRACE_IDS + id);
    r[0] = ((unsigned long long)tag << 32) | id; r[1] = t0; r[2] = __builtin_amdgcn_s_memrealtime(); r[3] = extra;
  }
}

// ------------------------------------------------------------------------------------------------
// vector access: NI contiguous floats per lane
template <int NI>
__device__ __forceinline__ void vload(float (&d)[NI], const float* __restrict__ p) {
  if constexpr (NI == 1) {
    d[0] = p[0];
  } else if constexpr (NI == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    d[0] = t.x; d[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < NI / 4; ++q) {
      const float4 t = reinterpret_cast<const float4*>(p)[q];
      d[4 * q] = t.x; d[4 * q + 1] = t.y; d[4 * q + 2] = t.z; d[4 * q + 3] = t.w;
    }
  }
}
typedef float cdae_f4v __attribute__((ext_vector_type(4)));
template <int NI>
__device__ __forceinline__ void vstore(float* __restrict__ p, const float (&d)[NI]) {
  if constexpr (NI == 1) {
    p[0] = d[0];
  } else if constexpr (NI == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
  } else {
#pragma unroll
    for (int q = 0; q < NI / 4; ++q) {
#ifdef CDAE_NT_STORES        // experiment (profiles/r03_nt_stores.txt): streaming stores, so that less is dirty in the L2s when the launch ends
      const cdae_f4v v = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
      __builtin_nontemporal_store(v, reinterpret_cast<cdae_f4v*>(p) + q);
#else
      reinterpret_cast<float4*>(p)[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
#endif
    }
  }
}

// Write-through (sc1) forms of the row accesses, through a buffer descriptor (aux bit 4 = sc1 on gfx950): what the fused decode +
// gather launch uses for everything one workgroup writes and another reads before the launch ends.  Byte offsets stay below 2 GiB
// (the host checks the matrices' sizes before it selects the fused launch).
using cdae_b128 = decltype(__builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0), 0, 0, 0));
using cdae_b64 = decltype(__builtin_amdgcn_raw_buffer_load_b64(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0), 0, 0, 0));
using cdae_rsrc = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0));
__device__ __forceinline__ cdae_rsrc rows_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
constexpr int AUX_SC1 = 16;
template <int NI>
__device__ __forceinline__ void vstore_sc1(cdae_rsrc r, uint32_t byte_off, const float (&d)[NI]) {
  if constexpr (NI == 1) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, d[0]), r, (int)byte_off, 0, AUX_SC1);
  } else if constexpr (NI == 2) {
    cdae_b64 v; const float t[2] = {d[0], d[1]};
    __builtin_memcpy(&v, t, sizeof v);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)byte_off, 0, AUX_SC1);
  } else {
#pragma unroll
    for (int q = 0; q < NI / 4; ++q) {
      cdae_b128 v; const float t[4] = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
      __builtin_memcpy(&v, t, sizeof v);
      __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(byte_off + 16u * q), 0, AUX_SC1);
    }
  }
}
template <int NI>
__device__ __forceinline__ void vload_sc1(float (&d)[NI], cdae_rsrc r, uint32_t byte_off) {
  if constexpr (NI == 1) {
    d[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, AUX_SC1));
  } else if constexpr (NI == 2) {
    const cdae_b64 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, AUX_SC1);
    float t[2]; __builtin_memcpy(t, &v, sizeof t);
    d[0] = t[0]; d[1] = t[1];
  } else {
#pragma unroll
    for (int q = 0; q < NI / 4; ++q) {
      const cdae_b128 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(byte_off + 16u * q), 0, AUX_SC1);
      float t[4]; __builtin_memcpy(t, &v, sizeof t);
      d[4 * q] = t[0]; d[4 * q + 1] = t[1]; d[4 * q + 2] = t[2]; d[4 * q + 3] = t[3];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// scalar math (fp32, hardware transcendental units)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// loss.hpp:53-55 (SQUARE), loss.hpp:141-147 (CROSS_ENTROPY)
// The reference switches to e^y - t below -18 and to 1 - t above +18 (loss.hpp:142-145).  In fp32 the plain
// form is already those limits to within 1 ulp (|sigmoid(y) - e^y| <= e^{2y} < 3e-16 for y < -18; 1/(1+e^-y)
// rounds to 1 for y > 18; e^-y overflowing to +inf gives rcp(inf) = 0), so the hot loop stays branch-free.
__device__ __forceinline__ float loss_grad(uint32_t loss_type, float pred, float truth) {
  if (loss_type == 0u) return -2.f * (truth - pred);
  return fast_rcp(1.f + fast_exp(-pred)) - truth;
}
// loss.hpp:48-51, 132-139
__device__ __forceinline__ float loss_eval(uint32_t loss_type, float pred, float truth) {
  if (loss_type == 0u) { float e = truth - pred; return e * e; }
  float ret = (1.f - truth) * pred;
  if (pred > 18.f) return ret + fast_exp(-pred);
  if (pred < -18.f) return ret - pred;
  return ret + log1pf(fast_exp(-pred));
}
// cdae.hpp:391-414
__device__ __forceinline__ float activate(const HyperParams& hp, float x) {
  if (hp.linear) return x;
  if (!hp.tanh_act) return x > 18.f ? 1.f : (x < -18.f ? 0.f : fast_rcp(1.f + fast_exp(-x)));
  if (x > 9.f) return 1.f;
  if (x < -9.f) return -1.f;
  float r = fast_exp(-2.f * x);
  return (1.f - r) * fast_rcp(1.f + r);
}
// cdae.hpp:208-215
__device__ __forceinline__ float act_deriv(const HyperParams& hp, float z) {
  return hp.linear ? 1.f : (hp.tanh_act ? 1.f - z * z : z - z * z);
}
// one coordinate of every `if (using_adagrad_) {...} p -= lr*grad` block (e.g. cdae.hpp:252-257)
// ADA at compile time: a loop that steps several independent elements per iteration (the IMF / BPR kernels) keeps their
// sqrt -> rcp chains interleaved only when no branch sits between them
template <bool ADA>
__device__ __forceinline__ void ada_step_t(const HyperParams& hp, float& p, float& acc, float grad) {
  if (ADA) {
    acc = fmaf(grad, grad, acc);
    // -lr * grad is formed beside the sqrt -> rcp chain, so the parameter is one fma behind the rcp (every step of a row's or
    // of b's recurrence is this chain: six dependent instructions instead of seven)
    p = fmaf(-hp.lr * grad, fast_rcp(fast_sqrt(acc) + hp.beta), p);
  } else {
    p = fmaf(-hp.lr, grad, p);
  }
}
__device__ __forceinline__ void ada_step(const HyperParams& hp, float& p, float& acc, float grad) {
  if (hp.adagrad) ada_step_t<true>(hp, p, acc, grad); else ada_step_t<false>(hp, p, acc, grad);
}

// Wavefront all-reduce on the VALU's DPP lanes (no LDS crossbar): quad swaps, row mirrors, then the two
// row broadcasts; lane 63 ends up with the full sum and is read back as a scalar.  ~7 dependent VALU ops
// instead of six ds_bpermute round trips — this sits on the critical path of every decoded example.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, /*bound_ctrl*/ true);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);           // row_half_mirror
  v = dpp_add<0x140>(v);           // row_mirror        -> every lane holds its 16-lane row sum
  // row_bcast with the full row mask: rows that have no source lane add 0 (bound_ctrl), rows 0-2 end up with
  // partial sums nobody reads, and the instruction folds into one v_add_f32_dpp like the stages above
  v = dpp_add<0x142>(v);           // row_bcast:15 -> row r += sum(row r-1)
  v = dpp_add<0x143>(v);           // row_bcast:31 -> rows 2,3 += lane 31 (= rows 0+1) -> lane 63 = total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ------------------------------------------------------------------------------------------------
// Work units.  User activity is heavy-tailed (n_u from 16 to 1468 at ML-10M shape), and a kernel that gives
// every user one wavefront takes as long as its longest user.  The user-parallel kernels therefore run on
// UNITS of at most unit_pos positives of one user: unit k of a user covers positives [k*unit_pos, ...) and the
// num_neg x as many negatives that belong to them.  `uptr` is the batch's prefix array (nb + 1 entries, any
// base): user slot s owns units uptr[s] .. uptr[s+1]-1.
// bucket_sort_kernel's cells (cdae_sort_kernels.hpp): sample_kernel's wavefront (one work unit) drops every example into the cell of
// (its item's range, the unit): BKC_SLOTS words, word 0 = how many the unit put there, then (example index << BKC_ITEM_BITS | item - range start)
constexpr uint32_t BKC_SLOTS = 32, BKC_ITEM_BITS = 12, BKC_ITEM_MASK = (1u << BKC_ITEM_BITS) - 1u;
constexpr uint32_t BKC_MAX_RANGES = 256;      // LDS counters per wavefront in sample_kernel
constexpr uint32_t UNIT_POS_MAX = 128;     // HyperParams::unit_pos (set per data set / batch size by the host) never exceeds it

struct UnitRef { uint32_t slot, p0, p1; };   // user slot and the positive range [p0, p1) of the unit

__device__ __forceinline__ UnitRef locate_unit(const uint32_t unit_pos, const uint32_t* __restrict__ uptr, uint32_t nb, uint32_t g,
                                               const uint32_t* __restrict__ unit_user = nullptr, uint64_t u0 = 0) {
  uint32_t lo = 0, hi = nb;                      // largest slot with uptr[slot] <= g
  if (unit_user) {                               // `uptr` is a window of the data set's own prefix: one table look-up
    lo = unit_user[g] - (uint32_t)u0;            // instead of log2(nb) dependent loads at the head of every wavefront
  } else {
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (uptr[mid] <= g) lo = mid; else hi = mid;
    }
  }
  UnitRef r;
  r.slot = lo;
  r.p0 = (g - uptr[lo]) * unit_pos;
  r.p1 = r.p0 + unit_pos;
  return r;
}

// ------------------------------------------------------------------------------------------------
// K1  sample: dropout keep-mask + rejection-sampled negatives -> example list of the batch.
// get_corrputed_input (cdae.hpp:361-371) and sample_negative_item (recsys_model_base.hpp:46-57,
// call site cdae.hpp:217-220).  One wavefront per user; integer-only.
// Example e of user slot s sits at ex_base(s) + j: j < n_u positives, then n_u*num_neg negatives.
// membership test in a sorted row staged in LDS (device twin of cdae_row_contains)
__device__ __forceinline__ int lds_row_contains(const uint32_t* row, uint32_t n, uint32_t item) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (row[mid] < item) lo = mid + 1; else hi = mid;
  }
  return lo < n && row[lo] == item;
}
constexpr uint32_t SAMPLE_LDS_ROW = 2048;   // items of a user's row staged per wavefront (8 KiB); longer rows search global memory

__global__ void __launch_bounds__(256)
sample_kernel(HyperParams hp, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
              const uint32_t* __restrict__ uptr, uint32_t n_units, uint64_t u0, uint32_t nb, uint32_t cidx,
              uint64_t seed, uint32_t epoch, uint32_t* __restrict__ ex_item, uint64_t* __restrict__ ex_val,
              uint16_t* __restrict__ ex_key16 /* sort key copy when num_items <= 65536, else nullptr */,
              uint32_t* __restrict__ seg /* [seg_words] cleared here for segment_kernel */, uint32_t seg_words,
              uint32_t* __restrict__ dup_count, uint32_t* __restrict__ dup_of_ex,
              const uint32_t* __restrict__ unit_user /* global unit -> user id */,
              uint32_t* __restrict__ wg_state /* bucket_sort_kernel's per-range words (BK_MAX_RANGES = 1024), cleared here; or nullptr */,
              const uint32_t* __restrict__ gpos /* item shard: [2 U] (length of the user's WHOLE row, position of the first local item in it); else nullptr */,
              // Item shard of the SAMPLED decode: row_ptr / col are the WHOLE rows with global item ids (negatives are rejected
              // against the whole row and drawn from all `draw_items` items, exactly as on one GPU), the example list keeps the
              // single-GPU layout and numbering, and an example whose item lies outside [shard_item0, shard_item0 + shard_items)
              // is VOID: item = key = shard_items (one past the last local row) — it sorts behind every local row, segment_kernel
              // and hidden_gather_kernel skip it.  shard_items = 0: not a shard (ids pass through).
              uint32_t shard_item0 = 0, uint32_t shard_items = 0, uint32_t draw_items = 0,
              // bucket_sort_kernel's cells (nullptr: none): item -> range, the ranges' first items, [ranges][n_units][BKC_SLOTS] words, the
              // word that receives `cell_tag` when a unit puts more than BKC_SLOTS - 1 examples into one range (the sort then scans instead)
              const uint16_t* __restrict__ range_of = nullptr, const uint32_t* __restrict__ range_cut = nullptr, uint32_t n_ranges = 0,
              uint32_t* __restrict__ cells = nullptr, uint32_t* __restrict__ cell_flag = nullptr, uint32_t cell_tag = 0) {
#ifdef CDAE_PREP_SETPRIO
  __builtin_amdgcn_s_setprio(CDAE_PREP_SETPRIO);
#endif
  __shared__ uint32_t lds_rows[4][SAMPLE_LDS_ROW];
  __shared__ uint32_t cell_cnt[4][BKC_MAX_RANGES];
  __shared__ uint32_t cell_lo[BKC_MAX_RANGES];
  if (cells) {
    for (uint32_t i = threadIdx.x; i < n_ranges; i += blockDim.x) cell_lo[i] = range_cut[i];
    for (uint32_t i = threadIdx.x; i < 4u * BKC_MAX_RANGES; i += blockDim.x) (&cell_cnt[0][0])[i] = 0u;
    __syncthreads();
  }
  // the per-batch clears ride along (no memset launches on the prep stream)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < seg_words; i += gridDim.x * blockDim.x) seg[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < DUP_STRIPES) dup_count[threadIdx.x] = 0u;
  if (blockIdx.x == 0 && wg_state)
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) wg_state[i] = 0u;
  const uint32_t wid = threadIdx.x / WAVE;
  const uint32_t unit = blockIdx.x * (blockDim.x / WAVE) + wid;
  const uint32_t lane = threadIdx.x % WAVE;
  if (unit >= n_units) return;
  const UnitRef ur = locate_unit(hp.unit_pos, uptr, nb, uptr[0] + unit, unit_user, u0);
  const uint32_t slot = ur.slot;
  const uint64_t uid = u0 + slot;
  const int64_t r0 = row_ptr[uid];
  const uint32_t n = (uint32_t)(row_ptr[uid + 1] - r0);
  const uint32_t m = n * hp.num_neg;
  const uint32_t p0 = ur.p0, p1 = min(ur.p1, n);
  const uint32_t* row = col + r0;
  const uint64_t base = (uint64_t)(r0 - row_ptr[u0]) * (1u + hp.num_neg);
  // the example's cell: an LDS counter per range gives its slot (lanes of the wavefront may meet in one range: the atomic sorts that out)
  auto cell_add = [&](uint64_t e, uint32_t it /* local item id; >= the local item count: VOID */, uint32_t n_local) {
    if (!cells || it >= n_local) return;
    const uint32_t r = range_of[it];
    const uint32_t slot = 1u + atomicAdd(&cell_cnt[wid][r], 1u);
    if (slot < BKC_SLOTS) cells[((size_t)r * n_units + unit) * BKC_SLOTS + slot] = ((uint32_t)e << BKC_ITEM_BITS) | (it - cell_lo[r]);
  };
  const uint32_t n_local = shard_items ? shard_items : hp.num_items;
  const uint64_t key_c = cdae_rng_key(seed, epoch, uid + hp.uid_offset, CDAE_STREAM_CORRUPT);
  const uint64_t key_n = cdae_rng_key(seed, epoch, uid + hp.uid_offset, CDAE_STREAM_NEGATIVE);
  const bool staged = n <= SAMPLE_LDS_ROW;
  uint32_t* lrow = lds_rows[wid];
  if (staged)
    for (uint32_t p = lane; p < n; p += WAVE) lrow[p] = row[p];           // the whole row: negatives are tested against it
  // the dropout stream is indexed by the item's position in the user's WHOLE row: an item shard holds a slice of it
  const uint32_t n_rng = gpos ? gpos[2 * uid] : n, p_rng0 = gpos ? gpos[2 * uid + 1] : 0u;
  for (uint32_t p = p0 + lane; p < p1; p += WAVE) {
    const int keep = cdae_keep(cdae_rng_draw(key_c, (uint64_t)cidx * n_rng + p_rng0 + p), hp.keep_thr);
    const uint64_t e = base + p;
    uint32_t it = row[p];
    if (shard_items) it = it - shard_item0 < shard_items ? it - shard_item0 : shard_items;
    ex_item[e] = it;
    if (ex_key16) ex_key16[e] = (uint16_t)it;
    dup_of_ex[e] = DUP_NONE;
    ex_val[e] = (e << 32) | (uint64_t)(slot | TARGET_BIT | (keep ? INPUT_BIT : 0u));
    cell_add(e, it, n_local);
  }
  const uint32_t n_items = draw_items ? draw_items : hp.num_items;
  // the wavefront's own LDS writes are visible to it once they are issued in order (no cross-wave sharing)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = p0 * hp.num_neg + lane; i < p1 * hp.num_neg; i += WAVE) {
    const uint64_t e = base + n + i;
    uint32_t cand;
    if (staged) {
      // recsys_model_base.hpp:46-57 on the LDS copy of the row; identical draws to cdae_sample_negative
      const uint64_t draw = (uint64_t)cidx * m + i;
      bool found = false;
      cand = 0;
      for (uint32_t t = 0; t < CDAE_NEG_MAX_TRY && !found; ++t) {
        cand = cdae_item_from_draw(cdae_rng_draw(key_n, draw * CDAE_NEG_MAX_TRY + t), n_items);
        found = !lds_row_contains(lrow, n, cand);
      }
      for (uint32_t t = 0; t < n_items && !found; ++t) {
        cand = cand + 1u == n_items ? 0u : cand + 1u;
        found = !lds_row_contains(lrow, n, cand);
      }
    } else {
      cand = cdae_sample_negative(key_n, (uint64_t)cidx * m + i, row, n, n_items);
    }
    if (shard_items) cand = cand - shard_item0 < shard_items ? cand - shard_item0 : shard_items;
    ex_item[e] = cand;
    if (ex_key16) ex_key16[e] = (uint16_t)cand;
    dup_of_ex[e] = DUP_NONE;
    ex_val[e] = (e << 32) | (uint64_t)slot;
    cell_add(e, cand, n_local);
  }
  if (cells) {
    // the unit's counts, EVERY range (the cells are not cleared between batches); a count beyond the cell raises the batch's tag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool over = false;
    for (uint32_t r = lane; r < n_ranges; r += WAVE) {
      const uint32_t c = cell_cnt[wid][r];
      over = over || c >= BKC_SLOTS;
      cells[((size_t)r * n_units + unit) * BKC_SLOTS] = min(c, BKC_SLOTS - 1u);
    }
    if (over) *cell_flag = cell_tag;
  }
}

// first / one-past-last sorted position of every item that has examples (others keep 0,0).
// Also finds runs of one user's examples inside a row (duplicate negatives: the sampler draws with replacement like
// recsys_model_base.hpp:46-57) and marks them in the example word.  Every second-or-later example of a run gets a row
// `dup_idx` of the correction buffer (decode writes g * (row now - row at the user's first visit) there, the
// hidden-gradient gather adds it): dup_of_pos[p] for decode, dup_of_ex[e] for the gather (DUP_NONE for all other
// examples and when the buffer is full — decode then falls back to atomics).
constexpr uint32_t SEG_PER_THREAD = 4;               // positions per thread: one global counter bump per 1024 positions
template <typename KeyT>
__global__ void __launch_bounds__(256)
segment_kernel(const KeyT* __restrict__ sorted_item, uint64_t* sorted_val, uint32_t n_ex, uint32_t* __restrict__ seg_begin,
               uint32_t* __restrict__ seg_end, uint32_t* __restrict__ dup_count, uint32_t dup_cap,
               uint32_t* __restrict__ dup_of_pos, uint32_t* __restrict__ dup_of_ex /* pre-filled with DUP_NONE */,
               uint32_t stripes /* 1..DUP_STRIPES counters in use */,
               const uint32_t* __restrict__ rank_of /* item -> popularity rank, or nullptr */,
               uint32_t* __restrict__ segr_begin /* the same table indexed by RANK (decode / input rows read it beside item_order[rank]: */,
               uint32_t* __restrict__ segr_end /* one dependent round trip less at the head of every row) */,
               uint32_t void_key = 0xFFFFFFFFu /* item shard: examples of other shards' rows (sorted last); no segment, no duplicate marks */) {
  __shared__ uint32_t blk_count, blk_base;
  if (threadIdx.x == 0) blk_count = 0u;
  __syncthreads();
  const uint32_t p0 = blockIdx.x * (blockDim.x * SEG_PER_THREAD) + threadIdx.x;
  uint32_t dups = 0;                                  // bit i: position p0 + i * blockDim.x continues a run
#pragma unroll 4
  for (uint32_t i = 0; i < SEG_PER_THREAD; ++i) {
    const uint32_t p = p0 + i * blockDim.x;
    if (p >= n_ex) break;
    const uint32_t it = sorted_item[p];
    if (it == void_key) continue;
    const bool first = p == 0 || sorted_item[p - 1] != it, last = p + 1 == n_ex || sorted_item[p + 1] != it;
    if (first) { seg_begin[it] = p; if (rank_of) segr_begin[rank_of[it]] = p; }
    if (last) { seg_end[it] = p + 1; if (rank_of) segr_end[rank_of[it]] = p + 1; }
    const uint64_t v = sorted_val[p];
    const uint32_t slot = (uint32_t)v & SLOT_MASK;
    uint32_t flags = 0;                               // neighbours may be mid-update: only their slot bits are compared
    if (!first && ((uint32_t)sorted_val[p - 1] & SLOT_MASK) == slot) flags |= DUP_PREV_BIT;
    if (!last && ((uint32_t)sorted_val[p + 1] & SLOT_MASK) == slot) flags |= DUP_NEXT_BIT;
    if (flags) sorted_val[p] = v | flags;
    if (flags & DUP_PREV_BIT) dups |= 1u << i;
  }
  const uint32_t mine = (uint32_t)__popc(dups);
  uint32_t off = mine ? atomicAdd(&blk_count, mine) : 0u;        // LDS
  __syncthreads();
  if (threadIdx.x == 0 && blk_count) blk_base = atomicAdd(dup_count + blockIdx.x % stripes, blk_count);
  __syncthreads();
  off += blk_base;
  const uint32_t stripe_cap = dup_cap / stripes, stripe0 = (blockIdx.x % stripes) * stripe_cap;
  while (dups) {
    const uint32_t i = (uint32_t)__ffs((int)dups) - 1u;
    dups &= dups - 1u;
    const uint32_t p = p0 + i * blockDim.x;
    const uint32_t idx = off < stripe_cap ? stripe0 + off : DUP_NONE;
    ++off;
    dup_of_pos[p] = idx;
    dup_of_ex[(uint32_t)(sorted_val[p] >> 32)] = idx;
  }
}

// Full-output path (round 3): the encode writes the bf16 operand images of z itself — Zb[slot][k] (the lane's NI elements are
// contiguous) and ZTb[k][slot] — instead of a conversion launch behind it.  Rows >= the batch's users stay zero (host: zb_rows).
typedef __bf16 BF16_T;
template <int NI>
__device__ __forceinline__ void store_z_bf16(const float (&z)[NI], uint32_t slot, uint32_t lo, uint32_t Kp, uint32_t Bp,
                                             BF16_T* __restrict__ Zb, BF16_T* __restrict__ ZTb) {
  BF16_T hb[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) hb[i] = (BF16_T)z[i];
  BF16_T* zrow = Zb + (size_t)slot * Kp + lo;
#pragma unroll
  for (int i = 0; i < NI; ++i) zrow[i] = hb[i];
#pragma unroll
  for (int i = 0; i < NI; ++i) ZTb[(size_t)(lo + i) * Bp + slot] = hb[i];
}

// ------------------------------------------------------------------------------------------------
// K2  encode: z_u = act(scale * sum_{i in In(u)} W[i] + b + Wu[u])   (get_hidden_values, cdae.hpp:373-416)
// Two launches: encode_partial_kernel — one wavefront per unit sums the kept rows of its <= 128 positives
// (coalesced gather: each row is one 256*NI-byte wave load, eight rows in flight) — and
// encode_finish_kernel — one wavefront per user adds its units' partial sums in order, applies scale, b,
// Wu[u] and the activation.
// mode 0: all train items, scale 1 (inference, cdae.hpp:169); mode 1: dropout mask of stream `stream`, scale
// hp.scale (training cdae.hpp:207, data_loss cdae.hpp:92).  explicit_in: the caller supplies the corrupted
// input set itself, like the reference's public train_one_user_corruption(uid, input_set, output_set)
// (cdae.hpp:198-200): one user, one unit, mode-1 scale, no mask.
template <int NI>
__global__ void __launch_bounds__(256)
encode_partial_kernel(HyperParams hp, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                      const float* __restrict__ W, const uint32_t* __restrict__ uptr, uint32_t n_units,
                      const uint32_t* __restrict__ uids, uint64_t u0, uint32_t nb, int mode, uint32_t stream,
                      uint32_t cidx, uint64_t seed, uint32_t epoch, float* __restrict__ Hpart,
                      const uint32_t* __restrict__ explicit_in, uint32_t n_explicit,
                      const uint32_t* __restrict__ unit_user /* nullptr unless uptr is a window of the data set's prefix */,
                      const uint32_t* __restrict__ gpos = nullptr /* item shard: see sample_kernel */) {
  const uint32_t unit = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (unit >= n_units) return;
  const unsigned long long t0 = trace_begin(hp);
  const UnitRef ur = locate_unit(hp.unit_pos, uptr, nb, uptr[0] + unit, unit_user, u0);
  const uint64_t uid = uids ? (uint64_t)uids[ur.slot] : u0 + ur.slot;
  const int64_t r0 = row_ptr[uid];
  const uint32_t n = explicit_in ? n_explicit : (uint32_t)(row_ptr[uid + 1] - r0);
  const uint32_t* row = explicit_in ? explicit_in : col + r0;
  const uint32_t p_begin = explicit_in ? 0u : ur.p0, p_end = explicit_in ? n : min(ur.p1, n);
  const uint64_t key_c = cdae_rng_key(seed, epoch, uid + hp.uid_offset, stream);
  const bool none = (mode == 0 && hp.keep_thr == 0x100000000ull);   // cdae.hpp:168-172 (q == 1 -> empty input)
  const uint32_t lo = lane * NI;
  const uint32_t n_rng = gpos ? gpos[2 * uid] : n, p_rng0 = gpos ? gpos[2 * uid + 1] : 0u;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  constexpr int UN = 8;
  for (uint32_t q0 = p_begin; q0 < p_end && !none; q0 += WAVE) {
    const uint32_t p = q0 + lane;
    uint32_t item = 0;
    int keep = 0;
    if (p < p_end) {
      item = row[p];
      keep = (mode == 0 || explicit_in) ? 1 : cdae_keep(cdae_rng_draw(key_c, (uint64_t)cidx * n_rng + p_rng0 + p), hp.keep_thr);
    }
    unsigned long long mask = __ballot(keep);
    while (mask) {
      // up to UN kept rows per trip, summed in ascending item order
      float v[UN][NI];
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        if (mask) {                                          // wave-uniform
          const int src = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)item, src);
          vload<NI>(v[j], W + (size_t)it * hp.Kp + lo);
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) v[j][i] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < UN; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] += v[j][i];
    }
  }
  vstore<NI>(Hpart + (size_t)unit * hp.Kp + lo, acc);
  trace_end(hp, 1, unit, t0);
}

template <int NI>
__global__ void __launch_bounds__(256)
encode_finish_kernel(HyperParams hp, const float* __restrict__ Hpart, const uint32_t* __restrict__ uptr,
                     const float* __restrict__ Wu, const float* __restrict__ b, const uint32_t* __restrict__ uids,
                     uint64_t u0, uint32_t nb, int mode, float* __restrict__ Z, float* __restrict__ Dz,
                     float* __restrict__ HGzero /* training: the batch's duplicate-correction rows start at 0 */,
                     const float* __restrict__ Uu /* linear_function only */,
                     float* __restrict__ Ssum /* linear_function training: the unscaled input sums, for the Uu step */,
                     BF16_T* __restrict__ Zb = nullptr /* full-output path: bf16 images of z, [Bp][Kp] and (ZTb) [Kp][Bp], written here */,
                     BF16_T* __restrict__ ZTb = nullptr, uint32_t Bp = 0,
                     // training with late rows (DecodeLate): the batch's Ghot / hotdup rows start at 0 / DUP_NONE; fused launch: G starts as G_PENDING
                     float* __restrict__ Ghot = nullptr, uint32_t* __restrict__ hotdup = nullptr,
                     uint32_t* __restrict__ Gfill = nullptr, uint32_t n_fill = 0) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_fill; i += gridDim.x * blockDim.x) Gfill[i] = G_PENDING;
  const uint32_t slot = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (slot >= nb) return;
  const unsigned long long t0 = trace_begin(hp);
  if (Ghot) { Ghot[(size_t)slot * LATE_MAX + lane] = 0.f; hotdup[(size_t)slot * LATE_MAX + lane] = DUP_NONE; }
  const uint64_t uid = uids ? (uint64_t)uids[slot] : u0 + slot;
  const uint32_t lo = lane * NI;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  // b and Wu[u] are requested up front, beside the partial rows (they were one more round trip behind them)
  float bb[NI], wu[NI], z[NI], dz[NI];
  vload<NI>(bb, b + lo);
  if (hp.user_factor) vload<NI>(wu, Wu + (size_t)uid * hp.Kp + lo);
  const uint32_t ub = uptr[slot] - uptr[0], ue = uptr[slot + 1] - uptr[0];
  // unit order == item order: deterministic sum.  UF partial rows are in flight at a time (index clamped, the surplus
  // added as 0): a user has 2-3 units at ML-10M shape, and one L2 round trip per unit was most of this launch
  constexpr uint32_t UF = 8;
  for (uint32_t u = ub; u < ue; u += UF) {
    float part[UF][NI];
#pragma unroll
    for (uint32_t j = 0; j < UF; ++j) vload<NI>(part[j], Hpart + (size_t)min(u + j, ue - 1u) * hp.Kp + lo);
#pragma unroll
    for (uint32_t j = 0; j < UF; ++j) {
      const bool on = u + j < ue;                               // wave-uniform
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] += on ? part[j][i] : 0.f;
    }
  }
  const float sc = mode == 0 ? 1.f : hp.scale;
  if (hp.linear_function) {                                  // h1 = Uu[u] (.) h1   cdae.hpp:382-384
    float uu[NI];
    vload<NI>(uu, Uu + (size_t)uid * hp.Kp + lo);
    if (Ssum) vstore<NI>(Ssum + (size_t)slot * hp.Kp + lo, acc);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] *= uu[i];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float h = fmaf(acc[i], sc, bb[i]);
    if (hp.user_factor) h += wu[i];
    const float zz = activate(hp, h);
    const bool live = lo + i < hp.K;                         // pad elements of z must be 0
    z[i] = live ? zz : 0.f;
    dz[i] = live ? act_deriv(hp, zz) : 0.f;
  }
  vstore<NI>(Z + (size_t)slot * hp.Kp + lo, z);
  if (Zb) store_z_bf16<NI>(z, slot, lo, hp.Kp, Bp, Zb, ZTb);
  if (Dz) vstore<NI>(Dz + (size_t)slot * hp.Kp + lo, dz);
  if (HGzero) {
#pragma unroll
    for (int i = 0; i < NI; ++i) z[i] = 0.f;
    vstore<NI>(HGzero + (size_t)slot * hp.Kp + lo, z);
  }
  trace_end(hp, 2, slot, t0);
}

// K2 (training path, one launch): one WORKGROUP of ENC_WAVES wavefronts per user of the batch.  Wavefront w sums the kept rows of
// the user's units w, w + ENC_WAVES, ... (each exactly as encode_partial_kernel does), the per-wavefront sums meet in LDS, and
// wavefront 0 adds them in wavefront order and finishes like encode_finish_kernel.  For a user with at most ENC_WAVES units
// (1024 positives at 64 per unit) that is the same sum in the same order as the two-launch form: bit-identical z.  It removes a
// launch boundary and a launch from every training step (8.3 + 1.4 + 4.3 us -> one launch, profiles/r02_main_stream.txt); heavy
// users are still spread over wavefronts, now of their own workgroup.  The two-launch form stays for arbitrary user lists,
// caller-supplied input sets and item shards.
constexpr int ENC_WAVES = 16;
template <int NI>
__global__ void __launch_bounds__(ENC_WAVES * WAVE)
encode_users_kernel(HyperParams hp, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                    const float* __restrict__ W, const uint32_t* __restrict__ uptr, uint64_t u0, uint32_t nb,
                    uint32_t cidx, uint64_t seed, uint32_t epoch, const float* __restrict__ Wu, const float* __restrict__ b,
                    float* __restrict__ Z, float* __restrict__ Dz, float* __restrict__ HGzero,
                    const float* __restrict__ Uu /* linear_function only */, float* __restrict__ Ssum /* linear_function only */,
                    BF16_T* __restrict__ Zb = nullptr, BF16_T* __restrict__ ZTb = nullptr, uint32_t Bp = 0 /* as encode_finish_kernel */,
                    // item shard, phase 0 (round 4): stop at the RAW input sum of the local rows — block 0 of the all-reduce buffer
                    // `raw_out` [blocks][nb][Kp] — and stage the owner's rows of the private matrices (`raw_a`, then `raw_b`; zeros when
                    // another shard owns the user) behind it: what encode_partial_kernel + unit_sum_stage_kernel wrote, in one launch
                    float* __restrict__ raw_out = nullptr, const float* __restrict__ raw_a = nullptr, const float* __restrict__ raw_b = nullptr,
                    const uint32_t* __restrict__ gpos = nullptr /* item shard: see encode_partial_kernel */,
                    float* __restrict__ Ghot = nullptr, uint32_t* __restrict__ hotdup = nullptr /* as encode_finish_kernel */,
                    uint32_t* __restrict__ Gfill = nullptr, uint32_t n_fill = 0) {
  __shared__ float part[ENC_WAVES][64 * NI];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_fill; i += gridDim.x * blockDim.x) Gfill[i] = G_PENDING;
  const uint32_t slot = blockIdx.x, wid = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const unsigned long long t0 = trace_begin(hp);
  if (Ghot && wid == 1) { Ghot[(size_t)slot * LATE_MAX + lane] = 0.f; hotdup[(size_t)slot * LATE_MAX + lane] = DUP_NONE; }
  const uint64_t uid = u0 + slot;
  const uint32_t lo = lane * NI;
  const uint32_t n_units = uptr[slot + 1] - uptr[slot];
  // wavefront 0 requests what the finish needs before anything else (it was a round trip behind the sums)
  float bb[NI], wu[NI], uu[NI];
  const bool own = raw_out && cdae_xa::owns_user(uid, hp.own_u0, hp.own_u1);          // (raw mode: the private rows this shard holds)
  if (wid == 0 && !raw_out) {
    vload<NI>(bb, b + lo);
    if (hp.user_factor) vload<NI>(wu, Wu + (size_t)uid * hp.Kp + lo);
    if (hp.linear_function) vload<NI>(uu, Uu + (size_t)uid * hp.Kp + lo);
  }
  if (wid == 0 && raw_out) {
#pragma unroll
    for (int i = 0; i < NI; ++i) wu[i] = uu[i] = 0.f;
    if (own && raw_a) vload<NI>(wu, raw_a + (size_t)(uid - hp.own_u0) * hp.Kp + lo);
    if (own && raw_b) vload<NI>(uu, raw_b + (size_t)(uid - hp.own_u0) * hp.Kp + lo);
  }
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  if (wid < n_units) {
    const int64_t r0 = row_ptr[uid];
    const uint32_t n = (uint32_t)(row_ptr[uid + 1] - r0);
    const uint32_t* row = col + r0;
    const uint64_t key_c = cdae_rng_key(seed, epoch, uid + hp.uid_offset, CDAE_STREAM_CORRUPT);
    // item shard: `row` is the slice of the user's row on this shard's items; the dropout stream is indexed by position in the WHOLE row
    const uint32_t n_rng = gpos ? gpos[2 * uid] : n, p_rng0 = gpos ? gpos[2 * uid + 1] : 0u;
#ifndef CDAE_ENCODE_UN
#define CDAE_ENCODE_UN 8
#endif
    constexpr int UN = CDAE_ENCODE_UN;                                  // kept rows in flight per round trip
    for (uint32_t unit = wid; unit < n_units; unit += ENC_WAVES) {
      const uint32_t p_begin = unit * hp.unit_pos, p_end = min(p_begin + hp.unit_pos, n);
      float ua[NI];                                              // the unit's own sum first, as encode_partial_kernel forms it
#pragma unroll
      for (int i = 0; i < NI; ++i) ua[i] = 0.f;
      for (uint32_t q0 = p_begin; q0 < p_end; q0 += WAVE) {
        const uint32_t p = q0 + lane;
        uint32_t item = 0;
        int keep = 0;
        if (p < p_end) {
          item = row[p];
          keep = cdae_keep(cdae_rng_draw(key_c, (uint64_t)cidx * n_rng + p_rng0 + p), hp.keep_thr);
        }
        unsigned long long mask = __ballot(keep);
        while (mask) {
          float v[UN][NI];
#pragma unroll
          for (int j = 0; j < UN; ++j) {
            if (mask) {                                          // wave-uniform
              const int src = __ffsll((long long)mask) - 1;
              mask &= mask - 1;
              const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)item, src);
              vload<NI>(v[j], W + (size_t)it * hp.Kp + lo);
            } else {
#pragma unroll
              for (int i = 0; i < NI; ++i) v[j][i] = 0.f;
            }
          }
#pragma unroll
          for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) ua[i] += v[j][i];
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] += ua[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) part[wid][lo + i] = acc[i];
  __syncthreads();
  if (wid != 0) return;
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  const uint32_t nw = min(n_units, (uint32_t)ENC_WAVES);
  for (uint32_t w = 0; w < nw; ++w)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] += part[w][lo + i];
  if (raw_out) {
    vstore<NI>(raw_out + (size_t)slot * hp.Kp + lo, acc);
    uint32_t blk = 1;
    if (raw_a) {
#pragma unroll
      for (int i = 0; i < NI; ++i) wu[i] = cdae_xa::own_row_contribution(own, wu[i]);
      vstore<NI>(raw_out + ((size_t)(blk++) * nb + slot) * hp.Kp + lo, wu);
    }
    if (raw_b) {
#pragma unroll
      for (int i = 0; i < NI; ++i) uu[i] = cdae_xa::own_row_contribution(own, uu[i]);
      vstore<NI>(raw_out + ((size_t)blk * nb + slot) * hp.Kp + lo, uu);
    }
    trace_end(hp, 2, slot, t0);
    return;
  }
  if (hp.linear_function) {                                      // h1 = Uu[u] (.) h1   cdae.hpp:382-384
    if (Ssum) vstore<NI>(Ssum + (size_t)slot * hp.Kp + lo, acc);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] *= uu[i];
  }
  float z[NI], dz[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float h = fmaf(acc[i], hp.scale, bb[i]);
    if (hp.user_factor) h += wu[i];
    const float zz = activate(hp, h);
    const bool live = lo + i < hp.K;                             // pad elements of z must be 0
    z[i] = live ? zz : 0.f;
    dz[i] = live ? act_deriv(hp, zz) : 0.f;
  }
  vstore<NI>(Z + (size_t)slot * hp.Kp + lo, z);
  if (Zb) store_z_bf16<NI>(z, slot, lo, hp.Kp, Bp, Zb, ZTb);
  vstore<NI>(Dz + (size_t)slot * hp.Kp + lo, dz);
#pragma unroll
  for (int i = 0; i < NI; ++i) z[i] = 0.f;
  vstore<NI>(HGzero + (size_t)slot * hp.Kp + lo, z);
  trace_end(hp, 2, slot, t0);
}

// ------------------------------------------------------------------------------------------------
// K3  decode, row-major: positives loop cdae.hpp:225-260 and negatives loop cdae.hpp:262-293, for all
// users of the batch at once.  One wavefront (or one 16-lane group of it, see below) owns item row j: D[j],
// D_ag[j], b'[j], b'_ag[j] stay in registers while it walks the row's examples in user order:
//   y = D[j].z_u + b'[j] (cdae.hpp:227/263, 418-426);  g = loss'(y, t) (:228/:265);
//   b'[j] step (:230-237/:267-274);
//   row step grad = g z_u + lambda D[j] (:241-246,:252-257/:278-283,:286-291), or, when j is one of u's
//   kept inputs in tied mode, no step: g is deferred into the merged input-row step (:249-250).
// g is written to G[e] (user-major) for the hidden-gradient gather (K4) and the input rows (K5).
// hg_u += g D[j] (:240,:248/:277,:285) is NOT accumulated here with one atomic per element (measured:
// 150 G atomics/s saturates the L2 atomic units, 3.7 ms per 4096-user batch); K4 gathers it from the
// batch-start snapshot D0 instead, and this kernel only adds the exact correction for a user's own
// duplicate negatives — g * (row now - row at the user's first visit) — which is what makes
// batch_users == 1 reproduce the reference in exact arithmetic.  segment_kernel marks those (rare) examples
// in the example word, so the common path carries no bookkeeping for them.
// D = V when asymmetric else W.  Rows are visited in `item_order` (popular rows first: their example
// chains are the longest and bound the kernel).
// Memory pipeline: the row's example words are fetched 64 at a time (one coalesced 512-byte load, next
// chunk in flight), broadcast with v_readlane; z_u rows run PF examples ahead in a register ring; g is
// parked in lane (p mod 64) of one VGPR and written once per chunk.  The inner loop therefore issues
// loads only, which is what lets s_waitcnt vmcnt(N) be counted instead of drained (on gfx9 stores share
// the counter and complete out of order with loads).
constexpr int WAIT_VM0 = 0x0F70;   // s_waitcnt vmcnt(0) (expcnt 7, lgkmcnt 15 = don't care), gfx9 encoding

#define CDAE_DECODE_PARAMS                                                                                         \
  const uint32_t *__restrict__ item_order, const uint32_t *__restrict__ seg_begin,                                  \
      const uint32_t *__restrict__ seg_end, const uint64_t *__restrict__ sorted_val, const float *__restrict__ Z,  \
      float *__restrict__ D, float *__restrict__ D_ag, float *__restrict__ bp, float *__restrict__ bp_ag,           \
      float *__restrict__ HGcorr, float *__restrict__ G, float *__restrict__ D0, uint32_t *__restrict__ touched,    \
      const uint32_t *__restrict__ dup_of_pos, float *__restrict__ dup_corr
#define CDAE_DECODE_PASS \
  item_order, seg_begin, seg_end, sorted_val, Z, D, D_ag, bp, bp_ag, HGcorr, G, D0, touched, dup_of_pos, dup_corr

// One row per wavefront; lane holds NI contiguous floats of the row.
// BIAS_IN_PAD (K < Kp, e.g. K = 200 or 50): b'[j] rides in the last pad element of the row registers with a
// constant 1 as its "z": its AdaGrad step grad = g*1 + lambda*b' (cdae.hpp:230-237) is then the row step's own
// arithmetic and y = D[j].z + b'[j] needs no separate add — the per-example chain loses the scalar bias
// recurrence (two transcendentals).  Memory images of D and D0 keep their pad elements 0.
// g_park / park_cap: wave-private LDS words where g is parked until the row is finished (nullptr / 0: written out per 64-example
// chunk, which costs a store + full vmcnt drain — ~1 us of the row's serial chain — per chunk)
// FUSED: the launch that also gathers (decode_gather_kernel) — G, the D0 row and the correction rows are written through (sc1).
// late: rows below late.late_rows leave their g in late.Ghot and one correction row per duplicate run (late.hotdup), see DecodeLate.
// ref_lds: 2 NI x 64 wave-private LDS words for the rarely used duplicate-run state (the row at the user's first visit, the run's
// corrections so far), or nullptr: kept in registers (the K > 256 launch, which has no LDS).
template <int NI, int LOSS, bool ADAGRAD, bool BIAS_IN_PAD, bool FUSED = false, bool REF_IN_LDS = false>
__device__ __forceinline__ void decode_row64(HyperParams hp, const uint32_t rank, float* __restrict__ g_park, const uint32_t park_cap,
                                             const DecodeLate late, CDAE_DECODE_PARAMS, float* __restrict__ ref_lds = nullptr) {
  const uint32_t lane = threadIdx.x % WAVE;
  if (rank >= hp.num_items) return;
  const bool is_late = rank < late.late_rows;                     // wave-uniform
  const uint32_t item = item_order[rank];
  const uint32_t beg = seg_begin[rank], end = seg_end[rank];      // (segment tables indexed by rank: read beside item_order[rank])
  hp.loss_type = LOSS;                 // compile-time specialisation of the per-example branches
  hp.adagrad = ADAGRAD;
#ifdef CDAE_DECODE_TIMING   // developer aid (tools/decode_timeline.py): s_memtime stamps of one row's timeline
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(touched);
  int dbg_n = 0;
#define CDAE_STAMP() do { if (rank == hp.debug_rank && lane == 0 && dbg_n < 60) dbg[dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define CDAE_STAMP() do {} while (0)
#endif
  CDAE_STAMP();
  const uint32_t lo = lane * NI;
  const bool tied = !hp.asymmetric;
  // the row is requested beside its segment bounds (one round trip of the chain's prologue instead of two); a row without
  // examples leaves here with nothing written
  float w[NI], a[NI];
  // (REF_IN_LDS: element i of this lane at ref[64 i]; else registers, one "word" apart)
  float ref_regs[REF_IN_LDS ? 1 : 2 * NI];
  float* const wref = REF_IN_LDS ? ref_lds + lane : ref_regs;
  float* const csum = REF_IN_LDS ? ref_lds + 64 * NI + lane : ref_regs + NI;
  constexpr int RS = REF_IN_LDS ? 64 : 1;                          // stride between a lane's elements
  vload<NI>(w, D + (size_t)item * hp.Kp + lo);
  vload<NI>(a, D_ag + (size_t)item * hp.Kp + lo);
  float bias = bp[item], bias_ag = bp_ag[item];
  if (beg == end) return;
  // batch-start snapshot of this row for the hidden-gradient gather
  if constexpr (FUSED) {
    vstore_sc1<NI>(rows_rsrc(D0), (item * hp.Kp + lo) * 4u, w);
    __builtin_amdgcn_s_waitcnt(WAIT_VM0);            // the row is in memory before any g of it can be (the gather reads the row once it has seen a g)
  } else {
    vstore<NI>(D0 + (size_t)item * hp.Kp + lo, w);
  }
  const bool pad_lane = BIAS_IN_PAD && lane == WAVE - 1;
  const float pad_one = pad_lane ? 1.f : 0.f;
  if (pad_lane) { w[NI - 1] = bias; a[NI - 1] = bias_ag; }
  // (a run's first example sets wref before its duplicates read it: no initial value needed)
  // late rows: ONE correction row per run of a user's duplicates — the sum of the run's corrections so far, re-written at every
  // duplicate into the row of the run's first one (hidden_finish_kernel adds it by (user, late row), late.hotdup)
  uint32_t run_di = DUP_NONE;

#ifndef CDAE_DECODE_PF
#define CDAE_DECODE_PF 8
#endif
  constexpr int PF = CDAE_DECODE_PF;   // z rows in flight per wavefront (must divide 64)
  // cur / nxt: this lane's example word (slot | flags), example index and z-row byte offset of the current / next
  // 64-example chunk; the chunk after that is in flight (`far`).  The byte offset is computed once per chunk by all 64
  // lanes, so the per-example address is a v_readlane + a scalar add onto the Z base (32-bit: the batch's Z is at
  // most 4 GiB).
  const uint32_t row_bytes = hp.Kp * 4u;
  const char* Zb = reinterpret_cast<const char*>(Z) + (size_t)lo * 4u;
  const uint64_t v0 = beg + lane < end ? sorted_val[beg + lane] : 0ull;
  const uint64_t v1 = beg + WAVE + lane < end ? sorted_val[beg + WAVE + lane] : 0ull;
  uint32_t cur_w = (uint32_t)v0, cur_e = (uint32_t)(v0 >> 32), cur_o = (cur_w & SLOT_MASK) * row_bytes;
  uint32_t nxt_w = (uint32_t)v1, nxt_e = (uint32_t)(v1 >> 32), nxt_o = (nxt_w & SLOT_MASK) * row_bytes;
  float z[PF][NI];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
#pragma unroll
    for (int i = 0; i < NI; ++i) z[j][i] = 0.f;
    if (beg + j < end) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)cur_o, j);
      vload<NI>(z[j], reinterpret_cast<const float*>(Zb + off));
    }
  }
  float gbuf = 0.f;
  uint32_t c0 = beg;
  uint32_t look = 0;                   // z offsets of the examples PF ahead: lanes >= PF this chunk's, lanes < PF the next chunk's
  float cur_t = 0.f;                   // this lane's example target (1 positive / 0 negative) of the current chunk

  // One example of the row.  FAST (compile-time): the straight-line form used while the row is long — no duplicate in
  // the group of PF, PF more examples behind it — where the deferred-input case is a 0/1 factor on the step instead of a
  // branch and the look-ahead address is one v_readlane; the serial chain of a popular row (hundreds of examples per
  // batch) is what bounds the launch, and each taken branch costs it an instruction-buffer refill.
  auto example = [&](auto fast_tag, const int t, const uint32_t idx) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur_w, idx);
    if (BIAS_IN_PAD) z[t][NI - 1] += pad_one;                  // the bias element's "z" is 1 (Z's pad elements are 0)
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i += 2) {
      d0 = fmaf(w[i], z[t][i], d0);
      if (i + 1 < NI) d1 = fmaf(w[i + 1], z[t][i + 1], d1);
    }
    float y = wave_sum(d0 + d1);
    if (!BIAS_IN_PAD) y += bias;
    const float tgt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur_t), idx));
    const float g = loss_grad(hp.loss_type, y, tgt);
    if (!BIAS_IN_PAD) ada_step(hp, bias, bias_ag, fmaf(hp.lambda, bias, g));
    gbuf = lane == idx ? g : gbuf;
    const bool deferred = (word & INPUT_BIT) && tied;          // cdae.hpp:249-250: the row step waits for the input-row merge
    if (FAST) {
      // two straight-line bodies behind one wave-uniform branch: a deferred example (a kept input of its user, ~40 % of a
      // popular row's examples) only steps b'
      if (!deferred) {
#pragma unroll
        for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(g, z[t][i], hp.lambda * w[i]));
      } else if (BIAS_IN_PAD) {
        float bw = w[NI - 1], ba = a[NI - 1];
        ada_step(hp, bw, ba, fmaf(hp.lambda, bw, g));
        if (pad_lane) { w[NI - 1] = bw; a[NI - 1] = ba; }
      }
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)look, (idx + PF) & 63u);
      vload<NI>(z[t], reinterpret_cast<const float*>(Zb + off));
    } else {
      if (word & (DUP_PREV_BIT | DUP_NEXT_BIT)) {              // duplicate negative of the same user (rare, wave-uniform)
        if (word & DUP_PREV_BIT) {
          // g * (row now - row at the user's first visit): a plain row store into the correction buffer (the
          // gather adds it to hg_u); fire-and-wait atomics here cost ~25 us per duplicate (measured)
          uint32_t di = dup_of_pos[c0 + idx];
          float corr[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) corr[i] = (pad_lane && i == NI - 1) ? 0.f : g * (w[i] - wref[RS * i]);
          if (is_late && di != DUP_NONE) {                     // (wave-uniform) the run's corrections so far, in the run's first row
            if (run_di == DUP_NONE) {
              run_di = di;
#pragma unroll
              for (int i = 0; i < NI; ++i) csum[RS * i] = corr[i];
              if (lane == 0) late.hotdup[(size_t)(word & SLOT_MASK) * LATE_MAX + rank] = di;
            } else {
              di = run_di;
#pragma unroll
              for (int i = 0; i < NI; ++i) { corr[i] += csum[RS * i]; csum[RS * i] = corr[i]; }
            }
          }
          if (di != DUP_NONE) {
            if constexpr (FUSED) vstore_sc1<NI>(rows_rsrc(dup_corr), (di * hp.Kp + lo) * 4u, corr);
            else vstore<NI>(dup_corr + (size_t)di * hp.Kp + lo, corr);
          } else {                                             // correction buffer full: slow path
            float* hc = HGcorr + (size_t)(word & SLOT_MASK) * hp.Kp + lo;
#pragma unroll
            for (int i = 0; i < NI; ++i) unsafeAtomicAdd(hc + i, corr[i]);
          }
          __builtin_amdgcn_s_waitcnt(WAIT_VM0);                // keep the loop's VMEM stream loads-only
        } else {                                               // first of a run: remember the row at the user's first visit
#pragma unroll
          for (int i = 0; i < NI; ++i) wref[RS * i] = w[i];
          run_di = DUP_NONE;
        }
      }
      if (!deferred) {
#pragma unroll
        for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(g, z[t][i], hp.lambda * w[i]));
      } else if (BIAS_IN_PAD) {                                // deferred row step: b' still steps now
        float bw = w[NI - 1], ba = a[NI - 1];
        ada_step(hp, bw, ba, fmaf(hp.lambda, bw, g));
        if (pad_lane) { w[NI - 1] = bw; a[NI - 1] = ba; }
      }
    }
  };

  // Speculative software pipeline of the straight-line body (rows whose b' is a scalar, i.e. !BIAS_IN_PAD — the popular rows of
  // decode_hybrid_kernel).  The chain of a popular row is what bounds the launch, and ~47 % of its examples are DEFERRED
  // (positives that are kept inputs of their user, cdae.hpp:249-250): they step b' but leave the row alone.  So the dot
  // product + wave reduction of example i+1 is issued against the row as it stands while example i's scalar chain
  // (y -> loss' -> b' step) is still in flight; if example i turns out to step the row, the dot is simply redone after the
  // step (the wasted instructions fill stall slots of the dependent chain).  A deferred example then costs the scalar chain
  // only (~11 dependent instructions instead of ~23).  Same arithmetic, same order: results are bit-identical.
  float s_carry = 0.f;                 // reduced dot of the NEXT example with the current row, valid when s_valid
  bool s_valid = false;                // wave-uniform
  auto row_dot = [&](const float (&zz)[NI]) -> float {
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i += 2) {
      d0 = fmaf(w[i], zz[i], d0);
      if (i + 1 < NI) d1 = fmaf(w[i + 1], zz[i + 1], d1);
    }
    return wave_sum(d0 + d1);
  };
  // A single wavefront issues about one instruction every 9 cycles on this chain (535 cycles per example measured on the most
  // popular row ALONE on its SIMD, for ~35 instructions of a deferred example and ~80 of a stepping one — hand-interleaving
  // the independent chains changed nothing, profiles/r02_main_stream.txt), so the popular rows' serial chain is shortened by
  // issuing fewer instructions:
  //  * an example that is DEFERRED (known from its word before anything is computed) overlaps the next example's dot + wave
  //    reduction with its own scalar chain; an example that steps the row computes that dot once, after the step (round 2's
  //    first version speculated for every example and redid the reduction for the 53 % that step);
  //  * the target comes out of the example word on the scalar unit (TARGET_BIT is the bit pattern of 2.0f), the look-ahead z row
  //    is one buffer load with a scalar row offset (no 64-bit address arithmetic), no per-lane target copy per chunk.
  // Same arithmetic, same operands, same order of roundings as loss_grad / ada_step: bit-identical results.
#define CDAE_SB() __builtin_amdgcn_sched_barrier(0)
  const auto zrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z), 0, 0x7FFFFFFF, 0x00020000);
  const int zvoff = (int)(lo * 4u);
  auto z_fetch = [&](float (&zz)[NI], uint32_t soff) {           // soff: wave-uniform byte offset of the user's z row
    if constexpr (NI == 1) {
      zz[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrsrc, zvoff, (int)soff, 0));
    } else if constexpr (NI == 2) {
      const auto raw = __builtin_amdgcn_raw_buffer_load_b64(zrsrc, zvoff, (int)soff, 0);
      float2 q;
      __builtin_memcpy(&q, &raw, sizeof q);
      zz[0] = q.x; zz[1] = q.y;
    } else {
#pragma unroll
      for (int v = 0; v < NI / 4; ++v) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, zvoff + 16 * v, (int)soff, 0);
        float4 q;
        __builtin_memcpy(&q, &raw, sizeof q);
        zz[4 * v] = q.x; zz[4 * v + 1] = q.y; zz[4 * v + 2] = q.z; zz[4 * v + 3] = q.w;
      }
    }
  };
  auto fast_group_spec = [&](const uint32_t j0) {
    float sdot = s_valid ? s_carry : row_dot(z[0]);
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      const uint32_t idx = j0 + t;
      const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur_w, idx);
      const float tgt2 = __builtin_bit_cast(float, word & TARGET_BIT);       // 2.0 (positive) or 0.0, on the scalar unit
      const float (&zn)[NI] = z[(t + 1) % PF];
      float g;
      if ((word & INPUT_BIT) && tied) {
        // deferred (cdae.hpp:249-250): the row does not move.  chain A: y -> g -> b' step | chain B: next example's dot
        float spec;
        const float y = sdot + bias;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; i += 2) {
          d0 = fmaf(w[i], zn[i], d0);
          if (i + 1 < NI) d1 = fmaf(w[i + 1], zn[i + 1], d1);
        }
        float d = d0 + d1;
        CDAE_SB();
        if constexpr (LOSS == 5) {                               // loss_grad: rcp(1 + exp(-y)) - t
          const float e = fast_exp(-y);
          d = dpp_add<0xB1>(d);
          CDAE_SB();
          const float u = 1.f + e;
          d = dpp_add<0x4E>(d);
          CDAE_SB();
          const float sg = fast_rcp(u);
          d = dpp_add<0x141>(d);
          CDAE_SB();
          g = fmaf(-0.5f, tgt2, sg);
          d = dpp_add<0x140>(d);
          CDAE_SB();
        } else {
          g = -2.f * (0.5f * tgt2 - y);
          d = dpp_add<0xB1>(d);
          CDAE_SB();
          d = dpp_add<0x4E>(d);
          d = dpp_add<0x141>(d);
          d = dpp_add<0x140>(d);
          CDAE_SB();
        }
        const float gb = fmaf(hp.lambda, bias, g);               // b' step (cdae.hpp:230-237)
        d = dpp_add<0x142>(d);
        CDAE_SB();
        if constexpr (ADAGRAD) bias_ag = fmaf(gb, gb, bias_ag);
        d = dpp_add<0x143>(d);
        CDAE_SB();
        spec = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 63));
        if constexpr (ADAGRAD) bias = fmaf(-hp.lr * gb, fast_rcp(fast_sqrt(bias_ag) + hp.beta), bias);
        else bias = fmaf(-hp.lr, gb, bias);
        sdot = spec;
      } else {
        const float y = sdot + bias;
        if constexpr (LOSS == 5) g = fmaf(-0.5f, tgt2, fast_rcp(1.f + fast_exp(-y)));
        else g = -2.f * (0.5f * tgt2 - y);
        const float gb = fmaf(hp.lambda, bias, g);
        if constexpr (ADAGRAD) {
          // row step in five stages, the b' step riding between them
          float gr[NI], rr[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) gr[i] = fmaf(g, z[t][i], hp.lambda * w[i]);
#pragma unroll
          for (int i = 0; i < NI; ++i) a[i] = fmaf(gr[i], gr[i], a[i]);
          bias_ag = fmaf(gb, gb, bias_ag);
#pragma unroll
          for (int i = 0; i < NI; ++i) rr[i] = fast_sqrt(a[i]);
          float rb = fast_sqrt(bias_ag);
#pragma unroll
          for (int i = 0; i < NI; ++i) rr[i] += hp.beta;
          rb += hp.beta;
#pragma unroll
          for (int i = 0; i < NI; ++i) rr[i] = fast_rcp(rr[i]);
          rb = fast_rcp(rb);
#pragma unroll
          for (int i = 0; i < NI; ++i) w[i] = fmaf(-hp.lr * gr[i], rr[i], w[i]);
          bias = fmaf(-hp.lr * gb, rb, bias);
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) w[i] = fmaf(-hp.lr, fmaf(g, z[t][i], hp.lambda * w[i]), w[i]);
          bias = fmaf(-hp.lr, gb, bias);
        }
        sdot = row_dot(zn);
      }
      gbuf = lane == idx ? g : gbuf;
      z_fetch(z[t], (uint32_t)__builtin_amdgcn_readlane((int)look, (idx + PF) & 63u));
    }
    s_carry = sdot;
    s_valid = true;
  };
#undef CDAE_SB

  CDAE_STAMP();
  for (; c0 < end; c0 += WAVE) {
    // chunk after next: in flight for a whole chunk before anything reads it
    const uint32_t qf = c0 + 2 * WAVE + lane;
    const uint64_t far = qf < end ? sorted_val[qf] : 0ull;
    look = lane < (uint32_t)PF ? nxt_o : cur_o;
    cur_t = (cur_w & TARGET_BIT) ? 1.f : 0.f;
    const unsigned long long dupmask = __ballot((cur_w & (DUP_PREV_BIT | DUP_NEXT_BIT)) != 0u);
    const uint32_t cnt = min((uint32_t)WAVE, end - c0);
    gbuf = 0.f;
    for (uint32_t j0 = 0; j0 < cnt; j0 += PF) {
      const bool fast = c0 + j0 + 2 * PF <= end && ((dupmask >> j0) & ((1ull << PF) - 1ull)) == 0ull;
      if (fast) {
        if constexpr (!BIAS_IN_PAD) {
          fast_group_spec(j0);
        } else {
#pragma unroll
          for (int t = 0; t < PF; ++t) example(std::true_type{}, t, j0 + t);
        }
      } else {
        s_valid = false;
#pragma unroll
        for (int t = 0; t < PF; ++t) {
          const uint32_t idx = j0 + t;
          if (idx < cnt) example(std::false_type{}, t, idx);   // wave-uniform
          // refill ring slot t with the example PF ahead (clamped to the row's last example; the value is never
          // consumed past the end).  Issued after the slot's last use so that the load lands in the same registers
          // and the compiler can wait with a counted vmcnt instead of draining.
          const uint32_t rel = min(c0 + idx + PF, end - 1u) - c0;
          const uint32_t off = rel < (uint32_t)WAVE ? (uint32_t)__builtin_amdgcn_readlane((int)cur_o, rel & 63u)
                                                    : (uint32_t)__builtin_amdgcn_readlane((int)nxt_o, rel & 63u);
          vload<NI>(z[t], reinterpret_cast<const float*>(Zb + off));
        }
      }
    }
    CDAE_STAMP();
    if (c0 - beg + WAVE <= park_cap) {                           // wave-uniform: parked in LDS, written out after the row's last example
      g_park[c0 - beg + lane] = gbuf;
    } else {
      if (lane < cnt) store_f32<FUSED>(G + cur_e, gbuf);
      if (c0 + WAVE < end || is_late) __builtin_amdgcn_s_waitcnt(WAIT_VM0);   // (nothing follows the last chunk but the row's own stores — and a late row's read-back below)
    }
    CDAE_STAMP();
    cur_w = nxt_w; cur_e = nxt_e; cur_o = nxt_o;
    nxt_w = (uint32_t)far; nxt_e = (uint32_t)(far >> 32); nxt_o = (nxt_w & SLOT_MASK) * row_bytes;
  }
  {
    // the parked g: example ids are re-read (coalesced).  A late row also leaves g in Ghot[user slot][rank] — for a run of one user's
    // duplicates ONE entry, the sum of the run's g (the run's first example writes it) — for hidden_finish_kernel's dense sum.
    const uint32_t n_row = end - beg, n_parked = min(n_row, park_cap & ~63u);
    for (uint32_t q = lane; q < (is_late ? n_row : n_parked); q += WAVE) {
      const uint64_t v = sorted_val[beg + q];
      const uint32_t e = (uint32_t)(v >> 32), word = (uint32_t)v;
      float g;
      if (q < n_parked) { g = g_park[q]; store_f32<FUSED>(G + e, g); }
      else g = load_f32_sc1(G + e);                               // (a row longer than the parking space: written per chunk above)
      if (is_late && !(word & DUP_PREV_BIT)) {
        if (word & DUP_NEXT_BIT) {
          for (uint32_t k = q + 1u; k < n_row; ++k) {
            const uint64_t v2 = sorted_val[beg + k];
            if (!((uint32_t)v2 & DUP_PREV_BIT)) break;
            g += k < n_parked ? g_park[k] : load_f32_sc1(G + (uint32_t)(v2 >> 32));
          }
        }
        late.Ghot[(size_t)(word & SLOT_MASK) * LATE_MAX + rank] = g;
      }
    }
  }
  if (BIAS_IN_PAD) {
    if (pad_lane) {
      bp[item] = w[NI - 1];
      bp_ag[item] = a[NI - 1];
      w[NI - 1] = 0.f; a[NI - 1] = 1.f;                     // pad images in memory: weight 0, accumulator 1
    }
  } else if (lane == 0) {
    bp[item] = bias;
    bp_ag[item] = bias_ag;
  }
  vstore<NI>(D + (size_t)item * hp.Kp + lo, w);
  vstore<NI>(D_ag + (size_t)item * hp.Kp + lo, a);
#ifdef CDAE_DECODE_TIMING
  CDAE_STAMP();
  if (rank == hp.debug_rank && lane == 0) dbg[63] = (unsigned long long)dbg_n | ((unsigned long long)(end - beg) << 32);
#else
  if (lane == 0 && touched) touched[item] = 1u;
#endif
#undef CDAE_STAMP
}

template <int NI, int LOSS, bool ADAGRAD, bool BIAS_IN_PAD>
__global__ void __launch_bounds__(256)
decode_rows_kernel(HyperParams hp, CDAE_DECODE_PARAMS) {
  const uint32_t rank = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE);
  decode_row64<NI, LOSS, ADAGRAD, BIAS_IN_PAD>(hp, rank, nullptr, 0u, DecodeLate{nullptr, nullptr, 0u}, CDAE_DECODE_PASS);
}

// ------------------------------------------------------------------------------------------------
// K3 (K <= 256)  FOUR item rows per wavefront, one per 16-lane group (= one DPP row).
// The rows below the popular ones carry 98 % of the examples; what they cost is instruction issue (PMC: ~34 M
// instructions per launch with one row per wavefront, ~22 M here, at 5-6 cycles per instruction per SIMD;
// profiles/r01_decode_bisect.txt), a third of it (wave reduction, sigmoid, flag decoding, ring bookkeeping) independent of
// K, and a K = 200 row fills only 50 of 64 lanes x 4 elements in the one-row layout.  Here a row is held as
// NV float4 pieces (lane l of the group: elements 64 v + 4 l .. + 3) plus NT tail scalars (elements 64 NV + l + 16 i),
// so K = 200 is 3 x float4 + 1 scalar = 13 registers with 200 of 208 slots used; the reduction is four intra-row DPP
// steps with no read-back, and every per-example instruction serves four rows.  The last tail slot (lane 15 of tail
// register NT-1, element 64 NV + 16 NT - 1 >= K) carries b' as in BIAS_IN_PAD above; NT == 0 (K a multiple of 64)
// keeps b' as a scalar.  Same arithmetic, same order of examples inside a row, same G / HGcorr / D0 side effects as
// decode_row64.  Each group walks its own segment (groups of one wavefront hold neighbouring popularity ranks, i.e.
// segments of similar length); example words are fetched 64 per group at a time (lane l holds examples l, l+16, l+32,
// l+48) and broadcast inside the group with ds_bpermute; z rows run PF examples ahead; g is parked in four VGPRs and
// stored once per 64 examples so the loop issues loads only (counted vmcnt, see above).
template <int NV, int NT>
__device__ __forceinline__ void row16_load(float (&r)[4 * NV + NT], const float* __restrict__ base, uint32_t l) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float4 q = *reinterpret_cast<const float4*>(base + 64 * v + 4 * l);
    r[4 * v] = q.x; r[4 * v + 1] = q.y; r[4 * v + 2] = q.z; r[4 * v + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) r[4 * NV + i] = base[64 * NV + l + 16 * i];
}
template <int NV, int NT>
__device__ __forceinline__ void row16_store(float* __restrict__ base, const float (&r)[4 * NV + NT], uint32_t l) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#ifdef CDAE_NT_STORES
    const cdae_f4v q = {r[4 * v], r[4 * v + 1], r[4 * v + 2], r[4 * v + 3]};
    __builtin_nontemporal_store(q, reinterpret_cast<cdae_f4v*>(base + 64 * v + 4 * l));
#else
    *reinterpret_cast<float4*>(base + 64 * v + 4 * l) = make_float4(r[4 * v], r[4 * v + 1], r[4 * v + 2], r[4 * v + 3]);
#endif
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) base[64 * NV + l + 16 * i] = r[4 * NV + i];
}

// the same row image written through (sc1): the fused launch's D0 and correction rows (read by gather wavefronts of the same launch)
template <int NV, int NT>
__device__ __forceinline__ void row16_store_sc1(float* __restrict__ matrix, uint32_t row_elem, const float (&r)[4 * NV + NT], uint32_t l) {
  const cdae_rsrc rs = rows_rsrc(matrix);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    cdae_b128 q; const float t[4] = {r[4 * v], r[4 * v + 1], r[4 * v + 2], r[4 * v + 3]};
    __builtin_memcpy(&q, t, sizeof q);
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)((row_elem + 64u * v + 4u * l) * 4u), 0, AUX_SC1);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, r[4 * NV + i]), rs, (int)((row_elem + 64u * NV + l + 16u * i) * 4u), 0, AUX_SC1);
}

// Wave-private LDS of decode_rows16 (words): example words and example ids of the current and the next 64-example chunk of each
// of the four groups, and the parked g of the current chunk.  (Round 6 tried the rows-at-first-visit state of both row roles in LDS as
// well — 13-16 registers per lane, wanted for an occupancy of four — and took it back: 36 KiB per workgroup instead of 20 left no
// room on a CU for bucket_sort_kernel's 88 KiB workgroups beside three of this launch's, and the prep chain's sort went 56 -> 78 us.)
constexpr uint32_t ROWS16_LDS_WORDS = 4u * 128u + 4u * 128u + 4u * 64u;
constexpr uint32_t ROW64_PARK_WORDS = ROWS16_LDS_WORDS;          // the same words as decode_row64 uses them: g parked until the row's end

template <int NV, int NT, int LOSS, bool ADAGRAD, bool FUSED = false>
__device__ __forceinline__ void decode_rows16(HyperParams hp, const uint32_t rank0, uint32_t* __restrict__ lds, CDAE_DECODE_PARAMS) {
  constexpr int GRP = 16, NE = 4 * NV + NT;
  constexpr bool HAS_PAD = NT > 0;
  const uint32_t lane = threadIdx.x % WAVE, l = lane & (GRP - 1), sub = lane / GRP;
  const uint32_t rank = rank0 + sub;
  const bool row_ok = rank < hp.num_items;
  const uint32_t item = row_ok ? item_order[rank] : 0u;
  const uint32_t beg = row_ok ? seg_begin[rank] : 0u, end = row_ok ? seg_end[rank] : 0u;    // (rank-indexed tables)
  const uint32_t n = end - beg;
  // longest segment of the wavefront's four groups (wave-uniform loop bound)
  const uint32_t nmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)n, 0), (uint32_t)__builtin_amdgcn_readlane((int)n, 16)),
                            max((uint32_t)__builtin_amdgcn_readlane((int)n, 32), (uint32_t)__builtin_amdgcn_readlane((int)n, 48)));
  if (nmax == 0) return;
  hp.loss_type = LOSS;
  hp.adagrad = ADAGRAD;
#ifdef CDAE_DECODE_TIMING   // stamps: start | loop start | every 16 steps | end   (rank0 == debug_rank)
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(touched);
  int dbg_n = 0;
#define CDAE_STAMP() do { if (rank0 == hp.debug_rank && lane == 0 && dbg_n < 60) dbg[dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define CDAE_STAMP() do {} while (0)
#endif
  CDAE_STAMP();
  const bool tied = !hp.asymmetric;
  const bool pad_lane = HAS_PAD && l == GRP - 1;
  const float pad_one = pad_lane ? 1.f : 0.f;

  float w[NE], a[NE], wref[NE];
  const size_t row_off = (size_t)item * hp.Kp;
  if (n) {
    row16_load<NV, NT>(w, D + row_off, l);
    row16_load<NV, NT>(a, D_ag + row_off, l);
    // batch-start snapshot of the row (hidden-gradient gather)
    if constexpr (FUSED) row16_store_sc1<NV, NT>(D0, (uint32_t)row_off, w, l);
    else row16_store<NV, NT>(D0 + row_off, w, l);
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) { w[i] = 0.f; a[i] = 1.f; }
  }
  float bias = n ? bp[item] : 0.f, bias_ag = n ? bp_ag[item] : 1.f;
  if (pad_lane) { w[NE - 1] = bias; a[NE - 1] = bias_ag; }
#pragma unroll
  for (int i = 0; i < NE; ++i) wref[i] = w[i];

#ifndef CDAE_DECODE16_PF
#define CDAE_DECODE16_PF 4
#endif
  constexpr int PF = CDAE_DECODE16_PF;
  static_assert(128 % PF == 0 && 64 % PF == 0, "ring offsets are immediates inside one PF-group");
  // Example words live in LDS: per group a ring of two 64-example chunks (the current one and the next), staged 64 at a time
  // from sorted_val (lane l of the group fetches examples l, l+16, l+32, l+48 — coalesced), read back as one broadcast
  // ds_read per example; g is parked there too and written out once per chunk, so the loop issues global LOADS only (counted
  // vmcnt, see above) and no per-example select / permute bookkeeping.  Positions past the group's segment hold word 0
  // (user slot 0: a valid z row that is never consumed).
  uint32_t* const wl = lds + sub * 128u;                          // words  [128]
  uint32_t* const el = lds + 512u + sub * 128u;                   // example ids [128]
  float* const gl = reinterpret_cast<float*>(lds + 1024u + sub * 64u);   // parked g [64]
  auto stage_chunk = [&](uint32_t c0) {                           // c0: wave-uniform multiple of 64; every lane of the wavefront takes part
    uint64_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t q = c0 + l + GRP * j;
      v[j] = q < n ? sorted_val[beg + q] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t q = (c0 + l + GRP * j) & 127u;
      wl[q] = (uint32_t)v[j];
      el[q] = (uint32_t)(v[j] >> 32);
    }
  };
  auto flush_chunk = [&](uint32_t c0) {                           // G of the chunk starting at c0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t q = c0 + l + GRP * j;
      if (q < n) store_f32<FUSED>(G + el[q & 127u], gl[q & 63u]);
    }
  };
  // FUSED: g leaves every FLUSH examples instead of every 64 — the gather wavefronts of the same launch take the users in order, and
  // a row's early examples are the early users' (one more store + drain per 16 steps of a wavefront whose SIMD has other work)
#ifndef CDAE_FUSED_FLUSH
#define CDAE_FUSED_FLUSH 64
#endif
  constexpr uint32_t FLUSH = FUSED ? (uint32_t)CDAE_FUSED_FLUSH : 64u;
  auto flush_sixteen = [&](uint32_t c0) {                         // G of the 16 examples starting at c0
    const uint32_t q = c0 + l;
    if (q < n) store_f32<FUSED>(G + el[q & 127u], gl[q & 63u]);
  };
  if constexpr (FUSED) __builtin_amdgcn_s_waitcnt(WAIT_VM0);      // the D0 rows are in memory before any g of them can be (decode_row64)
  stage_chunk(0);
  stage_chunk(64);                                               // (unconditional: the look-ahead reads up to PF words past nmax, and LDS starts as garbage)
  const uint32_t shift = 31u - (uint32_t)__builtin_clz(hp.Kp * 4u);      // row stride is a power of two bytes (Kp = 64 NI)
  // z rows by buffer loads: descriptor (SGPRs) + one 32-bit offset per piece kind — no 64-bit address arithmetic per example
  const auto zrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z), 0, 0x7FFFFFFF, 0x00020000);
  const uint32_t lane_v = 16u * l, lane_t = 256u * NV + 4u * l;  // byte offsets of this lane's float4 pieces / tail scalars in a row
  auto z_load = [&](float (&zz)[NE], uint32_t word) {
    const uint32_t row = (word & SLOT_MASK) << shift;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const auto raw = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, (int)(row + lane_v + 256u * v), 0, 0);
      float4 q;                                                  // (indexing the builtin's vector type directly narrows the load to one dword on hipcc 7.2)
      __builtin_memcpy(&q, &raw, sizeof q);
      zz[4 * v] = q.x; zz[4 * v + 1] = q.y; zz[4 * v + 2] = q.z; zz[4 * v + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
      zz[4 * NV + i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrsrc, (int)(row + lane_t + 64u * i), 0, 0));
  };
  float z[PF][NE];
  uint32_t zw[PF];                                               // word of the example whose z sits in ring slot
#pragma unroll
  for (int r = 0; r < PF; ++r) {
    zw[r] = wl[r];
    z_load(z[r], zw[r]);
  }
  CDAE_STAMP();
  for (uint32_t t0 = 0; t0 < nmax; t0 += PF) {
    if ((t0 & 15u) == 0u && t0) CDAE_STAMP();
    if (FLUSH == 16u && (t0 & 15u) == 0u && t0 != 0u) {
      flush_sixteen(t0 - 16u);
      __builtin_amdgcn_s_waitcnt(WAIT_VM0);
      if ((t0 & 63u) == 0u) stage_chunk(t0 + 64u);
    } else
    if ((t0 & 63u) == 0u && t0 != 0u) {                          // chunk boundary (wave-uniform): write the finished chunk's g, stage the chunk after next
      flush_chunk(t0 - 64u);
      __builtin_amdgcn_s_waitcnt(WAIT_VM0);                      // keep the loop's VMEM stream loads-only
      stage_chunk(t0 + 64u);                                     // (zeros past the segments' ends)
    }
    const uint32_t* const wnext = wl + ((t0 + PF) & 127u);       // words of the examples PF ahead: immediates r inside the group
    float* const gpark = gl + (t0 & 63u);
#pragma unroll
    for (int r = 0; r < PF; ++r) {
      const uint32_t t = t0 + r;                                 // wave-uniform
      const uint32_t word = zw[r];
      if (t < n) {                                               // group-uniform
        if (HAS_PAD) z[r][NE - 1] += pad_one;
        float d0 = 0.f, d1 = 0.f;                                // two chains: the dot is on every example's critical path
#pragma unroll
        for (int i = 0; i < NE; i += 2) {
          d0 = fmaf(w[i], z[r][i], d0);
          if (i + 1 < NE) d1 = fmaf(w[i + 1], z[r][i + 1], d1);
        }
        float dot = d0 + d1;
        dot = dpp_add<0xB1>(dot);                                // quad_perm [1,0,3,2]
        dot = dpp_add<0x4E>(dot);                                // quad_perm [2,3,0,1]
        dot = dpp_add<0x141>(dot);                               // row_half_mirror
        float y = dpp_add<0x140>(dot);                           // row_mirror: every lane of the group holds the row sum
        if (!HAS_PAD) y += bias;
        // target 1 / 0: TARGET_BIT is bit 30, i.e. (word & TARGET_BIT) read as a float is 2.0 or 0.0
        const float tgt = 0.5f * __builtin_bit_cast(float, word & TARGET_BIT);
        const float g = loss_grad(hp.loss_type, y, tgt);
        if (!HAS_PAD) ada_step(hp, bias, bias_ag, fmaf(hp.lambda, bias, g));
        gpark[r] = g;                                            // all 16 lanes of the group write the same word
        if (word & (DUP_PREV_BIT | DUP_NEXT_BIT)) {              // duplicate negative of the same user (rare)
          if (word & DUP_PREV_BIT) {
            const uint32_t di = dup_of_pos[beg + t];
            float corr[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) corr[i] = (pad_lane && i == NE - 1) ? 0.f : g * (w[i] - wref[i]);
            if (di != DUP_NONE) {
              if constexpr (FUSED) row16_store_sc1<NV, NT>(dup_corr, di * hp.Kp, corr, l);
              else row16_store<NV, NT>(dup_corr + (size_t)di * hp.Kp, corr, l);
            } else {                                             // correction buffer full: slow path
              float* hc = HGcorr + (size_t)(word & SLOT_MASK) * hp.Kp;
#pragma unroll
              for (int i = 0; i < NE; ++i) {
                const uint32_t k = i < 4 * NV ? 64u * (i / 4) + 4u * l + (i % 4) : 64u * NV + l + 16u * (i - 4 * NV);
                if (k < hp.K) unsafeAtomicAdd(hc + k, corr[i]);
              }
            }
            __builtin_amdgcn_s_waitcnt(WAIT_VM0);
          } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) wref[i] = w[i];
          }
        }
        if (!(word & INPUT_BIT) || !tied) {
#pragma unroll
          for (int i = 0; i < NE; ++i) ada_step(hp, w[i], a[i], fmaf(g, z[r][i], hp.lambda * w[i]));
        } else if (HAS_PAD) {                                    // deferred row step (cdae.hpp:249-250): b' still steps now
          float bw = w[NE - 1], ba = a[NE - 1];
          ada_step(hp, bw, ba, fmaf(hp.lambda, bw, g));
          if (pad_lane) { w[NE - 1] = bw; a[NE - 1] = ba; }
        }
      }
      // refill ring slot r with the example PF ahead (past the segment: word 0, never consumed)
      zw[r] = wnext[r];
      z_load(z[r], zw[r]);
    }
  }
  // g of the chunk the loop ended in (chunks before it were written at their boundary)
  if (FLUSH == 16u) flush_sixteen((nmax - 1u) & ~15u);
  else flush_chunk((nmax - 1u) & ~63u);
  if (n) {
    if (HAS_PAD) {
      if (pad_lane) { bp[item] = w[NE - 1]; bp_ag[item] = a[NE - 1]; w[NE - 1] = 0.f; a[NE - 1] = 1.f; }
    } else if (l == 0) {
      bp[item] = bias; bp_ag[item] = bias_ag;
    }
    row16_store<NV, NT>(D + row_off, w, l);
    row16_store<NV, NT>(D_ag + row_off, a, l);
#ifndef CDAE_DECODE_TIMING
    if (l == 0 && touched) touched[item] = 1u;
#endif
  }
#ifdef CDAE_DECODE_TIMING
  CDAE_STAMP();
  if (rank0 == hp.debug_rank && lane == 0) dbg[63] = (unsigned long long)dbg_n | ((unsigned long long)nmax << 32);
#endif
#undef CDAE_STAMP
}

// The launch for K <= 256: the `hot_rows` most popular rows (longest chains; static popularity order) take one
// wavefront each at raised priority — their serial chain bounds the launch, and 64 lanes make its per-example latency
// shortest — and all other rows go four to a wavefront.
template <int NV, int NT, int LOSS, bool ADAGRAD>
__global__ void __launch_bounds__(256)
decode_hybrid_kernel(HyperParams hp, uint32_t hot_rows, DecodeLate late, CDAE_DECODE_PARAMS) {
  constexpr int CH = NT == 0 ? NV : NV + 1;                       // 64-element chunks of the row
  constexpr int NI = CH <= 1 ? 1 : (CH <= 2 ? 2 : 4);
  __shared__ uint32_t rows16_lds[4][ROWS16_LDS_WORDS];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE);
  if (wave < hot_rows) {
    if (CDAE_SKIP_ROLE(hp, 4u)) return;
    const unsigned long long t0 = trace_begin(hp);
#ifndef CDAE_HOT_PRIO
#define CDAE_HOT_PRIO 2
#endif
    __builtin_amdgcn_s_setprio(CDAE_HOT_PRIO);
    decode_row64<NI, LOSS, ADAGRAD, false>(hp, wave, reinterpret_cast<float*>(rows16_lds[threadIdx.x / WAVE]), ROW64_PARK_WORDS, late,
                                           CDAE_DECODE_PASS);   // b' as a scalar: the speculative pipeline needs the row untouched by deferred examples
    trace_end(hp, 3, wave, t0);
  } else {
    if (CDAE_SKIP_ROLE(hp, 8u)) return;
    const unsigned long long t0 = trace_begin(hp);
    decode_rows16<NV, NT, LOSS, ADAGRAD>(hp, hot_rows + (wave - hot_rows) * 4u, rows16_lds[threadIdx.x / WAVE], CDAE_DECODE_PASS);
    trace_end(hp, 4, wave, t0);
  }
}

// ------------------------------------------------------------------------------------------------
// K4a  hidden gradient, XCD-partitioned gather:  hg_u = sum_e g_e D0[j_e]     cdae.hpp:240,248,277,285
// D0 is the decoder matrix as it was at batch start (I x Kp fp32, 10.8 MB at ML-10M/K=200: larger than one
// XCD's 4 MiB L2).  Workgroup b serves item partition x = b mod 8 — the dispatcher places consecutive
// workgroups round-robin on the 8 XCDs (a speed assumption only; any placement gives the same result) —
// and gathers, for its 4 users, only the rows with (item mod 8) == x.  Each XCD's L2 then holds 1/8 of
// D0 (1.35 MB) and the 684 row reads per user hit L2 instead of the fabric.  The 8 partial sums per user
// per unit are combined by hidden_finish_kernel.  One wavefront per (unit, partition); ids and g staged 64
// at a time, matching rows compacted with a ballot, 8 row loads in flight.
struct GatherArgs {
  const int64_t* row_ptr; const uint32_t* uptr; uint32_t n_units; uint64_t u0; uint32_t nb;
  const uint32_t* ex_item; const float* G; const float* D0; float* HGpart;
  uint32_t explicit_examples;                    // != 0: one user, one unit, that many examples
  const uint32_t* dup_of_ex; const float* dup_corr; const uint32_t* unit_user;
  uint32_t halves;                               // 1 or 2: wavefronts per (unit, partition), each walking half of the unit's 64-example chunks
  const uint32_t* late_bits; uint32_t late_words;   // bitmap over the items: the late rows (DecodeLate), whose examples are NOT gathered; nullptr: none
  uint32_t* err;                                 // fused launch: raised by a wavefront that gives up waiting for a g
};
constexpr uint32_t GATHER_CAP = 448;             // list entries per wavefront; a 128-example step adds at most 256 (flushed when fewer are free).  448,
                                                 // not 512: three workgroups of the fused launch and one of bucket_sort_kernel (88 KiB) then share a CU's LDS
constexpr uint32_t LATE_BITS_WORDS = 2048;       // LDS copy of the late-row bitmap: item spaces up to 65 536

// every thread of the workgroup: the late-row bitmap into LDS (before any wavefront leaves)
__device__ __forceinline__ void stage_late_bits(uint32_t* lbits, const uint32_t* __restrict__ late_bits, uint32_t late_words) {
  if (late_bits) {
    for (uint32_t i = threadIdx.x; i < late_words; i += blockDim.x) lbits[i] = late_bits[i];
    __syncthreads();
  }
}

// One wavefront: the partial hidden gradient of (unit, item partition `part`[, half]).
// FUSED (decode_gather_kernel): the decode of the same launch is still writing G — a g that reads G_PENDING is waited for (sc1
// polls) — and D0 / correction rows are read past the L1 (sc1), since another CU wrote them during this launch.
template <int NI, bool FUSED>
__device__ __forceinline__ void hidden_gather_role(const HyperParams& hp, const GatherArgs& ga, const uint32_t part, const uint32_t half,
                                                   const uint32_t unit, uint32_t* const lrow, float* const lg,
                                                   const uint32_t* const lbits /* LDS late-row bitmap, or nullptr */) {
  const uint32_t lane = threadIdx.x % WAVE;
  const uint32_t halves = ga.halves;
  const unsigned long long t0 = trace_begin(hp);
  const UnitRef ur = locate_unit(hp.unit_pos, ga.uptr, ga.nb, ga.uptr[0] + unit, ga.unit_user, ga.u0);
  const uint64_t uid = ga.u0 + ur.slot;
  const int64_t r0 = ga.row_ptr[uid];
  const uint32_t n = (uint32_t)(ga.row_ptr[uid + 1] - r0);
  const uint64_t base = (uint64_t)(r0 - ga.row_ptr[ga.u0]) * (1u + hp.num_neg);
  // the unit's examples: positives [p0, p1) then negatives [num_neg*p0, num_neg*p1) (stored after the n positives)
  const uint32_t p0 = ga.explicit_examples ? 0u : ur.p0, p1 = ga.explicit_examples ? n : min(ur.p1, n);
  const uint32_t n_posu = p1 - p0;
  const uint32_t n_negu = ga.explicit_examples ? ga.explicit_examples - n : n_posu * hp.num_neg;
  const uint32_t neg0 = n + p0 * hp.num_neg;
  const uint32_t n_ex = n_posu + n_negu;
  const uint32_t lo = lane * NI;
  // pad lanes (lo >= K, e.g. 14 of 64 at K = 200) re-read lane 0's bytes instead of the row's zero padding: same cache
  // line, no extra L2 traffic (-22 % at K = 200); their sums are discarded at the end
  const uint32_t lo_ld = lo < hp.K ? lo : 0u;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  // Round 5.  Through round 4 a wavefront walked its 64-example chunks one L2 round trip at a time — ballot, up to 12 row loads, wait, FMA —
  // and the launch moved its ~140 MB of rows at 7-9 TB/s, which was taken for the L2s' limit.  It is not: tools/l2_delivery.hip gathers
  // random 1 KiB rows out of an XCD-resident region at 30-32 TB/s once 8+ wavefronts per CU keep 8+ rows in flight each
  // (profiles/r05_l2_delivery.txt).  So the walk is now in two steps: the chunk's rows of this partition (and its duplicate
  // corrections, rare) are COMPACTED into a per-wavefront LDS list in the order the old loop added them — corrections of a chunk in
  // lane order, then its rows in lane order — and the list is drained 2 x UN rows per trip, all of a trip's loads issued before the
  // first FMA.  Same additions in the same order: the partial rows are bit-identical to round 4's.
  // rows in flight per trip (registers: TRIP x NI floats).  The launch of its own must be resident as a whole: <= 96 registers.  The fused
  // launch holds three wavefronts per SIMD whatever they need below 168 registers: twice the rows per trip (round 5 measured 16 rows /
  // 130 registers at 6.6 us per wavefront against 9 us for 8 rows).  The list order, hence every sum, is the same.
  constexpr int TRIP = NI >= 8 ? 4 : (FUSED ? 12 : 8);
  constexpr uint32_t GCAP = GATHER_CAP;
  constexpr uint32_t DUP_ROW = 0x80000000u;                      // list entry: a row of dup_corr (added as it is) instead of D0 (times g)
  const cdae_rsrc d0_rs = rows_rsrc(ga.D0), dc_rs = rows_rsrc(ga.dup_corr);      // (FUSED: sc1 row loads)
  // this lane's (item, g, correction row) of two consecutive 64-example chunks; the next pair is requested before this one is compacted
  struct Meta { uint32_t item[2], di[2]; float g[2]; };
  auto example_of = [&](uint32_t v) -> uint32_t { return v < n_posu ? p0 + v : neg0 + (v - n_posu); };
  auto load_meta = [&](uint32_t c, uint32_t c_end, Meta& m) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const uint32_t v = c + (uint32_t)h2 * WAVE + lane;
      const uint32_t e = example_of(v);
      const bool in = v < c_end;
      m.item[h2] = in ? ga.ex_item[base + e] : 0xFFFFFFFFu;
      if constexpr (FUSED) m.g[h2] = in ? load_f32_sc1(ga.G + base + e) : 0.f;
      else m.g[h2] = in ? ga.G[base + e] : 0.f;
      m.di[h2] = in ? ga.dup_of_ex[base + e] : DUP_NONE;
    }
  };
  // an example of this wavefront: a real item of its partition that is not a late row
  auto is_mine = [&](uint32_t item) -> bool {
    bool mine = item < hp.num_items && (item & 7u) == part;      // (fillers are 0xFFFFFFFF, an item shard's VOID examples num_items)
    if (lbits && mine) mine = ((lbits[item >> 5] >> (item & 31u)) & 1u) == 0u;
    return mine;
  };
  uint32_t head = 0, len = 0;                                    // list entries [head, len) are waiting (wave-uniform)
  auto trip = [&](uint32_t n_rows) {                             // the next n_rows <= TRIP entries: every load issued, then the FMAs in list order
    float vv[TRIP][NI], gg[TRIP];
#pragma unroll
    for (int t = 0; t < TRIP; ++t) {
      if ((uint32_t)t < n_rows) {                                // wave-uniform
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)lrow[head + t]);
        gg[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lg[head + t])));
        if constexpr (FUSED) {
          vload_sc1<NI>(vv[t], (e & DUP_ROW) ? dc_rs : d0_rs, ((e & ~DUP_ROW) * hp.Kp + lo_ld) * 4u);
        } else {
          const float* src = (e & DUP_ROW) ? ga.dup_corr : ga.D0;
          vload<NI>(vv[t], src + (size_t)(e & ~DUP_ROW) * hp.Kp + lo_ld);
        }
      } else {
        gg[t] = 0.f;
#pragma unroll
        for (int k = 0; k < NI; ++k) vv[t][k] = 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < TRIP; ++t)
#pragma unroll
      for (int k = 0; k < NI; ++k) acc[k] = fmaf(gg[t], vv[t][k], acc[k]);          // (a correction row: g = 1, i.e. acc + row exactly)
    head += n_rows;
  };
  // `halves` = 2: two wavefronts share a (unit, partition), each walking half of its 64-example chunks; measured slower in round 2
  // (twice the partial sums); default 1.
  const uint32_t chunks_per = ((n_ex + WAVE - 1) / WAVE + halves - 1) / halves;
  const uint32_t c_begin = half * chunks_per * WAVE, c_end = min(n_ex, c_begin + chunks_per * WAVE);
  Meta cur, nxt;
  load_meta(c_begin, c_end, cur);
  for (uint32_t c0 = c_begin; c0 < c_end; c0 += 2 * WAVE) {
    load_meta(c0 + 2 * WAVE, c_end, nxt);                        // (past the end: fillers, no loads)
    if constexpr (FUSED) {
      // g of a row the decode has not finished yet: wait for it (bounded), re-reading only the lanes that need it
      for (uint32_t spin = 0;; ++spin) {
        const bool pend0 = __builtin_bit_cast(uint32_t, cur.g[0]) == G_PENDING && is_mine(cur.item[0]);
        const bool pend1 = __builtin_bit_cast(uint32_t, cur.g[1]) == G_PENDING && is_mine(cur.item[1]);
        if (!__ballot(pend0 || pend1)) break;
        // (an earlier wavefront or launch has already given up: this handle's parameters are lost, the host will say so — do not make
        // the rest of the epoch wait for that news)
        if (spin == 64u && __hip_atomic_load(ga.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
        if (spin >= FUSED_SPIN_CAP) { if (lane == 0) __hip_atomic_store(ga.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }      // (host memory: a plain store, no PCIe atomic)
        __builtin_amdgcn_s_sleep(24);
        if (pend0) cur.g[0] = load_f32_sc1(ga.G + base + example_of(c0 + lane));
        if (pend1) cur.g[1] = load_f32_sc1(ga.G + base + example_of(c0 + WAVE + lane));
      }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const bool mine = is_mine(cur.item[h2]);
      const bool dup = mine && cur.di[h2] != DUP_NONE;           // duplicate negatives (rare): decode's correction rows, in example order
      const unsigned long long mask = __ballot(mine), dmask = __ballot(dup);
      const uint32_t n_d = (uint32_t)__popcll(dmask), n_r = (uint32_t)__popcll(mask);
      if (dup) { const uint32_t k = len + (uint32_t)__popcll(dmask & below); lrow[k] = cur.di[h2] | DUP_ROW; lg[k] = 1.f; }
      if (mine) { const uint32_t k = len + n_d + (uint32_t)__popcll(mask & below); lrow[k] = cur.item[h2]; lg[k] = cur.g[h2]; }
      len += n_d + n_r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (the wavefront's own LDS writes, in order)
    __builtin_amdgcn_wave_barrier();
    // full trips go out now — their loads fly while the next pair's (item, g) are still on their way; the remainder waits for more
    const bool last = c0 + 2 * WAVE >= c_end;
    while (len - head >= (uint32_t)TRIP) trip((uint32_t)TRIP);
    if (last || len + 4u * WAVE > GCAP) {
      if (len > head) trip(len - head);
      head = 0; len = 0;
      __builtin_amdgcn_wave_barrier();
    }
    cur = nxt;
  }
  if (lo >= hp.K) {
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  }
  vstore<NI>(ga.HGpart + ((size_t)(part * halves + half) * ga.n_units + unit) * hp.Kp + lo, acc);
  trace_end(hp, 5, (unit * 8u + part) * halves + half, t0, n_ex);
}

template <int NI>
__global__ void __launch_bounds__(256)
hidden_gather_kernel(HyperParams hp, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ uptr,
                     uint32_t n_units, uint64_t u0, uint32_t nb, const uint32_t* __restrict__ ex_item,
                     const float* __restrict__ G, const float* __restrict__ D0,
                     float* __restrict__ HGpart /* [8 * halves][n_units][Kp] */,
                     uint32_t explicit_examples /* != 0: one user, one unit, that many examples */,
                     const uint32_t* __restrict__ dup_of_ex, const float* __restrict__ dup_corr,
                     const uint32_t* __restrict__ unit_user,
                     uint32_t halves /* 1 or 2: wavefronts per (unit, partition), each walking half of the unit's 64-example chunks */,
                     const uint32_t* __restrict__ late_bits = nullptr /* the late rows (DecodeLate): skipped here, added by hidden_finish_kernel */,
                     uint32_t late_words = 0) {
  __shared__ uint32_t gl_row[4][GATHER_CAP];
  __shared__ float gl_g[4][GATHER_CAP];
  __shared__ uint32_t lbits[LATE_BITS_WORDS];
  stage_late_bits(lbits, late_bits, late_words);
  const uint32_t part = blockIdx.x & 7u;
  const uint32_t rest = blockIdx.x >> 3;
  const uint32_t half = rest % halves;
  const uint32_t unit = (rest / halves) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  if (unit >= n_units) return;
  const GatherArgs ga{row_ptr, uptr, n_units, u0, nb, ex_item, G, D0, HGpart, explicit_examples, dup_of_ex, dup_corr, unit_user, halves,
                      late_bits, late_words, nullptr};
  hidden_gather_role<NI, false>(hp, ga, part, half, unit, gl_row[threadIdx.x / WAVE], gl_g[threadIdx.x / WAVE], late_bits ? lbits : nullptr);
}

// ------------------------------------------------------------------------------------------------
// K3 + K4a in ONE launch (round 6).  The decode launch is as long as its most popular row's serial chain (~134 examples x 250 ns at
// 256 users per batch) while nine tenths of its other wavefronts are done in half that time, and the hidden-gradient gather that
// follows needs only g — so the gather of everything but the late rows (DecodeLate) runs INSIDE the decode launch, as trailing
// workgroups that are dispatched as the row wavefronts retire and wait, example by example, for the g they need (G_PENDING).
// Workgroups of four wavefronts, in this order of workgroup index (S = the chip's CU count):
//   [0, H)                      popular rows (decode_row64), four per workgroup, one per SIMD;
//   [rS, rS + H), r = 1..rounds BLOCKERS: idle wavefronts that sleep until popular workgroup (index mod S) has finished.  A popular row's
//                               chain keeps its SIMD's VALU ~80 % busy; through round 5 the four-row wavefronts that shared the SIMD
//                               lived as long as the launch — harmless then, but here every gather wavefront would end up waiting for
//                               one of their rows.  Workgroup b runs on the CU that b mod S names while the CUs hold 1 + rounds
//                               workgroups of this launch each (observed placement, a speed assumption only: wherever a blocker
//                               lands it just sleeps);
//   every other index below D   the other rows, four per wavefront (decode_rows16): which group a wavefront takes is the host's table
//                               (FusedGeom::cold_map: the groups dealt over the SIMDs by expected length, longest first);
//   [D, ...)                    gather: four (unit, partition) wavefronts each (hidden_gather_role<FUSED>).
// Workgroups are dispatched in index order (as bucket_sort_kernel assumes): whatever a gather wavefront waits for is running or done.
// Everything one role writes and another reads inside the launch is written through (sc1) and read past the L1 (sc1), so no
// agent-scope fence — a walk of the XCD's whole L2 — is needed (MI355X_MICROARCH.md, inter-workgroup visibility).
struct FusedGeom {
  uint32_t hot_wgs;        // H
  uint32_t stride;         // S
  uint32_t blocked;        // popular workgroups [0, blocked) have blockers (at most FUSED_BLOCK_MAX)
  uint32_t rounds;         // blocker rounds: workgroups of this launch a CU holds, less one
  uint32_t decode_wgs;     // D
  uint32_t hot_target;     // value every popular workgroup's counter reaches when its four wavefronts of THIS launch are done (wraps)
  uint32_t* hot_cnt;       // [hot_wgs] wavefronts finished since the handle was created
  const uint32_t* cold_map; // [decode_wgs][4]: the four-row group of every wavefront of a row workgroup, 0xFFFFFFFF = none (balanced by the host)
};
constexpr uint32_t FUSED_BLOCK_MAX = 16;         // blockers for the workgroups of the 64 most popular rows
constexpr uint32_t FUSED_LDS_WORDS = 4u * 2u * GATHER_CAP + LATE_BITS_WORDS;     // gather role: 22 KiB (the row roles need 4 x ROWS16_LDS_WORDS = 20 KiB)
static_assert(4u * ROWS16_LDS_WORDS <= FUSED_LDS_WORDS, "row roles' LDS");
static_assert(3u * (FUSED_LDS_WORDS + 8u) * 4u + 90376u <= 160u * 1024u, "three workgroups of this launch + one of bucket_sort_kernel per CU");
// is workgroup index b a blocker?
__host__ __device__ inline bool fused_is_blocker(const FusedGeom& g, uint32_t b) {
  return b >= g.stride && b < (1u + g.rounds) * g.stride && b % g.stride < g.blocked;
}

#ifndef CDAE_FUSED_WAVES_PER_SIMD
#define CDAE_FUSED_WAVES_PER_SIMD 3      // (4 = 128 registers: gather wavefronts resident from the start — measured slower, they take issue slots the row roles need)
#endif
template <int NV, int NT, int LOSS, bool ADAGRAD>
__global__ void __launch_bounds__(256, CDAE_FUSED_WAVES_PER_SIMD)
decode_gather_kernel(HyperParams hp, uint32_t hot_rows, FusedGeom geo, DecodeLate late, GatherArgs ga, CDAE_DECODE_PARAMS) {
  constexpr int CH = NT == 0 ? NV : NV + 1;                       // 64-element chunks of the row
  constexpr int NI = CH <= 1 ? 1 : (CH <= 2 ? 2 : 4);
  __shared__ uint32_t fused_lds[FUSED_LDS_WORDS + 8u];
  const uint32_t wg = blockIdx.x;
  const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  if (wg < geo.hot_wgs) {
    // ---- popular rows ----
    const uint32_t row = wg * 4u + wid;
    if (row < hot_rows && !CDAE_SKIP_ROLE(hp, 4u)) {
      const unsigned long long t0 = trace_begin(hp);
      __builtin_amdgcn_s_setprio(CDAE_HOT_PRIO);
      decode_row64<NI, LOSS, ADAGRAD, false, true>(hp, row, reinterpret_cast<float*>(fused_lds + wid * ROWS16_LDS_WORDS), ROW64_PARK_WORDS, late,
                                                   CDAE_DECODE_PASS);
      __builtin_amdgcn_s_setprio(0);
      trace_end(hp, 3, row, t0, hp.trace ? hw_place() : 0u);
    }
    if (threadIdx.x % WAVE == 0) atomicAdd(geo.hot_cnt + wg, 1u);          // (the blockers of this workgroup)
  } else if (wg < geo.decode_wgs) {
    if (fused_is_blocker(geo, wg)) {
      // ---- blocker: hold this CU's slots until the popular workgroup is done ----
      const uint32_t* cnt = geo.hot_cnt + wg % geo.stride;
      const unsigned long long t0 = trace_begin(hp);
      for (uint32_t spin = 0; spin < (1u << 16); ++spin) {
        if ((int32_t)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - geo.hot_target) >= 0) break;
        __builtin_amdgcn_s_sleep(127);
      }
      trace_end(hp, 10, wg * 4u + wid, t0, hp.trace ? hw_place() : 0u);
      return;
    }
    // ---- all other rows, four per wavefront ----
    if (CDAE_SKIP_ROLE(hp, 8u)) return;
    // the table is by SIMD (the host balanced the SIMDs' loads): a wavefront takes the entry of the SIMD it finds itself on.  The four
    // wavefronts of a workgroup sit on the CU's four SIMDs; should two ever share one, the entry nobody claimed goes to the loser.
    uint32_t* const claim = fused_lds + FUSED_LDS_WORDS;
    if (threadIdx.x < 4) claim[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t simd = (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) /* HW_ID.SIMD_ID */) & 3u;
    if (threadIdx.x % WAVE == 0) atomicCAS(&claim[simd], 0u, 1u + wid);
    __syncthreads();
    const uint32_t cs[4] = {claim[0], claim[1], claim[2], claim[3]};
    int mine = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (cs[q] == 1u + wid) mine = q;
    if (mine < 0) {
      uint32_t before = 0, cnt = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (cs[q] && cs[q] - 1u < wid) ++before;
      const uint32_t j = wid - before;                            // my place among the wavefronts that claimed nothing
#pragma unroll
      for (int q = 0; q < 4; ++q) if (!cs[q]) { if (cnt == j) mine = q; ++cnt; }
    }
    const uint32_t g = geo.cold_map[wg * 4u + (uint32_t)mine];
    if (g == 0xFFFFFFFFu) return;
    const unsigned long long t0 = trace_begin(hp);
    decode_rows16<NV, NT, LOSS, ADAGRAD, true>(hp, hot_rows + g * 4u, fused_lds + wid * ROWS16_LDS_WORDS, CDAE_DECODE_PASS);
    trace_end(hp, 4, hot_rows / 4u + g, t0, hp.trace ? hw_place() : 0u);
  } else {
    // ---- hidden-gradient gather of the rows that are not late ----
    uint32_t* const lbits = fused_lds + 4u * 2u * GATHER_CAP;
    stage_late_bits(lbits, ga.late_bits, ga.late_words);
    const uint32_t gw = wg - geo.decode_wgs;
    const uint32_t part = gw & 7u, unit = (gw >> 3) * 4u + wid;
    if (unit >= ga.n_units) return;
    hidden_gather_role<NI, true>(hp, ga, part, 0u, unit, fused_lds + wid * 2u * GATHER_CAP,
                                 reinterpret_cast<float*>(fused_lds + wid * 2u * GATHER_CAP + GATHER_CAP), ga.late_bits ? lbits : nullptr);
  }
}

// hidden_gather_kernel's partial rows of the units [ub, ue) of one user added to hg in a FIXED order (unit-major, then partition):
// deterministic, and the same sum in hidden_finish_kernel (single handle) and hg_raw_kernel (item shard).  Eight partitions (the
// training path): two units' 16 partial rows are in flight per trip (a user has 2-3 units at ML-10M shape; the second unit is clamped
// and added as 0 past the end)
template <int NI>
__device__ __forceinline__ void add_partial_rows(float (&hg)[NI], const float* __restrict__ HGpart, uint32_t n_units, uint32_t n_parts,
                                                 uint32_t ub, uint32_t ue, uint32_t Kp, uint32_t lo) {
  if (n_parts == 8u) {
    const size_t slab = (size_t)n_units * Kp;
    for (uint32_t u = ub; u < ue; u += 2) {
      const bool two = u + 1u < ue;                                // wave-uniform
      const float* p0 = HGpart + (size_t)u * Kp + lo;
      const float* p1 = HGpart + (size_t)(two ? u + 1u : u) * Kp + lo;
      float a0[8][NI], a1[8][NI];
#pragma unroll
      for (int x = 0; x < 8; ++x) vload<NI>(a0[x], p0 + x * slab);
#pragma unroll
      for (int x = 0; x < 8; ++x) vload<NI>(a1[x], p1 + x * slab);
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int i = 0; i < NI; ++i) hg[i] += a0[x][i];
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int i = 0; i < NI; ++i) hg[i] += two ? a1[x][i] : 0.f;
    }
  } else if (n_parts == 16u) {                                     // two wavefronts per (unit, partition): one unit's 16 rows per trip
    const size_t slab = (size_t)n_units * Kp;
    for (uint32_t u = ub; u < ue; ++u) {
      const float* p0 = HGpart + (size_t)u * Kp + lo;
      float a0[16][NI];
#pragma unroll
      for (int x = 0; x < 16; ++x) vload<NI>(a0[x], p0 + x * slab);
#pragma unroll
      for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int i = 0; i < NI; ++i) hg[i] += a0[x][i];
    }
  } else {                                                         // any other count (the full-output slabs): eight rows in flight per trip
    const size_t slab = (size_t)n_units * Kp;
    for (uint32_t u = ub; u < ue; ++u) {
      const float* p0 = HGpart + (size_t)u * Kp + lo;
      for (uint32_t x0 = 0; x0 < n_parts; x0 += 8u) {
        float a0[8][NI];
#pragma unroll
        for (int x = 0; x < 8; ++x) vload<NI>(a0[x], p0 + (size_t)min(x0 + (uint32_t)x, n_parts - 1u) * slab);
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int i = 0; i < NI; ++i) hg[i] = x0 + (uint32_t)x < n_parts ? hg[i] + a0[x][i] : hg[i];     // (wave-uniform; no + 0.f: -0 stays -0)
      }
    }
  }
}

// The late rows' terms of hg_u (DecodeLate; cdae.hpp:240,248,277,285 for the rows hidden_gather_kernel leaves out): lane r of the
// caller holds row r's entries of the user — gh = Ghot[slot][r] (0: no example), hd = hotdup[slot][r], it = item_order[r].
// hg += sum_r gh_r D0[it_r] in rank order, sixteen rows in flight, then the duplicate runs' correction rows in rank order: a fixed
// order of additions (deterministic; the same in hidden_finish_kernel and hg_raw_kernel).
template <int NI>
__device__ __forceinline__ void add_late_rows(float (&hg)[NI], const LateFinish& lf, float gh, uint32_t hd, uint32_t it, uint32_t Kp, uint32_t lo,
                                              unsigned long long ranks = ~0ull /* the ranks this wavefront takes */) {
  constexpr int TR = NI >= 8 ? 8 : 16;
  unsigned long long mask = __ballot(gh != 0.f) & ranks;
  while (mask) {
    float vv[TR][NI], gg[TR];
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      if (mask) {                                                  // wave-uniform
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        gg[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gh), src));
        const uint32_t item = (uint32_t)__builtin_amdgcn_readlane((int)it, src);
        vload<NI>(vv[t], lf.D0 + (size_t)item * Kp + lo);
      } else {
        gg[t] = 0.f;
#pragma unroll
        for (int k = 0; k < NI; ++k) vv[t][k] = 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
      for (int k = 0; k < NI; ++k) hg[k] = fmaf(gg[t], vv[t][k], hg[k]);
  }
  unsigned long long dm = __ballot(hd != DUP_NONE) & ranks;
  while (dm) {                                                     // (rare)
    const int src = __ffsll((long long)dm) - 1;
    dm &= dm - 1;
    const uint32_t di = (uint32_t)__builtin_amdgcn_readlane((int)hd, src);
    float row[NI];
    vload<NI>(row, lf.dup_corr + (size_t)di * Kp + lo);
#pragma unroll
    for (int k = 0; k < NI; ++k) hg[k] += row[k];
  }
}
// the caller's part: this lane's entries of the user's late rows, requested beside everything else the wavefront needs
__device__ __forceinline__ void load_late_entries(const LateFinish& lf, uint32_t slot, uint32_t lane, float& gh, uint32_t& hd, uint32_t& it) {
  gh = 0.f; hd = DUP_NONE; it = 0u;
  if (lane < lf.late_rows) {
    gh = lf.Ghot[(size_t)slot * LATE_MAX + lane];
    hd = lf.hotdup[(size_t)slot * LATE_MAX + lane];
    it = lf.items[lane];
  }
}

// hg of ONE user by the four wavefronts of its workgroup (round 6; through round 5 one wavefront per user walked all of it, and the
// launch was as long as the batch's most active user: 7 us against a median of 3).  The user's partial rows — (unit, partition) pairs in
// unit-major order, numbered 0 .. R-1 — are dealt round-robin: wavefront w adds rows w, w + 4, ... in ascending order, then the late rows
// of rank = w (mod 4) in rank order and their correction rows; the four sums meet in LDS and wavefront 0 adds them to HG[slot] in
// wavefront order.  A fixed order of additions: deterministic, and the same in hidden_finish_kernel and hg_raw_kernel (an item shard
// of one is the single handle bit for bit).  Returns true on wavefront 0, whose `hg` holds the sum; the others are done.
template <int NI>
__device__ __forceinline__ bool user_hg(float (&hg)[NI], float* __restrict__ lds /* [3][64 NI] */, const float* __restrict__ HG_in,
                                        const float* __restrict__ HGpart, uint32_t n_units, uint32_t n_parts, uint32_t ub, uint32_t ue,
                                        const LateFinish& lf, uint32_t slot, uint32_t Kp) {
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE, lo = lane * NI;
  if (wid == 0) vload<NI>(hg, HG_in + (size_t)slot * Kp + lo);           // (requested first: it is added last)
  float late_g; uint32_t late_d, late_i;
  load_late_entries(lf, slot, lane, late_g, late_d, late_i);
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  const uint32_t R = (ue - ub) * n_parts;
  const size_t slab = (size_t)n_units * Kp;
  constexpr uint32_t TR = NI >= 8 ? 4 : 8;
  for (uint32_t r0 = wid; r0 < R; r0 += 4u * TR) {
    float v[TR][NI];
#pragma unroll
    for (uint32_t t = 0; t < TR; ++t) {
      const uint32_t r = min(r0 + 4u * t, R - 1u);                         // (clamped: the surplus is added as 0)
      vload<NI>(v[t], HGpart + (size_t)(r % n_parts) * slab + (size_t)(ub + r / n_parts) * Kp + lo);
    }
#pragma unroll
    for (uint32_t t = 0; t < TR; ++t) {
      const bool on = r0 + 4u * t < R;                                     // wave-uniform
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = on ? acc[i] + v[t][i] : acc[i];
    }
  }
  if (lf.late_rows) add_late_rows<NI>(acc, lf, late_g, late_d, late_i, Kp, lo, 0x1111111111111111ull << wid);
  if (wid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) lds[(wid - 1u) * (64u * NI) + lo + i] = acc[i];
  }
  __syncthreads();
  if (wid) return false;
#pragma unroll
  for (int i = 0; i < NI; ++i) hg[i] += acc[i];
  for (uint32_t w = 0; w < 3u; ++w)
#pragma unroll
    for (int i = 0; i < NI; ++i) hg[i] += lds[w * (64u * NI) + lo + i];
  return true;
}

// K4a'  delta_u = (sum of the 8 partials + duplicate corrections) (.) act'(z_u)   cdae.hpp:305,321,337
//       and the private user-node step Wu[u]                                      cdae.hpp:317-331
// One WORKGROUP per user of the batch (grid = nb): user_hg above, then wavefront 0 finishes.
template <int NI>
__global__ void __launch_bounds__(256)
hidden_finish_kernel(HyperParams hp, const uint32_t* __restrict__ uptr, uint32_t n_units, uint64_t u0, uint32_t nb,
                     const float* __restrict__ HGpart, const float* __restrict__ Dz,
                     float* __restrict__ HG /* in: corrections (or the whole hg), out: delta */,
                     float* __restrict__ Wu, float* __restrict__ Wu_ag,
                     uint32_t n_parts /* slabs of HGpart [n_parts][n_units][Kp] to add to HG (0: HG already holds hg) */,
                     float* __restrict__ Uu, float* __restrict__ Uu_ag, const float* __restrict__ Ssum,
                     float* __restrict__ DELTA_ROWS /* linear_function only: Uu[u] (.) delta_u for the input rows */,
                     const float* __restrict__ Uu_batch = nullptr /* item shard: the batch's gathered Uu rows [nb][Kp] (a user this
                                                                     shard does not own still needs Uu[u] (.) delta for its input rows) */,
                     LateFinish lf = LateFinish{nullptr, nullptr, nullptr, nullptr, nullptr, 0u} /* the late rows' terms, added behind the partial rows */) {
  __shared__ float sums[3 * 64 * NI];
  const uint32_t slot = blockIdx.x;                                 // (grid = nb)
  const uint32_t lane = threadIdx.x % WAVE;
  const bool lead = threadIdx.x < WAVE;
  const unsigned long long t0 = trace_begin(hp);
  const uint64_t uid = u0 + slot;
  const bool own = cdae_xa::owns_user(uid, hp.own_u0, hp.own_u1);   // wave-uniform: the private rows of this user live here
  const uint32_t lo = lane * NI;
  const size_t o = (size_t)slot * hp.Kp + lo;
  float hg[NI], dz[NI], delta[NI];
  // everything the tail needs is requested up front, beside the partial rows (it was two more round trips behind them)
  const size_t ou = (size_t)(own ? uid - hp.own_u0 : 0) * hp.Kp + lo;
  float p[NI], pa[NI];
  if (lead) {
    vload<NI>(dz, Dz + o);
    if (hp.user_factor && own) { vload<NI>(p, Wu + ou); vload<NI>(pa, Wu_ag + ou); }
  }
  const uint32_t ub = n_parts ? uptr[slot] - uptr[0] : 0u, ue = n_parts ? uptr[slot + 1] - uptr[0] : 0u;
  if (!user_hg<NI>(hg, sums, HG, HGpart, n_units, n_parts, ub, ue, lf, slot, hp.Kp)) return;
#pragma unroll
  for (int i = 0; i < NI; ++i) delta[i] = hg[i] * dz[i];
  vstore<NI>(HG + o, delta);
  if (hp.user_factor && own) {
#pragma unroll
    for (int i = 0; i < NI; ++i) ada_step(hp, p[i], pa[i], fmaf(hp.lambda, p[i], delta[i]));
    vstore<NI>(Wu + ou, p);
    vstore<NI>(Wu_ag + ou, pa);
  }
  if (hp.linear_function && !own) {
    // another shard's user: only the input rows' delta, with the gathered Uu[u] (bit-equal to the owner's row before its step)
    float uu[NI], dr[NI];
    vload<NI>(uu, Uu_batch + o);
#pragma unroll
    for (int i = 0; i < NI; ++i) dr[i] = uu[i] * delta[i];
    vstore<NI>(DELTA_ROWS + o, dr);
  } else if (hp.linear_function) {
    // The input rows take Uu[u] (.) delta (cdae.hpp:339) with Uu[u] from BEFORE its own step (:351-357 comes last).
    // Uu_grad = lambda Uu[u] + sum_k delta (.) W[k] (cdae.hpp:295-299, 340 — no input scale there): the kept rows have
    // not moved since the encode inside one user's step, so the sum is delta (.) Ssum with the encode's own row sum.
    float ss[NI], dr[NI];
    vload<NI>(p, Uu + ou);
    vload<NI>(pa, Uu_ag + ou);
    vload<NI>(ss, Ssum + o);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      dr[i] = p[i] * delta[i];
      ada_step(hp, p[i], pa[i], fmaf(hp.lambda, p[i], delta[i] * ss[i]));
    }
    vstore<NI>(DELTA_ROWS + o, dr);
    vstore<NI>(Uu + ou, p);
    vstore<NI>(Uu_ag + ou, pa);
  }
  trace_end(hp, 6, slot, t0);
}

// K4b  hidden bias b: the one parameter every user updates, strictly in user order (cdae.hpp:301-315).
// One thread per coordinate; the recurrence is elementwise, the delta loads run 16 users ahead of it.
// It needs only delta, like the input rows, so it runs as the leading workgroup(s) of input_rows_kernel
// instead of a launch of its own.
// FULL: the full-output callers (same chain; the flag only names the call site).
template <bool ADAGRAD, bool FULL = false>
__device__ __forceinline__ void hidden_bias_role(HyperParams hp, uint32_t k, uint32_t nb,
                                                 const float* __restrict__ DELTA, float* __restrict__ b,
                                                 float* __restrict__ b_ag) {
  if (k >= hp.Kp || nb == 0) return;
  hp.adagrad = ADAGRAD;
  float p = b[k], acc = b_ag[k];
  // One dependent AdaGrad chain per coordinate (7 instructions per user) bounds this role: the loop is branch-free, and
  // the deltas run UN users ahead of it in a register ring — slot j is refilled (index clamped, never guarded) as soon as
  // its value has been taken, so the role holds UN values, not 2 UN (the role's registers set the occupancy of the whole
  // input_rows_kernel launch).
  constexpr uint32_t UN = 16;
  float d[UN];
  // buffer loads: descriptor + 32-bit lane offset + scalar row offset, i.e. no 64-bit address registers per load in flight
  // (plain pointer arithmetic cost the role 69 VGPRs, and with it the whole launch a third of its occupancy)
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DELTA), 0, (int)(nb * hp.Kp * 4u), 0x00020000);
  const int voff = (int)(k * 4u);
  auto delta_of = [&](uint32_t user) -> float {                // user is wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (int)(min(user, nb - 1u) * hp.Kp * 4u), 0));
  };
#pragma unroll
  for (uint32_t j = 0; j < UN; ++j) d[j] = delta_of(j);
  uint32_t u = 0;
  {
    for (; u + UN <= nb; u += UN) {
#pragma unroll
      for (uint32_t j = 0; j < UN; ++j) {
        const float x = d[j];
        d[j] = delta_of(u + UN + j);
        ada_step(hp, p, acc, fmaf(hp.lambda, p, x));
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < UN; ++j)                            // the last nb % UN users
      if (u + j < nb) ada_step(hp, p, acc, fmaf(hp.lambda, p, d[j]));
  }
  b[k] = p;
  b_ag[k] = acc;
}

// ------------------------------------------------------------------------------------------------
// K5  input rows, row-major (cdae.hpp:333-349): for every kept input (u, j) in user order
//     grad = scale * delta_u + lambda W[j] + g_uj z_u   (the last term is the deferred decoder
//     gradient input_gradient[j], cdae.hpp:249-250, 342-343; absent when asymmetric)
// The row's example words are scanned 64 at a time; the kept inputs among them are taken in groups of UN.
// Three groups are kept in registers: while the (elementwise, reduction-free) AdaGrad chain of one runs, the
// delta / z / g loads of the next two are in flight (loads-only loop => counted vmcnt, see K3).
template <int NE, int UN>
struct InputGroup {
  float dl[UN][NE], zz[UN][NE], gg[UN];
  uint32_t cnt;                                                  // kept inputs in the group (wave-uniform); 0: the row is done
};

// One wavefront walks row `rank`; lane holds NI contiguous elements.  Most rows have one or two kept inputs per batch, so a
// wavefront is a short chain of L2 round trips: segment bounds -> example words -> delta / z / W rows -> step -> store; the
// next 64-example chunk of words is requested while the current one is worked on.  A group never straddles a chunk.
// Tried and measured slower (round 2, profiles/r02_main_stream.txt): three groups in a register ring (the round-1 form, 140
// VGPRs: same time), two alternating groups, and splitting the popular rows over a workgroup's four wavefronts (K quarters, 24
// examples in flight): the launch is bound by the sum of its serial latencies (dispatch 4 us, prologue 3 round trips, the
// popular rows' 60-odd dependent AdaGrad steps), not by any one of them.  What did help, once the `b` role's registers were out of
// the way (buffer loads): groups of 12 kept inputs instead of 4 — a popular row then pays 5 round trips instead of 16 (step
// 0.0943 -> 0.0921 ms at 256 users per batch, 0.1407 -> 0.1354 at 512; 8: 0.0922 / 0.1371).
template <int NE, int UN, bool ADAGRAD>
__device__ __forceinline__ void input_row_role(HyperParams hp, const uint32_t rank, const uint32_t lo,
                                               const uint32_t* __restrict__ item_order,
                                               const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end,
                                               const uint64_t* __restrict__ sorted_val, const float* __restrict__ Z,
                                               const float* __restrict__ DELTA, const float* __restrict__ G,
                                               float* __restrict__ W, float* __restrict__ W_ag, uint32_t* __restrict__ touched) {
  const uint32_t lane = threadIdx.x % WAVE;
  if (rank >= hp.num_items) return;
  hp.adagrad = ADAGRAD;
  const uint32_t item = item_order[rank];
  const uint32_t beg = seg_begin[rank], end = seg_end[rank];      // (rank-indexed tables)
  if (beg == end) return;

  // stream state: the kept-input mask / words / example ids of the current 64-example chunk; the chunk at `next_chunk` is
  // already in flight (`ahead`), so moving on to it costs no round trip
  uint32_t next_chunk = beg;
  uint64_t ahead = beg + lane < end ? sorted_val[beg + lane] : 0ull;
  unsigned long long mask = 0ull;
  uint32_t word = 0u, ex = 0u;
  auto fetch = [&](InputGroup<NE, UN>& grp) {
    while (mask == 0ull && next_chunk < end) {                   // wave-uniform: move on to the chunk in flight, request the one after
      word = (uint32_t)ahead;
      ex = (uint32_t)(ahead >> 32);
      mask = __ballot((word & INPUT_BIT) != 0u);
      next_chunk += WAVE;
      const uint32_t p = next_chunk + lane;
      ahead = p < end ? sorted_val[p] : 0ull;
    }
    grp.cnt = min((uint32_t)__popcll(mask), (uint32_t)UN);
#pragma unroll
    for (int t = 0; t < UN; ++t) {
      if ((uint32_t)t < grp.cnt) {                               // wave-uniform
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)word, src) & SLOT_MASK;
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)ex, src);
        const size_t o = (size_t)slot * hp.Kp + lo;
        vload<NE>(grp.dl[t], DELTA + o);
        vload<NE>(grp.zz[t], Z + o);
        grp.gg[t] = hp.asymmetric ? 0.f : G[e];
      }
    }
  };
  float w[NE], a[NE];
  auto apply = [&](const InputGroup<NE, UN>& grp) {
#pragma unroll
    for (int t = 0; t < UN; ++t) {
      if ((uint32_t)t < grp.cnt) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
          ada_step(hp, w[i], a[i], fmaf(hp.scale, grp.dl[t][i], fmaf(grp.gg[t], grp.zz[t][i], hp.lambda * w[i])));
      }
    }
  };

  InputGroup<NE, UN> A;
  fetch(A);
  if (A.cnt == 0u) return;                                       // no kept input on this row: W is not touched
  vload<NE>(w, W + (size_t)item * hp.Kp + lo);
  vload<NE>(a, W_ag + (size_t)item * hp.Kp + lo);
  do {
    apply(A);
    fetch(A);
  } while (A.cnt != 0u);
  vstore<NE>(W + (size_t)item * hp.Kp + lo, w);
  vstore<NE>(W_ag + (size_t)item * hp.Kp + lo, a);
  if (lane == 0 && touched) touched[item] = 1u;
}

#ifndef CDAE_INPUT_UN
#define CDAE_INPUT_UN 12
#endif
// grid: [bias_blocks: K4b] [the rows, one per wavefront, popular rows first]
template <int NI>
__global__ void __launch_bounds__(256)
input_rows_kernel(HyperParams hp, const uint32_t* __restrict__ item_order,
                  const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end,
                  const uint64_t* __restrict__ sorted_val, const float* __restrict__ Z,
                  const float* __restrict__ DELTA, const float* __restrict__ G,
                  float* __restrict__ W, float* __restrict__ W_ag, uint32_t* __restrict__ touched,
                  uint32_t nb, float* __restrict__ b, float* __restrict__ b_ag,
                  const float* __restrict__ DELTA_ROWS /* == DELTA unless linear_function (then Uu[u] (.) delta_u) */) {
  const uint32_t bias_blocks = (hp.Kp + blockDim.x - 1) / blockDim.x;     // leading workgroups: K4b
  if (blockIdx.x < bias_blocks) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (CDAE_SKIP_ROLE(hp, 1u)) return;
    const unsigned long long t0 = trace_begin(hp);
    if (hp.adagrad) hidden_bias_role<true>(hp, k, nb, DELTA, b, b_ag);
    else hidden_bias_role<false>(hp, k, nb, DELTA, b, b_ag);
    trace_end(hp, 7, k / 64u, t0);
    return;
  }
  if (CDAE_SKIP_ROLE(hp, 2u)) return;
  const uint32_t lane = threadIdx.x % WAVE;
  const uint32_t rank = __builtin_amdgcn_readfirstlane((blockIdx.x - bias_blocks) * (blockDim.x / WAVE) + threadIdx.x / WAVE);
  const unsigned long long t0 = trace_begin(hp);
  if (hp.adagrad) input_row_role<NI, CDAE_INPUT_UN, true>(hp, rank, lane * NI, item_order, seg_begin, seg_end, sorted_val, Z, DELTA_ROWS, G, W, W_ag, touched);
  else input_row_role<NI, CDAE_INPUT_UN, false>(hp, rank, lane * NI, item_order, seg_begin, seg_end, sorted_val, Z, DELTA_ROWS, G, W, W_ag, touched);
  trace_end(hp, 9, rank, t0);
}

// ------------------------------------------------------------------------------------------------
// K6  data_loss (cdae.hpp:78-101): per user  sum_{i in P(u)} loss(D[i].z_u + b'[i], 1) ; Z from K2.
// One wavefront per work unit (<= 128 positives of one user; a wavefront per user took as long as its most active
// user, 1468 dependent row visits at ML-10M shape), eight decoder rows in flight.
template <int NI>
__global__ void __launch_bounds__(256)
data_loss_kernel(HyperParams hp, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                 const uint32_t* __restrict__ uptr, uint32_t n_units, const uint32_t* __restrict__ unit_user,
                 uint64_t u0, uint32_t nb, const float* __restrict__ Z, const float* __restrict__ D,
                 const float* __restrict__ bp, double* __restrict__ out) {
  const uint32_t unit = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (unit >= n_units) return;
  const UnitRef ur = locate_unit(hp.unit_pos, uptr, nb, uptr[0] + unit, unit_user, u0);
  const uint64_t uid = u0 + ur.slot;
  const int64_t r0 = row_ptr[uid];
  const uint32_t n = (uint32_t)(row_ptr[uid + 1] - r0);
  const uint32_t p0 = ur.p0, p1 = min(ur.p1, n);
  const uint32_t lo = lane * NI;
  float z[NI];
  vload<NI>(z, Z + (size_t)ur.slot * hp.Kp + lo);
  double total = 0.;
  constexpr int UN = 8;
  for (uint32_t q0 = p0; q0 < p1; q0 += WAVE) {
    const uint32_t p = q0 + lane;
    const uint32_t item = p < p1 ? col[r0 + p] : 0u;
    const float bias = p < p1 ? bp[item] : 0.f;
    const uint32_t cnt = min((uint32_t)WAVE, p1 - q0);
    for (uint32_t j0 = 0; j0 < cnt; j0 += UN) {
      float d[UN][NI];
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)item, min(j0 + t, cnt - 1u));
        vload<NI>(d[t], D + (size_t)it * hp.Kp + lo);
      }
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        if (j0 + t < cnt) {                                      // wave-uniform
          float dot = 0.f;
#pragma unroll
          for (int i = 0; i < NI; ++i) dot = fmaf(d[t][i], z[i], dot);
          const float y = wave_sum(dot) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias), j0 + t));
          total += (double)loss_eval(hp.loss_type, y, 1.f);
        }
      }
    }
  }
  if (lane == 0) atomicAdd(out, total);
}

// sum of squares of a float array into a double (penalty.hpp:36-39)
__global__ void __launch_bounds__(256)
sqnorm_kernel(const float* __restrict__ x, size_t n, double* __restrict__ out) {
  double s = 0.;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
  if (threadIdx.x % WAVE == 0) atomicAdd(out, s);
}

// ------------------------------------------------------------------------------------------------
// K7  recommend (cdae.hpp:162-196), general path: scores of all items for one user per block, rated items masked,
// top-k by repeated block arg-max (k = 10, evaluation.hpp:145).  Ties resolve to the lower item id.
// One block (256 threads) per user: each wavefront scores items wave-strided.  The scores live in LDS when
// num_items * 4 bytes fit (score_ws == nullptr), else in a global workspace row of the user (1 M items = 4 MB, L2
// resident) — any num_dim <= 512, any topk, any item count.  Each thread caches the best of the items it owns
// (item % 256 == thread); after a winner is taken only its owner rescans, so a user costs one pass over the scores plus
// topk * num_items / 256 reads instead of topk passes.
// rated_override != nullptr (one user per launch): the caller's rated set replaces the train row as the mask
// (recommend(uid, topk, rated_item_set) with a set that is not the train row).
template <int NI>
__global__ void __launch_bounds__(256)
recommend_kernel(HyperParams hp, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                 uint64_t u0, const float* __restrict__ Z, const float* __restrict__ D,
                 const float* __restrict__ bp, uint32_t topk, uint32_t* __restrict__ out,
                 float* __restrict__ score_ws /* [gridDim.x][num_items] or nullptr */,
                 const uint32_t* __restrict__ rated_override, uint32_t n_override,
                 float* __restrict__ out_score /* [gridDim.x][topk] scores of the winners, or nullptr (item-sharded top-k merge) */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red_v = reinterpret_cast<float*>(smem_raw);                         // [4]
  uint32_t* red_i = reinterpret_cast<uint32_t*>(red_v + 4);                  // [4] + winner at [4]
  const uint32_t slot = blockIdx.x;
  float* score = score_ws ? score_ws + (size_t)slot * hp.num_items : reinterpret_cast<float*>(smem_raw + 64);
  const uint64_t uid = u0 + slot;
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
  const uint32_t lo = lane * NI;
  float z[NI];
  vload<NI>(z, Z + (size_t)slot * hp.Kp + lo);
  for (uint32_t item = wid; item < hp.num_items; item += nw) {
    float d[NI];
    vload<NI>(d, D + (size_t)item * hp.Kp + lo);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) dot = fmaf(d[i], z[i], dot);
    const float y = wave_sum(dot) + bp[item];
    if (lane == 0) score[item] = y;
  }
  __syncthreads();
  if (rated_override) {
    for (uint32_t p = threadIdx.x; p < n_override; p += blockDim.x) score[rated_override[p]] = -INFINITY;
  } else {
    const int64_t r0 = row_ptr[uid];
    const uint32_t n = (uint32_t)(row_ptr[uid + 1] - r0);
    for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) score[col[r0 + p]] = -INFINITY;   // cdae.hpp:177-179
  }
  __syncthreads();
  float best = -INFINITY;
  uint32_t best_i = 0xFFFFFFFFu;
  auto rescan = [&]() {
    best = -INFINITY; best_i = 0xFFFFFFFFu;
    for (uint32_t item = threadIdx.x; item < hp.num_items; item += blockDim.x) {
      const float v = score[item];
      if (v > best || (v == best && item < best_i)) { best = v; best_i = item; }
    }
  };
  rescan();
  for (uint32_t t = 0; t < topk; ++t) {
    float wv = best;
    uint32_t wi = best_i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(wv, off, WAVE);
      const uint32_t oi = __shfl_xor(wi, off, WAVE);
      if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    if (lane == 0) { red_v[wid] = wv; red_i[wid] = wi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float bv = red_v[0];
      uint32_t bi = red_i[0];
      for (uint32_t w2 = 1; w2 < nw; ++w2)
        if (red_v[w2] > bv || (red_v[w2] == bv && red_i[w2] < bi)) { bv = red_v[w2]; bi = red_i[w2]; }
      out[(size_t)slot * topk + t] = bi;
      if (out_score) out_score[(size_t)slot * topk + t] = bv;
      red_i[4] = bi;
    }
    __syncthreads();
    const uint32_t win = red_i[4];
    if (win != 0xFFFFFFFFu && win % blockDim.x == threadIdx.x) {             // the owner retires the winner and rescans
      score[win] = -INFINITY;
      rescan();
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// parameter init (cdae.hpp:109-134) from the CDAE_STREAM_INIT counter stream; pad elements get `pad`
__global__ void __launch_bounds__(256)
init_matrix_kernel(float* __restrict__ M, size_t rows, uint32_t K, uint32_t Kp, uint64_t key, double init_scale,
                   uint64_t row0 /* global index of local row 0: a data-parallel shard's Wu rows draw what the whole matrix would */) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * Kp) return;
  const size_t r = idx / Kp;
  const uint32_t k = (uint32_t)(idx % Kp);
  M[idx] = k < K ? (float)(cdae_init_uniform(key, (row0 + r) * K + k) * init_scale) : 0.f;
}
__global__ void __launch_bounds__(256)
fill_matrix_kernel(float* __restrict__ M, size_t rows, uint32_t K, uint32_t Kp, float value, float pad) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * Kp) return;
  M[idx] = (uint32_t)(idx % Kp) < K ? value : pad;
}

// Item-sharded layout: the per-user input sum and hidden gradient of a shard cover ITS item rows only and are
// all-reduced across shards, so the two reductions that encode_finish / hidden_finish fold in are also available alone.
// Hsum[slot] = sum of the slot's unit partials (fixed order), raw: no scale, bias or activation
template <int NI>
__global__ void __launch_bounds__(256)
unit_sum_kernel(HyperParams hp, const float* __restrict__ Hpart, const uint32_t* __restrict__ uptr, uint32_t nb, float* __restrict__ Hsum) {
  const uint32_t slot = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (slot >= nb) return;
  const uint32_t lo = lane * NI;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  const uint32_t ub = uptr[slot] - uptr[0], ue = uptr[slot + 1] - uptr[0];
  for (uint32_t u = ub; u < ue; ++u) {
    float part[NI];
    vload<NI>(part, Hpart + (size_t)u * hp.Kp + lo);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] += part[i];
  }
  vstore<NI>(Hsum + (size_t)slot * hp.Kp + lo, acc);
}
// Item shard, phase 0 in ONE launch (round 4; was unit_sum_kernel + one staging launch per private matrix): block 0 of the
// all-reduce buffer = the slot's input sum as above, blocks 1.. = the slot's rows of the private matrices the encode needs (Wu, then
// Uu) when this shard owns the user, zeros otherwise — the same values in the same places, one launch boundary instead of two or three.
template <int NI>
__global__ void __launch_bounds__(256)
unit_sum_stage_kernel(HyperParams hp, const float* __restrict__ Hpart, const uint32_t* __restrict__ uptr, uint32_t nb, uint64_t u0,
                      const float* __restrict__ table_a, const float* __restrict__ table_b, float* __restrict__ out /* [blocks][nb][Kp] */) {
  const uint32_t slot = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (slot >= nb) return;
  const uint32_t lo = lane * NI;
  // the rows of the private matrices first: their loads travel while the partial sums are added
  const uint64_t uid = u0 + slot;
  const bool own = cdae_xa::owns_user(uid, hp.own_u0, hp.own_u1);          // wave-uniform
  float ra[NI], rb[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) ra[i] = rb[i] = 0.f;
  if (own && table_a) vload<NI>(ra, table_a + (size_t)(uid - hp.own_u0) * hp.Kp + lo);
  if (own && table_b) vload<NI>(rb, table_b + (size_t)(uid - hp.own_u0) * hp.Kp + lo);
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  const uint32_t ub = uptr[slot] - uptr[0], ue = uptr[slot + 1] - uptr[0];
  for (uint32_t u = ub; u < ue; ++u) {
    float part[NI];
    vload<NI>(part, Hpart + (size_t)u * hp.Kp + lo);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] += part[i];
  }
  vstore<NI>(out + (size_t)slot * hp.Kp + lo, acc);
  uint32_t blk = 1;
  if (table_a) {
#pragma unroll
    for (int i = 0; i < NI; ++i) ra[i] = cdae_xa::own_row_contribution(own, ra[i]);
    vstore<NI>(out + ((size_t)(blk++) * nb + slot) * hp.Kp + lo, ra);
  }
  if (table_b) {
#pragma unroll
    for (int i = 0; i < NI; ++i) rb[i] = cdae_xa::own_row_contribution(own, rb[i]);
    vstore<NI>(out + ((size_t)blk * nb + slot) * hp.Kp + lo, rb);
  }
}
// HG[slot] = sum over the n_parts slabs of HGpart[x][rows][Kp] (fixed order)
template <int NI>
__global__ void __launch_bounds__(256)
slab_sum_kernel(HyperParams hp, const float* __restrict__ HGpart, uint32_t n_parts, uint32_t rows, uint32_t nb, float* __restrict__ HG) {
  const uint32_t slot = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (slot >= nb) return;
  const uint32_t lo = lane * NI;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  add_partial_rows<NI>(acc, HGpart, rows, n_parts, slot, slot + 1u, hp.Kp, lo);      // slab x, row `slot`: x ascending
  vstore<NI>(HG + (size_t)slot * hp.Kp + lo, acc);
}

// Item shard, sampled decode: the RAW local hidden gradient of every user of the batch — HG (decode's overflow corrections) plus
// hidden_gather_kernel's partial rows, added in exactly hidden_finish_kernel's order (unit-major, then partition), so that with one
// shard the all-reduced sum is the single-GPU hg bit for bit.  delta is formed after the all-reduce (hidden_finish, n_parts = 0).
template <int NI>
__global__ void __launch_bounds__(256)
hg_raw_kernel(HyperParams hp, const uint32_t* __restrict__ uptr, uint32_t n_units, uint32_t nb, const float* __restrict__ HGpart,
              uint32_t n_parts, float* __restrict__ HG, LateFinish lf = LateFinish{nullptr, nullptr, nullptr, nullptr, nullptr, 0u}) {
  __shared__ float sums[3 * 64 * NI];
  const uint32_t slot = blockIdx.x;                                 // (grid = nb: one workgroup per user, as hidden_finish_kernel)
  const uint32_t lo = (threadIdx.x % WAVE) * NI;
  float hg[NI];
  const uint32_t ub = uptr[slot] - uptr[0], ue = uptr[slot + 1] - uptr[0];
  if (!user_hg<NI>(hg, sums, HG, HGpart, n_units, n_parts, ub, ue, lf, slot, hp.Kp)) return;
  vstore<NI>(HG + (size_t)slot * hp.Kp + lo, hg);
}

// data-parallel exchange helpers (no reference counterpart; DESIGN.md "Multi-GPU")
__global__ void __launch_bounds__(256)
delta_kernel(const float* __restrict__ cur, const float* __restrict__ base, float* __restrict__ delta, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) delta[i] = cur[i] - base[i];
}
__global__ void __launch_bounds__(256)
touch_to_float_kernel(const uint32_t* __restrict__ touched, float* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = touched[i] ? 1.f : 0.f;
}
// cur = base + sum_delta * weight.  rule 0 (CDAE_DELTA_SUM): weight 1 — the all-reduced sum of every
// rank's accumulated steps, i.e. what summing gradients does.  rule 1 (CDAE_DELTA_TOUCH_MEAN): item rows are
// divided by the number of ranks that touched them, the hidden bias by world_size.
// Shared block layout: [I x Kp matrices ...][bp | bp_ag (I each)][b | b_ag (Kp each)].
__global__ void __launch_bounds__(256)
apply_delta_kernel(float* __restrict__ cur, const float* __restrict__ base, const float* __restrict__ sum,
                   const float* __restrict__ touch_sum, size_t n_matrix, uint32_t Kp, uint32_t num_items,
                   size_t n_total, uint32_t world_size, uint32_t rule) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  float wgt = 1.f;
  if (rule == 1u) {
    if (i < n_matrix) wgt = 1.f / fmaxf(1.f, touch_sum[(uint32_t)((i / Kp) % num_items)]);
    else if (i < n_total - 2u * Kp) wgt = 1.f / fmaxf(1.f, touch_sum[(uint32_t)((i - n_matrix) % num_items)]);
    else wgt = 1.f / (float)world_size;
  }
  cur[i] = fmaf(sum[i], wgt, base[i]);
}

// Pipelined exchange (cdae_hip_delta_stage / _merge / _merge_stage): peers' contributions arrive one exchange period late.
// Per element: cur = this replica's parameter, A (`base`) = the state all replicas AGREE on bit for bit, snap = cur as it was
// when the last delta was staged.
//   STAGE:        send = recv = cur - A ; snap = cur                  (recv is all-reduced in place while training continues)
//   MERGE:        A += recv ; cur = A + (cur - snap)                  (everybody's staged deltas, then this replica's progress since)
//   MERGE_STAGE:  MERGE of the previous period, then STAGE of this one, in a single pass over the block
// A moves only by the all-reduced sums, which are the same bits on every rank, so the replicas' A never drift apart; and
// whenever nothing was trained between a STAGE and its MERGE (the synchronous exchange, and the flush that ends an epoch)
// cur - snap is exactly 0 and cur == A on every replica, bit for bit.  (The first version folded the peers' part in as
// cur += recv - send: algebraically the same, but each replica rounded differently and the copies drifted by ulps.)
// send / recv are COMPACT: the matrices' pad columns (56 of 256 at K = 200) are neither staged nor all-reduced — row r of a
// matrix occupies Kc = round_up(K, 4) floats there — followed by the block's tail [b' | b'_ag | b | b_ag] as it is.
enum { DELTA_STAGE = cdae_xa::STAGE, DELTA_MERGE = cdae_xa::MERGE, DELTA_MERGE_STAGE = cdae_xa::MERGE_STAGE };

// (the algebra itself lives in cdae_exchange_algebra.h, shared with the CPU slice the two-rank gloo test drives)
template <int MODE>
__device__ __forceinline__ void delta_pipe_elem(float& c, float& A, float& snap, float& s, float& r) {
  cdae_xa::pipe_elem<MODE>(c, A, snap, s, r);
}

// SYNC (round 6): the synchronous exchange of a rank that owns its device — nothing is trained between a STAGE and its MERGE, so
// c - snap is exactly 0 and c == A after the merge: the STAGE pass writes neither snap nor the send copy (the all-reduce works in place
// on recv), the MERGE pass reads neither cur nor snap.  Same values (c = A + 0), 7 array passes over the block per step instead of 11.
template <int MODE, bool SYNC = false>
__global__ void __launch_bounds__(256)
delta_pipe_kernel(float* __restrict__ cur, float* __restrict__ base, float* __restrict__ snap, float* __restrict__ send,
                  float* __restrict__ recv, size_t n_matrix /* padded floats of all matrices */, uint32_t Kp, uint32_t Kc,
                  size_t n_tail) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q_per_row = Kc / 4;
  const size_t n_rows = n_matrix / Kp, n_mat4 = n_rows * q_per_row, n_tail4 = n_tail / 4;
  size_t pad_off, cmp_off;                                        // float offsets of this thread's element(s)
  int width;
  if (i < n_mat4) {
    const size_t row = i / q_per_row;
    const uint32_t q = (uint32_t)(i - row * q_per_row);
    pad_off = row * Kp + 4u * q; cmp_off = row * Kc + 4u * q; width = 4;
  } else if (i - n_mat4 < n_tail4) {
    const size_t j = i - n_mat4;
    pad_off = n_matrix + 4 * j; cmp_off = n_rows * Kc + 4 * j; width = 4;
  } else if (i - n_mat4 - n_tail4 < n_tail % 4) {
    const size_t j = 4 * n_tail4 + (i - n_mat4 - n_tail4);
    pad_off = n_matrix + j; cmp_off = n_rows * Kc + j; width = 1;
  } else {
    return;
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (width == 4) {
    float4 c = (SYNC && MODE == DELTA_MERGE) ? zero4 : *reinterpret_cast<float4*>(cur + pad_off), b = *reinterpret_cast<float4*>(base + pad_off);
    float4 sn = (MODE == DELTA_STAGE || SYNC) ? zero4 : *reinterpret_cast<float4*>(snap + pad_off);
    float4 s = zero4;
    float4 r = MODE == DELTA_STAGE ? zero4 : *reinterpret_cast<float4*>(recv + cmp_off);
    delta_pipe_elem<MODE>(c.x, b.x, sn.x, s.x, r.x); delta_pipe_elem<MODE>(c.y, b.y, sn.y, s.y, r.y);
    delta_pipe_elem<MODE>(c.z, b.z, sn.z, s.z, r.z); delta_pipe_elem<MODE>(c.w, b.w, sn.w, s.w, r.w);
    if (MODE != DELTA_STAGE) { *reinterpret_cast<float4*>(cur + pad_off) = c; *reinterpret_cast<float4*>(base + pad_off) = b; }
    if (MODE != DELTA_MERGE) {
      if (!SYNC) { *reinterpret_cast<float4*>(snap + pad_off) = sn; *reinterpret_cast<float4*>(send + cmp_off) = s; }
      *reinterpret_cast<float4*>(recv + cmp_off) = r;
    }
  } else {
    float c = (SYNC && MODE == DELTA_MERGE) ? 0.f : cur[pad_off], b = base[pad_off];
    float sn = (MODE == DELTA_STAGE || SYNC) ? 0.f : snap[pad_off], s = 0.f, r = MODE == DELTA_STAGE ? 0.f : recv[cmp_off];
    delta_pipe_elem<MODE>(c, b, sn, s, r);
    if (MODE != DELTA_STAGE) { cur[pad_off] = c; base[pad_off] = b; }
    if (MODE != DELTA_MERGE) { if (!SYNC) { snap[pad_off] = sn; send[cmp_off] = s; } recv[cmp_off] = r; }
  }
}

// The same three passes under the GLOBAL-ACCUMULATOR combine rule (cdae_xa::pipe_pair; cdae_hip_delta_set_combine): a thread takes a
// parameter element TOGETHER with its AdaGrad accumulator — matrix 2j with matrix 2j + 1 ([W | W_ag | (V | V_ag)]), b' with b'_ag,
// b with b_ag — same compact send / recv layout as delta_pipe_kernel, so the all-reduce between the passes does not change.
template <int MODE, bool SYNC = false>
__global__ void __launch_bounds__(256)
delta_pipe_pair_kernel(float* __restrict__ cur, float* __restrict__ base, float* __restrict__ snap, float* __restrict__ send,
                       float* __restrict__ recv, size_t n_matrix, uint32_t Kp, uint32_t Kc, uint32_t num_items, float beta) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q_per_row = Kc / 4;
  const size_t n_rows = n_matrix / Kp, n_pair_rows = n_rows / 2, n_mat4 = n_pair_rows * q_per_row;
  size_t pp, pa, cpo, cao;                                        // padded / compact float offsets of the parameter and its accumulator
  int width;
  if (i < n_mat4) {
    const size_t prow = i / q_per_row;                            // row of the pair space: (pair index, item)
    const uint32_t q = (uint32_t)(i - prow * q_per_row);
    const size_t pair = prow / num_items, r = prow - pair * num_items;
    const size_t row_p = 2 * pair * num_items + r, row_a = row_p + num_items;
    pp = row_p * Kp + 4u * q; pa = row_a * Kp + 4u * q; cpo = row_p * Kc + 4u * q; cao = row_a * Kc + 4u * q; width = 4;
  } else if (i - n_mat4 < (size_t)num_items + Kp) {
    const size_t j = i - n_mat4;
    const size_t rel_p = j < num_items ? j : 2 * (size_t)num_items + (j - num_items);
    const size_t rel_a = rel_p + (j < num_items ? num_items : Kp);
    pp = n_matrix + rel_p; pa = n_matrix + rel_a; cpo = n_rows * Kc + rel_p; cao = n_rows * Kc + rel_a; width = 1;
  } else {
    return;
  }
  if (width == 4) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr bool NO_CUR = SYNC && MODE == DELTA_MERGE;            // (c - snap is exactly 0: neither is read)
    float4 c = NO_CUR ? zero4 : *reinterpret_cast<float4*>(cur + pp), ca = NO_CUR ? zero4 : *reinterpret_cast<float4*>(cur + pa);
    float4 A = *reinterpret_cast<float4*>(base + pp), Aa = *reinterpret_cast<float4*>(base + pa);
    float4 sn = zero4, sna = zero4, s = zero4, sa = zero4, r = zero4, ra = zero4;
    if (MODE != DELTA_STAGE) {
      if (!SYNC) { sn = *reinterpret_cast<float4*>(snap + pp); sna = *reinterpret_cast<float4*>(snap + pa); }
      r = *reinterpret_cast<float4*>(recv + cpo); ra = *reinterpret_cast<float4*>(recv + cao);
    }
    cdae_xa::pipe_pair<MODE>(c.x, ca.x, A.x, Aa.x, sn.x, sna.x, s.x, sa.x, r.x, ra.x, beta);
    cdae_xa::pipe_pair<MODE>(c.y, ca.y, A.y, Aa.y, sn.y, sna.y, s.y, sa.y, r.y, ra.y, beta);
    cdae_xa::pipe_pair<MODE>(c.z, ca.z, A.z, Aa.z, sn.z, sna.z, s.z, sa.z, r.z, ra.z, beta);
    cdae_xa::pipe_pair<MODE>(c.w, ca.w, A.w, Aa.w, sn.w, sna.w, s.w, sa.w, r.w, ra.w, beta);
    if (MODE != DELTA_STAGE) {
      *reinterpret_cast<float4*>(cur + pp) = c; *reinterpret_cast<float4*>(cur + pa) = ca;
      *reinterpret_cast<float4*>(base + pp) = A; *reinterpret_cast<float4*>(base + pa) = Aa;
    }
    if (MODE != DELTA_MERGE) {
      if (!SYNC) {
        *reinterpret_cast<float4*>(snap + pp) = sn; *reinterpret_cast<float4*>(snap + pa) = sna;
        *reinterpret_cast<float4*>(send + cpo) = s; *reinterpret_cast<float4*>(send + cao) = sa;
      }
      *reinterpret_cast<float4*>(recv + cpo) = r; *reinterpret_cast<float4*>(recv + cao) = ra;
    }
  } else {
    constexpr bool NO_CUR = SYNC && MODE == DELTA_MERGE;
    float c = NO_CUR ? 0.f : cur[pp], ca = NO_CUR ? 0.f : cur[pa], A = base[pp], Aa = base[pa];
    float sn = 0.f, sna = 0.f, s = 0.f, sa = 0.f, r = 0.f, ra = 0.f;
    if (MODE != DELTA_STAGE) { if (!SYNC) { sn = snap[pp]; sna = snap[pa]; } r = recv[cpo]; ra = recv[cao]; }
    cdae_xa::pipe_pair<MODE>(c, ca, A, Aa, sn, sna, s, sa, r, ra, beta);
    if (MODE != DELTA_STAGE) { cur[pp] = c; cur[pa] = ca; base[pp] = A; base[pa] = Aa; }
    if (MODE != DELTA_MERGE) { if (!SYNC) { snap[pp] = sn; snap[pa] = sna; send[cpo] = s; send[cao] = sa; } recv[cpo] = r; recv[cao] = ra; }
  }
}


}  // namespace cdae
