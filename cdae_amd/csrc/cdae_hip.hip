// cdae_hip.hip — host side of libcdae_hip.so: the C ABI of include/cdae_hip.h on top of the kernels
// in cdae_kernels.hpp.  gfx950 only; no CPU fallback: every entry point needs a HIP device and fails
// loudly without one.
#include <hip/hip_runtime.h>

#include <cstring>   // rocprim's texture_cache_iterator needs ::memset declared first

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/cdae_hip.h"
#include "cdae_internal.hpp"
#include "cdae_kernels.hpp"
#include "cdae_full_kernels.hpp"
#include "cdae_recommend_kernels.hpp"
#include "cdae_sort_kernels.hpp"
#include "cdae_mf_kernels.hpp"

#ifdef CDAE_DECODE_TIMING
#define CDAE_TOUCHED_ARG ((uint32_t*)nullptr)     // the timing build borrows `touched` for its stamps
#else
#define CDAE_TOUCHED_ARG h->d_touched
#endif

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}

#define HIPCHK(expr)                                                                           \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

#define CHK(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

// launch `KERNEL<NI>` for the handle's NI (elements of a K-vector per lane)
#define DISPATCH_NI(ni, KERNEL, grid, block, shmem, stream, ...)                                  \
  do {                                                                                            \
    switch (ni) {                                                                                 \
      case 1: hipLaunchKernelGGL(KERNEL<1>, grid, block, shmem, stream, __VA_ARGS__); break;      \
      case 2: hipLaunchKernelGGL(KERNEL<2>, grid, block, shmem, stream, __VA_ARGS__); break;      \
      case 4: hipLaunchKernelGGL(KERNEL<4>, grid, block, shmem, stream, __VA_ARGS__); break;      \
      default: hipLaunchKernelGGL(KERNEL<8>, grid, block, shmem, stream, __VA_ARGS__); break;     \
    }                                                                                             \
  } while (0)

constexpr uint32_t MF_SEQ_USERS = 256;     // IMF / BPR sequential default: users per launch window (cdae_hip::mf_seq)
enum Family { F_SAMPLE = 0, F_SORT, F_ENCODE, F_DECODE, F_HIDDEN, F_INPUT, F_COUNT };

struct Span { int family; hipEvent_t a, b; };

}  // namespace

struct cdae_hip {
  cdae_hip_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t U = 0, I = 0;
  uint32_t K = 0, Kp = 0, NI = 1, B = 0;
  uint64_t uid_offset = 0;
  cdae::HyperParams hp{};

  std::vector<int64_t> h_row_ptr;
  int64_t* d_row_ptr = nullptr;
  uint32_t* d_col = nullptr;
  uint32_t* d_item_order = nullptr;
  uint32_t hot_rows = 0;            // decode: rows [0, hot_rows) of item_order get a wavefront of their own
  // late rows (cdae_kernels.hpp DecodeLate): the first late_rows of them leave their hidden-gradient terms to hidden_finish_kernel
  uint32_t late_rows = 0;
  float* d_Ghot = nullptr;          // [B][LATE_MAX]
  uint32_t* d_hotdup = nullptr;     // [B][LATE_MAX]
  uint32_t* d_late_bits = nullptr;  // [(I + 31) / 32] bitmap of the late rows' items
  uint32_t late_words = 0;
  bool fused_decode = false;        // decode + gather as ONE launch (decode_gather_kernel); CDAE_DECODE_UNFUSED turns it off (developer switch)
  bool fused_possible = false;      // ... what set_interactions found possible; fused_decode = fused_possible && allow_fused
  bool allow_fused = true;          // cdae_hip_set_decode_fused: off for handles whose launches may overlap another handle's on the same device
  // device error word: raised by a gather wavefront of the fused launch (bit 0) or a workgroup of bucket_sort_kernel (bit 1) that gave up
  // waiting; checked after every synchronisation of the main stream.  HOST memory mapped into the device (the kernels write it on the
  // error path only): the check is a plain read, not a 30 us device-to-host copy behind every synchronize
  uint32_t* h_err = nullptr;        // host address (a slot of the process-wide pool below: one mapped page per device, not one per handle)
  int err_slot = -1;
  uint32_t* d_fused_err = nullptr;  // the same word as the device sees it
  uint32_t* d_hot_cnt = nullptr;    // [hot workgroups] wavefronts of the popular rows finished so far (the fused launch's blockers wait on it)
  uint32_t fused_seq = 0;           // fused launches so far (wraps with the counters)
  cdae::FusedGeom fused_geo{};      // geometry of this handle's fused launch (set at the first one)
  bool fused_geo_set = false;
  std::vector<float> h_rank_len;    // [I] expected examples per batch of the row of popularity rank r (fused launch: balancing the four-row groups over the SIMDs)
  uint32_t* d_cold_map = nullptr;   // [decode workgroups x 4] four-row group of every wavefront of the fused launch's row workgroups (0xFFFFFFFF: none)
  uint32_t num_cus = 256;
  cdae::DecodeLate decode_late() const { return cdae::DecodeLate{d_Ghot, d_hotdup, late_rows}; }
  cdae::LateFinish late_finish() { return cdae::LateFinish{d_Ghot, d_hotdup, d_item_order, d_D0, d_dup_corr, late_rows}; }
  // developer switches, read once in cdae_hip_create (DESIGN.md lists them)
  bool one_row_per_wave = false;    // CDAE_DECODE_ONE_ROW_PER_WAVE: every row on the 64-lane decode path
  bool full_unfused = false;        // CDAE_FULL_UNFUSED: full-output decode as three separate GEMMs
  bool gemm_direct = false;         // CDAE_GEMM_DIRECT: the fragment-from-L1 GEMM kernel instead of the LDS-staged one (A/B switch)
  bool gemm_two_stage = false;      // CDAE_GEMM_TWO_STAGE: always the 128 x 128 two-stage LDS kernel (A/B switch)
  bool recommend_per_user = false;  // CDAE_RECOMMEND_PER_USER: recommend_kernel instead of the MFMA path
  std::vector<uint32_t> h_unit_ptr;     // prefix of work units (<= hp.unit_pos positives each) per user
  uint32_t* d_rank_of = nullptr;        // [I] inverse of d_item_order
  uint32_t* d_unit_user = nullptr;      // [total units] user of every unit (kernels' unit -> user look-up)
  uint32_t* d_unit_ptr = nullptr;
  uint32_t unit_cap = 0;                // most units in any window of batch_users users
  float* d_Hpart = nullptr;             // [unit_cap][Kp] encode partial sums
  uint32_t* d_uptr_tmp = nullptr;       // per-call prefix for cdae_hip_encode's arbitrary user lists

  // shared (item-side) parameters, one allocation: [W | W_ag | (V | V_ag) | bp | bp_ag | b | b_ag]
  float* d_shared = nullptr;
  size_t n_shared = 0, n_matrix = 0;
  size_t off[CDAE_P_COUNT] = {0};
  size_t cnt[CDAE_P_COUNT] = {0};   // padded element counts
  float* d_Wu = nullptr;
  float* d_Wu_ag = nullptr;
  float* d_Uu = nullptr;                // linear_function only: [U][Kp] per-user gate (cdae.hpp:437-438)
  float* d_Uu_ag = nullptr;
  float* d_Ssum = nullptr;              // linear_function only: [B][Kp] unscaled input sums of the batch (Uu step)
  float* d_delta_rows = nullptr;        // linear_function only: [B][Kp] Uu[u] (.) delta_u (what the input rows receive)

  // batch workspace
  uint64_t Ecap = 0;
  // example lists are kept in NSETS sets: batch q uses set q % NSETS.  Batch t+1 (and, with the second prep lane, t+2) is
  // sampled and sorted on a prep stream while batch t trains
  struct ExBuf {
    uint32_t* item = nullptr; uint64_t* val = nullptr;            // user-major example list
    uint32_t* sorted_item = nullptr; uint64_t* sorted_val = nullptr;   // the same, stably sorted by item
    uint32_t* seg = nullptr;                                      // [2*I]: first | one-past-last sorted position per item
    uint16_t* key16 = nullptr; uint16_t* sorted_key16 = nullptr;      // 16-bit sort keys when I <= 65536 (rocPRIM then runs onesweep)
    uint32_t* dup_of_pos = nullptr; uint32_t* dup_of_ex = nullptr;   // duplicate-negative correction rows (segment_kernel)
    uint32_t* dup_count = nullptr;
    // counting sort (cdae_sort_kernels.hpp): per-item counts / prefix / (unused cursor), item-bucketed values, per-tile counts
    uint32_t* item_count = nullptr; uint32_t* prefix = nullptr; uint32_t* rank = nullptr; uint64_t* bucketed = nullptr;
    uint32_t* tile_hist = nullptr; uint32_t* block_total = nullptr;      // (`rank` holds the in-block item prefix)
    uint32_t* wg_state = nullptr;                                    // bucket_sort_kernel: one word per item range (cleared by the sample kernel)
    uint32_t* cells = nullptr; uint32_t* cell_flag = nullptr;        // ... and its cells [ranges][units][BKC_SLOTS], filled by sample_kernel; overflow word
    uint32_t cell_tag = 0;                                           // (host) the tag of the batch last prepared into this set
    hipEvent_t ready = nullptr, released = nullptr;
  } ex[3];
  static constexpr int NSETS = 3;
  hipStream_t prep2 = nullptr;          // second prep lane (batches with odd sequence number), or nullptr: one lane, look-ahead 1
  bool prep2_own = false;               // prep2 is a stream of its own (else it aliases `aux`)
  bool prep2_auto = false;              // use the second lane only for batches of at most PREP2_AUTO_MAX_USERS users
  float* d_D0 = nullptr;                // decoder matrix at batch start (hidden-gradient gather)
  uint32_t dup_stripes = 1;             // counters the correction rows are numbered from (cdae_kernels.hpp DUP_STRIPES)
  float* d_dup_corr = nullptr; uint32_t dup_cap = 0;   // [dup_cap][Kp] hidden-gradient corrections of duplicate negatives
  float* d_HGpart = nullptr;            // [8][B][Kp] per-XCD partial hidden gradients
  // full-output decode (MFMA path): bf16 operand copies and the dense gradient, padded to 128-multiples
  uint32_t Bp = 0, Ip = 0;
  __bf16 *d_Zb = nullptr, *d_ZTb = nullptr, *d_Db = nullptr, *d_DTb = nullptr, *d_Gb = nullptr, *d_GTb = nullptr;
  float* d_dD = nullptr;
  uint8_t* d_has_in = nullptr;          // [I]: item has a kept input in the block being stepped (set by full_positive_fixup_kernel, cleared by full_rows_inputs_kernel)
  uint32_t* d_iota = nullptr;           // 0..B: identity unit prefix (fused full-output path: one hg partial row per user)
  uint32_t* d_bits_train = nullptr;     // [NSETS][B x ceil(I/32)] rated-item bitmap of the batch (targets of the fused full-output decode), per example-buffer set
  size_t bits_stride = 0;
  uint32_t full_slices = 1;
  // Full-output path, item spaces below 32768 (full_rows_kernel): the row step writes the bf16 images of every decoder row it
  // has just stepped and the encode writes those of z, so a steady-state batch needs no conversion launch.  db_valid: d_Db / d_DTb
  // hold the CURRENT decoder (cleared by everything else that writes parameters); zb_rows: rows of d_Zb / columns of d_ZTb that may
  // be non-zero (the encode only writes the batch's users: a shorter batch takes the zero-filling conversion kernel once)
  bool db_valid = false;
  bool db_rows_valid = false;           // item spaces >= 32768: d_Db (only) holds the current decoder — the batch starts with a bf16 -> bf16 transposition
  uint32_t zb_rows = 0xFFFFFFFFu;
  // full-output, small item spaces: blocks of at most this many users run on ONE stream, the b recurrence as leading workgroups of the
  // GEMM 3 launch (first half of the users) and of the row launch (the rest) (CDAE_FULL_ONE_STREAM_MAX; 0 = always the two-stream
  // order).  Measured (Yelp shape K=50 / ML-10M shape K=200, ms per block, two streams -> one): 64 users 0.074 -> 0.051, 256: 0.076 ->
  // 0.055, 512: 0.083 -> 0.067 / 0.124 -> 0.106, 1024: 0.099 -> 0.088 / 0.145 -> 0.132, 2048: - / 0.197 -> 0.188, 4096: - / 0.327 -> 0.327
  // — a hand-off between two streams costs ~15 us, the recurrence 41 ns per user
  uint32_t full_one_stream_max = 2048;
  bool full_bias_unsplit = false;       // CDAE_FULL_BIAS_UNSPLIT: the whole recurrence beside the row launch (A/B)
  bool fused_images = true;             // CDAE_FULL_SEPARATE_COPIES turns it off (developer switch: conversion launches as in round 2)
  hipStream_t aux = nullptr;            // full-output path: the hidden-bias recurrence beside GEMM 3
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_delta = nullptr;
  bool join_pending = false;            // full-output path: the aux stream's b recurrence of the last batch has not been joined yet
  hipStream_t prep = nullptr;           // sampling + sorting of the next batch
  void* d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0, sort_tmp_stride = 0;   // two workspaces, one per prep lane
  float* d_Z = nullptr; float* d_Dz = nullptr; float* d_HG = nullptr; float* d_G = nullptr;
  uint32_t* d_touched = nullptr;
  double* d_scalar = nullptr;
  uint32_t* d_uids = nullptr;
  uint32_t* d_rec = nullptr; size_t rec_cap = 0;
  float* d_score = nullptr; size_t score_cap = 0;      // recommend, general path: score rows when num_items * 4 B exceed the LDS
  float* d_zeval = nullptr; float* d_hpart_eval = nullptr; uint32_t eval_cap = 0, eval_unit_cap = 0;   // evaluation workspace
  uint32_t* d_bits = nullptr; size_t bits_cap = 0;                                                     // recommend: rated-item bitmap
  // TOPN metrics on the device (cdae_hip_set_test_rows / cdae_hip_eval_topn): the validation rows as CSR, per-user metric terms
  int64_t* d_test_ptr = nullptr; uint32_t* d_test_col = nullptr; double* d_topn_pu = nullptr; double* d_topn_out = nullptr;
  uint64_t test_users_with_rows = 0;
  bool topn_active = false;             // recommend paths: score every chunk's lists with topn_user_kernel (cdae_hip_eval_topn)
  int sort_bits = 1;
  // prep worker: the ~12 launches that sample + sort a batch are issued by a second host thread (the training loop was bound by
  // the HOST's launch rate: ~21 runtime calls x 4.5 us per batch on one thread)
  struct PrepJob { int set; uint64_t s0; uint32_t nb, cidx; uint64_t E; uint64_t seed; uint32_t epoch; int lane; uint64_t prof_q; };
  bool prep_threaded = true;            // CDAE_PREP_THREAD=0: issue everything from the caller's thread
  std::thread worker;
  std::mutex job_mu;
  std::condition_variable job_cv;
  std::condition_variable issued_cv;    // worker -> caller: jobs_issued advanced (await_prep_upto sleeps here after a short spin)
  std::deque<PrepJob> jobs;
  bool worker_stop = false;
  uint64_t jobs_submitted = 0;          // (caller's thread only)
  std::atomic<uint64_t> jobs_issued{0}; // jobs whose launches + `ready` record have been issued (worker -> caller)
  std::atomic<int> worker_failed{0};
  std::string worker_error;             // valid once worker_failed != 0
  std::mutex prof_mu;                   // spans / event pool are touched by both threads when profiling
  uint32_t gather_halves = 1;           // wavefronts per (unit, item partition) in hidden_gather_kernel (CDAE_GATHER_HALVES = 1 | 2; 2 measured slower)
  uint32_t encode_users_max = 768;      // batches above this many users take the two-launch encode (a workgroup of 16 wavefronts per user is
                                        // mostly idle wavefronts; full-output: 512 users -7 % per step with one launch, 1024 +2 %, 2048 +11 %; sampled: equal at 1024, +4 % at 2048); CDAE_ENCODE_USERS_MAX
  bool full_separate_copies = false;    // CDAE_FULL_SEPARATE_COPIES: D and Z bf16 copies as two launches (developer switch)
  bool encode_two_launches = false;     // CDAE_ENCODE_TWO_LAUNCHES: the training encode as encode_partial + encode_finish (developer switch)
  bool debug_skip_prep = false;         // CDAE_DEBUG_SKIP_PREP (timing experiment only: batches reuse stale example lists -> WRONG results)
  // bucket_sort_kernel (cdae_sort_kernels.hpp): the default item-major ordering — one narrow launch; item ranges cut at set_interactions
  bool bucket_sort = false;
  uint32_t bucket_ranges = 0;
  uint32_t* d_bucket_cut = nullptr;     // [bucket_ranges + 1] item ids
  bool bucket_attr_set = false;         // dynamic-LDS attribute set on this handle's device
  uint16_t* d_range_of = nullptr;       // [I] item -> range (sample_kernel routes every example into its range's cell); nullptr: the sort scans the key list
  uint32_t cell_units = 0;              // units the cells are allocated for (the most of any batch)
  bool counting_sort = false;           // tile counting sort on the prep stream (cdae_sort_kernels.hpp) instead of rocPRIM: num_items <= TILE_SORT_MAX_ITEMS
  bool tile_attr_set = false;           // dynamic LDS above 64 KiB allowed for the two tile kernels (per handle: the attribute is per device)
  bool gemm3_attr_set[8] = {false, false, false, false, false, false, false, false};   // launch_gemm_lds: dynamic-LDS attribute set on this handle's device, per epilogue
  bool gemmw_attr_set[8] = {false, false, false, false, false, false, false, false};   // ... of the 256 x 256-tile kernel
  bool gemm_narrow = false;             // CDAE_GEMM_NARROW: never the 256 x 256-tile kernel (A/B switch)
  bool gemm1_tiled = false;             // CDAE_GEMM1_TILED: GEMM 1 as the 256 x 256-tile kernel where gemm1_loss_zreg_kernel would run (A/B switch)
  bool gemm1_zreg_attr_set[2] = {false, false};   // dynamic-LDS attribute of gemm1_loss_zreg_kernel<LOSS> set on this handle's device
  bool gemm1_zreg = false;              // CDAE_GEMM1_ZREG: round 3's gemm1_loss_zreg_kernel (eight wavefronts in lockstep) where gemm1_loss_duo_kernel runs (A/B switch)
  bool gemm1_duo_attr_set[2] = {false, false};
  bool fused_attr_set = false;          // dynamic-LDS attribute of this handle's full_decode_fused_kernel instance set (one K and loss per handle)
  bool gemm2_nt = false;                // CDAE_GEMM2_NT: hg = G D from G and D^T (gemm_nt_bf16_ldsw_kernel) where gemm_tn_bf16_kernel would read G^T and D (A/B switch)
  int gemm2_stages = 2;                 // CDAE_GEMM2_STAGES: LDS stages of gemm_tn_bf16_kernel (2 = 64-row stages; 3 / 4 = 32-row stages, measured no faster)
  bool gemm_tn_attr_set = false;        // dynamic-LDS attribute of gemm_tn_bf16_kernel set on this handle's device
  bool rows_separate = false;           // CDAE_FULL_ROWS_SEPARATE: GEMM 3 and the row step as two launches where gemm3_rows_fused_kernel would run (A/B switch)
  bool fused_rows_attr_set[2][2] = {};      // dynamic-LDS attribute of gemm3_rows_fused_kernel<ADA, KH> set on this handle's device
  int rows_fused_kh = 2;                // CDAE_FULL_ROWS_KH: workgroups per item tile of the fused row step (1 | 2)

  // data-parallel exchange
  float* d_base = nullptr; float* d_delta = nullptr; float* d_recv = nullptr; float* d_snap = nullptr;   // agreed state, staged delta, all-reduced delta, parameters at the last stage
  void* xchg = nullptr; void (*xchg_free)(void*) = nullptr;   // communicator + schedule of the exchange (cdae_multi.hip)
  uint32_t delta_combine = CDAE_COMBINE_SUM;                  // cdae_hip_delta_set_combine: how staged deltas are folded in (cdae_exchange_algebra.h)
  // IMF / BPR handles (cdae_hip_create_mf, cdae_mf_kernels.hpp): 0 = CDAE, 1 = IMF, 2 = BPR
  uint32_t mf = 0, mf_bias = 1;
  // IMF / BPR, batch_users = 1 (the library default: the reference's strictly sequential loop): the users are still taken one after the
  // other and every instance still steps the user vector and the item row(s) in place — but LAUNCHES cover MF_SEQ_USERS users: one
  // sampling launch for all of them (the draws are keyed by user, not by batch) and one mf_user_kernel<IN_PLACE> launch in which a
  // single wavefront walks them in order.  Through round 3 every user was a batch of its own: a sampling launch, a library sort and
  // a segment pass nobody read, ~20 runtime calls.  h->B is then the launch window (capacities, plans); cdae_hip_batch_users still
  // answers 1 and the stats count one block per user.
  bool mf_seq = false;
  bool mf_auto = false;                 // IMF / BPR handle created with batch_users = 0: the block size is chosen per data set (set_interactions)
  bool prep_force_sort = false;         // cdae_hip_debug_sample_batch: the item-major lists are wanted although training would not read them
  // IMF / BPR block schedule (batch_users > 1): the users are trained in ACTIVITY-GROUPED order — sorted by train-row length, cut
  // into blocks of batch_users, the blocks visited in a fixed pseudo-random order — because a block lasts as long as its most active
  // user's serial chain (a heavy-tailed mix made every block as slow as its heaviest user: 3 ms against 0.24 ms for the average
  // one).  user_perm[position] = user id, user_inv its inverse; everything on the device is indexed by POSITION, the C ABI by user
  // id (get / set_param, recommend_all map through them).  Empty: identity (CDAE; IMF / BPR with one user per block = the reference order).
  std::vector<uint32_t> user_perm, user_inv;
  uint32_t ex_per_pos = 1;              // examples of the batch's item-sorted list per train interaction (CDAE: 1 + num_neg)
  float* d_ub = nullptr; float* d_ub_ag = nullptr;   // [U] user bias and its accumulator
  float* d_UVpre = nullptr;             // [instances of a batch][Kp]: user vector before each instance's step (phase I input)
  // item-sharded layout (cdae_multi.hip): this handle holds item rows [item0, item0 + I) of I_global; every user,
  // possibly with no local item.  The two per-user sums that cross shards live in d_Hsum (input sums) and d_HG (hidden gradient).
  bool item_shard = false; uint64_t item0 = 0, I_global = 0;
  uint32_t* d_gpos = nullptr;           // item shard: per user (length of the whole row, position of the first local item): the dropout stream's index space
  float* d_Hsum = nullptr; float* d_hsum_eval = nullptr; uint32_t* d_iota_eval = nullptr; float* d_rec_score = nullptr; size_t rec_score_cap = 0;
  uint32_t hsum_eval_cap = 0;
  uint64_t fs_prepped = 0;              // item-sharded training: batches whose example lists have been prepared (buffer set = parity)
  // item shard: users [own_u0, own_u1) keep their private rows (Wu, Wu_ag, Uu, Uu_ag) HERE, table row 0 = user own_u0 (SURVEY.md
  // §8(e): the user node is sharded by user; the rows of a batch's users reach the other shards through the input-sum all-reduce)
  uint64_t own_u0 = 0, own_u1 = ~0ull;
  uint64_t wu_rows() const { return item_shard ? std::min<uint64_t>(own_u1, U) - std::min<uint64_t>(own_u0, U) : U; }
  // item shard, SAMPLED decode: the WHOLE train rows with global item ids — negatives are drawn from all I_global items and
  // rejected against the whole row, the example list keeps the single-GPU layout (work units over the whole rows)
  const int64_t* g_row_ptr_src = nullptr; const uint32_t* g_col_src = nullptr;   // (set_item_shard_global -> the next set_interactions)
  std::vector<int64_t> h_grow_ptr;
  std::vector<uint32_t> h_gunit_ptr;
  int64_t* d_grow_ptr = nullptr; uint32_t* d_gcol = nullptr; uint32_t* d_gunit_ptr = nullptr; uint32_t* d_gunit_user = nullptr;
  bool shard_sampled() const { return item_shard && !cfg.full_output; }

  bool skip_ready_wait = false;         // compute_batch: enqueue_users has seen the set's `ready` event complete on the host (no wait packet on the main stream)
  uint32_t host_pace_us = 200;          // enqueue_users: how long the caller's thread looks for a batch's lists to be complete before it leaves the wait to the device (0: always the device; CDAE_HOST_PACE_US, developer switch)
  uint64_t seq = 0;                     // batches enqueued so far; batch q uses example-buffer set q % NSETS
  // sets (seq + t) % NSETS, t < pre_n, already hold (or have queued) the prepared batches pre[t] (cdae_hip_prefetch_users)
  struct PreBatch { uint64_t s0, seed; uint32_t nb, cidx, epoch; };
  PreBatch pre[2] = {};
  uint32_t pre_n = 0;
  uint64_t acc_users = 0, acc_examples = 0, acc_batches = 0;   // since the last stats collection
  int profiling = 0;                    // 0 off; k >= 1: HIP events around the kernel families of every k-th batch
  uint32_t prof_mask = 0xFFFFFFFFu;     // ... of the families whose bit is set (cdae_hip_set_profiling_families)
  uint64_t prof_q = 0;                  // sequence number of the batch being enqueued (sampling of the profile)
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;

  float* P(uint32_t which) {
    if (which == CDAE_P_WU) return d_Wu;
    if (which == CDAE_P_WU_AG) return d_Wu_ag;
    if (which == CDAE_P_UU) return d_Uu;
    if (which == CDAE_P_UU_AG) return d_Uu_ag;
    if (which == CDAE_P_UB) return d_ub;
    if (which == CDAE_P_UB_AG) return d_ub_ag;
    return cnt[which] ? d_shared + off[which] : nullptr;
  }
  float* dec() { return cfg.asymmetric ? P(CDAE_P_V) : P(CDAE_P_W); }
  float* dec_ag() { return cfg.asymmetric ? P(CDAE_P_V_AG) : P(CDAE_P_W_AG); }
  float* delta_rows() { return cfg.linear_function ? d_delta_rows : d_HG; }
  bool full_unfused_nt() const { return gemm_direct || gemm_two_stage || gemm_narrow; }   // developer switches that select one of the older NT kernels for all three products
};

namespace {

// Full-output path: the hidden-bias recurrence of a batch runs on the aux stream and is joined as LATE as possible — right
// before the next batch's encode_finish (the first consumer of b and of the delta buffer), so that the next batch's input
// gather and the bf16 copies of D overlap its tail.  Every other entry point that touches parameters or the main stream
// joins first.
int join_aux(cdae_hip* h) {
  if (h->join_pending) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    h->join_pending = false;
  }
  return 0;
}

// Device error words: ONE page of mapped host memory per device and process, a slot per handle (a pinned, mapped allocation per handle
// is a system-wide resource; hundreds of handles in one process — a test session — should not each hold one).
struct ErrPool {
  static constexpr int SLOTS = 1024;
  uint32_t* host = nullptr; uint32_t* dev = nullptr;
  std::vector<char> used;
};
std::mutex g_err_mu;
ErrPool g_err_pool[64];
int err_slot_acquire(int device, uint32_t** host, uint32_t** dev, int* slot) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  if (device < 0 || device >= 64) return fail("device id %d outside the error-word pool", device);
  ErrPool& p = g_err_pool[device];
  if (!p.host) {
    HIPCHK(hipHostMalloc((void**)&p.host, ErrPool::SLOTS * sizeof(uint32_t), hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&p.dev, p.host, 0));
    p.used.assign(ErrPool::SLOTS, 0);
  }
  for (int i = 0; i < ErrPool::SLOTS; ++i)
    if (!p.used[i]) { p.used[i] = 1; p.host[i] = 0u; *host = p.host + i; *dev = p.dev + i; *slot = i; return 0; }
  return fail("more than %d live handles on device %d", ErrPool::SLOTS, device);
}
void err_slot_release(int device, int slot) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  if (device >= 0 && device < 64 && slot >= 0 && g_err_pool[device].host) g_err_pool[device].used[slot] = 0;
}

// After the main stream has been synchronised: the handle's device error word.  Bit 0: a gather wavefront of a fused launch
// (decode_gather_kernel) gave up waiting for a g; bit 1: a workgroup of bucket_sort_kernel gave up waiting for the ranges in front of it.
// Both waits rest on workgroups being dispatched in index order and are bounded, so that a broken assumption is an error, not a hang.
int fused_check(cdae_hip* h) {
  if (!h->d_fused_err || !(h->fused_decode || h->bucket_sort)) return 0;
  const uint32_t err = *(volatile uint32_t*)h->h_err;
  if (err) {
    *(volatile uint32_t*)h->h_err = 0u;
    return fail("%s%s: workgroups not dispatched in index order, or a launch that was skipped?  The parameters of this handle are no longer valid",
                (err & 1u) ? "fused decode + gather launch: a gather wavefront gave up waiting for its g; " : "",
                (err & 2u) ? "bucket_sort_kernel: a workgroup gave up waiting for the ranges in front of it" : "");
  }
  return 0;
}

// Events between the library's own streams order work on ONE device: they need no system-scope fence.  A default HIP event
// writes back and invalidates the caches when it is recorded — measured here as ~14 us of idle main stream per batch around
// the `released` record and ~3 us at the `ready` wait (profiles/r02_wave_timeline_256.txt: 86 us of kernels in a 100 us step).
// CDAE_EVENT_SYSTEM_FENCE=1 restores the default.  (The exchange's events in cdae_multi.hip stay system-scope: RCCL peers read.)
inline unsigned sync_event_flags() {      // (the environment is read at every call: per handle, not per process)
  const bool sys = DEV_ENV("CDAE_EVENT_SYSTEM_FENCE") != nullptr;
  return sys ? (unsigned)hipEventDisableTiming : (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
}
inline unsigned timing_event_flags() {
  const bool sys = DEV_ENV("CDAE_EVENT_SYSTEM_FENCE") != nullptr;
  return sys ? (unsigned)hipEventDefault : (unsigned)hipEventDisableSystemFence;
}

int get_event(cdae_hip* h, hipEvent_t* ev) {
  {
    std::lock_guard<std::mutex> lk(h->prof_mu);
    if (!h->pool.empty()) { *ev = h->pool.back(); h->pool.pop_back(); return 0; }
  }
  HIPCHK(hipEventCreateWithFlags(ev, timing_event_flags()));
  return 0;
}
struct Prof {   // RAII-less helper: begin()/end() around one kernel family launch, on the stream it is launched on
  cdae_hip* h; Span s; bool on; hipStream_t st;
  int begin(cdae_hip* hh, int family, hipStream_t stream, uint64_t q = ~0ull /* batch sequence number; default: the caller thread's prof_q */) {
    if (q == ~0ull) q = hh->prof_q;
    h = hh; on = hh->profiling > 0 && ((hh->prof_mask >> family) & 1u) && q % (uint64_t)hh->profiling == 0; st = stream; if (!on) return 0;
    s.family = family;
    CHK(get_event(h, &s.a)); CHK(get_event(h, &s.b));
    HIPCHK(hipEventRecord(s.a, st));
    return 0;
  }
  int end() {
    if (!on) return 0;
    HIPCHK(hipEventRecord(s.b, st));
    std::lock_guard<std::mutex> lk(h->prof_mu);
    h->spans.push_back(s);
    return 0;
  }
};

int collect_profile(cdae_hip* h, cdae_hip_stats* st) {
  double ms[F_COUNT] = {0};
  uint64_t launches[F_COUNT] = {0};
  std::lock_guard<std::mutex> lk(h->prof_mu);
  for (Span& s : h->spans) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, s.a, s.b));
    ms[s.family] += t;
    launches[s.family]++;
    h->pool.push_back(s.a); h->pool.push_back(s.b);
  }
  h->spans.clear();
  if (st) {
    st->ms_sample = ms[F_SAMPLE]; st->ms_sort = ms[F_SORT]; st->ms_encode = ms[F_ENCODE];
    st->ms_decode = ms[F_DECODE]; st->ms_hidden = ms[F_HIDDEN]; st->ms_input = ms[F_INPUT];
    st->launches_decode = launches[F_DECODE];
  }
  return 0;
}

void free_all(cdae_hip* h) {
  void* ptrs[] = {h->d_row_ptr, h->d_col, h->d_item_order, h->d_shared, h->d_Wu, h->d_Wu_ag, h->d_D0, h->d_HGpart,
                  h->d_unit_ptr, h->d_Hpart, h->d_uptr_tmp, h->d_Zb, h->d_ZTb, h->d_Db, h->d_DTb, h->d_Gb, h->d_GTb, h->d_dD, h->d_has_in,
                  h->d_sort_tmp, h->d_Z, h->d_Dz, h->d_HG, h->d_G, h->d_touched, h->d_scalar, h->d_uids, h->d_rec,
                  h->d_base, h->d_delta, h->d_recv, h->d_snap, h->d_dup_corr, h->d_unit_user, h->d_zeval, h->d_bits, h->d_hpart_eval, h->d_iota, h->d_bits_train,
                  h->d_Uu, h->d_Uu_ag, h->d_Ssum, h->d_delta_rows, h->d_score, h->d_Hsum, h->d_hsum_eval, h->d_iota_eval, h->d_rec_score, h->d_gpos, h->d_ub, h->d_ub_ag, h->d_UVpre, h->d_rank_of,
                  h->d_grow_ptr, h->d_gcol, h->d_gunit_ptr, h->d_gunit_user, h->d_test_ptr, h->d_test_col, h->d_topn_pu, h->d_topn_out, h->d_bucket_cut, h->d_range_of,
                  h->d_Ghot, h->d_hotdup, h->d_late_bits, h->d_hot_cnt, h->d_cold_map};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& b : h->ex) {
    void* q[] = {b.item, b.val, b.sorted_item, b.sorted_val, b.seg, b.dup_of_pos, b.dup_of_ex, b.dup_count, b.key16, b.sorted_key16,
                 b.item_count, b.prefix, b.rank, b.bucketed, b.tile_hist, b.block_total, b.wg_state, b.cells, b.cell_flag};
    for (void* p : q) if (p) (void)hipFree(p);
    if (b.ready) (void)hipEventDestroy(b.ready);
    if (b.released) (void)hipEventDestroy(b.released);
  }
  if (h->h_err) { err_slot_release(h->device, h->err_slot); h->h_err = nullptr; h->d_fused_err = nullptr; h->err_slot = -1; }
  if (h->prep) (void)hipStreamDestroy(h->prep);
  if (h->prep2 && h->prep2_own) (void)hipStreamDestroy(h->prep2);
  if (h->aux) (void)hipStreamDestroy(h->aux);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_delta) (void)hipEventDestroy(h->ev_delta);
  for (hipEvent_t e : h->pool) (void)hipEventDestroy(e);
  for (Span& s : h->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  if (h->stream) (void)hipStreamDestroy(h->stream);
}

int await_prep(cdae_hip* h);     // (prep worker, below)
void stop_prep_worker(cdae_hip* h);
int quiesce(cdae_hip* h) {
  CHK(await_prep(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->prep) HIPCHK(hipStreamSynchronize(h->prep));
  if (h->prep2) HIPCHK(hipStreamSynchronize(h->prep2));
  if (h->aux) HIPCHK(hipStreamSynchronize(h->aux));
  return 0;
}

int free_interaction_state(cdae_hip* h) {
  void** ptrs[] = {(void**)&h->d_row_ptr, (void**)&h->d_col, (void**)&h->d_item_order, (void**)&h->d_shared,
                   (void**)&h->d_Wu, (void**)&h->d_Wu_ag, (void**)&h->d_D0, (void**)&h->d_HGpart, (void**)&h->d_sort_tmp,
                   (void**)&h->d_unit_ptr, (void**)&h->d_Hpart, (void**)&h->d_uptr_tmp, (void**)&h->d_Zb, (void**)&h->d_ZTb,
                   (void**)&h->d_Db, (void**)&h->d_DTb, (void**)&h->d_Gb, (void**)&h->d_GTb, (void**)&h->d_dD, (void**)&h->d_has_in,
                   (void**)&h->d_Z, (void**)&h->d_Dz, (void**)&h->d_HG, (void**)&h->d_G, (void**)&h->d_touched,
                   (void**)&h->d_uids, (void**)&h->d_rec, (void**)&h->d_base, (void**)&h->d_delta, (void**)&h->d_recv, (void**)&h->d_snap, (void**)&h->d_dup_corr, (void**)&h->d_unit_user, (void**)&h->d_zeval, (void**)&h->d_bits, (void**)&h->d_hpart_eval, (void**)&h->d_iota, (void**)&h->d_bits_train,
                   (void**)&h->d_Uu, (void**)&h->d_Uu_ag, (void**)&h->d_Ssum, (void**)&h->d_delta_rows, (void**)&h->d_score,
                   (void**)&h->d_Hsum, (void**)&h->d_hsum_eval, (void**)&h->d_iota_eval, (void**)&h->d_rec_score, (void**)&h->d_gpos, (void**)&h->d_ub, (void**)&h->d_ub_ag, (void**)&h->d_UVpre, (void**)&h->d_rank_of,
                   (void**)&h->d_grow_ptr, (void**)&h->d_gcol, (void**)&h->d_gunit_ptr, (void**)&h->d_gunit_user,
                   (void**)&h->d_test_ptr, (void**)&h->d_test_col, (void**)&h->d_topn_pu, (void**)&h->d_topn_out, (void**)&h->d_bucket_cut, (void**)&h->d_range_of,
                   (void**)&h->d_Ghot, (void**)&h->d_hotdup, (void**)&h->d_late_bits, (void**)&h->d_hot_cnt, (void**)&h->d_cold_map};
  for (auto& b : h->ex) {
    void** q[] = {(void**)&b.item, (void**)&b.val, (void**)&b.sorted_item, (void**)&b.sorted_val, (void**)&b.seg,
                  (void**)&b.dup_of_pos, (void**)&b.dup_of_ex, (void**)&b.dup_count, (void**)&b.key16, (void**)&b.sorted_key16,
                  (void**)&b.item_count, (void**)&b.prefix, (void**)&b.rank, (void**)&b.bucketed, (void**)&b.tile_hist, (void**)&b.block_total,
                  (void**)&b.wg_state, (void**)&b.cells, (void**)&b.cell_flag};
    for (void** p : q) if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
  }
  for (void** p : ptrs) if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
  h->rec_cap = 0; h->score_cap = 0; h->rec_score_cap = 0; h->hsum_eval_cap = 0;
  h->db_valid = false; h->db_rows_valid = false; h->zb_rows = 0xFFFFFFFFu;
  h->eval_cap = 0; h->eval_unit_cap = 0; h->bits_cap = 0;
  return 0;
}

template <class T> int dev_alloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
  return 0;
}

// one batch of train_one_user_corruption for users [s0, s0+nb), corruption cidx
struct Batch { uint64_t s0; uint32_t nb; uint32_t cidx; uint64_t E; };

inline uint32_t units_of(const cdae_hip* h, const Batch& b) { return h->h_unit_ptr[b.s0 + b.nb] - h->h_unit_ptr[b.s0]; }
// item shard, sampled decode: units over the WHOLE rows (sample_kernel, hidden_gather_kernel)
inline uint32_t gunits_of(const cdae_hip* h, const Batch& b) { return h->h_gunit_ptr[b.s0 + b.nb] - h->h_gunit_ptr[b.s0]; }

// K1 + sort on the prep stream into example-buffer set `b`
// lane 1: the second prep stream with the second half of the sort workspace (two batches are prepared side by side)
int prep_batch(cdae_hip* h, int b, const Batch& bt, uint64_t seed, uint32_t epoch, int lane = 0, uint64_t prof_q = ~0ull) {
  using namespace cdae;
  cdae_hip::ExBuf& x = h->ex[b];
  hipStream_t st = lane ? h->prep2 : h->prep;
  void* sort_tmp = (char*)h->d_sort_tmp + (lane ? h->sort_tmp_stride : 0);
  const uint32_t I = (uint32_t)h->I;
  Prof pr;
  HIPCHK(hipStreamWaitEvent(st, x.released, 0));               // the batch that last used this set is done with it
  CHK(pr.begin(h, F_SAMPLE, st, prof_q));
  const bool shs = h->shard_sampled();
  const uint32_t n_units = shs ? gunits_of(h, bt) : units_of(h, bt);
  const bool bucket = h->bucket_sort && x.key16;      // bucket_sort_kernel writes every entry of the segment tables itself: nothing to clear
  const bool use_cells = bucket && x.cells && n_units <= h->cell_units;   // sample_kernel routes the examples into the ranges' cells
  if (use_cells) x.cell_tag = x.cell_tag + 1u ? x.cell_tag + 1u : 1u;     // a fresh non-zero tag per batch: nobody has to clear the overflow word
  if (n_units == 0) {            // (an item shard none of whose rows the batch's users rated: only the per-batch clears)
    HIPCHK(hipMemsetAsync(x.seg, 0, 4 * (size_t)I * sizeof(uint32_t), st));
    HIPCHK(hipMemsetAsync(x.dup_count, 0, cdae::DUP_STRIPES * sizeof(uint32_t), st));
  } else if (h->mf) {
    hipLaunchKernelGGL(mf_sample_kernel, dim3((n_units + 3) / 4), dim3(256), 0, st, h->hp, h->mf == 2 ? 1u : 0u, h->d_row_ptr, h->d_col,
                       h->d_unit_ptr + bt.s0, n_units, bt.s0, bt.nb, seed, epoch, x.item, x.val, x.key16, x.seg, bucket ? 0u : 2u * I, x.dup_count,
                       x.dup_of_ex, h->d_unit_user, x.wg_state);
  } else if (shs) {
    // item shard, sampled decode: the single-GPU example list of the batch from the WHOLE rows, other shards' examples VOID
    hipLaunchKernelGGL(sample_kernel, dim3((n_units + 3) / 4), dim3(256), 0, st, h->hp, h->d_grow_ptr, h->d_gcol,
                       h->d_gunit_ptr + bt.s0, n_units, bt.s0, bt.nb, bt.cidx, seed, epoch, x.item, x.val, x.key16,
                       x.seg, h->counting_sort || bucket ? 0u : 4u * I, x.dup_count, x.dup_of_ex, h->d_gunit_user,
                       x.wg_state, (const uint32_t*)nullptr, (uint32_t)h->item0, I, (uint32_t)h->I_global,
                       (const uint16_t*)(use_cells ? h->d_range_of : nullptr), (const uint32_t*)h->d_bucket_cut, h->bucket_ranges,
                       use_cells ? x.cells : (uint32_t*)nullptr, x.cell_flag, x.cell_tag);
  } else
  hipLaunchKernelGGL(sample_kernel, dim3((n_units + 3) / 4), dim3(256), 0, st, h->hp, h->d_row_ptr, h->d_col,
                     h->d_unit_ptr + bt.s0, n_units, bt.s0, bt.nb, bt.cidx, seed, epoch, x.item, x.val, x.key16,
                     x.seg, h->counting_sort || bucket ? 0u : 4u * I, x.dup_count, x.dup_of_ex, h->d_unit_user,
                     x.wg_state, (const uint32_t*)h->d_gpos, 0u, 0u, 0u,
                     (const uint16_t*)(use_cells ? h->d_range_of : nullptr), (const uint32_t*)h->d_bucket_cut, h->bucket_ranges,
                     use_cells ? x.cells : (uint32_t*)nullptr, x.cell_flag, x.cell_tag);
  CHK(pr.end());
  CHK(pr.begin(h, F_SORT, st, prof_q));
  const dim3 seg_grid((uint32_t)((bt.E + 256 * SEG_PER_THREAD - 1) / (256 * SEG_PER_THREAD)));
  if (bt.E == 0) {
    // nothing to order
  } else if (h->mf_seq && !h->prep_force_sort) {
    // the in-place loop reads the user-major list only: no item-major order, no segment table
  } else if (bucket) {
    // item-major order, segment tables, duplicate marks: ONE narrow launch (cdae_sort_kernels.hpp bucket_sort_kernel)
    if (!h->bucket_attr_set) {
      HIPCHK(hipFuncSetAttribute((const void*)bucket_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BK_LDS_BYTES));
      h->bucket_attr_set = true;
    }
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(h->bucket_ranges), dim3(BK_THREADS), BK_LDS_BYTES, st, (const uint16_t*)x.key16, (const uint64_t*)x.val,
                       (uint32_t)bt.E, (const uint32_t*)h->d_bucket_cut, x.wg_state, (const uint32_t*)(use_cells ? x.cells : nullptr), n_units,
                       (const uint32_t*)x.cell_flag, x.cell_tag, x.seg, x.seg + I, (const uint32_t*)h->d_rank_of,
                       x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I, x.sorted_val, x.bucketed, x.dup_count, h->dup_cap, x.dup_of_pos, x.dup_of_ex,
                       h->dup_stripes, h->d_fused_err);
  } else if (h->counting_sort) {
    // item-major order by counting, four launches (cdae_sort_kernels.hpp)
    const uint32_t n_tiles = (uint32_t)((bt.E + TILE_EX - 1) / TILE_EX);
    const size_t tile_lds = (size_t)I * sizeof(uint32_t);
    if (tile_lds + 1024 > 64 * 1024 && !h->tile_attr_set) {
      HIPCHK(hipFuncSetAttribute((const void*)tile_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)tile_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds + 1024));
      h->tile_attr_set = true;
    }
    hipLaunchKernelGGL(tile_hist_kernel, dim3(n_tiles), dim3(TILE_THREADS), tile_lds / 2 + 4, st, x.item, (uint32_t)bt.E, I, x.tile_hist);
    hipLaunchKernelGGL(item_tile_scan_kernel, dim3((I + 255) / 256), dim3(256), 0, st, x.tile_hist, n_tiles, I, x.item_count, x.rank, x.block_total);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(n_tiles), dim3(TILE_THREADS), tile_lds + 128 * sizeof(uint32_t), st, x.item, x.val,
                       (uint32_t)bt.E, I, x.tile_hist, x.item_count, x.rank, x.block_total, x.prefix, x.seg, x.seg + I, x.dup_count, x.bucketed,
                       (const uint32_t*)h->d_rank_of, x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I);
    hipLaunchKernelGGL(segment_sort_kernel, dim3((I + SEGSORT_ITEMS - 1) / SEGSORT_ITEMS), dim3(SEGSORT_THREADS), 0, st, I, x.prefix, x.bucketed,
                       x.sorted_val, x.item_count, x.dup_count, h->dup_cap, x.dup_of_pos, x.dup_of_ex, h->dup_stripes);
  } else if (x.key16) {
    // 16-bit keys: rocPRIM picks onesweep (2 digit passes) instead of block sort + log2(tiles) merge passes
    HIPCHK(rocprim::radix_sort_pairs(sort_tmp, h->sort_tmp_bytes, x.key16, x.sorted_key16, x.val, x.sorted_val,
                                     (size_t)bt.E, 0u, (unsigned)h->sort_bits, st));
    hipLaunchKernelGGL(segment_kernel<uint16_t>, seg_grid, dim3(256), 0, st, x.sorted_key16, x.sorted_val, (uint32_t)bt.E,
                       x.seg, x.seg + I, x.dup_count, h->dup_cap, x.dup_of_pos, x.dup_of_ex, h->dup_stripes,
                       (const uint32_t*)h->d_rank_of, x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I, shs ? I : 0xFFFFFFFFu);
  } else {
    HIPCHK(rocprim::radix_sort_pairs(sort_tmp, h->sort_tmp_bytes, x.item, x.sorted_item, x.val, x.sorted_val,
                                     (size_t)bt.E, 0u, (unsigned)h->sort_bits, st));
    hipLaunchKernelGGL(segment_kernel<uint32_t>, seg_grid, dim3(256), 0, st, x.sorted_item, x.sorted_val, (uint32_t)bt.E,
                       x.seg, x.seg + I, x.dup_count, h->dup_cap, x.dup_of_pos, x.dup_of_ex, h->dup_stripes,
                       (const uint32_t*)h->d_rank_of, x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I, shs ? I : 0xFFFFFFFFu);
  }
  CHK(pr.end());
  if (h->cfg.full_output && h->d_bits_train) {
    // fused full-output decode: its targets — one bit per (batch user, item) — depend on the data set only, so they are
    // built here, beside the previous batch's training, instead of in front of the decode (memset + kernel: 22 us per batch)
    const uint32_t words = (I + 31) / 32;
    uint32_t* bits = h->d_bits_train + (size_t)b * h->bits_stride;
    hipLaunchKernelGGL(rated_bits_kernel, dim3((bt.nb + 3) / 4), dim3(256), 0, st, h->d_row_ptr, h->d_col, bt.s0, bt.nb, words, bits);
  }
  HIPCHK(hipEventRecord(x.ready, st));
  HIPCHK(hipGetLastError());
  return 0;
}

// K3 on the main stream: the decode of example-buffer set `x` over this handle's item rows (shared by the single-handle step and
// the sampled item-shard step)
// Geometry of the fused launch, once per handle: how many workgroups of the launch a CU holds decides the blocker rounds.
template <int NV, int NT>
int fused_geometry(cdae_hip* h, uint32_t hot, uint32_t I) {
  using namespace cdae;
  FusedGeom g{};
  g.hot_wgs = (hot + 3) / 4;
  g.stride = h->num_cus;
  g.blocked = std::min<uint32_t>(g.hot_wgs, FUSED_BLOCK_MAX);
  int per_cu = 0;
  const bool ce = h->cfg.loss_type == CDAE_LOSS_CROSS_ENTROPY, ada = h->cfg.using_adagrad != 0;
  if (ce && ada) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_gather_kernel<NV, NT, 5, true>, 256, 0));
  else if (ce) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_gather_kernel<NV, NT, 5, false>, 256, 0));
  else if (ada) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_gather_kernel<NV, NT, 0, true>, 256, 0));
  else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_gather_kernel<NV, NT, 0, false>, 256, 0));
  // Blockers are OFF in the shipped launch (rounds = 0): measured, they buy nothing (0.0924 with, 0.0919 without: the long-lived four-row
  // wavefronts are not the popular rows' neighbours), and a blocker that waits in vain is the one wait of this launch that costs time
  // without raising an error.  CDAE_FUSED_BLOCK_ROUNDS = n (developer switch) brings them back; -1 = one round per resident workgroup.
  g.rounds = 0;
  if (const char* ev = DEV_ENV("CDAE_FUSED_BLOCK_ROUNDS")) g.rounds = std::atoi(ev) < 0 ? (uint32_t)std::max(0, std::min(per_cu, 8) - 1) : (uint32_t)std::atoi(ev);
  // The four-row groups are dealt to (CU, SIMD) bins so that every SIMD gets about the same number of example steps (longest group first,
  // each to the lightest bin): a group lasts as long as its longest row, the SIMDs are VALU-bound on these wavefronts, and in index
  // order (round 5) the SIMDs that held the most popular groups finished 10-15 us after the others — which every gather wavefront of
  // this launch then waits for.  Workgroup b's wavefront w runs on SIMD w of CU b mod S (observed placement: only the balance depends
  // on it): the k-th group of bin (cu, w) goes to workgroup cu + k S.  The popular rows' CUs (and their blockers') take no group.
  const uint32_t n_groups = (I - hot + 3) / 4, S = g.stride;
  std::vector<float> len(n_groups, 0.f);
  for (uint32_t q = 0; q < n_groups; ++q)
    for (uint32_t j = 0; j < 4 && hot + 4 * q + j < I; ++j) len[q] = std::max(len[q], h->h_rank_len[hot + 4 * q + j]);
  std::vector<uint32_t> by_len(n_groups);
  std::iota(by_len.begin(), by_len.end(), 0u);
  std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b2) { return len[a] > len[b2]; });
  const uint32_t cu0 = std::min(g.hot_wgs, S > 8 ? S - 8 : 0u);                 // CUs [0, cu0) belong to the popular rows
  const uint32_t n_bins = (S - cu0) * 4;
  std::vector<std::vector<uint32_t>> bin(n_bins);
  {
    // lightest bin first: a heap of (load, bin)
    using Ent = std::pair<float, uint32_t>;
    std::vector<Ent> heap;
    for (uint32_t i = 0; i < n_bins; ++i) heap.push_back({0.f, i});
    auto cmp = [](const Ent& a, const Ent& b2) { return a.first > b2.first || (a.first == b2.first && a.second > b2.second); };
    std::make_heap(heap.begin(), heap.end(), cmp);
    const bool balance = DEV_ENV("CDAE_FUSED_INDEX_ORDER") == nullptr;                   // (developer switch: round 5's index order)
    for (uint32_t k = 0; k < n_groups; ++k) {
      const uint32_t q = balance ? by_len[k] : k;
      if (!balance) { bin[k % n_bins].push_back(q); continue; }
      std::pop_heap(heap.begin(), heap.end(), cmp);
      Ent e = heap.back();
      bin[e.second].push_back(q);
      e.first += 4.f + len[q];                                                    // (+ a wavefront's fixed cost: prologue, epilogue)
      heap.back() = e;
      std::push_heap(heap.begin(), heap.end(), cmp);
    }
  }
  size_t rounds_cold = 0;
  for (auto& v : bin) rounds_cold = std::max(rounds_cold, v.size());
  // cold round k of CU cu is workgroup index cu + k S, except that on the popular CUs' indices there are blockers in rounds 1..rounds
  g.decode_wgs = (uint32_t)std::max<size_t>(rounds_cold, 1) * S;
  std::vector<uint32_t> map((size_t)g.decode_wgs * 4, 0xFFFFFFFFu);
  for (uint32_t i = 0; i < n_bins; ++i) {
    const uint32_t cu = cu0 + i / 4, w = i % 4;
    for (size_t k = 0; k < bin[i].size(); ++k) map[((size_t)k * S + cu) * 4 + w] = bin[i][k];
  }
  if (h->d_cold_map) { HIPCHK(hipFree(h->d_cold_map)); h->d_cold_map = nullptr; }
  CHK(dev_alloc(&h->d_cold_map, map.size()));
  HIPCHK(hipMemcpy(h->d_cold_map, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  g.cold_map = h->d_cold_map;
  h->fused_geo = g;
  h->fused_geo_set = true;
  return 0;
}

// fused != nullptr: the fused launch (decode_gather_kernel) with these gather arguments; the caller then launches no hidden_gather_kernel
int launch_decode(cdae_hip* h, cdae_hip::ExBuf& x, const cdae::GatherArgs* fused = nullptr) {
  using namespace cdae;
  hipStream_t st = h->stream;
  const uint32_t I = (uint32_t)h->I;
  const dim3 blk(256);
  const dim3 grid_rows((I + 3) / 4);
#define DECODE_TAIL h->d_item_order, x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I, x.sorted_val, h->d_Z, h->dec(), h->dec_ag(), \
                    h->P(CDAE_P_BP), h->P(CDAE_P_BP_AG), h->d_HG, h->d_G, h->d_D0, h->d_touched, x.dup_of_pos, h->d_dup_corr
#define DECODE_ARGS h->hp, DECODE_TAIL
#define DECODE_LA(NI_, L_, A_)                                                                                       \
  do {                                                                                                               \
    if (pad) hipLaunchKernelGGL((decode_rows_kernel<NI_, L_, A_, true>), grid_rows, blk, 0, st, DECODE_ARGS);        \
    else hipLaunchKernelGGL((decode_rows_kernel<NI_, L_, A_, false>), grid_rows, blk, 0, st, DECODE_ARGS);           \
  } while (0)
#define DECODE_NI(NI_)                                          \
  do {                                                          \
    if (ce && ada) DECODE_LA(NI_, 5, true);                     \
    else if (ce) DECODE_LA(NI_, 5, false);                      \
    else if (ada) DECODE_LA(NI_, 0, true);                      \
    else DECODE_LA(NI_, 0, false);                              \
  } while (0)
  // K <= 256: hot rows one per wavefront, all others four per wavefront (NV float4 pieces + NT tail scalars per lane)
#define DECODE_HY(NV_, NT_)                                                                                           \
  do {                                                                                                                \
    if (fused && !h->fused_geo_set) { CHK((fused_geometry<NV_, NT_>(h, hot, I))); geo = h->fused_geo; geo.hot_target = 4u * h->fused_seq; geo.hot_cnt = h->d_hot_cnt; \
                                      grid_fu = dim3(geo.decode_wgs + gather_wgs); }                                  \
    if (fused) {                                                                                                       \
      if (ce && ada) hipLaunchKernelGGL((decode_gather_kernel<NV_, NT_, 5, true>), grid_fu, blk, 0, st, h->hp, hot, geo, late, *fused, DECODE_TAIL);   \
      else if (ce) hipLaunchKernelGGL((decode_gather_kernel<NV_, NT_, 5, false>), grid_fu, blk, 0, st, h->hp, hot, geo, late, *fused, DECODE_TAIL);    \
      else if (ada) hipLaunchKernelGGL((decode_gather_kernel<NV_, NT_, 0, true>), grid_fu, blk, 0, st, h->hp, hot, geo, late, *fused, DECODE_TAIL);    \
      else hipLaunchKernelGGL((decode_gather_kernel<NV_, NT_, 0, false>), grid_fu, blk, 0, st, h->hp, hot, geo, late, *fused, DECODE_TAIL);            \
    }                                                                                                                  \
    else if (ce && ada) hipLaunchKernelGGL((decode_hybrid_kernel<NV_, NT_, 5, true>), grid_hy, blk, 0, st, h->hp, hot, late, DECODE_TAIL);   \
    else if (ce) hipLaunchKernelGGL((decode_hybrid_kernel<NV_, NT_, 5, false>), grid_hy, blk, 0, st, h->hp, hot, late, DECODE_TAIL);    \
    else if (ada) hipLaunchKernelGGL((decode_hybrid_kernel<NV_, NT_, 0, true>), grid_hy, blk, 0, st, h->hp, hot, late, DECODE_TAIL);    \
    else hipLaunchKernelGGL((decode_hybrid_kernel<NV_, NT_, 0, false>), grid_hy, blk, 0, st, h->hp, hot, late, DECODE_TAIL);            \
  } while (0)
#define DECODE_HY_NT(NV_)                                                                        \
  do {                                                                                           \
    if (nt == 1) DECODE_HY(NV_, 1); else if (nt == 2) DECODE_HY(NV_, 2); else DECODE_HY(NV_, 4); \
  } while (0)
  {
    const bool ce = h->cfg.loss_type == CDAE_LOSS_CROSS_ENTROPY, ada = h->cfg.using_adagrad != 0, pad = h->K < h->Kp;
    const uint32_t K = h->K;
    if (K <= 256 && !h->one_row_per_wave) {
      const uint32_t hot = std::min<uint32_t>(h->hot_rows, I);
      const uint32_t waves = hot + (I - hot + 3) / 4;
      const dim3 grid_hy((waves + 3) / 4);
      const DecodeLate late = h->decode_late();
      // fused launch (decode_gather_kernel): [popular rows] [the other rows, with the blockers' indices among them] [gather]
      FusedGeom geo = h->fused_geo;
      if (fused) {
        geo.hot_target = 4u * ++h->fused_seq;
        geo.hot_cnt = h->d_hot_cnt;
      }
      const uint32_t gather_wgs = fused ? 8u * ((fused->n_units + 3u) / 4u) : 0u;
      dim3 grid_fu(geo.decode_wgs + gather_wgs);
      const uint32_t nv = K / 64, tail = K % 64;
      const uint32_t nt = tail == 0 ? 0u : (tail < 16 ? 1u : (tail < 32 ? 2u : 4u));   // 16 nt > tail: room for b'
      if (nt == 0) {
        switch (nv) { case 1: DECODE_HY(1, 0); break; case 2: DECODE_HY(2, 0); break; case 3: DECODE_HY(3, 0); break; default: DECODE_HY(4, 0); break; }
      } else {
        switch (nv) { case 0: DECODE_HY_NT(0); break; case 1: DECODE_HY_NT(1); break; case 2: DECODE_HY_NT(2); break; default: DECODE_HY_NT(3); break; }
      }
    } else {
      switch (h->NI) { case 1: DECODE_NI(1); break; case 2: DECODE_NI(2); break; case 4: DECODE_NI(4); break; default: DECODE_NI(8); break; }
    }
  }
#undef DECODE_HY_NT
#undef DECODE_HY
#undef DECODE_LA
#undef DECODE_NI
#undef DECODE_ARGS
#undef DECODE_TAIL
  HIPCHK(hipGetLastError());
  return 0;
}

// K2..K5 on the main stream from example-buffer set `b`.
// explicit_in != nullptr: single user whose example list (already sorted into set b) and input set come from the caller
int compute_batch(cdae_hip* h, int b, const Batch& bt, uint64_t seed, uint32_t epoch,
                  const uint32_t* explicit_in = nullptr, uint32_t n_explicit = 0) {
  using namespace cdae;
  cdae_hip::ExBuf& x = h->ex[b];
  hipStream_t st = h->stream;
  const uint32_t I = (uint32_t)h->I, nb = bt.nb;
  const uint64_t s0 = bt.s0;
  const dim3 blk(256);
  const dim3 grid_users((nb + 3) / 4), grid_rows((I + 3) / 4);
  Prof pr;

  h->hp.trace_odd = (uint32_t)(h->seq & 1);
  CHK(pr.begin(h, F_ENCODE, st));
  // explicit mode: one user, one unit (the caller's lists need not follow the num_neg proportion)
  const uint32_t n_units = explicit_in ? 1u : units_of(h, bt);
  const uint32_t* uptr = explicit_in ? h->d_uptr_tmp : h->d_unit_ptr + s0;
  const dim3 grid_units((n_units + 3) / 4);
  // The fused launch (decode + gather, decode_gather_kernel): plain batches of a handle that has late rows.  The batch's encode then
  // fills G with G_PENDING (the gather wavefronts wait on it), whichever encode form runs.
  const uint32_t halves = h->gather_halves;
  const bool fused = h->fused_decode && !explicit_in && halves == 1u && n_units > 0u && bt.E > 0u && bt.E < (1ull << 32);
  float* const ghot = h->late_rows ? h->d_Ghot : nullptr;
  uint32_t* const gfill = fused ? reinterpret_cast<uint32_t*>(h->d_G) : nullptr;
  const uint32_t n_fill = fused ? (uint32_t)bt.E : 0u;
  if (!explicit_in && !h->encode_two_launches && nb <= h->encode_users_max) {
    // one launch: a workgroup per user (encode_users_kernel)
    DISPATCH_NI(h->NI, encode_users_kernel, dim3(nb), dim3(ENC_WAVES * WAVE), 0, st, h->hp, h->d_row_ptr, h->d_col, h->P(CDAE_P_W), uptr, s0, nb,
                bt.cidx, seed, epoch, h->d_Wu, h->P(CDAE_P_B), h->d_Z, h->d_Dz, h->d_HG, h->d_Uu, h->d_Ssum,
                (BF16_T*)nullptr, (BF16_T*)nullptr, 0u, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr,
                ghot, h->d_hotdup, gfill, n_fill);
  } else {
    DISPATCH_NI(h->NI, encode_partial_kernel, grid_units, blk, 0, st, h->hp, h->d_row_ptr, h->d_col, h->P(CDAE_P_W), uptr, n_units,
                (const uint32_t*)nullptr, s0, nb, 1, CDAE_STREAM_CORRUPT, bt.cidx, seed, epoch, h->d_Hpart, explicit_in, n_explicit,
                explicit_in ? (const uint32_t*)nullptr : (const uint32_t*)h->d_unit_user);
    DISPATCH_NI(h->NI, encode_finish_kernel, grid_users, blk, 0, st, h->hp, h->d_Hpart, uptr, h->d_Wu, h->P(CDAE_P_B),
                (const uint32_t*)nullptr, s0, nb, 1, h->d_Z, h->d_Dz, h->d_HG, h->d_Uu, h->d_Ssum,
                (BF16_T*)nullptr, (BF16_T*)nullptr, 0u, ghot, h->d_hotdup, gfill, n_fill);
  }
  CHK(pr.end());

  if (!h->skip_ready_wait) HIPCHK(hipStreamWaitEvent(st, x.ready, 0));
  const GatherArgs ga{h->d_row_ptr, uptr, n_units, s0, nb, x.item, h->d_G, h->d_D0, h->d_HGpart, explicit_in ? (uint32_t)bt.E : 0u, x.dup_of_ex,
                      h->d_dup_corr, explicit_in ? (const uint32_t*)nullptr : (const uint32_t*)h->d_unit_user, halves,
                      h->late_rows ? (const uint32_t*)h->d_late_bits : (const uint32_t*)nullptr, h->late_words, h->d_fused_err};
  CHK(pr.begin(h, F_DECODE, st));
  CHK(launch_decode(h, x, fused ? &ga : nullptr));
  CHK(pr.end());

  CHK(pr.begin(h, F_HIDDEN, st));
  if (!fused)
    DISPATCH_NI(h->NI, hidden_gather_kernel, dim3(8 * halves * ((n_units + 3) / 4)), blk, 0, st, h->hp, ga.row_ptr, uptr, n_units, s0, nb,
                x.item, h->d_G, h->d_D0, h->d_HGpart, ga.explicit_examples, x.dup_of_ex, h->d_dup_corr, ga.unit_user, halves,
                ga.late_bits, ga.late_words);
  DISPATCH_NI(h->NI, hidden_finish_kernel, dim3(nb), blk, 0, st, h->hp, uptr, n_units, s0, nb, h->d_HGpart, h->d_Dz, h->d_HG,
              h->d_Wu, h->d_Wu_ag, 8u * halves, h->d_Uu, h->d_Uu_ag, h->d_Ssum, h->d_delta_rows, (const float*)nullptr, h->late_finish());
  CHK(pr.end());
  // input rows + (leading workgroups) the strictly sequential hidden-bias recurrence: both need only delta
  CHK(pr.begin(h, F_INPUT, st));
  {
    const uint32_t bias_blocks = (h->Kp + 255u) / 256u;
    DISPATCH_NI(h->NI, input_rows_kernel, dim3(bias_blocks + (I + 3) / 4), blk, 0, st, h->hp, h->d_item_order, x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I,
                x.sorted_val, h->d_Z, h->d_HG, h->d_G, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), CDAE_TOUCHED_ARG, nb, h->P(CDAE_P_B),
                h->P(CDAE_P_B_AG), h->delta_rows());
  }
  CHK(pr.end());
  HIPCHK(hipEventRecord(x.released, st));
  HIPCHK(hipGetLastError());
  return 0;
}

// LDS-staged NT GEMM: the 256 x 128 three-stage kernel where the rows allow it (M % 256 == 0), else the 128 x 128 one.
template <int EPI>
int launch_gemm_lds(cdae_hip* h, hipStream_t st, const __bf16* A, const __bf16* Bm, uint32_t M, uint32_t N, uint32_t Kd, uint32_t lda,
                    uint32_t ldb, uint32_t kps, const cdae::GemmEpilogue& ep, uint32_t splits, uint32_t mode) {
  using namespace cdae;
  const uint32_t Nt = (N + 127) / 128;
  if (M % 256 == 0 && N % 256 == 0 && !h->gemm_two_stage && !h->gemm_narrow) {
    // 256 x 256 tiles (round 3): 128 flop per byte staged into LDS instead of 85
    if (!h->gemmw_attr_set[EPI]) {
      HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_bf16_ldsw_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemmw_lds_bytes()));
      h->gemmw_attr_set[EPI] = true;
    }
    const GemmGrid gg{M / 256, N / 256, splits, mode};
    hipLaunchKernelGGL((gemm_nt_bf16_ldsw_kernel<EPI>), dim3(gg.workgroups()), dim3(512), gemmw_lds_bytes(), st, A, Bm, M, N, Kd, lda, ldb,
                       kps, ep, gg);
    return 0;
  }
  if (M % 256 == 0 && !h->gemm_two_stage) {
    if (!h->gemm3_attr_set[EPI]) {     // per handle: the attribute belongs to the (function, device) pair
      HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_bf16_lds3_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm3s_lds_bytes()));
      h->gemm3_attr_set[EPI] = true;
    }
    const GemmGrid gg{M / 256, Nt, splits, mode};
    hipLaunchKernelGGL((gemm_nt_bf16_lds3_kernel<EPI>), dim3(gg.workgroups()), dim3(512), gemm3s_lds_bytes(), st, A, Bm, M, N, Kd, lda, ldb,
                       kps, ep, gg);
  } else {
    const GemmGrid gg{M / 128, Nt, splits, mode};
    hipLaunchKernelGGL((gemm_nt_bf16_lds_kernel<EPI>), dim3(gg.workgroups() + (EPI == EPI_STORE ? ep.bias_blocks : 0u)), dim3(256), 0, st, A, Bm, M, N,
                       Kd, lda, ldb, kps, ep, gg);
  }
  return 0;
}
// does launch_gemm_lds take the 128 x 128 kernel — the one that can host a bias role (GemmEpilogue::bias_blocks) — for this shape?
bool gemm_lds_is_128(const cdae_hip* h, uint32_t M, uint32_t N) {
  if (M % 256 == 0 && N % 256 == 0 && !h->gemm_two_stage && !h->gemm_narrow) return false;
  return !(M % 256 == 0 && !h->gemm_two_stage);
}

// contraction split of GEMM 2 (hg = G D, K > 256 / unfused path): about 2048 workgroups in all, splits a multiple of 64 items
uint32_t gemm2_k_per_split(const cdae_hip* h) {
  const uint32_t tiles2 = ((h->Kp + 127) / 128) * (h->Bp / 128);
  const uint32_t want = std::max<uint32_t>(1, std::min<uint32_t>(h->Ip / 64, (2048 + tiles2 - 1) / tiles2));
  return (((h->Ip + want - 1) / want + 63) / 64) * 64;
}

// GEMM 2 of the K > 256 path from G^T and the row-major decoder image (gemm_tn_bf16_kernel): then GEMM 1 writes no G and nothing reads D^T
bool gemm2_tn_path(const cdae_hip* h) {
  // (rows and columns of GEMM 1 in multiples of 256: it is then the 256 x 256-tile kernel, whose loss epilogue knows how to leave G out)
  return h->Kp > 256 && h->Kp % 256 == 0 && h->Bp % 256 == 0 && h->Ip % 256 == 0 && !h->full_unfused_nt() && !h->gemm2_nt;
}
// GEMM 3 + row step in one launch (gemm3_rows_fused_kernel): Kp = 512 over item spaces >= 32768, i.e. BASELINE configs[4]'s path
bool rows_fused_path(const cdae_hip* h) {
  return h->Kp == 512 && h->I >= 32768 && h->Ip % cdae::FR_ITEMS == 0 && h->fused_images && !h->gemm_direct && !h->rows_separate;
}
int launch_rows_fused(cdae_hip* h, hipStream_t st, const cdae_hip::ExBuf& x, uint32_t nb, __bf16* Db) {
  using namespace cdae;
  const uint32_t I = (uint32_t)h->I;
#define FR_LAUNCH(ADA_, KH_)                                                                                                              \
  do {                                                                                                                                    \
    if (!h->fused_rows_attr_set[ADA_][KH_ - 1]) {                                                                                         \
      HIPCHK(hipFuncSetAttribute((const void*)gemm3_rows_fused_kernel<ADA_, KH_>, hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                 (int)fused_rows_lds_bytes<KH_>()));                                                                      \
      h->fused_rows_attr_set[ADA_][KH_ - 1] = true;                                                                                       \
    }                                                                                                                                     \
    const uint32_t tiles = h->Ip / FR_ITEMS, grid = KH_ == 1 ? tiles : 16u * ((tiles + 7u) / 8u);                                         \
    hipLaunchKernelGGL((gemm3_rows_fused_kernel<ADA_, KH_>), dim3(grid), dim3(256 / KH_), fused_rows_lds_bytes<KH_>(), st, h->hp,         \
                       (const __bf16*)h->d_ZTb, (const __bf16*)h->d_GTb, h->Bp, h->Bp, nb, (const uint8_t*)h->d_has_in,                   \
                       h->d_dD, h->dec(), h->dec_ag(),                                                                                    \
                       h->P(CDAE_P_BP), h->P(CDAE_P_BP_AG), h->d_touched, Db, h->Ip);                                                     \
  } while (0)
  if (h->cfg.using_adagrad) { if (h->rows_fused_kh == 1) FR_LAUNCH(true, 1); else FR_LAUNCH(true, 2); }
  else { if (h->rows_fused_kh == 1) FR_LAUNCH(false, 1); else FR_LAUNCH(false, 2); }
#undef FR_LAUNCH
  // the rows kept as inputs (tied weights: one step with dD + the summed input gradient)
  DISPATCH_NI(h->NI, full_rows_inputs_kernel, dim3((I + 255) / 256), dim3(256), 0, st, h->hp, h->d_has_in, (const uint32_t*)x.seg, (const uint32_t*)(x.seg + I),
              (const uint64_t*)x.sorted_val, (const float*)h->delta_rows(), (const float*)h->d_dD, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), Db, (__bf16*)nullptr, h->Ip);
  return 0;
}

// The unfused (K > 256, or CDAE_FULL_UNFUSED) forward product with its loss epilogue, the positive fix-up and the hidden-gradient
// product of one block — shared by the single-handle step (compute_batch_full) and the item shard's phase 1 (fs_phase1: the same
// launches over the shard's own item rows).  *parts / *rows: the slabs of HGpart holding the partial hg and their row count
// (0 parts: accumulated into d_HG by atomics, the CDAE_GEMM_DIRECT developer path).
int full_products_k512(cdae_hip* h, hipStream_t st, cdae_hip::ExBuf& x, const Batch& bt, uint32_t nb, uint32_t* parts, uint32_t* rows) {
  using namespace cdae;
  const uint32_t I = (uint32_t)h->I, Kp = h->Kp, Bp = h->Bp, Ip = h->Ip;
  const dim3 blk(256);
  const bool tn2 = gemm2_tn_path(h);                                      // GEMM 2 reads G^T and D: no G, no D^T
  GemmEpilogue ep{};
  ep.bp = h->P(CDAE_P_BP); ep.G = tn2 ? (__bf16*)nullptr : h->d_Gb; ep.ldg = Ip; ep.GT = h->d_GTb; ep.ldgt = Bp;
  ep.rows_live = nb; ep.cols_live = I; ep.loss_type = h->cfg.loss_type;
  // GEMM 1: Y = Z D^T (+ b'), g = loss'(y, 0) -> G [Bp x Ip] (unless GEMM 2 reads G^T), G^T [Ip x Bp]
  if (h->gemm_direct)
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPI_LOSS>), dim3(Ip / 128, Bp / 128, 1), blk, 0, st, h->d_Zb, h->d_Db, Bp, Ip, Kp, Kp, Kp,
                       Kp, ep);
  else if (tn2 && Kp == 512 && !h->gemm1_tiled) {
    // the z rows of 256 users in registers, only D staged (gemm1_loss_zreg_kernel): G^T alone, which is all GEMM 2 (TN) and GEMM 3 read
    const uint32_t user_tiles = Bp / 256, n_tiles = Ip / 128;
    const uint32_t item_groups = std::min<uint32_t>(((std::max<uint32_t>(1u, 256u / user_tiles) + 7u) / 8u) * 8u, ((n_tiles + 7u) / 8u) * 8u);
    const uint32_t tiles_per_group = (n_tiles + item_groups - 1) / item_groups;
    const dim3 grid(8u * user_tiles * ((item_groups + 7u) / 8u));
    const bool ce = h->cfg.loss_type == CDAE_LOSS_CROSS_ENTROPY;
    if (!h->gemm1_zreg) {
      // round 5 default: the two wavefronts of a SIMD in opposite phases — one contracts while the other runs its loss epilogue (gemm1_loss_duo_kernel)
      if (!h->gemm1_duo_attr_set[ce]) {
        if (ce) HIPCHK(hipFuncSetAttribute((const void*)gemm1_loss_duo_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm1_duo_lds_bytes()));
        else HIPCHK(hipFuncSetAttribute((const void*)gemm1_loss_duo_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm1_duo_lds_bytes()));
        h->gemm1_duo_attr_set[ce] = true;
      }
      if (ce)
        hipLaunchKernelGGL(gemm1_loss_duo_kernel<5>, grid, dim3(512), gemm1_duo_lds_bytes(), st, (const __bf16*)h->d_Zb, (const __bf16*)h->d_Db,
                           (const float*)h->P(CDAE_P_BP), h->d_GTb, Bp, nb, I, Ip, user_tiles, item_groups, tiles_per_group);
      else
        hipLaunchKernelGGL(gemm1_loss_duo_kernel<0>, grid, dim3(512), gemm1_duo_lds_bytes(), st, (const __bf16*)h->d_Zb, (const __bf16*)h->d_Db,
                           (const float*)h->P(CDAE_P_BP), h->d_GTb, Bp, nb, I, Ip, user_tiles, item_groups, tiles_per_group);
    } else {
    if (!h->gemm1_zreg_attr_set[ce]) {
      if (ce) HIPCHK(hipFuncSetAttribute((const void*)gemm1_loss_zreg_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm1_zreg_lds_bytes()));
      else HIPCHK(hipFuncSetAttribute((const void*)gemm1_loss_zreg_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm1_zreg_lds_bytes()));
      h->gemm1_zreg_attr_set[ce] = true;
    }
    if (ce)
      hipLaunchKernelGGL(gemm1_loss_zreg_kernel<5>, grid, dim3(512), gemm1_zreg_lds_bytes(), st, (const __bf16*)h->d_Zb, (const __bf16*)h->d_Db,
                         (const float*)h->P(CDAE_P_BP), h->d_GTb, Bp, nb, I, Ip, user_tiles, item_groups, tiles_per_group);
    else
      hipLaunchKernelGGL(gemm1_loss_zreg_kernel<0>, grid, dim3(512), gemm1_zreg_lds_bytes(), st, (const __bf16*)h->d_Zb, (const __bf16*)h->d_Db,
                         (const float*)h->P(CDAE_P_BP), h->d_GTb, Bp, nb, I, Ip, user_tiles, item_groups, tiles_per_group);
    }
  } else
    CHK(launch_gemm_lds<EPI_LOSS>(h, st, h->d_Zb, h->d_Db, Bp, Ip, Kp, Kp, Kp, Kp, ep, 1, 0));
  HIPCHK(hipStreamWaitEvent(st, x.ready, 0));
  if (bt.E)
    hipLaunchKernelGGL(full_positive_fixup_kernel, dim3((uint32_t)((bt.E + 255) / 256)), blk, 0, st, x.item, x.val, (uint32_t)bt.E,
                       h->cfg.loss_type == CDAE_LOSS_CROSS_ENTROPY ? 1.f : 2.f, tn2 ? (__bf16*)nullptr : h->d_Gb, Ip, h->d_GTb, Bp,
                       rows_fused_path(h) ? h->d_has_in : (uint8_t*)nullptr);
  // GEMM 2: hg = G D  (contraction over items, split).  Every split stores its partial [Bp x Kp] product into its own slab of
  // HGpart and the consumer adds the slabs in fixed order: deterministic (the first version accumulated with fp32 atomics into
  // HG, whose order — and therefore rounding — changed from run to run)
  const uint32_t kps = gemm2_k_per_split(h);
  GemmEpilogue e2{};
  if (h->gemm_direct) {
    e2.Cout = h->d_HG; e2.ldc = Kp; e2.rows_live = nb;
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPI_ATOMIC>), dim3((Kp + 127) / 128, Bp / 128, (Ip + 2047) / 2048), blk, 0, st, h->d_Gb,
                       h->d_DTb, Bp, Kp, Ip, Ip, Ip, 2048u, e2);
    *parts = 0; *rows = nb;
    return 0;
  }
  const uint32_t splits = (Ip + kps - 1) / kps;
  e2.Cout = h->d_HGpart; e2.ldc = Kp; e2.rows_live = nb; e2.split_stride = (size_t)Bp * Kp;
  if (tn2) {                                                     // sum over items of G^T[item][user] D[item][k]: both images as they are
    const GemmGrid gg{Bp / 256, Kp / 256, splits, 2};
#define GEMM_TN(ROWS_, NST_)                                                                                                              \
  do {                                                                                                                                    \
    constexpr size_t lds_ = gemm_tn_lds_bytes<ROWS_, NST_>();                                                                             \
    if (!h->gemm_tn_attr_set) {                                                                                                           \
      HIPCHK(hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<ROWS_, NST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));  \
      h->gemm_tn_attr_set = true;                                                                                                         \
    }                                                                                                                                     \
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<ROWS_, NST_>), dim3(gg.workgroups()), dim3(512), lds_, st,                                    \
                       (const __bf16*)h->d_GTb, (const __bf16*)h->d_Db, Bp, Kp, Ip, Bp, Kp, kps, e2, gg);                                 \
  } while (0)
    // two 64-row stages; CDAE_GEMM2_STAGES=3 / 4: the same 136 KiB as 32-row stages with two / three of them in flight behind the one
    // being contracted — bit-identical, 0.97 against 0.94 ms at 1 M items (A/B switch; cdae_full_kernels.hpp has the table)
    if (h->gemm2_stages == 3) GEMM_TN(32, 3);
    else if (h->gemm2_stages == 4) GEMM_TN(32, 4);
    else GEMM_TN(64, 2);
#undef GEMM_TN
  } else {
    CHK(launch_gemm_lds<EPI_STORE>(h, st, h->d_Gb, h->d_DTb, Bp, Kp, Ip, Ip, Ip, kps, e2, splits, 2));
  }
  *parts = splits; *rows = Bp;
  return 0;
}

// Full-output decode of one batch (MFMA path, cdae_full_kernels.hpp).  The example list holds the positives only.
int compute_batch_full(cdae_hip* h, int b, const Batch& bt, uint64_t seed, uint32_t epoch) {
  using namespace cdae;
  cdae_hip::ExBuf& x = h->ex[b];
  hipStream_t st = h->stream;
  const uint32_t I = (uint32_t)h->I, nb = bt.nb, Kp = h->Kp, Bp = h->Bp, Ip = h->Ip;
  const uint64_t s0 = bt.s0;
  const dim3 blk(256);
  const dim3 grid_users((nb + 3) / 4);
  const uint32_t n_units = units_of(h, bt);
  const uint32_t* uptr = h->d_unit_ptr + s0;
  Prof pr;

  CHK(pr.begin(h, F_ENCODE, st));
  const bool two_launches = h->encode_two_launches || nb > h->encode_users_max;
  if (two_launches) {
    DISPATCH_NI(h->NI, encode_partial_kernel, dim3((n_units + 3) / 4), blk, 0, st, h->hp, h->d_row_ptr, h->d_col, h->P(CDAE_P_W), uptr,
                n_units, (const uint32_t*)nullptr, s0, nb, 1, CDAE_STREAM_CORRUPT, bt.cidx, seed, epoch, h->d_Hpart,
                (const uint32_t*)nullptr, 0u, (const uint32_t*)h->d_unit_user);
  }
  // bf16 copies D, D^T (rows >= I zero) of this batch: like the input gather they need the previous batch's row steps but not
  // its b recurrence, which may still be running on the aux stream — joined behind them, in front of its first consumer.
  // Small item spaces: converted together with Z behind the encode instead (one launch less; a launch is ~9 us there)
  // Round 3: for item spaces below 32768 the row step (full_rows_kernel) leaves the bf16 images of the decoder behind and the
  // encode writes those of z: no conversion launch in the steady state.  D is converted here only when something else wrote the
  // parameters (init, set_param, an exchange), Z only when the batch is shorter than the rows the images may hold.
  const bool rows_fused = rows_fused_path(h);                             // GEMM 3 + row step in one launch (Kp = 512, >= 32768 items)
  const bool rows_write_images = h->fused_images && I < 32768u;
  const bool rows_write_db = h->fused_images && I >= 32768u && !rows_write_images;   // full_rows_wave_kernel: the row-major image only
  const bool need_d = !(rows_write_images && h->db_valid) && !(rows_write_db && h->db_rows_valid);
  const bool tn2 = gemm2_tn_path(h);                                      // GEMM 2 reads G^T and D: no G, no D^T
  if (rows_write_db && h->db_rows_valid && !tn2)                          // D^T from the bf16 rows the row step left (2 GB instead of 4 at 1 M x 512)
    hipLaunchKernelGGL(bf16_transpose_kernel, dim3(Kp / 64, Ip / 64), blk, 0, st, (const __bf16*)h->d_Db, I, Kp, Ip, h->d_DTb);
  const bool z_in_encode = h->fused_images && h->zb_rows == nb;
  const bool pair_copy = Ip <= 65536 && !h->full_separate_copies && need_d && !z_in_encode;
  if (need_d && !pair_copy) hipLaunchKernelGGL(to_bf16_transpose_kernel, dim3(Kp / 64, Ip / 64), blk, 0, st, h->dec(), I, Kp, Kp, Ip, h->d_Db, h->d_DTb);
  CHK(join_aux(h));
  __bf16* zb = z_in_encode ? h->d_Zb : nullptr;
  __bf16* ztb = z_in_encode ? h->d_ZTb : nullptr;
  if (two_launches) {
    DISPATCH_NI(h->NI, encode_finish_kernel, grid_users, blk, 0, st, h->hp, h->d_Hpart, uptr, h->d_Wu, h->P(CDAE_P_B),
                (const uint32_t*)nullptr, s0, nb, 1, h->d_Z, h->d_Dz, h->d_HG, h->d_Uu, h->d_Ssum, zb, ztb, Bp);
  } else {                       // one launch: a workgroup per user (encode_users_kernel)
    DISPATCH_NI(h->NI, encode_users_kernel, dim3(nb), dim3(ENC_WAVES * WAVE), 0, st, h->hp, h->d_row_ptr, h->d_col, h->P(CDAE_P_W), uptr, s0, nb,
                bt.cidx, seed, epoch, h->d_Wu, h->P(CDAE_P_B), h->d_Z, h->d_Dz, h->d_HG, h->d_Uu, h->d_Ssum, zb, ztb, Bp);
  }
  CHK(pr.end());

  CHK(pr.begin(h, F_DECODE, st));
  // bf16 operand copies Z, Z^T (rows >= nb zero) where the encode did not write them
  if (pair_copy)
    hipLaunchKernelGGL(to_bf16_transpose_pair_kernel, dim3(Kp / 64, Ip / 64 + Bp / 64), blk, 0, st, (const float*)h->dec(), I, Ip, h->d_Db, h->d_DTb,
                       (const float*)h->d_Z, nb, Bp, h->d_Zb, h->d_ZTb, Kp, Kp);
  else if (!z_in_encode)
    hipLaunchKernelGGL(to_bf16_transpose_kernel, dim3(Kp / 64, Bp / 64), blk, 0, st, h->d_Z, nb, Kp, Kp, Bp, h->d_Zb, h->d_ZTb);
  h->zb_rows = nb;
  const bool fused = Kp <= 256 && !h->full_unfused;
  uint32_t hg_parts = 0, hg_rows = nb;           // slabs of HGpart holding hg and their row count (0: accumulated into HG by atomics)
  if (fused) {
    // targets: one bit per (batch user, item); then forward + loss' + hidden gradient in one launch (cdae_full_kernels.hpp)
    const uint32_t words = (I + 31) / 32, slices = h->full_slices, tiles = Ip / (32 * FUSED_SUB);   // staged steps of 64 items
    const uint32_t* bits = h->d_bits_train + (size_t)b * h->bits_stride;      // built by prep_batch on the prep stream
    HIPCHK(hipStreamWaitEvent(st, x.ready, 0));
    const uint32_t tps = (tiles + slices - 1) / slices;
    const dim3 grid(slices, Bp / 128);
    const size_t lds = full_fused_lds_bytes(Kp);
#define FUSED_LAUNCH2(NKS_, L_)                                                                                                         \
  do {                                                                                                                                  \
    if (!h->fused_attr_set) {      /* per handle (the attribute belongs to the (function, device) pair): a runtime call per batch otherwise */ \
      HIPCHK(hipFuncSetAttribute((const void*)full_decode_fused_kernel<NKS_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      h->fused_attr_set = true;                                                                                                          \
    }                                                                                                                                     \
    hipLaunchKernelGGL((full_decode_fused_kernel<NKS_, L_>), grid, blk, lds, st, h->hp, h->d_Zb, h->d_Db, h->d_DTb, Ip, h->P(CDAE_P_BP), \
                       bits, words, nb, tps, h->d_GTb, Bp, h->d_HGpart);                                                     \
  } while (0)
#define FUSED_LAUNCH(NKS_)                                                                       \
  do {                                                                                           \
    if (h->cfg.loss_type == CDAE_LOSS_CROSS_ENTROPY) FUSED_LAUNCH2(NKS_, 5); else FUSED_LAUNCH2(NKS_, 0); \
  } while (0)
    if ((uint64_t)Ip * Bp > 0xFFFFFFFFull) return fail("full-output decode: G^T of %u x %u exceeds 2^32 elements; lower batch_users", Ip, Bp);
    switch (Kp) { case 64: FUSED_LAUNCH(4); break; case 128: FUSED_LAUNCH(8); break; default: FUSED_LAUNCH(16); break; }
#undef FUSED_LAUNCH2
#undef FUSED_LAUNCH
    hg_parts = slices;
  } else {
  CHK(full_products_k512(h, st, x, bt, nb, &hg_parts, &hg_rows));
  }
  // Small item spaces, short blocks (round 4): everything on ONE stream, the b recurrence as the leading workgroups of the row launch
  // (full_rows_kernel's bias role).  A hand-off between two streams through an event costs ~15 us on this part against 2.7 us for
  // a launch boundary (tools/grid_barrier_cost.hip), and at <= one_stream_max users per block the recurrence (41 ns per user) is
  // shorter than the two hand-offs that would put it beside the row step.
  const bool one_stream = !rows_fused && I < 32768u && nb <= h->full_one_stream_max && hg_parts != 0 && !h->cfg.linear_function;   // (the gate's rows receive Uu (.) delta, b the plain delta)
  if (one_stream) {
    Prof pa;
    CHK(pa.begin(h, F_HIDDEN, st));
    DISPATCH_NI(h->NI, hidden_finish_kernel, dim3(nb), blk, 0, st, h->hp, (const uint32_t*)h->d_iota, hg_rows, s0, nb, h->d_HGpart, h->d_Dz,
                h->d_HG, h->d_Wu, h->d_Wu_ag, hg_parts, h->d_Uu, h->d_Uu_ag, h->d_Ssum, h->d_delta_rows);
    CHK(pa.end());
    GemmEpilogue e3{};
    e3.Cout = h->d_dD; e3.ldc = Kp;
    // the b recurrence (41 ns per user, strictly in user order) in two parts: the first half of the block's users as leading workgroups
    // of the GEMM 3 launch, the rest as leading workgroups of the row launch — each launch then lasts about as long as its own work
    uint32_t bias_u0 = 0;
    if (!h->gemm_direct && gemm_lds_is_128(h, Ip, Kp) && nb >= 64 && !h->full_bias_unsplit) {
      bias_u0 = nb / 2;
      e3.bias_blocks = (Kp + 255u) / 256u; e3.bias_nb = bias_u0; e3.bias_delta = h->d_HG; e3.bias_b = h->P(CDAE_P_B); e3.bias_b_ag = h->P(CDAE_P_B_AG);
      e3.bias_hp = h->hp;
    }
    if (h->gemm_direct)
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPI_STORE>), dim3((Kp + 127) / 128, Ip / 64, 1), dim3(128), 0, st, h->d_GTb, h->d_ZTb, Ip, Kp,
                         Bp, Bp, Bp, Bp, e3);
    else
      CHK(launch_gemm_lds<EPI_STORE>(h, st, h->d_GTb, h->d_ZTb, Ip, Kp, Bp, Bp, Bp, Bp, e3, 1, 1));
    CHK(pr.end());
    CHK(pr.begin(h, F_INPUT, st));
    const uint32_t bias_blocks = (Kp + 255u) / 256u;
    DISPATCH_NI(h->NI, full_rows_kernel, dim3(bias_blocks + I), blk, 0, st, h->hp, x.seg, x.seg + I, x.sorted_val, h->delta_rows(),
                h->d_dD, h->d_GTb, Bp, nb, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_V), h->P(CDAE_P_V_AG), h->P(CDAE_P_BP),
                h->P(CDAE_P_BP_AG), h->P(CDAE_P_B), h->P(CDAE_P_B_AG), h->d_touched,
                rows_write_images ? h->d_Db : (__bf16*)nullptr, rows_write_images ? h->d_DTb : (__bf16*)nullptr, Ip, bias_u0, (const float*)h->d_HG);
    h->db_valid = rows_write_images;
    h->db_rows_valid = false;
    CHK(pr.end());
    HIPCHK(hipEventRecord(x.released, st));
    HIPCHK(hipGetLastError());
    return 0;
  }
  // Second stream: delta_u, the Wu steps and then the strictly sequential hidden-bias recurrence (2048 users x 58 ns) need
  // hg only; they run beside GEMM 3, and the row steps join them.
  // (Round 4 measured GEMM 2 on this stream too, BESIDE the fused row launch of the K = 512 path, which needs G^T and Z^T only — two
  // bf16 images of the decoder, swapped per block: 5.0 ms per 1024-user block against 4.75 in order.  The two launches stretch each
  // other — GEMM 2 + hidden layer 1.2 -> 3.7 ms, row launch 2.3 -> 3.8 — because GEMM 2's LDS fill and the row launch's streams share
  // the L2s and the fabric, and a CU holds one or the other, not both (139 KiB / 2 x 72 KiB of LDS).  Removed.)
  HIPCHK(hipEventRecord(h->ev_fork, st));
  HIPCHK(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
  {
    Prof pa;
    CHK(pa.begin(h, F_HIDDEN, h->aux));
    DISPATCH_NI(h->NI, hidden_finish_kernel, dim3(nb), blk, 0, h->aux, h->hp, hg_parts ? (const uint32_t*)h->d_iota : uptr,
                hg_parts ? hg_rows : n_units, s0, nb, h->d_HGpart, h->d_Dz, h->d_HG, h->d_Wu, h->d_Wu_ag, hg_parts,
                h->d_Uu, h->d_Uu_ag, h->d_Ssum, h->d_delta_rows);
    CHK(pa.end());
  }
  HIPCHK(hipEventRecord(h->ev_delta, h->aux));
  hipLaunchKernelGGL(hidden_bias_kernel, dim3((Kp + 255u) / 256u), blk, 0, h->aux, h->hp, nb, h->d_HG, h->P(CDAE_P_B), h->P(CDAE_P_B_AG));
  HIPCHK(hipEventRecord(h->ev_join, h->aux));
  // GEMM 3: dD = G^T Z  (contraction over the batch's users)
  {
    GemmEpilogue e3{};
    e3.Cout = h->d_dD; e3.ldc = Kp;
    // 64-row workgroups (two wavefronts): 2 x Ip/64 of them spread over all CUs, 128-row ones would occupy only 166 at ML-10M shape
    // (developer switch CDAE_GEMM_DIRECT; the default is the LDS-staged kernel: 73 -> 31 us at ML-10M shape, 2048 users)
    if (rows_fused)
      ;                                                          // (the product stays in the accumulators of the row-step launch below)
    else if (h->gemm_direct)
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPI_STORE>), dim3((Kp + 127) / 128, Ip / 64, 1), dim3(128), 0, st, h->d_GTb, h->d_ZTb, Ip, Kp,
                         Bp, Bp, Bp, Bp, e3);
    else
      CHK(launch_gemm_lds<EPI_STORE>(h, st, h->d_GTb, h->d_ZTb, Ip, Kp, Bp, Bp, Bp, Bp, e3, 1, 1));
  }
  CHK(pr.end());

  HIPCHK(hipStreamWaitEvent(st, h->ev_delta, 0));
  CHK(pr.begin(h, F_INPUT, st));
  if (rows_fused)      // dD = G^T Z and the row steps from its accumulators (gemm3_rows_fused_kernel)
    CHK(launch_rows_fused(h, st, x, nb, h->d_Db));
  else if (I >= 32768u)     // rows are plentiful and mostly without kept inputs: one wavefront per row
    DISPATCH_NI(h->NI, full_rows_wave_kernel, dim3((I + 3) / 4), blk, 0, st, h->hp, x.seg, x.seg + I, x.sorted_val, h->delta_rows(),
                h->d_dD, h->d_GTb, Bp, nb, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_V), h->P(CDAE_P_V_AG), h->P(CDAE_P_BP),
                h->P(CDAE_P_BP_AG), h->d_touched, rows_write_db ? h->d_Db : (__bf16*)nullptr);
  else
    DISPATCH_NI(h->NI, full_rows_kernel, dim3(I), blk, 0, st, h->hp, x.seg, x.seg + I, x.sorted_val, h->delta_rows(),
                h->d_dD, h->d_GTb, Bp, nb, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_V), h->P(CDAE_P_V_AG), h->P(CDAE_P_BP),
                h->P(CDAE_P_BP_AG), (float*)nullptr, (float*)nullptr, h->d_touched,
                rows_write_images ? h->d_Db : (__bf16*)nullptr, rows_write_images ? h->d_DTb : (__bf16*)nullptr, Ip);
  h->db_valid = rows_write_images;                             // every decoder row was stepped and imaged by this launch
  h->db_rows_valid = rows_write_db;
  h->join_pending = true;                                      // the aux stream (b recurrence) is joined by its next consumer: join_aux
  CHK(pr.end());
  HIPCHK(hipEventRecord(x.released, st));
  HIPCHK(hipGetLastError());
  return 0;
}

// IMF / BPR: one block of users (cdae_mf_kernels.hpp).  A block of one user is the reference loop itself (in place).
int compute_batch_mf(cdae_hip* h, int b, const Batch& bt) {
  using namespace cdae;
  cdae_hip::ExBuf& x = h->ex[b];
  hipStream_t st = h->stream;
  const uint32_t I = (uint32_t)h->I, nb = bt.nb;
  // in place = the reference loop: a block of one user, or (mf_seq) a launch window of users that ONE wavefront walks in order
  const bool in_place = nb == 1 || h->mf_seq;
  const dim3 blk(in_place ? 64 : 256), grid_users(in_place ? 1 : (nb + 3) / 4);
  Prof pr;
  HIPCHK(hipStreamWaitEvent(st, x.ready, 0));
  CHK(pr.begin(h, F_DECODE, st));
#define MF_USER(NI_, PAIR_, INP_)                                                                                                   \
  hipLaunchKernelGGL((mf_user_kernel<NI_, PAIR_, INP_>), grid_users, blk, 0, st, h->hp, h->mf_bias, h->d_row_ptr, bt.s0, nb, x.item, \
                     h->d_Wu, h->d_Wu_ag, h->d_ub, h->d_ub_ag, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_BP), h->P(CDAE_P_BP_AG), \
                     h->d_UVpre, h->d_G)
#define MF_USER_NI(NI_)                                                              \
  do {                                                                               \
    if (h->mf == 2) { if (in_place) MF_USER(NI_, true, true); else MF_USER(NI_, true, false); }     \
    else { if (in_place) MF_USER(NI_, false, true); else MF_USER(NI_, false, false); }              \
  } while (0)
  switch (h->NI) { case 1: MF_USER_NI(1); break; case 2: MF_USER_NI(2); break; case 4: MF_USER_NI(4); break; default: MF_USER_NI(8); break; }
#undef MF_USER_NI
#undef MF_USER
  CHK(pr.end());
  if (!in_place) {
    CHK(pr.begin(h, F_INPUT, st));
    DISPATCH_NI(h->NI, mf_item_kernel, dim3((I + 3) / 4), blk, 0, st, h->hp, h->mf_bias, h->d_item_order, x.seg, x.seg + I, x.sorted_val,
                h->d_UVpre, h->d_G, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_BP), h->P(CDAE_P_BP_AG));
    CHK(pr.end());
  }
  HIPCHK(hipEventRecord(x.released, st));
  HIPCHK(hipGetLastError());
  return 0;
}

// z for nb users: a contiguous range [u0, u0+nb) (d_uids == nullptr) or the list d_uids (prefix in d_uptr_tmp)
int encode_chunk(cdae_hip* h, const uint32_t* d_uids, uint64_t u0, uint32_t nb, int mode, uint32_t stream_id,
                 uint32_t cidx, uint64_t seed, uint32_t epoch, uint32_t n_units_list = 0, float* z_out = nullptr,
                 float* hpart = nullptr, uint32_t hpart_cap = 0) {
  if (h->item_shard) return fail("an item shard encodes in phases under the multi-shard handle (its rows and its users' private rows are partial)");
  const uint32_t n_units = d_uids ? n_units_list : h->h_unit_ptr[u0 + nb] - h->h_unit_ptr[u0];
  const uint32_t* uptr = d_uids ? h->d_uptr_tmp : h->d_unit_ptr + u0;
  if (!hpart) { hpart = h->d_Hpart; hpart_cap = h->unit_cap; }
  if (n_units > hpart_cap) return fail("%u work units exceed the capacity %u", n_units, hpart_cap);
  DISPATCH_NI(h->NI, cdae::encode_partial_kernel, dim3((n_units + 3) / 4), dim3(256), 0, h->stream, h->hp, h->d_row_ptr, h->d_col,
              h->P(CDAE_P_W), uptr, n_units, d_uids, u0, nb, mode, stream_id, cidx, seed, epoch, hpart,
              (const uint32_t*)nullptr, 0u, d_uids ? (const uint32_t*)nullptr : (const uint32_t*)h->d_unit_user);
  DISPATCH_NI(h->NI, cdae::encode_finish_kernel, dim3((nb + 3) / 4), dim3(256), 0, h->stream, h->hp, hpart, uptr, h->d_Wu,
              h->P(CDAE_P_B), d_uids, u0, nb, mode, z_out ? z_out : h->d_Z, (float*)nullptr, (float*)nullptr, h->d_Uu, (float*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

// Evaluation workspace (data_loss, recommend): z rows and encode partial sums for up to EVAL_CHUNK users per launch — the
// training workspace holds only batch_users of them, far too few wavefronts to fill the chip.
constexpr uint32_t EVAL_CHUNK = 32768;
// item shard: blocks of [users x Kp] floats in the input-sum all-reduce buffer: the input sums, then — the user node being sharded by
// user — the batch's Wu rows (user_factor) and Uu rows (linear_function) contributed by their owner
constexpr uint32_t SHARD_BLOCKS = 3;
inline uint32_t shard_blocks_of(const cdae_hip* h) { return 1u + (h->cfg.user_factor ? 1u : 0u) + (h->cfg.linear_function ? 1u : 0u); }
int ensure_eval_ws(cdae_hip* h, uint32_t users, uint32_t units) {
  if (h->eval_cap < users) {
    if (h->d_zeval) HIPCHK(hipFree(h->d_zeval));
    h->d_zeval = nullptr; h->eval_cap = 0;
    CHK(dev_alloc(&h->d_zeval, (size_t)users * h->Kp));
    h->eval_cap = users;
  }
  if (h->eval_unit_cap < units) {
    if (h->d_hpart_eval) HIPCHK(hipFree(h->d_hpart_eval));
    h->d_hpart_eval = nullptr; h->eval_unit_cap = 0;
    CHK(dev_alloc(&h->d_hpart_eval, (size_t)units * h->Kp));
    h->eval_unit_cap = units;
  }
  return 0;
}

inline bool is_user_indexed(uint32_t which) {
  return which == CDAE_P_WU || which == CDAE_P_WU_AG || which == CDAE_P_UU || which == CDAE_P_UU_AG || which == CDAE_P_UB || which == CDAE_P_UB_AG;
}
int copy_param_out(cdae_hip* h, uint32_t which, float* host, size_t count) {
  float* d = h->P(which);
  if (!d) return count == 0 ? 0 : fail("parameter %u is not allocated in this configuration", which);
  const bool vec = (which == CDAE_P_BP || which == CDAE_P_BP_AG || which == CDAE_P_UB || which == CDAE_P_UB_AG);
  const size_t rows = (which == CDAE_P_WU || which == CDAE_P_WU_AG || which == CDAE_P_UU || which == CDAE_P_UU_AG) ? (size_t)h->wu_rows() : ((which == CDAE_P_B || which == CDAE_P_B_AG) ? 1 : h->I);
  if (vec) {
    const size_t want = (which == CDAE_P_UB || which == CDAE_P_UB_AG) ? h->U : h->I;
    if (count != want) return fail("parameter %u has %llu elements, got %zu", which, (unsigned long long)want, count);
    HIPCHK(hipMemcpyAsync(host, d, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  } else {
    if (count != rows * h->K) return fail("parameter %u has %zu elements, got %zu", which, rows * h->K, count);
    if (rows == 0) return 0;
    HIPCHK(hipMemcpy2DAsync(host, h->K * sizeof(float), d, h->Kp * sizeof(float), h->K * sizeof(float), rows,
                            hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

}  // namespace

// The library keeps three HIP streams busy (main, prep, aux) next to the host application's own (e.g. RCCL's): more than
// HIP's default of four hardware queues, beyond which streams share a queue and serialise.  Runs at load time, i.e. before
// this process's first HIP call when the library is loaded first; a no-op if the variable is already set.
__attribute__((constructor)) static void cdae_hip_raise_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" {

const char* cdae_hip_last_error(void) { return g_err.c_str(); }
int cdae_hip_abi_version(void) { return CDAE_HIP_ABI_VERSION; }

int cdae_hip_create(const cdae_hip_config* cfg, int device_id, cdae_hip_t** out) {
  if (!cfg || !out) return fail("null argument");
  if (cfg->struct_size != sizeof(cdae_hip_config)) return fail("cdae_hip_config size mismatch: got %u, want %zu", cfg->struct_size, sizeof(cdae_hip_config));
  if (cfg->num_dim == 0 || cfg->num_dim > 512) return fail("num_dim must be in [1, 512], got %u", cfg->num_dim);
  if (cfg->loss_type != CDAE_LOSS_SQUARE && cfg->loss_type != CDAE_LOSS_CROSS_ENTROPY)
    return fail("loss_type %u unsupported: CDAE's linear output only works with SQUARE (0) and CROSS_ENTROPY (5); "
                "LOGISTIC aborts in the reference (loss.hpp:96)", cfg->loss_type);
  if (cfg->num_corruptions == 0) return fail("num_corruptions must be >= 1");
  if (cfg->scaled && !(cfg->corruption_ratio < 1.0)) return fail("scaled input needs corruption_ratio < 1");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail("device %d not available (%d HIP devices)", device_id, ndev);
  HIPCHK(hipSetDevice(device_id));
  if (cfg->full_output > 1u) return fail("bad full_output");
  if (cfg->batch_users > cdae::SLOT_MASK) return fail("batch_users %u exceeds the example word's slot field (2^28 - 1)", cfg->batch_users);
  cdae_hip* h = new cdae_hip();
  h->cfg = *cfg;
  h->device = device_id;
  h->K = cfg->num_dim;
  h->NI = cfg->num_dim <= 64 ? 1 : (cfg->num_dim <= 128 ? 2 : (cfg->num_dim <= 256 ? 4 : 8));
  h->Kp = 64u * h->NI;
  h->B = cfg->batch_users ? cfg->batch_users : 1024u;
  h->one_row_per_wave = DEV_ENV("CDAE_DECODE_ONE_ROW_PER_WAVE") != nullptr;
  h->full_unfused = DEV_ENV("CDAE_FULL_UNFUSED") != nullptr;
  h->gemm_direct = DEV_ENV("CDAE_GEMM_DIRECT") != nullptr;
  h->gemm_two_stage = DEV_ENV("CDAE_GEMM_TWO_STAGE") != nullptr;
  h->gemm_narrow = DEV_ENV("CDAE_GEMM_NARROW") != nullptr;
  h->rows_separate = DEV_ENV("CDAE_FULL_ROWS_SEPARATE") != nullptr;
  h->gemm2_nt = DEV_ENV("CDAE_GEMM2_NT") != nullptr;
  if (const char* ev = DEV_ENV("CDAE_GEMM2_STAGES")) h->gemm2_stages = std::max(2, std::min(4, std::atoi(ev)));
  h->gemm1_tiled = DEV_ENV("CDAE_GEMM1_TILED") != nullptr;
  if (const char* e = DEV_ENV("CDAE_FULL_ROWS_KH")) h->rows_fused_kh = std::atoi(e) == 1 ? 1 : 2;
  h->recommend_per_user = DEV_ENV("CDAE_RECOMMEND_PER_USER") != nullptr;
  h->full_bias_unsplit = DEV_ENV("CDAE_FULL_BIAS_UNSPLIT") != nullptr;
  if (const char* v = DEV_ENV("CDAE_FULL_ONE_STREAM_MAX")) h->full_one_stream_max = (uint32_t)std::strtoul(v, nullptr, 10);
  h->gemm1_zreg = DEV_ENV("CDAE_GEMM1_ZREG") != nullptr;
  h->debug_skip_prep = DEV_ENV("CDAE_DEBUG_SKIP_PREP") != nullptr;
  h->encode_two_launches = DEV_ENV("CDAE_ENCODE_TWO_LAUNCHES") != nullptr;
  h->full_separate_copies = DEV_ENV("CDAE_FULL_SEPARATE_COPIES") != nullptr;
  h->fused_images = !h->full_separate_copies;
  if (const char* ev = DEV_ENV("CDAE_ENCODE_USERS_MAX")) h->encode_users_max = (uint32_t)std::atoi(ev);
  if (const char* ev = DEV_ENV("CDAE_GATHER_HALVES")) h->gather_halves = std::atoi(ev) == 2 ? 2u : 1u;
  if (const char* ev = DEV_ENV("CDAE_PREP_THREAD")) h->prep_threaded = std::atoi(ev) != 0;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return fail("hipStreamCreate failed: %s", hipGetErrorString(e)); }
  if (const char* ev = DEV_ENV("CDAE_STREAM_PAD")) {     // developer experiment: shift which hardware queues the library's other streams get
    static std::vector<hipStream_t> pads;                 // (kept for the life of the process)
    static void* padbuf = nullptr;
    if (!padbuf) (void)hipMalloc(&padbuf, 256);
    for (int i = 0, n = std::atoi(ev); i < n && padbuf; ++i) {
      hipStream_t s;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
      (void)hipMemsetAsync(padbuf, 0, 256, s);            // (a stream gets its hardware queue when it is first used)
      (void)hipStreamSynchronize(s);
      pads.push_back(s);
    }
  }
  {
    // the prep stream's queue priority (CDAE_PREP_PRIORITY = 1: the device's highest; developer switch)
    int prio_lo = 0, prio_hi = 0;
    const bool high = DEV_ENV("CDAE_PREP_PRIORITY") != nullptr && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess;
    if (e == hipSuccess) e = high ? hipStreamCreateWithPriority(&h->prep, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&h->prep, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking);
  {
    // Second prep lane.  Sampling + sorting a batch is a chain of ~12 small launches, ~95 us on the prep stream whatever the
    // batch size; once a training step is shorter than that the decode waits for `ready` (profiles/r02_wave_timeline_256.txt: 15 us
    // per step at 256 users).  Two lanes prepare consecutive batches side by side.  Measured (tools/ab_bench.py, same box):
    // 128 users per batch 0.0972 -> 0.0755 ms per step, 256: 0.0995 -> 0.0942, 512: 0.1409 -> 0.1429 (the prep kernels then
    // only take issue slots from a step that was not waiting for them).  Default "auto": on up to 384 users per batch, on the
    // handle's aux stream (idle in the sampled path; an exchange's collective shares it).  CDAE_PREP2 = off | aux | own | auto.
    const char* sel = DEV_ENV("CDAE_PREP2");
    if (sel && !std::strcmp(sel, "own")) { if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->prep2, hipStreamNonBlocking); h->prep2_own = true; }
    else if (sel && !std::strcmp(sel, "aux")) h->prep2 = h->aux;
    else if (sel && !std::strcmp(sel, "off")) h->prep2 = nullptr;
    else { h->prep2 = h->aux; h->prep2_auto = true; }
    if (const char* hp_ = DEV_ENV("CDAE_HOST_PACE_US")) h->host_pace_us = (uint32_t)std::max(0, std::atoi(hp_));
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, sync_event_flags());
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, sync_event_flags());
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_delta, sync_event_flags());
  if (e != hipSuccess) { free_all(h); delete h; return fail("stream/event setup failed: %s", hipGetErrorString(e)); }
  e = hipMalloc((void**)&h->d_scalar, 8 * sizeof(double));
  if (e != hipSuccess) { free_all(h); delete h; return fail("hipMalloc failed: %s", hipGetErrorString(e)); }
  cdae::HyperParams& hp = h->hp;
  hp.lambda = (float)cfg->lambda; hp.lr = (float)cfg->learn_rate; hp.beta = (float)cfg->beta;
  hp.scale = cfg->scaled ? (float)(1.0 / (1.0 - cfg->corruption_ratio)) : 1.f;     // cdae.hpp:202-205
  hp.num_neg = cfg->full_output ? 0u : cfg->num_neg;     // full output: the example list holds the positives only
  hp.loss_type = cfg->loss_type;
  hp.adagrad = cfg->using_adagrad; hp.asymmetric = cfg->asymmetric; hp.user_factor = cfg->user_factor;
  hp.linear = cfg->linear; hp.tanh_act = cfg->tanh_act; hp.linear_function = cfg->linear_function;
  hp.keep_thr = cdae_keep_threshold(cfg->corruption_ratio);
  hp.uid_offset = 0; hp.num_items = 0; hp.K = h->K; hp.Kp = h->Kp;
  hp.own_u0 = 0; hp.own_u1 = ~0ull;
  if (DEV_ENV("CDAE_WAVE_TRACE")) {      // developer aid (tools/wave_trace.py): the last training batch's wavefront timeline
    if (hipMalloc((void**)&hp.trace, 4 * cdae::TRACE_CAP * sizeof(unsigned long long)) != hipSuccess) hp.trace = nullptr;
    if (hp.trace) (void)hipMemset(hp.trace, 0, 4 * cdae::TRACE_CAP * sizeof(unsigned long long));
  }
  hp.debug_skip = DEV_ENV("CDAE_DEBUG_SKIP_ROLES") ? (uint32_t)std::strtoul(DEV_ENV("CDAE_DEBUG_SKIP_ROLES"), nullptr, 10) : 0u;
  hp.debug_rank = DEV_ENV("CDAE_DEBUG_RANK") ? (uint32_t)std::strtoul(DEV_ENV("CDAE_DEBUG_RANK"), nullptr, 10) : 0u;
  *out = h;
  return 0;
}

int cdae_hip_destroy(cdae_hip_t* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  stop_prep_worker(h);
  (void)hipStreamSynchronize(h->stream);
  if (h->hp.trace) {      // records of the LAST batch every slot saw
    std::vector<unsigned long long> rec(4 * cdae::TRACE_CAP);
    (void)hipMemcpy(rec.data(), h->hp.trace, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen(DEV_ENV("CDAE_WAVE_TRACE"), "wb")) { std::fwrite(rec.data(), sizeof(unsigned long long), rec.size(), f); std::fclose(f); }
    (void)hipFree(h->hp.trace);
    h->hp.trace = nullptr;
  }
  if (h->xchg && h->xchg_free) { h->xchg_free(h->xchg); h->xchg = nullptr; }
  free_all(h);
  delete h;
  return 0;
}

int cdae_hip_create_mf(const cdae_mf_config* mc, int device_id, cdae_hip_t** out) {
  if (!mc || !out) return fail("null argument");
  if (mc->struct_size != sizeof(cdae_mf_config)) return fail("cdae_mf_config size mismatch: got %u, want %zu", mc->struct_size, sizeof(cdae_mf_config));
  if (mc->loss_type > 3u && mc->loss_type != CDAE_LOSS_CROSS_ENTROPY)
    return fail("loss_type %u unsupported: SQUARE (0), LOGISTIC (1), LOG (2), HINGE (3), CROSS_ENTROPY (5)", mc->loss_type);
  if (mc->num_neg == 0) return fail("num_neg must be >= 1");
  cdae_hip_config c = cdae_hip_config();
  c.struct_size = sizeof(c);
  c.num_dim = mc->num_dim; c.num_neg = mc->num_neg; c.num_corruptions = 1;
  c.loss_type = CDAE_LOSS_SQUARE;                          // (validated above; the MF kernels read hp.loss_type, set below)
  c.using_adagrad = mc->using_adagrad; c.user_factor = 1;
  // default (cdae_hip_mf_default_batch_users): the largest block measured inside the +-0.002 mean-over-seeds Recall@10 bound against the
  // sequential loop (imf.hpp:71-115, bpr.hpp:56-106; DESIGN.md §8b, tests/test_gpu_mf.py) at the BASELINE shapes, one user per block
  // — the reference loop itself — on data sets smaller than the ones it was measured on.  Larger blocks are a throughput setting
  // (batch_users = 0: decided in cdae_hip_set_interactions, where the number of users is known — cdae_hip_mf_default_batch_users)
  c.batch_users = mc->batch_users ? mc->batch_users : 1u;
  c.lambda = mc->lambda; c.learn_rate = mc->learn_rate; c.corruption_ratio = 0.; c.beta = mc->beta;
  CHK(cdae_hip_create(&c, device_id, out));
  cdae_hip* h = *out;
  h->mf = mc->pairwise ? 2u : 1u;
  h->mf_bias = mc->using_bias_term ? 1u : 0u;
  h->mf_auto = mc->batch_users == 0u;
  if (c.batch_users == 1u && DEV_ENV("CDAE_MF_ONE_LAUNCH_PER_USER") == nullptr) { h->mf_seq = true; h->B = MF_SEQ_USERS; }   // (the switch: round 3's launches, A/B)
  h->hp.loss_type = mc->loss_type;
  h->hp.lambda = (float)(2.0 * mc->lambda);                // imf.hpp:92-95, bpr.hpp:78-82: the gradients regularise with 2 * lambda
  return 0;
}

uint32_t cdae_hip_row_stride(const cdae_hip_t* h) { return h ? h->Kp : 0; }
uint32_t cdae_hip_mf_default_batch_users(uint64_t U, uint32_t pairwise) {
  if (pairwise) return U >= CDAE_BPR_DEFAULT_MIN_USERS ? CDAE_BPR_DEFAULT_BATCH_USERS : 1u;
  return U >= CDAE_IMF_DEFAULT_MIN_USERS ? CDAE_IMF_DEFAULT_BATCH_USERS : 1u;
}
uint32_t cdae_hip_default_batch_users(uint64_t U) {
  return (uint32_t)std::min<uint64_t>(CDAE_DEFAULT_BATCH_USERS_MAX, std::max<uint64_t>(32, (U / 160) & ~(uint64_t)31));
}
int cdae_hip_user_order(cdae_hip_t* h, uint32_t* out, size_t count) {
  if (!h || !h->d_shared || !out) return fail("set_interactions must be called first");
  if (count != h->U) return fail("user order has %llu entries, got %zu", (unsigned long long)h->U, count);
  for (uint64_t pos = 0; pos < h->U; ++pos) out[pos] = h->user_perm.empty() ? (uint32_t)pos : h->user_perm[pos];
  return 0;
}
int cdae_hip_set_decode_fused(cdae_hip_t* h, int allow) {
  if (!h) return fail("null handle");
  h->allow_fused = allow != 0;
  h->fused_decode = h->fused_possible && h->allow_fused;
  return 0;
}

int cdae_hip_decode_plan(const cdae_hip_t* h, uint32_t* hot_rows, uint32_t* late_rows, uint32_t* fused) {
  if (!h) return fail("null handle");
  const bool set = h->d_shared != nullptr;
  if (hot_rows) *hot_rows = set && h->K <= 256 && !h->one_row_per_wave && !h->mf && !h->cfg.full_output ? h->hot_rows : 0u;
  if (late_rows) *late_rows = set ? h->late_rows : 0u;
  if (fused) *fused = set && h->fused_decode ? 1u : 0u;
  return 0;
}

uint32_t cdae_hip_batch_users(const cdae_hip_t* h) { return !h || (h->cfg.batch_users == 0 && h->U == 0) ? 0 : (h->mf_seq ? 1u : h->B); }
uint32_t cdae_hip_full_output_plan(const cdae_hip_t* h) {
  if (!h || !h->cfg.full_output || h->U == 0) return 0;
  if (h->Kp <= 256 && !h->full_unfused) return CDAE_PLAN_FUSED_DECODE;
  return (gemm2_tn_path(h) ? CDAE_PLAN_GEMM2_TN : 0u) | (rows_fused_path(h) ? CDAE_PLAN_ROWS_FUSED : 0u);
}

int cdae_hip_set_user_id_offset(cdae_hip_t* h, uint64_t offset) {
  if (!h) return fail("null handle");
  h->uid_offset = offset; h->hp.uid_offset = offset;
  return 0;
}

int cdae_hip_set_interactions(cdae_hip_t* h, uint64_t U, uint64_t I, const int64_t* row_ptr, const uint32_t* col) {
  if (!h || !row_ptr || (!col && U && row_ptr[U])) return fail("null argument");
  if (h->shard_sampled() && (!h->g_row_ptr_src || !h->g_col_src)) return fail("an item shard of the sampled decode needs the whole rows (set_item_shard_global)");
  if (h->shard_sampled() && h->mf) return fail("the item-sharded layout is CDAE's");
  if (U == 0 || I == 0) return fail("empty interaction matrix (%llu users, %llu items)", (unsigned long long)U, (unsigned long long)I);
  if (I >= (1ull << 30) || U >= (1ull << 30)) return fail("at most 2^30 users and items");
  if (row_ptr[0] != 0) return fail("row_ptr[0] must be 0");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  std::vector<int64_t> perm_ptr;
  std::vector<uint32_t> perm_col;
  h->user_perm.clear(); h->user_inv.clear();
  if (h->mf && h->mf_auto) {
    const uint32_t b = cdae_hip_mf_default_batch_users(U, h->mf == 2u ? 1u : 0u);
    h->cfg.batch_users = b;
    h->mf_seq = b == 1u && DEV_ENV("CDAE_MF_ONE_LAUNCH_PER_USER") == nullptr;
    h->B = h->mf_seq ? MF_SEQ_USERS : b;
  }
  if (h->mf && !h->mf_seq && h->B > 1 && U > h->B) {
    for (uint64_t u = 0; u < U; ++u)
      if (row_ptr[u + 1] < row_ptr[u]) return fail("row_ptr is not monotone at user %llu", (unsigned long long)u);
    // activity-grouped training order (see cdae_hip::user_perm)
    std::vector<uint32_t> by_len(U);
    std::iota(by_len.begin(), by_len.end(), 0u);
    std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b) { return row_ptr[a + 1] - row_ptr[a] > row_ptr[b + 1] - row_ptr[b]; });
    const uint64_t nblk = (U + h->B - 1) / h->B;
    std::vector<uint32_t> blk(nblk);
    std::iota(blk.begin(), blk.end(), 0u);
    // the short last group (U % B users, the least active) stays LAST: make_plan cuts batches at multiples of B from position 0, so
    // a short group in the middle made every later batch straddle two unrelated activity groups and last as long as the heavier one
    const uint64_t shuffled = (U % h->B) ? nblk - 1 : nblk;
    std::stable_sort(blk.begin(), blk.begin() + shuffled, [](uint32_t a, uint32_t b) { return (uint32_t)(a * 2654435761u) < (uint32_t)(b * 2654435761u); });
    h->user_perm.reserve(U);
    for (uint32_t k : blk)
      for (uint64_t t = (uint64_t)k * h->B; t < std::min<uint64_t>(U, (uint64_t)(k + 1) * h->B); ++t) h->user_perm.push_back(by_len[t]);
    h->user_inv.resize(U);
    for (uint64_t pos = 0; pos < U; ++pos) h->user_inv[h->user_perm[pos]] = (uint32_t)pos;
    perm_ptr.assign(U + 1, 0);
    perm_col.resize((size_t)row_ptr[U]);
    for (uint64_t pos = 0; pos < U; ++pos) {
      const uint32_t u = h->user_perm[pos];
      std::copy(col + row_ptr[u], col + row_ptr[u + 1], perm_col.begin() + perm_ptr[pos]);
      perm_ptr[pos + 1] = perm_ptr[pos] + (row_ptr[u + 1] - row_ptr[u]);
    }
    row_ptr = perm_ptr.data(); col = perm_col.data();       // from here on: rows by POSITION
  }
  std::vector<uint64_t> pop(I, 0);
  for (uint64_t u = 0; u < U; ++u) {
    const int64_t a = row_ptr[u], b = row_ptr[u + 1];
    if (b < a) return fail("row_ptr is not monotone at user %llu", (unsigned long long)u);
    if (!h->item_shard) {      // an item shard sees only its slice of every row: empty and full slices are legal there
      if (b <= a) return fail("user %llu has no training item (the reference CHECK-fails too, cdae.hpp:139)", (unsigned long long)u);
      if ((uint64_t)(b - a) >= I) return fail("user %llu rated every item: no negative can be sampled", (unsigned long long)u);
    }
    for (int64_t p = a; p < b; ++p) {
      if (col[p] >= I) return fail("item id %u out of range at position %lld", col[p], (long long)p);
      if (p > a && col[p] <= col[p - 1]) return fail("row %llu is not strictly ascending at position %lld", (unsigned long long)u, (long long)p);
      pop[col[p]]++;
    }
  }
  // nothing may still be reading the old data set: a prefetched batch in the prep worker's queue or in the prep streams included
  CHK(quiesce(h));
  CHK(free_interaction_state(h));
  if (h->cfg.batch_users == 0) {
    // default: ~U/160 users per parameter snapshot, in [32, 256].  256 is the largest size whose Recall@10 shows no systematic
    // offset against the strictly sequential reference schedule (paired multi-seed study at ML-10M shape, DESIGN.md §2: 384 and
    // 512 sit 0.001-0.002 low); tests/test_gpu_accuracy.py certifies exactly this value at the BASELINE shapes.
    h->B = cdae_hip_default_batch_users(U);
  }
  h->U = U; h->I = I; h->hp.num_items = (uint32_t)I;
  if (h->item_shard) {
    if (h->own_u1 == ~0ull) { h->own_u0 = 0; h->own_u1 = U; }
    if (h->own_u0 > h->own_u1 || h->own_u1 > U) return fail("owned user range [%llu, %llu) outside the %llu users", (unsigned long long)h->own_u0, (unsigned long long)h->own_u1, (unsigned long long)U);
    h->hp.own_u0 = h->own_u0; h->hp.own_u1 = h->own_u1;
  }
  {
    // Work-unit size of the user-parallel kernels (sample, encode, hidden gather, data_loss: one wavefront per unit).  Those
    // launches are latency-bound per wavefront — a unit's rows are gathered a few at a time — so a batch should offer the chip
    // (256 CUs x 4 SIMDs) several thousand wavefronts: small batches take small units.
    const uint64_t Bu = std::min<uint64_t>(h->B, U);
    uint32_t up = Bu <= 1024 ? 64u : cdae::UNIT_POS_MAX;          // measured at ML-10M shape, batch_users 256 / 512: 64 best (profiles/r02_unit_size.txt)
    if (const char* ev = DEV_ENV("CDAE_UNIT_POS")) up = (uint32_t)std::atoi(ev);
    h->hp.unit_pos = std::max<uint32_t>(1u, std::min<uint32_t>(up, cdae::UNIT_POS_MAX));
  }
  h->h_row_ptr.assign(row_ptr, row_ptr + U + 1);
  const size_t nnz = (size_t)row_ptr[U];
  CHK(dev_alloc(&h->d_row_ptr, U + 1));
  CHK(dev_alloc(&h->d_col, nnz));
  HIPCHK(hipMemcpy(h->d_row_ptr, row_ptr, (U + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_col, col, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
  // decode launch order: most popular rows first (their example chains are the longest)
  std::vector<uint32_t> order(I);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pop[a] > pop[b]; });
  CHK(dev_alloc(&h->d_item_order, (size_t)I));
  HIPCHK(hipMemcpy(h->d_item_order, order.data(), I * sizeof(uint32_t), hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> rank_of(I);
    for (uint32_t r = 0; r < I; ++r) rank_of[order[r]] = r;
    CHK(dev_alloc(&h->d_rank_of, (size_t)I));
    HIPCHK(hipMemcpy(h->d_rank_of, rank_of.data(), I * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  {
    // hot rows: expected positives per batch (popularity x batch share) of at least CDAE_DECODE_HOT_POS (default 48);
    // every row also receives ~ B * mean(n_u) * num_neg / I uniformly spread negatives
    const char* ev = DEV_ENV("CDAE_DECODE_HOT_POS");
    const double hot_pos = ev ? std::atof(ev) : 48.0;
    const double share = (double)std::min<uint64_t>(h->B, U) / (double)U;
    uint32_t hot = 0;
    while (hot < I && (double)pop[order[hot]] * share >= hot_pos) ++hot;
    h->hot_rows = std::min<uint32_t>((hot + 3u) & ~3u, (uint32_t)I);
  }
  {
    // Late rows: the most popular of the hot rows (at most one lane of hidden_finish_kernel each).  Their terms of the hidden gradient are
    // added by hidden_finish_kernel / hg_raw_kernel from Ghot, not gathered — in EVERY launch order (so that the fused launch, the
    // separate launches and an item shard of one agree bit for bit); the gather tells them by a bitmap it stages in LDS (item spaces up
    // to 65 536).  CDAE_NO_LATE_ROWS: round 5's arithmetic (developer switch).
    const bool hybrid = h->K <= 256 && !h->one_row_per_wave && !h->mf && !h->cfg.full_output;
    h->late_rows = 0;
    if (hybrid && I <= 32u * cdae::LATE_BITS_WORDS && !DEV_ENV("CDAE_NO_LATE_ROWS")) {
      // late: at least CDAE_DECODE_LATE_POS (default 48, i.e. the hot rows: measured 16 / 24 / 32 / 48 -> 0.0915 / 0.0918 / 0.0907 / 0.0898 ms per step) expected positives per batch, at most LATE_MAX rows; all of them take a
      // wavefront of their own (a late row in the four-rows-per-wavefront format would need its own Ghot bookkeeping there)
      const char* ev = DEV_ENV("CDAE_DECODE_LATE_POS");
      const double late_pos = ev ? std::atof(ev) : 48.0;
      const double share = (double)std::min<uint64_t>(h->B, U) / (double)U;
      uint32_t late = 0;
      while (late < I && late < cdae::LATE_MAX && (double)pop[order[late]] * share >= late_pos) ++late;
      // (a batch must have examples enough to make the split worth its bookkeeping: none for tiny batches)
      h->late_rows = h->hot_rows ? std::min<uint32_t>(std::min<uint32_t>((late + 3u) & ~3u, cdae::LATE_MAX), (uint32_t)I) : 0u;
      h->hot_rows = std::max(h->hot_rows, h->late_rows);
    }
    h->late_words = (uint32_t)((I + 31) / 32);
    {
      // expected examples per batch of every row, by popularity rank: positives (popularity x batch share) + the uniformly drawn negatives
      const double share = (double)std::min<uint64_t>(h->B, U) / (double)U;
      const double neg_per_item = share * (double)row_ptr[U] * (double)h->hp.num_neg / (double)std::max<uint64_t>(I, 1);
      h->h_rank_len.resize(I);
      for (uint32_t r = 0; r < I; ++r) h->h_rank_len[r] = (float)((double)pop[order[r]] * share + neg_per_item);
    }
    if (h->late_rows) {
      std::vector<uint32_t> bits(h->late_words, 0u);
      for (uint32_t r = 0; r < h->late_rows; ++r) bits[order[r] >> 5] |= 1u << (order[r] & 31u);
      CHK(dev_alloc(&h->d_late_bits, (size_t)h->late_words));
      HIPCHK(hipMemcpy(h->d_late_bits, bits.data(), h->late_words * sizeof(uint32_t), hipMemcpyHostToDevice));
      const size_t nB = (size_t)std::max<uint64_t>(std::min<uint64_t>(h->B, U), 1) * cdae::LATE_MAX;
      CHK(dev_alloc(&h->d_Ghot, nB)); CHK(dev_alloc(&h->d_hotdup, nB));
      HIPCHK(hipMemset(h->d_Ghot, 0, nB * sizeof(float)));
      HIPCHK(hipMemset(h->d_hotdup, 0xFF, nB * sizeof(uint32_t)));
    }
    if (!h->h_err) CHK(err_slot_acquire(h->device, &h->h_err, &h->d_fused_err, &h->err_slot));
    *(volatile uint32_t*)h->h_err = 0u;
    CHK(dev_alloc(&h->d_hot_cnt, (size_t)h->hot_rows / 4 + 1));
    HIPCHK(hipMemset(h->d_hot_cnt, 0, ((size_t)h->hot_rows / 4 + 1) * sizeof(uint32_t)));
    h->fused_seq = 0; h->fused_geo_set = false;
    // The fused launch: needs late rows (else the gather would wait for the longest chains); hot rows take a CU per four of them, so
    // at most half the chip's; the row matrices are addressed through 32-bit buffer offsets.
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    h->num_cus = (uint32_t)std::max(prop.multiProcessorCount, 16);
    h->fused_possible = h->late_rows && h->hot_rows <= 2u * h->num_cus && !h->item_shard && (uint64_t)I * h->Kp * 4u < (1ull << 31) &&
                        !DEV_ENV("CDAE_DECODE_UNFUSED");
#ifdef CDAE_DECODE_TIMING
    h->fused_possible = false;
#endif
    h->fused_decode = h->fused_possible && h->allow_fused;
  }

  // parameters
  const size_t IK = (size_t)I * h->Kp;
  size_t o = 0;
  std::memset(h->off, 0, sizeof h->off); std::memset(h->cnt, 0, sizeof h->cnt);
  auto place = [&](uint32_t id, size_t n) { h->off[id] = o; h->cnt[id] = n; o += n; };
  place(CDAE_P_W, IK); place(CDAE_P_W_AG, IK);
  if (h->cfg.asymmetric) { place(CDAE_P_V, IK); place(CDAE_P_V_AG, IK); }
  h->n_matrix = o;
  place(CDAE_P_BP, I); place(CDAE_P_BP_AG, I);
  place(CDAE_P_B, h->Kp); place(CDAE_P_B_AG, h->Kp);
  h->n_shared = o;
  CHK(dev_alloc(&h->d_shared, h->n_shared));
  HIPCHK(hipMemset(h->d_shared, 0, h->n_shared * sizeof(float)));
  const size_t WR = (size_t)h->wu_rows();                   // private rows held here: every user, or an item shard's own user range
  h->cnt[CDAE_P_WU] = h->cnt[CDAE_P_WU_AG] = WR * h->Kp;
  CHK(dev_alloc(&h->d_Wu, WR * h->Kp));
  CHK(dev_alloc(&h->d_Wu_ag, WR * h->Kp));
  HIPCHK(hipMemset(h->d_Wu, 0, std::max<size_t>(WR * h->Kp, 1) * sizeof(float)));
  HIPCHK(hipMemset(h->d_Wu_ag, 0, std::max<size_t>(WR * h->Kp, 1) * sizeof(float)));
  if (h->cfg.linear_function) {
    h->cnt[CDAE_P_UU] = h->cnt[CDAE_P_UU_AG] = WR * h->Kp;
    CHK(dev_alloc(&h->d_Uu, WR * h->Kp));
    CHK(dev_alloc(&h->d_Uu_ag, WR * h->Kp));
  }

  // batch workspace sized for the largest batch of B consecutive users
  const uint32_t B = (uint32_t)std::min<uint64_t>(h->B, U);
  uint64_t emax = 0;
  for (uint64_t s0 = 0; s0 < U; ++s0) {   // any window [s0, s0+B) may be requested by train_users
    const uint64_t s1 = std::min<uint64_t>(U, s0 + B);
    emax = std::max<uint64_t>(emax, (uint64_t)(row_ptr[s1] - row_ptr[s0]));
  }
  h->ex_per_pos = h->mf == 2 ? 2u * h->hp.num_neg : 1u + h->hp.num_neg;
  h->Ecap = emax * h->ex_per_pos;
  h->h_grow_ptr.clear(); h->h_gunit_ptr.clear();
  if (h->shard_sampled()) {
    // the example list of a batch is the single-GPU one (every position of the WHOLE rows, every negative draw), entries of other
    // shards' rows VOID: capacity, units and offsets come from the whole rows
    const int64_t* grp = h->g_row_ptr_src;
    const uint32_t* gcl = h->g_col_src;
    h->g_row_ptr_src = nullptr; h->g_col_src = nullptr;
    if (grp[0] != 0) return fail("whole-row row_ptr[0] must be 0");
    h->h_grow_ptr.assign(grp, grp + U + 1);
    uint64_t gmax = 0;
    for (uint64_t s0 = 0; s0 < U; ++s0) gmax = std::max<uint64_t>(gmax, (uint64_t)(grp[std::min<uint64_t>(U, s0 + B)] - grp[s0]));
    h->Ecap = gmax * h->ex_per_pos;
    const size_t gnnz = (size_t)grp[U];
    CHK(dev_alloc(&h->d_grow_ptr, U + 1)); CHK(dev_alloc(&h->d_gcol, gnnz));
    HIPCHK(hipMemcpy(h->d_grow_ptr, grp, (U + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_gcol, gcl, gnnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    h->h_gunit_ptr.assign(U + 1, 0u);
    for (uint64_t u = 0; u < U; ++u)
      h->h_gunit_ptr[u + 1] = h->h_gunit_ptr[u] + (uint32_t)((grp[u + 1] - grp[u] + h->hp.unit_pos - 1) / h->hp.unit_pos);
    CHK(dev_alloc(&h->d_gunit_ptr, U + 1));
    HIPCHK(hipMemcpy(h->d_gunit_ptr, h->h_gunit_ptr.data(), (U + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    std::vector<uint32_t> guser(h->h_gunit_ptr[U]);
    for (uint64_t u = 0; u < U; ++u)
      for (uint32_t g = h->h_gunit_ptr[u]; g < h->h_gunit_ptr[u + 1]; ++g) guser[g] = (uint32_t)u;
    CHK(dev_alloc(&h->d_gunit_user, std::max<size_t>(guser.size(), 1)));
    HIPCHK(hipMemcpy(h->d_gunit_user, guser.data(), guser.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  h->seq = 0; h->pre_n = 0;
  // opt-in (CDAE_SORT_COUNTING=1): measured slower than rocPRIM's onesweep beside the training kernels (profiles/r02_*)
  // opt-in (CDAE_SORT_TILE=1).  Measured (profiles/r02_tile_sort.txt): the four launches take 72 us against rocPRIM's ten launches / ~100 us
  // per batch on the prep stream, yet the training step is the same within 1 % at 256 users and 3 % slower at 512 — the prep
  // stream runs beside the training kernels, and what counts there is how much it disturbs them, not its own length
  // (round 4: the tile kernels skip VOID examples, so a sampled item shard can take them too — there the prep chain is NOT hidden
  // behind a look-ahead lane and the library sort's launches and fills are the larger part of it)
  h->counting_sort = I <= cdae::TILE_SORT_MAX_ITEMS && DEV_ENV("CDAE_SORT_TILE") != nullptr;
  // DEFAULT since round 5: bucket_sort_kernel — one narrow launch, every workgroup owns a range of item ids cut HERE so that the
  // ranges expect equal numbers of examples per batch: B * pop[i] / U positives + the uniformly drawn negatives.  It scans the batch's
  // whole 16-bit key list once per range and pass, so it is for batches up to BUCKET_MAX_EXAMPLES examples (256-512 users at the
  // BASELINE shapes) over at most 65536 items; beyond that the library sort below stays.  CDAE_SORT_LIBRARY (developer build): the
  // library sort + segment_kernel everywhere (the A/B side of the bit-equality tests).
  h->bucket_sort = false; h->bucket_ranges = 0; h->cell_units = 0;
  {
    // full-output blocks list their positives only and are long (thousands of users).  There the one narrow launch is the wrong shape: its
    // workgroups (1024 threads, 90 KB of LDS: a CU each) run ~130 us beside a fused decode whose grid wants every CU (ML-10M shape, 2048
    // users per block: full_decode_fused_kernel 57 -> 93 us, step 0.203 -> 0.24 ms), and from ~300 K examples on it is longer than the
    // block it prepares (4096 users: 0.43 ms against a 0.33 ms step).  Blocks above 100 K examples keep the library sort's short launches.
    const uint64_t BUCKET_MAX_EXAMPLES = h->cfg.full_output ? 100000 : 600000;
    const uint64_t keys = I + (h->shard_sampled() ? 1u : 0u);
    if (keys <= 65536 && !h->counting_sort && !h->mf_seq && h->Ecap <= BUCKET_MAX_EXAMPLES && h->Ecap > 0 && !DEV_ENV("CDAE_SORT_LIBRARY")) {
      const double b_share = (double)B / (double)U;
      const double draw_items = (double)(h->shard_sampled() ? h->I_global : I);
      const double whole_nnz = h->shard_sampled() ? (double)h->h_grow_ptr[U] : (double)nnz;
      const uint32_t negs = h->mf == 2 ? 2u * h->hp.num_neg - 0u : h->hp.num_neg;      // (BPR lists the positive once per pair too: weights only balance, any estimate is legal)
      const double neg_per_item = b_share * whole_nnz * (double)negs / draw_items;
      std::vector<double> wgt(I);
      double total = 0.0;
      for (uint64_t i = 0; i < I; ++i) { wgt[i] = b_share * (double)pop[i] + neg_per_item; total += wgt[i]; }
      const double per_range = std::max(total / 256.0, std::min(2730.0, total / 8.0));   // ~2.7 K examples per range: half the LDS window
      std::vector<uint32_t> cut{0u};
      double acc = 0.0;
      for (uint64_t i = 0; i < I; ++i) {
        acc += wgt[i];
        if (i + 1 < I && (acc >= per_range || (i + 1) - cut.back() == cdae::BK_ITEMS)) { cut.push_back((uint32_t)(i + 1)); acc = 0.0; }
      }
      cut.push_back((uint32_t)I);
      if (cut.size() - 1 <= cdae::BK_MAX_RANGES) {
        h->bucket_sort = true;
        h->bucket_ranges = (uint32_t)cut.size() - 1;
        CHK(dev_alloc(&h->d_bucket_cut, cut.size()));
        HIPCHK(hipMemcpy(h->d_bucket_cut, cut.data(), cut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        // cells: sample_kernel routes every example into the cell of (its item's range, its unit), so that the sort reads its range's cells
        // instead of scanning the batch's key list.  The CDAE sampler only; <= 256 ranges (LDS counters per wavefront); example indices must
        // fit the entry's 20 bits (Ecap <= 600 000 does)
        if (!h->mf && h->bucket_ranges <= cdae::BKC_MAX_RANGES && h->Ecap < (1ull << (32 - cdae::BKC_ITEM_BITS)) && !DEV_ENV("CDAE_SORT_SCAN")) {
          std::vector<uint16_t> rof(I);
          for (uint32_t r = 0; r + 1 < cut.size(); ++r)
            for (uint32_t i = cut[r]; i < cut[r + 1]; ++i) rof[i] = (uint16_t)r;
          CHK(dev_alloc(&h->d_range_of, (size_t)I));
          HIPCHK(hipMemcpy(h->d_range_of, rof.data(), I * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
      }
    }
  }
  h->h_unit_ptr.assign(U + 1, 0u);
  for (uint64_t u = 0; u < U; ++u)
    h->h_unit_ptr[u + 1] = h->h_unit_ptr[u] + (uint32_t)((row_ptr[u + 1] - row_ptr[u] + h->hp.unit_pos - 1) / h->hp.unit_pos);
  h->unit_cap = 0;
  for (uint64_t s0 = 0; s0 < U; ++s0) {
    const uint64_t s1 = std::min<uint64_t>(U, s0 + B);
    h->unit_cap = std::max(h->unit_cap, h->h_unit_ptr[s1] - h->h_unit_ptr[s0]);
  }
  // cdae_hip_encode takes arbitrary user lists of up to B users: the heaviest B users bound its unit count
  {
    std::vector<uint32_t> per(U);
    for (uint64_t u = 0; u < U; ++u) per[u] = h->h_unit_ptr[u + 1] - h->h_unit_ptr[u];
    std::partial_sort(per.begin(), per.begin() + B, per.end(), std::greater<uint32_t>());
    uint64_t top = 0;
    for (uint32_t i = 0; i < B; ++i) top += per[i];
    h->unit_cap = (uint32_t)std::max<uint64_t>(h->unit_cap, top);
  }
  if (h->shard_sampled())                                   // hidden_gather's partial rows are per unit of the WHOLE rows
    for (uint64_t s0 = 0; s0 < U; ++s0)
      h->unit_cap = std::max(h->unit_cap, h->h_gunit_ptr[std::min<uint64_t>(U, s0 + B)] - h->h_gunit_ptr[s0]);
  CHK(dev_alloc(&h->d_unit_ptr, U + 1));
  HIPCHK(hipMemcpy(h->d_unit_ptr, h->h_unit_ptr.data(), (U + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> unit_user(h->h_unit_ptr[U]);
    for (uint64_t u = 0; u < U; ++u)
      for (uint32_t g = h->h_unit_ptr[u]; g < h->h_unit_ptr[u + 1]; ++g) unit_user[g] = (uint32_t)u;
    CHK(dev_alloc(&h->d_unit_user, std::max<size_t>(unit_user.size(), 1)));
    HIPCHK(hipMemcpy(h->d_unit_user, unit_user.data(), unit_user.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  CHK(dev_alloc(&h->d_Hpart, (size_t)h->unit_cap * h->Kp));
  CHK(dev_alloc(&h->d_uptr_tmp, (size_t)B + 1));
  {
    const uint32_t one_unit[2] = {0u, 1u};          // explicit-input step: one user, one unit
    HIPCHK(hipMemcpy(h->d_uptr_tmp, one_unit, sizeof one_unit, hipMemcpyHostToDevice));
  }
  if ((uint64_t)B * h->Kp * sizeof(float) > 0x7FFFFFFFull)      // decode addresses z rows with 31-bit byte offsets (buffer loads, cdae_kernels.hpp)
    return fail("batch_users %u x row stride %u floats exceeds the 2 GiB the batch's Z may occupy; lower batch_users", B, h->Kp);
  if (h->Ecap > 0xFFFFFFF0ull) return fail("batch of %u users holds %llu examples (> 2^32); lower batch_users", B, (unsigned long long)h->Ecap);
  for (auto& b : h->ex) {
    CHK(dev_alloc(&b.item, h->Ecap)); CHK(dev_alloc(&b.val, h->Ecap));
    CHK(dev_alloc(&b.sorted_item, h->Ecap)); CHK(dev_alloc(&b.sorted_val, h->Ecap));
    CHK(dev_alloc(&b.seg, 4 * (size_t)I));                   // first | one-past-last position by item, then the same by popularity rank
    CHK(dev_alloc(&b.dup_of_pos, h->Ecap)); CHK(dev_alloc(&b.dup_of_ex, h->Ecap)); CHK(dev_alloc(&b.dup_count, cdae::DUP_STRIPES));
    if (h->counting_sort) {
      CHK(dev_alloc(&b.item_count, (size_t)I)); CHK(dev_alloc(&b.prefix, (size_t)I + 1));
      CHK(dev_alloc(&b.rank, (size_t)I)); CHK(dev_alloc(&b.bucketed, h->Ecap));
      CHK(dev_alloc(&b.tile_hist, (size_t)((h->Ecap + cdae::TILE_EX - 1) / cdae::TILE_EX + 1) * I));
      CHK(dev_alloc(&b.block_total, 128));
    } else if (I + (h->shard_sampled() ? 1u : 0u) <= 65536) { CHK(dev_alloc(&b.key16, h->Ecap + 8)); CHK(dev_alloc(&b.sorted_key16, h->Ecap)); }   // (a sampled item shard sorts one more key: VOID = I)
    if (h->bucket_sort) { CHK(dev_alloc(&b.bucketed, h->Ecap)); }
    b.cell_tag = 0;
    if (h->bucket_sort && h->d_range_of) {
      // [ranges][units of the largest batch][BKC_SLOTS] words; beyond 256 MiB per set (or 16 384 units) the sort scans instead
      const size_t words = (size_t)h->bucket_ranges * h->unit_cap * cdae::BKC_SLOTS;
      if (h->unit_cap <= cdae::BK_CELL_UNITS_MAX && words * sizeof(uint32_t) <= (256ull << 20)) {
        CHK(dev_alloc(&b.cells, words));
        CHK(dev_alloc(&b.cell_flag, 1));
        HIPCHK(hipMemset(b.cell_flag, 0, sizeof(uint32_t)));
        h->cell_units = h->unit_cap;
      }
    }
    CHK(dev_alloc(&b.wg_state, cdae::BK_MAX_RANGES));
    HIPCHK(hipMemset(b.wg_state, 0, cdae::BK_MAX_RANGES * sizeof(uint32_t)));
    if (!b.ready) { HIPCHK(hipEventCreateWithFlags(&b.ready, sync_event_flags())); HIPCHK(hipEventCreateWithFlags(&b.released, sync_event_flags())); }
    HIPCHK(hipEventRecord(b.released, h->stream));
  }
  CHK(dev_alloc(&h->d_D0, IK));
  {
    // one correction row per duplicate negative of a batch (~2 % of the examples at ML-10M shape); beyond the
    // capacity decode falls back to atomics.  Zero-filled once: decode's 16-lane path leaves elements >= 64 NV + 16 NT
    // of a row untouched and the gather reads whole rows.
    const char* ev = DEV_ENV("CDAE_DUP_CAP");
    const uint64_t want = h->mf ? 1 : (ev ? std::strtoull(ev, nullptr, 10) : std::max<uint64_t>(65536, h->Ecap / 4));   // small problems: every example (IMF / BPR have no correction rows)
    h->dup_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want, 1), std::max<uint64_t>(h->Ecap, 1));
    h->dup_stripes = std::max<uint32_t>(1u, std::min<uint32_t>(cdae::DUP_STRIPES, h->dup_cap / 4096u));
    if ((uint64_t)h->dup_cap * h->Kp * 4u >= (1ull << 31)) h->fused_possible = h->fused_decode = false;     // (32-bit buffer offsets in the fused launch)
    CHK(dev_alloc(&h->d_dup_corr, (size_t)h->dup_cap * h->Kp));
    HIPCHK(hipMemset(h->d_dup_corr, 0, (size_t)h->dup_cap * h->Kp * sizeof(float)));
  }
  CHK(dev_alloc(&h->d_G, h->Ecap));
  if (h->mf) {
    const uint64_t inst_cap = emax * (h->mf == 2 ? h->hp.num_neg : 1u + h->hp.num_neg);
    CHK(dev_alloc(&h->d_UVpre, (size_t)inst_cap * h->Kp));
    CHK(dev_alloc(&h->d_ub, (size_t)U)); CHK(dev_alloc(&h->d_ub_ag, (size_t)U));
    h->cnt[CDAE_P_UB] = h->cnt[CDAE_P_UB_AG] = (size_t)U;
    HIPCHK(hipMemset(h->d_ub, 0, (size_t)U * sizeof(float)));
  }
  h->sort_bits = 1;
  while ((1ull << h->sort_bits) < I + (h->shard_sampled() ? 1u : 0u)) h->sort_bits++;
  h->sort_tmp_bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, h->sort_tmp_bytes, h->ex[0].item, h->ex[0].sorted_item, h->ex[0].val,
                                   h->ex[0].sorted_val, (size_t)std::max<uint64_t>(h->Ecap, 1), 0u, (unsigned)h->sort_bits, h->stream));
  if (h->ex[0].key16) {
    size_t bytes16 = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes16, h->ex[0].key16, h->ex[0].sorted_key16, h->ex[0].val, h->ex[0].sorted_val,
                                     (size_t)std::max<uint64_t>(h->Ecap, 1), 0u, (unsigned)h->sort_bits, h->stream));
    h->sort_tmp_bytes = std::max(h->sort_tmp_bytes, bytes16);
  }
  h->sort_tmp_stride = (h->sort_tmp_bytes + 255) & ~(size_t)255;
  CHK(dev_alloc((char**)&h->d_sort_tmp, 2 * h->sort_tmp_stride));
  const size_t BK = (size_t)B * h->Kp;
  CHK(dev_alloc(&h->d_Z, BK)); CHK(dev_alloc(&h->d_Dz, BK)); CHK(dev_alloc(&h->d_HG, BK));
  if (h->cfg.linear_function) { CHK(dev_alloc(&h->d_Ssum, BK)); CHK(dev_alloc(&h->d_delta_rows, BK)); }
  if (h->cfg.full_output) {
    h->Bp = (B + 127u) & ~127u;
    // big item spaces and K > 256: 256-row GEMM tiles (the K > 256 launches of round 3 — gemm1_loss_zreg_kernel, gemm_tn_bf16_kernel — want them)
    h->Ip = (I >= 32768 || h->Kp > 256) ? (((uint32_t)I + 255u) & ~255u) : (((uint32_t)I + 127u) & ~127u);
    CHK(dev_alloc(&h->d_Zb, (size_t)h->Bp * h->Kp)); CHK(dev_alloc(&h->d_ZTb, (size_t)h->Kp * h->Bp));
    CHK(dev_alloc(&h->d_Db, (size_t)h->Ip * h->Kp)); CHK(dev_alloc(&h->d_DTb, (size_t)h->Kp * h->Ip));
    // G [Bp x Ip] only where a launch reads it: the NT form of GEMM 2 (K > 256 without gemm_tn_bf16_kernel, or CDAE_FULL_UNFUSED);
    // the fused K <= 256 kernel and the TN form read G^T alone (2 GB less per handle at 1 M items x 1024 users)
    if ((h->Kp > 256 || h->full_unfused) && !gemm2_tn_path(h)) CHK(dev_alloc(&h->d_Gb, (size_t)h->Bp * h->Ip));
    CHK(dev_alloc(&h->d_GTb, (size_t)h->Ip * h->Bp));
    CHK(dev_alloc(&h->d_dD, (size_t)h->Ip * h->Kp));
    CHK(dev_alloc(&h->d_has_in, (size_t)h->Ip));
    HIPCHK(hipMemsetAsync(h->d_has_in, 0, (size_t)h->Ip, h->stream));
    {
      // the rated-items bitmap is read by the fused K <= 256 decode only (the K = 512 launches patch their positives in place:
      // full_positive_fixup_kernel) — round 5: no longer built where nothing reads it (128 MB and ~1 ms of prep stream per
      // 1024-user block at 1 M items)
      if (h->Kp <= 256 && !h->full_unfused) {
        h->bits_stride = (size_t)B * ((I + 31) / 32);
        CHK(dev_alloc(&h->d_bits_train, cdae_hip::NSETS * h->bits_stride));     // one per example-buffer set
      }
      // item slices of the fused decode: slices x Bp/128 workgroups ~ one per CU (measured best at B = 2048: 16 slices; every
      // slice adds a [B x Kp] partial of hg)
      const uint32_t tiles = h->Ip / (32 * cdae::FUSED_SUB), ublocks = h->Bp / 128;
      h->full_slices = std::max<uint32_t>(1, std::min<uint32_t>({32u, tiles, (256u + ublocks - 1) / ublocks}));
      if (const char* ev = DEV_ENV("CDAE_FULL_SLICES")) h->full_slices = std::max<uint32_t>(1, std::min<uint32_t>(tiles, (uint32_t)std::atoi(ev)));
    }
  }
  if (h->cfg.full_output || h->item_shard) {
    std::vector<uint32_t> iota((size_t)std::max<uint32_t>(B, h->item_shard ? EVAL_CHUNK : 0u) + 1);
    std::iota(iota.begin(), iota.end(), 0u);
    CHK(dev_alloc(&h->d_iota, iota.size()));
    HIPCHK(hipMemcpy(h->d_iota, iota.data(), iota.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    // item shard: [input sums | gathered Wu rows | gathered Uu rows] of a batch, one all-reduce buffer
    if (h->item_shard) CHK(dev_alloc(&h->d_Hsum, (size_t)SHARD_BLOCKS * B * h->Kp));
  }
  {
    size_t rows = 8 * (size_t)h->gather_halves * h->unit_cap;
    if (h->cfg.full_output) {
      rows = std::max(rows, (size_t)h->full_slices * B);
      const uint32_t kps = gemm2_k_per_split(h);                                 // unfused path: one [Bp x Kp] slab per contraction split
      rows = std::max(rows, (size_t)((h->Ip + kps - 1) / kps) * h->Bp);
    }
    CHK(dev_alloc(&h->d_HGpart, rows * h->Kp));
  }
  CHK(dev_alloc(&h->d_touched, (size_t)I));
  HIPCHK(hipMemset(h->d_touched, 0, (size_t)I * sizeof(uint32_t)));
  CHK(dev_alloc(&h->d_uids, (size_t)B));
  // weights 0, accumulators 1e-4 (cdae.hpp:114,...), accumulator pad lanes 1: a well-defined state even
  // before init_params / set_param (a zero accumulator pad would make beta == 0 divide 0 by 0)
  auto fillm = [&](float* M, size_t rows, uint32_t K, uint32_t Kp, float v, float pad) {
    if (M && rows) hipLaunchKernelGGL(cdae::fill_matrix_kernel, dim3((uint32_t)((rows * Kp + 255) / 256)), dim3(256), 0, h->stream, M, rows,
                                      K, Kp, v, pad);
  };
  fillm(h->P(CDAE_P_W_AG), I, h->K, h->Kp, 1e-4f, 1.f);
  fillm(h->P(CDAE_P_V_AG), I, h->K, h->Kp, 1e-4f, 1.f);
  fillm(h->d_Wu_ag, WR, h->K, h->Kp, 1e-4f, 1.f);
  fillm(h->d_Uu, WR, h->K, h->Kp, 1.f, 0.f);                // cdae.hpp:131-132
  fillm(h->d_Uu_ag, WR, h->K, h->Kp, 1e-4f, 1.f);
  fillm(h->P(CDAE_P_B_AG), 1, h->K, h->Kp, 1e-4f, 1.f);
  fillm(h->P(CDAE_P_BP_AG), I, 1, 1, 1e-4f, 1.f);
  fillm(h->d_ub_ag, U, 1, 1, 1e-4f, 1.f);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int cdae_hip_init_params(cdae_hip_t* h, uint64_t seed) {
  if (h) h->db_valid = h->db_rows_valid = false;      // the bf16 images of the decoder no longer match it (full-output path, compute_batch_full)
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  using namespace cdae;
  const double init_scale = 4. * std::sqrt(6. / (double)((h->item_shard ? h->I_global : h->I) + h->K));          // cdae.hpp:112 (an item shard: the whole item space)
  auto blocks = [](size_t n) { return dim3((uint32_t)((n + 255) / 256)); };
  auto init = [&](float* M, size_t rows, uint32_t id, uint64_t row0 = 0) {
    hipLaunchKernelGGL(init_matrix_kernel, blocks(rows * h->Kp), dim3(256), 0, h->stream, M, rows, h->K, h->Kp,
                       cdae_rng_key(seed, 0, id, CDAE_STREAM_INIT), init_scale, row0);
  };
  auto fill = [&](float* M, size_t rows, uint32_t K, uint32_t Kp, float v) {     // accumulator pads are 1, others 0
    hipLaunchKernelGGL(fill_matrix_kernel, blocks(rows * Kp), dim3(256), 0, h->stream, M, rows, K, Kp, v, v == 0.f ? 0.f : 1.f);
  };
  if (h->mf) {                                             // imf.hpp:57-69: Random() * 0.01, accumulators 1e-4, biases 0
    auto init01 = [&](float* M, size_t rows, uint32_t id, uint64_t row0) {
      hipLaunchKernelGGL(init_matrix_kernel, blocks(rows * h->Kp), dim3(256), 0, h->stream, M, rows, h->K, h->Kp,
                         cdae_rng_key(seed, 0, id, CDAE_STREAM_INIT), 0.01, row0);
    };
    init01(h->d_Wu, h->U, CDAE_P_WU, h->uid_offset); fill(h->d_Wu_ag, h->U, h->K, h->Kp, 1e-4f);
    init01(h->P(CDAE_P_W), h->I, CDAE_P_W, 0); fill(h->P(CDAE_P_W_AG), h->I, h->K, h->Kp, 1e-4f);
    fill(h->d_ub, h->U, 1, 1, 0.f); fill(h->d_ub_ag, h->U, 1, 1, 1e-4f);
    fill(h->P(CDAE_P_BP), h->I, 1, 1, 0.f); fill(h->P(CDAE_P_BP_AG), h->I, 1, 1, 1e-4f);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  init(h->P(CDAE_P_W), h->I, CDAE_P_W, h->item0); fill(h->P(CDAE_P_W_AG), h->I, h->K, h->Kp, 1e-4f);     // :113-114
  if (h->cfg.asymmetric) { init(h->P(CDAE_P_V), h->I, CDAE_P_V, h->item0); fill(h->P(CDAE_P_V_AG), h->I, h->K, h->Kp, 1e-4f); }   // :115-118
  const size_t WR = (size_t)h->wu_rows();                   // (an item shard: its own user range, rows keyed by global user id)
  const uint64_t wu_row0 = h->uid_offset + (h->item_shard ? h->own_u0 : 0);
  if (h->cfg.user_factor && WR) { init(h->d_Wu, WR, CDAE_P_WU, wu_row0); fill(h->d_Wu_ag, WR, h->K, h->Kp, 1e-4f); }   // :119-122
  else if (WR) { fill(h->d_Wu, WR, h->K, h->Kp, 0.f); fill(h->d_Wu_ag, WR, h->K, h->Kp, 1e-4f); }
  fill(h->P(CDAE_P_B), 1, h->K, h->Kp, 0.f); fill(h->P(CDAE_P_B_AG), 1, h->K, h->Kp, 1e-4f);    // :123-124
  fill(h->P(CDAE_P_BP), h->I, 1, 1, 0.f); fill(h->P(CDAE_P_BP_AG), h->I, 1, 1, 1e-4f);          // :125-126
  if (h->cfg.linear_function && WR) {                                                           // :130-133
    hipLaunchKernelGGL(fill_matrix_kernel, blocks(WR * h->Kp), dim3(256), 0, h->stream, h->d_Uu, WR, h->K, h->Kp, 1.f, 0.f);
    fill(h->d_Uu_ag, WR, h->K, h->Kp, 1e-4f);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int cdae_hip_set_param(cdae_hip_t* h, uint32_t which, const float* host, size_t count) {
  if (h) h->db_valid = h->db_rows_valid = false;      // the bf16 images of the decoder no longer match it (full-output path, compute_batch_full)
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  if (which >= CDAE_P_COUNT || !host) return fail("bad argument");
  std::vector<float> by_pos;
  if (!h->user_perm.empty() && is_user_indexed(which) && count % h->U == 0) {   // by user id -> by training position
    by_pos.resize(count);
    const size_t w = count / h->U;
    for (uint64_t pos = 0; pos < h->U; ++pos) std::copy(host + (size_t)h->user_perm[pos] * w, host + ((size_t)h->user_perm[pos] + 1) * w, by_pos.begin() + pos * w);
    host = by_pos.data();
  }
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  float* d = h->P(which);
  if (!d) return fail("parameter %u is not allocated in this configuration", which);
  if (which == CDAE_P_BP || which == CDAE_P_BP_AG || which == CDAE_P_UB || which == CDAE_P_UB_AG) {
    const size_t want = (which == CDAE_P_UB || which == CDAE_P_UB_AG) ? h->U : h->I;
    if (count != want) return fail("parameter %u has %llu elements, got %zu", which, (unsigned long long)want, count);
    HIPCHK(hipMemcpy(d, host, count * sizeof(float), hipMemcpyHostToDevice));
    return 0;
  }
  const size_t rows = (which == CDAE_P_WU || which == CDAE_P_WU_AG || which == CDAE_P_UU || which == CDAE_P_UU_AG) ? (size_t)h->wu_rows() : ((which == CDAE_P_B || which == CDAE_P_B_AG) ? 1 : h->I);
  if (count != rows * h->K) return fail("parameter %u has %zu elements, got %zu", which, rows * h->K, count);
  if (rows == 0) return 0;
  const bool is_acc = (which & 1u) != 0u;                 // odd ids are the *_AG accumulators: pad lanes stay 1
  hipLaunchKernelGGL(cdae::fill_matrix_kernel, dim3((uint32_t)((rows * h->Kp + 255) / 256)), dim3(256), 0, h->stream, d, rows,
                     h->K, h->Kp, 0.f, is_acc ? 1.f : 0.f);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy2D(d, h->Kp * sizeof(float), host, h->K * sizeof(float), h->K * sizeof(float), rows, hipMemcpyHostToDevice));
  return 0;
}

int cdae_hip_get_param(cdae_hip_t* h, uint32_t which, float* host, size_t count) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  if (which >= CDAE_P_COUNT || !host) return fail("bad argument");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  if (!h->user_perm.empty() && is_user_indexed(which)) {    // device rows are in training order: hand them out by user id
    std::vector<float> tmp(count);
    CHK(copy_param_out(h, which, tmp.data(), count));
    const size_t w = count / h->U;
    for (uint64_t pos = 0; pos < h->U; ++pos) std::copy(tmp.begin() + pos * w, tmp.begin() + (pos + 1) * w, host + (size_t)h->user_perm[pos] * w);
    return 0;
  }
  return copy_param_out(h, which, host, count);
}

int cdae_hip_param_device_ptr(cdae_hip_t* h, uint32_t which, void** device_ptr, size_t* padded_count) {
  if (!h || !h->d_shared || which >= CDAE_P_COUNT || !device_ptr) return fail("bad argument");
  const bool user_indexed = which == CDAE_P_WU || which == CDAE_P_WU_AG || which == CDAE_P_UU || which == CDAE_P_UU_AG || which == CDAE_P_UB || which == CDAE_P_UB_AG;
  if (user_indexed && !h->user_perm.empty())
    return fail("cdae_hip_param_device_ptr: the user-indexed arrays of an IMF / BPR block-schedule handle are stored in TRAINING order "
                "(cdae_hip_user_order), not by user id: use cdae_hip_get_param / cdae_hip_set_param");
  // the pointer is writable: whoever writes the parameters through it must find the full-output path re-imaging the decoder
  h->db_valid = h->db_rows_valid = false;
  *device_ptr = h->P(which);
  if (padded_count) *padded_count = h->cnt[which];
  return 0;
}

int cdae_hip_set_profiling(cdae_hip_t* h, int enabled) {
  if (!h) return fail("null handle");
  h->profiling = enabled < 0 ? 0 : enabled;
  return 0;
}

int cdae_hip_set_profiling_families(cdae_hip_t* h, uint32_t mask) {
  if (!h) return fail("null handle");
  h->prof_mask = mask;
  return 0;
}

int cdae_hip_synchronize(cdae_hip_t* h) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  return fused_check(h);
}

}  // extern "C"

namespace {

int make_plan(cdae_hip* h, uint64_t u_begin, uint64_t u_end, std::vector<Batch>& plan) {
  if (u_begin > u_end || u_end > h->U) return fail("bad user range [%llu, %llu)", (unsigned long long)u_begin, (unsigned long long)u_end);
  const uint32_t B = (uint32_t)std::min<uint64_t>(h->B, h->U);
  for (uint64_t s0 = u_begin; s0 < u_end; s0 += B) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, u_end - s0);
    const uint64_t E = (uint64_t)(h->h_row_ptr[s0 + nb] - h->h_row_ptr[s0]) * h->ex_per_pos;
    if (E > h->Ecap || E > 0xFFFFFFF0ull) return fail("batch has %llu examples, capacity %llu", (unsigned long long)E, (unsigned long long)h->Ecap);
    for (uint32_t c = 0; c < h->cfg.num_corruptions; ++c) plan.push_back(Batch{s0, nb, c, E});       // cdae.hpp:141
  }
  return 0;
}

inline int set_of(uint64_t q) { return (int)(q % cdae_hip::NSETS); }
// two lanes only where the second one is free: the sampled CDAE path (the full-output path runs its b recurrence on aux, an
// item shard trains in phases)
constexpr uint64_t PREP2_AUTO_MAX_USERS = 384;
inline size_t prep_depth(const cdae_hip* h) {
  if (!h->prep2 || h->mf || h->cfg.full_output || h->item_shard) return 1;
  if (h->prep2_auto && std::min<uint64_t>(h->B, h->U) > PREP2_AUTO_MAX_USERS) return 1;
  // an exchange runs its collective on the aux stream too, issued by the caller's thread while the worker issues lane 1: whichever
  // is queued first decides whether the next batch's lists wait behind an all-reduce (measured one-rank: 0.115 -> 0.151 ms)
  if (h->prep2 == h->aux && h->xchg) return 1;
  return 2;
}
inline int prep_lane(const cdae_hip* h, uint64_t q) { return prep_depth(h) == 2 ? (int)(q & 1) : 0; }
// ---- prep worker ---------------------------------------------------------------------------------------------------
// The caller's thread hands a batch's sampling + sorting to the worker (submit_prep) and, before it issues the training
// kernels that wait on the set's `ready` event, makes sure the worker has ISSUED that record (await_prep: host-side order
// of hipEventRecord before hipStreamWaitEvent).  The other direction needs no handshake: a set's `released` record is issued
// by the caller before it submits the job that reuses the set (NSETS > look-ahead depth).
void prep_worker_main(cdae_hip* h) {
  (void)hipSetDevice(h->device);
  for (;;) {
    cdae_hip::PrepJob j;
    {
      std::unique_lock<std::mutex> lk(h->job_mu);
      h->job_cv.wait(lk, [&] { return h->worker_stop || !h->jobs.empty(); });
      if (h->jobs.empty()) return;                           // stop requested and nothing left
      j = h->jobs.front();
      h->jobs.pop_front();
    }
    if (!h->worker_failed.load(std::memory_order_relaxed)) {
      const int rc = prep_batch(h, j.set, Batch{j.s0, j.nb, j.cidx, j.E}, j.seed, j.epoch, j.lane, j.prof_q);
      if (rc) {
        h->worker_error = cdae_hip_last_error();             // (thread-local on the worker; published by the release store below)
        h->worker_failed.store(1, std::memory_order_release);
      }
    }
    {
      // (under the mutex so that a caller between its predicate check and its wait cannot miss the wake-up)
      std::lock_guard<std::mutex> lk(h->job_mu);
      h->jobs_issued.fetch_add(1, std::memory_order_release);
    }
    h->issued_cv.notify_all();
  }
}

int submit_prep(cdae_hip* h, int set, const Batch& bt, uint64_t seed, uint32_t epoch, int lane, uint64_t prof_q) {
  if (!h->prep_threaded) return prep_batch(h, set, bt, seed, epoch, lane, prof_q);
  if (!h->worker.joinable()) h->worker = std::thread(prep_worker_main, h);
  {
    std::lock_guard<std::mutex> lk(h->job_mu);
    h->jobs.push_back(cdae_hip::PrepJob{set, bt.s0, bt.nb, bt.cidx, bt.E, seed, epoch, lane, prof_q});
  }
  h->jobs_submitted++;
  h->job_cv.notify_one();
  return 0;
}

// jobs 1..id have been issued (their launches and their `ready` records are in the prep stream)
// A failed job is reported ONCE, to the call that was waiting for it: the jobs queued behind it were skipped (their example sets were
// never written), so the queue is drained, whatever was prefetched is forgotten and the flag is cleared — the next call starts
// clean instead of finding the handle bricked by one transient launch error.
int await_prep_upto(cdae_hip* h, uint64_t id) {
  if (!h->prep_threaded) return 0;
  auto wait_for = [&](uint64_t n) {
    for (uint32_t spins = 0; spins < 4096; ++spins) {        // the worker is normally a few microseconds behind at most
      if (h->jobs_issued.load(std::memory_order_acquire) >= n) return;
      __builtin_ia32_pause();
    }
    std::unique_lock<std::mutex> lk(h->job_mu);
    h->issued_cv.wait(lk, [&] { return h->jobs_issued.load(std::memory_order_acquire) >= n; });
  };
  wait_for(id);
  if (h->worker_failed.load(std::memory_order_acquire)) {
    wait_for(h->jobs_submitted);
    const std::string msg = h->worker_error;
    h->pre_n = 0;
    h->worker_failed.store(0, std::memory_order_release);
    return fail("prep worker: %s", msg.c_str());
  }
  return 0;
}
int await_prep(cdae_hip* h) { return await_prep_upto(h, h->jobs_submitted); }

void stop_prep_worker(cdae_hip* h) {
  if (!h->worker.joinable()) return;
  {
    std::lock_guard<std::mutex> lk(h->job_mu);
    h->worker_stop = true;
  }
  h->job_cv.notify_one();
  h->worker.join();
}

int sync_prep(cdae_hip* h) {
  CHK(await_prep(h));
  HIPCHK(hipStreamSynchronize(h->prep));
  if (h->prep2) HIPCHK(hipStreamSynchronize(h->prep2));
  return 0;
}

bool is_prefetched(const cdae_hip* h, size_t t, const Batch& b, uint64_t seed, uint32_t epoch) {
  if (t >= h->pre_n) return false;
  const cdae_hip::PreBatch& p = h->pre[t];
  return p.s0 == b.s0 && p.nb == b.nb && p.cidx == b.cidx && p.seed == seed && p.epoch == epoch;
}
// leading batches of `plan` that are already prepared, at most the look-ahead depth
size_t prefetched_prefix(const cdae_hip* h, const std::vector<Batch>& plan, uint64_t seed, uint32_t epoch) {
  size_t k = 0;
  while (k < prep_depth(h) && k < plan.size() && is_prefetched(h, k, plan[k], seed, epoch)) ++k;
  return k;
}
// Prepared batches nobody will train on are dropped.  Their launches may still be in flight, and the batch that takes their
// set over may be prepared from the other lane: wait for them (a rare path: a caller that prefetches one range and trains another)
int drop_prefetched(cdae_hip* h, size_t keep) {
  if (h->pre_n <= keep) return 0;
  h->pre_n = (uint32_t)keep;
  return sync_prep(h);
}

// Enqueue (no synchronisation with the MAIN stream; the caller's thread may look for up to host_pace_us per batch at the prep chain's
// event, see below) one pass over users [u_begin, u_end): a software pipeline in which the
// sampling + sorting of batch t+1 (prep stream) overlaps the training of batch t (main stream).
int enqueue_users(cdae_hip* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end) {
  if (h->item_shard) return fail("an item shard trains in phases under cdae_hip_multi_train_epoch, not on its own");
  std::vector<Batch> plan;
  CHK(make_plan(h, u_begin, u_end, plan));
  if (plan.empty()) return 0;
  // look-ahead: batch t + depth is prepared while batch t trains; with two prep lanes consecutive batches alternate between them
  const size_t depth = prep_depth(h);
  const uint64_t q0 = h->seq;
  std::vector<uint64_t> job_of(plan.size(), h->jobs_submitted);   // the job (count) that must be issued before batch t may train
  auto prep = [&](size_t t) -> int {
    const uint64_t q = q0 + t;
    if (!(h->debug_skip_prep && q >= 2 * cdae_hip::NSETS)) CHK(submit_prep(h, set_of(q), plan[t], seed, epoch, prep_lane(h, q), q));
    job_of[t] = h->jobs_submitted;
    return 0;
  };
  const size_t have = prefetched_prefix(h, plan, seed, epoch);
  // what was prepared beyond this call's batches stays for the next call (one-batch calls with two batches of look-ahead)
  const bool carry = have == plan.size() && h->pre_n > have;
  if (carry) {
    for (uint32_t t = (uint32_t)have; t < h->pre_n; ++t) h->pre[t - have] = h->pre[t];
    h->pre_n -= (uint32_t)have;
  } else {
    CHK(drop_prefetched(h, have));       // (prepared for a range the caller did not come back to)
    h->pre_n = 0;
  }
  for (size_t t = have; t < depth && t < plan.size(); ++t) CHK(prep(t));
  for (size_t t = 0; t < plan.size(); ++t) {
    if (t + depth < plan.size()) CHK(prep(t + depth));
    CHK(await_prep_upto(h, job_of[t]));                          // the set's `ready` record is in the prep stream
    h->prof_q = h->seq;
    const int set = set_of(h->seq);
    if (h->host_pace_us && !h->mf && !h->cfg.full_output) {
      // Host pacing (round 6): a wait for `ready` is a barrier packet between the encode and the decode launch, ~2-3 us of idle main
      // stream per batch (HISTORY.md "What the two event packets of a step cost").  When the lists are complete by the time the batch is
      // enqueued there is nothing to wait for on the device: the caller's thread looks (for at most host_pace_us — the prep chain
      // of a batch is ~90 us), and only a batch whose lists are late leaves the wait to the device as before.  Measured 0.0906 ->
      // 0.0879 ms per 256-user step at ML-10M shape (the thread is then ~2 batches ahead of the device instead of a whole call).
      const auto t_in = std::chrono::steady_clock::now();
      for (;;) {
        const hipError_t e = hipEventQuery(h->ex[set].ready);
        if (e == hipSuccess) { h->skip_ready_wait = true; break; }
        (void)hipGetLastError();                                  // (hipErrorNotReady is not an error)
        if (e != hipErrorNotReady) return fail("hipEventQuery: %s", hipGetErrorString(e));
        if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_in).count() >= (long)h->host_pace_us) break;
        for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
      }
    }
    int rc = 0;
    if (h->mf) rc = compute_batch_mf(h, set, plan[t]);
    else if (h->cfg.full_output) rc = compute_batch_full(h, set, plan[t], seed, epoch);
    else rc = compute_batch(h, set, plan[t], seed, epoch);
    h->skip_ready_wait = false;
    if (rc) return rc;
    h->seq++;
    h->acc_examples += plan[t].E; h->acc_batches += h->mf_seq ? plan[t].nb : 1u; h->acc_users += plan[t].nb;     // (mf_seq: a block is one user)
  }
  return 0;
}

int fill_stats(cdae_hip* h, cdae_hip_stats* stats) {
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->users = h->acc_users; stats->examples = h->acc_examples; stats->batches = h->acc_batches;
  }
  h->acc_users = h->acc_examples = h->acc_batches = 0;
  if (h->profiling) CHK(collect_profile(h, stats));
  return 0;
}

}  // namespace

// recommend(), general path (cdae_kernels.hpp recommend_kernel): any num_dim / topk / item count; one workgroup per user.
// rated != nullptr: ONE user whose input set and mask are the caller's list (sorted, unique) instead of the train row.
namespace {
// cdae_hip_eval_topn: the lists of users [u0, u0 + nu) sit in h->d_rec — score them against the test rows (same stream, no host round trip)
void topn_chunk(cdae_hip* h, uint32_t topk, uint64_t u0, uint32_t nu) {
  hipLaunchKernelGGL(cdae::topn_user_kernel, dim3((nu + 255) / 256), dim3(256), 0, h->stream, (const uint32_t*)h->d_rec, topk, u0, nu,
                     (const int64_t*)h->d_test_ptr, (const uint32_t*)h->d_test_col, (double)h->test_users_with_rows, h->d_topn_pu,
                     reinterpret_cast<unsigned long long*>(h->d_topn_out + 8));
}
int recommend_general(cdae_hip* h, uint64_t u_begin, uint64_t u_end, uint32_t topk, uint32_t* out, const uint32_t* rated, uint32_t n_rated) {
  const size_t lds_scores = (size_t)h->I * sizeof(float) + 64;
  const bool in_lds = lds_scores <= 160 * 1024;
  const size_t shmem = in_lds ? lds_scores : 64;
  // users per launch: a chunk of the EVALUATION workspace, not of the training batch (through round 3 it was batch_users: a handle
  // with one user per block — the reference schedule — evaluated 16 384 users in 16 384 launches + host round trips, 75 s)
  uint32_t B = rated ? 1u : (uint32_t)std::min<uint64_t>(h->mf ? EVAL_CHUNK : 4096u, std::max<uint64_t>(h->U, 1));
  if (!in_lds) {     // scores of a launch in a global workspace of <= 256 MiB
    B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(B, (256ull << 20) / ((uint64_t)h->I * sizeof(float))));
    if (h->score_cap < (size_t)B * h->I) {
      if (h->d_score) HIPCHK(hipFree(h->d_score));
      h->d_score = nullptr; h->score_cap = 0;
      CHK(dev_alloc(&h->d_score, (size_t)B * h->I));
      h->score_cap = (size_t)B * h->I;
    }
  }
  if (h->rec_cap < (size_t)B * topk) {
    if (h->d_rec) HIPCHK(hipFree(h->d_rec));
    h->d_rec = nullptr; h->rec_cap = 0;
    CHK(dev_alloc(&h->d_rec, (size_t)B * topk));
    h->rec_cap = (size_t)B * topk;
  }
#define SET_SHMEM(NI_) HIPCHK(hipFuncSetAttribute((const void*)cdae::recommend_kernel<NI_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
  switch (h->NI) { case 1: SET_SHMEM(1); break; case 2: SET_SHMEM(2); break; case 4: SET_SHMEM(4); break; default: SET_SHMEM(8); break; }
#undef SET_SHMEM
  uint32_t* d_rated = nullptr;
  if (rated) {
    CHK(dev_alloc(&d_rated, std::max<uint32_t>(n_rated, 1)));
    if (n_rated) HIPCHK(hipMemcpyAsync(d_rated, rated, n_rated * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    const uint32_t one_unit[2] = {0u, 1u};
    HIPCHK(hipMemcpyAsync(h->d_uptr_tmp, one_unit, sizeof one_unit, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  int rc = 0;
  for (uint64_t s0 = u_begin; s0 < u_end && !rc; s0 += B) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, u_end - s0);
    if (rated) {
      // get_hidden_values(uid, rated_items) with the default scale 1 (cdae.hpp:169; q == 1 -> empty input, :168-172)
      const bool none = h->hp.keep_thr == 0x100000000ull;
      DISPATCH_NI(h->NI, cdae::encode_partial_kernel, dim3(1), dim3(256), 0, h->stream, h->hp, h->d_row_ptr, h->d_col, h->P(CDAE_P_W),
                  (const uint32_t*)h->d_uptr_tmp, 1u, (const uint32_t*)nullptr, s0, 1u, 0, CDAE_STREAM_CORRUPT, 0u, (uint64_t)0, 0u, h->d_Hpart,
                  (const uint32_t*)d_rated, none ? 0u : n_rated, (const uint32_t*)nullptr);
      DISPATCH_NI(h->NI, cdae::encode_finish_kernel, dim3(1), dim3(256), 0, h->stream, h->hp, h->d_Hpart, (const uint32_t*)h->d_uptr_tmp, h->d_Wu,
                  h->P(CDAE_P_B), (const uint32_t*)nullptr, s0, 1u, 0, h->d_Z, (float*)nullptr, (float*)nullptr, h->d_Uu, (float*)nullptr);
    } else if (!h->mf) {
      rc = ensure_eval_ws(h, nb, h->h_unit_ptr[s0 + nb] - h->h_unit_ptr[s0]);
      if (!rc) rc = encode_chunk(h, nullptr, s0, nb, 0, CDAE_STREAM_CORRUPT, 0, 0, 0, 0, h->d_zeval, h->d_hpart_eval, h->eval_unit_cap);      // cdae.hpp:167-172
      if (rc) break;
    }
    DISPATCH_NI(h->NI, cdae::recommend_kernel, dim3(nb), dim3(256), shmem, h->stream, h->hp, h->d_row_ptr, h->d_col, s0,
                h->mf ? h->d_Wu + (size_t)s0 * h->Kp : (rated ? h->d_Z : h->d_zeval), h->dec(), h->P(CDAE_P_BP), topk, h->d_rec, in_lds ? (float*)nullptr : h->d_score, (const uint32_t*)d_rated, n_rated, (float*)nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && h->topn_active) { topn_chunk(h, topk, s0, nb); e = hipGetLastError(); }
    if (e == hipSuccess && out) e = hipMemcpyAsync(out + (s0 - u_begin) * topk, h->d_rec, (size_t)nb * topk * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && out) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail("recommend: %s", hipGetErrorString(e));
  }
  if (d_rated) (void)hipFree(d_rated);
  return rc;
}
}  // namespace

// geometry and launch of the pipelined-exchange kernels (delta_pipe_kernel)
namespace {
struct PipeGeom { uint32_t Kc; size_t n_tail, n_compact, threads; };
PipeGeom pipe_geom(const cdae_hip* h) {
  PipeGeom g;
  g.Kc = (h->K + 3u) & ~3u;
  g.n_tail = h->n_shared - h->n_matrix;
  const size_t n_rows = h->n_matrix / h->Kp;
  g.n_compact = n_rows * g.Kc + g.n_tail;
  g.threads = n_rows * (g.Kc / 4) + g.n_tail / 4 + g.n_tail % 4;
  return g;
}
// SYNC: the synchronous exchange's forms of STAGE / MERGE (cdae_kernels.hpp delta_pipe_kernel: no snap, no send copy)
template <int MODE, bool SYNC = false>
int launch_pipe(cdae_hip* h) {
  const PipeGeom g = pipe_geom(h);
  if (h->delta_combine == CDAE_COMBINE_GLOBAL_ACC && h->cfg.using_adagrad) {
    // (parameter, accumulator) pairs: same compact buffers, half the threads of the element-wise pass each moving two elements
    const size_t n_rows = h->n_matrix / h->Kp, threads = (n_rows / 2) * (g.Kc / 4) + h->I + h->Kp;
    hipLaunchKernelGGL((cdae::delta_pipe_pair_kernel<MODE, SYNC>), dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, h->stream, h->d_shared,
                       h->d_base, h->d_snap, h->d_delta, h->d_recv, h->n_matrix, h->Kp, g.Kc, (uint32_t)h->I, (float)h->cfg.beta);
    HIPCHK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((cdae::delta_pipe_kernel<MODE, SYNC>), dim3((uint32_t)((g.threads + 255) / 256)), dim3(256), 0, h->stream, h->d_shared,
                     h->d_base, h->d_snap, h->d_delta, h->d_recv, h->n_matrix, h->Kp, g.Kc, g.n_tail);
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" {

int cdae_hip_train_users(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end, cdae_hip_stats* stats) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  HIPCHK(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  CHK(enqueue_users(h, seed, epoch, u_begin, u_end));
  CHK(join_aux(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  CHK(fused_check(h));
#ifdef CDAE_DECODE_TIMING
  {   // developer aid: cycle stamps of row `debug_rank` in the last decode launch (see decode_rows_kernel)
    unsigned long long t[64];
    HIPCHK(hipMemcpy(t, h->d_touched, sizeof t, hipMemcpyDeviceToHost));
    const int n = (int)(t[63] & 0xffffffffu);
    fprintf(stderr, "[decode timing] rank %u: %llu examples, %d stamps (cycles since first):", h->hp.debug_rank, t[63] >> 32, n);
    for (int i = 1; i < n; ++i) fprintf(stderr, " %llu", t[i] - t[0]);
    fprintf(stderr, "\n");
  }
#endif
  CHK(fill_stats(h, stats));
  if (stats) stats->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int cdae_hip_enqueue_users(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  HIPCHK(hipSetDevice(h->device));
  return enqueue_users(h, seed, epoch, u_begin, u_end);
}

int cdae_hip_prefetch_users(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  HIPCHK(hipSetDevice(h->device));
  std::vector<Batch> plan;
  CHK(make_plan(h, u_begin, u_end, plan));
  if (plan.empty()) return 0;
  // up to the look-ahead depth: with two prep lanes the second batch is prepared ahead as well
  const size_t n = std::min(plan.size(), prep_depth(h));
  const size_t have = prefetched_prefix(h, plan, seed, epoch);
  if (have >= n) return 0;
  CHK(drop_prefetched(h, have));
  for (size_t t = have; t < n; ++t) {
    const uint64_t q = h->seq + t;
    if (!(h->debug_skip_prep && q >= 2 * cdae_hip::NSETS)) CHK(submit_prep(h, set_of(q), plan[t], seed, epoch, prep_lane(h, q), q));
    h->pre[t] = cdae_hip::PreBatch{plan[t].s0, seed, plan[t].nb, plan[t].cidx, epoch};
    h->pre_n = (uint32_t)t + 1;
  }
  return 0;
}

int cdae_hip_collect_stats(cdae_hip_t* h, cdae_hip_stats* stats) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  return fill_stats(h, stats);
}

int cdae_hip_debug_sample_batch(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint32_t n_users,
                                uint32_t cidx, uint32_t* ex_item, uint64_t* ex_val, uint32_t* sorted_item,
                                uint64_t* sorted_val, uint32_t* seg_begin, uint32_t* seg_end, uint32_t* dup_of_pos,
                                uint32_t* dup_of_ex, uint64_t* n_examples) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  if (!n_examples) return fail("null argument");
  if (n_users == 0 || u_begin + n_users > h->U) return fail("bad user range [%llu, +%u)", (unsigned long long)u_begin, n_users);
  if (n_users > std::min<uint64_t>(h->B, h->U)) return fail("%u users exceed batch_users %u", n_users, h->B);
  if (cidx >= h->cfg.num_corruptions) return fail("corruption index %u out of range", cidx);
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  const uint64_t E = (uint64_t)(h->h_row_ptr[u_begin + n_users] - h->h_row_ptr[u_begin]) * h->ex_per_pos;
  if (E > h->Ecap) return fail("batch has %llu examples, capacity %llu", (unsigned long long)E, (unsigned long long)h->Ecap);
  if (E > *n_examples) return fail("batch has %llu examples, the caller's arrays hold %llu", (unsigned long long)E, (unsigned long long)*n_examples);
  *n_examples = E;
  HIPCHK(hipStreamSynchronize(h->stream));
  CHK(sync_prep(h));
  h->pre_n = 0;                                           // the set is overwritten: a prefetched batch is gone
  const int set = set_of(h->seq);
  const int prof = h->profiling;
  h->profiling = 0;
  h->prep_force_sort = true;
  const int rc = prep_batch(h, set, Batch{u_begin, n_users, cidx, E}, seed, epoch);
  h->prep_force_sort = false;
  h->profiling = prof;
  if (rc) return rc;
  CHK(sync_prep(h));
  cdae_hip::ExBuf& x = h->ex[set];
  const size_t I = (size_t)h->I;
  auto out = [&](void* dst, const void* src, size_t bytes) -> int {
    if (dst && bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
  };
  CHK(out(ex_item, x.item, E * sizeof(uint32_t)));
  CHK(out(ex_val, x.val, E * sizeof(uint64_t)));
  CHK(out(sorted_val, x.sorted_val, E * sizeof(uint64_t)));
  CHK(out(seg_begin, x.seg, I * sizeof(uint32_t)));
  CHK(out(seg_end, x.seg + I, I * sizeof(uint32_t)));
  CHK(out(dup_of_pos, x.dup_of_pos, E * sizeof(uint32_t)));
  CHK(out(dup_of_ex, x.dup_of_ex, E * sizeof(uint32_t)));
  if (sorted_item && E) {
    if (h->counting_sort || (h->bucket_sort && x.key16)) {   // the counting / bucket sorts write no sorted key array: the segments say the same
      std::vector<uint32_t> sb(I), se(I);
      HIPCHK(hipMemcpy(sb.data(), x.seg, I * sizeof(uint32_t), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(se.data(), x.seg + I, I * sizeof(uint32_t), hipMemcpyDeviceToHost));
      for (size_t it = 0; it < I; ++it)
        for (uint32_t q = sb[it]; q < se[it]; ++q) sorted_item[q] = (uint32_t)it;
    } else if (x.key16) {                                        // I <= 65536: the sort ran on 16-bit copies of the item ids
      std::vector<uint16_t> k16(E);
      HIPCHK(hipMemcpy(k16.data(), x.sorted_key16, E * sizeof(uint16_t), hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < E; ++i) sorted_item[i] = k16[i];
    } else {
      CHK(out(sorted_item, x.sorted_item, E * sizeof(uint32_t)));
    }
  }
  // dup_of_pos is only written at the positions segment_kernel numbered: report DUP_NONE everywhere else
  if (dup_of_pos && sorted_val) {
    for (uint64_t p = 0; p < E; ++p)
      if (!((uint32_t)sorted_val[p] & cdae::DUP_PREV_BIT)) dup_of_pos[p] = cdae::DUP_NONE;
  } else if (dup_of_pos) {
    std::vector<uint64_t> sv(E);
    if (E) HIPCHK(hipMemcpy(sv.data(), x.sorted_val, E * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (uint64_t p = 0; p < E; ++p)
      if (!((uint32_t)sv[p] & cdae::DUP_PREV_BIT)) dup_of_pos[p] = cdae::DUP_NONE;
  }
  HIPCHK(hipEventRecord(x.released, h->stream));          // nothing trains on this set: hand it back
  return 0;
}

int cdae_hip_train_epoch(cdae_hip_t* h, uint64_t seed, uint32_t epoch, cdae_hip_stats* stats) {
  if (!h) return fail("null handle");
  return cdae_hip_train_users(h, seed, epoch, 0, h->U, stats);
}

int cdae_hip_encode(cdae_hip_t* h, uint64_t seed, uint32_t epoch, int mode, const uint32_t* uids, size_t n, float* Z) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  if (h->mf) return fail("cdae_hip_encode does not apply to an IMF / BPR handle");
  if ((!uids || !Z) && n) return fail("null argument");
  if (mode != 0 && mode != 1) return fail("mode must be 0 or 1");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  const uint32_t B = (uint32_t)std::min<uint64_t>(h->B, h->U);
  for (size_t i = 0; i < n; ++i) if (uids[i] >= h->U) return fail("user id %u out of range", uids[i]);
  for (size_t c0 = 0; c0 < n; c0 += B) {
    const uint32_t nb = (uint32_t)std::min<size_t>(B, n - c0);
    HIPCHK(hipMemcpyAsync(h->d_uids, uids + c0, nb * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    std::vector<uint32_t> prefix(nb + 1, 0u);
    for (uint32_t i = 0; i < nb; ++i) prefix[i + 1] = prefix[i] + (h->h_unit_ptr[uids[c0 + i] + 1] - h->h_unit_ptr[uids[c0 + i]]);
    HIPCHK(hipMemcpyAsync(h->d_uptr_tmp, prefix.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));                     // `prefix` is a stack temporary
    CHK(encode_chunk(h, h->d_uids, 0, nb, mode, CDAE_STREAM_CORRUPT, 0, seed, epoch, prefix[nb]));
    HIPCHK(hipMemcpy2DAsync(Z + c0 * h->K, h->K * sizeof(float), h->d_Z, h->Kp * sizeof(float), h->K * sizeof(float), nb,
                            hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

int cdae_hip_data_loss(cdae_hip_t* h, uint64_t seed, uint32_t epoch, double* out) {
  if (!h || !h->d_shared || !out) return fail("bad argument");
  if (h->mf) { *out = 0.; return 0; }                      // IMF / BPR do not override ModelBase::data_loss (model_base.hpp:36-40)
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  HIPCHK(hipMemsetAsync(h->d_scalar, 0, sizeof(double), h->stream));
  for (uint64_t s0 = 0; s0 < h->U; s0 += EVAL_CHUNK) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(EVAL_CHUNK, h->U - s0);
    const uint32_t n_units = h->h_unit_ptr[s0 + nb] - h->h_unit_ptr[s0];
    CHK(ensure_eval_ws(h, nb, n_units));
    for (uint32_t c = 0; c < h->cfg.num_corruptions; ++c) {       // cdae.hpp:86
      CHK(encode_chunk(h, nullptr, s0, nb, 1, CDAE_STREAM_LOSS_CORRUPT, c, seed, epoch, 0, h->d_zeval, h->d_hpart_eval, h->eval_unit_cap));
      DISPATCH_NI(h->NI, cdae::data_loss_kernel, dim3((n_units + 3) / 4), dim3(256), 0, h->stream, h->hp, h->d_row_ptr, h->d_col,
                  h->d_unit_ptr + s0, n_units, (const uint32_t*)h->d_unit_user, s0, nb, h->d_zeval, h->dec(), h->P(CDAE_P_BP), h->d_scalar);
    }
  }
  HIPCHK(hipGetLastError());
  double v = 0;
  HIPCHK(hipMemcpyAsync(&v, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *out = v / (double)h->cfg.num_corruptions;                      // cdae.hpp:98
  return 0;
}

int cdae_hip_penalty_loss(cdae_hip_t* h, double* out) {
  if (!h || !h->d_shared || !out) return fail("bad argument");
  if (h->mf) { *out = 0.; return 0; }                      // ModelBase::penalty_loss default (model_base.hpp:43-45)
  double a = 0, b = 0;
  CHK(cdae_internal::shared_penalty(h, &a));
  CHK(cdae_internal::private_penalty(h, &b));
  *out = a + b;
  return 0;
}

int cdae_hip_recommend_all(cdae_hip_t* h, uint64_t u_begin, uint64_t u_end, uint32_t topk, uint32_t* out) {
  if (!h || !h->d_shared || (!out && !h->topn_active)) return fail("bad argument");
  if (u_begin > u_end || u_end > h->U) return fail("bad user range");
  if (topk == 0 || topk > h->I) return fail("topk must be in [1, num_items]");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  if (!h->user_perm.empty()) {
    // device rows are in training order: rank every position, hand the lists out by user id
    std::vector<uint32_t> perm, inv;
    perm.swap(h->user_perm); inv.swap(h->user_inv);          // (the recursive call sees an identity order)
    std::vector<uint32_t> all((size_t)h->U * topk);
    const int rc = cdae_hip_recommend_all(h, 0, h->U, topk, all.data());
    h->user_perm.swap(perm); h->user_inv.swap(inv);
    if (rc) return rc;
    for (uint64_t u = u_begin; u < u_end; ++u)
      std::copy(all.begin() + (size_t)h->user_inv[u] * topk, all.begin() + ((size_t)h->user_inv[u] + 1) * topk, out + (u - u_begin) * topk);
    return 0;
  }
  if (topk <= (uint32_t)cdae::REC_TOPK_MAX && h->K <= 256 && !h->recommend_per_user) {
    // matrix-core path: all users of a chunk in one launch (cdae_recommend_kernels.hpp)
    const uint32_t nch = h->K <= 32 ? 4 : (h->K <= 64 ? 8 : (h->K <= 128 ? 16 : (h->K <= 200 ? 25 : 32)));
    const uint32_t words = (uint32_t)((h->I + 31) / 32);
    const uint64_t n_all = u_end - u_begin;
    const uint32_t UC = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_all, 1), EVAL_CHUNK);
    if (h->bits_cap < (size_t)UC * words) {
      if (h->d_bits) HIPCHK(hipFree(h->d_bits));
      h->d_bits = nullptr; h->bits_cap = 0;
      CHK(dev_alloc(&h->d_bits, (size_t)UC * words));
      h->bits_cap = (size_t)UC * words;
    }
    if (h->rec_cap < (size_t)UC * topk) {
      if (h->d_rec) HIPCHK(hipFree(h->d_rec));
      h->d_rec = nullptr; h->rec_cap = 0;
      CHK(dev_alloc(&h->d_rec, (size_t)UC * topk));
      h->rec_cap = (size_t)UC * topk;
    }
    const size_t lds = cdae::recommend_mfma_lds_bytes((int)nch);
    for (uint64_t c0 = u_begin; c0 < u_end; c0 += UC) {
      const uint32_t nu = (uint32_t)std::min<uint64_t>(UC, u_end - c0);
      if (!h->mf) CHK(ensure_eval_ws(h, nu, h->h_unit_ptr[c0 + nu] - h->h_unit_ptr[c0]));
      hipLaunchKernelGGL(cdae::rated_bits_kernel, dim3((nu + 3) / 4), dim3(256), 0, h->stream, h->d_row_ptr, h->d_col, c0, nu, words, h->d_bits);
      const float* zsrc = h->d_zeval;
      if (h->mf) zsrc = h->d_Wu + (size_t)c0 * h->Kp;      // IMF / BPR: score = ub + ib + uv . iv (imf.hpp:117-119); ub does not rank
      else CHK(encode_chunk(h, nullptr, c0, nu, 0, CDAE_STREAM_CORRUPT, 0, 0, 0, 0, h->d_zeval, h->d_hpart_eval, h->eval_unit_cap));   // cdae.hpp:167-172, full rows
      const dim3 grid((nu + cdae::REC_USERS_PER_BLOCK - 1) / cdae::REC_USERS_PER_BLOCK);
#define REC_LAUNCH(NCH_)                                                                                                              \
  do {                                                                                                                                \
    HIPCHK(hipFuncSetAttribute((const void*)cdae::recommend_mfma_kernel<NCH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(cdae::recommend_mfma_kernel<NCH_>, grid, dim3(256), lds, h->stream, h->hp, zsrc, nu, h->dec(),                 \
                       h->P(CDAE_P_BP), h->d_bits, words, topk, h->d_rec);                                                           \
  } while (0)
      switch (nch) { case 4: REC_LAUNCH(4); break; case 8: REC_LAUNCH(8); break; case 16: REC_LAUNCH(16); break; case 25: REC_LAUNCH(25); break; default: REC_LAUNCH(32); break; }
#undef REC_LAUNCH
      HIPCHK(hipGetLastError());
      if (h->topn_active) { topn_chunk(h, topk, c0, nu); HIPCHK(hipGetLastError()); }
      if (out) {
        HIPCHK(hipMemcpyAsync(out + (c0 - u_begin) * topk, h->d_rec, (size_t)nu * topk * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
      }
    }
    return 0;
  }
  return recommend_general(h, u_begin, u_end, topk, out, nullptr, 0);
}

// The validation rows TOPN_Evaluation scores against (evaluation.hpp:118-120 builds them as a hashtable on every call): CSR over
// the handle's users, items ascending and unique inside a row; copied to the device once per data set.
int cdae_hip_set_test_rows(cdae_hip_t* h, const int64_t* test_row_ptr, const uint32_t* test_col) {
  if (!h || !h->d_shared || !test_row_ptr) return fail("bad argument (cdae_hip_set_interactions first)");
  HIPCHK(hipSetDevice(h->device));
  const uint64_t U = h->U;
  uint64_t with_rows = 0;
  CHK(cdae_internal::validate_test_rows(test_row_ptr, test_col, U, h->item_shard ? h->I_global : h->I, &with_rows));
  const uint64_t nnz = (uint64_t)test_row_ptr[U];
  CHK(quiesce(h));
  void** old[] = {(void**)&h->d_test_ptr, (void**)&h->d_test_col, (void**)&h->d_topn_pu, (void**)&h->d_topn_out};
  for (void** p : old) if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
  CHK(dev_alloc(&h->d_test_ptr, U + 1));
  CHK(dev_alloc(&h->d_test_col, std::max<uint64_t>(nnz, 1)));
  CHK(dev_alloc(&h->d_topn_pu, U * 8));
  CHK(dev_alloc(&h->d_topn_out, 16));
  HIPCHK(hipMemcpy(h->d_test_ptr, test_row_ptr, (U + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (nnz) HIPCHK(hipMemcpy(h->d_test_col, test_col, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
  h->test_users_with_rows = with_rows;
  return 0;
}

// TOPN_Evaluation::evaluate (evaluation.hpp:113-181): every user's top-`topk` list over the unrated items (cdae_hip_recommend_all)
// scored against the test rows on the device; rets = P@1 P@5 P@10 R@1 R@5 R@10 MAP@5 MAP@10 averaged over the users with test
// items, summed in user order (the bits of a sequential host loop); hits = summed hit counts in the first 1 / 5 / 10 places.
int cdae_hip_eval_topn(cdae_hip_t* h, uint32_t topk, double* rets8, uint64_t* hits3, uint32_t* ids_out) {
  if (!h || !h->d_shared || !rets8) return fail("bad argument");
  if (!h->d_test_ptr) return fail("cdae_hip_eval_topn: cdae_hip_set_test_rows first");
  if (h->item_shard) return fail("cdae_hip_eval_topn applies to a whole model (item shards: cdae_hip_multi_eval_topn)");
  if (h->test_users_with_rows == 0) return fail("no user has test items");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  if (!h->user_perm.empty()) {
    // IMF / BPR block schedules keep their rows in training order: take the lists by user id, score them from a staged copy
    std::vector<uint32_t> all((size_t)h->U * topk);
    CHK(cdae_hip_recommend_all(h, 0, h->U, topk, all.data()));
    if (ids_out) std::copy(all.begin(), all.end(), ids_out);
    const uint32_t UC = (uint32_t)std::min<uint64_t>(h->U, EVAL_CHUNK);
    if (h->rec_cap < (size_t)UC * topk) {
      if (h->d_rec) HIPCHK(hipFree(h->d_rec));
      h->d_rec = nullptr; h->rec_cap = 0;
      CHK(dev_alloc(&h->d_rec, (size_t)UC * topk));
      h->rec_cap = (size_t)UC * topk;
    }
    HIPCHK(hipMemsetAsync(h->d_topn_out, 0, 16 * sizeof(double), h->stream));
    for (uint64_t c0 = 0; c0 < h->U; c0 += UC) {
      const uint32_t nu = (uint32_t)std::min<uint64_t>(UC, h->U - c0);
      HIPCHK(hipMemcpyAsync(h->d_rec, all.data() + c0 * topk, (size_t)nu * topk * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
      topn_chunk(h, topk, c0, nu);
      HIPCHK(hipStreamSynchronize(h->stream));
    }
  } else {
    HIPCHK(hipMemsetAsync(h->d_topn_out, 0, 16 * sizeof(double), h->stream));
    h->topn_active = true;
    const int rc = cdae_hip_recommend_all(h, 0, h->U, topk, ids_out);
    h->topn_active = false;
    if (rc) return rc;
  }
  hipLaunchKernelGGL(cdae::topn_sum_kernel, dim3(1), dim3(64), 0, h->stream, (const double*)h->d_topn_pu, h->U, h->d_topn_out);
  HIPCHK(hipGetLastError());
  double host[16];
  HIPCHK(hipMemcpyAsync(host, h->d_topn_out, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int c = 0; c < 8; ++c) rets8[c] = host[c];
  if (hits3) std::memcpy(hits3, host + 8, 3 * sizeof(uint64_t));
  return 0;
}

int cdae_hip_recommend_user(cdae_hip_t* h, uint64_t uid, const uint32_t* rated_items, size_t n_rated, uint32_t topk, uint32_t* out) {
  if (!h || !h->d_shared || !out) return fail("bad argument");
  if (uid >= h->U) return fail("user id %llu out of range", (unsigned long long)uid);
  if (topk == 0 || topk > h->I) return fail("topk must be in [1, num_items]");
  if (n_rated && !rated_items) return fail("null argument");
  if (n_rated + topk > h->I) return fail("%zu rated items leave fewer than topk = %u candidates", n_rated, topk);
  if (h->mf) return fail("cdae_hip_recommend_user does not apply to an IMF / BPR handle (its score does not depend on the rated set)");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  std::vector<uint32_t> rated(rated_items, rated_items + n_rated);
  std::sort(rated.begin(), rated.end());                         // encode sums in ascending item order like the train row
  for (size_t i = 0; i < n_rated; ++i) {
    if (rated[i] >= h->I) return fail("rated item %u out of range", rated[i]);
    if (i && rated[i] == rated[i - 1]) return fail("duplicate rated item %u", rated[i]);
  }
  return recommend_general(h, uid, uid + 1, topk, out, rated.data(), (uint32_t)n_rated);
}

int cdae_hip_train_one_user_corruption(cdae_hip_t* h, uint64_t uid, const uint32_t* input_items, size_t n_in,
                                       const uint32_t* negative_items, size_t n_neg) {
  if (h) h->db_valid = h->db_rows_valid = false;      // the bf16 images of the decoder no longer match it (full-output path, compute_batch_full)
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  if (uid >= h->U) return fail("user id %llu out of range", (unsigned long long)uid);
  if (h->cfg.full_output) return fail("train_one_user_corruption takes an explicit negative list; it is not available in full_output mode");
  if (h->mf) return fail("train_one_user_corruption does not apply to an IMF / BPR handle");
  if ((n_in && !input_items) || (n_neg && !negative_items)) return fail("null argument");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  const int64_t r0 = h->h_row_ptr[uid];
  const size_t n_pos = (size_t)(h->h_row_ptr[uid + 1] - r0);
  const size_t E = n_pos + n_neg;
  if (E > h->Ecap) return fail("%zu examples exceed the batch capacity %llu", E, (unsigned long long)h->Ecap);
  std::vector<uint32_t> pos(n_pos), items(E), in_sorted(input_items, input_items + n_in);
  HIPCHK(hipMemcpy(pos.data(), h->d_col + r0, n_pos * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::sort(in_sorted.begin(), in_sorted.end());
  for (size_t i = 0; i < n_in; ++i) {
    if (!std::binary_search(pos.begin(), pos.end(), in_sorted[i])) return fail("input item %u is not a training item of user %llu", in_sorted[i], (unsigned long long)uid);
    if (i && in_sorted[i] == in_sorted[i - 1]) return fail("duplicate input item %u", in_sorted[i]);
  }
  std::vector<uint64_t> vals(E);
  for (size_t p = 0; p < n_pos; ++p) {
    const bool kept = std::binary_search(in_sorted.begin(), in_sorted.end(), pos[p]);
    items[p] = pos[p];
    vals[p] = ((uint64_t)p << 32) | (uint64_t)(cdae::TARGET_BIT | (kept ? cdae::INPUT_BIT : 0u));    // slot 0
  }
  for (size_t i = 0; i < n_neg; ++i) {
    if (negative_items[i] >= h->I) return fail("negative item %u out of range", negative_items[i]);
    if (std::binary_search(pos.begin(), pos.end(), negative_items[i])) return fail("negative item %u is a training item", negative_items[i]);
    items[n_pos + i] = negative_items[i];
    vals[n_pos + i] = (uint64_t)(n_pos + i) << 32;
  }
  h->pre_n = 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  CHK(sync_prep(h));
  const int set = set_of(h->seq);
  cdae_hip::ExBuf& x = h->ex[set];
  HIPCHK(hipMemcpyAsync(x.item, items.data(), E * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(x.val, vals.data(), E * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(rocprim::radix_sort_pairs(h->d_sort_tmp, h->sort_tmp_bytes, x.item, x.sorted_item, x.val, x.sorted_val, E, 0u,
                                   (unsigned)h->sort_bits, h->stream));
  HIPCHK(hipMemsetAsync(x.seg, 0, 4 * (size_t)h->I * sizeof(uint32_t), h->stream));
  HIPCHK(hipMemsetAsync(x.dup_count, 0, cdae::DUP_STRIPES * sizeof(uint32_t), h->stream));
  HIPCHK(hipMemsetAsync(x.dup_of_ex, 0xFF, E * sizeof(uint32_t), h->stream));
  hipLaunchKernelGGL(cdae::segment_kernel<uint32_t>, dim3((uint32_t)((E + 256 * cdae::SEG_PER_THREAD - 1) / (256 * cdae::SEG_PER_THREAD))),
                     dim3(256), 0, h->stream, x.sorted_item, x.sorted_val,
                     (uint32_t)E, x.seg, x.seg + h->I, x.dup_count, h->dup_cap, x.dup_of_pos, x.dup_of_ex, h->dup_stripes,
                     (const uint32_t*)h->d_rank_of, x.seg + 2 * (size_t)h->I, x.seg + 3 * (size_t)h->I);
  HIPCHK(hipEventRecord(x.ready, h->stream));
  const uint32_t one_unit[2] = {0u, 1u};                         // one user, one unit
  HIPCHK(hipMemcpyAsync(h->d_uptr_tmp, one_unit, sizeof one_unit, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  uint32_t* d_in = h->d_uids;                     // capacity min(B, U) >= 1; longer input sets get their own buffer
  uint32_t* d_in_owned = nullptr;
  if (n_in > std::min<uint64_t>(h->B, h->U)) { CHK(dev_alloc(&d_in_owned, n_in)); d_in = d_in_owned; }
  if (n_in) HIPCHK(hipMemcpyAsync(d_in, in_sorted.data(), n_in * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  h->prof_q = h->seq;
  int rc = compute_batch(h, set, Batch{uid, 1, 0, E}, 0, 0, d_in, (uint32_t)n_in);
  h->seq++;
  hipError_t se = hipStreamSynchronize(h->stream);
  if (d_in_owned) (void)hipFree(d_in_owned);
  if (rc) return rc;
  if (se != hipSuccess) return fail("stream synchronize failed: %s", hipGetErrorString(se));
  return 0;
}

// ---- data-parallel exchange ---------------------------------------------------------------------
int cdae_hip_delta_begin(cdae_hip_t* h) {
  if (!h || !h->d_shared) return fail("set_interactions must be called first");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  if (!h->d_base) { CHK(dev_alloc(&h->d_base, h->n_shared)); CHK(dev_alloc(&h->d_delta, h->n_shared + h->I)); }
  HIPCHK(hipMemcpyAsync(h->d_base, h->d_shared, h->n_shared * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->d_touched, 0, h->I * sizeof(uint32_t), h->stream));
  return 0;
}

int cdae_hip_stream(cdae_hip_t* h, void** hip_stream) {
  if (!h || !hip_stream) return fail("bad argument");
  *hip_stream = (void*)h->stream;
  return 0;
}

int cdae_hip_delta_compute(cdae_hip_t* h) {
  if (!h || !h->d_base) return fail("delta_begin must be called first");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  hipLaunchKernelGGL(cdae::delta_kernel, dim3((uint32_t)((h->n_shared + 255) / 256)), dim3(256), 0, h->stream, h->d_shared,
                     h->d_base, h->d_delta, h->n_shared);
  hipLaunchKernelGGL(cdae::touch_to_float_kernel, dim3((uint32_t)((h->I + 255) / 256)), dim3(256), 0, h->stream, h->d_touched,
                     h->d_delta + h->n_shared, (uint32_t)h->I);
  HIPCHK(hipGetLastError());
  return 0;                                  // stream-ordered: callers that read the buffer on another stream synchronise
}

int cdae_hip_delta_device_ptr(cdae_hip_t* h, void** device_ptr, size_t* count_floats) {
  if (!h || !h->d_delta || !device_ptr) return fail("delta_begin must be called first");
  *device_ptr = h->d_delta;
  if (count_floats) *count_floats = h->n_shared + h->I;
  return 0;
}

int cdae_hip_delta_apply(cdae_hip_t* h, uint32_t world_size, uint32_t rule) {
  if (h) h->db_valid = h->db_rows_valid = false;      // the bf16 images of the decoder no longer match it (full-output path, compute_batch_full)
  if (!h || !h->d_delta) return fail("delta_begin must be called first");
  if (world_size == 0) return fail("world_size must be >= 1");
  if (rule > CDAE_DELTA_TOUCH_MEAN) return fail("unknown delta rule %u", rule);
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  hipLaunchKernelGGL(cdae::apply_delta_kernel, dim3((uint32_t)((h->n_shared + 255) / 256)), dim3(256), 0, h->stream, h->d_shared,
                     h->d_base, h->d_delta, h->d_delta + h->n_shared, h->n_matrix, h->Kp, (uint32_t)h->I, h->n_shared, world_size, rule);
  HIPCHK(hipGetLastError());
  return 0;
}

int cdae_hip_delta_set_combine(cdae_hip_t* h, uint32_t combine) {
  if (!h) return fail("null handle");
  if (combine > CDAE_COMBINE_GLOBAL_ACC) return fail("unknown combine rule %u", combine);
  h->delta_combine = combine;      // (takes effect at the next stage: call it between a merge and the following stage)
  return 0;
}

// pipelined variant: see delta_pipe_kernel.  The staged / received buffers are compact (no pad columns).
int cdae_hip_delta_stage(cdae_hip_t* h) {
  if (!h || !h->d_base) return fail("delta_begin must be called first");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  if (!h->d_recv) {
    CHK(dev_alloc(&h->d_recv, h->n_shared + 4096));            // slack: collectives may round the count up
    CHK(dev_alloc(&h->d_snap, h->n_shared));
    HIPCHK(hipMemsetAsync(h->d_recv, 0, (h->n_shared + 4096) * sizeof(float), h->stream));
  }
  return launch_pipe<cdae::DELTA_STAGE>(h);
}

int cdae_hip_delta_recv_device_ptr(cdae_hip_t* h, void** device_ptr, size_t* count_floats) {
  if (!h || !h->d_recv || !device_ptr) return fail("delta_stage must be called first");
  *device_ptr = h->d_recv;
  if (count_floats) *count_floats = pipe_geom(h).n_compact;
  return 0;
}

int cdae_hip_delta_merge(cdae_hip_t* h) {
  if (h) h->db_valid = h->db_rows_valid = false;      // the bf16 images of the decoder no longer match it (full-output path, compute_batch_full)
  if (!h || !h->d_recv) return fail("delta_stage must be called first");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  return launch_pipe<cdae::DELTA_MERGE>(h);
}

int cdae_hip_delta_merge_stage(cdae_hip_t* h) {
  if (h) h->db_valid = h->db_rows_valid = false;      // the bf16 images of the decoder no longer match it (full-output path, compute_batch_full)
  if (!h || !h->d_recv) return fail("delta_stage must be called first");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  return launch_pipe<cdae::DELTA_MERGE_STAGE>(h);
}

}  // extern "C"

// ---- accessor surface for cdae_multi.hip (cdae_internal.hpp) --------------------------------------------------------
namespace cdae_internal {

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
int device_of(const cdae_hip_t* h) { return h->device; }
hipStream_t main_stream(cdae_hip_t* h) { return h->stream; }
hipStream_t aux_stream(cdae_hip_t* h) { return h->aux; }
uint64_t num_users(const cdae_hip_t* h) { return h->U; }
uint64_t num_items(const cdae_hip_t* h) { return h->I; }
uint32_t batch_users(const cdae_hip_t* h) { return (uint32_t)std::min<uint64_t>(h->B, h->U); }
bool ready(const cdae_hip_t* h) { return h->d_shared != nullptr; }
float* send_buf(cdae_hip_t* h) { return h->d_delta; }
float* recv_buf(cdae_hip_t* h) { return h->d_recv; }
size_t compact_count(const cdae_hip_t* h) { return pipe_geom(h).n_compact; }
int validate_test_rows(const int64_t* test_row_ptr, const uint32_t* test_col, uint64_t U, uint64_t I, uint64_t* users_with_rows) {
  if (!test_row_ptr) return fail("null test_row_ptr");
  if (test_row_ptr[0] != 0) return fail("test_row_ptr[0] must be 0");
  for (uint64_t u = 0; u < U; ++u)
    if (test_row_ptr[u + 1] < test_row_ptr[u]) return fail("test_row_ptr must be non-decreasing");
  if (test_row_ptr[U] > 0 && !test_col) return fail("null test_col with %lld test interactions", (long long)test_row_ptr[U]);
  uint64_t with_rows = 0;
  for (uint64_t u = 0; u < U; ++u) {
    with_rows += test_row_ptr[u + 1] > test_row_ptr[u];
    for (int64_t p = test_row_ptr[u]; p < test_row_ptr[u + 1]; ++p) {
      if (test_col[p] >= I) return fail("test item id %u of user %llu out of range", test_col[p], (unsigned long long)u);
      if (p > test_row_ptr[u] && test_col[p] <= test_col[p - 1]) return fail("test row %llu is not ascending and unique", (unsigned long long)u);
    }
  }
  if (users_with_rows) *users_with_rows = with_rows;
  return 0;
}
int delta_stage_sync(cdae_hip_t* h) {
  if (!h || !h->d_recv) return fail("delta_stage must have been called once (it allocates the exchange buffers)");
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  return launch_pipe<cdae::DELTA_STAGE, true>(h);
}
int delta_merge_sync(cdae_hip_t* h) {
  if (!h || !h->d_recv) return fail("delta_stage must be called first");
  h->db_valid = h->db_rows_valid = false;
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  return launch_pipe<cdae::DELTA_MERGE, true>(h);
}

int adopt_shared_block(cdae_hip_t* dst, cdae_hip_t* src) {
  if (!dst || !src || !dst->d_shared || !src->d_shared || dst->n_shared != src->n_shared) return fail("adopt_shared_block: the handles do not hold the same model");
  CHK(cdae_hip_synchronize(src));
  HIPCHK(hipSetDevice(dst->device));
  CHK(quiesce(dst));
  dst->db_valid = dst->db_rows_valid = false;      // (full-output path) the bf16 images of the decoder no longer match it
  if (dst->device == src->device)
    HIPCHK(hipMemcpyAsync(dst->d_shared, src->d_shared, dst->n_shared * sizeof(float), hipMemcpyDeviceToDevice, dst->stream));
  else
    HIPCHK(hipMemcpyPeerAsync(dst->d_shared, dst->device, src->d_shared, src->device, dst->n_shared * sizeof(float), dst->stream));
  return 0;
}
void*& exchange_slot(cdae_hip_t* h) { return h->xchg; }
void set_exchange_deleter(cdae_hip_t* h, void (*deleter)(void*)) { h->xchg_free = deleter; }

static int sqnorm_sum(cdae_hip* h, std::initializer_list<std::pair<const float*, size_t>> parts, double* out) {
  HIPCHK(hipSetDevice(h->device));
  CHK(join_aux(h));
  HIPCHK(hipMemsetAsync(h->d_scalar, 0, sizeof(double), h->stream));
  for (auto& pr : parts)
    if (pr.first && pr.second)
      hipLaunchKernelGGL(cdae::sqnorm_kernel, dim3((uint32_t)std::min<size_t>(2048, (pr.second + 255) / 256)), dim3(256), 0, h->stream,
                         pr.first, pr.second, h->d_scalar);
  HIPCHK(hipGetLastError());
  double v = 0;
  HIPCHK(hipMemcpyAsync(&v, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *out = 0.5 * h->cfg.lambda * v;
  return 0;
}
// cdae.hpp:104-106: W, V, Wu, b, b_prime (pad lanes are zero)
int shared_penalty(cdae_hip_t* h, double* out) {
  return sqnorm_sum(h, {{h->P(CDAE_P_W), h->cnt[CDAE_P_W]}, {h->P(CDAE_P_V), h->cnt[CDAE_P_V]}, {h->P(CDAE_P_B), h->cnt[CDAE_P_B]},
                        {h->P(CDAE_P_BP), h->cnt[CDAE_P_BP]}}, out);
}
int item_rows_penalty(cdae_hip_t* h, double* out) {
  return sqnorm_sum(h, {{h->P(CDAE_P_W), h->cnt[CDAE_P_W]}, {h->P(CDAE_P_V), h->cnt[CDAE_P_V]}, {h->P(CDAE_P_BP), h->cnt[CDAE_P_BP]}}, out);
}
int hidden_bias_penalty(cdae_hip_t* h, double* out) { return sqnorm_sum(h, {{h->P(CDAE_P_B), h->cnt[CDAE_P_B]}}, out); }
int private_penalty(cdae_hip_t* h, double* out) {
  if (!h->cfg.user_factor) { *out = 0.; return 0; }
  return sqnorm_sum(h, {{h->d_Wu, h->cnt[CDAE_P_WU]}}, out);
}

// ---- item-sharded layout: phases of a full-output batch and of the evaluation passes (cdae_internal.hpp) ---------------
int set_item_shard(cdae_hip_t* h, uint64_t item0, uint64_t num_items_global) {
  if (!h) return fail("null handle");
  if (h->mf) return fail("the item-sharded layout is CDAE's");
  h->item_shard = true; h->item0 = item0; h->I_global = num_items_global;
  return 0;
}
int set_item_shard_owner(cdae_hip_t* h, uint64_t u_begin, uint64_t u_end) {
  if (!h || !h->item_shard) return fail("set_item_shard must be called first");
  if (u_begin > u_end) return fail("bad owned user range");
  h->own_u0 = u_begin; h->own_u1 = u_end;
  return 0;
}
int set_item_shard_global(cdae_hip_t* h, const int64_t* row_ptr, const uint32_t* col) {
  if (!h || !h->item_shard) return fail("set_item_shard must be called first");
  h->g_row_ptr_src = row_ptr; h->g_col_src = col;
  return 0;
}
uint32_t shard_blocks(const cdae_hip_t* h) { return shard_blocks_of(h); }
int set_item_shard_positions(cdae_hip_t* h, const uint32_t* len_and_first /* [2 U] */) {
  if (!h || !h->d_shared || !h->item_shard || !len_and_first) return fail("set_item_shard and set_interactions must be called first");
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_gpos) CHK(dev_alloc(&h->d_gpos, 2 * (size_t)h->U));
  HIPCHK(hipMemcpy(h->d_gpos, len_and_first, 2 * (size_t)h->U * sizeof(uint32_t), hipMemcpyHostToDevice));
  return 0;
}
uint32_t row_stride(const cdae_hip_t* h) { return h->Kp; }
float* hsum_buf(cdae_hip_t* h) { return h->d_Hsum; }
float* hg_buf(cdae_hip_t* h) { return h->d_HG; }
float* ev_hsum_buf(cdae_hip_t* h) { return h->d_hsum_eval; }
uint32_t eval_chunk() { return EVAL_CHUNK; }

// block 0 = the users' input sums over this shard's rows; blocks 1.. = the Wu / Uu rows of the users this shard owns (zeros for everybody
// else's: after the all-reduce(sum) every shard holds the owners' rows, bit for bit) — one launch (unit_sum_stage_kernel)
static int sum_and_stage(cdae_hip* h, const float* hpart, const uint32_t* uptr, uint64_t u0, uint32_t n, float* buf) {
  const float* ta = h->cfg.user_factor ? h->d_Wu : (h->cfg.linear_function ? h->d_Uu : nullptr);
  const float* tb = h->cfg.user_factor && h->cfg.linear_function ? h->d_Uu : nullptr;
  DISPATCH_NI(h->NI, cdae::unit_sum_stage_kernel, dim3((n + 3) / 4), dim3(256), 0, h->stream, h->hp, hpart, uptr, n, u0, ta, tb, buf);
  return 0;
}
// the gathered rows inside an all-reduced buffer (nullptr when the configuration has none)
static const float* gathered_wu(const cdae_hip* h, const float* buf, uint32_t n) { return h->cfg.user_factor ? buf + (size_t)n * h->Kp : nullptr; }
static const float* gathered_uu(const cdae_hip* h, const float* buf, uint32_t n) {
  return h->cfg.linear_function ? buf + (size_t)(h->cfg.user_factor ? 2 : 1) * n * h->Kp : nullptr;
}

static int fs_batch(cdae_hip* h, uint64_t s0, uint32_t nb, uint32_t cidx, Batch* bt) {
  if (!h->d_shared || !h->item_shard) return fail("not an item-sharded handle with data");
  if (nb == 0 || s0 + nb > h->U || nb > std::min<uint64_t>(h->B, h->U)) return fail("bad batch [%llu, +%u)", (unsigned long long)s0, nb);
  const uint64_t E = h->shard_sampled() ? (uint64_t)(h->h_grow_ptr[s0 + nb] - h->h_grow_ptr[s0]) * h->ex_per_pos
                                        : (uint64_t)(h->h_row_ptr[s0 + nb] - h->h_row_ptr[s0]);
  if (E > h->Ecap) return fail("batch has %llu examples, capacity %llu", (unsigned long long)E, (unsigned long long)h->Ecap);
  *bt = Batch{s0, nb, cidx, E};
  return 0;
}

int fs_prep(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t s0, uint32_t nb, uint32_t cidx) {
  HIPCHK(hipSetDevice(h->device));
  Batch bt;
  CHK(fs_batch(h, s0, nb, cidx, &bt));
  h->prof_q = h->seq;
  CHK(prep_batch(h, (int)(h->fs_prepped & 1), bt, seed, epoch));       // fs_prepped counts prepared batches: set = parity
  h->fs_prepped++;
  return 0;
}

int fs_phase0(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t s0, uint32_t nb, uint32_t cidx) {
  HIPCHK(hipSetDevice(h->device));
  Batch bt;
  CHK(fs_batch(h, s0, nb, cidx, &bt));
  const uint32_t n_units = units_of(h, bt);
  const uint32_t* uptr = h->d_unit_ptr + s0;
  if (!h->encode_two_launches && nb <= h->encode_users_max) {
    // one launch (round 4): the training encode's workgroup-per-user kernel, stopped at the raw input sum of the local rows, with the
    // owner's private rows staged behind it — the same sums in the same order as the single handle's encode for every user
    const float* ta = h->cfg.user_factor ? h->d_Wu : (h->cfg.linear_function ? h->d_Uu : nullptr);
    const float* tb = h->cfg.user_factor && h->cfg.linear_function ? h->d_Uu : nullptr;
    DISPATCH_NI(h->NI, cdae::encode_users_kernel, dim3(nb), dim3(cdae::ENC_WAVES * cdae::WAVE), 0, h->stream, h->hp, h->d_row_ptr, h->d_col,
                h->P(CDAE_P_W), uptr, s0, nb, cidx, seed, epoch, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                (float*)nullptr, (const float*)nullptr, (float*)nullptr, (cdae::BF16_T*)nullptr, (cdae::BF16_T*)nullptr, 0u, h->d_Hsum, ta, tb,
                (const uint32_t*)h->d_gpos);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (n_units)
    DISPATCH_NI(h->NI, cdae::encode_partial_kernel, dim3((n_units + 3) / 4), dim3(256), 0, h->stream, h->hp, h->d_row_ptr, h->d_col,
                h->P(CDAE_P_W), uptr, n_units, (const uint32_t*)nullptr, s0, nb, 1, CDAE_STREAM_CORRUPT, cidx, seed, epoch, h->d_Hpart,
                (const uint32_t*)nullptr, 0u, (const uint32_t*)h->d_unit_user, (const uint32_t*)h->d_gpos);
  CHK(sum_and_stage(h, h->d_Hpart, uptr, s0, nb, h->d_Hsum));
  HIPCHK(hipGetLastError());
  return 0;
}

int fs_phase1(cdae_hip_t* h, uint64_t s0, uint32_t nb) {
  using namespace cdae;
  HIPCHK(hipSetDevice(h->device));
  Batch bt;
  CHK(fs_batch(h, s0, nb, 0, &bt));
  const int b = (int)(h->seq & 1);
  cdae_hip::ExBuf& x = h->ex[b];
  hipStream_t st = h->stream;
  const uint32_t I = (uint32_t)h->I, Kp = h->Kp, Bp = h->Bp, Ip = h->Ip;
  const dim3 blk(256), grid_users((nb + 3) / 4);
  // the batch's Wu / Uu rows arrived with the all-reduced input sums (stage_own_rows): encode_finish reads them by slot (uids = iota)
  const float* wu_b = gathered_wu(h, h->d_Hsum, nb);
  const float* uu_b = gathered_uu(h, h->d_Hsum, nb);
  if (!h->cfg.full_output) {
    // ---- sampled decode over the local item rows: the single-GPU step's kernels on this shard's rows / examples ----
    DISPATCH_NI(h->NI, encode_finish_kernel, grid_users, blk, 0, st, h->hp, h->d_Hsum, (const uint32_t*)h->d_iota, wu_b, h->P(CDAE_P_B),
                (const uint32_t*)h->d_iota, s0, nb, 1, h->d_Z, h->d_Dz, h->d_HG, uu_b, h->d_Ssum,
                (BF16_T*)nullptr, (BF16_T*)nullptr, 0u, h->late_rows ? h->d_Ghot : (float*)nullptr, h->d_hotdup);
    HIPCHK(hipStreamWaitEvent(st, x.ready, 0));
    Prof pr;                                                  // (cdae_hip_set_profiling on a shard's handle: the decode launch of this shard's rows)
    CHK(pr.begin(h, F_DECODE, st, h->seq));
    CHK(launch_decode(h, x));
    CHK(pr.end());
    // local hidden gradient: the user's examples on THIS shard's rows (the others are VOID), partial rows per unit of the whole rows
    const uint32_t n_gunits = gunits_of(h, bt), halves = h->gather_halves;
    const uint32_t* guptr = h->d_gunit_ptr + s0;
    if (n_gunits)
      DISPATCH_NI(h->NI, hidden_gather_kernel, dim3(8 * halves * ((n_gunits + 3) / 4)), blk, 0, st, h->hp, h->d_grow_ptr, guptr, n_gunits, s0, nb,
                  x.item, h->d_G, h->d_D0, h->d_HGpart, 0u, x.dup_of_ex, h->d_dup_corr, (const uint32_t*)h->d_gunit_user, halves,
                  h->late_rows ? (const uint32_t*)h->d_late_bits : (const uint32_t*)nullptr, h->late_words);
    DISPATCH_NI(h->NI, hg_raw_kernel, dim3(nb), blk, 0, st, h->hp, guptr, n_gunits, nb, h->d_HGpart, 8u * halves, h->d_HG, h->late_finish());
    HIPCHK(hipGetLastError());
    return 0;
  }
  // bf16 images of the shard's decoder rows: the fused row step (gemm3_rows_fused_kernel) leaves the row-major one current, and with
  // GEMM 2 reading G^T and D nothing needs D^T — converted only when something else wrote the parameters
  if (!(rows_fused_path(h) && gemm2_tn_path(h) && h->db_rows_valid))
    hipLaunchKernelGGL(to_bf16_transpose_kernel, dim3(Kp / 64, Ip / 64), blk, 0, st, h->dec(), I, Kp, Kp, Ip, h->d_Db, h->d_DTb);
  CHK(join_aux(h));
  // z from the ALL-REDUCED input sums: encode_finish with one "unit" per user (identity prefix)
  DISPATCH_NI(h->NI, encode_finish_kernel, grid_users, blk, 0, st, h->hp, h->d_Hsum, (const uint32_t*)h->d_iota, wu_b, h->P(CDAE_P_B),
              (const uint32_t*)h->d_iota, s0, nb, 1, h->d_Z, h->d_Dz, h->d_HG, uu_b, h->d_Ssum);
  hipLaunchKernelGGL(to_bf16_transpose_kernel, dim3(Kp / 64, Bp / 64), blk, 0, st, h->d_Z, nb, Kp, Kp, Bp, h->d_Zb, h->d_ZTb);
  HIPCHK(hipStreamWaitEvent(st, x.ready, 0));
  uint32_t parts = 0, rows = nb;
  if (Kp <= 256 && !h->full_unfused) {
    const uint32_t words = (I + 31) / 32, slices = h->full_slices, tiles = Ip / (32 * FUSED_SUB);
    const uint32_t* bits = h->d_bits_train + (size_t)b * h->bits_stride;
    const uint32_t tps = (tiles + slices - 1) / slices;
    const dim3 grid(slices, Bp / 128);
    const size_t lds = full_fused_lds_bytes(Kp);
    if ((uint64_t)Ip * Bp > 0xFFFFFFFFull) return fail("full-output decode: G^T of %u x %u exceeds 2^32 elements; lower batch_users", Ip, Bp);
#define FS_FUSED2(NKS_, L_)                                                                                                              \
  do {                                                                                                                                  \
    HIPCHK(hipFuncSetAttribute((const void*)full_decode_fused_kernel<NKS_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
    hipLaunchKernelGGL((full_decode_fused_kernel<NKS_, L_>), grid, blk, lds, st, h->hp, h->d_Zb, h->d_Db, h->d_DTb, Ip, h->P(CDAE_P_BP), \
                       bits, words, nb, tps, h->d_GTb, Bp, h->d_HGpart);                                                                 \
  } while (0)
#define FS_FUSED(NKS_) do { if (h->cfg.loss_type == CDAE_LOSS_CROSS_ENTROPY) FS_FUSED2(NKS_, 5); else FS_FUSED2(NKS_, 0); } while (0)
    switch (Kp) { case 64: FS_FUSED(4); break; case 128: FS_FUSED(8); break; default: FS_FUSED(16); break; }
#undef FS_FUSED
#undef FS_FUSED2
    parts = slices; rows = nb;
  } else {
    Prof prd;
    CHK(prd.begin(h, F_DECODE, st, h->seq));
    CHK(full_products_k512(h, st, x, bt, nb, &parts, &rows));      // the single handle's launches over this shard's item rows
    CHK(prd.end());
  }
  // local hidden gradient of the batch, raw: the shards' sums are all-reduced before delta is formed
  DISPATCH_NI(h->NI, slab_sum_kernel, grid_users, blk, 0, st, h->hp, h->d_HGpart, parts, rows, nb, h->d_HG);
  HIPCHK(hipGetLastError());
  return 0;
}

int fs_phase2(cdae_hip_t* h, uint64_t s0, uint32_t nb) {
  using namespace cdae;
  HIPCHK(hipSetDevice(h->device));
  Batch bt;
  CHK(fs_batch(h, s0, nb, 0, &bt));
  const int b = (int)(h->seq & 1);
  cdae_hip::ExBuf& x = h->ex[b];
  hipStream_t st = h->stream;
  const uint32_t I = (uint32_t)h->I, Kp = h->Kp, Bp = h->Bp, Ip = h->Ip;
  const dim3 blk(256), grid_users((nb + 3) / 4);
  const float* uu_b = gathered_uu(h, h->d_Hsum, nb);
  if (!h->cfg.full_output) {
    // ---- sampled decode: delta from the all-reduced hg, the Wu / Uu steps of the users this shard owns, then the local input rows
    // and (replicated, identical everywhere) the b recurrence — the tail of the single-GPU step ----
    DISPATCH_NI(h->NI, hidden_finish_kernel, dim3(nb), blk, 0, st, h->hp, (const uint32_t*)h->d_iota, nb, s0, nb, h->d_HGpart, h->d_Dz,
                h->d_HG, h->d_Wu, h->d_Wu_ag, 0u, h->d_Uu, h->d_Uu_ag, h->d_Ssum, h->d_delta_rows, uu_b);
    const uint32_t bias_blocks = (h->Kp + 255u) / 256u;
    DISPATCH_NI(h->NI, input_rows_kernel, dim3(bias_blocks + (I + 3) / 4), blk, 0, st, h->hp, h->d_item_order, x.seg + 2 * (size_t)I, x.seg + 3 * (size_t)I,
                x.sorted_val, h->d_Z, h->d_HG, h->d_G, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), CDAE_TOUCHED_ARG, nb, h->P(CDAE_P_B),
                h->P(CDAE_P_B_AG), h->delta_rows());
    HIPCHK(hipEventRecord(x.released, st));
    HIPCHK(hipGetLastError());
    h->seq++;
    h->acc_examples += bt.E; h->acc_batches++; h->acc_users += nb;
    return 0;
  }
  HIPCHK(hipEventRecord(h->ev_fork, st));
  HIPCHK(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
  // d_HG holds the all-reduced hg: delta, the Wu steps of the users this shard owns, then the b recurrence (replicated)
  DISPATCH_NI(h->NI, hidden_finish_kernel, dim3(nb), blk, 0, h->aux, h->hp, (const uint32_t*)h->d_iota, nb, s0, nb, h->d_HGpart, h->d_Dz,
              h->d_HG, h->d_Wu, h->d_Wu_ag, 0u, h->d_Uu, h->d_Uu_ag, h->d_Ssum, h->d_delta_rows, uu_b);
  HIPCHK(hipEventRecord(h->ev_delta, h->aux));
  hipLaunchKernelGGL(hidden_bias_kernel, dim3((Kp + 255u) / 256u), blk, 0, h->aux, h->hp, nb, h->d_HG, h->P(CDAE_P_B), h->P(CDAE_P_B_AG));
  HIPCHK(hipEventRecord(h->ev_join, h->aux));
  const bool rows_fused = rows_fused_path(h);
  if (!rows_fused) {
    GemmEpilogue e3{};
    e3.Cout = h->d_dD; e3.ldc = Kp;
    CHK(launch_gemm_lds<EPI_STORE>(h, st, h->d_GTb, h->d_ZTb, Ip, Kp, Bp, Bp, Bp, Bp, e3, 1, 1));
  }
  HIPCHK(hipStreamWaitEvent(st, h->ev_delta, 0));
  Prof pri;
  CHK(pri.begin(h, F_INPUT, st, h->seq));
  if (rows_fused)      // dD = G^T Z and the row steps from its accumulators, the row-major bf16 image left current
    CHK(launch_rows_fused(h, st, x, nb, h->d_Db));
  else if (I >= 32768u)
    DISPATCH_NI(h->NI, full_rows_wave_kernel, dim3((I + 3) / 4), blk, 0, st, h->hp, x.seg, x.seg + I, x.sorted_val, h->delta_rows(),
                h->d_dD, h->d_GTb, Bp, nb, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_V), h->P(CDAE_P_V_AG), h->P(CDAE_P_BP),
                h->P(CDAE_P_BP_AG), h->d_touched);
  else
    DISPATCH_NI(h->NI, full_rows_kernel, dim3(I), blk, 0, st, h->hp, x.seg, x.seg + I, x.sorted_val, h->delta_rows(),
                h->d_dD, h->d_GTb, Bp, nb, h->P(CDAE_P_W), h->P(CDAE_P_W_AG), h->P(CDAE_P_V), h->P(CDAE_P_V_AG), h->P(CDAE_P_BP),
                h->P(CDAE_P_BP_AG), (float*)nullptr, (float*)nullptr, h->d_touched);
  CHK(pri.end());
  h->db_valid = false;
  h->db_rows_valid = rows_fused;                                 // the fused row step imaged every decoder row it stepped
  h->join_pending = true;
  HIPCHK(hipEventRecord(x.released, st));
  HIPCHK(hipGetLastError());
  h->seq++;
  h->acc_examples += bt.E; h->acc_batches++; h->acc_users += nb;
  return 0;
}

int ev_phase0(cdae_hip_t* h, uint64_t u0, uint32_t nu, int mode, uint32_t cidx, uint64_t seed, uint32_t epoch) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_shared || !h->item_shard) return fail("not an item-sharded handle with data");
  if (nu == 0 || nu > EVAL_CHUNK || u0 + nu > h->U) return fail("bad evaluation chunk");
  CHK(join_aux(h));
  const uint32_t n_units = h->h_unit_ptr[u0 + nu] - h->h_unit_ptr[u0];
  CHK(ensure_eval_ws(h, nu, std::max<uint32_t>(n_units, 1u)));
  if (h->hsum_eval_cap < nu) {
    if (h->d_hsum_eval) HIPCHK(hipFree(h->d_hsum_eval));
    h->d_hsum_eval = nullptr; h->hsum_eval_cap = 0;
    CHK(dev_alloc(&h->d_hsum_eval, (size_t)SHARD_BLOCKS * nu * h->Kp));
    h->hsum_eval_cap = nu;
  }
  const uint32_t* uptr = h->d_unit_ptr + u0;
  if (n_units)
    DISPATCH_NI(h->NI, cdae::encode_partial_kernel, dim3((n_units + 3) / 4), dim3(256), 0, h->stream, h->hp, h->d_row_ptr, h->d_col,
                h->P(CDAE_P_W), uptr, n_units, (const uint32_t*)nullptr, u0, nu, mode, mode ? CDAE_STREAM_LOSS_CORRUPT : CDAE_STREAM_CORRUPT, cidx,
                seed, epoch, h->d_hpart_eval, (const uint32_t*)nullptr, 0u, (const uint32_t*)h->d_unit_user, (const uint32_t*)h->d_gpos);
  CHK(sum_and_stage(h, h->d_hpart_eval, uptr, u0, nu, h->d_hsum_eval));
  HIPCHK(hipGetLastError());
  return 0;
}

int ev_finish(cdae_hip_t* h, uint64_t u0, uint32_t nu, int mode) {
  HIPCHK(hipSetDevice(h->device));
  DISPATCH_NI(h->NI, cdae::encode_finish_kernel, dim3((nu + 3) / 4), dim3(256), 0, h->stream, h->hp, h->d_hsum_eval, (const uint32_t*)h->d_iota,
              gathered_wu(h, h->d_hsum_eval, nu), h->P(CDAE_P_B), (const uint32_t*)h->d_iota, u0, nu, mode, h->d_zeval, (float*)nullptr,
              (float*)nullptr, gathered_uu(h, h->d_hsum_eval, nu), (float*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

int ev_data_loss(cdae_hip_t* h, uint64_t u0, uint32_t nu, double* sum) {
  HIPCHK(hipSetDevice(h->device));
  const uint32_t n_units = h->h_unit_ptr[u0 + nu] - h->h_unit_ptr[u0];
  if (n_units == 0) return 0;
  HIPCHK(hipMemsetAsync(h->d_scalar, 0, sizeof(double), h->stream));
  DISPATCH_NI(h->NI, cdae::data_loss_kernel, dim3((n_units + 3) / 4), dim3(256), 0, h->stream, h->hp, h->d_row_ptr, h->d_col,
              h->d_unit_ptr + u0, n_units, (const uint32_t*)h->d_unit_user, u0, nu, h->d_zeval, h->dec(), h->P(CDAE_P_BP), h->d_scalar);
  HIPCHK(hipGetLastError());
  double v = 0;
  HIPCHK(hipMemcpyAsync(&v, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *sum += v;
  return 0;
}

int ev_recommend(cdae_hip_t* h, uint64_t u0, uint32_t nu, uint32_t topk, uint32_t* ids, float* scores) {
  HIPCHK(hipSetDevice(h->device));
  if (topk == 0 || topk > h->I) return fail("topk %u exceeds the %llu items of this shard", topk, (unsigned long long)h->I);
  const size_t lds_scores = (size_t)h->I * sizeof(float) + 64;
  const bool in_lds = lds_scores <= 160 * 1024;
  const size_t shmem = in_lds ? lds_scores : 64;
  uint32_t Bq = nu;
  if (!in_lds) {
    Bq = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nu, (256ull << 20) / ((uint64_t)h->I * sizeof(float))));
    if (h->score_cap < (size_t)Bq * h->I) {
      if (h->d_score) HIPCHK(hipFree(h->d_score));
      h->d_score = nullptr; h->score_cap = 0;
      CHK(dev_alloc(&h->d_score, (size_t)Bq * h->I));
      h->score_cap = (size_t)Bq * h->I;
    }
  }
  if (h->rec_cap < (size_t)Bq * topk) {
    if (h->d_rec) HIPCHK(hipFree(h->d_rec));
    h->d_rec = nullptr; h->rec_cap = 0;
    CHK(dev_alloc(&h->d_rec, (size_t)Bq * topk));
    h->rec_cap = (size_t)Bq * topk;
  }
  if (h->rec_score_cap < (size_t)Bq * topk) {
    if (h->d_rec_score) HIPCHK(hipFree(h->d_rec_score));
    h->d_rec_score = nullptr; h->rec_score_cap = 0;
    CHK(dev_alloc(&h->d_rec_score, (size_t)Bq * topk));
    h->rec_score_cap = (size_t)Bq * topk;
  }
#define SET_SHMEM(NI_) HIPCHK(hipFuncSetAttribute((const void*)cdae::recommend_kernel<NI_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
  switch (h->NI) { case 1: SET_SHMEM(1); break; case 2: SET_SHMEM(2); break; case 4: SET_SHMEM(4); break; default: SET_SHMEM(8); break; }
#undef SET_SHMEM
  for (uint32_t c0 = 0; c0 < nu; c0 += Bq) {
    const uint32_t nb = std::min(Bq, nu - c0);
    DISPATCH_NI(h->NI, cdae::recommend_kernel, dim3(nb), dim3(256), shmem, h->stream, h->hp, h->d_row_ptr, h->d_col, u0 + c0,
                h->d_zeval + (size_t)c0 * h->Kp, h->dec(), h->P(CDAE_P_BP), topk, h->d_rec, in_lds ? (float*)nullptr : h->d_score,
                (const uint32_t*)nullptr, 0u, h->d_rec_score);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ids + (size_t)c0 * topk, h->d_rec, (size_t)nb * topk * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(scores + (size_t)c0 * topk, h->d_rec_score, (size_t)nb * topk * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

}  // namespace cdae_internal
