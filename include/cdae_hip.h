/* cdae_hip.h — C ABI of libcdae_hip.so, the MI355X (gfx950) implementation of the CDAE training hot path.
 *
 * The reference (jasonyaw/CDAE, "libcf") has no FFI: its boundary is C++ template duck-typing —
 * Solver<Model> and Evaluation<Model> instantiated with Model = libcf::CDAE
 * (/root/reference/src/solver/solver.hpp:11-46, src/model/evaluation.hpp:113-181,
 * apps/yelp/yelp.cpp:168-199).  This header is the plain-C surface that sits *underneath* that
 * class: the repo's own host C++ (src/model/recsys/cdae.hpp, same class name and methods as the
 * reference) forwards each Model-concept call to exactly one entry point below, and CHECKs the status
 * so that the reference's abort-on-error convention (glog CHECK / LOG(FATAL)) is preserved.
 *
 *   reference interface (file:line)                         entry point
 *   -------------------------------------------------------------------------------------------
 *   CDAE::CDAE(const CDAEConfig&)        cdae.hpp:39-74      cdae_hip_create
 *   RecsysModelBase::reset               recsys_model_base.hpp:29-34
 *                                                            cdae_hip_set_interactions
 *   CDAE::reset (parameter init)         cdae.hpp:109-134    cdae_hip_init_params / cdae_hip_set_param
 *   CDAE::train_one_iteration            cdae.hpp:136-146    cdae_hip_train_epoch
 *     get_corrputed_input                cdae.hpp:361-371      (device, include/cdae_rng.h)
 *     sample_negative_item               recsys_model_base.hpp:46-57  (device, include/cdae_rng.h)
 *     train_one_user_corruption          cdae.hpp:198-358      (device kernels; explicit-input form:
 *                                                           cdae_hip_train_one_user_corruption)
 *   CDAE::get_hidden_values              cdae.hpp:373-416    cdae_hip_encode
 *   CDAE::data_loss                      cdae.hpp:78-101     cdae_hip_data_loss
 *   CDAE::penalty_loss                   cdae.hpp:103-107    cdae_hip_penalty_loss
 *   CDAE::recommend (all users, top-k)   cdae.hpp:162-196    cdae_hip_recommend_all
 *   TOPN_Evaluation::evaluate            evaluation.hpp:113-181, evaluate_rec_list :183-219
 *                                                            cdae_hip_set_test_rows + cdae_hip_eval_topn
 *   (data-parallel exchange; no reference counterpart)       cdae_hip_delta_*, cdae_hip_comm_*, cdae_hip_exchange_*,
 *                                                            cdae_hip_multi_* (Solver<CDAE>::train on N GPUs)
 *
 * Conventions: every function returns 0 on success, non-zero on failure; cdae_hip_last_error()
 * returns a thread-local message for the last failure.  No exceptions cross the boundary.  All
 * pointers are HOST pointers unless the name says `device`.  The library owns all device memory.
 * A handle is used from one host thread at a time (internally it owns one worker thread that issues the
 * sampling + sorting launches of the coming batches; cdae_hip_destroy joins it).  Parameters are fp32 row-major with the row
 * stride returned by cdae_hip_row_stride() (64, 128, 256 or 512 floats — the smallest that holds
 * num_dim; pad lanes are 0, or 1 in the AdaGrad accumulators, and stay so).
 */
#ifndef CDAE_HIP_H_
#define CDAE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: cdae_hip_config.linear_function (in what was tail padding of the uint32 block: zero the struct before filling it),
 *    parameters CDAE_P_UU / CDAE_P_UU_AG; pipelined delta exchange entry points */
/* 3: cdae_hip_debug_sample_batch (integer-parity test hook), cdae_hip_recommend_user
 * 4: library-owned RCCL communicator + exchange schedule (cdae_hip_comm_*, cdae_hip_exchange_*), cdae_hip_multi_*
 * 5: cdae_hip_create_mf (IMF / BPR handles), CDAE_P_UB / CDAE_P_UB_AG, item-rows layout of cdae_hip_multi_*
 * 6: cdae_hip_set_profiling_families
 * 7: default batch_users capped at 256 (was 512); cdae_hip_default_batch_users, cdae_hip_batch_users; cdae_hip_user_order (IMF / BPR
 *    block schedules train in activity-grouped order; their default is one user per block)
 * 8: cdae_hip_full_output_plan
 * 9: cdae_hip_set_test_rows, cdae_hip_eval_topn, cdae_hip_multi_eval_topn (TOPN metrics on the device)
 * 10: IMF / BPR handles created with batch_users = 0 train the certified block of their model (cdae_hip_mf_default_batch_users: 16 / 8
 *     users on the BASELINE-sized data sets, was 1); BPR's phase U carries the positive item's row from pair to pair
 * 11: schedule of the user-sharded layout: cdae_hip_delta_set_combine (CDAE_COMBINE_GLOBAL_ACC), cdae_hip_multi_set_schedule
 *     (relay warm-up epochs on the single-GPU schedule, users per shard of the exchanged steps, combine rule); the drop-in IMF / BPR
 *     classes pass batch_users = 1 (the reference loop) unless CDAE_BATCH_USERS says otherwise
 * 12: cdae_hip_decode_plan, cdae_hip_set_decode_fused (which launches the sampled decode + hidden-gradient step of a handle is made of);
 *     cdae_hip_multi_steps_per_epoch, cdae_hip_multi_train_steps (a range of the exchanged steps of an epoch: what bench.py times) */
#define CDAE_HIP_ABI_VERSION 12

/* numeric values follow libcf::LossType (/root/reference/src/model/loss.hpp:10-18) */
#define CDAE_LOSS_SQUARE 0u
#define CDAE_LOSS_CROSS_ENTROPY 5u

/* parameter ids (/root/reference/src/model/recsys/cdae.hpp:430-439) */
#define CDAE_P_W 0u      /* items x K  : input embedding and, when !asymmetric, tied decoder  */
#define CDAE_P_W_AG 1u   /* AdaGrad accumulator of W                                          */
#define CDAE_P_V 2u      /* items x K  : decoder when asymmetric                              */
#define CDAE_P_V_AG 3u
#define CDAE_P_WU 4u     /* users x K  : per-user input node (north-star "V_u"), cdae.hpp:434 */
#define CDAE_P_WU_AG 5u
#define CDAE_P_B 6u      /* K          : hidden bias                                          */
#define CDAE_P_B_AG 7u
#define CDAE_P_BP 8u     /* items      : output bias b_prime                                  */
#define CDAE_P_BP_AG 9u
#define CDAE_P_UU 10u     /* users x K  : per-user gate on the input sum, linear_function only, cdae.hpp:437 */
#define CDAE_P_UU_AG 11u
#define CDAE_P_UB 12u     /* users      : user bias ub_ of the IMF / BPR handles (imf.hpp:132); not allocated for CDAE */
#define CDAE_P_UB_AG 13u
#define CDAE_P_COUNT 14u

typedef struct cdae_hip_config {
  uint32_t struct_size;      /* sizeof(cdae_hip_config), for ABI checking                  */
  uint32_t num_dim;          /* CDAEConfig::num_dim          cdae.hpp:19                   */
  uint32_t num_neg;          /* CDAEConfig::num_neg          cdae.hpp:26                   */
  uint32_t num_corruptions;  /* CDAEConfig::num_corruptions  cdae.hpp:22                   */
  uint32_t loss_type;        /* CDAE_LOSS_*                  cdae.hpp:17                   */
  uint32_t using_adagrad;    /* cdae.hpp:20 (0 -> plain SGD with L2)                       */
  uint32_t asymmetric;       /* cdae.hpp:23                                                */
  uint32_t user_factor;      /* cdae.hpp:24                                                */
  uint32_t linear;           /* cdae.hpp:25 (identity hidden activation)                   */
  uint32_t scaled;           /* cdae.hpp:27 (input scale 1/(1-q))                          */
  uint32_t tanh_act;         /* cdae.hpp:30                                                */
  uint32_t batch_users;      /* users whose encode sees the same parameter snapshot; 1 ==  */
                             /* the reference's strictly sequential schedule; 0 -> default */
                             /* (cdae_hip_default_batch_users: num_users/160 rounded down   */
                             /* to 32, within [32, 256] — the accuracy envelope's bound)    */
  uint32_t full_output;      /* 1: every unrated item is a negative with target 0 (north-star */
                             /* extension; num_neg is ignored): dense decode on the MFMA cores, */
                             /* per-block summed decoder gradient            */
  uint32_t linear_function;  /* cdae.hpp:29: h = Uu[u] (.) (scale * sum W[k]) + b + Wu[u], Uu trained */
                             /* (fills what used to be alignment padding: struct_size is unchanged)   */
  double lambda;             /* cdae.hpp:15 */
  double learn_rate;         /* cdae.hpp:16 */
  double corruption_ratio;   /* cdae.hpp:21 */
  double beta;               /* cdae.hpp:28 */
} cdae_hip_config;

typedef struct cdae_hip_stats {
  double wall_seconds;       /* host wall-clock of the call, stream-synchronised            */
  uint64_t users;            /* user-corruption units trained                               */
  uint64_t examples;         /* (user, output item) pairs decoded = sum (1+num_neg) n_u     */
  uint64_t batches;
  /* accumulated HIP-event milliseconds per kernel family; filled only when profiling is on */
  double ms_sample, ms_sort, ms_encode, ms_decode, ms_hidden, ms_input;
  uint64_t launches_decode;
} cdae_hip_stats;

typedef struct cdae_hip cdae_hip_t;

const char* cdae_hip_last_error(void);
int cdae_hip_abi_version(void);

int cdae_hip_create(const cdae_hip_config* cfg, int device_id, cdae_hip_t** out);
int cdae_hip_destroy(cdae_hip_t* h);

/* CSR of the training interactions: row_ptr[num_users+1], col_idx[nnz] sorted ascending and unique
 * inside each row (the reference's uid -> {iid -> label} hashtable, data-inl.hpp:414-429, with the
 * always-1 labels of yelp.cpp:60-66 dropped).  Copied; the caller keeps its arrays. */
int cdae_hip_set_interactions(cdae_hip_t* h, uint64_t num_users, uint64_t num_items,
                              const int64_t* row_ptr, const uint32_t* col_idx);

uint32_t cdae_hip_row_stride(const cdae_hip_t* h);

/* The batch_users a handle created with batch_users = 0 takes for `num_users` users (chosen at cdae_hip_set_interactions):
 * num_users / 160 rounded down to a multiple of 32, within [32, CDAE_DEFAULT_BATCH_USERS_MAX].  The upper bound is the largest
 * size for which the driver-run accuracy tests hold the Recall@10 / loss-curve tolerance against the strictly sequential
 * reference schedule (cdae.hpp:136-146) at the BASELINE shapes; larger values are accepted when asked for explicitly and are
 * a throughput setting outside that envelope.  cdae_hip_batch_users: the value a handle is using (0 before
 * cdae_hip_set_interactions when the default was asked for). */
#define CDAE_DEFAULT_BATCH_USERS_MAX 256u
uint32_t cdae_hip_default_batch_users(uint64_t num_users);
uint32_t cdae_hip_batch_users(const cdae_hip_t* h);

/* Which launches the SAMPLED step's decode (cdae.hpp:225-293) and hidden-gradient sum (cdae.hpp:240,248,277,285) of this handle are
 * made of, chosen at cdae_hip_set_interactions from the data set and batch_users (0 before it):
 *   *hot_rows   item rows that take a wavefront of their own (the most popular: >= 48 expected positives per batch);
 *   *late_rows  the first of them (at most 64) whose hidden-gradient terms the per-user finish adds itself instead of the gather;
 *   *fused      1: decode and gather are ONE launch (the gather waits, example by example, for the g of rows that finish early
 *               and never for the late rows); 0: two launches.  Either order gives the same bits.
 * Any pointer may be NULL. */
int cdae_hip_decode_plan(const cdae_hip_t* h, uint32_t* hot_rows, uint32_t* late_rows, uint32_t* fused);
/* allow = 0: this handle takes the two separate launches (same bits).  REQUIRED for handles whose training calls may be in flight on the
 * SAME device at the same time as another handle's (two models trained from two threads; the logical shards of cdae_hip_multi_* with
 * equal device ids, for which the library does it itself): inside the fused launch the gather wavefronts hold their slots while they wait
 * for rows of the same launch, and with two such launches on one device each can keep the other's row workgroups from being dispatched
 * — the waits are bounded and end in an error from cdae_hip_synchronize, not in a hang, but the epoch is lost.  One handle per device
 * at a time (the reference's own use: one model, one training loop) needs nothing. */
int cdae_hip_set_decode_fused(cdae_hip_t* h, int allow);

/* Which launches the full-output decode (cdae_hip_config.full_output; the reference has no counterpart: its training decode is
 * always sampled, cdae.hpp:217-293) of this handle is made of, once cdae_hip_set_interactions has run — for a caller that prices
 * the kernel families cdae_hip_get_stats times (bench.py: which of the three products are inside the "decode" family):
 *   CDAE_PLAN_FUSED_DECODE  K <= 256: forward product, loss' and hidden-gradient product in one launch (full_decode_fused_kernel)
 *   CDAE_PLAN_GEMM2_TN      K > 256: hg = G D read from G^T and the row-major decoder image (gemm_tn_bf16_kernel; GEMM 1 writes no G)
 *   CDAE_PLAN_ROWS_FUSED    K > 256, >= 32768 items: dD = G^T Z and the row steps in one launch (gemm3_rows_fused_kernel),
 *                           timed in the "input" family — the "decode" family then holds two of the three products
 * 0 for a sampled-decode handle. */
#define CDAE_PLAN_FUSED_DECODE 1u
#define CDAE_PLAN_GEMM2_TN 2u
#define CDAE_PLAN_ROWS_FUSED 4u
uint32_t cdae_hip_full_output_plan(const cdae_hip_t* h);

/* A data-parallel rank holds only its own users (rows re-based to 0).  The random streams of
 * include/cdae_rng.h are keyed by GLOBAL user id = offset + local row, so a sharded run draws the
 * same masks and negatives as a single-GPU run over all users.  Default 0. */
int cdae_hip_set_user_id_offset(cdae_hip_t* h, uint64_t global_id_of_local_user_0);

/* reset(): W, V, Wu ~ U(-1,1) * 4*sqrt(6/(I+K)), accumulators 1e-4, biases 0 (cdae.hpp:109-134),
 * drawn from the CDAE_STREAM_INIT counter stream of include/cdae_rng.h.  Wu rows are keyed by GLOBAL user id
 * (cdae_hip_set_user_id_offset): a shard initialises its rows to what a single-GPU run would. */
int cdae_hip_init_params(cdae_hip_t* h, uint64_t seed);

/* dense [rows x num_dim] (or [n]) fp32 host arrays, unpadded */
int cdae_hip_set_param(cdae_hip_t* h, uint32_t which, const float* host, size_t count);
int cdae_hip_get_param(cdae_hip_t* h, uint32_t which, float* host, size_t count);
/* The device array itself (row stride cdae_hip_row_stride(), padded_count elements).  The pointer is WRITABLE: the call marks the
 * full-output path's bf16 images of the decoder stale, so a caller that writes parameters through it trains on what it wrote (write
 * before the next training call, not during one).  Refused for the user-indexed arrays (Wu, Uu, user bias) of an IMF / BPR handle with
 * batch_users > 1, whose rows are stored in training order (cdae_hip_user_order), not by user id. */
int cdae_hip_param_device_ptr(cdae_hip_t* h, uint32_t which, void** device_ptr, size_t* padded_count);

/* One pass over users [u_begin, u_end) x num_corruptions in batches of batch_users
 * (train_one_iteration, cdae.hpp:136-146).  train_epoch == train_users over all users. */
int cdae_hip_train_epoch(cdae_hip_t* h, uint64_t seed, uint32_t epoch, cdae_hip_stats* stats);
int cdae_hip_train_users(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin,
                         uint64_t u_end, cdae_hip_stats* stats);
/* Asynchronous forms.  enqueue_users = train_users without the final host synchronisation (several calls
 * queue back to back on the library's stream; the call returns when its batches are QUEUED, and since round 6 it paces
 * itself on the side streams: before it queues a batch it looks, for at most 200 us, whether that batch's sampled and
 * sorted lists are complete, so that the training stream need not wait for them — a call of n batches returns about
 * two batches before the device has trained them, not n); prefetch_users runs only the sampling + sorting of the leading
 * batch(es) of a range on the side stream(s) — as many as the library looks ahead: one, or two where the second
 * prep lane is on — so that it overlaps whatever the caller does next (e.g. the RCCL all-reduce of the
 * data-parallel exchange); later train/enqueue calls for the same (seed, epoch) starting at the same user
 * pick the prepared batches up, one-batch calls included.  collect_stats synchronises and returns the counters (and, with profiling
 * on, the HIP-event kernel times) accumulated since the previous train_users / collect_stats. */
int cdae_hip_enqueue_users(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end);
int cdae_hip_prefetch_users(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end);
int cdae_hip_collect_stats(cdae_hip_t* h, cdae_hip_stats* stats);

/* The reference's public per-user step with the caller's own corrupted input set,
 * train_one_user_corruption(uid, input_set, output_set) (cdae.hpp:198-200; output_set is the user's
 * train row), plus the negatives the reference would draw at cdae.hpp:217-220.  Known-answer tests use it
 * to feed fixed masks and negatives (duplicates allowed, processed in the given order). */
int cdae_hip_train_one_user_corruption(cdae_hip_t* h, uint64_t uid, const uint32_t* input_items,
                                       size_t n_input, const uint32_t* negative_items, size_t n_negative);
/* Developer/benchmark aid: period 0 = off; k >= 1 = HIP events (on the stream each kernel is launched on) around the
 * kernel families of every k-th batch; cdae_hip_stats.ms_* / launches_decode then cover the sampled batches only. */
int cdae_hip_set_profiling(cdae_hip_t* h, int period);
/* ... restricted to the families in `mask` (bit CDAE_FAMILY_*; default all): an event pair costs ~6 us of stream time, so a
 * benchmark times only the family its roofline needs inside its timed region. */
#define CDAE_FAMILY_SAMPLE 0
#define CDAE_FAMILY_SORT 1
#define CDAE_FAMILY_ENCODE 2
#define CDAE_FAMILY_DECODE 3
#define CDAE_FAMILY_HIDDEN 4
#define CDAE_FAMILY_INPUT 5
int cdae_hip_set_profiling_families(cdae_hip_t* h, uint32_t mask);
int cdae_hip_synchronize(cdae_hip_t* h);

/* Test hook for the INTEGER work of the path (bit-exact parity, tests/test_gpu_integer.py): runs the sampling, the
 * item sort and the segmentation of ONE batch — users [u_begin, u_begin + n_users) (n_users <= batch_users),
 * corruption `cidx` — exactly as cdae_hip_train_users would (get_corrputed_input cdae.hpp:361-371, sample_negative_item
 * recsys_model_base.hpp:46-57 / cdae.hpp:217-220) and copies the example lists back instead of training on them:
 *   ex_item / ex_val [E]          user-major: per user its n_u train items, then its n_u * num_neg negatives;
 *                                 val = example index << 32 | slot | target << 30 | is_input << 31
 *   sorted_item / sorted_val [E]  the same list stably sorted by item (user order inside an item); sorted_val also
 *                                 carries dup_prev << 28 | dup_next << 29 (neighbour in the row is the same user's)
 *   seg_begin / seg_end [num_items]   sorted range of every item (0, 0 for items without examples)
 *   dup_of_pos / dup_of_ex [E]    correction-row number of every second-or-later duplicate negative, by sorted
 *                                 position / by example (0xFFFFFFFF elsewhere); the numbering itself is arbitrary
 * E = (1 + num_neg) * (train items of the batch's users) (full_output: the positives only); *n_examples is the
 * capacity of the arrays on entry and E on return.  Any output pointer may be NULL. */
int cdae_hip_debug_sample_batch(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint32_t n_users,
                                uint32_t cidx, uint32_t* ex_item, uint64_t* ex_val, uint32_t* sorted_item,
                                uint64_t* sorted_val, uint32_t* seg_begin, uint32_t* seg_end, uint32_t* dup_of_pos,
                                uint32_t* dup_of_ex, uint64_t* n_examples);

/* z for `n` users (get_hidden_values, cdae.hpp:373-416).  mode 0: full train row, scale 1 (the
 * inference form, cdae.hpp:169); mode 1: training corruption of (seed, epoch, corruption 0) with the
 * configured scale.  Z is [n x num_dim] fp32 on the host. */
int cdae_hip_encode(cdae_hip_t* h, uint64_t seed, uint32_t epoch, int mode, const uint32_t* uids,
                    size_t n, float* Z);

/* data_loss (cdae.hpp:78-101) with the CDAE_STREAM_LOSS_CORRUPT masks; penalty_loss (cdae.hpp:103-107) */
int cdae_hip_data_loss(cdae_hip_t* h, uint64_t seed, uint32_t epoch, double* out);
int cdae_hip_penalty_loss(cdae_hip_t* h, double* out);

/* recommend() for users [u_begin, u_end): top-k unrated items by W'[i].z + b'[i], descending score,
 * ties -> lower item id first (heap.hpp:44-52 + utils.hpp:16-19 with ascending scan order).  num_dim <= 256 and topk <= 16
 * run on the matrix cores (all users of a chunk per launch); any other combination — and any item count — takes the
 * general one-workgroup-per-user path.
 * out is [(u_end-u_begin) x topk] uint32 on the host. */
int cdae_hip_recommend_all(cdae_hip_t* h, uint64_t u_begin, uint64_t u_end, uint32_t topk,
                           uint32_t* out);

/* TOPN_Evaluation::evaluate (evaluation.hpp:113-181) with evaluate_rec_list (evaluation.hpp:183-219) on the device: the top-`topk`
 * list of EVERY user over the items outside the user's train row (what cdae_hip_recommend_all returns) is scored against the
 * user's validation row without leaving the GPU.  cdae_hip_set_test_rows hands the validation rows over once per data set — CSR
 * over the handle's users, items ascending and unique inside a row (the uid -> {iid} table evaluation.hpp:118-120 rebuilds on
 * every call); copied.  cdae_hip_eval_topn returns
 *   rets[8] = P@1 P@5 P@10 R@1 R@5 R@10 MAP@5 MAP@10, each the sum over the users WITH test items of r / (number of such users),
 *             added in user order: the bits of the reference's sequential loop (evaluation.hpp:160-166) in fp64;
 *   hits[3] = hits in the first 1 / 5 / 10 places summed over all users (integers; may be NULL);
 *   ids_out = the [num_users x topk] table itself (may be NULL: then nothing but 16 numbers crosses PCIe).
 * topk is what the evaluation asks recommend() for (10, evaluation.hpp:145); only the first 20 places are scored (:186). */
int cdae_hip_set_test_rows(cdae_hip_t* h, const int64_t* test_row_ptr, const uint32_t* test_col);
int cdae_hip_eval_topn(cdae_hip_t* h, uint32_t topk, double* rets8, uint64_t* hits3, uint32_t* ids_out);

/* recommend(uid, topk, rated_item_set) (cdae.hpp:162-196) for a rated set that is NOT the user's train row: the hidden
 * layer is encoded from `rated_items` (scale 1, cdae.hpp:169) and exactly those items are excluded (cdae.hpp:177-179).
 * Items need not be sorted; duplicates are an error.  out is [topk] uint32 on the host. */
int cdae_hip_recommend_user(cdae_hip_t* h, uint64_t uid, const uint32_t* rated_items, size_t n_rated, uint32_t topk,
                            uint32_t* out);

/* ---- data-parallel exchange (north star: RCCL all-reduce of the shared W / W' / bias gradients;
 * Wu never leaves its GPU).  Each rank trains its own users from a common snapshot, then
 *   cdae_hip_delta_begin   : snapshot the shared parameters
 *   ... cdae_hip_train_users ...
 *   cdae_hip_delta_compute : delta = current - snapshot, packed in one device buffer together with a
 *                            per-row touch indicator
 *   <all-reduce(sum) of that buffer by the caller, e.g. torch.distributed over RCCL>
 *   cdae_hip_delta_apply   : current = snapshot + combine(summed delta)
 * Layout of the buffer: cdae_hip_delta_device_ptr(). */
/* All four delta calls are stream-ordered on the library's HIP stream (cdae_hip_stream) and do not block the
 * host; run the all-reduce on that stream, or synchronise (cdae_hip_synchronize) before touching the buffer
 * from another one. */
int cdae_hip_stream(cdae_hip_t* h, void** hip_stream);
int cdae_hip_delta_begin(cdae_hip_t* h);
int cdae_hip_delta_compute(cdae_hip_t* h);
int cdae_hip_delta_device_ptr(cdae_hip_t* h, void** device_ptr, size_t* count_floats);
#define CDAE_DELTA_SUM 0u         /* current = snapshot + sum over ranks (summed gradients; default)      */
#define CDAE_DELTA_TOUCH_MEAN 1u  /* item rows / #ranks that touched them, hidden bias / world_size      */
int cdae_hip_delta_apply(cdae_hip_t* h, uint32_t world_size, uint32_t rule);

/* Pipelined form of the same exchange (sum rule): the all-reduce of one period's deltas overlaps the next period's
 * training, and the other ranks' part is folded in one period late.  After cdae_hip_delta_begin() (agreed state A = current):
 *   cdae_hip_delta_stage : send = recv = current - A ; snap = current
 *   (caller all-reduces the recv buffer, cdae_hip_delta_recv_device_ptr, n floats, asynchronously; the buffer is
 *    compact: the matrices' pad columns are not exchanged)
 *   cdae_hip_delta_merge : A += recv ; current = A + (current - snap)     (before the next _stage)
 * A moves only by the all-reduced sums, so it is the same BITS on every rank; when nothing was trained between a stage and
 * its merge (synchronous exchange, final flush) every rank ends with current == A, bit for bit.
 * Stream-ordered on cdae_hip_stream like the calls above.  (cdae_hip_exchange_* below drive these with a library-owned
 * RCCL communicator; these entry points remain for hosts that bring their own collective.) */
int cdae_hip_delta_stage(cdae_hip_t* h);
/* How _merge folds the all-reduced buffer in (set it before the first _stage, or between a _merge and the next _stage):
 *   CDAE_COMBINE_SUM         (default) the algebra above: the replicas' accumulated steps are summed.
 *   CDAE_COMBINE_GLOBAL_ACC  AdaGrad handles only (others keep the sum): per (parameter, accumulator) pair the replicas exchange their
 *       accumulator growth — exactly their sum of squared gradients, cdae.hpp:254 — and their step with their own preconditioner taken
 *       back out, (current - A) * (beta + sqrt(acc)); _merge takes ONE step with the accumulator that has seen every replica:
 *       A_acc += sum ; A += sum / (beta + sqrt(A_acc))  (cdae_exchange_algebra.h pipe_pair).  Same buffers, same all-reduce. */
#define CDAE_COMBINE_SUM 0u
#define CDAE_COMBINE_GLOBAL_ACC 1u
int cdae_hip_delta_set_combine(cdae_hip_t* h, uint32_t combine);
int cdae_hip_delta_recv_device_ptr(cdae_hip_t* h, void** device_ptr, size_t* count_floats);
int cdae_hip_delta_merge(cdae_hip_t* h);
int cdae_hip_delta_merge_stage(cdae_hip_t* h);   /* _merge of the previous period then _stage of this one, in one pass */

/* ---- the sibling SGD models IMF and BPR on the same handle type (yelp.cpp:122-165 --method=MF / BPR) ----
 *   IMF::reset / BPR::reset        imf.hpp:57-69                cdae_hip_set_interactions + cdae_hip_init_params
 *   IMF::train_one_iteration       imf.hpp:71-86 (+ :88-115)    cdae_hip_train_epoch / _train_users
 *   BPR::train_one_iteration       bpr.hpp:56-70 (+ :72-106)    (same; pairwise = 1)
 *   RecsysModelBase::recommend     recsys_model_base.hpp:77-104 cdae_hip_recommend_all   (score = ub + ib + uv . iv, imf.hpp:117-119)
 * Parameters through cdae_hip_get/set_param: CDAE_P_WU / _WU_AG = uv_ / uv_ag_ (users x K), CDAE_P_W / _W_AG = iv_ / iv_ag_
 * (items x K), CDAE_P_UB / _UB_AG = ub_ / ub_ag_, CDAE_P_BP / _BP_AG = ib_ / ib_ag_.  loss_type: 0 SQUARE, 1 LOGISTIC, 2 LOG,
 * 3 HINGE, 5 CROSS_ENTROPY (loss.hpp:10-18; what yelp.cpp lets these models choose).  batch_users: users whose chains run
 * concurrently against the block-start item rows; 1 == the reference's strictly sequential loop.  data_loss / penalty_loss are
 * 0 for these models, as in the reference (ModelBase defaults, model_base.hpp:36-45); cdae_hip_encode, the explicit-input step
 * and the full-output decode do not apply. */
/* Training order.  A CDAE handle, and an IMF / BPR handle with one user per block (batch_users = 1: what the drop-in classes
 * src/model/recsys/imf.hpp / bpr.hpp pass by default), visit the users in id order like the reference.
 * An IMF / BPR handle with batch_users > 1 (the block schedule: what batch_users = 0 selects on BASELINE-sized data sets, a throughput setting beyond it) trains them in ACTIVITY-GROUPED order — users
 * sorted by train-row length, cut into blocks of batch_users, the blocks visited in a fixed pseudo-random order — because a block
 * lasts as long as its most active user's serial chain.  cdae_hip_user_order returns that order: out[position] = user id
 * (count = num_users; the identity for every other handle).  The random streams, cdae_hip_train_users' range and
 * cdae_hip_debug_sample_batch's window are in POSITIONS; get / set_param and recommend_all are by user id as everywhere. */
int cdae_hip_user_order(cdae_hip_t* h, uint32_t* out, size_t count);

typedef struct cdae_mf_config {
  uint32_t struct_size;      /* sizeof(cdae_mf_config)                                   */
  uint32_t num_dim;          /* IMFConfig::num_dim          imf.hpp:19                   */
  uint32_t num_neg;          /* imf.hpp:20                                               */
  uint32_t loss_type;        /* imf.hpp:17 / bpr.hpp:17                                  */
  uint32_t using_adagrad;    /* imf.hpp:22                                               */
  uint32_t using_bias_term;  /* imf.hpp:21                                               */
  uint32_t pairwise;         /* 0: IMF (pointwise instances), 1: BPR (pairs)             */
  uint32_t batch_users;      /* 1: the reference's sequential loop; > 1: block schedule  */
                             /* in activity-grouped order (cdae_hip_user_order).  0 ->   */
                             /* cdae_hip_mf_default_batch_users(num_users, pairwise)      */
  double lambda;             /* imf.hpp:16                                               */
  double learn_rate;         /* imf.hpp:14                                               */
  double beta;               /* imf.hpp:15                                               */
} cdae_mf_config;
/* The default blocks (batch_users = 0): the largest sizes measured INSIDE the sampled CDAE path's accuracy bound — Recall@10 within
 * +-0.002 of the sequential loop at every epoch as a mean over six seeds, against fp64 fixtures of the loop at ML-10M shape K=200
 * (tests/test_gpu_mf.py; tools/mf_envelope.py has the other sizes and Yelp shape) — chosen when the data set is known
 * (cdae_hip_set_interactions):
 *   IMF  16 users per block from 8 192 users on (measured at 10 000 and 70 000 users; 32 users per block sit up to +0.0035, 64 up to
 *        +0.005): 43 x the loop's users/s at ML-10M shape;
 *   BPR   8 users per block from 65 536 users on (70 000 users: inside the bound; 10 000 users: +0.003 in the first two epochs, so
 *        smaller data sets keep the loop): 22 x the loop's users/s.  Since ABI 10 a BPR user's pairs see their shared positive item's
 *        row as the loop leaves it between them (a private copy carried through phase U); before, every block size was 0.008 low
 *        in the first two epochs.
 * Smaller data sets train one user per block: the reference loop itself. */
#define CDAE_IMF_DEFAULT_BATCH_USERS 16u
#define CDAE_IMF_DEFAULT_MIN_USERS 8192u
#define CDAE_BPR_DEFAULT_BATCH_USERS 8u
#define CDAE_BPR_DEFAULT_MIN_USERS 65536u
uint32_t cdae_hip_mf_default_batch_users(uint64_t num_users, uint32_t pairwise);
int cdae_hip_create_mf(const cdae_mf_config* cfg, int device_id, cdae_hip_t** out);

/* ---- library-owned RCCL communicator and exchange schedule (one process per GPU: bench.py --gpus N) -----------------
 * The all-reduce of the staged deltas runs inside the library, on its own communicator and HIP stream, overlapped with the
 * training kernels; the host application only distributes the 128-byte unique id (any channel: bench.py uses a gloo
 * broadcast) and calls _exchange_step after every enqueued batch.  period 0: synchronous exchange at every step (stage,
 * all-reduce, merge); period k >= 1: pipelined — every k steps the accumulated delta is staged and reduced asynchronously,
 * the other ranks' part is merged k steps later.  _exchange_flush reduces and merges what is outstanding: afterwards every
 * rank holds the same shared parameters.  A handle without a communicator behaves as a one-rank group.
 * _exchange_time_all_reduce: seconds per all-reduce of the exchange buffer with the device otherwise idle (call it before
 * the deltas matter: it flushes, and restarts the exchange from the current parameters). */
#define CDAE_COMM_ID_BYTES 128
int cdae_hip_comm_unique_id(void* out, size_t bytes);
int cdae_hip_comm_init_rank(cdae_hip_t* h, int world_size, int rank, const void* unique_id, size_t bytes);
int cdae_hip_exchange_configure(cdae_hip_t* h, int period);
int cdae_hip_exchange_step(cdae_hip_t* h);
int cdae_hip_exchange_flush(cdae_hip_t* h);
int cdae_hip_exchange_time_all_reduce(cdae_hip_t* h, int repeats, double* seconds);

/* ---- several user shards behind one handle (one process, N GPUs): what Solver<CDAE>::train (solver-inl.hpp:51-55) drives
 * when src/model/recsys/cdae.hpp is given CDAE_DEVICES=0,1,... ------------------------------------------------------------
 * device_ids all distinct: one shard per GPU, one host thread per shard during an epoch, RCCL communicator from
 * ncclCommInitAll.  device_ids all equal: logical shards of ONE GPU — same schedule, the all-reduce is a fixed-order sum kernel
 * (tests, accuracy envelope).  Users are cut into contiguous ranges balanced by interactions; Wu / Wu_ag stay on their
 * shard; get/set_param address the global matrices.  Every entry point mirrors its single-handle namesake. */
typedef struct cdae_hip_multi cdae_hip_multi_t;
/* How the shards divide the model (set before _set_interactions):
 *   CDAE_LAYOUT_USERS      (default) user shards + exchange of shared-parameter deltas, as described above.
 *   CDAE_LAYOUT_ITEM_ROWS  every shard owns a contiguous range of ITEM rows — W / W_ag / (V / V_ag) / b' and the decode over them —
 *       and sees every user; b is replicated and stepped identically everywhere; the user node (Wu, Uu) is sharded by contiguous
 *       USER ranges (the owner's rows of a batch ride the first all-reduce).  Per batch two all-reduces of [batch_users x row_stride]
 *       floats cross the shards (the input sums of the encode, the hidden gradient); no item-row parameter ever does.  The schedule
 *       is the single-GPU schedule EXACTLY (same steps, same order; only the two sums are associated differently), so this layout
 *       has no accuracy cost — for the sampled decode (every shard samples the batch's whole example list against the whole rows and
 *       keeps the examples of its rows; one shard is the single handle bit for bit) and for the full-output decode (BASELINE
 *       configs[4]: 1 M items x K=512, whose dense delta would be 2 GB).  _shard() reports item ranges in this layout. */
#define CDAE_LAYOUT_USERS 0u
#define CDAE_LAYOUT_ITEM_ROWS 1u
int cdae_hip_multi_set_layout(cdae_hip_multi_t* m, uint32_t layout);
int cdae_hip_multi_create(const cdae_hip_config* cfg, int n_shards, const int* device_ids, cdae_hip_multi_t** out);
int cdae_hip_multi_destroy(cdae_hip_multi_t* m);
int cdae_hip_multi_num_shards(const cdae_hip_multi_t* m);
int cdae_hip_multi_shard(cdae_hip_multi_t* m, int shard, cdae_hip_t** handle, uint64_t* u_begin, uint64_t* u_end);
int cdae_hip_multi_set_interactions(cdae_hip_multi_t* m, uint64_t num_users, uint64_t num_items, const int64_t* row_ptr,
                                    const uint32_t* col_idx);
int cdae_hip_multi_init_params(cdae_hip_multi_t* m, uint64_t seed);
int cdae_hip_multi_set_exchange(cdae_hip_multi_t* m, int period);
/* The whole schedule of CDAE_LAYOUT_USERS (ABI 11).  An epoch e of `train_epoch` runs in two parts:
 *   RELAY  the first  R = clamp((relay_epochs - e) * num_users, 0, num_users)  users (fractions of an epoch allowed) are trained on the
 *          SINGLE-GPU schedule — Solver<CDAE>::train's order (cdae.hpp:136-146): users 0, 1, 2, ... in blocks of the handles'
 *          batch_users, every item row's chain sequential — by the shard that holds them, and the shared block is handed from shard to
 *          shard when the range crosses a cut (one device-to-device copy of [W | W_ag | b' | ... ] per shard and epoch; the other GPUs
 *          wait).  The blocks restart at every shard cut (the cuts balance interactions, they are not multiples of batch_users), so the
 *          trajectory is the single-GPU SCHEDULE — bit for bit one handle walked range by range — not the single handle's own block grid.
 *          It costs a single GPU's time and has the single-GPU schedule's accuracy: the warm-up that DESIGN.md §7 measured to
 *          bring the exchanged steps back towards the envelope (young AdaGrad accumulators are what the summed steps overshoot on).
 *   EXCHANGE  the remaining users of every shard, `sync_batch_users` (0 = the handles' batch_users) per shard and step, deltas
 *          exchanged every step (period 0) or pipelined (period k), folded in by `combine` (CDAE_COMBINE_*).
 * cdae_hip_multi_set_exchange(m, p) changes the PERIOD only: combine rule, sync_batch_users and relay_epochs stay as the last
 * set_schedule left them (a fresh handle: CDAE_COMBINE_SUM, 0, 0.0). */
typedef struct cdae_multi_schedule {
  int32_t period;
  uint32_t combine;
  uint32_t sync_batch_users;
  uint32_t reserved;
  double relay_epochs;
} cdae_multi_schedule;
int cdae_hip_multi_set_schedule(cdae_hip_multi_t* m, const cdae_multi_schedule* schedule);
int cdae_hip_multi_train_epoch(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, cdae_hip_stats* stats);
/* users [u_begin, u_end) only — CDAE_LAYOUT_ITEM_ROWS (every shard sees every user); the user-sharded layout trains whole epochs */
int cdae_hip_multi_train_users(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end,
                               cdae_hip_stats* stats);
/* User-sharded layout: the exchanged part of an epoch is `steps_per_epoch` steps (every shard trains sync_batch_users — or batch_users — of
 * its users, then the exchange).  cdae_hip_multi_train_steps runs steps [step_begin, step_end) of epoch `epoch` WITHOUT the relay part and
 * ends with a flush (replicas identical): a measurement hook — bench.py times K steps with it; training goes through train_epoch. */
uint64_t cdae_hip_multi_steps_per_epoch(const cdae_hip_multi_t* m);
int cdae_hip_multi_train_steps(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, uint64_t step_begin, uint64_t step_end,
                               cdae_hip_stats* stats);
int cdae_hip_multi_data_loss(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, double* out);
int cdae_hip_multi_penalty_loss(cdae_hip_multi_t* m, double* out);
int cdae_hip_multi_recommend_all(cdae_hip_multi_t* m, uint64_t u_begin, uint64_t u_end, uint32_t topk, uint32_t* out);
/* cdae_hip_eval_topn for a sharded model: the shards' lists are merged on the host (cdae_hip_multi_recommend_all), the eight means
 * are summed there in user order (same expressions and order of additions as the single-handle kernels) */
int cdae_hip_multi_eval_topn(cdae_hip_multi_t* m, const int64_t* test_row_ptr, const uint32_t* test_col, uint32_t topk,
                             double* rets8, uint64_t* hits3, uint32_t* ids_out);
int cdae_hip_multi_get_param(cdae_hip_multi_t* m, uint32_t which, float* host, size_t count);
int cdae_hip_multi_set_param(cdae_hip_multi_t* m, uint32_t which, const float* host, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* CDAE_HIP_H_ */
