// cdae_internal.hpp — what cdae_multi.hip (data-parallel exchange, multi-shard handle) needs from cdae_hip.hip.
// Not part of the C ABI.  The kernels stay in ONE translation unit (cdae_hip.hip); this is a narrow accessor surface.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/cdae_hip.h"

// Developer switches (A/B paths the bit-equality tests flip, tuning knobs, timing experiments) are environment variables of the
// DEVELOPER build only (-DCDAE_DEVELOPER: build/libcdae_hip_dev.so, made by __graft_entry__.build() beside the shipped library and
// loaded by the tests that need it).  The shipped library reads NO environment variable: the names are not even in its string table.
#ifdef CDAE_DEVELOPER
#define DEV_ENV(name) std::getenv(name)
#else
#define DEV_ENV(name) ((const char*)nullptr)
#endif

namespace cdae_internal {

int fail(const char* fmt, ...);                 // sets cdae_hip_last_error(), returns 1

int device_of(const cdae_hip_t* h);
hipStream_t main_stream(cdae_hip_t* h);
hipStream_t aux_stream(cdae_hip_t* h);         // second stream of the handle (full-output path: the b recurrence); idle in the sampled path
uint64_t num_users(const cdae_hip_t* h);
uint64_t num_items(const cdae_hip_t* h);
uint32_t batch_users(const cdae_hip_t* h);      // min(batch_users, num_users)
bool ready(const cdae_hip_t* h);                // set_interactions has run

// shared block, staged delta (`send`) and receive buffer of the pipelined exchange; all compact_count() floats long
// (allocated by the first pipe_stage)
float* send_buf(cdae_hip_t* h);
float* recv_buf(cdae_hip_t* h);
size_t compact_count(const cdae_hip_t* h);

// 0.5 * lambda * (|Wu|^2) of this handle's users (0 when !user_factor): the private part of penalty_loss
int private_penalty(cdae_hip_t* h, double* out);
// 0.5 * lambda * (|W|^2 + |V|^2 + |b|^2 + |b'|^2): the shared part
int shared_penalty(cdae_hip_t* h, double* out);
// 0.5 * lambda * (|W|^2 + |V|^2 + |b'|^2) (item rows only) and 0.5 * lambda * |b|^2: the pieces an item-sharded model adds up
int item_rows_penalty(cdae_hip_t* h, double* out);
int hidden_bias_penalty(cdae_hip_t* h, double* out);

// the validation rows of cdae_hip_set_test_rows / cdae_hip_multi_eval_topn: row_ptr starts at 0 and does not decrease, col is there when
// there are interactions, every row ascending, unique and inside [0, I) (the metric's binary search needs the order)
int validate_test_rows(const int64_t* test_row_ptr, const uint32_t* test_col, uint64_t U, uint64_t I, uint64_t* users_with_rows);

// dst's shared block [W | W_ag | (V | V_ag) | b' | b'_ag | b | b_ag] := src's (same model, any two devices of this process); waits for
// src's work, stream-ordered on dst's main stream.  The relay part of cdae_hip_multi_set_schedule hands the parameters on with it.
int adopt_shared_block(cdae_hip_t* dst, cdae_hip_t* src);
// the synchronous exchange's forms of cdae_hip_delta_stage / _merge (nothing trained between them: no snapshot, no send copy — the
// all-reduce works in place on recv_buf); a stage_sync must be followed by merge_sync before anything else touches the exchange state
int delta_stage_sync(cdae_hip_t* h);
int delta_merge_sync(cdae_hip_t* h);

// exchange state owned by cdae_multi.hip, destroyed with the handle
void*& exchange_slot(cdae_hip_t* h);
void set_exchange_deleter(cdae_hip_t* h, void (*deleter)(void*));


// ---- item-sharded layout: full-output decode and, since round 3, the sampled decode ------------------------
// A shard is a complete handle over item rows [item0, item0 + I_local) and ALL users; b is replicated and stepped identically
// everywhere; the user node (Wu, Uu) is sharded by user range — the owner contributes a batch's rows to the first all-reduce.
// Two per-user sums cross shards in every batch — the input sum of the encode and the hidden gradient — so a batch runs in
// three phases with an all-reduce(sum) of [n_users x row_stride] floats (x shard_blocks for the first) after the first two.
int set_item_shard(cdae_hip_t* h, uint64_t item0, uint64_t num_items_global);      // before set_interactions
// users [u_begin, u_end) keep their private rows (Wu, Wu_ag, Uu, Uu_ag) on this shard (before set_interactions; default: all)
int set_item_shard_owner(cdae_hip_t* h, uint64_t u_begin, uint64_t u_end);
// sampled decode: the WHOLE train rows (global item ids) — read by the NEXT set_interactions call only, not kept by pointer
int set_item_shard_global(cdae_hip_t* h, const int64_t* row_ptr, const uint32_t* col);
// [users x row_stride] blocks in the input-sum all-reduce buffers (hsum_buf, ev_hsum_buf): the sums, then the gathered Wu / Uu rows
uint32_t shard_blocks(const cdae_hip_t* h);
// per user: (length of the WHOLE train row, position of the first local item in it) — the dropout stream is indexed by position
// in the whole row, so a shard must know where its slice sits (after set_interactions)
int set_item_shard_positions(cdae_hip_t* h, const uint32_t* len_and_first /* [2 U] */);
uint32_t row_stride(const cdae_hip_t* h);
int fs_prep(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t s0, uint32_t nb, uint32_t cidx);   // positives list + bitmap, prep stream
int fs_phase0(cdae_hip_t* h, uint64_t seed, uint32_t epoch, uint64_t s0, uint32_t nb, uint32_t cidx); // local input sums -> hsum_buf
int fs_phase1(cdae_hip_t* h, uint64_t s0, uint32_t nb);                                                // z, decode of the local items, local hg -> hg_buf
int fs_phase2(cdae_hip_t* h, uint64_t s0, uint32_t nb);                                                // hidden-layer steps, local row steps
float* hsum_buf(cdae_hip_t* h);
float* hg_buf(cdae_hip_t* h);
// evaluation over users [u0, u0 + nu), nu <= eval_chunk(): mode 0 = inference encode, 1 = loss corruption `cidx`
uint32_t eval_chunk();
int ev_phase0(cdae_hip_t* h, uint64_t u0, uint32_t nu, int mode, uint32_t cidx, uint64_t seed, uint32_t epoch);   // local input sums -> ev_hsum_buf
float* ev_hsum_buf(cdae_hip_t* h);
int ev_finish(cdae_hip_t* h, uint64_t u0, uint32_t nu, int mode);                                                  // z of the chunk
int ev_data_loss(cdae_hip_t* h, uint64_t u0, uint32_t nu, double* sum);                                            // += loss of the LOCAL positives
int ev_recommend(cdae_hip_t* h, uint64_t u0, uint32_t nu, uint32_t topk, uint32_t* ids, float* scores);            // local top-k (local ids) with scores, host arrays

}  // namespace cdae_internal
