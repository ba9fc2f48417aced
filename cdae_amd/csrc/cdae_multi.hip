// cdae_multi.hip — data parallelism behind the C ABI (include/cdae_hip.h, "multi-GPU" section).
//
// The reference trains one user after the other in one thread (cdae.hpp:136-146); Solver<CDAE>::train
// (solver-inl.hpp:51-55) calls train_one_iteration once per epoch.  Here an epoch may be split over several user shards:
//   * users are cut into contiguous ranges balanced by interactions; shard s holds its CSR rows and its private Wu / Wu_ag
//     rows (they never leave their GPU) and a replica of the shared block [W | W_ag | (V | V_ag) | b' | b'_ag | b | b_ag];
//   * every step each shard trains <= batch_users of its users, then the shards exchange the accumulated DELTA of the
//     shared block (cdae_kernels.hpp delta_pipe_kernel): synchronously (period 0) or pipelined — the all-reduce of one
//     period overlaps the next period's training and the peers' part is merged one period late;
//   * the all-reduce(sum) is RCCL on a library-owned communicator and stream (ncclCommInitAll for the shards of one
//     process, ncclCommInitRank when every rank is its own process: bench.py), or — shards that share ONE device, the
//     form the single-GPU tests and the accuracy envelope use — a fixed-order sum kernel over the peers' staged buffers.
// No torch, no Python in this path.  Accuracy of the data-parallel schedule: DESIGN.md §7 (it is NOT inside the +-0.002
// Recall@10 envelope of the single-GPU schedule; the table there says by how much).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "cdae_exchange_algebra.h"
#include "cdae_internal.hpp"

using cdae_internal::fail;

#define HIPCHK(expr)                                                                           \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)
#define NCCLCHK(expr)                                                                            \
  do {                                                                                           \
    ncclResult_t r__ = (expr);                                                                   \
    if (r__ != ncclSuccess)                                                                      \
      return fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, __LINE__);  \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

namespace {

// recv[i] = sum over the peers' staged buffers, in shard order (deterministic; the single-device stand-in for ncclAllReduce)
constexpr int MAX_LOCAL_PEERS = 16;
struct PeerPtrs { const float* p[MAX_LOCAL_PEERS]; };
struct PeerOuts { float* p[MAX_LOCAL_PEERS]; };
__global__ void __launch_bounds__(256)
local_sum_kernel(PeerPtrs peers, int n_peers, float* __restrict__ out, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n) {
    float4 acc = *reinterpret_cast<const float4*>(peers.p[0] + i);
    for (int r = 1; r < n_peers; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(peers.p[r] + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = acc;
  } else {
    for (size_t j = i; j < n; ++j) {
      float acc = peers.p[0][j];
      for (int r = 1; r < n_peers; ++r) acc += peers.p[r][j];
      out[j] = acc;
    }
  }
}
// the same sum, written IN PLACE to every peer's buffer by the launch itself (item-rows layout, logical shards: the all-reduce of
// [batch x row stride] floats was one sum kernel + one device-to-device copy per shard — sixteen copy launches per batch at eight
// shards, more stream time than the phases they separate).  An element is read from every peer and then written to every peer by
// ONE thread, so the in-place form has no hazard.
__global__ void __launch_bounds__(256)
local_allreduce_kernel(PeerOuts bufs, int n_peers, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n) {
    float4 acc = *reinterpret_cast<const float4*>(bufs.p[0] + i);
    for (int r = 1; r < n_peers; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(bufs.p[r] + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    for (int r = 0; r < n_peers; ++r) *reinterpret_cast<float4*>(bufs.p[r] + i) = acc;
  } else {
    for (size_t j = i; j < n; ++j) {
      float acc = bufs.p[0][j];
      for (int r = 1; r < n_peers; ++r) acc += bufs.p[r][j];
      for (int r = 0; r < n_peers; ++r) bufs.p[r][j] = acc;
    }
  }
}

// One process, one host thread per GPU (cdae_hip_multi_*): a thread whose shard fails inside an epoch must not leave its peers
// waiting in a collective that will never complete.  It raises `failed` and ABORTS every communicator of the group
// (ncclCommAbort is the one call RCCL allows from another thread while a rank is blocked) under the exclusive side of `mu`;
// the other threads hold the shared side only while they ISSUE a collective and check the flag first, so nobody issues on a
// communicator that is being aborted.
struct GroupGuard {
  std::shared_timed_mutex mu;
  std::atomic<int> failed{0};
};

// Exchange state of one handle (rank).
struct Exchange {
  GroupGuard* guard = nullptr;           // set while the shards of one process run an epoch on their own threads
  cdae_hip_t* h = nullptr;
  int world = 1, rank = 0;
  ncclComm_t comm = nullptr;             // RCCL communicator (nullptr: single rank, or a local group)
  bool owns_comm = false;
  std::vector<Exchange*> local;          // non-empty: the ranks of a single-device group, in shard order (includes this one)
  hipStream_t cstream = nullptr;         // the collective runs here, beside the training kernels
  bool owns_stream = true;
  hipEvent_t ev_staged = nullptr, ev_reduced = nullptr;
  int period = 0;                        // 0: synchronous exchange after every step; k >= 1: pipelined, every k steps
  uint64_t steps = 0;
  bool pending = false;                  // an all-reduce of a staged delta is in flight / not merged yet
  bool begun = false;
  // Synchronous exchange of a rank that owns its device (round 6): nothing can overlap the collective — the merge behind it is the next
  // thing the main stream does — so it is issued ON the main stream: stage -> all-reduce -> merge in order, no event hand-off to the
  // collective stream and back (two system-scope hops per step, ~10 us each).  Same buffers, same arithmetic.
  bool reduced_on_main = false;          // the pending all-reduce was issued on the main stream: the merge needs no event wait
  bool staged_sync = false;              // the pending delta was staged by delta_stage_sync (no snapshot): it must be merged by delta_merge_sync
};

void free_exchange(void* p) {
  Exchange* x = (Exchange*)p;
  if (!x) return;
  (void)hipSetDevice(cdae_internal::device_of(x->h));
  if (x->cstream) { (void)hipStreamSynchronize(x->cstream); if (x->owns_stream) (void)hipStreamDestroy(x->cstream); }
  if (x->ev_staged) (void)hipEventDestroy(x->ev_staged);
  if (x->ev_reduced) (void)hipEventDestroy(x->ev_reduced);
  if (x->comm && x->owns_comm) (void)ncclCommDestroy(x->comm);
  delete x;
}

int make_exchange(cdae_hip_t* h, Exchange** out) {
  if (cdae_internal::exchange_slot(h)) { *out = (Exchange*)cdae_internal::exchange_slot(h); return 0; }
  HIPCHK(hipSetDevice(cdae_internal::device_of(h)));
  Exchange* x = new Exchange();
  x->h = h;
  hipError_t e = hipSuccess;
  // The collective runs on the handle's SECOND stream (idle in the sampled path; the full-output path's b recurrence lives
  // there).  A fourth library stream is poison on this stack: with more than four hardware queues every kernel of the
  // step ran ~3x slower as soon as it existed and carried cross-stream waits (0.132 -> 0.395 ms per 256-user step with no
  // communicator at all; profiles/r02_exchange_streams.txt).  CDAE_XCHG_STREAM = own | main are developer switches.
  const char* sel = DEV_ENV("CDAE_XCHG_STREAM");
  if (sel && !std::strcmp(sel, "own")) e = hipStreamCreateWithFlags(&x->cstream, hipStreamNonBlocking);
  else if (sel && !std::strcmp(sel, "main")) { x->cstream = cdae_internal::main_stream(h); x->owns_stream = false; }
  else { x->cstream = cdae_internal::aux_stream(h); x->owns_stream = false; }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_staged, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_reduced, hipEventDisableTiming);
  if (e != hipSuccess) { free_exchange(x); return fail("exchange set-up failed: %s", hipGetErrorString(e)); }
  cdae_internal::exchange_slot(h) = x;
  cdae_internal::set_exchange_deleter(h, free_exchange);
  *out = x;
  return 0;
}

int begin_if_needed(Exchange* x) {
  if (x->begun) return 0;
  CHK(cdae_hip_delta_begin(x->h));       // base = current
  CHK(cdae_hip_delta_stage(x->h));       // allocates send / recv (stages a zero delta)
  x->begun = true;
  x->pending = false;
  return 0;
}

// the synchronous exchange of a rank that owns its device issues its collective on the main stream (Exchange::reduced_on_main)
static bool sync_on_main(const Exchange* x) { return x->period == 0 && x->local.empty() && !DEV_ENV("CDAE_XCHG_COLLECTIVE_STREAM"); }

// phase 1 of a boundary, on the handle's main stream: fold the previous period's peers in and / or stage this period's delta
int boundary_stage(Exchange* x, bool start_next) {
  HIPCHK(hipSetDevice(cdae_internal::device_of(x->h)));
  hipStream_t st = cdae_internal::main_stream(x->h);
  if (x->pending) {
    // the reduced buffer must be complete; in a local group nobody may restage (overwrite its send buffer) before every
    // peer's sum kernel has read it
    if (x->local.empty()) { if (!x->reduced_on_main) HIPCHK(hipStreamWaitEvent(st, x->ev_reduced, 0)); }
    else for (Exchange* p : x->local) HIPCHK(hipStreamWaitEvent(st, p->ev_reduced, 0));
  }
  if (x->pending && x->staged_sync) {                           // (a synchronous step's merge: directly behind its stage and all-reduce)
    CHK(cdae_internal::delta_merge_sync(x->h));
    x->staged_sync = false;
    if (start_next) CHK(cdae_hip_delta_stage(x->h));
  }
  else if (x->pending && start_next) CHK(cdae_hip_delta_merge_stage(x->h));
  else if (x->pending) CHK(cdae_hip_delta_merge(x->h));
  else if (start_next && sync_on_main(x) && !DEV_ENV("CDAE_XCHG_FULL_PASSES")) { CHK(cdae_internal::delta_stage_sync(x->h)); x->staged_sync = true; }
  else if (start_next) CHK(cdae_hip_delta_stage(x->h));
  x->pending = false;
  if (start_next && !sync_on_main(x)) HIPCHK(hipEventRecord(x->ev_staged, st));
  return 0;
}

// phase 2: the all-reduce(sum) of the staged deltas on the collective stream (every rank's phase 1 has been ENQUEUED:
// trivially true across processes / threads with RCCL, guaranteed by the caller's ordering in a local group)
int boundary_reduce(Exchange* x) {
  HIPCHK(hipSetDevice(cdae_internal::device_of(x->h)));
  const size_t n = cdae_internal::compact_count(x->h);
  float* recv = cdae_internal::recv_buf(x->h);
  if (!x->local.empty()) {
    PeerPtrs pp{};
    for (size_t r = 0; r < x->local.size(); ++r) {
      HIPCHK(hipStreamWaitEvent(x->cstream, x->local[r]->ev_staged, 0));
      pp.p[r] = cdae_internal::send_buf(x->local[r]->h);
    }
    hipLaunchKernelGGL(local_sum_kernel, dim3((unsigned)((n / 4 + 1 + 255) / 256)), dim3(256), 0, x->cstream, pp, (int)x->local.size(), recv, n);
    HIPCHK(hipGetLastError());
  } else {
    const bool on_main = sync_on_main(x);      // (CDAE_XCHG_COLLECTIVE_STREAM, developer switch: round 5's hand-off to the collective stream)
    hipStream_t cs = on_main ? cdae_internal::main_stream(x->h) : x->cstream;
    if (!on_main) HIPCHK(hipStreamWaitEvent(cs, x->ev_staged, 0));
    if (x->guard) {
      std::shared_lock<std::shared_timed_mutex> lk(x->guard->mu);
      if (x->guard->failed.load()) return fail("a peer shard failed: epoch abandoned");
      NCCLCHK(ncclAllReduce(recv, recv, n, ncclFloat32, ncclSum, x->comm, cs));
    } else if (x->comm) {
      NCCLCHK(ncclAllReduce(recv, recv, n, ncclFloat32, ncclSum, x->comm, cs));   // (one rank: identity, same stream semantics)
    }
    x->reduced_on_main = on_main;
    if (on_main) { x->pending = true; return 0; }
  }
  HIPCHK(hipEventRecord(x->ev_reduced, x->cstream));
  x->pending = true;
  return 0;
}

// a whole boundary of ONE rank that does not share its device with peers (RCCL or single rank)
int boundary(Exchange* x, bool start_next) {
  CHK(boundary_stage(x, start_next));
  if (start_next) CHK(boundary_reduce(x));
  return 0;
}

int step_single(Exchange* x) {
  CHK(begin_if_needed(x));
  x->steps++;
  if (x->period == 0) { CHK(boundary(x, true)); CHK(boundary(x, false)); }
  else if (x->steps % (uint64_t)x->period == 0) CHK(boundary(x, true));
  return 0;
}
int flush_single(Exchange* x) {
  CHK(begin_if_needed(x));
  if (x->period != 0 || x->pending) { CHK(boundary(x, true)); CHK(boundary(x, false)); }
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// per-handle communicator and exchange schedule (one process per GPU: bench.py --gpus N)
extern "C" {

int cdae_hip_comm_unique_id(void* out, size_t bytes) {
  if (!out || bytes < sizeof(ncclUniqueId)) return fail("unique id buffer must hold %zu bytes", sizeof(ncclUniqueId));
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  std::memset(out, 0, bytes);
  std::memcpy(out, &id, sizeof id);
  return 0;
}

int cdae_hip_comm_init_rank(cdae_hip_t* h, int world_size, int rank, const void* unique_id, size_t bytes) {
  if (!h) return fail("null handle");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail("bad rank %d of %d", rank, world_size);
  if (!unique_id || bytes < sizeof(ncclUniqueId)) return fail("unique id must hold %zu bytes", sizeof(ncclUniqueId));
  Exchange* x = nullptr;
  CHK(make_exchange(h, &x));
  if (x->comm || !x->local.empty()) return fail("this handle already has a communicator");
  HIPCHK(hipSetDevice(cdae_internal::device_of(h)));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  NCCLCHK(ncclCommInitRank(&x->comm, world_size, id, rank));
  x->owns_comm = true;
  x->world = world_size; x->rank = rank;
  return 0;
}

int cdae_hip_exchange_configure(cdae_hip_t* h, int period) {
  if (!h) return fail("null handle");
  if (period < 0) return fail("exchange period must be >= 0");
  Exchange* x = nullptr;
  CHK(make_exchange(h, &x));
  if (!x->local.empty()) return fail("this handle belongs to a multi-shard group: use cdae_hip_multi_set_exchange");
  if (x->pending) CHK(flush_single(x));
  x->period = period;
  x->steps = 0;
  return 0;
}

int cdae_hip_exchange_step(cdae_hip_t* h) {
  if (!h || !cdae_internal::ready(h)) return fail("set_interactions must be called first");
  Exchange* x = nullptr;
  CHK(make_exchange(h, &x));
  if (!x->local.empty()) return fail("this handle belongs to a multi-shard group");
  return step_single(x);
}

int cdae_hip_exchange_flush(cdae_hip_t* h) {
  if (!h || !cdae_internal::ready(h)) return fail("set_interactions must be called first");
  Exchange* x = nullptr;
  CHK(make_exchange(h, &x));
  if (!x->local.empty()) return fail("this handle belongs to a multi-shard group");
  return flush_single(x);
}

int cdae_hip_exchange_time_all_reduce(cdae_hip_t* h, int repeats, double* seconds) {
  if (!h || !cdae_internal::ready(h) || !seconds) return fail("bad argument");
  if (repeats < 1) repeats = 1;
  Exchange* x = nullptr;
  CHK(make_exchange(h, &x));
  if (!x->local.empty()) return fail("this handle belongs to a multi-shard group");
  CHK(flush_single(x));
  CHK(cdae_hip_synchronize(h));
  HIPCHK(hipSetDevice(cdae_internal::device_of(h)));
  HIPCHK(hipStreamSynchronize(x->cstream));
  *seconds = 0.;
  if (!x->comm) return 0;
  const size_t n = cdae_internal::compact_count(h);
  float* recv = cdae_internal::recv_buf(h);
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) NCCLCHK(ncclAllReduce(recv, recv, n, ncclFloat32, ncclSum, x->comm, x->cstream));
  HIPCHK(hipEventRecord(a, x->cstream));
  for (int i = 0; i < repeats; ++i) NCCLCHK(ncclAllReduce(recv, recv, n, ncclFloat32, ncclSum, x->comm, x->cstream));
  HIPCHK(hipEventRecord(b, x->cstream));
  HIPCHK(hipEventSynchronize(b));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *seconds = 1e-3 * ms / repeats;
  // the timing runs summed garbage into the receive buffer: restart from a fresh base with a zero staged delta
  x->begun = false; x->pending = false; x->staged_sync = false; x->steps = 0;
  CHK(begin_if_needed(x));
  CHK(cdae_hip_synchronize(h));
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// several shards behind one handle (one process): Solver<CDAE>::train on N GPUs
struct cdae_hip_multi {
  cdae_hip_config cfg{};
  std::vector<int> devices;
  std::vector<cdae_hip_t*> shard;
  std::vector<uint64_t> cut;             // users of shard s: [cut[s], cut[s+1])
  std::vector<ncclComm_t> comms;         // ncclCommInitAll (distinct devices)
  bool single_device = false;            // all shards on one device: in-process sum instead of RCCL, one driver thread
  uint64_t U = 0, I = 0;
  int period = 0;
  uint32_t B = 0;
  // cdae_hip_multi_set_schedule (user-sharded layout): combine rule of the exchange, users per shard of an exchanged step
  // (0 = B), epochs (fractions allowed) that start with the relay of the single-GPU schedule
  uint32_t combine = CDAE_COMBINE_SUM, sync_B = 0;
  double relay_epochs = 0.0;
  // CDAE_LAYOUT_ITEM_ROWS: the shards cut the ITEM rows (W / W_ag / V / V_ag / b' and the decode over them); icut = item ranges
  uint32_t layout = 0;
  std::vector<uint64_t> icut;
  float* d_tmp = nullptr; size_t tmp_cap = 0;      // single-device all-reduce: the sum before it is copied back to every shard
  hipEvent_t ev_done = nullptr;
  bool one_rank_comms = false;           // (developer build) every shard on its own one-rank communicator, thread-per-shard host logic
  int fail_shard = -1; uint64_t fail_step = 0;   // (developer build) forced failure of one shard's thread at a step
  bool comm_aborted = false;             // a shard's thread failed inside an epoch: the communicators were aborted so that its peers
                                         // could unwind; the handle answers every later call with that error
  std::string abort_reason;
  GroupGuard guard;
  // called by the failing shard's thread
  // `failed` goes up FIRST (issuers check it under the shared side before they enqueue a collective).  The exclusive side is then
  // taken with a TIME LIMIT: a peer may be blocked INSIDE ncclAllReduce holding the shared side (RCCL connects lazily on the first
  // collective and waits there for the rank that just failed) — waiting for it would be waiting for ourselves, and the abort is
  // exactly what releases it, so after the limit the abort goes ahead without the lock (ncclCommAbort is the one call RCCL allows
  // from another thread while a rank is blocked).  ncclCommAbort also frees the communicator: nothing is left to destroy.
  // Known residual window (error path only; round-5 advice): a peer that has passed the `failed` check under the shared side and is then
  // descheduled for longer than the limit BEFORE it enters ncclAllReduce would call into a freed communicator.  RCCL offers no abort that
  // keeps the object alive, and waiting without a limit is the deadlock described above; the limit is 200 ms against the nanoseconds
  // between the check and the call.
  void give_up() {
    if (guard.failed.exchange(1)) return;
    std::unique_lock<std::shared_timed_mutex> lk(guard.mu, std::defer_lock);
    (void)lk.try_lock_for(std::chrono::milliseconds(200));
    for (ncclComm_t& c : comms) if (c) { (void)ncclCommAbort(c); c = nullptr; }
    comm_aborted = true;
  }
};

namespace {

Exchange* xof(cdae_hip_t* h) { return (Exchange*)cdae_internal::exchange_slot(h); }

// one step of every shard's epoch: shard s trains users [a, b) of its own
// `first[s]`: users of shard s already trained in this epoch (the relay part of cdae_hip_multi_set_schedule); the plan covers the rest
struct StepPlan { uint64_t steps; std::vector<uint64_t> per, first; uint64_t t0 = 0, t1 = ~0ull; /* steps [t0, min(t1, steps)) are run (cdae_hip_multi_train_steps) */ };
StepPlan plan_of(const cdae_hip_multi* m, const std::vector<uint64_t>& first) {
  StepPlan p;
  p.first = first;
  const uint64_t B = m->sync_B ? std::min<uint64_t>(m->sync_B, m->B) : m->B;
  uint64_t longest = 0;
  for (size_t s = 0; s < m->shard.size(); ++s) longest = std::max(longest, m->cut[s + 1] - m->cut[s] - first[s]);
  p.steps = std::max<uint64_t>(1, (longest + B - 1) / B);
  for (size_t s = 0; s < m->shard.size(); ++s) p.per.push_back((m->cut[s + 1] - m->cut[s] - first[s] + p.steps - 1) / p.steps);   // all shards finish together
  return p;
}

// which shard's error to report when several threads of a group came back with one: the first that is a CAUSE — the peers of a
// failing shard return "a peer shard failed" once the group has been abandoned, and which of them has the lower index is chance
int first_cause(const std::vector<int>& rc, const std::vector<std::string>& err) {
  int any = -1;
  for (size_t s = 0; s < rc.size(); ++s)
    if (rc[s]) {
      if (err[s].find("a peer shard failed") == std::string::npos) return (int)s;
      if (any < 0) any = (int)s;
    }
  return any;
}

// the epoch of ONE shard that owns its device (RCCL group): runs on its own host thread
int shard_epoch(cdae_hip_multi* m, size_t s, const StepPlan& pl, uint64_t seed, uint32_t epoch) {
  cdae_hip_t* h = m->shard[s];
  Exchange* x = xof(h);
  const uint64_t n = m->cut[s + 1] - m->cut[s], f = pl.first[s];
  x->period = m->period;
  for (uint64_t t = pl.t0; t < std::min(pl.t1, pl.steps); ++t) {
    const uint64_t a = std::min(n, f + t * pl.per[s]), b = std::min(n, f + (t + 1) * pl.per[s]);
    if ((int)s == m->fail_shard && t == m->fail_step) return fail("forced failure of shard %zu at step %llu (developer build)", s, (unsigned long long)t);
    if (b > a) CHK(cdae_hip_enqueue_users(h, seed, epoch, a, b));
    const uint64_t a2 = std::min(n, f + (t + 1) * pl.per[s]), b2 = std::min(n, f + (t + 2) * pl.per[s]);
    if (b2 > a2) CHK(cdae_hip_prefetch_users(h, seed, epoch, a2, b2));
    CHK(step_single(x));                                   // every shard takes every step: the collective needs all ranks
  }
  CHK(flush_single(x));                                    // the epoch ends with identical shared parameters everywhere
  return cdae_hip_synchronize(h);
}

// all shards on ONE device: one driver thread, boundaries in lockstep (phase 1 of every shard, then phase 2 of every shard)
int local_boundary(cdae_hip_multi* m, bool start_next) {
  for (cdae_hip_t* h : m->shard) CHK(boundary_stage(xof(h), start_next));
  if (start_next) for (cdae_hip_t* h : m->shard) CHK(boundary_reduce(xof(h)));
  return 0;
}
int local_epoch(cdae_hip_multi* m, const StepPlan& pl, uint64_t seed, uint32_t epoch) {
  for (cdae_hip_t* h : m->shard) CHK(begin_if_needed(xof(h)));
  uint64_t steps = 0;
  for (uint64_t t = pl.t0; t < std::min(pl.t1, pl.steps); ++t) {
    for (size_t s = 0; s < m->shard.size(); ++s) {
      const uint64_t n = m->cut[s + 1] - m->cut[s], f = pl.first[s];
      const uint64_t a = std::min(n, f + t * pl.per[s]), b = std::min(n, f + (t + 1) * pl.per[s]);
      if (b > a) CHK(cdae_hip_enqueue_users(m->shard[s], seed, epoch, a, b));
    }
    ++steps;
    if (m->period == 0) { CHK(local_boundary(m, true)); CHK(local_boundary(m, false)); }
    else if (steps % (uint64_t)m->period == 0) CHK(local_boundary(m, true));
  }
  if (m->period != 0 || xof(m->shard[0])->pending) { CHK(local_boundary(m, true)); CHK(local_boundary(m, false)); }
  for (cdae_hip_t* h : m->shard) CHK(cdae_hip_synchronize(h));
  return 0;
}

// RELAY part of an epoch (cdae_hip_multi_set_schedule): global users [0, R) on the single-GPU schedule, by the shards that hold them,
// one after the other; the shared block travels with the training (shard s starts from what shard s - 1 ended with) and ends up on
// every shard.  first[s] = users of shard s trained here.  Replicas agree on entry (an epoch ends with a flush).
int relay_part(cdae_hip_multi* m, uint64_t seed, uint32_t epoch, uint64_t R, std::vector<uint64_t>& first, cdae_hip_stats* sum) {
  const size_t S = m->shard.size();
  first.assign(S, 0);
  if (R == 0) return 0;
  size_t last = 0;
  for (size_t s = 0; s < S && m->cut[s] < R; ++s) {
    const uint64_t n = std::min(R, m->cut[s + 1]) - m->cut[s];
    if (s > 0) CHK(cdae_internal::adopt_shared_block(m->shard[s], m->shard[s - 1]));
    cdae_hip_stats st;
    CHK(cdae_hip_train_users(m->shard[s], seed, epoch, 0, n, &st));     // (synchronises: the next shard copies a finished block)
    sum->users += st.users; sum->examples += st.examples; sum->batches += st.batches;
    first[s] = n;
    last = s;
  }
  for (size_t s = 0; s < S; ++s) if (s != last) CHK(cdae_internal::adopt_shared_block(m->shard[s], m->shard[last]));
  for (cdae_hip_t* h : m->shard) {
    CHK(cdae_hip_synchronize(h));
    Exchange* x = xof(h);
    if (x) { x->begun = false; x->pending = false; x->staged_sync = false; x->steps = 0; }      // the exchange restarts from the relayed parameters
  }
  return 0;
}

// all-reduce(sum) of one [n]-float buffer per shard, stream-ordered on the shards' main streams (item-sharded layout: the
// phases of a batch are sequential anyway, nothing to overlap).  Distinct devices: RCCL, one group call from this thread.
// One device: sum kernel into a scratch buffer on shard 0's stream, copied back to every shard.
int all_reduce_bufs(cdae_hip_multi* m, const std::vector<float*>& bufs, size_t n) {
  const size_t S = m->shard.size();
  if (S == 1) return 0;
  if (!m->single_device) {
    NCCLCHK(ncclGroupStart());
    for (size_t s = 0; s < S; ++s) {
      ncclResult_t r = ncclAllReduce(bufs[s], bufs[s], n, ncclFloat32, ncclSum, m->comms[s], cdae_internal::main_stream(m->shard[s]));
      if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail("ncclAllReduce failed: %s", ncclGetErrorString(r)); }
    }
    NCCLCHK(ncclGroupEnd());
    return 0;
  }
  HIPCHK(hipSetDevice(m->devices[0]));
  if (!m->ev_done) HIPCHK(hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming));
  hipStream_t s0 = cdae_internal::main_stream(m->shard[0]);
  PeerOuts pp{};
  for (size_t s = 0; s < S; ++s) {
    Exchange* x = xof(m->shard[s]);
    HIPCHK(hipEventRecord(x->ev_staged, cdae_internal::main_stream(m->shard[s])));
    HIPCHK(hipStreamWaitEvent(s0, x->ev_staged, 0));
    pp.p[s] = bufs[s];
  }
  hipLaunchKernelGGL(local_allreduce_kernel, dim3((unsigned)((n / 4 + 1 + 255) / 256)), dim3(256), 0, s0, pp, (int)S, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(m->ev_done, s0));
  for (size_t s = 1; s < S; ++s) HIPCHK(hipStreamWaitEvent(cdae_internal::main_stream(m->shard[s]), m->ev_done, 0));
  return 0;
}

// one epoch of the item-sharded layout: every batch of users runs on EVERY shard (each over its item rows) in three phases
int item_epoch(cdae_hip_multi* m, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end) {
  const size_t S = m->shard.size();
  const uint32_t Kp = cdae_internal::row_stride(m->shard[0]), blocks = cdae_internal::shard_blocks(m->shard[0]);
  struct Bt { uint64_t s0; uint32_t nb, c; };
  std::vector<Bt> plan;
  for (uint64_t s0 = u_begin; s0 < u_end; s0 += m->B)
    for (uint32_t c = 0; c < m->cfg.num_corruptions; ++c) plan.push_back(Bt{s0, (uint32_t)std::min<uint64_t>(m->B, u_end - s0), c});
  if (plan.empty()) return 0;
  std::vector<float*> hs(S), hg(S);
  for (size_t s = 0; s < S; ++s) { hs[s] = cdae_internal::hsum_buf(m->shard[s]); hg[s] = cdae_internal::hg_buf(m->shard[s]); }
  if (S > 1 && !m->single_device) {
    // one shard per GPU: one host thread per shard, like the user-sharded layout (a batch is ~15 launches per shard; from one
    // thread the HOST would pace N GPUs).  Every thread issues the same sequence — phases on its shard's main stream, the two
    // all-reduces on its own communicator of the ncclCommInitAll group — so the collectives pair up without a group call.
    // (a failing shard's thread aborts the group's communicators so that its peers unwind: GroupGuard above)
    std::vector<int> rc(S, 0);
    std::vector<std::string> err(S);
    std::vector<std::thread> th;
    auto all_reduce = [&](size_t s, float* buf, size_t n, hipStream_t st) -> int {
      std::shared_lock<std::shared_timed_mutex> lk(m->guard.mu);
      if (m->guard.failed.load()) return fail("item shard %zu: a peer shard failed, epoch abandoned", s);
      NCCLCHK(ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, m->comms[s], st));
      return 0;
    };
    auto run = [&](size_t s) -> int {
      cdae_hip_t* h = m->shard[s];
      HIPCHK(hipSetDevice(m->devices[s]));
      hipStream_t st = cdae_internal::main_stream(h);
      CHK(cdae_internal::fs_prep(h, seed, epoch, plan[0].s0, plan[0].nb, plan[0].c));
      for (size_t t = 0; t < plan.size(); ++t) {
        const Bt& b = plan[t];
        if ((int)s == m->fail_shard && t == m->fail_step) return fail("forced failure of item shard %zu at batch %zu (developer build)", s, t);
        CHK(cdae_internal::fs_phase0(h, seed, epoch, b.s0, b.nb, b.c));
        CHK(all_reduce(s, hs[s], (size_t)b.nb * Kp * blocks, st));
        if (t + 1 < plan.size()) CHK(cdae_internal::fs_prep(h, seed, epoch, plan[t + 1].s0, plan[t + 1].nb, plan[t + 1].c));
        CHK(cdae_internal::fs_phase1(h, b.s0, b.nb));
        CHK(all_reduce(s, hg[s], (size_t)b.nb * Kp, st));
        CHK(cdae_internal::fs_phase2(h, b.s0, b.nb));
      }
      return cdae_hip_synchronize(h);
    };
    for (size_t s = 0; s < S; ++s)
      th.emplace_back([&, s] {
        rc[s] = run(s);
        if (rc[s]) { err[s] = cdae_hip_last_error(); m->give_up(); }
      });
    for (std::thread& t : th) t.join();
    if (const int s = first_cause(rc, err); s >= 0) {
      if (m->comm_aborted) m->abort_reason = err[s];
      return fail("item shard %d: %s", s, err[s].c_str());
    }
    return 0;
  }
  for (cdae_hip_t* h : m->shard) CHK(cdae_internal::fs_prep(h, seed, epoch, plan[0].s0, plan[0].nb, plan[0].c));
  for (size_t t = 0; t < plan.size(); ++t) {
    const Bt& b = plan[t];
    for (cdae_hip_t* h : m->shard) CHK(cdae_internal::fs_phase0(h, seed, epoch, b.s0, b.nb, b.c));
    CHK(all_reduce_bufs(m, hs, (size_t)b.nb * Kp * blocks));              // input sums over ALL item rows (+ the owners' Wu / Uu rows of the batch)
    if (t + 1 < plan.size())                                             // the next batch's example lists: prep streams, beside the decode
      for (cdae_hip_t* h : m->shard) CHK(cdae_internal::fs_prep(h, seed, epoch, plan[t + 1].s0, plan[t + 1].nb, plan[t + 1].c));
    for (cdae_hip_t* h : m->shard) CHK(cdae_internal::fs_phase1(h, b.s0, b.nb));
    CHK(all_reduce_bufs(m, hg, (size_t)b.nb * Kp));                       // hidden gradient over ALL item rows
    for (cdae_hip_t* h : m->shard) CHK(cdae_internal::fs_phase2(h, b.s0, b.nb));
  }
  for (cdae_hip_t* h : m->shard) CHK(cdae_hip_synchronize(h));
  return 0;
}

// z of users [u0, u0 + nu) on every shard (evaluation): local input sums, all-reduce, activation
int item_encode_chunk(cdae_hip_multi* m, uint64_t u0, uint32_t nu, int mode, uint32_t cidx, uint64_t seed, uint32_t epoch) {
  const size_t S = m->shard.size();
  const uint32_t Kp = cdae_internal::row_stride(m->shard[0]);
  std::vector<float*> bufs(S);
  for (size_t s = 0; s < S; ++s) {
    CHK(cdae_internal::ev_phase0(m->shard[s], u0, nu, mode, cidx, seed, epoch));
    bufs[s] = cdae_internal::ev_hsum_buf(m->shard[s]);
  }
  CHK(all_reduce_bufs(m, bufs, (size_t)nu * Kp * cdae_internal::shard_blocks(m->shard[0])));
  for (cdae_hip_t* h : m->shard) CHK(cdae_internal::ev_finish(h, u0, nu, mode));
  return 0;
}

// the communicators of a group of shards that each own a device (both layouts): ncclCommInitAll — or, in the developer build's
// one-rank mode, a communicator of world size 1 per shard
int init_group_comms(cdae_hip_multi* m) {
  const size_t S = m->shard.size();
  if (!m->comms.empty()) return 0;
  m->comms.assign(S, nullptr);
  if (m->one_rank_comms) {
    for (size_t s = 0; s < S; ++s) {
      ncclUniqueId id;
      NCCLCHK(ncclGetUniqueId(&id));
      HIPCHK(hipSetDevice(m->devices[s]));
      NCCLCHK(ncclCommInitRank(&m->comms[s], 1, id, 0));
    }
    return 0;
  }
  NCCLCHK(ncclCommInitAll(m->comms.data(), (int)S, m->devices.data()));
  return 0;
}

int check_multi(const cdae_hip_multi* m, bool need_data) {
  if (!m) return fail("null multi handle");
  if (m->comm_aborted) return fail("the communicators of this handle were aborted after a shard failed (%s): destroy it", m->abort_reason.c_str());
  if (need_data && m->U == 0) return fail("cdae_hip_multi_set_interactions must be called first");
  return 0;
}

}  // namespace

extern "C" {

int cdae_hip_multi_create(const cdae_hip_config* cfg, int n_shards, const int* device_ids, cdae_hip_multi_t** out) {
  if (!cfg || !out || !device_ids) return fail("null argument");
  if (n_shards < 1 || n_shards > MAX_LOCAL_PEERS) return fail("n_shards must be in [1, %d]", MAX_LOCAL_PEERS);
  std::vector<int> devs(device_ids, device_ids + n_shards), uniq(devs);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  bool single = uniq.size() == 1;
  // DEVELOPER build only (tests/test_gpu_multi.py): equal device ids, but driven like distinct GPUs — one host thread per shard, each
  // shard on a ONE-RANK communicator of its own (RCCL refuses two ranks of one communicator on one device).  The collectives are then
  // identities, so every shard trains as if it were alone: what runs is the thread-per-shard host logic, the guard and the RCCL calls.
  const bool one_rank_comms = single && n_shards > 1 && DEV_ENV("CDAE_MULTI_ONE_RANK_COMMS") != nullptr;
  if (one_rank_comms) single = false;
  if (!single && !one_rank_comms && (int)uniq.size() != n_shards)
    return fail("device_ids must be all distinct (one shard per GPU, RCCL) or all equal (logical shards of one GPU)");
  std::unique_ptr<cdae_hip_multi> m(new cdae_hip_multi());
  m->cfg = *cfg; m->devices = devs; m->single_device = single; m->one_rank_comms = one_rank_comms;
  if (const char* f = DEV_ENV("CDAE_MULTI_FAIL_AT")) {        // DEVELOPER build only: "shard:step" — that shard's thread fails there (GroupGuard test)
    m->fail_shard = std::atoi(f);
    if (const char* c = std::strchr(f, ':')) m->fail_step = std::strtoull(c + 1, nullptr, 10);
  }
  for (int s = 0; s < n_shards; ++s) {
    cdae_hip_t* h = nullptr;
    int rc = cdae_hip_create(cfg, devs[s], &h);
    if (rc) { for (cdae_hip_t* q : m->shard) cdae_hip_destroy(q); return rc; }
    // shards that share a device run their steps side by side on it: the fused decode + gather launch must not (cdae_hip_set_decode_fused)
    if ((int)uniq.size() != n_shards) (void)cdae_hip_set_decode_fused(h, 0);
    m->shard.push_back(h);
  }
  *out = m.release();
  return 0;
}

int cdae_hip_multi_destroy(cdae_hip_multi_t* m) {
  if (!m) return 0;
  for (cdae_hip_t* h : m->shard) cdae_hip_destroy(h);      // (exchange state goes with the handle; communicators below)
  for (ncclComm_t c : m->comms) if (c) (void)ncclCommDestroy(c);
  if (m->d_tmp) { (void)hipSetDevice(m->devices[0]); (void)hipFree(m->d_tmp); }
  if (m->ev_done) (void)hipEventDestroy(m->ev_done);
  delete m;
  return 0;
}

int cdae_hip_multi_num_shards(const cdae_hip_multi_t* m) { return m ? (int)m->shard.size() : 0; }

int cdae_hip_multi_shard(cdae_hip_multi_t* m, int shard, cdae_hip_t** handle, uint64_t* u_begin, uint64_t* u_end) {
  CHK(check_multi(m, false));
  if (shard < 0 || shard >= (int)m->shard.size()) return fail("shard %d out of range", shard);
  if (handle) *handle = m->shard[shard];
  const std::vector<uint64_t>& c = m->layout == CDAE_LAYOUT_ITEM_ROWS ? m->icut : m->cut;    // item ranges in the item-rows layout
  if (u_begin) *u_begin = c.size() ? c[shard] : 0;
  if (u_end) *u_end = c.size() ? c[shard + 1] : 0;
  return 0;
}

int cdae_hip_multi_set_layout(cdae_hip_multi_t* m, uint32_t layout) {
  CHK(check_multi(m, false));
  if (layout > CDAE_LAYOUT_ITEM_ROWS) return fail("unknown layout %u", layout);
  if (m->U) return fail("set the layout before cdae_hip_multi_set_interactions");
  m->layout = layout;
  return 0;
}

static int set_interactions_item_rows(cdae_hip_multi* m, uint64_t U, uint64_t I, const int64_t* row_ptr, const uint32_t* col) {
  const size_t S = m->shard.size();
  if (I < S) return fail("%llu items cannot be split into %zu shards", (unsigned long long)I, S);
  // contiguous item ranges balanced by interactions (column counts)
  std::vector<uint64_t> cnt(I + 1, 0);
  const int64_t nnz = row_ptr[U];
  for (int64_t p = 0; p < nnz; ++p) {
    if (col[p] >= I) return fail("item id %u out of range at position %lld", col[p], (long long)p);
    cnt[col[p] + 1]++;
  }
  for (uint64_t i = 0; i < I; ++i) cnt[i + 1] += cnt[i];
  m->icut.assign(S + 1, 0);
  {
    std::vector<int64_t> pre(cnt.begin(), cnt.end());
    cdae_xa::balanced_cuts(pre.data(), I, S, true, m->icut.data());
  }
  // the user node (Wu, Uu) is sharded by USER: contiguous user ranges balanced by interactions; a shard may own no user
  m->cut.assign(S + 1, 0);
  cdae_xa::balanced_cuts(row_ptr, U, S, false, m->cut.data());
  std::vector<int64_t> rp(U + 1);
  std::vector<uint32_t> lc, pos(2 * U);
  for (size_t s = 0; s < S; ++s) {
    const uint32_t i0 = (uint32_t)m->icut[s], i1 = (uint32_t)m->icut[s + 1];
    lc.clear();
    rp[0] = 0;
    for (uint64_t u = 0; u < U; ++u) {                     // rows are ascending: the slice is one sub-range
      const uint32_t* a = col + row_ptr[u];
      const uint32_t* b = col + row_ptr[u + 1];
      if (b <= a) return fail("user %llu has no training item (the reference CHECK-fails too, cdae.hpp:139)", (unsigned long long)u);
      const uint32_t* lo = std::lower_bound(a, b, i0);
      const uint32_t* hi = std::lower_bound(lo, b, i1);
      for (const uint32_t* q = lo; q < hi; ++q) lc.push_back(*q - i0);
      rp[u + 1] = (int64_t)lc.size();
      pos[2 * u] = (uint32_t)(b - a); pos[2 * u + 1] = (uint32_t)(lo - a);
    }
    CHK(cdae_internal::set_item_shard(m->shard[s], i0, I));
    CHK(cdae_internal::set_item_shard_owner(m->shard[s], m->cut[s], m->cut[s + 1]));
    if (!m->cfg.full_output) CHK(cdae_internal::set_item_shard_global(m->shard[s], row_ptr, col));   // sampled decode: negatives against the whole rows
    static const uint32_t none = 0;
    CHK(cdae_hip_set_interactions(m->shard[s], U, i1 - i0, rp.data(), lc.empty() ? &none : lc.data()));
    CHK(cdae_internal::set_item_shard_positions(m->shard[s], pos.data()));
  }
  m->U = U; m->I = I;
  m->B = cdae_internal::batch_users(m->shard[0]);
  std::vector<Exchange*> xs(S, nullptr);
  for (size_t s = 0; s < S; ++s) CHK(make_exchange(m->shard[s], &xs[s]));
  if (S > 1 && !m->single_device) CHK(init_group_comms(m));
  return 0;
}

int cdae_hip_multi_set_interactions(cdae_hip_multi_t* m, uint64_t U, uint64_t I, const int64_t* row_ptr, const uint32_t* col) {
  CHK(check_multi(m, false));
  if (!row_ptr || U == 0) return fail("bad argument");
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) return set_interactions_item_rows(m, U, I, row_ptr, col);
  const size_t S = m->shard.size();
  if (U < S) return fail("%llu users cannot be split into %zu shards", (unsigned long long)U, S);
  // contiguous user ranges balanced by interactions; every shard gets at least one user
  m->cut.assign(S + 1, 0);
  cdae_xa::balanced_cuts(row_ptr, U, S, true, m->cut.data());   // first user whose prefix reaches s/S of the interactions
  std::vector<int64_t> rp;
  for (size_t s = 0; s < S; ++s) {
    const uint64_t a = m->cut[s], b = m->cut[s + 1];
    rp.assign(row_ptr + a, row_ptr + b + 1);
    const int64_t base = rp[0];
    for (int64_t& v : rp) v -= base;
    CHK(cdae_hip_set_interactions(m->shard[s], b - a, I, rp.data(), col + base));
    CHK(cdae_hip_set_user_id_offset(m->shard[s], a));      // random streams and Wu init keyed by GLOBAL user id
  }
  m->U = U; m->I = I;
  m->B = cdae_internal::batch_users(m->shard[0]);
  for (cdae_hip_t* h : m->shard) m->B = std::min(m->B, cdae_internal::batch_users(h));
  // exchange state: a local group (one device) or one RCCL communicator per shard
  std::vector<Exchange*> xs(S, nullptr);
  for (size_t s = 0; s < S; ++s) CHK(make_exchange(m->shard[s], &xs[s]));
  if (S > 1 && m->single_device) {
    for (Exchange* x : xs) { x->local = xs; x->world = (int)S; }
    for (size_t s = 0; s < S; ++s) xs[s]->rank = (int)s;
  } else if (S > 1) {
    CHK(init_group_comms(m));
    for (size_t s = 0; s < S; ++s) { xs[s]->comm = m->comms[s]; xs[s]->owns_comm = false; xs[s]->world = (int)S; xs[s]->rank = (int)s; }
  }
  for (Exchange* x : xs) { x->begun = false; x->pending = false; x->staged_sync = false; x->steps = 0; }
  return 0;
}

int cdae_hip_multi_init_params(cdae_hip_multi_t* m, uint64_t seed) {
  CHK(check_multi(m, true));
  for (cdae_hip_t* h : m->shard) CHK(cdae_hip_init_params(h, seed));     // identical shared blocks; Wu rows by global user id
  for (cdae_hip_t* h : m->shard) { Exchange* x = xof(h); if (x) { x->begun = false; x->pending = false; x->staged_sync = false; x->steps = 0; } }
  return 0;
}

int cdae_hip_multi_set_exchange(cdae_hip_multi_t* m, int period) {
  CHK(check_multi(m, false));
  if (period < 0) return fail("exchange period must be >= 0");
  m->period = period;
  return 0;
}

int cdae_hip_multi_set_schedule(cdae_hip_multi_t* m, const cdae_multi_schedule* sc) {
  CHK(check_multi(m, false));
  if (!sc) return fail("null schedule");
  if (sc->period < 0) return fail("exchange period must be >= 0");
  if (sc->combine > CDAE_COMBINE_GLOBAL_ACC) return fail("unknown combine rule %u", sc->combine);
  if (!(sc->relay_epochs >= 0.0)) return fail("relay_epochs must be >= 0");
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS && (sc->combine || sc->sync_batch_users || sc->relay_epochs > 0.0))
    return fail("the item-rows layout IS the single-GPU schedule: it has no relay, no combine rule and no exchanged steps to size");
  m->period = sc->period; m->combine = sc->combine; m->sync_B = sc->sync_batch_users; m->relay_epochs = sc->relay_epochs;
  for (cdae_hip_t* h : m->shard) CHK(cdae_hip_delta_set_combine(h, m->combine));
  return 0;
}

int cdae_hip_multi_train_epoch(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, cdae_hip_stats* stats) {
  CHK(check_multi(m, true));
  return cdae_hip_multi_train_users(m, seed, epoch, 0, m->U, stats);
}

int cdae_hip_multi_train_users(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, uint64_t u_begin, uint64_t u_end, cdae_hip_stats* stats) {
  CHK(check_multi(m, true));
  if (u_begin > u_end || u_end > m->U) return fail("bad user range");
  if (m->layout != CDAE_LAYOUT_ITEM_ROWS && (u_begin != 0 || u_end != m->U))
    return fail("the user-sharded layout trains whole epochs (every shard walks its own users): use cdae_hip_multi_train_epoch");
  const auto t0 = std::chrono::steady_clock::now();
  const size_t S = m->shard.size();
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) {
    CHK(item_epoch(m, seed, epoch, u_begin, u_end));
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
      for (size_t s = 0; s < S; ++s) {
        cdae_hip_stats st;
        CHK(cdae_hip_collect_stats(m->shard[s], &st));
        if (s == 0) {                                                             // every shard sees every user
          stats->users = st.users; stats->batches = st.batches;
          // HIP-event kernel times are shard 0's (cdae_hip_set_profiling on that shard's handle): its own launches over its rows
          stats->ms_sample = st.ms_sample; stats->ms_sort = st.ms_sort; stats->ms_encode = st.ms_encode; stats->ms_decode = st.ms_decode;
          stats->ms_hidden = st.ms_hidden; stats->ms_input = st.ms_input; stats->launches_decode = st.launches_decode;
        }
        // full output: a shard lists its own positives; sampled: every shard walks the whole list (other shards' examples VOID)
        if (m->cfg.full_output || s == 0) stats->examples += st.examples;
      }
      stats->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
  }
  if (S == 1) {
    CHK(cdae_hip_train_epoch(m->shard[0], seed, epoch, stats));
    return 0;
  }
  // relay part: the first (relay_epochs - epoch) epochs' worth of users on the single-GPU schedule
  std::vector<uint64_t> first(S, 0);
  const double left = m->relay_epochs - (double)epoch;
  const uint64_t R = left <= 0.0 ? 0 : (left >= 1.0 ? m->U : std::min<uint64_t>(m->U, (uint64_t)(left * (double)m->U + 1e-6)));   // (1.4 - 1 is 0.39999...: the fraction a caller wrote, not one user less)
  cdae_hip_stats relay_stats;
  std::memset(&relay_stats, 0, sizeof relay_stats);
  if (R) CHK(relay_part(m, seed, epoch, R, first, &relay_stats));
  const StepPlan pl = plan_of(m, first);
  if (R == m->U) {
    // the whole epoch was relayed: nothing to exchange
  } else if (m->single_device) {
    CHK(local_epoch(m, pl, seed, epoch));
  } else {
    // one host thread per device: the launches of a step cost tens of microseconds of host time per shard
    std::vector<int> rc(S, 0);
    std::vector<std::string> err(S);
    std::vector<std::thread> th;
    for (size_t s = 0; s < S; ++s)
      th.emplace_back([&, s] {
        xof(m->shard[s])->guard = &m->guard;
        rc[s] = shard_epoch(m, s, pl, seed, epoch);
        if (rc[s]) { err[s] = cdae_hip_last_error(); m->give_up(); }   // thread-local message: carry it to the caller's thread; free the peers
        xof(m->shard[s])->guard = nullptr;
      });
    for (std::thread& t : th) t.join();
    if (const int s = first_cause(rc, err); s >= 0) {
      if (m->comm_aborted) { m->abort_reason = err[s]; for (cdae_hip_t* h : m->shard) xof(h)->comm = nullptr; }
      return fail("shard %d: %s", s, err[s].c_str());
    }
  }
  if (stats) {
    *stats = relay_stats;
    for (cdae_hip_t* h : m->shard) {
      cdae_hip_stats st;
      CHK(cdae_hip_collect_stats(h, &st));
      stats->users += st.users; stats->examples += st.examples; stats->batches += st.batches;
    }
    stats->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return 0;
}

uint64_t cdae_hip_multi_steps_per_epoch(const cdae_hip_multi_t* m) {
  if (!m || m->shard.empty() || !m->U || m->layout == CDAE_LAYOUT_ITEM_ROWS) return 0;
  return plan_of(m, std::vector<uint64_t>(m->shard.size(), 0)).steps;
}

// Steps [step_begin, step_end) of epoch `epoch`'s EXCHANGED part (no relay), then a flush: what a benchmark times.  Training proper
// goes through cdae_hip_multi_train_epoch, which runs the relay part first and all the steps.
int cdae_hip_multi_train_steps(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, uint64_t step_begin, uint64_t step_end, cdae_hip_stats* stats) {
  CHK(check_multi(m, true));
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) return fail("cdae_hip_multi_train_steps: user-sharded layout only (item rows: cdae_hip_multi_train_users)");
  const size_t S = m->shard.size();
  StepPlan pl = plan_of(m, std::vector<uint64_t>(S, 0));
  if (step_begin > step_end || step_begin > pl.steps) return fail("bad step range [%llu, %llu) of %llu", (unsigned long long)step_begin, (unsigned long long)step_end, (unsigned long long)pl.steps);
  pl.t0 = step_begin; pl.t1 = step_end;
  const auto t0 = std::chrono::steady_clock::now();
  if (S == 1) {
    const uint64_t n = m->U, a = std::min(n, pl.t0 * pl.per[0]), b = std::min(n, std::min(pl.t1, pl.steps) * pl.per[0]);
    CHK(cdae_hip_train_users(m->shard[0], seed, epoch, a, b, stats));
    return 0;
  }
  if (m->single_device) {
    CHK(local_epoch(m, pl, seed, epoch));
  } else {
    std::vector<int> rc(S, 0);
    std::vector<std::string> err(S);
    std::vector<std::thread> th;
    for (size_t s = 0; s < S; ++s)
      th.emplace_back([&, s] {
        xof(m->shard[s])->guard = &m->guard;
        rc[s] = shard_epoch(m, s, pl, seed, epoch);
        if (rc[s]) { err[s] = cdae_hip_last_error(); m->give_up(); }
        xof(m->shard[s])->guard = nullptr;
      });
    for (std::thread& t : th) t.join();
    if (const int s = first_cause(rc, err); s >= 0) {
      if (m->comm_aborted) { m->abort_reason = err[s]; for (cdae_hip_t* h : m->shard) xof(h)->comm = nullptr; }
      return fail("shard %d: %s", s, err[s].c_str());
    }
  }
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    for (cdae_hip_t* h : m->shard) {
      cdae_hip_stats st;
      CHK(cdae_hip_collect_stats(h, &st));
      stats->users += st.users; stats->examples += st.examples; stats->batches += st.batches;
    }
    stats->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return 0;
}

int cdae_hip_multi_data_loss(cdae_hip_multi_t* m, uint64_t seed, uint32_t epoch, double* out) {
  CHK(check_multi(m, true));
  if (!out) return fail("null argument");
  double total = 0;
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) {                  // every shard adds the loss of ITS positives, z from the all-reduced sums
    const uint32_t chunk = cdae_internal::eval_chunk();
    for (uint64_t u0 = 0; u0 < m->U; u0 += chunk) {
      const uint32_t nu = (uint32_t)std::min<uint64_t>(chunk, m->U - u0);
      for (uint32_t c = 0; c < m->cfg.num_corruptions; ++c) {
        CHK(item_encode_chunk(m, u0, nu, 1, c, seed, epoch));
        for (cdae_hip_t* h : m->shard) CHK(cdae_internal::ev_data_loss(h, u0, nu, &total));
      }
    }
    *out = total / (double)m->cfg.num_corruptions;           // cdae.hpp:98
    return 0;
  }
  for (cdae_hip_t* h : m->shard) { double v = 0; CHK(cdae_hip_data_loss(h, seed, epoch, &v)); total += v; }   // sum over users, cdae.hpp:99
  *out = total;
  return 0;
}

int cdae_hip_multi_penalty_loss(cdae_hip_multi_t* m, double* out) {
  CHK(check_multi(m, true));
  if (!out) return fail("null argument");
  double total = 0, v = 0;
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) {
    for (cdae_hip_t* h : m->shard) { CHK(cdae_internal::item_rows_penalty(h, &v)); total += v; }     // every shard's own rows
    CHK(cdae_internal::hidden_bias_penalty(m->shard[0], &v)); total += v;                             // b is replicated: once
    for (cdae_hip_t* h : m->shard) { CHK(cdae_internal::private_penalty(h, &v)); total += v; }        // Wu: every shard's own users
    *out = total;
    return 0;
  }
  CHK(cdae_internal::shared_penalty(m->shard[0], &v)); total += v;         // replicas agree after every epoch's flush
  for (cdae_hip_t* h : m->shard) { CHK(cdae_internal::private_penalty(h, &v)); total += v; }
  *out = total;
  return 0;
}

int cdae_hip_multi_recommend_all(cdae_hip_multi_t* m, uint64_t u_begin, uint64_t u_end, uint32_t topk, uint32_t* out) {
  CHK(check_multi(m, true));
  if (u_begin > u_end || u_end > m->U || !out) return fail("bad user range");
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) {
    // every shard ranks ITS items for the chunk's users (scores kept), the host merges the candidates: descending score, ties to
    // the lower item id (heap.hpp:44-52 + utils.hpp:16-19 over ascending ids)
    const size_t S = m->shard.size();
    for (size_t s = 0; s < S; ++s)
      if (topk > m->icut[s + 1] - m->icut[s]) return fail("topk %u exceeds the %llu items of shard %zu", topk, (unsigned long long)(m->icut[s + 1] - m->icut[s]), s);
    const uint32_t chunk = cdae_internal::eval_chunk();
    std::vector<uint32_t> ids;
    std::vector<float> sc;
    std::vector<std::pair<float, uint32_t>> cand(S * topk);
    for (uint64_t u0 = u_begin; u0 < u_end; u0 += chunk) {
      const uint32_t nu = (uint32_t)std::min<uint64_t>(chunk, u_end - u0);
      CHK(item_encode_chunk(m, u0, nu, 0, 0, 0, 0));
      ids.resize(S * (size_t)nu * topk); sc.resize(ids.size());
      for (size_t s = 0; s < S; ++s)
        CHK(cdae_internal::ev_recommend(m->shard[s], u0, nu, topk, ids.data() + s * (size_t)nu * topk, sc.data() + s * (size_t)nu * topk));
      for (uint32_t u = 0; u < nu; ++u) {
        for (size_t s = 0; s < S; ++s)
          for (uint32_t t = 0; t < topk; ++t) {
            const size_t at = s * (size_t)nu * topk + (size_t)u * topk + t;
            cand[s * topk + t] = std::make_pair(sc[at], ids[at] == 0xFFFFFFFFu ? 0xFFFFFFFFu : ids[at] + (uint32_t)m->icut[s]);
          }
        std::sort(cand.begin(), cand.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
          return a.first > b.first || (a.first == b.first && a.second < b.second);
        });
        for (uint32_t t = 0; t < topk; ++t) out[((u0 - u_begin) + u) * topk + t] = cand[t].second;
      }
    }
    return 0;
  }
  for (size_t s = 0; s < m->shard.size(); ++s) {
    const uint64_t a = std::max(u_begin, m->cut[s]), b = std::min(u_end, m->cut[s + 1]);
    if (b > a) CHK(cdae_hip_recommend_all(m->shard[s], a - m->cut[s], b - m->cut[s], topk, out + (a - u_begin) * topk));
  }
  return 0;
}

// TOPN_Evaluation::evaluate over the shards (evaluation.hpp:113-181, evaluate_rec_list :183-219).  The merged top-k table of a
// sharded model is assembled on the host (the item shards' candidates meet there), so the eight columns are summed there too:
// one tight pass in user order — the same expressions, the same order of additions as cdae_hip_eval_topn's kernels and the
// reference's sequential loop — instead of num_thread workers each building a vector per user.
int cdae_hip_multi_eval_topn(cdae_hip_multi_t* m, const int64_t* test_row_ptr, const uint32_t* test_col, uint32_t topk,
                             double* rets8, uint64_t* hits3, uint32_t* ids_out) {
  CHK(check_multi(m, true));
  if (!test_row_ptr || !rets8 || topk == 0) return fail("bad argument");
  CHK(cdae_internal::validate_test_rows(test_row_ptr, test_col, m->U, m->I, nullptr));     // (the single handle's checks: cdae_hip_set_test_rows)
  std::vector<uint32_t> own;
  uint32_t* ids = ids_out;
  if (!ids) { own.resize((size_t)m->U * topk); ids = own.data(); }
  CHK(cdae_hip_multi_recommend_all(m, 0, m->U, topk, ids));
  double n_test_users = 0;
  for (uint64_t u = 0; u < m->U; ++u) n_test_users += test_row_ptr[u + 1] > test_row_ptr[u] ? 1. : 0.;
  if (n_test_users == 0) return fail("no user has test items");
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t h1 = 0, h5 = 0, h10 = 0;
  const uint32_t top = std::min<uint32_t>(20u, topk);
  for (uint64_t u = 0; u < m->U; ++u) {
    const int64_t t0 = test_row_ptr[u], t1 = test_row_ptr[u + 1];
    if (t1 <= t0) continue;
    const double nt = (double)(t1 - t0);
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hit = 0, map5 = 0, map10 = 0;
    for (uint32_t i = 0; i < top; ++i) {
      if (std::binary_search(test_col + t0, test_col + t1, ids[u * topk + i])) {
        hit += 1.;
        if (i < 5) map5 += hit / (double)(i + 1);
        if (i < 10) map10 += hit / (double)(i + 1);
        h1 += i < 1; h5 += i < 5; h10 += i < 10;
      }
      if (i == 0) { r[0] = hit; r[3] = hit / nt; }
      else if (i == 4) { r[1] = hit / 5.; r[4] = hit / nt; }
      else if (i == 9) { r[2] = hit / 10.; r[5] = hit / nt; }
    }
    r[6] = map5 / std::min(5., nt);
    r[7] = map10 / std::min(10., nt);
    for (int c = 0; c < 8; ++c) acc[c] += r[c] / n_test_users;
  }
  for (int c = 0; c < 8; ++c) rets8[c] = acc[c];
  if (hits3) { hits3[0] = h1; hits3[1] = h5; hits3[2] = h10; }
  return 0;
}

static bool is_private(uint32_t which) { return which == CDAE_P_WU || which == CDAE_P_WU_AG || which == CDAE_P_UU || which == CDAE_P_UU_AG; }

static bool is_item_rows(uint32_t which) {
  return which == CDAE_P_W || which == CDAE_P_W_AG || which == CDAE_P_V || which == CDAE_P_V_AG || which == CDAE_P_BP || which == CDAE_P_BP_AG;
}

int cdae_hip_multi_get_param(cdae_hip_multi_t* m, uint32_t which, float* host, size_t count) {
  CHK(check_multi(m, true));
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS && is_private(which)) {                                 // user node: sharded by user range
    const size_t K = m->cfg.num_dim;
    if (count != m->U * K) return fail("parameter %u has %zu elements, got %zu", which, (size_t)(m->U * K), count);
    for (size_t s = 0; s < m->shard.size(); ++s)
      CHK(cdae_hip_get_param(m->shard[s], which, host + m->cut[s] * K, (m->cut[s + 1] - m->cut[s]) * K));
    return 0;
  }
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) {
    if (!is_item_rows(which)) return cdae_hip_get_param(m->shard[0], which, host, count);          // replicated (b)
    const size_t w = (which == CDAE_P_BP || which == CDAE_P_BP_AG) ? 1 : m->cfg.num_dim;
    if (count != m->I * w) return fail("parameter %u has %zu elements, got %zu", which, (size_t)(m->I * w), count);
    for (size_t s = 0; s < m->shard.size(); ++s)
      CHK(cdae_hip_get_param(m->shard[s], which, host + m->icut[s] * w, (m->icut[s + 1] - m->icut[s]) * w));
    return 0;
  }
  if (!is_private(which)) return cdae_hip_get_param(m->shard[0], which, host, count);
  const size_t K = m->cfg.num_dim;
  if (count != m->U * K) return fail("parameter %u has %zu elements, got %zu", which, (size_t)(m->U * K), count);
  for (size_t s = 0; s < m->shard.size(); ++s)
    CHK(cdae_hip_get_param(m->shard[s], which, host + m->cut[s] * K, (m->cut[s + 1] - m->cut[s]) * K));
  return 0;
}

int cdae_hip_multi_set_param(cdae_hip_multi_t* m, uint32_t which, const float* host, size_t count) {
  CHK(check_multi(m, true));
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS && is_private(which)) {
    const size_t K = m->cfg.num_dim;
    if (count != m->U * K) return fail("parameter %u has %zu elements, got %zu", which, (size_t)(m->U * K), count);
    for (size_t s = 0; s < m->shard.size(); ++s)
      CHK(cdae_hip_set_param(m->shard[s], which, host + m->cut[s] * K, (m->cut[s + 1] - m->cut[s]) * K));
    return 0;
  }
  if (m->layout == CDAE_LAYOUT_ITEM_ROWS) {
    if (!is_item_rows(which)) { for (cdae_hip_t* h : m->shard) CHK(cdae_hip_set_param(h, which, host, count)); return 0; }
    const size_t w = (which == CDAE_P_BP || which == CDAE_P_BP_AG) ? 1 : m->cfg.num_dim;
    if (count != m->I * w) return fail("parameter %u has %zu elements, got %zu", which, (size_t)(m->I * w), count);
    for (size_t s = 0; s < m->shard.size(); ++s)
      CHK(cdae_hip_set_param(m->shard[s], which, host + m->icut[s] * w, (m->icut[s + 1] - m->icut[s]) * w));
    return 0;
  }
  if (!is_private(which)) {
    for (cdae_hip_t* h : m->shard) CHK(cdae_hip_set_param(h, which, host, count));
  } else {
    const size_t K = m->cfg.num_dim;
    if (count != m->U * K) return fail("parameter %u has %zu elements, got %zu", which, (size_t)(m->U * K), count);
    for (size_t s = 0; s < m->shard.size(); ++s)
      CHK(cdae_hip_set_param(m->shard[s], which, host + m->cut[s] * K, (m->cut[s + 1] - m->cut[s]) * K));
  }
  for (cdae_hip_t* h : m->shard) { Exchange* x = xof(h); if (x) { x->begun = false; x->pending = false; x->staged_sync = false; x->steps = 0; } }
  return 0;
}

}  // extern "C"
