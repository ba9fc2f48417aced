// cdae_exchange_algebra.h — the arithmetic of the multi-GPU layouts that does not need a GPU to be stated or checked:
//   * the per-element STAGE / MERGE / MERGE_STAGE algebra of the pipelined shared-parameter exchange (user-sharded layout),
//   * the rule by which a shard contributes its users' private rows to an all-reduce(sum) (item-rows layout: owner row + zeros),
//   * the balanced contiguous cuts both layouts shard by.
// ONE source, two compilers: hipcc includes it from the device kernels (cdae_kernels.hpp: delta_pipe_kernel, own_rows_stage_kernel)
// and from cdae_multi.hip; g++ compiles it into the CPU slice that tests/test_distributed_cpu.py drives with two gloo ranks —
// the world_size-2 test exercises the shipped functions, not a Python stand-in.  No reference counterpart: the reference is
// single-process (cdae.hpp:136-146).
#ifndef CDAE_EXCHANGE_ALGEBRA_H_
#define CDAE_EXCHANGE_ALGEBRA_H_

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CDAE_XA_FN __host__ __device__ __forceinline__
#else
#define CDAE_XA_FN inline
#endif

namespace cdae_xa {

// Pipelined exchange.  Per element: c = this replica's parameter, A = the state all replicas AGREE on bit for bit, snap = c as it
// was when the last delta was staged, s / r = the staged delta (send copy / the copy that is all-reduced in place).
//   STAGE:        s = r = c - A ; snap = c
//   MERGE:        A += r ; c = A + (c - snap)          (everybody's staged deltas, then this replica's progress since the stage)
//   MERGE_STAGE:  MERGE of the previous period, then STAGE of this one
// A moves only by the all-reduced sums — the same bits on every rank — so the replicas' A never drift; whenever nothing was
// trained between a STAGE and its MERGE, c - snap is exactly 0 and c == A on every replica, bit for bit.
enum { STAGE = 0, MERGE = 1, MERGE_STAGE = 2 };

template <int MODE>
CDAE_XA_FN void pipe_elem(float& c, float& A, float& snap, float& s, float& r) {
  if (MODE != STAGE) {
    A += r;
    c = A + (c - snap);
  }
  if (MODE != MERGE) {
    const float d = c - A;
    s = d; r = d; snap = c;
  }
}

// COMBINE RULE "global accumulator" (CDAE_COMBINE_GLOBAL_ACC, AdaGrad only).  The plain rule above sums every replica's accumulated
// steps, each preconditioned by an accumulator that has seen only that replica's examples (eight summed first steps of a
// young accumulator are not one sequence of eight steps).  This rule exchanges, per (parameter p, accumulator a) pair, the replica's
// accumulator growth da = a - A_a (exactly its sum of squared gradients, cdae.hpp:254) and its step with the replica's own
// preconditioner taken back out, dp * (beta + sqrt(a)) ~ -lr * (the replica's summed gradient); after the all-reduce(sum) everybody
// takes ONE step with the accumulator that has seen all replicas:  A_a += sum da ;  A_p += sum(...) / (beta + sqrt(A_a)).
// A row that a single replica touched moves exactly as that replica moved it (a == A_a + da).  With lr folded into dp nothing else
// is needed from the optimiser.  Same agreement property as pipe_elem: A moves only by all-reduced bits.
template <int MODE>
CDAE_XA_FN void pipe_pair(float& cp, float& ca, float& Ap, float& Aa, float& snp, float& sna, float& sp, float& sa, float& rp, float& ra,
                          float beta) {
  if (MODE != STAGE) {
    Aa += ra;
    const float den = beta + sqrtf(Aa);
    Ap += den > 0.f ? rp / den : 0.f;
    cp = Ap + (cp - snp);
    ca = Aa + (ca - sna);
  }
  if (MODE != MERGE) {
    const float da = ca - Aa, dp = (cp - Ap) * (beta + sqrtf(ca));
    sp = dp; rp = dp; sa = da; ra = da;
    snp = cp; sna = ca;
  }
}

// Item-rows layout, user node sharded by user: what a shard puts into the all-reduce(sum) for user `uid`'s private row element —
// its own value when it owns the user, +0 otherwise.  x + 0 + ... + 0 == x exactly, so every shard receives the owner's value.
CDAE_XA_FN bool owns_user(uint64_t uid, uint64_t own_u0, uint64_t own_u1) { return uid >= own_u0 && uid < own_u1; }
CDAE_XA_FN float own_row_contribution(bool own, float value) { return own ? value : 0.f; }

// Contiguous cuts of n rows into S ranges balanced by a prefix sum (prefix[n] = total weight): range s starts at the first row whose
// prefix reaches s/S of the total.  at_least_one: every range keeps one row or more (user shards, item shards); otherwise a range
// may be empty (the item-rows layout's user ownership).  cuts has S + 1 entries.
inline void balanced_cuts(const int64_t* prefix, uint64_t n, uint64_t S, bool at_least_one, uint64_t* cuts) {
  const int64_t total = prefix[n];
  cuts[0] = 0;
  for (uint64_t s = 1; s < S; ++s) {
    const int64_t want = (int64_t)(((__int128)total * (int64_t)s + (int64_t)S - 1) / (int64_t)S);
    uint64_t lo = 0, hi = n + 1;                       // first index with prefix[index] >= want
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (prefix[mid] < want) lo = mid + 1; else hi = mid; }
    uint64_t u = lo;
    if (at_least_one) {
      if (u < cuts[s - 1] + 1) u = cuts[s - 1] + 1;
      if (u > n - (S - s)) u = n - (S - s);
    } else {
      if (u < cuts[s - 1]) u = cuts[s - 1];
      if (u > n) u = n;
    }
    cuts[s] = u;
  }
  cuts[S] = n;
}

}  // namespace cdae_xa
#endif  // CDAE_EXCHANGE_ALGEBRA_H_
