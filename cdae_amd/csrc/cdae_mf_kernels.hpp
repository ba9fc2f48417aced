// cdae_mf_kernels.hpp — the reference's sibling SGD models on the same row-step machinery:
//   IMF  implicit-feedback matrix factorisation   /root/reference/src/model/recsys/imf.hpp:71-119
//   BPR  pairwise ranking                          /root/reference/src/model/recsys/bpr.hpp:56-106
// Both walk the users in order; per positive item they take one pointwise instance + num_neg sampled negatives (IMF) or
// num_neg (positive, sampled negative) pairs (BPR); every instance steps the USER vector and the ITEM row(s) at once.  Unlike
// CDAE the user side moves with every instance, so a user's chain cannot be transposed away; the block schedule is
//   phase U  mf_user_kernel: one wavefront per user of the block — uv[u], uv_ag[u], ub[u] live in registers while the user's
//            instances run strictly in order; the item side is READ (the item rows are not written in this phase, so no
//            snapshot copy is needed); every instance leaves its loss gradient g and the user vector before its step;
//   phase I  mf_item_kernel: one wavefront per item row — the row and its accumulators live in registers while the row's
//            contributions run in (user, instance) order: grad = +-g * uv_before + 2 lambda * row   (the item-major order comes
//            from the same sort + segment table as CDAE's decode).
// A block of ONE user runs phase U with IN_PLACE = true: the item rows are stepped at once — the reference loop itself,
// duplicate negatives of the user included.  tests/test_gpu_mf.py checks both against oracle/mf_oracle.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cdae_kernels.hpp"

#include <type_traits>

namespace cdae {

// loss.hpp gradients of every loss the two models accept (yelp.cpp:122-165): SQUARE 0, LOGISTIC 1, LOG 2, HINGE 3, CROSS_ENTROPY 5
__device__ __forceinline__ float mf_loss_grad(uint32_t loss_type, float pred, float truth) {
  switch (loss_type) {
    case 0u: return -2.f * (truth - pred);                                             // loss.hpp:54
    case 1u: return (pred - truth) * fast_rcp(pred * (1.f - pred));                    // loss.hpp:95-98 (the reference CHECKs 0 < pred < 1)
    case 2u: return -truth * fast_rcp(1.f + fast_exp(pred * truth));                   // loss.hpp:189-197 (the +-18 branches are its fp32 limits)
    case 3u: return pred * truth > 1.f ? 0.f : -truth;                                 // loss.hpp:283-288
    default: return fast_rcp(1.f + fast_exp(-pred)) - truth;                           // loss.hpp:141-147
  }
}
__device__ __forceinline__ float mf_negative_label(uint32_t loss_type) { return (loss_type == 2u || loss_type == 3u) ? -1.f : 0.f; }

// IN_PLACE reads of rows this wavefront may have just written (a user's duplicate negative): bypass the L1, which a store does
// not update
template <int NI>
__device__ __forceinline__ void vload_coherent(float (&d)[NI], const float* p) {
#pragma unroll
  for (int i = 0; i < NI; ++i) d[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// K1 for IMF / BPR: the block's instance list, user-major in the order the reference takes them (imf.hpp:77-84, bpr.hpp:62-68):
//   IMF  instance t of a user = positive p = t / (1 + num_neg) itself when t % (1 + num_neg) == 0, else its (t % .. - 1)-th negative;
//        one example per instance (index = instance)
//   BPR  instance (pair) q = p * num_neg + k; two examples per pair: 2 q = (positive item, +), 2 q + 1 = (negative item, -)
// Negatives: recsys_model_base.hpp:46-57 on the counter stream (draw index p * num_neg + k), as in sample_kernel.
// val = instance index << 32 | slot | (negative side of a pair ? TARGET_BIT : 0).  One wavefront per work unit of positives.
__global__ void __launch_bounds__(256)
mf_sample_kernel(HyperParams hp, uint32_t pairwise, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                 const uint32_t* __restrict__ uptr, uint32_t n_units, uint64_t u0, uint32_t nb, uint64_t seed, uint32_t epoch,
                 uint32_t* __restrict__ ex_item, uint64_t* __restrict__ ex_val, uint16_t* __restrict__ ex_key16,
                 uint32_t* __restrict__ seg, uint32_t seg_words, uint32_t* __restrict__ dup_count, uint32_t* __restrict__ dup_of_ex,
                 const uint32_t* __restrict__ unit_user, uint32_t* __restrict__ wg_state /* bucket_sort_kernel's per-range words, or nullptr */) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < seg_words; i += gridDim.x * blockDim.x) seg[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < DUP_STRIPES) dup_count[threadIdx.x] = 0u;
  if (blockIdx.x == 0 && wg_state)
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) wg_state[i] = 0u;
  const uint32_t unit = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (unit >= n_units) return;
  const UnitRef ur = locate_unit(hp.unit_pos, uptr, nb, uptr[0] + unit, unit_user, u0);
  const uint32_t slot = ur.slot;
  const uint64_t uid = u0 + slot;
  const int64_t r0 = row_ptr[uid];
  const uint32_t n = (uint32_t)(row_ptr[uid + 1] - r0);
  const uint32_t p0 = ur.p0, p1 = min(ur.p1, n);
  const uint32_t* row = col + r0;
  const uint32_t per = pairwise ? hp.num_neg : 1u + hp.num_neg;                 // instances per positive
  const uint64_t inst0 = (uint64_t)(r0 - row_ptr[u0]) * per;                   // first instance of the user in the block
  const uint64_t key_n = cdae_rng_key(seed, epoch, uid + hp.uid_offset, CDAE_STREAM_NEGATIVE);
  const uint32_t work = (p1 - p0) * (1u + hp.num_neg);                         // (positive, k) with k = 0: the positive itself
  for (uint32_t w = lane; w < work; w += WAVE) {
    const uint32_t p = p0 + w / (1u + hp.num_neg), k = w % (1u + hp.num_neg);
    const uint32_t item_pos = row[p];
    uint32_t item = item_pos;
    if (k) item = cdae_sample_negative(key_n, (uint64_t)p * hp.num_neg + (k - 1u), row, n, hp.num_items);
    if (!pairwise) {
      const uint64_t e = inst0 + (uint64_t)p * per + k;
      ex_item[e] = item;
      if (ex_key16) ex_key16[e] = (uint16_t)item;
      ex_val[e] = (e << 32) | (uint64_t)slot;
      dup_of_ex[e] = DUP_NONE;
    } else if (k) {
      const uint64_t q = inst0 + (uint64_t)p * per + (k - 1u), e = 2u * q;
      ex_item[e] = item_pos; ex_item[e + 1] = item;
      if (ex_key16) { ex_key16[e] = (uint16_t)item_pos; ex_key16[e + 1] = (uint16_t)item; }
      ex_val[e] = (q << 32) | (uint64_t)slot;
      ex_val[e + 1] = (q << 32) | (uint64_t)(slot | TARGET_BIT);
      dup_of_ex[q] = DUP_NONE;
    }
  }
}

// phase U (and, IN_PLACE, the whole reference loop for a block of one user)
template <int NI, bool PAIR, bool IN_PLACE>
__global__ void __launch_bounds__(256)
mf_user_kernel(HyperParams hp, uint32_t bias_term, const int64_t* __restrict__ row_ptr, uint64_t u0, uint32_t nb,
               const uint32_t* __restrict__ ex_item, float* __restrict__ UV, float* __restrict__ UV_ag, float* __restrict__ UB,
               float* __restrict__ UB_ag, float* __restrict__ IV, float* __restrict__ IV_ag, float* __restrict__ IB,
               float* __restrict__ IB_ag, float* __restrict__ UVpre /* [instances][Kp] */, float* __restrict__ G /* [instances] */) {
  // wave-uniform by construction; said so, everything derived from it (row bounds, counts, loop control) stays scalar
  const uint32_t wave_slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE));
  const uint32_t lane = threadIdx.x % WAVE;
  // IN_PLACE: ONE wavefront walks the launch's users in order (the reference's `for uid` loop, imf.hpp:71-86 / bpr.hpp:56-70: a block
  // of one user, or the sequential default's launch window); otherwise a wavefront per user of the block
  if (wave_slot >= (IN_PLACE ? 1u : nb)) return;
  const uint32_t slot_first = IN_PLACE ? 0u : wave_slot, slot_end = IN_PLACE ? nb : wave_slot + 1u;
  const uint32_t per = PAIR ? hp.num_neg : 1u + hp.num_neg;
  const uint32_t lo = lane * NI;
  const float lam2 = hp.lambda;                      // the host stores 2 * lambda here (imf.hpp:92-95 regularise with 2 lambda)
  const float neg_label = mf_negative_label(hp.loss_type);
  // the whole loop several times over: AdaGrad or plain SGD fixed at compile time (see ada_step_t), and so is the loss for the two
  // the models default to (LT < 0: the run-time switch).  A lone wavefront per user issues one instruction every ~5 cycles whatever
  // its kind, so the instance chain is its instruction count: the switch and its branches were a seventh of it.
  auto run = [&](auto ada_tag, auto lt_tag) __attribute__((always_inline)) {
  constexpr bool ADA = decltype(ada_tag)::value;
  constexpr int LT = decltype(lt_tag)::value;
  for (uint32_t slot = slot_first; slot < slot_end; ++slot) {
  const uint64_t uid = u0 + slot;
  const int64_t r0 = row_ptr[uid];
  const uint32_t n = (uint32_t)(row_ptr[uid + 1] - r0);
  const uint64_t inst0 = (uint64_t)(r0 - row_ptr[u0]) * per;
  const uint32_t n_inst = n * per;
  float uv[NI], ua[NI];
  vload<NI>(uv, UV + (size_t)uid * hp.Kp + lo);
  vload<NI>(ua, UV_ag + (size_t)uid * hp.Kp + lo);
  float ub = UB[uid], uba = UB_ag[uid];
  // item rows of the NEXT group of PF instances travel while the current group is stepped (IN_PLACE: rows move under the loop,
  // nothing is fetched ahead).  Two register sets and a copy at the group boundary: the ring-of-slots form (fetch x + PF into
  // slot x % PF) made the compiler wait for every load in flight at each instance, the one just issued included.
  // IN_PLACE (round 4): the rows AND accumulators of the next group of 4 instances travel too; an instance whose item was
  // stepped inside the window its prefetch could not see — the 2 PF instances before it — is re-read after the stores have landed
  constexpr int PF = IN_PLACE ? 4 : (PAIR || NI > 4) ? 8 : 16;
  // this lane's item id(s) of the current chunk of 64 instances and, when the item side does not move under this kernel, their
  // biases (one gather per chunk instead of one dependent scalar load per instance)
  uint32_t ci = 0, cj = 0, ni = 0, nj = 0;
  float cib = 0.f, cjb = 0.f, ciba = 0.f;
  // BPR block schedule (round 4): the user's num_neg pairs of one positive share that item, and the loop steps its row between them
  // (bpr.hpp:84-105).  The wavefront carries a PRIVATE copy of the positive's row, accumulators and bias from pair to pair — the
  // block-start values at the positive's first pair, then stepped with each pair's own g and the user vector from before its step —
  // and forms the next pair's prediction and user step from it.  Phase I is unchanged (it steps the real row with the same g's).
  // Without the copy every block size sat 0.008 low in Recall@10 for the first two epochs (tools/mf_envelope.py).
  constexpr bool CARRY = PAIR && !IN_PLACE;
  float pw[NI], pa[NI], pb = 0.f, pba = 0.f;
  uint32_t p_item = 0xFFFFFFFFu;                     // wave-uniform: the positive whose private copy the registers hold
#pragma unroll
  for (int i = 0; i < NI; ++i) { pw[i] = 0.f; pa[i] = 1.f; }
  auto load_ids = [&](uint32_t c0, uint32_t& a, uint32_t& b) {
    const uint32_t t = c0 + lane;
    a = 0; b = 0;
    if (t < n_inst) {
      if (PAIR) { a = ex_item[2u * (inst0 + t)]; b = ex_item[2u * (inst0 + t) + 1u]; }
      else a = ex_item[inst0 + t];
    }
  };
  load_ids(0, ni, nj);
  for (uint32_t c0 = 0; c0 < n_inst; c0 += WAVE) {
    ci = ni; cj = nj;
    if (c0 + WAVE < n_inst) load_ids(c0 + WAVE, ni, nj);     // the next chunk's ids travel while this one is stepped
    if (!IN_PLACE) { cib = IB[ci]; if (PAIR) cjb = IB[cj]; if (CARRY) ciba = IB_ag[ci]; }
    const uint32_t cnt = min((uint32_t)WAVE, n_inst - c0);
    // per lane: the label of chunk instance `lane` (imf.hpp:80-84: a positive, then its negatives), and the slot its loss
    // gradient is parked in until the chunk's 64 go out in one store
    const float tl = PAIR ? 1.f : ((c0 + lane) % per == 0u) ? 1.f : neg_label;
    float gl = 0.f;
    float ri[PF][NI], rj[PF][NI], qi[PF][NI], qj[PF][NI];
    // IN_PLACE only: the rows' AdaGrad accumulators and the item biases (value, accumulator) of the same instances
    constexpr int PA = (IN_PLACE || CARRY) ? PF : 1, PJ = IN_PLACE && PAIR ? PF : 1;
    float ai[PA][NI], aj[PJ][NI], pai[PA][NI], paj[PJ][NI];
    float bi[PA][2], bj[PJ][2], pbi[PA][2], pbj[PJ][2];
    auto fetch = [&](float (&di)[NI], float (&dj)[NI], float (&dai)[NI], uint32_t idx) {      // rows of chunk instance idx
      const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)ci, idx);
      if (IN_PLACE) vload_coherent<NI>(di, IV + (size_t)it * hp.Kp + lo); else vload<NI>(di, IV + (size_t)it * hp.Kp + lo);
      if (CARRY) vload<NI>(dai, IV_ag + (size_t)it * hp.Kp + lo);          // (the positive's accumulators: its private copy starts from them)
      if (PAIR) {
        const uint32_t jt = (uint32_t)__builtin_amdgcn_readlane((int)cj, idx);
        if (IN_PLACE) vload_coherent<NI>(dj, IV + (size_t)jt * hp.Kp + lo); else vload<NI>(dj, IV + (size_t)jt * hp.Kp + lo);
      }
    };
    // IN_PLACE: everything an instance's step reads from the item side — rows, accumulators, biases — in one go (L1-bypassing loads:
    // this wavefront wrote some of these lines earlier and a store does not update the L1)
    auto fetch_all = [&](int s, float (&di)[NI], float (&dj)[NI], float (&dai)[NI], float (&daj)[NI], float (&dbi)[2], float (&dbj)[2], uint32_t idx) {
      (void)s;
      const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)ci, idx);
      vload_coherent<NI>(di, IV + (size_t)it * hp.Kp + lo);
      vload_coherent<NI>(dai, IV_ag + (size_t)it * hp.Kp + lo);
      dbi[0] = __hip_atomic_load(IB + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dbi[1] = __hip_atomic_load(IB_ag + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (PAIR) {
        const uint32_t jt = (uint32_t)__builtin_amdgcn_readlane((int)cj, idx);
        vload_coherent<NI>(dj, IV + (size_t)jt * hp.Kp + lo);
        vload_coherent<NI>(daj, IV_ag + (size_t)jt * hp.Kp + lo);
        dbj[0] = __hip_atomic_load(IB + jt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dbj[1] = __hip_atomic_load(IB_ag + jt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    auto fetch_group = [&](uint32_t g0) {              // past the end of the chunk: the last row again (in bounds, never used)
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        if (IN_PLACE) fetch_all(s, qi[s], qj[s], pai[s % PA], paj[s % PJ], pbi[s % PA], pbj[s % PJ], min(g0 + (uint32_t)s, cnt - 1u));
        else fetch(qi[s], qj[s], pai[s % PA], min(g0 + (uint32_t)s, cnt - 1u));
      }
    };
    auto take_group = [&]() {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
#pragma unroll
        for (int i = 0; i < NI; ++i) { ri[s][i] = qi[s][i]; if (PAIR) rj[s][i] = qj[s][i]; }
        if (IN_PLACE) {
#pragma unroll
          for (int i = 0; i < NI; ++i) { ai[s % PA][i] = pai[s % PA][i]; if (PAIR) aj[s % PJ][i] = paj[s % PJ][i]; }
          bi[s % PA][0] = pbi[s % PA][0]; bi[s % PA][1] = pbi[s % PA][1];
          if (PAIR) { bj[s % PJ][0] = pbj[s % PJ][0]; bj[s % PJ][1] = pbj[s % PJ][1]; }
        } else if (CARRY) {
#pragma unroll
          for (int i = 0; i < NI; ++i) ai[s % PA][i] = pai[s % PA][i];
        }
      }
    };
    auto step = [&](int s, uint32_t x) __attribute__((always_inline)) {   // instance x of the chunk, its rows in register set s
      const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)ci, x);
      const uint32_t jt = PAIR ? (uint32_t)__builtin_amdgcn_readlane((int)cj, x) : 0u;
      if (IN_PLACE) {
        // this instance's rows were requested up to 2 PF - 1 instances ago (fetch_group of the group before its own ran in front of
        // that group's steps): if one of the instances x - 2 PF .. x - 1 of this chunk stepped one of its rows, what sits in the
        // registers is from before that step — wait for the stores and read again.  (Negatives are rejected against the user's
        // positives, so this is a duplicate negative inside eight instances, or BPR's positive item, which its num_neg pairs share.)
        const uint32_t w0 = x > 2u * (uint32_t)PF ? x - 2u * (uint32_t)PF : 0u;
        const bool in_window = lane >= w0 && lane < x;
        const bool hit = in_window && (ci == it || (PAIR && (cj == it || ci == jt || cj == jt)));
        if (__builtin_amdgcn_ballot_w64(hit) != 0ull || (PAIR && it == jt)) {
          __builtin_amdgcn_s_waitcnt(WAIT_VM0);
          fetch_all(s, ri[s], rj[s], ai[s % PA], aj[s % PJ], bi[s % PA], bj[s % PJ], x);
        }
      }
      const bool carried = CARRY && it == p_item;    // wave-uniform: a later pair of the positive whose private copy is held
      if (CARRY && !carried) {                       // the positive's first pair: the block-start row, accumulators and bias
#pragma unroll
        for (int i = 0; i < NI; ++i) { pw[i] = ri[s][i]; pa[i] = ai[s % PA][i]; }
        pb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cib), x));
        pba = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ciba), x));
        p_item = it;
      }
      float d[NI];                                   // the item-side vector of the user step: iv[i] (IMF) or iv[i] - iv[j] (BPR)
#pragma unroll
      for (int i = 0; i < NI; ++i) d[i] = CARRY ? pw[i] - rj[s][i] : PAIR ? ri[s][i] - rj[s][i] : ri[s][i];
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) dot = fmaf(uv[i], d[i], dot);
      float pred = wave_sum(dot);
      float truth;
      const float ibi = IN_PLACE ? bi[s % PA][0]
                        : CARRY  ? pb
                                 : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cib), x));
      if (PAIR) {                                                                  // bpr.hpp:73-76 (ub cancels in the difference)
        pred += ibi - (IN_PLACE ? bj[s % PJ][0]
                                : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cjb), x)));
        truth = 1.f;
      } else {                                                                     // imf.hpp:117-119, 80-84
        pred += ub + ibi;
        truth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tl), x));
      }
      const float g = mf_loss_grad(LT < 0 ? hp.loss_type : (uint32_t)LT, pred, truth);
      const uint64_t inst = inst0 + c0 + x;
      if (!IN_PLACE) {
        vstore<NI>(UVpre + (size_t)inst * hp.Kp + lo, uv);
        gl = lane == x ? g : gl;
        if (CARRY) {                                 // the private copy takes the loop's step of the positive (bpr.hpp:79, 84-87, 93-96)
          if (bias_term) ada_step_t<ADA>(hp, pb, pba, fmaf(lam2, pb, g));
#pragma unroll
          for (int i = 0; i < NI; ++i) ada_step_t<ADA>(hp, pw[i], pa[i], fmaf(g, uv[i], lam2 * pw[i]));
        }
      } else {
        // the reference loop: item row(s) stepped at once with the user vector from BEFORE its own step (imf.hpp:94-114)
        float w[NI], a[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) { w[i] = ri[s][i]; a[i] = ai[s % PA][i]; ada_step_t<ADA>(hp, w[i], a[i], fmaf(g, uv[i], lam2 * w[i])); }
        vstore<NI>(IV + (size_t)it * hp.Kp + lo, w);
        vstore<NI>(IV_ag + (size_t)it * hp.Kp + lo, a);
        if (bias_term && lane == 0) {
          float b = ibi, ba = bi[s % PA][1];
          ada_step_t<ADA>(hp, b, ba, fmaf(lam2, b, g)); IB[it] = b; IB_ag[it] = ba;
        }
        if (PAIR) {
          // (it == jt cannot happen: a negative is never one of the user's positives; the re-read above covers it all the same)
#pragma unroll
          for (int i = 0; i < NI; ++i) { w[i] = rj[s][i]; a[i] = aj[s % PJ][i]; ada_step_t<ADA>(hp, w[i], a[i], fmaf(-g, uv[i], lam2 * w[i])); }
          vstore<NI>(IV + (size_t)jt * hp.Kp + lo, w);
          vstore<NI>(IV_ag + (size_t)jt * hp.Kp + lo, a);
          if (bias_term && lane == 0) {
            float b = bj[s % PJ][0], ba = bj[s % PJ][1];
            ada_step_t<ADA>(hp, b, ba, fmaf(lam2, b, -g)); IB[jt] = b; IB_ag[jt] = ba;
          }
        }
        // (no wait here: the stores are only waited for where a later instance's re-read needs them, and at the chunk's end)
      }
      if (!PAIR && bias_term) ada_step_t<ADA>(hp, ub, uba, fmaf(lam2, ub, g));            // imf.hpp:97-101, 108-111 (BPR never steps ub)
#pragma unroll
      for (int i = 0; i < NI; ++i) ada_step_t<ADA>(hp, uv[i], ua[i], fmaf(g, d[i], lam2 * uv[i]));
    };
    if (IN_PLACE) {
      // the grouped walk below, with the item side stepped in place; the conflict window is counted inside ONE chunk, so every chunk
      // starts with nothing of the previous one in flight
      __builtin_amdgcn_s_waitcnt(WAIT_VM0);
      fetch_group(0);
      take_group();
      uint32_t x0 = 0;
      for (; x0 + PF < cnt; x0 += PF) {
        fetch_group(x0 + PF);
#pragma unroll
        for (int s = 0; s < PF; ++s) step(s, x0 + (uint32_t)s);
        take_group();
      }
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        if (x0 + (uint32_t)s >= cnt) break;
        step(s, x0 + (uint32_t)s);
      }
      __builtin_amdgcn_s_waitcnt(WAIT_VM0);
    } else {
      // a full group with a successor: fetch the successor, step the group, take the successor -- one straight path, so the wait
      // before the take is a count of the stores issued since (not "everything")
      fetch_group(0);
      take_group();
      uint32_t x0 = 0;
      for (; x0 + PF < cnt; x0 += PF) {
        fetch_group(x0 + PF);
#pragma unroll
        for (int s = 0; s < PF; ++s) step(s, x0 + (uint32_t)s);
        take_group();
      }
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        if (x0 + (uint32_t)s >= cnt) break;
        step(s, x0 + (uint32_t)s);
      }
      if (lane < cnt) G[inst0 + c0 + lane] = gl;
    }
  }
  vstore<NI>(UV + (size_t)uid * hp.Kp + lo, uv);
  vstore<NI>(UV_ag + (size_t)uid * hp.Kp + lo, ua);
  if (lane == 0) { UB[uid] = ub; UB_ag[uid] = uba; }
  }
  };
  using any_loss = std::integral_constant<int, -1>;
  if (IN_PLACE) {                                      // the literal loop (blocks of one user): not a speed path
    if (hp.adagrad) run(std::true_type{}, any_loss{}); else run(std::false_type{}, any_loss{});
  } else if (hp.adagrad) {
    if (hp.loss_type == 0u) run(std::true_type{}, std::integral_constant<int, 0>{});
    else if (hp.loss_type == 2u) run(std::true_type{}, std::integral_constant<int, 2>{});
    else run(std::true_type{}, any_loss{});
  } else {
    run(std::false_type{}, any_loss{});
  }
}

// phase I: one wavefront per item row, contributions in (user, instance) order
template <int NI>
__global__ void __launch_bounds__(256)
mf_item_kernel(HyperParams hp, uint32_t bias_term, const uint32_t* __restrict__ item_order, const uint32_t* __restrict__ seg_begin,
               const uint32_t* __restrict__ seg_end, const uint64_t* __restrict__ sorted_val, const float* __restrict__ UVpre,
               const float* __restrict__ G, float* __restrict__ IV, float* __restrict__ IV_ag, float* __restrict__ IB,
               float* __restrict__ IB_ag) {
  const uint32_t rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE));
  const uint32_t lane = threadIdx.x % WAVE;
  if (rank >= hp.num_items) return;
  const uint32_t item = item_order[rank];
  const uint32_t beg = seg_begin[item], end = seg_end[item];
  if (beg == end) return;
  const uint32_t lo = lane * NI;
  const float lam2 = hp.lambda;
  auto run = [&](auto ada_tag) __attribute__((always_inline)) {
  constexpr bool ADA = decltype(ada_tag)::value;
  float w[NI], a[NI];
  vload<NI>(w, IV + (size_t)item * hp.Kp + lo);
  vload<NI>(a, IV_ag + (size_t)item * hp.Kp + lo);
  float b = IB[item], ba = IB_ag[item];
  constexpr int PF = NI > 4 ? 8 : 16;                  // contributions per group; the next group's rows travel under this one's steps
  for (uint32_t c0 = beg; c0 < end; c0 += WAVE) {
    const uint32_t cnt = min((uint32_t)WAVE, end - c0);
    const uint64_t v = c0 + lane < end ? sorted_val[c0 + lane] : 0ull;
    const uint32_t inst = (uint32_t)(v >> 32);
    float gl = c0 + lane < end ? G[inst] : 0.f;
    if ((uint32_t)v & TARGET_BIT) gl = -gl;                                        // the negative item of a pair (bpr.hpp:80, 83)
    float up[PF][NI], uq[PF][NI];
    auto fetch_group = [&](uint32_t g0) {              // past the end: the last contribution again (never used)
#pragma unroll
      for (int s = 0; s < PF; ++s)
        vload<NI>(uq[s], UVpre + (size_t)(uint32_t)__builtin_amdgcn_readlane((int)inst, min(g0 + (uint32_t)s, cnt - 1u)) * hp.Kp + lo);
    };
    fetch_group(0);
    for (uint32_t x0 = 0; x0 < cnt; x0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int i = 0; i < NI; ++i) up[s][i] = uq[s][i];
      if (x0 + PF < cnt) fetch_group(x0 + PF);
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const uint32_t x = x0 + (uint32_t)s;
        if (x >= cnt) break;
        const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl), x));
        if (bias_term) ada_step_t<ADA>(hp, b, ba, fmaf(lam2, b, g));                      // imf.hpp:93, 98-100, 110
#pragma unroll
        for (int i = 0; i < NI; ++i) ada_step_t<ADA>(hp, w[i], a[i], fmaf(g, up[s][i], lam2 * w[i]));   // imf.hpp:95, 103-106, 114
      }
    }
  }
  vstore<NI>(IV + (size_t)item * hp.Kp + lo, w);
  vstore<NI>(IV_ag + (size_t)item * hp.Kp + lo, a);
  if (lane == 0) { IB[item] = b; IB_ag[item] = ba; }
  };
  if (hp.adagrad) run(std::true_type{}); else run(std::false_type{});
}

}  // namespace cdae
