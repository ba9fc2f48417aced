// K7 on the matrix cores: top-k recommendation for many users at once (cdae.hpp:162-196, evaluation.hpp:135-145).
//
// scores[user][item] = z_user . D[item] + b'[item] is a [users x K] x [K x items] product — GEMM-shaped, unlike the
// sampled training path — so it runs on MFMA with fp32 operands (v_mfma_f32_32x32x2_f32, exact fp32 products and
// accumulation: ranking parity with the fp64 oracle needs more than bf16).  A 256-thread workgroup serves 128 users:
// wavefront w keeps the z rows of its 32 users in registers as the MFMA's B operand for the whole launch; the
// workgroup streams the decoder matrix through LDS in tiles of 32 items (double-buffered, rows padded to 8 NCH + 4
// floats so that the ds_read_b128 of 16 consecutive rows hit distinct banks) as the A operand.  After K / 2 MFMAs a
// lane holds 16 item scores of ONE user (C[item][user]: column = lane & 31), adds b', drops the user's training items
// (one bit per (user, item), built by rated_bits_kernel; a tile of 32 items is one 32-bit word) and keeps its own sorted
// top-16; the two lanes of a user merge through LDS at the end.  Ties resolve to the lower item id like the reference's
// heap walk over ascending ids.
//
// Contraction order: the K axis is cut into chunks of 8; lane half h = lane >> 5 owns elements 4h..4h+3 of every chunk,
// so both operands are plain float4 loads (A and B only have to agree on the order in which k is summed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cdae_kernels.hpp"

namespace cdae {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int REC_TOPK_MAX = 16;       // per-lane list length; larger topk falls back to recommend_kernel
constexpr int REC_USERS_PER_BLOCK = 128;

// bits[(u - u0) * words + item / 32] = OR of 1 << (item % 32) over the training items of users [u0, u0 + nu).  No memset launch in front
// (round 5): the wavefront that owns a user's row of words builds it in LDS and writes every word once (item spaces up to 65 536: 8 KiB
// per wavefront); larger rows are cleared in global memory by their wavefront, which waits for its own stores (s_waitcnt: same-wave order
// is all that is needed — an agent-scope fence here wrote back the L2 under the training kernels of the other stream) before its atomics.
constexpr uint32_t RATED_LDS_WORDS = 2048;
__global__ void __launch_bounds__(256)
rated_bits_kernel(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, uint64_t u0, uint32_t nu,
                  uint32_t words, uint32_t* __restrict__ bits) {
  __shared__ uint32_t lrow[4][RATED_LDS_WORDS];
  const uint32_t wid = threadIdx.x / WAVE;
  const uint32_t slot = blockIdx.x * (blockDim.x / WAVE) + wid;
  const uint32_t lane = threadIdx.x % WAVE;
  if (slot >= nu) return;
  const int64_t r0 = row_ptr[u0 + slot], r1 = row_ptr[u0 + slot + 1];
  uint32_t* out = bits + (size_t)slot * words;
  if (words <= RATED_LDS_WORDS) {
    uint32_t* w = lrow[wid];
    for (uint32_t i = lane; i < words; i += WAVE) w[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t p = r0 + lane; p < r1; p += WAVE) {
      const uint32_t item = col[p];
      atomicOr(&w[item >> 5], 1u << (item & 31u));                 // LDS
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < words; i += WAVE) out[i] = w[i];
    return;
  }
  {                                                                // 16-byte stores over the aligned middle of the row, words at its ends
    const uint32_t head = min(words, (uint32_t)((16u - ((uintptr_t)out & 15u)) & 15u) / 4u), quads = (words - head) / 4u;
    for (uint32_t i = lane; i < head; i += WAVE) out[i] = 0u;
    uint4* o4 = reinterpret_cast<uint4*>(out + head);
    for (uint32_t i = lane; i < quads; i += WAVE) o4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t i = head + 4u * quads + lane; i < words; i += WAVE) out[i] = 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int64_t p = r0 + lane; p < r1; p += WAVE) {
    const uint32_t item = col[p];
    atomicOr(out + (item >> 5), 1u << (item & 31u));
  }
}

constexpr size_t recommend_mfma_lds_bytes(int nch) {
  const size_t tiles = 2 * 32 * (size_t)(8 * nch + 4) * sizeof(float), merge = 2 * 4 * 64 * (size_t)REC_TOPK_MAX * sizeof(float);
  return tiles > merge ? tiles : merge;
}

template <int NCH>
__global__ void __launch_bounds__(256)
recommend_mfma_kernel(HyperParams hp, const float* __restrict__ Z /* [nu x Kp] */, uint32_t nu,
                      const float* __restrict__ D, const float* __restrict__ bp,
                      const uint32_t* __restrict__ bits, uint32_t words, uint32_t topk, uint32_t* __restrict__ out) {
  constexpr int KC = 8 * NCH;                     // contraction length (>= K; pad columns are zero in Z and D)
  constexpr int ROW = KC + 4;                     // LDS row stride in floats
  constexpr int TILE = 32;
  extern __shared__ __attribute__((aligned(16))) char rec_smem[];   // max(2 tiles, merge scratch): see recommend_mfma_lds_bytes
  float (*tile)[TILE * ROW] = reinterpret_cast<float (*)[TILE * ROW]>(rec_smem);
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  const uint32_t col_u = lane & 31u, half = lane >> 5;
  const uint32_t user = blockIdx.x * REC_USERS_PER_BLOCK + wave * 32u + col_u;
  const uint32_t user_ld = min(user, nu - 1u);

  // B operand: this lane's user, elements 8c + 4 half .. + 3 of every chunk
  float4 bz[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    bz[c] = *reinterpret_cast<const float4*>(Z + (size_t)user_ld * hp.Kp + 8 * c + 4 * half);

  float tv[REC_TOPK_MAX];
  uint32_t ti[REC_TOPK_MAX];
#pragma unroll
  for (int j = 0; j < REC_TOPK_MAX; ++j) { tv[j] = -INFINITY; ti[j] = 0xFFFFFFFFu; }

  const uint32_t n_tiles = (hp.num_items + TILE - 1) / TILE;
  // cooperative tile load: TILE rows x (KC / 4) float4 = 8 NCH float4 per row; thread t takes float4 t, t + 256, ...
  constexpr int F4_PER_ROW = KC / 4;
  constexpr int F4_PER_TILE = TILE * F4_PER_ROW;
  constexpr int F4_PER_THREAD = (F4_PER_TILE + 255) / 256;
  float4 stage[F4_PER_THREAD];
  auto fetch = [&](uint32_t t) {
#pragma unroll
    for (int q = 0; q < F4_PER_THREAD; ++q) {
      const uint32_t f = threadIdx.x + 256u * q;
      const uint32_t r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
      const uint32_t item = min(t * TILE + r, hp.num_items - 1u);
      stage[q] = f < (uint32_t)F4_PER_TILE ? *reinterpret_cast<const float4*>(D + (size_t)item * hp.Kp + 4 * c4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int q = 0; q < F4_PER_THREAD; ++q) {
      const uint32_t f = threadIdx.x + 256u * q;
      const uint32_t r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
      if (f < (uint32_t)F4_PER_TILE) *reinterpret_cast<float4*>(&tile[buf][r * ROW + 4 * c4]) = stage[q];
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();

  for (uint32_t t = 0; t < n_tiles; ++t) {
    const int buf = (int)(t & 1u);
    if (t + 1 < n_tiles) fetch(t + 1);                              // global loads of the next tile fly under the MFMAs
    const uint32_t word = bits[(size_t)user_ld * words + t];        // this user's training items among the tile's 32
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* arow = &tile[buf][col_u * ROW + 4 * half];         // A operand: item row (lane & 31) of the tile
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 8 * c);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bz[c].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bz[c].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bz[c].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bz[c].w, acc, 0, 0, 0);
    }
    // C[item][user]: lane holds items 8 (r / 4) + 4 half + (r % 4), r = 0..15, of user column lane & 31
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t i0 = t * TILE + 8u * q + 4u * half;
      const float4 b4 = i0 + 3u < hp.num_items ? *reinterpret_cast<const float4*>(bp + i0)
                                                : make_float4(i0 < hp.num_items ? bp[i0] : 0.f, i0 + 1u < hp.num_items ? bp[i0 + 1u] : 0.f,
                                                              i0 + 2u < hp.num_items ? bp[i0 + 2u] : 0.f, 0.f);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t item = i0 + (uint32_t)e;
        const float s = acc[4 * q + e] + bb[e];
        const bool ok = item < hp.num_items && !((word >> (8u * q + 4u * half + (uint32_t)e)) & 1u);
        if (ok && s > tv[REC_TOPK_MAX - 1]) {                       // strict: an equal score keeps the earlier (lower) id
          tv[REC_TOPK_MAX - 1] = s; ti[REC_TOPK_MAX - 1] = item;
#pragma unroll
          for (int j = REC_TOPK_MAX - 1; j >= 1; --j) {
            const bool up = tv[j] > tv[j - 1];
            const float fv = tv[j]; const uint32_t fi = ti[j];
            tv[j] = up ? tv[j - 1] : fv; ti[j] = up ? ti[j - 1] : fi;
            tv[j - 1] = up ? fv : tv[j - 1]; ti[j - 1] = up ? fi : ti[j - 1];
          }
        }
      }
    }
    if (t + 1 < n_tiles) commit(buf ^ 1);
    __syncthreads();
  }

  // merge the two lanes of every user (lists are sorted by score desc, id asc within a lane)
  float* mv = &tile[0][0];                                           // [4 waves][64 lanes][16] scores, then ids
  uint32_t* mi = reinterpret_cast<uint32_t*>(mv + 4 * 64 * REC_TOPK_MAX);
#pragma unroll
  for (int j = 0; j < REC_TOPK_MAX; ++j) {
    mv[(wave * 64 + lane) * REC_TOPK_MAX + j] = tv[j];
    mi[(wave * 64 + lane) * REC_TOPK_MAX + j] = ti[j];
  }
  __syncthreads();
  if (half == 0 && user < nu) {
    const float* va = mv + (wave * 64 + lane) * REC_TOPK_MAX;
    const float* vb = mv + (wave * 64 + lane + 32) * REC_TOPK_MAX;
    const uint32_t* ia = mi + (wave * 64 + lane) * REC_TOPK_MAX;
    const uint32_t* ib = mi + (wave * 64 + lane + 32) * REC_TOPK_MAX;
    uint32_t pa = 0, pb = 0;
    for (uint32_t j = 0; j < topk; ++j) {
      const float a = pa < (uint32_t)REC_TOPK_MAX ? va[pa] : -INFINITY, b = pb < (uint32_t)REC_TOPK_MAX ? vb[pb] : -INFINITY;
      const uint32_t xa = pa < (uint32_t)REC_TOPK_MAX ? ia[pa] : 0xFFFFFFFFu, xb = pb < (uint32_t)REC_TOPK_MAX ? ib[pb] : 0xFFFFFFFFu;
      const bool take_a = a > b || (a == b && xa <= xb);
      out[(size_t)user * topk + j] = take_a ? xa : xb;
      if (take_a) ++pa; else ++pb;
    }
  }
}

// ---- TOPN metrics on the device (TOPN_Evaluation::evaluate evaluation.hpp:113-181, evaluate_rec_list :183-219) ------------------
// The reference scores each user's top-10 list against the user's test items on num_thread host threads and averages the eight
// columns P@1 P@5 P@10 R@1 R@5 R@10 MAP@5 MAP@10 over the users that have test items (:160-166).  Here the lists never leave the
// device: topn_user_kernel turns the list of one user (a thread each) into that user's eight fp64 terms r[c] / n_test_users —
// the same expressions in the same order as evaluate_rec_list, membership by binary search in the sorted test row — and
// topn_sum_kernel adds the terms of all users IN USER ORDER, one lane per column, so that the eight means carry the bits of the
// reference's sequential `rets[c] += r[c] / n` loop (users without test items contribute +0.0, which changes no bit of a
// non-negative sum).  The integer hit counts at 1 / 5 / 10 are summed with atomics (order-free).
__global__ void __launch_bounds__(256)
topn_user_kernel(const uint32_t* __restrict__ rec, uint32_t topk, uint64_t u0, uint32_t nu, const int64_t* __restrict__ test_ptr,
                 const uint32_t* __restrict__ test_col, double n_test_users, double* __restrict__ per_user,
                 unsigned long long* __restrict__ hits) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nu) return;
  const uint64_t u = u0 + slot;
  const int64_t t0 = test_ptr[u], t1 = test_ptr[u + 1];
  double r[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  if (t1 > t0) {
    const double nt = (double)(t1 - t0);
    const uint32_t top = topk < 20u ? topk : 20u;                       // evaluation.hpp:186,191
    double hit = 0., map5 = 0., map10 = 0.;
    uint32_t h1 = 0, h5 = 0, h10 = 0;
    for (uint32_t i = 0; i < top; ++i) {
      const uint32_t item = rec[(size_t)slot * topk + i];
      int64_t lo = t0, hi = t1;                                         // first position with test_col >= item
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (test_col[mid] < item) lo = mid + 1; else hi = mid; }
      if (lo < t1 && test_col[lo] == item) {
        hit += 1.;
        if (i < 5) map5 += hit / (double)(i + 1);
        if (i < 10) map10 += hit / (double)(i + 1);
        h1 += i < 1; h5 += i < 5; h10 += i < 10;
      }
      if (i == 0) { r[0] = hit; r[3] = hit / nt; }
      else if (i == 4) { r[1] = hit / 5.; r[4] = hit / nt; }
      else if (i == 9) { r[2] = hit / 10.; r[5] = hit / nt; }
    }
    r[6] = map5 / (nt < 5. ? nt : 5.);
    r[7] = map10 / (nt < 10. ? nt : 10.);
    if (h1) atomicAdd(hits + 0, (unsigned long long)h1);
    if (h5) atomicAdd(hits + 1, (unsigned long long)h5);
    if (h10) atomicAdd(hits + 2, (unsigned long long)h10);
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = r[c] / n_test_users;             // evaluation.hpp:162-166 divides per user, then adds
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) per_user[u * 8 + c] = r[c];
}

// out[c] = ((per_user[0][c] + per_user[1][c]) + per_user[2][c]) + ...  — one wavefront, lane c < 8 owns column c; 16 users'
// terms are requested ahead of the dependent chain of fp64 adds
__global__ void __launch_bounds__(64)
topn_sum_kernel(const double* __restrict__ per_user, uint64_t num_users, double* __restrict__ out) {
  const uint32_t c = threadIdx.x;
  if (c >= 8) return;
  double acc = 0.;
  uint64_t u = 0;
  for (; u + 16 <= num_users; u += 16) {
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = per_user[(u + j) * 8 + c];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += v[j];
  }
  for (; u < num_users; ++u) acc += per_user[u * 8 + c];
  out[c] = acc;
}

}  // namespace cdae
