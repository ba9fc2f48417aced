// cdae_full_kernels.hpp — full-output decode on the MFMA matrix cores (BASELINE.json configs[1], configs[4]).
//
// The reference's training decode is always sampled (cdae.hpp:217-293); its only dense decode over all items
// is recommend() (cdae.hpp:176-186).  The north star extends training to "every unrated item is a negative with
// target 0".  Per block of B users, from the block-start parameters (oracle: Oracle::train_users_full):
//     Y  = Z D^T + b'            [B x I]    GEMM 1, loss' fused in the epilogue -> G (and G^T), bf16
//     hg = G D                   [B x K]    GEMM 2, split along the item dimension, fp32 atomics
//     dD = G^T Z                 [I x K]    GEMM 3
// then one AdaGrad/SGD step per decoder row with dD[j] + lambda D[j] (+ the summed input gradient in tied mode),
// one per b'[j], and the hidden-layer steps of the sampled path.  With B = 1 this is exactly the reference loop
// fed all unrated items.  This is the compute-bound end of the path (6 K I flop per user), so it runs on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the operands are bf16 copies made per batch.
//
// All three products are "NT" GEMMs  C[m][n] = sum_k A[m][k] * Bm[n][k]  over row-major bf16 operands with the
// contraction index contiguous (the per-batch transposed copies D^T, Z^T, G^T make that true), so an MFMA
// fragment is one 16-byte global load per lane: lane l holds A[m0 + (l & 31)][k0 + 8 (l >> 5) .. +7] and the same
// slice of Bm's row n0 + (l & 31).  A wavefront owns a 64 x 64 tile of C (2 x 2 MFMA tiles, 64 accumulator
// VGPRs), a 256-thread workgroup 128 x 128.  Round 1 feeds the fragments straight from L1/L2 (no LDS staging).
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cdae_kernels.hpp"

namespace cdae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32 [R x C] (row stride ld_src) -> bf16 [Rp x C] with rows >= R zero, and its transpose [C x Rp].
// 64 x 64 tiles through LDS so that both images are written in coalesced runs.
template <typename SrcT>
__device__ __forceinline__ void to_bf16_transpose_body(const SrcT* __restrict__ src, uint32_t R, uint32_t C, uint32_t ld_src, uint32_t Rp,
                                                       __bf16* __restrict__ dst /* nullptr: only the transposed image */, __bf16* __restrict__ dstT) {
  __shared__ float tile[64][65];
  const uint32_t r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const uint32_t r = r0 + i / 64, c = c0 + i % 64;
    const float v = (r < R && c < C) ? (float)src[(size_t)r * ld_src + c] : 0.f;
    tile[i / 64][i % 64] = v;
    if (dst && r < Rp && c < C) dst[(size_t)r * C + c] = (__bf16)v;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const uint32_t c = c0 + i / 64, r = r0 + i % 64;
    if (r < Rp && c < C) dstT[(size_t)c * Rp + r] = (__bf16)tile[i % 64][i / 64];
  }
}
// D^T image from the row-major bf16 image the row step has just written (item spaces >= 32768: full_rows_wave_kernel writes Db, the
// 2-byte scattered stores of D^T would not pay there): half the traffic of converting from the fp32 matrix
__global__ void __launch_bounds__(256)
bf16_transpose_kernel(const __bf16* __restrict__ src, uint32_t R, uint32_t C, uint32_t Rp, __bf16* __restrict__ dstT) {
  // 64 x 64 tile, 16-byte global accesses both ways (a 2-byte-per-thread form ran at a fifth of the HBM rate); LDS rows of 66
  // elements: the 2-byte column reads of the write pass walk 33 banks
  __shared__ uint16_t tile[64][66];
  const uint32_t r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const uint16_t* s16 = reinterpret_cast<const uint16_t*>(src);
  uint16_t* d16 = reinterpret_cast<uint16_t*>(dstT);
#pragma unroll
  for (uint32_t pass = 0; pass < 2; ++pass) {
    const uint32_t rl = threadIdx.x / 8 + 32 * pass, c8 = (threadIdx.x % 8) * 8, r = r0 + rl;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < R) v = *reinterpret_cast<const uint4*>(s16 + (size_t)r * C + c0 + c8);          // C is a multiple of 64: in bounds, aligned
    uint32_t* t32 = reinterpret_cast<uint32_t*>(&tile[rl][c8]);
    t32[0] = v.x; t32[1] = v.y; t32[2] = v.z; t32[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t pass = 0; pass < 2; ++pass) {
    const uint32_t cl = threadIdx.x / 8 + 32 * pass, i8 = (threadIdx.x % 8) * 8;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[i8 + 2 * j][cl] | ((uint32_t)tile[i8 + 2 * j + 1][cl] << 16);
    if (r0 + i8 < Rp) *reinterpret_cast<uint4*>(d16 + (size_t)(c0 + cl) * Rp + r0 + i8) = make_uint4(w[0], w[1], w[2], w[3]);   // Rp % 64 == 0
  }
}
__global__ void __launch_bounds__(256)
to_bf16_transpose_kernel(const float* __restrict__ src, uint32_t R, uint32_t C, uint32_t ld_src, uint32_t Rp,
                         __bf16* __restrict__ dst, __bf16* __restrict__ dstT) {
  __shared__ float tile[64][65];
  const uint32_t r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const uint32_t r = r0 + i / 64, c = c0 + i % 64;
    const float v = (r < R && c < C) ? src[(size_t)r * ld_src + c] : 0.f;
    tile[i / 64][i % 64] = v;
    if (r < Rp && c < C) dst[(size_t)r * C + c] = (__bf16)v;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const uint32_t c = c0 + i / 64, r = r0 + i % 64;
    if (r < Rp && c < C) dstT[(size_t)c * Rp + r] = (__bf16)tile[i % 64][i / 64];
  }
}

// Two matrices of the same width in one launch (the batch's D and Z copies: a launch each was 2 x 9 us of a 100 us Yelp-shape step):
// tile rows [0, RpA / 64) belong to A, the rest to B.
__global__ void __launch_bounds__(256)
to_bf16_transpose_pair_kernel(const float* __restrict__ srcA, uint32_t RA, uint32_t RpA, __bf16* __restrict__ dstA, __bf16* __restrict__ dstTA,
                              const float* __restrict__ srcB, uint32_t RB, uint32_t RpB, __bf16* __restrict__ dstB, __bf16* __restrict__ dstTB,
                              uint32_t C, uint32_t ld_src) {
  __shared__ float tile[64][65];
  const bool second = blockIdx.y >= RpA / 64;
  const float* src = second ? srcB : srcA;
  const uint32_t R = second ? RB : RA, Rp = second ? RpB : RpA;
  __bf16* dst = second ? dstB : dstA;
  __bf16* dstT = second ? dstTB : dstTA;
  const uint32_t r0 = (second ? blockIdx.y - RpA / 64 : blockIdx.y) * 64, c0 = blockIdx.x * 64;
  for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const uint32_t r = r0 + i / 64, c = c0 + i % 64;
    const float v = (r < R && c < C) ? src[(size_t)r * ld_src + c] : 0.f;
    tile[i / 64][i % 64] = v;
    if (r < Rp && c < C) dst[(size_t)r * C + c] = (__bf16)v;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const uint32_t c = c0 + i / 64, r = r0 + i % 64;
    if (r < Rp && c < C) dstT[(size_t)c * Rp + r] = (__bf16)tile[i % 64][i / 64];
  }
}

enum { EPI_LOSS = 0, EPI_ATOMIC = 1, EPI_STORE = 2 };

struct GemmEpilogue {
  // EPI_LOSS: g = loss'(acc + bp[n], 0) for m < rows_live, n < cols_live, else 0; G[m][n] and GT[n][m] (bf16)
  const float* bp;
  __bf16* G; uint32_t ldg;
  __bf16* GT; uint32_t ldgt;
  uint32_t rows_live, cols_live, loss_type;
  // EPI_ATOMIC / EPI_STORE: fp32 C with row stride ldc (ATOMIC: only rows < rows_live)
  float* Cout; uint32_t ldc;
  // EPI_STORE with a split contraction: split z stores its partial product at Cout + z * split_stride (summed in fixed order
  // by the consumer: deterministic, unlike EPI_ATOMIC); 0 when the contraction is not split
  size_t split_stride;
  // EPI_STORE in gemm_nt_bf16_lds_kernel only (round 4, full-output small shapes on one stream): bias_blocks > 0 = the launch's leading
  // workgroups run the hidden-bias recurrence (hidden_bias_role) over the users whose delta rows start at bias_delta — the first half
  // of the block's users beside GEMM 3, the second half beside the row launch — instead of all of them beside the row launch alone
  uint32_t bias_blocks, bias_nb;
  const float* bias_delta; float* bias_b; float* bias_b_ag;
  HyperParams bias_hp;
};

// Epilogue of one wavefront's 64 x 64 tile (2 x 2 MFMA tiles), shared by the direct and the LDS-staged kernel.
template <int EPI>
__device__ __forceinline__ void gemm_tile_epilogue(f32x16 (&acc)[2][2], const uint32_t m_base, const uint32_t n_base, const uint32_t lane,
                                                   const GemmEpilogue& ep) {
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t n = n_base + j * 32 + (lane & 31);
      if constexpr (EPI == EPI_LOSS) {
        const float bias = n < ep.cols_live ? ep.bp[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                          // rows m0 .. m0+3 are consecutive: one 8-byte G^T store
          const uint32_t m0 = m_base + i * 32 + 8 * q + 4 * (lane >> 5);
          bf16x4 gt;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t m = m0 + r;
            float g = 0.f;
            if (m < ep.rows_live && n < ep.cols_live) g = loss_grad(ep.loss_type, acc[i][j][4 * q + r] + bias, 0.f);
            gt[r] = (__bf16)g;
            ep.G[(size_t)m * ep.ldg + n] = (__bf16)g;
          }
          *reinterpret_cast<bf16x4*>(ep.GT + (size_t)n * ep.ldgt + m0) = gt;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if constexpr (EPI == EPI_ATOMIC) {
            if (m < ep.rows_live) unsafeAtomicAdd(ep.Cout + (size_t)m * ep.ldc + n, acc[i][j][r]);
          } else {
            ep.Cout[(size_t)m * ep.ldc + n] = acc[i][j][r];
          }
        }
      }
    }
}

template <int EPI>
__global__ void __launch_bounds__(256)
gemm_nt_bf16_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ Bm, uint32_t M, uint32_t N, uint32_t Kd,
                    uint32_t lda, uint32_t ldb, uint32_t k_per_split, GemmEpilogue ep) {
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;
  const uint32_t m_base = blockIdx.y * (blockDim.x / 2) + (wid >> 1) * 64;   // 256 threads: 128 rows per workgroup; 128 threads: 64
  const uint32_t n_base = blockIdx.x * 128 + (wid & 1) * 64;
  if (m_base >= M || n_base >= N) return;
  const uint32_t k_begin = blockIdx.z * k_per_split;
  const uint32_t k_end = min(Kd, k_begin + k_per_split);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const uint32_t frag_row = lane & 31, frag_k = (lane >> 5) * 8;
  const __bf16* a0 = A + (size_t)(m_base + frag_row) * lda + frag_k;
  const __bf16* a1 = a0 + (size_t)32 * lda;
  const __bf16* b0 = Bm + (size_t)(n_base + frag_row) * ldb + frag_k;
  const __bf16* b1 = b0 + (size_t)32 * ldb;
  for (uint32_t k = k_begin; k < k_end; k += 16) {
    const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(a0 + k);
    const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(a1 + k);
    const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(b0 + k);
    const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(b1 + k);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
  }
  gemm_tile_epilogue<EPI>(acc, m_base, n_base, lane, ep);
}

// LDS-staged form of the same NT product for the shapes where it is the whole cost (K > 256: BASELINE configs[4], 1 M items x
// K = 512).  A 256-thread workgroup owns a 128 x 128 tile of C (wavefront: 64 x 64, as above) and walks the contraction in
// steps of 64: the 128 x 64 slices of A and Bm (16 KiB each) go global -> LDS by 16-byte LDS-DMA (`global_load_lds`: no
// staging registers, no ds_write pass), double-buffered, and every fragment is one ds_read_b128 shared by the two wavefronts
// that need it — the direct kernel above moved every fragment through L1 once per wavefront (4 KiB per 4 MFMAs: TA-bound at
// ~165 TFLOP/s).  LDS image of a slice: row-major, 128-byte rows, lane-linear per DMA instruction (8 rows x 8 sixteen-byte
// slots); slot c of row r holds source column c ^ ((r >> 1) & 7), which makes the 16-lane groups of ds_read_b128
// ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md §LDS) hit 16 distinct slots of the 256-byte bank row.  The permutation is
// applied to the per-lane SOURCE address (it stays inside one 128-byte line, so coalescing is unchanged).
// Requirements (host-checked): M % 128 == 0, contraction range % 64 == 0, rows of Bm beyond N are clamped (their C is dropped).
constexpr int GEMM_BK = 64;
constexpr int GEMM_SLICE_BYTES = 128 * GEMM_BK * 2;      // one operand slice: 128 rows x 64 bf16

// Workgroup -> tile mapping, XCD-aware: workgroup ids go round-robin over the 8 XCDs (id & 7), each with its own 4 MiB L2.
// The tiles that share the big streamed operand — GEMM 1: the M-tiles of one D tile; GEMM 3: the N-tiles of one G^T tile;
// GEMM 2: all (m, n) tiles of one contraction split — are given consecutive slots of ONE XCD, so that operand comes from
// HBM once and from that L2 afterwards (with the natural x-fastest order the 8 user-tiles of a D tile ran 8192 workgroups
// apart: 8 GB of D traffic per 1024-user block at 1 M items instead of 1 GB).
struct GemmGrid {
  uint32_t Mt, Nt, Zt;      // tiles along M, N and contraction splits
  uint32_t mode;            // 0: inner = M-tiles, outer = N-tiles; 1: inner = N-tiles, outer = M-tiles; 2: inner = (m, n), outer = split
  __host__ __device__ uint32_t inner() const { return mode == 0 ? Mt : (mode == 1 ? Nt : Mt * Nt); }
  __host__ __device__ uint32_t outer() const { return mode == 0 ? Nt : (mode == 1 ? Mt : Zt); }
  __host__ uint32_t workgroups() const { return 8u * inner() * ((outer() + 7u) / 8u); }
};

template <int EPI>
__global__ void __launch_bounds__(256)
gemm_nt_bf16_lds_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ Bm, uint32_t M, uint32_t N, uint32_t Kd,
                        uint32_t lda, uint32_t ldb, uint32_t k_per_split, GemmEpilogue ep, GemmGrid gg) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * GEMM_SLICE_BYTES];     // [buffer][A | B][128 rows][128 B]
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;
  uint32_t bid = blockIdx.x;
  if (EPI == EPI_STORE && ep.bias_blocks) {                                // leading workgroups: part of the hidden-bias recurrence (see GemmEpilogue)
    if (bid < ep.bias_blocks) {
      if (ep.bias_hp.adagrad) hidden_bias_role<true, true>(ep.bias_hp, bid * blockDim.x + threadIdx.x, ep.bias_nb, ep.bias_delta, ep.bias_b, ep.bias_b_ag);
      else hidden_bias_role<false, true>(ep.bias_hp, bid * blockDim.x + threadIdx.x, ep.bias_nb, ep.bias_delta, ep.bias_b, ep.bias_b_ag);
      return;
    }
    bid -= ep.bias_blocks;
  }
  uint32_t mt, nt, zt;
  {
    const uint32_t xcd = bid & 7u, j = bid >> 3, in = gg.inner();
    const uint32_t t = j % in, o = (j / in) * 8u + xcd;
    if (o >= gg.outer()) return;                                           // (whole workgroup)
    if (gg.mode == 0) { mt = t; nt = o; zt = 0; }
    else if (gg.mode == 1) { nt = t; mt = o; zt = 0; }
    else { mt = t / gg.Nt; nt = t % gg.Nt; zt = o; }
  }
  const uint32_t m_tile = mt * 128u, n_tile = nt * 128u;
  const uint32_t m_base = m_tile + (wid >> 1) * 64u, n_base = n_tile + (wid & 1u) * 64u;
  const uint32_t k_begin = zt * k_per_split;
  const uint32_t k_end = min(Kd, k_begin + k_per_split);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: a slice is 16 DMA instructions of 1 KiB (8 rows each); wavefront w issues instructions 4w .. 4w+3 of A and of B
  const uint32_t st_row = lane >> 3, st_slot = lane & 7u;
  const __bf16* a_src[4];
  const __bf16* b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t r = (wid * 4u + q) * 8u + st_row;                       // row of the slice this lane feeds
    const uint32_t c = st_slot ^ ((r >> 1) & 7u);                          // source column (16-byte units) for LDS slot st_slot
    a_src[q] = A + (size_t)(m_tile + r) * lda + 8u * c;
    b_src[q] = Bm + (size_t)min(n_tile + r, N - 1u) * ldb + 8u * c;
  }
  auto stage = [&](uint32_t k, int buf) {
    char* base = smem + buf * 2 * GEMM_SLICE_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t off = (wid * 4u + q) * 1024u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + GEMM_SLICE_BYTES + off), 16, 0, 0);
    }
  };
  // fragment addresses: row = lane & 31 (+32 i) of the wavefront's 64 rows, k sub-step s: source column 2 s + (lane >> 5)
  const uint32_t f_row = lane & 31u, f_half = lane >> 5;
  uint32_t a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t ra = (wid >> 1) * 64u + i * 32u + f_row, rb = (wid & 1u) * 64u + i * 32u + f_row;
    a_off[i] = ra * 128u; a_sw[i] = (ra >> 1) & 7u;
    b_off[i] = GEMM_SLICE_BYTES + rb * 128u; b_sw[i] = (rb >> 1) & 7u;
  }

  stage(k_begin, 0);
  __syncthreads();                                                         // (carries the vmcnt(0) that lands the DMA)
  int buf = 0;
  for (uint32_t k = k_begin; k < k_end; k += GEMM_BK, buf ^= 1) {
    if (k + GEMM_BK < k_end) stage(k + GEMM_BK, buf ^ 1);
    const char* base = smem + buf * 2 * GEMM_SLICE_BYTES;
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
      const uint32_t c = 2u * s + f_half;
      const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(base + a_off[0] + ((c ^ a_sw[0]) << 4));
      const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(base + a_off[1] + ((c ^ a_sw[1]) << 4));
      const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(base + b_off[0] + ((c ^ b_sw[0]) << 4));
      const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(base + b_off[1] + ((c ^ b_sw[1]) << 4));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();                                                       // next slice landed; this one is free to be overwritten
  }
  if constexpr (EPI == EPI_LOSS) {
    // g = loss'(y + b', 0) as bf16, written as G [users x items] AND G^T [items x users].  Straight from the accumulator
    // layout that is 64 two-byte + 16 eight-byte scattered stores per lane (4 GB per batch at 1 M items in 64- and 8-byte
    // pieces); instead the workgroup's 128 x 128 tile goes through LDS (the staging buffers are free now) once per
    // orientation and leaves as 16-byte pieces of whole 256-byte rows.
    constexpr uint32_t TS = 272;                                            // tile row stride (bytes): 16-byte aligned, 16 lanes x 16 B = one bank row
    const uint32_t half = lane >> 5, col = lane & 31u;
    __bf16 gb[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t n = n_base + j * 32 + col;
        const float bias = n < ep.cols_live ? ep.bp[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float g = 0.f;
          if (m < ep.rows_live && n < ep.cols_live) g = loss_grad(ep.loss_type, acc[i][j][r] + bias, 0.f);
          gb[i][j][r] = (__bf16)g;
        }
      }
    const uint32_t wm = (wid >> 1) * 64u, wn = (wid & 1u) * 64u;
    // pass 1: [m][n] image -> G rows
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t ml = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, nl = wn + j * 32 + col;
          *reinterpret_cast<__bf16*>(smem + ml * TS + nl * 2u) = gb[i][j][r];
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t pc = threadIdx.x + 256u * q, row = pc >> 4, c16 = pc & 15u;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * TS + c16 * 16u);
      *reinterpret_cast<bf16x8*>(ep.G + (size_t)(m_tile + row) * ep.ldg + n_tile + c16 * 8u) = v;
    }
    __syncthreads();
    // pass 2: [n][m] image -> G^T rows (a lane holds 4 consecutive m per (i, j, q))
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t nl = wn + j * 32 + col, m0 = wm + i * 32 + 8 * q + 4 * half;
          const bf16x4 v = {gb[i][j][4 * q], gb[i][j][4 * q + 1], gb[i][j][4 * q + 2], gb[i][j][4 * q + 3]};
          *reinterpret_cast<bf16x4*>(smem + nl * TS + m0 * 2u) = v;
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t pc = threadIdx.x + 256u * q, row = pc >> 4, c16 = pc & 15u;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * TS + c16 * 16u);
      if (n_tile + row < N) *reinterpret_cast<bf16x8*>(ep.GT + (size_t)(n_tile + row) * ep.ldgt + m_tile + c16 * 8u) = v;
    }
    return;
  } else {
    if (m_base >= M || n_base >= N) return;
    GemmEpilogue es = ep;
    if constexpr (EPI == EPI_STORE) es.Cout += (size_t)zt * ep.split_stride;
    gemm_tile_epilogue<EPI>(acc, m_base, n_base, lane, es);
  }
}

// Deeper form for the big shapes (M % 256 == 0): 512 threads own a 256 x 128 tile (eight wavefronts, 4 x 2, 64 x 64 each:
// 85 flop per staged byte instead of 64) and keep TWO slices in flight across each barrier — three LDS stages of 48 KiB, counted
// `s_waitcnt vmcnt(6)` (a wavefront issues 6 DMA instructions per slice) and raw `s_barrier`; `__syncthreads()` would carry a
// `vmcnt(0)` and drain the pipeline (cdna_hip_programming.md §5).  Order per step k: wait for this wavefront's slice-k DMAs ->
// barrier (everyone's slice k has landed, everyone is done reading slice k-1) -> issue slice k+2 into the stage slice k-1
// occupied -> MFMAs on slice k.
constexpr int GEMM3S_STAGE_BYTES = (256 + 128) * GEMM_BK * 2;     // 48 KiB: A slice 256 x 64 bf16, then B slice 128 x 64
constexpr size_t gemm3s_lds_bytes() { return 3 * (size_t)GEMM3S_STAGE_BYTES; }

template <int EPI>
__global__ void __launch_bounds__(512)
gemm_nt_bf16_lds3_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ Bm, uint32_t M, uint32_t N, uint32_t Kd,
                         uint32_t lda, uint32_t ldb, uint32_t k_per_split, GemmEpilogue ep, GemmGrid gg) {
  extern __shared__ __attribute__((aligned(1024))) char smem3[];
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;      // wid 0..7
  uint32_t mt, nt, zt;
  {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, in = gg.inner();
    const uint32_t t = j % in, o = (j / in) * 8u + xcd;
    if (o >= gg.outer()) return;
    if (gg.mode == 0) { mt = t; nt = o; zt = 0; }
    else if (gg.mode == 1) { nt = t; mt = o; zt = 0; }
    else { mt = t / gg.Nt; nt = t % gg.Nt; zt = o; }
  }
  const uint32_t m_tile = mt * 256u, n_tile = nt * 128u;
  const uint32_t wm = (wid >> 1) * 64u, wn = (wid & 1u) * 64u;
  const uint32_t m_base = m_tile + wm, n_base = n_tile + wn;
  const uint32_t k_begin = zt * k_per_split;
  const uint32_t k_end = min(Kd, k_begin + k_per_split);
  const uint32_t n_steps = (k_end - k_begin) / GEMM_BK;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: a stage is 48 DMA instructions of 1 KiB (8 rows each): 32 of A then 16 of B; wavefront w issues 4w..4w+3 of A and
  // 2w, 2w+1 of B
  const uint32_t st_row = lane >> 3, st_slot = lane & 7u;
  const __bf16* a_src[4];
  const __bf16* b_src[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t r = (wid * 4u + q) * 8u + st_row;
    a_src[q] = A + (size_t)(m_tile + r) * lda + 8u * (st_slot ^ ((r >> 1) & 7u));
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t r = (wid * 2u + q) * 8u + st_row;
    b_src[q] = Bm + (size_t)min(n_tile + r, N - 1u) * ldb + 8u * (st_slot ^ ((r >> 1) & 7u));
  }
  auto stage = [&](uint32_t step, uint32_t slot) {
    char* base = smem3 + slot * GEMM3S_STAGE_BYTES;
    const uint32_t k = k_begin + step * GEMM_BK;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + (wid * 4u + q) * 1024u), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + 256 * 128 + (wid * 2u + q) * 1024u), 16, 0, 0);
  };
  const uint32_t f_row = lane & 31u, f_half = lane >> 5;
  uint32_t a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t ra = wm + i * 32u + f_row, rb = wn + i * 32u + f_row;
    a_off[i] = ra * 128u; a_sw[i] = (ra >> 1) & 7u;
    b_off[i] = 256u * 128u + rb * 128u; b_sw[i] = (rb >> 1) & 7u;
  }

  // prologue: slices 0 and 1 in flight
  stage(0, 0);
  if (n_steps > 1) stage(1, 1);
  uint32_t slot = 0;
  for (uint32_t step = 0; step < n_steps; ++step) {
    // this wavefront's DMAs of slice `step` have landed: all but the 6 of slice step+1 (none behind the last slice)
    if (step + 1 < n_steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                          // ... and everyone else's; slice step-1 is free
    const uint32_t nslot = slot == 0 ? 2u : slot - 1u;                     // (slot + 2) % 3: the stage slice step-1 occupied
    if (step + 2 < n_steps) stage(step + 2u, nslot);
    const char* base = smem3 + slot * GEMM3S_STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
      const uint32_t c = 2u * s + f_half;
      const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(base + a_off[0] + ((c ^ a_sw[0]) << 4));
      const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(base + a_off[1] + ((c ^ a_sw[1]) << 4));
      const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(base + b_off[0] + ((c ^ b_sw[0]) << 4));
      const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(base + b_off[1] + ((c ^ b_sw[1]) << 4));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
    }
    slot = slot == 2 ? 0u : slot + 1u;
  }
  __syncthreads();                                                         // every wavefront is done with the stages
  if constexpr (EPI == EPI_LOSS) {
    // as in the 128 x 128 kernel: the tile leaves through LDS, once per orientation, as 16-byte pieces of whole rows
    const uint32_t half = lane >> 5, col = lane & 31u;
    __bf16 gb[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t n = n_base + j * 32 + col;
        const float bias = n < ep.cols_live ? ep.bp[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float g = 0.f;
          if (m < ep.rows_live && n < ep.cols_live) g = loss_grad(ep.loss_type, acc[i][j][r] + bias, 0.f);
          gb[i][j][r] = (__bf16)g;
        }
      }
    constexpr uint32_t TS1 = 272;                                           // [256 m][128 n] image, row stride 272 B
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t ml = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, nl = wn + j * 32 + col;
          *reinterpret_cast<__bf16*>(smem3 + ml * TS1 + nl * 2u) = gb[i][j][r];
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t pc = threadIdx.x + 512u * q, row = pc >> 4, c16 = pc & 15u;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem3 + row * TS1 + c16 * 16u);
      *reinterpret_cast<bf16x8*>(ep.G + (size_t)(m_tile + row) * ep.ldg + n_tile + c16 * 8u) = v;
    }
    __syncthreads();
    constexpr uint32_t TS2 = 528;                                           // [128 n][256 m] image, row stride 528 B
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t nl = wn + j * 32 + col, m0 = wm + i * 32 + 8 * q + 4 * half;
          const bf16x4 v = {gb[i][j][4 * q], gb[i][j][4 * q + 1], gb[i][j][4 * q + 2], gb[i][j][4 * q + 3]};
          *reinterpret_cast<bf16x4*>(smem3 + nl * TS2 + m0 * 2u) = v;
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t pc = threadIdx.x + 512u * q, row = pc >> 5, c16 = pc & 31u;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem3 + row * TS2 + c16 * 16u);
      if (n_tile + row < N) *reinterpret_cast<bf16x8*>(ep.GT + (size_t)(n_tile + row) * ep.ldgt + m_tile + c16 * 8u) = v;
    }
  } else {
    if (m_base >= M || n_base >= N) return;
    GemmEpilogue es = ep;
    if constexpr (EPI == EPI_STORE) es.Cout += (size_t)zt * ep.split_stride;
    gemm_tile_epilogue<EPI>(acc, m_base, n_base, lane, es);
  }
}

// Wide form (round 3; M % 256 == 0 and N % 256 == 0: every product of the K > 256 path at BASELINE configs[4]): 512 threads own a
// 256 x 256 tile — eight wavefronts as 4 (M) x 2 (N), 64 x 128 each, 2 x 4 MFMA tiles — so a staged byte feeds 128 flop instead of
// 85.  All three products were bound by the global -> LDS fill rate, not by the matrix cores: their time is linear in the block size
// and their rate is the tile's flop per staged byte (profiles/r03_full_cfg5_block_sweep.txt).  A stage is 64 KiB (A slice 256 x 64
// bf16, then B slice 256 x 64), two stages: one slice in flight across each barrier — a step now carries 2048 SIMD cycles of MFMA per
// CU, twice the 256 x 128 kernel's, which is what the third stage bought there.  Same LDS image, swizzle and fragment reads as the
// kernels above (6 ds_read_b128 per 8 MFMAs instead of 4 per 4).  EPI_LOSS leaves through a [256][256] bf16 image, once per
// orientation, like the narrower kernels.
constexpr int GEMMW_STAGE_BYTES = (256 + 256) * GEMM_BK * 2;      // 64 KiB
constexpr uint32_t GEMMW_TS = 528;                                // epilogue image: 256 bf16 per row + 16 B (bank spread)
constexpr size_t gemmw_lds_bytes() { return 256 * (size_t)GEMMW_TS > 2 * (size_t)GEMMW_STAGE_BYTES ? 256 * (size_t)GEMMW_TS : 2 * (size_t)GEMMW_STAGE_BYTES; }

template <int EPI>
__global__ void __launch_bounds__(512)
gemm_nt_bf16_ldsw_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ Bm, uint32_t M, uint32_t N, uint32_t Kd,
                         uint32_t lda, uint32_t ldb, uint32_t k_per_split, GemmEpilogue ep, GemmGrid gg) {
  extern __shared__ __attribute__((aligned(1024))) char smemw[];
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;      // wid 0..7
  uint32_t mt, nt, zt;
  {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, in = gg.inner();
    const uint32_t t = j % in, o = (j / in) * 8u + xcd;
    if (o >= gg.outer()) return;
    if (gg.mode == 0) { mt = t; nt = o; zt = 0; }
    else if (gg.mode == 1) { nt = t; mt = o; zt = 0; }
    else { mt = t / gg.Nt; nt = t % gg.Nt; zt = o; }
  }
  const uint32_t m_tile = mt * 256u, n_tile = nt * 256u;
  const uint32_t wm = (wid >> 1) * 64u, wn = (wid & 1u) * 128u;
  const uint32_t m_base = m_tile + wm, n_base = n_tile + wn;
  const uint32_t k_begin = zt * k_per_split;
  const uint32_t k_end = min(Kd, k_begin + k_per_split);
  const uint32_t n_steps = (k_end - k_begin) / GEMM_BK;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: a stage is 64 DMA instructions of 1 KiB (8 rows each): 32 of A then 32 of B; wavefront w issues 4w..4w+3 of each
  const uint32_t st_row = lane >> 3, st_slot = lane & 7u;
  const __bf16* a_src[4];
  const __bf16* b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t r = (wid * 4u + q) * 8u + st_row;
    a_src[q] = A + (size_t)(m_tile + r) * lda + 8u * (st_slot ^ ((r >> 1) & 7u));
    b_src[q] = Bm + (size_t)min(n_tile + r, N - 1u) * ldb + 8u * (st_slot ^ ((r >> 1) & 7u));
  }
  auto stage = [&](uint32_t step, uint32_t slot) {
    char* base = smemw + slot * GEMMW_STAGE_BYTES;
    const uint32_t k = k_begin + step * GEMM_BK;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + (wid * 4u + q) * 1024u), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + 256 * 128 + (wid * 4u + q) * 1024u), 16, 0, 0);
  };
  const uint32_t f_row = lane & 31u, f_half = lane >> 5;
  uint32_t a_off[2], a_sw[2], b_off[4], b_sw[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t ra = wm + i * 32u + f_row;
    a_off[i] = ra * 128u; a_sw[i] = (ra >> 1) & 7u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t rb = wn + j * 32u + f_row;
    b_off[j] = 256u * 128u + rb * 128u; b_sw[j] = (rb >> 1) & 7u;
  }

  stage(0, 0);
  uint32_t slot = 0;
  for (uint32_t step = 0; step < n_steps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wavefront's DMAs of slice `step` have landed
    __builtin_amdgcn_s_barrier();                                          // ... and everyone else's; the other stage (slice step-1) is free
    if (step + 1 < n_steps) stage(step + 1u, slot ^ 1u);
    const char* base = smemw + slot * GEMMW_STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
      const uint32_t c = 2u * s + f_half;
      bf16x8 fa[2], fb[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(base + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    slot ^= 1u;
  }
  __syncthreads();                                                         // every wavefront is done with the stages
  const uint32_t half = lane >> 5, col = lane & 31u;
  if constexpr (EPI == EPI_LOSS) {
    // [256 m][256 n] image -> G rows; then [256 n][256 m] -> G^T rows: whole 512-byte rows leave as 16-byte pieces
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t n = n_base + j * 32 + col;
      const float bias = n < ep.cols_live ? ep.bp[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t ml = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, nl = wn + j * 32 + col;
          float g = 0.f;
          if (m_tile + ml < ep.rows_live && n < ep.cols_live) g = loss_grad(ep.loss_type, acc[i][j][r] + bias, 0.f);
          acc[i][j][r] = g;                                                // (kept for the second orientation)
          if (ep.G) *reinterpret_cast<__bf16*>(smemw + ml * GEMMW_TS + nl * 2u) = (__bf16)g;
        }
    }
    if (ep.G) {                                                            // (G = nullptr: GEMM 2 reads G^T, gemm_tn_bf16_kernel)
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t pc = threadIdx.x + 512u * q, row = pc >> 5, c16 = pc & 31u;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(smemw + row * GEMMW_TS + c16 * 16u);
        *reinterpret_cast<bf16x8*>(ep.G + (size_t)(m_tile + row) * ep.ldg + n_tile + c16 * 8u) = v;
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t nl = wn + j * 32 + col, m0 = wm + i * 32 + 8 * q + 4 * half;
          const bf16x4 v = {(__bf16)acc[i][j][4 * q], (__bf16)acc[i][j][4 * q + 1], (__bf16)acc[i][j][4 * q + 2], (__bf16)acc[i][j][4 * q + 3]};
          *reinterpret_cast<bf16x4*>(smemw + nl * GEMMW_TS + m0 * 2u) = v;
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const uint32_t pc = threadIdx.x + 512u * q, row = pc >> 5, c16 = pc & 31u;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smemw + row * GEMMW_TS + c16 * 16u);
      if (n_tile + row < N) *reinterpret_cast<bf16x8*>(ep.GT + (size_t)(n_tile + row) * ep.ldgt + m_tile + c16 * 8u) = v;
    }
  } else {
    if (m_base >= M) return;
    float* C = ep.Cout;
    if constexpr (EPI == EPI_STORE) C += (size_t)zt * ep.split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t n = n_base + j * 32 + col;
        if (n >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if constexpr (EPI == EPI_ATOMIC) {
            if (m < ep.rows_live) unsafeAtomicAdd(C + (size_t)m * ep.ldc + n, acc[i][j][r]);
          } else {
            C[(size_t)m * ep.ldc + n] = acc[i][j][r];
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM 1 of the K = 512 path with the z rows in registers:  G^T[item][user] = loss'(D[item] . z_user + b'[item], 0).
//
// The 256 x 256-tile kernel stages BOTH operands of every tile through LDS (512 KiB per tile, 128 flop per staged byte) and is bound
// by that fill (~32 GB/s per CU), and its loss epilogue runs with nothing else on the CU.  Here a 512-thread workgroup owns 256
// users for its whole life: wavefront w holds the K = 512 fragments of its 32 users in 128 registers (the MFMA A operand) and the
// workgroup walks ITEM tiles of 128 — only D is staged (16 KiB slices of 64 k; four stages, three slices in flight: 256 flop per
// staged byte), and because the stages are a ring of their own the next tile's slices keep arriving while the current tile's loss
// epilogue runs.  C[user][item] puts four consecutive users of one item into a lane, so bf16(g) goes to a [128 items][256 users]
// LDS image as 8-byte pieces and leaves as whole 512-byte rows of G^T.  Workgroups that share an item range (the user tiles) sit on
// ONE XCD, so D comes from HBM once.  Every element is the same sum over k in the same order as gemm_nt_bf16_ldsw_kernel<EPI_LOSS>
// (16-wide steps ascending): identical G^T (test_gemm1_zreg_changes_no_bit).
constexpr int G1Z_STAGE_BYTES = 128 * 64 * 2;                    // one D slice: 128 items x 64 k (128-byte rows, swizzled as GEMM_SLICE)
constexpr uint32_t G1Z_IMG_RS = 528;                             // epilogue image row: 256 users bf16 + 16 B
constexpr size_t gemm1_zreg_lds_bytes() { return 4 * (size_t)G1Z_STAGE_BYTES + 128 * (size_t)G1Z_IMG_RS; }

template <int LOSS>
__global__ void __launch_bounds__(512)
gemm1_loss_zreg_kernel(const __bf16* __restrict__ Zb /* [Bp][512] */, const __bf16* __restrict__ Db /* [Ip][512] */,
                       const float* __restrict__ bp, __bf16* __restrict__ GT, uint32_t ldgt, uint32_t rows_live, uint32_t cols_live,
                       uint32_t Ip, uint32_t user_tiles, uint32_t item_groups, uint32_t tiles_per_group) {
  extern __shared__ __attribute__((aligned(1024))) char smemz[];
  char* const img = smemz + 4 * G1Z_STAGE_BYTES;
  const uint32_t lane = threadIdx.x % WAVE, wid = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);      // 0..7
  uint32_t ut, ig;
  {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    ut = j % user_tiles; ig = (j / user_tiles) * 8u + xcd;
    if (ig >= item_groups) return;
  }
  const uint32_t n_tiles_all = Ip / 128u;
  const uint32_t t_begin = ig * tiles_per_group, t_end = min(n_tiles_all, t_begin + tiles_per_group);
  if (t_begin >= t_end) return;
  const uint32_t n_tiles = t_end - t_begin, n_slices = n_tiles * 8u;
  const uint32_t u_tile = ut * 256u;
  const uint32_t f_row = lane & 31u, f_half = lane >> 5;

  // this wavefront's 32 z rows: fragment kk = k in [16 kk, 16 kk + 16), lane holds the 8 of its half
  bf16x8 zf[32];
  {
    const __bf16* zr = Zb + (size_t)(u_tile + wid * 32u + f_row) * 512u + 8u * f_half;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) zf[kk] = *reinterpret_cast<const bf16x8*>(zr + 16 * kk);
  }
  // staging: slice sl = (tile, ks) -> 16 DMA instructions of 1 KiB (8 rows of 128 bytes); wavefront w issues 2 w, 2 w + 1
  const uint32_t st_row = lane >> 3, st_slot = lane & 7u;
  auto stage = [&](uint32_t sl) {
    const uint32_t tile = t_begin + (sl >> 3), ks = sl & 7u;
    char* base = smemz + (sl & 3u) * G1Z_STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = (wid * 2u + q) * 8u + st_row;
      const __bf16* src = Db + (size_t)min(tile * 128u + r, Ip - 1u) * 512u + ks * 64u + 8u * (st_slot ^ ((r >> 1) & 7u));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + (wid * 2u + q) * 1024u), 16, 0, 0);
    }
  };
  uint32_t d_off[4], d_sw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const uint32_t r = j * 32u + f_row; d_off[j] = r * 128u; d_sw[j] = (r >> 1) & 7u; }

  stage(0);
  if (n_slices > 1) stage(1);
  if (n_slices > 2) stage(2);
  f32x16 acc[4];
  uint32_t sl = 0;
  for (uint32_t t = 0; t < n_tiles; ++t) {
    const uint32_t item0 = (t_begin + t) * 128u;
    float bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t n = item0 + j * 32u + f_row; bias[j] = n < cols_live ? bp[n] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks, ++sl) {
      // this wavefront's two DMAs of slice sl have landed.  Issued behind them: two per later slice in flight (sl + 1, sl + 2), the
      // bias loads of this tile (ks == 0: compiler-visible, counted by its own waits) and — for the first three slices of every
      // tile but the first — the 8 G^T stores of the previous tile's epilogue
      const uint32_t later = min(2u, n_slices - 1u - sl);
      const bool stores_behind = t > 0 && ks < 3;
      if (later == 2u) { if (stores_behind) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else if (later == 1u) { if (stores_behind) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();                                        // everyone's; the stage slice sl - 1 occupied is free
      if (sl + 3u < n_slices) stage(sl + 3u);
      const char* base = smemz + (sl & 3u) * G1Z_STAGE_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint32_t c = 2u * s + f_half;
        bf16x8 fd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fd[j] = *reinterpret_cast<const bf16x8*>(base + d_off[j] + ((c ^ d_sw[j]) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[ks * 4 + s], fd[j], acc[j], 0, 0, 0);
      }
    }
    // loss epilogue: lane = item item0 + 32 j + f_row, acc[j][4 q + e] = user u_tile + 32 wid + 8 q + 4 f_half + e
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool n_live = item0 + j * 32u + f_row < cols_live;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t ul = wid * 32u + 8u * q + 4u * f_half;
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = acc[j][4 * q + e] + bias[j];
          const float v = LOSS == 0 ? 2.f * y : fast_rcp(1.f + fast_exp(-y));
          g[e] = (n_live && u_tile + ul + e < rows_live) ? v : 0.f;
        }
        const bf16x4 hb = {(__bf16)g[0], (__bf16)g[1], (__bf16)g[2], (__bf16)g[3]};
        *reinterpret_cast<bf16x4*>(img + (j * 32u + f_row) * G1Z_IMG_RS + ul * 2u) = hb;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t pc = threadIdx.x + 512u * q, row = pc >> 5, c16 = pc & 31u;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(img + row * G1Z_IMG_RS + c16 * 16u);
      *reinterpret_cast<bf16x8*>(GT + (size_t)(item0 + row) * ldgt + u_tile + c16 * 8u) = v;      // (nontemporal here: no difference, measured)
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM 1 of the K = 512 path, round 5: the two wavefronts of a SIMD in OPPOSITE phases (gemm1_loss_duo_kernel; the default).
//
// gemm1_loss_zreg_kernel runs its eight wavefronts in lockstep: all contract a tile, then all run its loss epilogue — 64 x (exp, rcp,
// pack) per lane — with the matrix cores idle (rocprofv3: MFMA busy 42 %; 1.15 ms per launch at 1 M items x 1024 users against 0.52 ms
// for the contraction and its staging alone).  Round 4 tried the epilogue between the MFMAs of one wavefront (no gain: both wavefronts
// of a SIMD then want the same issue slots at the same time).  Here the workgroup's two halves — wavefronts 0-3 and 4-7, one of each
// on every SIMD — alternate roles by PERIOD, one workgroup barrier per period:
//   period p, t = p / 2:   half (p & 1)      contracts item tile t (64 items, two 32 x 32 accumulators per wavefront: 64 MFMAs back to
//                                            back, nothing else but their LDS reads in its stream)
//                          the other half    runs the loss epilogue of the tile it contracted in the period before, stores its G^T
//                                            piece, and issues the LDS DMA of tile t + 1
// so at any time exactly one wavefront per SIMD feeds the matrix pipe (one stream of independent MFMAs saturates it) and its partner's
// exp / rcp / packs / stores issue in the gaps.  A tile (64 items x K = 512: 64 KiB, eight slices of 64 k, rows swizzled as GEMM_SLICE)
// is double-buffered WHOLE: tile t is live for periods 2t (first half reads it) and 2t + 1 (second half), tile t + 1 lands meanwhile
// in the other buffer — no slice-level hand-shake, no image buffer: a lane holds four consecutive users of an item; v_permlane32_swap
// pairs it with the lane that holds the next four, and G^T leaves in 16-byte pieces (64 contiguous bytes per item row and wavefront).
// Every element is the same sum over k in the same order (16-wide steps ascending) and the same loss expression as the kernels above:
// G^T is bit-identical (test_gemm1_zreg_changes_no_bit).
__device__ __forceinline__ void wait_vmcnt_at_most(uint32_t n) {     // n is wave-uniform
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
    case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
    case 11: __builtin_amdgcn_s_waitcnt(0x0F7B); break;
    default: __builtin_amdgcn_s_waitcnt(0x0F7C); break;             // 12: the most this kernel ever has behind a slice
  }
}

constexpr int G1D_ITEMS = 64;
constexpr int G1D_TILE_BYTES = G1D_ITEMS * 512 * 2;              // 64 KiB: [8 slices][64 rows][128 B]
constexpr int G1D_PIECES = G1D_TILE_BYTES / 1024;                // 64 DMA instructions of 1 KiB (8 rows of one slice)
// developer builds (tools/build_variant.sh -DG1D_X_...=1: timing experiments, WRONG results): which part of a period costs what
#ifndef G1D_X_NOEPI
#define G1D_X_NOEPI 0
#endif
#ifndef G1D_X_NOSTORE
#define G1D_X_NOSTORE 0
#endif
#ifndef G1D_X_NODMA
#define G1D_X_NODMA 0
#endif
#ifndef G1D_X_NOMFMA
#define G1D_X_NOMFMA 0
#endif
constexpr int G1D_PIECES_EARLY = 10;                             // per wavefront: issued by the half whose epilogue period comes FIRST (a whole period to land) ...
constexpr int G1D_PIECES_LATE = G1D_PIECES / 4 - G1D_PIECES_EARLY;   // ... and by the half whose epilogue period ends at the barrier the tile is needed behind
constexpr size_t gemm1_duo_lds_bytes() { return 2 * (size_t)G1D_TILE_BYTES + 4 * G1D_ITEMS * sizeof(float); }     // two tiles + four tiles' b' values (tile t + 2's
                                                                                                                  // arrive while the second half still reads tile t's)

__device__ __forceinline__ uint32_t lds_addr_of(const void* p);
// step I of a tile's contraction reads rows f_row and 32 + f_row (offset 4096) of slice I / 4 (8 KiB each), 16-byte chunk 2 (I % 4) + f_half
template <int I>
__device__ __forceinline__ void g1d_read(bf16x8 (&f)[2], const uint32_t (&va)[4]) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[0]) : "v"(va[I & 3]), "n"((I >> 2) * 8192));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[1]) : "v"(va[I & 3]), "n"((I >> 2) * 8192 + 4096));
}
template <int N>
__device__ __forceinline__ void g1d_wait(bf16x8 (&f)[2]) {       // the MFMAs that take f are ordered behind the wait through its operands
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N));
}
template <int I>
__device__ __forceinline__ void g1d_steps(f32x16 (&acc)[2], const bf16x8 (&zf)[32], bf16x8 (&fd)[3][2], const uint32_t (&va)[4]) {
  if constexpr (I < 32) {
    if constexpr (I + 2 < 32) g1d_read<I + 2>(fd[(I + 2) % 3], va);
    g1d_wait<(I + 2 < 32) ? 4 : ((I + 1 < 32) ? 2 : 0)>(fd[I % 3]);     // LDS reads return in order: what was requested behind step I's pair may stay out
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[I], fd[I % 3][j], acc[j], 0, 0, 0);
    g1d_steps<I + 1>(acc, zf, fd, va);
  }
}

template <int LOSS>
__global__ void __launch_bounds__(512)
gemm1_loss_duo_kernel(const __bf16* __restrict__ Zb /* [Bp][512] */, const __bf16* __restrict__ Db /* [Ip][512] */,
                      const float* __restrict__ bp, __bf16* __restrict__ GT, uint32_t ldgt, uint32_t rows_live, uint32_t cols_live,
                      uint32_t Ip, uint32_t user_tiles, uint32_t item_groups, uint32_t tiles_per_group /* of 128 items */) {
  extern __shared__ __attribute__((aligned(1024))) char smemd[];
  const uint32_t lane = threadIdx.x % WAVE, wid = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);      // 0..7
  const uint32_t half = wid >> 2, hw = wid & 3u;                 // wavefronts w and w + 4 share a SIMD
  uint32_t ut, ig;
  {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    ut = j % user_tiles; ig = (j / user_tiles) * 8u + xcd;
    if (ig >= item_groups) return;
  }
  const uint32_t n_tiles_all = Ip / (uint32_t)G1D_ITEMS;
  const uint32_t t_begin = ig * tiles_per_group * 2u, t_end = min(n_tiles_all, t_begin + tiles_per_group * 2u);
  if (t_begin >= t_end) return;
  const uint32_t n_tiles = t_end - t_begin;
  const uint32_t u_tile = ut * 256u;
  const uint32_t f_row = lane & 31u, f_half = lane >> 5;

  // this wavefront's 32 z rows: fragment kk = k in [16 kk, 16 kk + 16), lane holds the 8 of its half
  bf16x8 zf[32];
  {
    const __bf16* zr = Zb + (size_t)(u_tile + wid * 32u + f_row) * 512u + 8u * f_half;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) zf[kk] = *reinterpret_cast<const bf16x8*>(zr + 16 * kk);
  }
  // DMA piece c of a tile: slice c / 8, rows 8 (c % 8) .. + 8 (1 KiB); lane -> (row of the piece, 16-byte slot of its 128-byte line).
  // Source address = a wave-uniform base (tile, piece) + a 32-bit lane offset that depends on the piece's parity only (the swizzle
  // term (r >> 1) & 7 of row r = 8 (c % 8) + st_row is 4 (c & 1) + st_row / 2): one scalar add and the DMA per piece.  Tiles are whole
  // (Ip is a multiple of 256 on this path): no row clamp.
  const uint32_t st_row = lane >> 3, st_slot = lane & 7u;
  uint32_t lane_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) lane_off[par] = st_row * 1024u + 16u * (st_slot ^ ((4u * par + (st_row >> 1)) & 7u));
  auto stage_piece = [&](uint32_t tile /* workgroup-local */, uint32_t c /* wave-uniform */) {
    const char* gbase = reinterpret_cast<const char*>(Db) + (size_t)(t_begin + tile) * (G1D_ITEMS * 1024u) + (c & 7u) * 8192u + (c >> 3) * 128u;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + lane_off[c & 1u]),
                                     (__attribute__((address_space(3))) void*)(smemd + (tile & 1u) * G1D_TILE_BYTES + c * 1024u), 16, 0, 0);
  };
  uint32_t d_off[2], d_sw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const uint32_t r = j * 32u + f_row; d_off[j] = r * 128u; d_sw[j] = (r >> 1) & 7u; }

  // b' of a tile's 64 items: one 256-byte DMA (4 bytes per lane) beside the tile's pieces
  auto stage_bias = [&](uint32_t tile) {
    const float* src = bp + min((t_begin + tile) * (uint32_t)G1D_ITEMS + lane, cols_live - 1u);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smemd + 2 * G1D_TILE_BYTES + (tile & 3u) * (G1D_ITEMS * 4)), 4, 0, 0);
  };
  // tile 0: everybody stages an eighth of it
#pragma unroll
  for (int q = 0; q < G1D_PIECES / 8; ++q) stage_piece(0u, wid * (G1D_PIECES / 8) + q);
  if (wid == 0u) stage_bias(0u);
  __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0), as the builtin: the compiler's wait-count pass then knows the z loads above are complete
  __builtin_amdgcn_s_barrier();

  f32x16 acc[2];
  const uint32_t n_periods = 2u * n_tiles + 1u;
  for (uint32_t p = 0; p < n_periods; ++p) {
    const uint32_t t = p >> 1;
    if ((p & 1u) == half) {
      // ---- contraction of tile t ----
      if (t < n_tiles) {
        const char* base = smemd + (t & 1u) * G1D_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // 32 steps of 16 k; the fragments of step i + 2 are requested before the MFMAs of step i — two steps = 128 matrix-pipe cycles
        // cover the LDS round trip (g1d_steps: asm reads and counted lgkmcnt waits; from the plain loop nest hipcc issued read pair ->
        // lgkmcnt(0) -> two MFMAs, and with sched_barriers it still drained the queue every third step)
        uint32_t va[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) va[s4] = lds_addr_of(base) + d_off[0] + (((2u * s4 + f_half) ^ d_sw[0]) << 4);
        bf16x8 fd[3][2];
        g1d_read<0>(fd[0], va);
        g1d_read<1>(fd[1], va);
        if (!G1D_X_NOMFMA) g1d_steps<0>(acc, zf, fd, va);
      }
      // this half's DMA pieces of the period before (the EARLY ones are needed behind this barrier) and its stores are a period old by
      // now.  (No vector-memory LOAD into registers anywhere in the loop: the compiler's wait for one — vmcnt(0) — would also wait
      // for the stores this wavefront issued just before the last barrier; b' comes through LDS with the tile.)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      // ---- epilogue of the tile this half contracted one period ago, and this wavefront's eighth of tile t + 1 ----
      // p even: this is the second half and its tile is t - 1; p odd: the first half, tile t.  Tile t + 1's buffer was last read in
      // period 2t - 1 (second half's contraction of t - 1): free in both.
      const bool early = (p & 1u) == 0u;
      const uint32_t et = early ? t - 1u : t;                    // (p == 0: no tile yet, wraps to 0xFFFFFFFF)
      // tile t + 1's pieces: p even — the second half, a whole period before the tile is needed — issues G1D_PIECES_EARLY per wavefront,
      // p odd — the first half — the rest, FIRST: they are needed behind this period's barrier
      uint32_t n_dma = 0;
      if (t + 1u < n_tiles && !G1D_X_NODMA) {
        if (early) {
#pragma unroll
          for (int q = 0; q < G1D_PIECES_EARLY; ++q) stage_piece(t + 1u, hw * G1D_PIECES_EARLY + q);
          if (hw == 0u) stage_bias(t + 1u);
          n_dma = G1D_PIECES_EARLY;
        } else {
#pragma unroll
          for (int q = 0; q < G1D_PIECES_LATE; ++q) stage_piece(t + 1u, 4u * G1D_PIECES_EARLY + hw * G1D_PIECES_LATE + q);
          n_dma = G1D_PIECES_LATE;
        }
      }
      const bool epi = et < n_tiles && !G1D_X_NOEPI;
      const uint32_t item0 = (t_begin + (epi ? et : 0u)) * (uint32_t)G1D_ITEMS;
      // lane = item item0 + 32 j + f_row, acc[j][4 q + e] = user u_tile + 32 wid + 8 q + 4 f_half + e.  A tile with all of its items and
      // all of this wavefront's users live (every tile but the ragged edges) skips the masks: the epilogue's instruction count is
      // what paces a period (measured: epilogue + stores alone 2670 cycles against the contraction's 2048)
      const bool interior = item0 + (uint32_t)G1D_ITEMS <= cols_live && u_tile + wid * 32u + 32u <= rows_live;      // wave-uniform
      uint4 outs[2][2];                                          // [j][piece]: bf16(g) of item row item0 + 32 j + f_row, this lane's two 16-byte pieces
      auto pieces = [&](int j, auto masked) {
        uint4& out0 = outs[j][0];
        uint4& out1 = outs[j][1];
        const float bias = *reinterpret_cast<const float*>(smemd + 2 * G1D_TILE_BYTES + (et & 3u) * (G1D_ITEMS * 4) + (j * 32u + f_row) * 4u);
        const bool n_live = item0 + j * 32u + f_row < cols_live;
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t ul = wid * 32u + 8u * q + 4u * f_half;
          float g[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = acc[j][4 * q + e] + bias;
            const float v = LOSS == 0 ? 2.f * y : fast_rcp(1.f + fast_exp(-y));
            g[e] = (!decltype(masked)::value || (n_live && u_tile + ul + e < rows_live)) ? v : 0.f;
          }
          const bf16x4 hb = {(__bf16)g[0], (__bf16)g[1], (__bf16)g[2], (__bf16)g[3]};
          pk[q] = __builtin_bit_cast(uint2, hb);
        }
        // v_permlane32_swap(a, b): a's upper 32 lanes <-> b's lower 32 lanes.  Lane (f_row, 0) holds users 8 q + 0..3, lane
        // (f_row, 1) users 8 q + 4..7 of item f_row: after swapping (q0, q1) the lower lane holds all eight of q0, the upper of q1
        {
          const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
          out0 = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
        {
          const auto sx = __builtin_amdgcn_permlane32_swap(pk[2].x, pk[3].x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pk[2].y, pk[3].y, false, false);
          out1 = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      };
      auto flush = [&](int j) {                                  // this wavefront's 32 users of the item's row: 64 bytes, 32 of them from this lane pair
        __bf16* grow = GT + (size_t)(item0 + j * 32u + f_row) * ldgt + u_tile + wid * 32u;
        if (!G1D_X_NOSTORE || outs[j][0].x == 0x12345u) {
          *reinterpret_cast<uint4*>(grow + 8u * f_half) = outs[j][0];
          *reinterpret_cast<uint4*>(grow + 8u * (2u + f_half)) = outs[j][1];
        }
      };
      if (epi) {
        if (interior) { pieces(0, std::false_type{}); pieces(1, std::false_type{}); }
        else { pieces(0, std::true_type{}); pieces(1, std::true_type{}); }
      }
      // LATE pieces must have landed behind this barrier: they are waited for EXACTLY — vmcnt(0) while they are the only vector-memory
      // operations this wavefront has outstanding (its previous stores are two periods old) — and the G^T stores go out behind the
      // wait, so nothing here depends on loads and stores completing in order with each other.  EARLY pieces are waited for by the
      // vmcnt(0) at the end of this wavefront's next contraction.  Measured at 1 M items x 1024 users
      // (profiles/r05_gemm1_duo_anatomy.txt): 0.98-1.04 ms per launch with 10 + 6 pieces per wavefront; 1.06 with 16 + 0 and no wait at
      // all — a wavefront whose DMA finds the CU's queue full stalls at the ISSUE, so the epilogue waits for the fill either way;
      // 1.31 with the tile staged through registers; 0.57 without staging, 0.42 without the epilogue (= the matrix-core floor), 0.78
      // without the G^T stores.
      if (!early && n_dma != 0u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (epi) { flush(0); flush(1); }
      __builtin_amdgcn_s_barrier();
    }
  }
}


// ------------------------------------------------------------------------------------------------
// "TN" product for GEMM 2 of the K > 256 path:  C[m][n] = sum_c A[c][m] * Bm[c][n]  with BOTH operands stored contraction-row-major
// — hg = G D as  sum_item G^T[item][user] * D[item][k]  straight from the two images the other launches already keep (G^T for
// GEMM 3 / the row step, the row-major bf16 decoder image for GEMM 1).  With it GEMM 1 no longer writes G (2 GB per 1024-user
// block at 1 M items, and one of its two LDS passes), and D^T — its 1 GB, and the 0.44 ms transposition per block that kept it
// current — is not needed by the K > 256 path at all.
// An MFMA fragment wants 8 consecutive contraction elements of one m per lane, i.e. a COLUMN of the staged slice: gfx950's
// ds_read_b64_tr_b16 reads, per 16-lane group, a [4 c][16 m] block — lane i supplies 4 contiguous m of row i / 4 — and hands lane
// j the four c of column j (measured, tools/tr_b16_semantics.hip); two of them are one fragment.  LDS image of a slice (64
// contraction rows): row c = [256 m of A | 256 n of Bm | 64 bytes], ONE 1 KiB DMA instruction per row (lanes 0-31 fetch the A
// piece, 32-63 the Bm piece), and the 1088-byte stride puts the four rows of a transposing read 64 bytes apart in the bank
// row: conflict-free.  Same 256 x 256 tile, wavefront layout, two 68 KiB stages and epilogue as gemm_nt_bf16_ldsw_kernel; every
// output element is the same sum in the same order (16-wide steps ascending, same contraction splits), so the slabs are
// bit-identical to the NT kernel's (test_tn_gemm2_changes_no_bit).
// Transposing LDS reads as inline asm (round 4).  Through the builtin (__builtin_amdgcn_ds_read_tr16_b64_*) the compiler's wait-count pass
// sees an LDS read with no alias information behind an LDS DMA (global_load_lds) and puts `s_waitcnt vmcnt(0)` in front of the first
// read after every stage() — i.e. the NEXT slice's fill was waited for before the current slice was contracted, fills and matrix work
// never overlapped, and a 64-row step lasted fill round trip + contraction (2.25 us for 0.85 us of MFMAs: the 0.37 of round 3).  As asm
// the reads are invisible to that pass; their own completion is waited for by hand (lgkm_wait: the fragments are in/out operands of
// the s_waitcnt, so every MFMA that consumes them is ordered behind it).
template <int OFF>
__device__ __forceinline__ bf16x4 lds_read_tr16(uint32_t addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_frag_tr16(uint32_t addr) {      // rows r..r+3 and r+4..r+7 of one 16-row sub-step
  const bf16x4 lo = lds_read_tr16<OFF>(addr);
  const bf16x4 hi = lds_read_tr16<OFF + 4 * 1088>(addr);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int N>
__device__ __forceinline__ void lgkm_wait(bf16x8 (&a)[2], bf16x8 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

constexpr uint32_t GTN_RS = 1088;
// Staging depth and wavefront shape (round 4, measured after the transposing reads became asm — see lds_read_tr16): two 64-row
// stages 0.94 ms per launch at 1 M items x 1024 users; the same 136 KiB as 32-row stages with two / three of them in flight behind the
// one being contracted (ROWS = 32, NST = 3 / 4, CDAE_GEMM2_STAGES; counted vmcnt) 0.97 / 0.97; four wavefronts of 128 x 128 outputs (a
// third fewer LDS reads, built and removed) 1.00.  With parts compiled out (GTN_X_*): no MFMAs 0.83, no fills 0.70, no LDS reads
// beyond the first sub-step 0.84 — the fills (8.6 GB from the L2s per launch: each G^T piece is staged by two workgroups, each decoder
// piece by four) and the contraction each take most of the launch and overlap only partly.  All variants bit-identical.
template <int ROWS, int NST>
constexpr size_t gemm_tn_lds_bytes() { return (size_t)NST * ROWS * GTN_RS; }

template <int ROWS /* contraction rows per stage: 64 or 32 */, int NST /* stages */>
__global__ void __launch_bounds__(512)
gemm_tn_bf16_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ Bm, uint32_t M, uint32_t N, uint32_t Kd,
                    uint32_t lda, uint32_t ldb, uint32_t k_per_split, GemmEpilogue ep, GemmGrid gg) {
  static_assert(ROWS % 16 == 0 && ROWS % 8 == 0 && NST >= 2 && NST <= 4, "stage shape");
  constexpr int STAGE_BYTES = ROWS * (int)GTN_RS;
  constexpr int PER_WAVE = ROWS / 8;                                       // DMA instructions per wavefront and stage
  extern __shared__ __attribute__((aligned(1024))) char smemt[];
  const uint32_t lane = threadIdx.x % WAVE, wid = threadIdx.x / WAVE;      // wid 0..7
  uint32_t mt, nt, zt;
  {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, in = gg.inner();
    const uint32_t t = j % in, o = (j / in) * 8u + xcd;
    if (o >= gg.outer()) return;
    if (gg.mode == 0) { mt = t; nt = o; zt = 0; }
    else if (gg.mode == 1) { nt = t; mt = o; zt = 0; }
    else { mt = t / gg.Nt; nt = t % gg.Nt; zt = o; }
  }
  const uint32_t m_tile = mt * 256u, n_tile = nt * 256u;
  const uint32_t wm = (wid >> 1) * 64u, wn = (wid & 1u) * 128u;
  const uint32_t m_base = m_tile + wm, n_base = n_tile + wn;
  const uint32_t k_begin = zt * k_per_split;
  const uint32_t k_end = min(Kd, k_begin + k_per_split);
  const uint32_t n_steps = (k_end - k_begin) / (uint32_t)ROWS;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: contraction row r of the slice is one DMA instruction; wavefront w issues rows PER_WAVE w .. PER_WAVE w + PER_WAVE - 1
  const __bf16* src = lane < 32u ? A + m_tile + 8u * lane : Bm + n_tile + 8u * (lane - 32u);
  const uint32_t ld = lane < 32u ? lda : ldb;
  auto stage = [&](uint32_t step, uint32_t slot) {
    char* base = smemt + slot * STAGE_BYTES;
    const uint32_t k = k_begin + step * (uint32_t)ROWS + wid * (uint32_t)PER_WAVE;
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(k + q) * ld),
                                       (__attribute__((address_space(3))) void*)(base + (wid * (uint32_t)PER_WAVE + q) * GTN_RS), 16, 0, 0);
  };
  // fragment addresses: lane L of a 16-lane group supplies row 8 (L >> 5) + ((L & 15) >> 2) (+ 4 for the second read), columns
  // 16 ((L >> 4) & 1) + 4 (L & 3) .. + 3 of the fragment's 32; lane L receives column L & 31, four rows per read
  const uint32_t f_off = (8u * (lane >> 5) + ((lane & 15u) >> 2)) * GTN_RS + (16u * ((lane >> 4) & 1u) + 4u * (lane & 3u)) * 2u;
  const uint32_t lds0 = lds_addr_of(smemt);
  const uint32_t a_off = lds0 + f_off + wm * 2u, b_off = lds0 + f_off + 512u + wn * 2u;
#define GTN_READ(FA_, FB_, S_)                                                          \
  do {                                                                                  \
    if (GTN_X_NOREAD && (S_) != 0) break;                                               \
    FA_[0] = lds_frag_tr16<(S_) * 16 * (int)GTN_RS>(a_cur);                             \
    FA_[1] = lds_frag_tr16<(S_) * 16 * (int)GTN_RS + 64>(a_cur);                        \
    FB_[0] = lds_frag_tr16<(S_) * 16 * (int)GTN_RS>(b_cur);                             \
    FB_[1] = lds_frag_tr16<(S_) * 16 * (int)GTN_RS + 64>(b_cur);                        \
    FB_[2] = lds_frag_tr16<(S_) * 16 * (int)GTN_RS + 128>(b_cur);                       \
    FB_[3] = lds_frag_tr16<(S_) * 16 * (int)GTN_RS + 192>(b_cur);                       \
  } while (0)
#ifndef GTN_X_NOMFMA      // developer switches (tools/build_variant.sh): parts of the launch compiled out, to see what bounds it
#define GTN_X_NOMFMA 0
#endif
#ifndef GTN_X_NODMA
#define GTN_X_NODMA 0
#endif
#ifndef GTN_X_NOREAD
#define GTN_X_NOREAD 0
#endif
#define GTN_MFMA(FA_, FB_)                                                                                        \
  do {                                                                                                            \
    if (GTN_X_NOMFMA) { acc[0][0][0] += (float)FA_[0][0] + (float)FA_[1][0] + (float)FB_[0][0] + (float)FB_[1][0] + (float)FB_[2][0] + (float)FB_[3][0]; break; } \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_[i], FB_[j], acc[i][j], 0, 0, 0);                  \
  } while (0)

#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if ((uint32_t)p < n_steps) stage((uint32_t)p, (uint32_t)p);
  uint32_t slot = 0;
  for (uint32_t step = 0; step < n_steps; ++step) {
    // this wavefront's DMAs of slice `step` have landed: the stages behind it (at most NST - 2 of them) stay in flight
    const uint32_t behind = min((uint32_t)(NST - 2), n_steps - 1u - step);
    wait_vmcnt_at_most(behind * (uint32_t)PER_WAVE);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                                          // ... and everyone else's; the stage of slice step-1 is free
    if (step + (uint32_t)(NST - 1) < n_steps && !(GTN_X_NODMA && step > 2u)) stage(step + (uint32_t)(NST - 1), (slot + (uint32_t)(NST - 1)) % (uint32_t)NST);
    // fragments of sub-step s + 1 are requested in front of sub-step s's MFMAs, behind the wait for sub-step s's own (never more
    // than 12 LDS reads outstanding: lgkmcnt counts to 15)
    const uint32_t a_cur = a_off + slot * (uint32_t)STAGE_BYTES, b_cur = b_off + slot * (uint32_t)STAGE_BYTES;
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    GTN_READ(fa0, fb0, 0);
    lgkm_wait<0>(fa0, fb0);
    GTN_READ(fa1, fb1, 1);
    __builtin_amdgcn_sched_barrier(0);
    GTN_MFMA(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ROWS == 64) {
      lgkm_wait<0>(fa1, fb1);
      GTN_READ(fa0, fb0, 2);
      __builtin_amdgcn_sched_barrier(0);
      GTN_MFMA(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      lgkm_wait<0>(fa0, fb0);
      GTN_READ(fa1, fb1, 3);
      __builtin_amdgcn_sched_barrier(0);
      GTN_MFMA(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
    }
    lgkm_wait<0>(fa1, fb1);                                                // (all of this stage's reads are done before the next barrier)
    GTN_MFMA(fa1, fb1);
    slot = slot + 1u == (uint32_t)NST ? 0u : slot + 1u;
  }
#undef GTN_READ
#undef GTN_MFMA
  if (m_base >= M) return;
  const uint32_t half = lane >> 5, col = lane & 31u;
  float* C = ep.Cout + (size_t)zt * ep.split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t n = n_base + j * 32 + col;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        C[(size_t)m * ep.ldc + n] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused forward + loss' + hidden gradient (replaces GEMM 1, the positive fix-up and GEMM 2 when Kp <= 256):
// a workgroup owns 128 users (wavefront w: 32 of them) and one slice of the item dimension, and walks the slice in tiles
// of 32 items.  Per tile and wavefront:
//   C1[item][user] = D_tile Z^T                      Kp/16 MFMAs; A = D tile from LDS (row-major, ds_read_b128),
//                                                    B = the wave's z rows, held in registers for the whole launch
//   g = loss'(C1 + b'[item], target)                 target = bit of the user's rated-items word of this tile
//                                                    (rated_bits_kernel), so positives need no second pass
//   G^T[item][user] = bf16(g)                        the only image of G that reaches memory (GEMM 3 and the row step
//                                                    read it); lanes of one item hold 32 consecutive users
//   hg[user][k] += sum_items g D[item][k]            2 * Kp/32 MFMAs; A = the 8 g values a lane already holds per
//                                                    16-item step (C1 puts 4 consecutive items of ONE user in a lane,
//                                                    and the contraction order is free as long as A and B agree),
//                                                    B = D^T tile from LDS (two 8-byte reads)
// so G never takes the [users x items] detour through HBM on its way into the second product.  The per-slice partial
// hg is stored to HGpart[slice][user] and summed in fixed order by hidden_finish_kernel.
constexpr int FUSED_TILE = 32;                       // items per MFMA tile
constexpr int FUSED_SUB = 2;                         // tiles staged through LDS per step (64 items: loads get two tiles of work to land)
constexpr size_t full_fused_lds_bytes(uint32_t Kp) {
  return 2 * ((size_t)FUSED_SUB * FUSED_TILE * (Kp + 8) + (size_t)Kp * (FUSED_SUB * FUSED_TILE + 4)) * sizeof(__bf16) +
         2 * ((size_t)FUSED_SUB * FUSED_TILE * sizeof(float) + (size_t)FUSED_SUB * 128 * sizeof(uint32_t));   // + b' and target words
}

template <int NKS /* Kp / 16 */, int LOSS>
__global__ void __launch_bounds__(256)
full_decode_fused_kernel(HyperParams hp, const __bf16* __restrict__ Zb /* [Bp x Kp] */, const __bf16* __restrict__ Db /* [Ip x Kp] */,
                         const __bf16* __restrict__ DTb /* [Kp x Ip] */, uint32_t Ip, const float* __restrict__ bp,
                         const uint32_t* __restrict__ bits /* [nb x words] */, uint32_t words, uint32_t nb,
                         uint32_t steps_per_slice, __bf16* __restrict__ GT /* [Ip x ldgt] */, uint32_t ldgt,
                         float* __restrict__ HGpart /* [slices][nb][Kp] */) {
  constexpr int Kp = 16 * NKS, NT = Kp / 32;
  constexpr int STEP = FUSED_SUB * FUSED_TILE;       // items per staged step
  constexpr int DROW = Kp + 8;                       // D tile row stride (bf16): 16-byte reads of 16 rows hit distinct banks
  constexpr int TROW = STEP + 4;                     // D^T tile row stride: 8-byte reads of 32 rows hit distinct banks
  extern __shared__ __attribute__((aligned(16))) char fused_smem[];
  __bf16* dt = reinterpret_cast<__bf16*>(fused_smem);                          // [2][STEP][DROW]
  __bf16* dtt = dt + 2 * STEP * DROW;                                          // [2][Kp][TROW]
  // b' of the step's items and the users' rated-items words also come through LDS: a global load inside the epilogue makes
  // the compiler drain vmcnt to 0 (loads and stores share the counter and complete out of order), i.e. wait for the G^T
  // stores just issued — eight HBM write round trips per step
  float* bpt = reinterpret_cast<float*>(dtt + 2 * Kp * TROW);                  // [2][STEP]
  uint32_t* wt = reinterpret_cast<uint32_t*>(bpt + 2 * STEP);                   // [2][FUSED_SUB][128]
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  const uint32_t col = lane & 31u, half = lane >> 5;
  const uint32_t user = blockIdx.y * 128u + wave * 32u + col;                  // batch slot
  const uint32_t n_steps = Ip / STEP;                                          // Ip is a multiple of 128
  const uint32_t s_begin = blockIdx.x * steps_per_slice, s_end = min(n_steps, s_begin + steps_per_slice);
  if (s_begin >= s_end) return;

  bf16x8 zf[NKS];                                    // B operand of product 1: Z[user][16 s + 8 half .. + 7]
#pragma unroll
  for (int s = 0; s < NKS; ++s) zf[s] = *reinterpret_cast<const bf16x8*>(Zb + (size_t)user * Kp + 16 * s + 8 * half);

  f32x16 hg[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) hg[nt][r] = 0.f;

  // staging: D rows = STEP rows x Kp bf16 (Kp / 8 sixteen-byte pieces per row), D^T = Kp rows x STEP items (STEP / 8 pieces)
  constexpr int D_PIECES = STEP * Kp / 8, T_PIECES = Kp * STEP / 8, T_PER_ROW = STEP / 8;
  constexpr int D_PER = (D_PIECES + 255) / 256, T_PER = (T_PIECES + 255) / 256;
  static_assert(D_PER == T_PER, "one staging register set serves both halves");
  bf16x8 sd[D_PER];
  float sb = 0.f;
  uint32_t sw = 0u;
  // per-thread piece addresses are fixed up to the step's offset: computed once, advanced by a constant per step
  uint32_t d_src[D_PER], d_dst[D_PER], t_src[T_PER], t_dst[T_PER];                 // element offsets (bf16)
#pragma unroll
  for (int q = 0; q < D_PER; ++q) {
    const uint32_t f = threadIdx.x + 256u * q, r = f / (Kp / 8), c = f % (Kp / 8);
    d_src[q] = r * Kp + 8 * c; d_dst[q] = r * DROW + 8 * c;
  }
#pragma unroll
  for (int q = 0; q < T_PER; ++q) {
    const uint32_t f = threadIdx.x + 256u * q, r = f / T_PER_ROW, c = f % T_PER_ROW;
    t_src[q] = r * Ip + 8 * c; t_dst[q] = r * TROW + 8 * c;
  }
  const uint32_t w_user = blockIdx.y * 128u + (threadIdx.x & 127u), w_sub = threadIdx.x >> 7;   // 256 threads = FUSED_SUB x 128 users
  // The next step's operands are fetched into registers in two halves — the D rows (+ b', target words) while the first tile
  // of the current step computes, the D^T rows while the second one does — so that only 32 staging registers are live at a
  // time (all 64 at once, beside zf, hg and the fragments, spilled to scratch at Kp = 256).  Both halves are committed to the
  // other LDS buffer inside the same step: every wavefront left that buffer at the barrier that ended the previous step.
  auto fetch_d = [&](uint32_t st) {
    const uint32_t i0 = st * STEP;
    const __bf16* dsrc = Db + (size_t)i0 * Kp;
    if (threadIdx.x < (uint32_t)STEP) sb = bp[i0 + threadIdx.x];               // in bounds up to Ip: b'_ag, b, b_ag follow b'
    {
      const uint32_t t = st * FUSED_SUB + w_sub;
      sw = (w_user < nb && t < words) ? bits[(size_t)w_user * words + t] : 0u;
    }
#pragma unroll
    for (int q = 0; q < D_PER; ++q)
      if (threadIdx.x + 256u * q < (uint32_t)D_PIECES) sd[q] = *reinterpret_cast<const bf16x8*>(dsrc + d_src[q]);
  };
  auto fetch_t = [&](uint32_t st) {
    const __bf16* tsrc = DTb + st * STEP;
#pragma unroll
    for (int q = 0; q < T_PER; ++q)
      if (threadIdx.x + 256u * q < (uint32_t)T_PIECES) sd[q] = *reinterpret_cast<const bf16x8*>(tsrc + t_src[q]);
  };
  auto commit_d = [&](int buf) {
    if (threadIdx.x < (uint32_t)STEP) bpt[buf * STEP + threadIdx.x] = sb;
    wt[buf * FUSED_SUB * 128 + threadIdx.x] = sw;
    __bf16* ddst = dt + (size_t)buf * STEP * DROW;
#pragma unroll
    for (int q = 0; q < D_PER; ++q)
      if (threadIdx.x + 256u * q < (uint32_t)D_PIECES) *reinterpret_cast<bf16x8*>(ddst + d_dst[q]) = sd[q];
  };
  auto commit_t = [&](int buf) {
    __bf16* tdst = dtt + (size_t)buf * Kp * TROW;
#pragma unroll
    for (int q = 0; q < T_PER; ++q)
      if (threadIdx.x + 256u * q < (uint32_t)T_PIECES) {
        // rows are 8 (mod 16) bytes apart: two 8-byte stores
        const bf16x4 lo = {sd[q][0], sd[q][1], sd[q][2], sd[q][3]}, hi = {sd[q][4], sd[q][5], sd[q][6], sd[q][7]};
        *reinterpret_cast<bf16x4*>(tdst + t_dst[q]) = lo;
        *reinterpret_cast<bf16x4*>(tdst + t_dst[q] + 4) = hi;
      }
  };
  fetch_d(s_begin);
  commit_d(0);
  fetch_t(s_begin);
  commit_t(0);
  __syncthreads();

  for (uint32_t st = s_begin; st < s_end; ++st) {
    const int buf = (int)((st - s_begin) & 1u);
    const bool more = st + 1 < s_end;                                          // workgroup-uniform
    if (more) fetch_d(st + 1);
#pragma unroll
    for (int sub = 0; sub < FUSED_SUB; ++sub) {
      if (sub == FUSED_SUB - 1 && more) { commit_d(buf ^ 1); fetch_t(st + 1); }
      const uint32_t t = st * FUSED_SUB + sub;                                 // global 32-item tile index
      const uint32_t word = wt[(buf * FUSED_SUB + sub) * 128 + wave * 32u + col];          // the user's training items in the tile
      // product 1: C1[item][user]
      // (all fragments are read from LDS first, then the MFMAs issue back to back on two alternating accumulators: with one
      // wavefront per SIMD nothing else hides a dependent MFMA's latency or an LDS wait between two of them)
      f32x16 c1, c1b;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c1b[r] = 0.f; }
      const __bf16* arow = dt + (size_t)buf * STEP * DROW + (sub * FUSED_TILE + col) * DROW + 8 * half;
      // fragments in groups of <= 8 (at Kp = 256 all 16 at once, next to zf, hg and the staging registers, overflowed the 256
      // architectural VGPRs: 9 dwords of scratch spills inside this loop)
      constexpr int AG = NKS < 8 ? NKS : 8;
#pragma unroll
      for (int s0 = 0; s0 < NKS; s0 += AG) {
        bf16x8 af[AG];
#pragma unroll
        for (int s = 0; s < AG; ++s) af[s] = *reinterpret_cast<const bf16x8*>(arow + 16 * (s0 + s));
#pragma unroll
        for (int s = 0; s < AG; s += 2) {
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], zf[s0 + s], c1, 0, 0, 0);
          if (s + 1 < AG) c1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s + 1], zf[s0 + s + 1], c1b, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) c1[r] += c1b[r];
      // loss gradient of the lane's 16 (item, user) pairs: items 8 q + 4 half + e of the tile.  Straight-line: rows of padding
      // users (z = 0) and their G^T columns are never read back, so only items beyond num_items (last tile; their b' slot
      // holds other parameters) are forced to g = 0, and the store address is a 32-bit offset from a uniform base.
      bf16x8 ga[2];                                                            // A operands of product 2 (16-item steps)
      const bool tail_tile = (t + 1u) * FUSED_TILE > hp.num_items;             // wave-uniform
      const uint32_t gt_off = (t * FUSED_TILE + 4u * half) * ldgt + user;      // G^T is < 2^32 elements (checked by the host)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(bpt + buf * STEP + sub * FUSED_TILE + 8 * q + 4 * half);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tgt = (float)((word >> (8u * q + 4u * half + (uint32_t)e)) & 1u);
          float g = loss_grad(LOSS, c1[4 * q + e] + bb[e], tgt);
          if (tail_tile && t * FUSED_TILE + 8u * q + 4u * half + (uint32_t)e >= hp.num_items) g = 0.f;
          const __bf16 gb = (__bf16)g;
          ga[q >> 1][4 * (q & 1) + e] = gb;
          GT[gt_off + (8u * q + (uint32_t)e) * ldgt] = gb;
        }
      }
      // product 2: hg[user][k] += sum over the tile's items; step ks covers tile items {16 ks + 4 half + 0..3, 16 ks + 8 + 4 half + 0..3}
      bf16x8 bf[2][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const __bf16* brow = dtt + (size_t)buf * Kp * TROW + (size_t)(32 * nt + col) * TROW + sub * FUSED_TILE + 4 * half;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x4 b_lo = *reinterpret_cast<const bf16x4*>(brow + 16 * ks);
          const bf16x4 b_hi = *reinterpret_cast<const bf16x4*>(brow + 16 * ks + 8);
          bf[ks][nt] = bf16x8{b_lo[0], b_lo[1], b_lo[2], b_lo[3], b_hi[0], b_hi[1], b_hi[2], b_hi[3]};
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)                                           // consecutive MFMAs write different accumulators
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) hg[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[ks], bf[ks][nt], hg[nt], 0, 0, 0);
    }
    if (more) commit_t(buf ^ 1);
    __syncthreads();
  }
  // C layout: column n = lane & 31 (hidden index 32 nt + n), rows = users 8 (r / 4) + 4 half + (r % 4) of the wave's 32
  float* out = HGpart + ((size_t)blockIdx.x * nb) * Kp;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t u = blockIdx.y * 128u + wave * 32u + 8u * (r >> 2) + 4u * half + (r & 3);
      if (u < nb) out[(size_t)u * Kp + 32 * nt + col] = hg[nt][r];
    }
}

// Targets: GEMM 1 computed every g against target 0.  For a positive, loss'(y, 1) = loss'(y, 0) - c with
// c = 1 (cross-entropy: sigmoid(y) - t) or 2 (square: -2 (t - y)), so the batch's positives are patched in place.
__global__ void __launch_bounds__(256)
full_positive_fixup_kernel(const uint32_t* __restrict__ ex_item, const uint64_t* __restrict__ ex_val, uint32_t n_ex,
                           float c, __bf16* __restrict__ G, uint32_t ldg, __bf16* __restrict__ GT, uint32_t ldgt,
                           uint8_t* __restrict__ has_in = nullptr /* [I]: marks the items some user of the block kept as an input (gemm3_rows_fused_kernel) */) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ex) return;
  const uint32_t item = ex_item[e];
  const uint32_t slot = (uint32_t)ex_val[e] & SLOT_MASK;
  if (has_in && ((uint32_t)ex_val[e] & INPUT_BIT)) has_in[item] = 1;
  const float g = (float)GT[(size_t)item * ldgt + slot] - c;     // (G and G^T hold the same value)
  if (G) G[(size_t)slot * ldg + item] = (__bf16)g;
  GT[(size_t)item * ldgt + slot] = (__bf16)g;
}

// K4b as a launch of its own: the full-output path runs it on a second stream beside GEMM 3 (2048 users x 58 ns of
// strictly sequential chain would otherwise sit on the critical path).
__global__ void __launch_bounds__(256)
hidden_bias_kernel(HyperParams hp, uint32_t nb, const float* __restrict__ DELTA, float* __restrict__ b, float* __restrict__ b_ag) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (hp.adagrad) hidden_bias_role<true, true>(hp, k, nb, DELTA, b, b_ag);
  else hidden_bias_role<false, true>(hp, k, nb, DELTA, b, b_ag);
}

// bf16 images of a decoder row that has just been stepped, written by the kernel that holds it in registers (round 3: the separate
// fp32 -> bf16 conversion pass over the whole matrix was 27 us of a 196 us ML-10M-shape step): Db[item][k] — the lane's NI
// elements are contiguous — and DTb[k][item], one 2-byte store per element (10.6 K rows x 256 columns per block: the L2 merges them;
// NOT used for item spaces >= 32768, where a million rows x 512 scattered 2-byte stores would cost more than the pass they replace).
template <int NI>
__device__ __forceinline__ void store_row_bf16(const float (&w)[NI], uint32_t item, uint32_t lo, uint32_t Kp, uint32_t Ip,
                                               __bf16* __restrict__ Db, __bf16* __restrict__ DTb) {
  __bf16 hb[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) hb[i] = (__bf16)w[i];
  __bf16* drow = Db + (size_t)item * Kp + lo;
#pragma unroll
  for (int i = 0; i < NI; ++i) drow[i] = hb[i];
#pragma unroll
  for (int i = 0; i < NI; ++i) DTb[(size_t)(lo + i) * Ip + item] = hb[i];
}

template <int NI>
__device__ __forceinline__ void store_row_only_bf16(const float (&w)[NI], uint32_t item, uint32_t lo, uint32_t Kp, __bf16* __restrict__ Db) {
  __bf16* drow = Db + (size_t)item * Kp + lo;
#pragma unroll
  for (int i = 0; i < NI; ++i) drow[i] = (__bf16)w[i];
}

// Row steps of the full-output schedule (+ the hidden-bias recurrence as the leading workgroups, like K5):
//   b'[j]: grad = sum_u G[u][j] + lambda b'[j]                                 cdae.hpp:230-237, summed over the block
//   tied : W[j]: grad = dD[j] + scale * sum_{u: j kept} delta_u + lambda W[j]     cdae.hpp:252-257 + 337-348 merged
//   asym : V[j]: grad = dD[j] + lambda V[j];  W[j] (only if some user kept j): scale * sum delta_u + lambda W[j]
// One workgroup per item row; the kept inputs of the row come from the item-sorted positives list.
template <int NI>
__global__ void __launch_bounds__(256)
full_rows_kernel(HyperParams hp, const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end,
                 const uint64_t* __restrict__ sorted_val, const float* __restrict__ DELTA,
                 const float* __restrict__ dD, const __bf16* __restrict__ GT, uint32_t ldgt, uint32_t nb,
                 float* __restrict__ W, float* __restrict__ W_ag, float* __restrict__ V, float* __restrict__ V_ag,
                 float* __restrict__ bp, float* __restrict__ bp_ag, float* __restrict__ b, float* __restrict__ b_ag,
                 uint32_t* __restrict__ touched,
                 __bf16* __restrict__ Db = nullptr /* [Ip][Kp] */, __bf16* __restrict__ DTb = nullptr /* [Kp][Ip] */, uint32_t Ip = 0,
                 uint32_t bias_u0 = 0 /* the bias role starts at this user of the block (the users before it were taken beside GEMM 3) */,
                 const float* __restrict__ BIAS_DELTA = nullptr /* the plain delta rows the recurrence reads (default: DELTA) */) {
  const uint32_t bias_blocks = b ? (hp.Kp + blockDim.x - 1) / blockDim.x : 0u;   // b == nullptr: the recurrence runs in hidden_bias_kernel
  if (blockIdx.x < bias_blocks) {
    const float* dl = (BIAS_DELTA ? BIAS_DELTA : DELTA) + (size_t)bias_u0 * hp.Kp;
    if (hp.adagrad) hidden_bias_role<true, true>(hp, blockIdx.x * blockDim.x + threadIdx.x, nb - bias_u0, dl, b, b_ag);
    else hidden_bias_role<false, true>(hp, blockIdx.x * blockDim.x + threadIdx.x, nb - bias_u0, dl, b, b_ag);
    return;
  }
  // One workgroup per item row: its four wavefronts split the row's kept inputs (a popular row has ~500 of them per
  // 2048-user block, one L2 round trip per 8 would otherwise be a 60 us chain) and the G^T row, and meet in LDS.
  __shared__ float part[4][WAVE * NI + 1];
  const uint32_t item = blockIdx.x - bias_blocks;
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  if (item >= hp.num_items) return;
  const uint32_t lo = lane * NI;
  // summed input gradient of the row
  float din[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) din[i] = 0.f;
  bool has_in = false;
  const uint32_t beg = seg_begin[item], end = seg_end[item];
  constexpr int UN = 8;
  for (uint32_t p0 = beg + wave * WAVE; p0 < end; p0 += 4 * WAVE) {
    const uint32_t p = p0 + lane;
    const uint32_t word = p < end ? (uint32_t)sorted_val[p] : 0u;
    unsigned long long mask = __ballot((word & INPUT_BIT) != 0u);
    has_in = has_in || mask != 0ull;
    while (mask) {
      float v[UN][NI];
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        if (mask) {
          const int src = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)word, src) & SLOT_MASK;
          vload<NI>(v[t], DELTA + (size_t)slot * hp.Kp + lo);
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) v[t][i] = 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < UN; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i) din[i] += v[t][i];        // fixed order per wavefront: deterministic
    }
  }
  // b'[j] gradient: sum of the row of G^T, also split over the four wavefronts
  float gsum = 0.f;
  for (uint32_t u = threadIdx.x; u < nb; u += blockDim.x) gsum += (float)GT[(size_t)item * ldgt + u];
  gsum = wave_sum(gsum);
#pragma unroll
  for (int i = 0; i < NI; ++i) part[wave][lo + i] = din[i];
  if (lane == 0) part[wave][WAVE * NI] = has_in ? 1.f : 0.f;
  __shared__ float gpart[4];
  if (lane == 0) gpart[wave] = gsum;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int i = 0; i < NI; ++i) din[i] = ((part[0][lo + i] + part[1][lo + i]) + part[2][lo + i]) + part[3][lo + i];
  has_in = part[0][WAVE * NI] + part[1][WAVE * NI] + part[2][WAVE * NI] + part[3][WAVE * NI] > 0.f;
  gsum = ((gpart[0] + gpart[1]) + gpart[2]) + gpart[3];
  // b'[j]
  {
    float p = bp[item], pa = bp_ag[item];
    ada_step(hp, p, pa, fmaf(hp.lambda, p, gsum));
    if (lane == 0) { bp[item] = p; bp_ag[item] = pa; }
  }
  float dd[NI];
  vload<NI>(dd, dD + (size_t)item * hp.Kp + lo);
  if (!hp.asymmetric) {
    float w[NI], a[NI];
    vload<NI>(w, W + (size_t)item * hp.Kp + lo);
    vload<NI>(a, W_ag + (size_t)item * hp.Kp + lo);
#pragma unroll
    for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.scale, din[i], fmaf(hp.lambda, w[i], dd[i])));
    vstore<NI>(W + (size_t)item * hp.Kp + lo, w);
    vstore<NI>(W_ag + (size_t)item * hp.Kp + lo, a);
    if (Db) store_row_bf16<NI>(w, item, lo, hp.Kp, Ip, Db, DTb);
  } else {
    float w[NI], a[NI];
    vload<NI>(w, V + (size_t)item * hp.Kp + lo);
    vload<NI>(a, V_ag + (size_t)item * hp.Kp + lo);
#pragma unroll
    for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.lambda, w[i], dd[i]));
    vstore<NI>(V + (size_t)item * hp.Kp + lo, w);
    vstore<NI>(V_ag + (size_t)item * hp.Kp + lo, a);
    if (Db) store_row_bf16<NI>(w, item, lo, hp.Kp, Ip, Db, DTb);
    if (has_in) {
      vload<NI>(w, W + (size_t)item * hp.Kp + lo);
      vload<NI>(a, W_ag + (size_t)item * hp.Kp + lo);
#pragma unroll
      for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.scale, din[i], hp.lambda * w[i]));
      vstore<NI>(W + (size_t)item * hp.Kp + lo, w);
      vstore<NI>(W_ag + (size_t)item * hp.Kp + lo, a);
    }
  }
  if (lane == 0 && touched) touched[item] = 1u;
}

// The same row step with one wavefront per row (four rows per workgroup, no LDS hand-off): for item spaces so large that
// rows are plentiful and nearly all of them have no kept input at all (1 M items: one workgroup per row was a million
// three-round-trip chains, 3.2 ms per 1024-user block).  Every load that does not depend on the row's example list is issued up front.
template <int NI>
__global__ void __launch_bounds__(256)
full_rows_wave_kernel(HyperParams hp, const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end,
                      const uint64_t* __restrict__ sorted_val, const float* __restrict__ DELTA,
                      const float* __restrict__ dD, const __bf16* __restrict__ GT, uint32_t ldgt, uint32_t nb,
                      float* __restrict__ W, float* __restrict__ W_ag, float* __restrict__ V, float* __restrict__ V_ag,
                      float* __restrict__ bp, float* __restrict__ bp_ag, uint32_t* __restrict__ touched,
                      __bf16* __restrict__ Db = nullptr /* [Ip][Kp]: the stepped decoder row's bf16 image (one 2 NI-byte store per lane) */) {
  const uint32_t item = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  if (item >= hp.num_items) return;
  const uint32_t lo = lane * NI;
  const uint32_t beg = seg_begin[item], end = seg_end[item];
  float* const P0 = hp.asymmetric ? V : W;
  float* const P0a = hp.asymmetric ? V_ag : W_ag;
  float dd[NI], w[NI], a[NI];
  vload<NI>(dd, dD + (size_t)item * hp.Kp + lo);
  vload<NI>(w, P0 + (size_t)item * hp.Kp + lo);
  vload<NI>(a, P0a + (size_t)item * hp.Kp + lo);
  float p = bp[item], pa = bp_ag[item];
  // b'[j] gradient: the row of G^T (columns >= nb belong to padding users and are not defined)
  float gsum = 0.f;
  const __bf16* grow = GT + (size_t)item * ldgt;
  for (uint32_t u0 = lane * 8u; u0 < nb; u0 += WAVE * 8u) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(grow + u0);          // ldgt is a multiple of 128: in bounds
#pragma unroll
    for (int e = 0; e < 8; ++e) gsum += (u0 + (uint32_t)e < nb) ? (float)v[e] : 0.f;
  }
  float din[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) din[i] = 0.f;
  bool has_in = false;
  constexpr int UN = 8;
  for (uint32_t p0 = beg; p0 < end; p0 += WAVE) {
    const uint32_t q = p0 + lane;
    const uint32_t word = q < end ? (uint32_t)sorted_val[q] : 0u;
    unsigned long long mask = __ballot((word & INPUT_BIT) != 0u);
    has_in = has_in || mask != 0ull;
    while (mask) {
      float v[UN][NI];
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        if (mask) {
          const int src = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)word, src) & SLOT_MASK;
          vload<NI>(v[t], DELTA + (size_t)slot * hp.Kp + lo);
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) v[t][i] = 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < UN; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i) din[i] += v[t][i];
    }
  }
  gsum = wave_sum(gsum);
  ada_step(hp, p, pa, fmaf(hp.lambda, p, gsum));
  if (lane == 0) { bp[item] = p; bp_ag[item] = pa; }
  if (!hp.asymmetric) {
#pragma unroll
    for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.scale, din[i], fmaf(hp.lambda, w[i], dd[i])));
    vstore<NI>(W + (size_t)item * hp.Kp + lo, w);
    vstore<NI>(W_ag + (size_t)item * hp.Kp + lo, a);
    if (Db) store_row_only_bf16<NI>(w, item, lo, hp.Kp, Db);
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.lambda, w[i], dd[i]));
    vstore<NI>(V + (size_t)item * hp.Kp + lo, w);
    vstore<NI>(V_ag + (size_t)item * hp.Kp + lo, a);
    if (Db) store_row_only_bf16<NI>(w, item, lo, hp.Kp, Db);
    if (has_in) {
      vload<NI>(w, W + (size_t)item * hp.Kp + lo);
      vload<NI>(a, W_ag + (size_t)item * hp.Kp + lo);
#pragma unroll
      for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.scale, din[i], hp.lambda * w[i]));
      vstore<NI>(W + (size_t)item * hp.Kp + lo, w);
      vstore<NI>(W_ag + (size_t)item * hp.Kp + lo, a);
    }
  }
  if (lane == 0 && touched) touched[item] = 1u;
}


// ------------------------------------------------------------------------------------------------
// K > 256, item spaces >= 32768 (BASELINE configs[4]): GEMM 3 and the row step in ONE launch.
//
// The separate launches moved dD twice (2 GB written by GEMM 3, 2 GB read by the row step at 1 M items x K = 512), the row step
// read the G^T row of every item a second time for the b' gradient (2 GB) — 13 GB for a step whose own operands are 10 GB.
// Here the workgroup(s) of an item tile own 128 items x 512 k of  dD^T[k][item] = sum_u Z^T[k][u] G^T[item][u]  (contraction over
// the block's users) and keep it in the accumulators (a wavefront: k in [128 kw, 128 kw + 128) x 128 items = 4 x 4 tiles of
// v_mfma_f32_32x32x16_bf16, 256 registers; one wavefront per SIMD), then step the rows from there.  The b' gradient (the row sum
// of G^T) is summed from the operand slices that pass through LDS anyway.
//   contraction: slices of 32 users, three LDS stages (the workgroup's Z^T rows + 128 G^T rows of 64 bytes; slot c of row r holds
//   source chunk c ^ ((r >> 2) & 3): the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots), two slices in flight
//   across every raw barrier, counted vmcnt (8 + GQ DMA instructions per wavefront and slice);
//   epilogue: fr_quarter (above) — a quarter of the tile at a time through LDS, stepped as whole rows; the same piece of the next
//   quarter is requested into the registers just read, the first quarter in front of the contraction.
// Every dD element is the same sum over users in the same order as gemm_nt_bf16_ldsw_kernel's and the row step is
// full_rows_wave_kernel's expression, so D / D_ag / W come out bit-identical to the separate launches; b' differs in the
// order its gradient is summed (test_fused_rows_kernel_matches_separate_launches).
constexpr int FR_ITEMS = 128, FR_BK = 32;
// KH: workgroups per item tile.  1: 256 threads, all 512 k (40 KiB stages).  2: 128 threads and 256 k each (24 KiB stages, the G^T
// tile staged by both) — two workgroups per CU, so that one's HBM-bound row steps run beside the other's contraction.
template <int KH> constexpr int fr_stage_bytes() { return (512 / KH + FR_ITEMS) * FR_BK * 2; }
template <int KH> constexpr size_t fused_rows_lds_bytes() { return 3 * (size_t)fr_stage_bytes<KH>(); }

// Epilogue of gemm3_rows_fused_kernel.  The C layout of the MFMA gives a lane 4 consecutive k of ONE item and its 32 lanes 32
// different items: stepping the rows from there means 32-byte pieces of 32 rows per memory instruction (measured: 4.0 TB/s over
// the epilogue's 9 GB, against 5.2 TB/s for full_rows_wave_kernel's whole rows).  So a quarter of the tile at a time — 32 items x
// the workgroup's k span — goes through LDS (row stride KS * 4 + 16 bytes: the 8 lanes of a ds_write_b128 group hit 8 distinct
// 16-byte bank groups) and comes back ROW-MAJOR: a wavefront then steps whole rows, 16 bytes per lane, every access a
// contiguous 1 KiB (the workgroup's k span of one row is 1 KiB at KH = 2, 2 KiB at KH = 1).
struct FusedRowsCtx {
  float* P0; float* P0a; float* dD;
  __bf16* Db;
  char* lds;
  uint32_t in_mask;     // lane l (and l + 32): bit c set if item tile0 + 32 c + l has a kept input (stepped by full_rows_inputs_kernel from dD)
  uint32_t tile0, kbase /* first k of the workgroup */, wid, lane, num_items;
};
// piece p (0..15) of a quarter, wavefront wid: which row of the quarter and which 4 floats of the workgroup's k span this lane holds
template <int KH>
__device__ __forceinline__ void fr_piece(const FusedRowsCtx& cx, int p, uint32_t& item_local, uint32_t& col) {
  if (KH == 2) { item_local = cx.wid * 16u + (uint32_t)p; col = cx.lane * 4u; }
  else { item_local = cx.wid * 8u + (uint32_t)(p >> 1); col = (uint32_t)(p & 1) * 256u + cx.lane * 4u; }
}
template <int C, int KH>
__device__ __forceinline__ void fr_load_piece(const FusedRowsCtx& cx, int p, float4& wv, float4& av) {
  uint32_t il, col;
  fr_piece<KH>(cx, p, il, col);
  const uint32_t row = min(cx.tile0 + (uint32_t)C * 32u + il, cx.num_items - 1u);
  const size_t o = (size_t)row * 512u + cx.kbase + col;
  // D / D_ag cross the chip once per block (8 GB at 1 M items x 512: no reuse any cache could serve): nontemporal both ways,
  // 4.82 -> 4.72 ms per block; the bf16 image and the G^T slices stay plain (the forward product and the tile's other half
  // re-read them from L2: nontemporal there measured 5.05 ms).  -DCDAE_FR_PLAIN: plain accesses (A/B build)
#ifndef CDAE_FR_PLAIN
  const cdae_f4v w_ = __builtin_nontemporal_load(reinterpret_cast<const cdae_f4v*>(cx.P0 + o));
  const cdae_f4v a_ = __builtin_nontemporal_load(reinterpret_cast<const cdae_f4v*>(cx.P0a + o));
  wv = make_float4(w_[0], w_[1], w_[2], w_[3]); av = make_float4(a_[0], a_[1], a_[2], a_[3]);
#else
  wv = *reinterpret_cast<const float4*>(cx.P0 + o);
  av = *reinterpret_cast<const float4*>(cx.P0a + o);
#endif
}
template <int C, bool ADA, int KH>
__device__ __forceinline__ void fr_quarter(const HyperParams& hp, const FusedRowsCtx& cx, const f32x16 (&acc)[4][4], float4 (&wq)[16], float4 (&aq)[16]) {
  constexpr uint32_t KS = 512u / KH, RS = KS * 4u + 16u;
  __builtin_amdgcn_s_barrier();                                // the contraction's last slice / the previous quarter has been read by every wavefront
  {
    const uint32_t f_row = cx.lane & 31u, f_half = cx.lane >> 5;
    char* wp = cx.lds + f_row * RS + (cx.wid * 128u + 4u * f_half) * 4u;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)     // the products stay in the accumulation registers until their quarter leaves (left to itself hipcc 7.2 moves
                                        // all 256 to VGPRs behind the loop and spills them)
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(d[e]) : "a"(acc[kb][C][4 * q + e]));
        *reinterpret_cast<float4*>(wp + (kb * 32 + q * 8) * 4) = make_float4(d[0], d[1], d[2], d[3]);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wavefront's LDS writes are done (NOT __syncthreads(): its vmcnt(0) would drain the row requests in flight)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    uint32_t il, col;
    fr_piece<KH>(cx, p, il, col);
    const uint32_t item = cx.tile0 + (uint32_t)C * 32u + il;                                   // wave-uniform
    const bool live = item < cx.num_items;
    const bool deferred = ((uint32_t)__builtin_amdgcn_readlane((int)cx.in_mask, (int)il) >> C) & 1u;      // (in_mask is 0 for an asymmetric decoder)
    const float4 dv = *reinterpret_cast<const float4*>(cx.lds + il * RS + col * 4u);
    const size_t o = (size_t)min(item, cx.num_items - 1u) * 512u + cx.kbase + col;
    float w4[4] = {wq[p].x, wq[p].y, wq[p].z, wq[p].w}, a4[4] = {aq[p].x, aq[p].y, aq[p].z, aq[p].w};
    const float d4[4] = {dv.x, dv.y, dv.z, dv.w};
    if (C < 3) fr_load_piece<(C < 3 ? C + 1 : 3), KH>(cx, p, wq[p], aq[p]);                   // the same piece of the next quarter, into the registers just read
    if (live && deferred) {
      *reinterpret_cast<float4*>(cx.dD + o) = dv;
    } else if (live) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ada_step_t<ADA>(hp, w4[e], a4[e], fmaf(hp.lambda, w4[e], d4[e]));
#ifndef CDAE_FR_PLAIN
      const cdae_f4v wo = {w4[0], w4[1], w4[2], w4[3]}, ao = {a4[0], a4[1], a4[2], a4[3]};
      __builtin_nontemporal_store(wo, reinterpret_cast<cdae_f4v*>(cx.P0 + o));
      __builtin_nontemporal_store(ao, reinterpret_cast<cdae_f4v*>(cx.P0a + o));
#else
      *reinterpret_cast<float4*>(cx.P0 + o) = make_float4(w4[0], w4[1], w4[2], w4[3]);
      *reinterpret_cast<float4*>(cx.P0a + o) = make_float4(a4[0], a4[1], a4[2], a4[3]);
#endif
      const bf16x4 hb = {(__bf16)w4[0], (__bf16)w4[1], (__bf16)w4[2], (__bf16)w4[3]};
      *reinterpret_cast<bf16x4*>(cx.Db + o) = hb;
    }
  }
}

template <bool ADA, int KH>
__global__ void __launch_bounds__(256 / KH)
gemm3_rows_fused_kernel(HyperParams hp, const __bf16* __restrict__ ZT /* [512][ldz] */, const __bf16* __restrict__ GT /* [Ip][ldgt] */,
                        uint32_t ldz, uint32_t ldgt, uint32_t nb, const uint8_t* __restrict__ has_in /* [I], full_positive_fixup_kernel */,
                        float* __restrict__ dD /* [Ip][512]: rows with a kept input only */,
                        float* __restrict__ W, float* __restrict__ W_ag,
                        float* __restrict__ bp, float* __restrict__ bp_ag, uint32_t* __restrict__ touched,
                        __bf16* __restrict__ Db /* [Ip][512] */, uint32_t Ip) {
  extern __shared__ __attribute__((aligned(1024))) char smemf[];
  constexpr uint32_t NW = 4 / KH, ZROWS = 512 / KH, GQ = 8 / NW;              // wavefronts, staged Z^T rows, G^T DMA instructions per wavefront
  constexpr uint32_t STAGE = (uint32_t)fr_stage_bytes<KH>();
  const uint32_t lane = threadIdx.x % WAVE, wid = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);      // 0 .. NW-1
  uint32_t tile, kh;
  if (KH == 1) { tile = blockIdx.x; kh = 0; }
  else { tile = (blockIdx.x >> 4) * 8u + (blockIdx.x & 7u); kh = (blockIdx.x >> 3) & 1u; }   // both halves of a tile on ONE XCD (ids go round-robin over the 8): G^T from HBM once
  const uint32_t tile0 = tile * FR_ITEMS;
  if (tile0 >= Ip) return;
  const uint32_t kw = kh * NW + wid;                                       // this wavefront's k block of 128: [128 kw, 128 kw + 128)
  const uint32_t n_steps = CDAE_SKIP_ROLE(hp, 32u) ? 0u : (nb + FR_BK - 1) / FR_BK;     // columns >= nb of Z^T and G^T are zero (debug_skip: timing experiments only)
  f32x16 acc[4][4];                                                        // [k block][item block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: a slice is ZROWS / 16 + 8 DMA instructions of 1 KiB (16 rows of 64 bytes); a wavefront issues the 8 of its own Z^T rows and GQ of G^T
  const uint32_t st_row = lane >> 2, st_src = 8u * ((lane & 3u) ^ ((lane >> 4) & 3u));
  const __bf16* z_src[8];
  const __bf16* g_src[GQ];
#pragma unroll
  for (int q = 0; q < 8; ++q) z_src[q] = ZT + (size_t)(kw * 128u + q * 16u + st_row) * ldz + st_src;
#pragma unroll
  for (int q = 0; q < (int)GQ; ++q) g_src[q] = GT + (size_t)min(tile0 + (wid * GQ + q) * 16u + st_row, Ip - 1u) * ldgt + st_src;
  auto stage = [&](uint32_t step, uint32_t slot) {
    char* base = smemf + slot * STAGE;
    const uint32_t k = step * FR_BK;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(z_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + (wid * 8u + q) * 1024u), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < (int)GQ; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g_src[q] + k),
                                       (__attribute__((address_space(3))) void*)(base + ZROWS * 64u + (wid * GQ + q) * 1024u), 16, 0, 0);
  };
  const uint32_t f_row = lane & 31u, f_half = lane >> 5, f_sw = (f_row >> 2) & 3u;
  const uint32_t a_off = (wid * 128u + f_row) * 64u;                       // + 32 rows per k block
  const uint32_t b_off = ZROWS * 64u + f_row * 64u;                        // + 32 rows per item block
  const uint32_t gblk = kh * NW + wid;                                     // the item block whose b' gradient this wavefront sums (KH = 2: blocks 2 kh, 2 kh + 1)
  float gs = 0.f;                                                          // this lane's share of sum_u G^T[item][u], item = tile0 + 32 gblk + f_row

  // the first quarter's row pieces are requested in front of the contraction and land under it
  FusedRowsCtx cx;
  cx.P0 = W; cx.P0a = W_ag; cx.dD = dD; cx.Db = Db; cx.lds = smemf; cx.in_mask = 0;
  cx.tile0 = tile0; cx.kbase = kh * (512u / KH); cx.wid = wid; cx.lane = lane; cx.num_items = hp.num_items;
  float4 wq[16], aq[16];
  stage(0, 0);
  if (n_steps > 1) stage(1, 1);
  // Tied weights: a row that some user of the block kept as an input takes ONE step with dD + the summed input gradient
  // (cdae.hpp:252-257 and 337-348 merged).  Those rows (a few per cent at this item count) are not stepped here: their dD pieces
  // go to memory and full_rows_inputs_kernel steps them behind this launch.  Found while the first slices are in flight.
  uint32_t in_mask = 0;                                                    // bit j: item tile0 + 32 j + f_row has a kept input
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t item = tile0 + (uint32_t)j * 32u + f_row;
    in_mask |= (item < hp.num_items && !hp.asymmetric ? (uint32_t)has_in[item] : 0u) << j;      // (asymmetric: the decoder row V[item] has no input term, every row is stepped here)
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) fr_load_piece<0, KH>(cx, p, wq[p], aq[p]);
  uint32_t slot = 0;
  for (uint32_t step = 0; step < n_steps; ++step) {
    if (step + 1 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (KH == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // (8 + GQ: the DMA instructions of the slice still in flight)
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint32_t nslot = slot == 0 ? 2u : slot - 1u;
    if (step + 2 < n_steps) stage(step + 2u, nslot);
    const char* base = smemf + slot * STAGE;
    // Both sub-steps' fragments are requested before the first MFMA (round 4): the workgroup runs ONE wavefront per SIMD, so a
    // read waited for right behind its issue (what hipcc schedules by itself: read pair, lgkmcnt(0), two MFMAs) is latency no
    // other wavefront covers.  CDAE_FR_READ_LATE: the loop as the compiler orders it (A/B build)
    static_assert(FR_BK == 32, "two 16-wide sub-steps per slice");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    bf16x8 fa[2][4], fb[2][4];
    u32x4 gu[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t c16 = ((2u * s + f_half) ^ f_sw) << 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 2048u + c16);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 2048u + c16);
      gu[s] = *reinterpret_cast<const u32x4*>(base + b_off + gblk * 2048u + c16);                  // G^T piece of this wavefront's b' rows
    }
#ifndef CDAE_FR_READ_LATE
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int d = 0; d < 4; ++d) gs += __builtin_bit_cast(float, gu[s][d] << 16) + __builtin_bit_cast(float, gu[s][d] & 0xFFFF0000u);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    }
    slot = slot == 2 ? 0u : slot + 1u;
  }

  __builtin_amdgcn_s_waitcnt(WAIT_VM0);      // (nothing is in flight here; said through the builtin so that hipcc's own wait insertion counts from a known state below)
  // b'[item] of the 32 items this wavefront summed
  gs += __shfl_xor(gs, 32);
  {
    const uint32_t item = tile0 + gblk * 32u + f_row;
    if (f_half == 0 && item < hp.num_items) {
      float p = bp[item], pa = bp_ag[item];
      ada_step_t<ADA>(hp, p, pa, fmaf(hp.lambda, p, gs));
      bp[item] = p; bp_ag[item] = pa;
      if (touched) touched[item] = 1u;
    }
  }

  if (CDAE_SKIP_ROLE(hp, 16u)) {                                               // (timing experiments only: contraction without the row steps)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][j][0])); t += v; }
    if (t == 123.456f) bp[0] = 0.f;
    return;
  }
  // row steps, a quarter of the tile (32 items) at a time through LDS (fr_quarter)
  cx.in_mask = in_mask;
  fr_quarter<0, ADA, KH>(hp, cx, acc, wq, aq);
  fr_quarter<1, ADA, KH>(hp, cx, acc, wq, aq);
  fr_quarter<2, ADA, KH>(hp, cx, acc, wq, aq);
  fr_quarter<3, ADA, KH>(hp, cx, acc, wq, aq);
}

// Behind gemm3_rows_fused_kernel: the rows some user of the block kept as an input — tied weights: dD from the fused launch, the
// summed delta rows added as full_rows_wave_kernel adds them, one step.  b' of these rows was stepped by the fused launch.
template <int NI>
__global__ void __launch_bounds__(256)
full_rows_inputs_kernel(HyperParams hp, uint8_t* __restrict__ has_in, const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end,
                        const uint64_t* __restrict__ sorted_val, const float* __restrict__ DELTA, const float* __restrict__ dD,
                        float* __restrict__ W, float* __restrict__ W_ag, __bf16* __restrict__ Db, __bf16* __restrict__ DTb, uint32_t Ip) {
  // a wavefront looks at the flags of 64 items, one per lane, then steps the few with a kept input one after the other with all its
  // lanes.  Its items are strided by the number of wavefronts: popular items have neighbouring ids (and hundreds of examples each),
  // 64 of them in one wavefront were a 340 us tail
  const uint32_t n_waves = gridDim.x * (blockDim.x / WAVE), wave = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x % WAVE;
  const uint32_t my_item = wave + lane * n_waves;
  uint32_t my_beg = 0, my_end = 0;
  const bool mine = my_item < hp.num_items && has_in[my_item] != 0;
  if (mine) {
    my_beg = seg_begin[my_item]; my_end = seg_end[my_item];
    has_in[my_item] = 0;                                          // (the flag is this block's: cleared by its only reader after the fused launch)
  }
  unsigned long long todo = __ballot(mine);
  const uint32_t lo = lane * NI;
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const uint32_t item = wave + (uint32_t)src * n_waves;
    const uint32_t beg = (uint32_t)__builtin_amdgcn_readlane((int)my_beg, src), end = (uint32_t)__builtin_amdgcn_readlane((int)my_end, src);
    // tied: the decoder row, with dD from the fused launch.  Asymmetric: the INPUT row W[item] alone takes this step (the fused
    // launch stepped the decoder row V[item] itself: its gradient has no input term), dD plays no part and no image is written
    float dd[NI], w[NI], a[NI], din[NI];
    if (!hp.asymmetric) {
      vload<NI>(dd, dD + (size_t)item * hp.Kp + lo);
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) dd[i] = 0.f;
    }
    vload<NI>(w, W + (size_t)item * hp.Kp + lo);
    vload<NI>(a, W_ag + (size_t)item * hp.Kp + lo);
#pragma unroll
    for (int i = 0; i < NI; ++i) din[i] = 0.f;
    constexpr int UN = 8;
    for (uint32_t p0 = beg; p0 < end; p0 += WAVE) {
      const uint32_t q = p0 + lane;
      const uint32_t word = q < end ? (uint32_t)sorted_val[q] : 0u;
      unsigned long long mask = __ballot((word & INPUT_BIT) != 0u);
      while (mask) {
        float v[UN][NI];
#pragma unroll
        for (int t = 0; t < UN; ++t) {
          if (mask) {
            const int s2 = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)word, s2) & SLOT_MASK;
            vload<NI>(v[t], DELTA + (size_t)slot * hp.Kp + lo);
          } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) v[t][i] = 0.f;
          }
        }
#pragma unroll
        for (int t = 0; t < UN; ++t)
#pragma unroll
          for (int i = 0; i < NI; ++i) din[i] += v[t][i];
      }
    }
    if (!hp.asymmetric) {
#pragma unroll
      for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.scale, din[i], fmaf(hp.lambda, w[i], dd[i])));
    } else {                                                       // full_rows_wave_kernel's expression for the input row
#pragma unroll
      for (int i = 0; i < NI; ++i) ada_step(hp, w[i], a[i], fmaf(hp.scale, din[i], hp.lambda * w[i]));
    }
    vstore<NI>(W + (size_t)item * hp.Kp + lo, w);
    vstore<NI>(W_ag + (size_t)item * hp.Kp + lo, a);
    if (Db && !hp.asymmetric) {
      store_row_only_bf16<NI>(w, item, lo, hp.Kp, Db);
      if (DTb) {
#pragma unroll
        for (int i = 0; i < NI; ++i) DTb[(size_t)(lo + i) * Ip + item] = (__bf16)w[i];
      }
    }
  }
}

}  // namespace cdae
