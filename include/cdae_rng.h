/* cdae_rng.h — counter-based random stream shared by every consumer of the CDAE hot path.
 *
 * The reference draws its randomness from two global, order-dependent generators:
 *   - dropout corruption:  Random::uniform() > q  on a process-wide mt19937_64
 *       (/root/reference/src/model/recsys/cdae.hpp:361-371, src/base/random.hpp:34-37,82)
 *   - negative sampling:   rand() % num_items_, rejecting the user's positives
 *       (/root/reference/src/model/recsys/recsys_model_base.hpp:46-57, call site cdae.hpp:217-220)
 * Both are sequential streams, so the n-th draw depends on every draw before it; that cannot be
 * reproduced by thousands of wavefronts.  This header replaces them by a *stateless* function of
 * (seed, epoch, user, stream, index): the HIP kernels, the C++ host layer and the CPU oracle all
 * evaluate the same function and therefore see identical keep-masks and identical negative items.
 *
 * Plain C99; also compiled as HIP device code (CDAE_RNG_FN expands to __host__ __device__ there).
 */
#ifndef CDAE_RNG_H_
#define CDAE_RNG_H_

#include <stdint.h>

#ifndef CDAE_RNG_FN
#if defined(__HIPCC__)
#define CDAE_RNG_FN static __host__ __device__ __forceinline__
#else
#define CDAE_RNG_FN static inline
#endif
#endif

/* stream ids */
#define CDAE_STREAM_CORRUPT 0u      /* training dropout mask            idx = c * n_u + pos            */
#define CDAE_STREAM_NEGATIVE 1u     /* negative item draws              idx = (c*m_u + i)*MAXTRY + try */
#define CDAE_STREAM_LOSS_CORRUPT 2u /* fresh mask used by data_loss     idx = c * n_u + pos            */
#define CDAE_STREAM_INIT 3u         /* parameter init                   uid = matrix id, idx = flat    */

#define CDAE_NEG_MAX_TRY 32u        /* rejection attempts before the deterministic linear fallback */

/* splitmix64 finaliser (Steele, Lea, Flood 2014) */
CDAE_RNG_FN uint64_t cdae_mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

/* per-(seed, epoch, user, stream) key; hoist it out of inner loops */
CDAE_RNG_FN uint64_t cdae_rng_key(uint64_t seed, uint32_t epoch, uint64_t uid, uint32_t stream) {
  uint64_t x = cdae_mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)epoch + 1ull));
  x = cdae_mix64(x ^ (uid * 0xD1B54A32D192ED03ull + (uint64_t)stream + 1ull));
  return x;
}

/* idx-th 32-bit draw of a keyed stream */
CDAE_RNG_FN uint32_t cdae_rng_draw(uint64_t key, uint64_t idx) {
  return (uint32_t)(cdae_mix64(key + 0x9E3779B97F4A7C15ull * (idx + 1ull)) >> 32);
}

/* Keep rule for dropout.  The reference keeps an item when uniform() > q.  With u = r / 2^32 this is
 * r > floor(q * 2^32); the threshold is computed once on the host (double arithmetic) and handed to
 * every consumer as an integer so that no floating-point compare can disagree between CPU and GPU.
 * q <= 0 keeps everything (the reference drops an item only when uniform() returns exactly 0). */
CDAE_RNG_FN uint64_t cdae_keep_threshold(double q) {
  if (!(q > 0.0)) return 0xFFFFFFFFFFFFFFFFull;         /* sentinel: keep all */
  if (q >= 1.0) return 0x100000000ull;                 /* r > 2^32 never holds: keep none */
  return (uint64_t)(q * 4294967296.0);
}
CDAE_RNG_FN int cdae_keep(uint32_t r, uint64_t thr) {
  return thr == 0xFFFFFFFFFFFFFFFFull ? 1 : ((uint64_t)r > thr);
}

/* candidate negative item in [0, num_items): multiply-shift instead of the reference's modulo */
CDAE_RNG_FN uint32_t cdae_item_from_draw(uint32_t r, uint32_t num_items) {
  return (uint32_t)(((uint64_t)r * (uint64_t)num_items) >> 32);
}

/* membership test in a sorted CSR row (the reference probes an unordered_map,
 * recsys_model_base.hpp:50) */
CDAE_RNG_FN int cdae_row_contains(const uint32_t* row, uint32_t n, uint32_t item) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    uint32_t v = row[mid];
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return lo < n && row[lo] == item;
}

/* i-th negative of corruption c for a user with n_u positives (sorted in `row`), m_u = n_u*num_neg.
 * Rejection sampling like recsys_model_base.hpp:46-57; after CDAE_NEG_MAX_TRY rejected draws it walks
 * forward from the last candidate to the next unrated item (the reference would spin forever on a
 * user who rated everything; callers must guarantee n_u < num_items). */
CDAE_RNG_FN uint32_t cdae_sample_negative(uint64_t key_neg, uint64_t draw_index, const uint32_t* row,
                                          uint32_t n_u, uint32_t num_items) {
  uint32_t cand = 0;
  for (uint32_t t = 0; t < CDAE_NEG_MAX_TRY; ++t) {
    cand = cdae_item_from_draw(cdae_rng_draw(key_neg, draw_index * CDAE_NEG_MAX_TRY + t), num_items);
    if (!cdae_row_contains(row, n_u, cand)) return cand;
  }
  for (uint32_t t = 0; t < num_items; ++t) {
    cand = cand + 1u == num_items ? 0u : cand + 1u;
    if (!cdae_row_contains(row, n_u, cand)) return cand;
  }
  return cand;
}

/* uniform in (-1, 1) for parameter init (reference: Eigen Random(), cdae.hpp:113,116,120) */
CDAE_RNG_FN double cdae_init_uniform(uint64_t key_init, uint64_t flat_index) {
  uint32_t r = cdae_rng_draw(key_init, flat_index);
  return (((double)r + 0.5) * (1.0 / 4294967296.0)) * 2.0 - 1.0;
}

#endif /* CDAE_RNG_H_ */
