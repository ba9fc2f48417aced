/* mf_oracle.cpp — CPU restatement (fp64, single thread) of the reference's sibling SGD models IMF and BPR
 * (SURVEY.md §8(f) rank 4), built into the same libcdae_oracle.so.
 *
 * TEST INFRASTRUCTURE ONLY (see cdae_oracle.cpp): checker for the HIP path of these models, never the product.
 * PARITY UNPINNED: the reference holds no tests or fixtures for IMF / BPR either (test/model_test.hpp names a class BPR_MF that
 * does not exist in src/), and it cannot be built here.  Anchored on line citations:
 *   IMF  /root/reference/src/model/recsys/imf.hpp:57-69 (reset), :71-86 (train_one_iteration), :88-115 (train_one_instance),
 *        :117-119 (predict_user_item_rating)
 *   BPR  /root/reference/src/model/recsys/bpr.hpp:56-70 (train_one_iteration), :72-106 (train_one_pair)
 *   loss /root/reference/src/model/loss.hpp:48-55 SQUARE, :85-98 LOGISTIC, :121-160 CROSS_ENTROPY, :166-211 LOG, :262-302 HINGE
 * Order of a user's positives: the reference walks an unordered_map (implementation-defined order); here ascending item id.
 * Negatives: the counter stream of include/cdae_rng.h (draw index = positive * num_neg + k), like the CDAE restatement.
 *
 * Two schedules:
 *   literal — the reference loops: users in order, every instance (pair) steps the user vector AND the item row(s) at once.
 *   batched — what the HIP path runs for a block of B users: phase U, per user (independent): its instances in order, the
 *             user side (uv[u], ub[u]) stepped at once, the item side READ from the block-start parameters; each instance
 *             leaves (g, the user vector before its step).  Phase I, per item row (independent): the row's instances in (user,
 *             instance) order step iv[i] / ib[i] with g * uv_before + 2 lambda * row.  A block of ONE user runs the literal loop
 *             (so B = 1 IS the reference, duplicate negatives of a user included).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/cdae_rng.h"

namespace {

enum { L_SQUARE = 0, L_LOGISTIC = 1, L_LOG = 2, L_HINGE = 3, L_CE = 5 };

struct MfCfg {
  uint32_t num_dim, num_neg, loss_type, using_adagrad, using_bias_term, pairwise;
  double lambda, learn_rate, beta;
};

struct Mf {
  MfCfg c;
  size_t U = 0, I = 0, K = 0;
  std::vector<int64_t> row_ptr;
  std::vector<uint32_t> col;
  std::vector<double> uv, uv_ag, iv, iv_ag, ub, ub_ag, ib, ib_ag;      // imf.hpp:131-132

  double grad(double pred, double truth) const {
    switch (c.loss_type) {
      case L_SQUARE: return -2. * (truth - pred);                                        // loss.hpp:54
      case L_LOGISTIC: return (pred - truth) / (pred * (1. - pred));                     // loss.hpp:95-98 (CHECKs 0 < pred < 1)
      case L_LOG: {                                                                      // loss.hpp:189-197
        const double z = pred * truth;
        if (z > 18) return -truth * std::exp(-z);
        if (z < -18) return -truth;
        return -truth / (1. + std::exp(z));
      }
      case L_HINGE: return pred * truth > 1 ? 0. : -truth;                               // loss.hpp:283-288
      default:                                                                           // CROSS_ENTROPY loss.hpp:141-147
        if (pred < -18) return std::exp(pred) - truth;
        if (pred > 18) return 1 - truth;
        return 1. / (1. + std::exp(-pred)) - truth;
    }
  }
  double pos_label() const { return 1.; }
  double neg_label() const { return (c.loss_type == L_LOG || c.loss_type == L_HINGE) ? -1. : 0.; }    // loss.hpp:65,109,157,208,299

  double predict(size_t u, size_t i) const {                                             // imf.hpp:117-119
    double s = ub[u] + ib[i];
    for (size_t k = 0; k < K; ++k) s += uv[u * K + k] * iv[i * K + k];
    return s;
  }
  // one coordinate of the `if (using_adagrad_) {...} p -= lr * grad` idiom (imf.hpp:96-114)
  void step(double& p, double& acc, double g) const {
    if (c.using_adagrad) { acc += g * g; g /= (c.beta + std::sqrt(acc)); }
    p -= c.learn_rate * g;
  }
  uint32_t negative(uint64_t seed, uint32_t epoch, size_t u, uint64_t draw) const {
    const uint64_t key = cdae_rng_key(seed, epoch, u, CDAE_STREAM_NEGATIVE);
    return cdae_sample_negative(key, draw, &col[row_ptr[u]], (uint32_t)(row_ptr[u + 1] - row_ptr[u]), (uint32_t)I);
  }

  // ---- imf.hpp:88-115 ----
  void instance_literal(size_t u, size_t i, double r) {
    const double g = grad(predict(u, i), r);
    double ub_grad = g + 2. * c.lambda * ub[u], ib_grad = g + 2. * c.lambda * ib[i];
    std::vector<double> ug(K), ig(K);
    for (size_t k = 0; k < K; ++k) {
      ug[k] = g * iv[i * K + k] + 2. * c.lambda * uv[u * K + k];
      ig[k] = g * uv[u * K + k] + 2. * c.lambda * iv[i * K + k];
    }
    if (c.using_bias_term) { step(ub[u], ub_ag[u], ub_grad); step(ib[i], ib_ag[i], ib_grad); }
    for (size_t k = 0; k < K; ++k) { step(uv[u * K + k], uv_ag[u * K + k], ug[k]); step(iv[i * K + k], iv_ag[i * K + k], ig[k]); }
  }
  // ---- bpr.hpp:72-106 ----
  void pair_literal(size_t u, size_t i, size_t j) {
    const double g = grad(predict(u, i) - predict(u, j), 1.);
    double ib_grad = g + 2. * c.lambda * ib[i], jb_grad = -g + 2. * c.lambda * ib[j];
    std::vector<double> ug(K), ig(K), jg(K);
    for (size_t k = 0; k < K; ++k) {
      ug[k] = g * (iv[i * K + k] - iv[j * K + k]) + 2. * c.lambda * uv[u * K + k];
      ig[k] = g * uv[u * K + k] + 2. * c.lambda * iv[i * K + k];
      jg[k] = -g * uv[u * K + k] + 2. * c.lambda * iv[j * K + k];
    }
    if (c.using_bias_term) { step(ib[i], ib_ag[i], ib_grad); step(ib[j], ib_ag[j], jb_grad); }
    for (size_t k = 0; k < K; ++k) {
      step(uv[u * K + k], uv_ag[u * K + k], ug[k]);
      step(iv[i * K + k], iv_ag[i * K + k], ig[k]);
      step(iv[j * K + k], iv_ag[j * K + k], jg[k]);
    }
  }
  void user_literal(uint64_t seed, uint32_t epoch, size_t u) {                           // imf.hpp:71-86 / bpr.hpp:56-70
    const size_t n = (size_t)(row_ptr[u + 1] - row_ptr[u]);
    for (size_t p = 0; p < n; ++p) {
      const size_t i = col[row_ptr[u] + p];
      if (!c.pairwise) instance_literal(u, i, pos_label());
      for (size_t k = 0; k < c.num_neg; ++k) {
        const size_t j = negative(seed, epoch, u, p * c.num_neg + k);
        if (c.pairwise) pair_literal(u, i, j); else instance_literal(u, j, neg_label());
      }
    }
  }

  // ---- the block schedule of the HIP path ----
  struct Rec { uint32_t item; double sign, g; size_t uv_at; };        // one item-side contribution: sign * g * uv_before
  void block(uint64_t seed, uint32_t epoch, size_t s0, size_t s1) {
    if (s1 - s0 == 1) { user_literal(seed, epoch, s0); return; }
    std::vector<Rec> recs;
    std::vector<double> uvs;                                           // user vectors before their steps, one per instance / pair
    for (size_t u = s0; u < s1; ++u) {
      const size_t n = (size_t)(row_ptr[u + 1] - row_ptr[u]);
      // BPR (round 4): a user's num_neg pairs share their positive item and the loop steps that item's row between them
      // (bpr.hpp:84-105), so the user side carries a PRIVATE copy of the positive's row, accumulators and bias from pair to pair — taken
      // from the block-start row at the positive's first pair, stepped with the pair's own g and the user vector from before its
      // step.  (Without it every block size sat 0.008 low in Recall@10 for the first two epochs.)  Phase I is unchanged: it steps the
      // real row with the same g's in (user, pair) order.
      std::vector<double> pw(K), pa(K);
      double pb = 0., pba = 0.;
      long carried = -1;
      auto user_side = [&](size_t i, long j, double r) {
        const bool pair = j >= 0;
        if (pair && carried != (long)i) {
          for (size_t k = 0; k < K; ++k) { pw[k] = iv[i * K + k]; pa[k] = iv_ag[i * K + k]; }
          pb = ib[i]; pba = ib_ag[i];
          carried = (long)i;
        }
        double pred = ub[u] + (pair ? pb : ib[i]);
        for (size_t k = 0; k < K; ++k) pred += uv[u * K + k] * (pair ? pw[k] : iv[i * K + k]);
        if (pair) {                                                    // pairwise: pred_i - pred_j (ub cancels)
          double pj = ub[u] + ib[(size_t)j];
          for (size_t k = 0; k < K; ++k) pj += uv[u * K + k] * iv[(size_t)j * K + k];
          pred -= pj;
        }
        const double g = grad(pred, r);
        const size_t at = uvs.size();
        uvs.insert(uvs.end(), uv.begin() + u * K, uv.begin() + (u + 1) * K);
        recs.push_back(Rec{(uint32_t)i, 1., g, at});
        if (pair) recs.push_back(Rec{(uint32_t)j, -1., g, at});
        if (c.using_bias_term && !pair) step(ub[u], ub_ag[u], g + 2. * c.lambda * ub[u]);     // (BPR never steps ub)
        for (size_t k = 0; k < K; ++k) {
          const double d = pair ? pw[k] - iv[(size_t)j * K + k] : iv[i * K + k];
          step(uv[u * K + k], uv_ag[u * K + k], g * d + 2. * c.lambda * uv[u * K + k]);
        }
        if (pair) {                                                    // the private copy of the positive: bpr.hpp:79, 84-87, 93-96
          if (c.using_bias_term) step(pb, pba, g + 2. * c.lambda * pb);
          for (size_t k = 0; k < K; ++k) step(pw[k], pa[k], g * uvs[at + k] + 2. * c.lambda * pw[k]);
        }
      };
      for (size_t p = 0; p < n; ++p) {
        const size_t i = col[row_ptr[u] + p];
        if (!c.pairwise) user_side(i, -1, pos_label());
        for (size_t k = 0; k < c.num_neg; ++k) {
          const size_t j = negative(seed, epoch, u, p * c.num_neg + k);
          if (c.pairwise) user_side(i, (long)j, 1.); else user_side(j, -1, neg_label());
        }
      }
    }
    // phase I: item rows, contributions in (user, instance) order = the order they were recorded in
    std::vector<size_t> order(recs.size());
    for (size_t t = 0; t < order.size(); ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return recs[a].item < recs[b].item; });
    for (size_t t : order) {
      const Rec& r = recs[t];
      const size_t i = r.item;
      if (c.using_bias_term) step(ib[i], ib_ag[i], r.sign * r.g + 2. * c.lambda * ib[i]);
      for (size_t k = 0; k < K; ++k) step(iv[i * K + k], iv_ag[i * K + k], r.sign * r.g * uvs[r.uv_at + k] + 2. * c.lambda * iv[i * K + k]);
    }
  }
};

std::vector<double>* mf_param(Mf* o, uint32_t which) {
  switch (which) {
    case 0: return &o->uv; case 1: return &o->uv_ag; case 2: return &o->iv; case 3: return &o->iv_ag;
    case 4: return &o->ub; case 5: return &o->ub_ag; case 6: return &o->ib; case 7: return &o->ib_ag;
  }
  return nullptr;
}

}  // namespace

extern "C" {

void* mf_oracle_create(const MfCfg* cfg, uint64_t U, uint64_t I, const int64_t* row_ptr, const uint32_t* col) {
  Mf* o = new Mf();
  o->c = *cfg; o->U = U; o->I = I; o->K = cfg->num_dim;
  o->row_ptr.assign(row_ptr, row_ptr + U + 1);
  o->col.assign(col, col + row_ptr[U]);
  return o;
}
void mf_oracle_destroy(void* h) { delete (Mf*)h; }

// imf.hpp:57-69: uv, iv = Random() * 0.01; accumulators 1e-4; biases 0 — drawn from the CDAE_STREAM_INIT counter stream
// (matrix ids 4 = user vectors, 0 = item vectors: the ids the HIP path uses for Wu and W)
void mf_oracle_init_params(void* h, uint64_t seed) {
  Mf* o = (Mf*)h;
  auto fill = [&](std::vector<double>& m, size_t rows, uint32_t id) {
    const uint64_t key = cdae_rng_key(seed, 0, id, CDAE_STREAM_INIT);
    m.resize(rows * o->K);
    for (size_t i = 0; i < m.size(); ++i) m[i] = cdae_init_uniform(key, i) * 0.01;
  };
  fill(o->uv, o->U, 4); fill(o->iv, o->I, 0);
  o->uv_ag.assign(o->U * o->K, 0.0001); o->iv_ag.assign(o->I * o->K, 0.0001);
  o->ub.assign(o->U, 0.); o->ib.assign(o->I, 0.);
  o->ub_ag.assign(o->U, 0.0001); o->ib_ag.assign(o->I, 0.0001);
}
size_t mf_oracle_param_size(void* h, uint32_t which) { auto* p = mf_param((Mf*)h, which); return p ? p->size() : 0; }
int mf_oracle_get_param(void* h, uint32_t which, double* out, size_t n) {
  auto* p = mf_param((Mf*)h, which); if (!p || p->size() != n) return 1;
  std::memcpy(out, p->data(), n * sizeof(double)); return 0;
}
int mf_oracle_set_param(void* h, uint32_t which, const double* in, size_t n) {
  auto* p = mf_param((Mf*)h, which); if (!p || p->size() != n) return 1;
  p->assign(in, in + n); return 0;
}
void mf_oracle_train_literal(void* h, uint64_t seed, uint32_t epoch, uint64_t u0, uint64_t u1) {
  Mf* o = (Mf*)h;
  for (uint64_t u = u0; u < u1; ++u) o->user_literal(seed, epoch, u);
}
void mf_oracle_train_batched(void* h, uint64_t seed, uint32_t epoch, uint64_t u0, uint64_t u1, uint64_t B) {
  Mf* o = (Mf*)h;
  for (uint64_t s0 = u0; s0 < u1; s0 += B) o->block(seed, epoch, s0, std::min<uint64_t>(u1, s0 + B));
}
double mf_oracle_predict(void* h, uint64_t u, uint64_t i) { return ((Mf*)h)->predict(u, i); }
double mf_oracle_loss_grad(void* h, double pred, double truth) { return ((Mf*)h)->grad(pred, truth); }
// RecsysModelBase::recommend (recsys_model_base.hpp:77-104): top-k unrated items by predict_user_item_rating, heap semantics:
// descending score, ties to the lower item id
void mf_oracle_recommend(void* h, uint64_t u0, uint64_t u1, uint32_t topk, uint32_t* out, double* scores) {
  Mf* o = (Mf*)h;
  std::vector<std::pair<double, uint32_t>> cand;
  for (uint64_t u = u0; u < u1; ++u) {
    cand.clear();
    const uint32_t* a = &o->col[o->row_ptr[u]];
    const uint32_t* b = &o->col[o->row_ptr[u + 1]];
    for (uint32_t i = 0; i < o->I; ++i)
      if (!std::binary_search(a, b, i)) cand.push_back(std::make_pair(o->predict(u, i), i));
    std::partial_sort(cand.begin(), cand.begin() + topk, cand.end(), [](const std::pair<double, uint32_t>& x, const std::pair<double, uint32_t>& y) {
      return x.first > y.first || (x.first == y.first && x.second < y.second);
    });
    for (uint32_t t = 0; t < topk; ++t) {
      out[(u - u0) * topk + t] = cand[t].second;
      if (scores) scores[(u - u0) * topk + t] = cand[t].first;
    }
  }
}

}  // extern "C"
