/* cdae_oracle.cpp — CPU restatement (fp64, single thread) of the reference CDAE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cdae_amd/, src/ or include/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline — never as the product path.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or fixtures for CDAE
 * (test/ never includes cdae.hpp, test/loss_test.hpp:10-13 is empty) and it cannot be built in this
 * image (Eigen, Boost, glog, gflags absent; no network), so this restatement is anchored only on
 * line-by-line citations of /root/reference/src/model/recsys/cdae.hpp and on its own
 * finite-difference self-checks (tests/test_oracle.py).  The one adjacent reference test,
 * test/heap_test.hpp:10-88 (top-k ordering), is re-run against oracle_heap_* in tests/test_oracle.py.
 *
 * Two schedules:
 *   literal  — train_one_iteration exactly as cdae.hpp:136-146 + 198-358: users strictly in order,
 *              every row updated the moment the reference updates it.
 *   batched  — the schedule the HIP path executes: users are taken in blocks of B; the hidden layer of
 *              every user of a block (z_u and the hidden gradient sum_e g_e D[j_e]) is evaluated against
 *              the block-start parameters (plus the user's own duplicate-negative updates); the decode
 *              runs row-major (for each item row, its (user, target) examples in user order — every dot
 *              product sees every earlier update of that row, exactly like the reference), then
 *              hidden-bias/user-node steps in user order, then the input-row steps row-major.  With
 *              B = 1 the two schedules are the same algorithm (tests assert agreement to 1e-12; only
 *              the summation order of the hidden gradient differs).
 * Randomness: include/cdae_rng.h counter streams (the reference's global mt19937_64 / rand() are
 * order-dependent and cannot be shared with a parallel implementation; see that header).  The literal
 * schedule can ALSO be driven with the reference's own generators in the reference's own order
 * (oracle_ref_*: rand() from srand(1), std::mt19937_64, std::unordered_map iteration order), for a reader
 * who can build the reference and wants to compare a run of it draw for draw.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../include/cdae_rng.h"

namespace {

enum { LOSS_SQUARE = 0, LOSS_CE = 5 };  // loss.hpp:10-18

struct Cfg {
  uint32_t num_dim, num_neg, num_corruptions, loss_type;
  uint32_t using_adagrad, asymmetric, user_factor, linear, scaled, tanh_act, linear_function;
  double lambda, learn_rate, corruption_ratio, beta;
};

struct Oracle {
  Cfg c;
  size_t U = 0, I = 0, K = 0;
  std::vector<int64_t> row_ptr;
  std::vector<uint32_t> col;
  // cdae.hpp:430-439
  std::vector<double> W, W_ag, V, V_ag, Wu, Wu_ag, b, b_ag, bp, bp_ag;
  std::vector<double> Uu, Uu_ag;   // cdae.hpp:437-438, linear_function only

  // ---- loss.hpp:48-55 (SQUARE), loss.hpp:132-147 (CROSS_ENTROPY) ----
  double loss_eval(double pred, double truth) const {
    if (c.loss_type == LOSS_SQUARE) { double err = truth - pred; return err * err; }
    double ret = (1 - truth) * pred;                      // loss.hpp:133
    if (pred > 18) return ret + std::exp(-pred);          // loss.hpp:134-135
    if (pred < -18) return ret - pred;                    // loss.hpp:136-137
    return ret + std::log1p(std::exp(-pred));             // loss.hpp:138
  }
  double loss_grad(double pred, double truth) const {
    if (c.loss_type == LOSS_SQUARE) return -2. * (truth - pred);   // loss.hpp:54
    if (pred < -18) return std::exp(pred) - truth;                 // loss.hpp:142-143
    if (pred > 18) return 1 - truth;                               // loss.hpp:144-145
    return 1. / (1. + std::exp(-pred)) - truth;                    // loss.hpp:146
  }

  double scale() const { return c.scaled ? 1. / (1. - c.corruption_ratio) : 1.; }   // cdae.hpp:202-205

  // ---- get_hidden_values, cdae.hpp:373-416 ----
  void hidden(size_t uid, const uint32_t* items, size_t n, double sc, double* h) const {
    for (size_t k = 0; k < K; ++k) h[k] = 0.;
    for (size_t t = 0; t < n; ++t) {                                        // :377-380
      const double* w = &W[(size_t)items[t] * K];
      for (size_t k = 0; k < K; ++k) h[k] += w[k] * sc;
    }
    if (c.linear_function) for (size_t k = 0; k < K; ++k) h[k] = Uu[uid * K + k] * h[k];   // :382-384
    for (size_t k = 0; k < K; ++k) h[k] += b[k];                            // :386
    if (c.user_factor) for (size_t k = 0; k < K; ++k) h[k] += Wu[uid * K + k];   // :387-389
    if (!c.linear) {
      if (!c.tanh_act) {
        for (size_t k = 0; k < K; ++k) {                                    // :393-401
          double x = h[k];
          h[k] = x > 18. ? 1. : (x < -18. ? 0. : 1. / (1. + std::exp(-x)));
        }
      } else {
        for (size_t k = 0; k < K; ++k) {                                    // :403-412
          double x = h[k];
          if (x > 9.) h[k] = 1.;
          else if (x < -9.) h[k] = -1.;
          else { double r = std::exp(-2. * x); h[k] = (1. - r) / (1. + r); }
        }
      }
    }
  }
  // z_1_z, cdae.hpp:208-215
  void act_deriv(const double* z, double* d) const {
    for (size_t k = 0; k < K; ++k) d[k] = c.linear ? 1. : (c.tanh_act ? 1. - z[k] * z[k] : z[k] - z[k] * z[k]);
  }
  const std::vector<double>& dec() const { return c.asymmetric ? V : W; }
  // get_output_values, cdae.hpp:418-426
  double output(const double* z, size_t idx) const {
    const double* w = &dec()[idx * K];
    double s = 0;
    for (size_t k = 0; k < K; ++k) s += w[k] * z[k];
    return s + bp[idx];
  }

  // scalar AdaGrad / SGD step used for b_prime (cdae.hpp:230-237, 267-274)
  void ada1(double& p, double& acc, double grad) const {
    if (c.using_adagrad) { acc += grad * grad; grad /= (c.beta + std::sqrt(acc)); }
    p -= c.learn_rate * grad;
  }
  // row step: grad[k] already holds the full gradient incl. lambda term (cdae.hpp:252-257 etc.)
  void ada_row(double* p, double* acc, const double* grad) const {
    for (size_t k = 0; k < K; ++k) {
      double g = grad[k];
      if (c.using_adagrad) { acc[k] += g * g; g = g / (std::sqrt(acc[k]) + c.beta); }
      p[k] -= c.learn_rate * g;
    }
  }

  // ---- corruption + negatives for (uid, corruption c) from the counter streams ----
  void draw_inputs(uint64_t seed, uint32_t epoch, size_t uid, uint32_t cidx, uint32_t stream,
                   std::vector<uint32_t>& in) const {
    const uint32_t* row = &col[row_ptr[uid]];
    size_t n = row_ptr[uid + 1] - row_ptr[uid];
    uint64_t key = cdae_rng_key(seed, epoch, uid, stream);
    uint64_t thr = cdae_keep_threshold(c.corruption_ratio);
    in.clear();
    for (size_t p = 0; p < n; ++p)                                           // cdae.hpp:365-369
      if (cdae_keep(cdae_rng_draw(key, (uint64_t)cidx * n + p), thr)) in.push_back(row[p]);
  }
  void draw_negatives(uint64_t seed, uint32_t epoch, size_t uid, uint32_t cidx,
                      std::vector<uint32_t>& neg) const {
    const uint32_t* row = &col[row_ptr[uid]];
    size_t n = row_ptr[uid + 1] - row_ptr[uid];
    size_t m = n * c.num_neg;                                                // cdae.hpp:217
    uint64_t key = cdae_rng_key(seed, epoch, uid, CDAE_STREAM_NEGATIVE);
    neg.resize(m);
    for (size_t i = 0; i < m; ++i)                                           // cdae.hpp:218-220
      neg[i] = cdae_sample_negative(key, (uint64_t)cidx * m + i, row, (uint32_t)n, (uint32_t)I);
  }

  // ---- train_one_user_corruption, cdae.hpp:198-358 ----
  // Optional taps (z, y per output, g per output, hg) for the known-answer fixtures.
  void train_user_literal(size_t uid, const uint32_t* in, size_t n_in, const uint32_t* neg,
                          size_t n_neg, double* tap_z, double* tap_y, double* tap_g, double* tap_hg) {
    train_user_seq(uid, &col[row_ptr[uid]], row_ptr[uid + 1] - row_ptr[uid], in, n_in, neg, n_neg, tap_z, tap_y, tap_g, tap_hg);
  }
  // The same step with the two container orders made explicit: `pos` = the order `for (auto& p : output_set)` visits the
  // user's train items (cdae.hpp:225), `in` = the order `for (auto& p : input_set)` visits the kept inputs (:333 and the
  // encode's :377).  train_user_literal passes both ascending (CSR order); the reference-sequenced mode below passes the
  // iteration orders of the reference's std::unordered_map containers.  The orders only change rounding (every row is touched
  // once per loop), but they decide WHICH item a dropout draw belongs to.
  void train_user_seq(size_t uid, const uint32_t* pos, size_t n_pos, const uint32_t* in, size_t n_in, const uint32_t* neg,
                      size_t n_neg, double* tap_z, double* tap_y, double* tap_g, double* tap_hg) {
    double sc = scale();
    std::vector<double> z(K), d(K), hg(K, 0.), grad(K);
    hidden(uid, in, n_in, sc, z.data());                                     // :207
    act_deriv(z.data(), d.data());                                           // :208-215
    if (tap_z) std::memcpy(tap_z, z.data(), K * sizeof(double));
    std::vector<double> defer_g(n_pos, 0.);                                  // input_gradient, :222
    std::vector<char> is_in(n_pos, 0);
    std::vector<size_t> pos_of_in(n_in, 0);                                  // position in `pos` of every kept input
    {                                                                        // input_set.count(iid), :249
      std::vector<std::pair<uint32_t, size_t>> by_item(n_pos);
      for (size_t p = 0; p < n_pos; ++p) by_item[p] = std::make_pair(pos[p], p);
      std::sort(by_item.begin(), by_item.end());
      for (size_t t = 0; t < n_in; ++t) {
        auto it = std::lower_bound(by_item.begin(), by_item.end(), std::make_pair(in[t], (size_t)0));
        pos_of_in[t] = it->second;                                           // (a kept input is one of the user's train items)
        is_in[it->second] = 1;
      }
    }
    std::vector<double>& D = c.asymmetric ? V : W;
    std::vector<double>& D_ag = c.asymmetric ? V_ag : W_ag;
    for (size_t p = 0; p < n_pos; ++p) {                                     // :225-260
      size_t iid = pos[p];
      double y = output(z.data(), iid);                                      // :227
      double g = loss_grad(y, 1.);                                           // :228
      if (tap_y) tap_y[p] = y;
      if (tap_g) tap_g[p] = g;
      ada1(bp[iid], bp_ag[iid], g + c.lambda * bp[iid]);                     // :230-237
      double* row = &D[iid * K];
      for (size_t k = 0; k < K; ++k) hg[k] += g * row[k];                    // :240 / :248 (pre-update row)
      if (!c.asymmetric && is_in[p]) { defer_g[p] = g; continue; }           // :249-250
      for (size_t k = 0; k < K; ++k) grad[k] = g * z[k] + c.lambda * row[k]; // :241 / :252
      ada_row(row, &D_ag[iid * K], grad.data());                             // :242-246 / :253-257
    }
    for (size_t i = 0; i < n_neg; ++i) {                                     // :262-293
      size_t iid = neg[i];
      double y = output(z.data(), iid);                                      // :263
      double g = loss_grad(y, 0.);                                           // :265
      if (tap_y) tap_y[n_pos + i] = y;
      if (tap_g) tap_g[n_pos + i] = g;
      ada1(bp[iid], bp_ag[iid], g + c.lambda * bp[iid]);                     // :267-274
      double* row = &D[iid * K];
      for (size_t k = 0; k < K; ++k) hg[k] += g * row[k];                    // :277 / :285
      for (size_t k = 0; k < K; ++k) grad[k] = g * z[k] + c.lambda * row[k]; // :278 / :286
      ada_row(row, &D_ag[iid * K], grad.data());                             // :279-283 / :287-291
    }
    if (tap_hg) std::memcpy(tap_hg, hg.data(), K * sizeof(double));
    std::vector<double> delta(K);
    for (size_t k = 0; k < K; ++k) delta[k] = hg[k] * d[k];
    for (size_t k = 0; k < K; ++k) grad[k] = delta[k] + c.lambda * b[k];     // :305
    ada_row(b.data(), b_ag.data(), grad.data());                             // :310-314
    if (c.user_factor) {                                                     // :317-331
      double* wu = &Wu[uid * K];
      for (size_t k = 0; k < K; ++k) grad[k] = delta[k] + c.lambda * wu[k];
      ada_row(wu, &Wu_ag[uid * K], grad.data());
    }
    std::vector<double> uu_grad;
    if (c.linear_function) {                                                 // :295-299
      uu_grad.resize(K);
      for (size_t k = 0; k < K; ++k) uu_grad[k] = Uu[uid * K + k] * c.lambda;
    }
    for (size_t t = 0; t < n_in; ++t) {                                      // :333-349 (input rows, in `in` order)
      const size_t p = pos_of_in[t];
      size_t jid = pos[p];
      double* row = &W[jid * K];
      for (size_t k = 0; k < K; ++k) {
        double g;
        if (!c.linear_function) g = delta[k] * sc + c.lambda * row[k];       // :337
        else {
          g = Uu[uid * K + k] * delta[k] * sc + c.lambda * row[k];           // :339
          uu_grad[k] += delta[k] * row[k];                                   // :340 (row before its step)
        }
        if (!c.asymmetric) g += defer_g[p] * z[k];                           // :342-343 (input_gradient = g*z, :250)
        grad[k] = g;
      }
      ada_row(row, &W_ag[jid * K], grad.data());                             // :344-348
    }
    if (c.linear_function) ada_row(&Uu[uid * K], &Uu_ag[uid * K], uu_grad.data());   // :351-357
  }

  // ---- train_one_iteration, cdae.hpp:136-146 ----
  void train_users_literal(uint64_t seed, uint32_t epoch, size_t u0, size_t u1) {
    std::vector<uint32_t> in, neg;
    for (size_t uid = u0; uid < u1; ++uid)                                    // :137
      for (uint32_t ci = 0; ci < c.num_corruptions; ++ci) {                  // :141
        draw_inputs(seed, epoch, uid, ci, CDAE_STREAM_CORRUPT, in);          // :142
        draw_negatives(seed, epoch, uid, ci, neg);                           // :217-220
        train_user_literal(uid, in.data(), in.size(), neg.data(), neg.size(), nullptr, nullptr, nullptr, nullptr);
      }
  }

  // ---- reference-SEQUENCED randomness (SURVEY.md §7 step 2 "glibc mode") -------------------------------------------
  // The reference draws from two process-global generators: C rand() — never seeded, i.e. srand(1) — for Eigen's
  // DMatrix::Random in reset() (cdae.hpp:112-113,116,120) and for sample_negative_item (recsys_model_base.hpp:46-57:
  // `rand() % num_items_` until unrated), and libcf::Random::rng, a std::mt19937_64 (random.hpp:14,82), for the dropout
  // masks (`Random::uniform() > corruption_ratio`, cdae.hpp:361-371; random.hpp:34-37: std::uniform_real_distribution<>).
  // This mode consumes both in exactly the reference's order, through the reference's own containers
  // (std::unordered_map<size_t,double> built by insertion as data-inl.hpp:414-429 builds it, so its iteration order — which
  // item a draw belongs to — is the reference's when both are compiled against the same libstdc++), so that a reader who CAN
  // build the reference (Eigen, Boost, glog, gflags) can line a run of it up with this oracle draw for draw.
  // rand() is restated (glibc stdlib/random_r.c, TYPE_3: r[i] = r[i-3] + r[i-31], output >> 1) instead of called, so that the
  // oracle neither depends on nor disturbs the process-global state; tests/test_oracle.py pins the restatement against this
  // platform's rand().  Assumed, not verifiable here (no Eigen): DenseBase::Random() = -1 + 2 * rand() / RAND_MAX per
  // coefficient (Eigen 3.x internal::random<double>), coefficients visited in storage order (row-major, mat.hpp:12).
  struct GlibcRand {
    uint32_t ring[31];
    int f = 3, b = 0;
    void seed(uint32_t s) {
      if (s == 0) s = 1;
      int32_t w[31];
      w[0] = (int32_t)s;
      for (int i = 1; i < 31; ++i) {                       // Schrage form of 16807 * w mod (2^31 - 1)
        const int32_t hi = w[i - 1] / 127773, lo = w[i - 1] % 127773;
        int32_t v = 16807 * lo - 2836 * hi;
        if (v < 0) v += 2147483647;
        w[i] = v;
      }
      for (int i = 0; i < 31; ++i) ring[i] = (uint32_t)w[i];
      f = 3; b = 0;
      for (int i = 0; i < 310; ++i) next();                // glibc discards 10 * degree outputs
    }
    uint32_t next() {
      ring[f] += ring[b];
      const uint32_t out = ring[f] >> 1;
      if (++f == 31) f = 0;
      if (++b == 31) b = 0;
      return out;
    }
  };
  GlibcRand ref_rand;
  std::mt19937_64 ref_mt;
  std::vector<uint32_t> ref_insertion;      // per user (row_ptr layout) its items in data order; empty: ascending
  void ref_seed(uint64_t mt_seed, uint32_t rand_seed) { ref_mt.seed(mt_seed); ref_rand.seed(rand_seed); }
  double ref_uniform() { std::uniform_real_distribution<> dist(0., 1.); return dist(ref_mt); }     // random.hpp:34-37
  void ref_item_set(size_t uid, std::unordered_map<size_t, double>& m) const {                      // data-inl.hpp:420-426
    const uint32_t* items = ref_insertion.empty() ? &col[row_ptr[uid]] : &ref_insertion[row_ptr[uid]];
    const size_t n = row_ptr[uid + 1] - row_ptr[uid];
    std::unordered_map<size_t, double> tmp;
    for (size_t p = 0; p < n; ++p) tmp.insert(std::make_pair((size_t)items[p], 1.));
    m = std::move(tmp);
  }
  // one user-corruption's draws, in the reference's order: the mask first (train_one_iteration, cdae.hpp:142), then the
  // negatives (train_one_user_corruption, :217-220)
  void ref_draw_user(const std::unordered_map<size_t, double>& item_set, std::vector<uint32_t>& pos_seq,
                     std::vector<uint32_t>& in_seq, std::vector<uint32_t>& neg) {
    std::unordered_map<size_t, double> rets;                                                        // cdae.hpp:363-370
    rets.reserve((size_t)(item_set.size() * (1. - c.corruption_ratio)));
    for (auto& p : item_set) if (ref_uniform() > c.corruption_ratio) rets.insert(p);
    pos_seq.clear(); in_seq.clear();
    for (auto& p : item_set) pos_seq.push_back((uint32_t)p.first);
    for (auto& p : rets) in_seq.push_back((uint32_t)p.first);
    neg.resize(item_set.size() * c.num_neg);                                                        // :217
    for (size_t i = 0; i < neg.size(); ++i) {                                                       // recsys_model_base.hpp:46-57
      size_t it;
      do { it = (size_t)ref_rand.next() % I; } while (item_set.count(it));
      neg[i] = (uint32_t)it;
    }
  }
  void train_users_reference_sequenced(size_t u0, size_t u1) {
    std::unordered_map<size_t, double> item_set;
    std::vector<uint32_t> pos_seq, in_seq, neg;
    for (size_t uid = u0; uid < u1; ++uid) {                                                        // cdae.hpp:137
      ref_item_set(uid, item_set);
      for (uint32_t ci = 0; ci < c.num_corruptions; ++ci) {                                         // :141
        ref_draw_user(item_set, pos_seq, in_seq, neg);
        train_user_seq(uid, pos_seq.data(), pos_seq.size(), in_seq.data(), in_seq.size(), neg.data(), neg.size(),
                       nullptr, nullptr, nullptr, nullptr);
      }
    }
  }

  // ---- the HIP path's schedule (see file header) ----
  struct Ex { uint32_t item; uint32_t slot; uint8_t target; uint8_t is_in; uint32_t order; };
  void train_users_batched(uint64_t seed, uint32_t epoch, size_t u0, size_t u1, size_t B) {
    if (B == 0) B = 1;
    double sc = scale();
    std::vector<double>& D = c.asymmetric ? V : W;
    std::vector<double>& D_ag = c.asymmetric ? V_ag : W_ag;
    std::vector<uint32_t> in, neg;
    std::vector<double> grad(K);
    for (size_t s0 = u0; s0 < u1; s0 += B) {
      size_t s1 = std::min(u1, s0 + B), nb = s1 - s0;
      for (uint32_t ci = 0; ci < c.num_corruptions; ++ci) {
        // phase A: sample + encode with block-start parameters
        std::vector<double> Z(nb * K), Dv(nb * K), HG(nb * K, 0.);
        std::vector<double> SSUM(c.linear_function ? nb * K : 0, 0.);   // unscaled input sums (block-start rows)
        std::vector<Ex> ex;
        std::vector<double> G;     // per example loss gradient (for deferred rows)
        for (size_t s = 0; s < nb; ++s) {
          size_t uid = s0 + s;
          const uint32_t* pos = &col[row_ptr[uid]];
          size_t n_pos = row_ptr[uid + 1] - row_ptr[uid];
          draw_inputs(seed, epoch, uid, ci, CDAE_STREAM_CORRUPT, in);
          draw_negatives(seed, epoch, uid, ci, neg);
          hidden(uid, in.data(), in.size(), sc, &Z[s * K]);
          act_deriv(&Z[s * K], &Dv[s * K]);
          if (c.linear_function) input_sum(in.data(), in.size(), &SSUM[s * K]);
          size_t t = 0;
          for (size_t p = 0; p < n_pos; ++p) {
            while (t < in.size() && in[t] < pos[p]) ++t;
            bool isin = t < in.size() && in[t] == pos[p];
            ex.push_back(Ex{pos[p], (uint32_t)s, 1, (uint8_t)isin, (uint32_t)ex.size()});
          }
          for (size_t i = 0; i < neg.size(); ++i) ex.push_back(Ex{neg[i], (uint32_t)s, 0, 0, (uint32_t)ex.size()});
        }
        G.assign(ex.size(), 0.);
        const std::vector<double> D0(D);     // decoder rows at block start (hidden-gradient snapshot)
        // row-major order: stable by item (order == user order, positives before negatives)
        std::vector<Ex> sorted(ex);
        std::stable_sort(sorted.begin(), sorted.end(), [](const Ex& a, const Ex& b) { return a.item < b.item; });
        // phase B: decode, every row sequential over its examples
        std::vector<double> wref(K);
        uint32_t prev_item = 0xFFFFFFFFu, prev_slot = 0xFFFFFFFFu;
        for (const Ex& e : sorted) {
          size_t iid = e.item;
          const double* z = &Z[(size_t)e.slot * K];
          double y = output(z, iid);
          double g = loss_grad(y, e.target ? 1. : 0.);
          ada1(bp[iid], bp_ag[iid], g + c.lambda * bp[iid]);
          double* row = &D[iid * K];
          G[e.order] = g;
          if (e.item == prev_item && e.slot == prev_slot) {
            // the same user's duplicate negative: its hidden gradient sees its own earlier update of the row
            double* hg = &HG[(size_t)e.slot * K];
            for (size_t k = 0; k < K; ++k) hg[k] += g * (row[k] - wref[k]);
          } else {
            prev_item = e.item; prev_slot = e.slot;
            for (size_t k = 0; k < K; ++k) wref[k] = row[k];
          }
          if (!c.asymmetric && e.is_in) continue;
          for (size_t k = 0; k < K; ++k) grad[k] = g * z[k] + c.lambda * row[k];
          ada_row(row, &D_ag[iid * K], grad.data());
        }
        // hidden gradient against the block-start decoder rows (+ the duplicate corrections above)
        for (const Ex& e : ex) {
          double* hg = &HG[(size_t)e.slot * K];
          const double* row0 = &D0[(size_t)e.item * K];
          for (size_t k = 0; k < K; ++k) hg[k] += G[e.order] * row0[k];
        }
        // phase C: hidden bias + user node(s), user order
        std::vector<double> DELTA(nb * K);
        for (size_t s = 0; s < nb; ++s) {
          size_t uid = s0 + s;
          double* delta = &DELTA[s * K];
          for (size_t k = 0; k < K; ++k) delta[k] = HG[s * K + k] * Dv[s * K + k];
          for (size_t k = 0; k < K; ++k) grad[k] = delta[k] + c.lambda * b[k];
          ada_row(b.data(), b_ag.data(), grad.data());
          if (c.user_factor) {
            double* wu = &Wu[uid * K];
            for (size_t k = 0; k < K; ++k) grad[k] = delta[k] + c.lambda * wu[k];
            ada_row(wu, &Wu_ag[uid * K], grad.data());
          }
          if (c.linear_function) uu_step(uid, delta, &SSUM[s * K], grad.data());
        }
        // phase D: input rows, row-major, user order inside a row
        for (const Ex& e : sorted) {
          if (!e.is_in) continue;
          size_t jid = e.item;
          double* row = &W[jid * K];
          const double* delta = &DELTA[(size_t)e.slot * K];
          const double* z = &Z[(size_t)e.slot * K];
          for (size_t k = 0; k < K; ++k) {
            double g = delta[k] * sc + c.lambda * row[k];          // delta already carries Uu[u] (uu_step)
            if (!c.asymmetric) g += G[e.order] * z[k];
            grad[k] = g;
          }
          ada_row(row, &W_ag[jid * K], grad.data());
        }
      }
    }
  }

  // linear_function in the block schedules.  The reference accumulates Uu_grad += delta (.) W[k] over the kept inputs
  // with each row read just before its own step (cdae.hpp:340); inside one user's step those rows have not moved since
  // the encode (tied mode defers them, cdae.hpp:249-250; asymmetric mode's decode never touches W), so the sum is
  // delta (.) sum_k W[k] with the rows of the encode — the block schedules use the block-start rows, which is the same
  // thing at batch_users == 1.  The input rows see Uu[u] from before its step (cdae.hpp:339 precedes :351-357).
  void input_sum(const uint32_t* items, size_t n, double* out) const {
    for (size_t k = 0; k < K; ++k) out[k] = 0.;
    for (size_t t = 0; t < n; ++t) {
      const double* w = &W[(size_t)items[t] * K];
      for (size_t k = 0; k < K; ++k) out[k] += w[k];
    }
  }
  void uu_step(size_t uid, double* delta /* in: delta, out: Uu_old (.) delta */, const double* ssum, double* grad) {
    double* uu = &Uu[uid * K];
    for (size_t k = 0; k < K; ++k) {
      grad[k] = c.lambda * uu[k] + delta[k] * ssum[k];
      delta[k] *= uu[k];
    }
    ada_row(uu, &Uu_ag[uid * K], grad);
  }

  // ---- full-output decode (north-star extension) ----
  // Every non-positive item is a negative with target 0, each exactly once per user.  The reference has no
  // such training mode (its decode is always sampled, cdae.hpp:217-293); this is the limit of that loop with
  // the negative list = all unrated items, restated in the dense form the MFMA path computes per block of B
  // users from the block-start parameters:
  //   Y = Z D^T + b' ;  G = loss'(Y, T) ;  hg = G D ;  dD = G^T Z ;  db' = column sums of G
  //   one step per decoder row with the block's summed gradient dD[j] + lambda D[j] (+, tied mode, the summed
  //   input gradient scale * sum_{u: j kept} delta_u, merged like cdae.hpp:337-343), one step per b'[j],
  //   b and Wu[u] steps in user order as in the sampled schedule.
  // With B = 1 every row is touched once per user and this IS the reference loop (tests compare it with
  // train_user_literal fed all unrated items as negatives).
  void train_users_full(uint64_t seed, uint32_t epoch, size_t u0, size_t u1, size_t B) {
    if (B == 0) B = 1;
    const double sc = scale();
    std::vector<double>& D = c.asymmetric ? V : W;
    std::vector<double>& D_ag = c.asymmetric ? V_ag : W_ag;
    std::vector<uint32_t> in;
    std::vector<double> grad(K);
    for (size_t s0 = u0; s0 < u1; s0 += B) {
      const size_t s1 = std::min(u1, s0 + B), nb = s1 - s0;
      for (uint32_t ci = 0; ci < c.num_corruptions; ++ci) {
        std::vector<double> Z(nb * K), Dv(nb * K), HG(nb * K, 0.), DELTA(nb * K);
        std::vector<double> SSUM(c.linear_function ? nb * K : 0, 0.);
        std::vector<double> dD(I * K, 0.), dbp(I, 0.), dIn(I * K, 0.);
        std::vector<char> has_in(I, 0);
        std::vector<std::vector<uint32_t>> kept(nb);
        for (size_t s = 0; s < nb; ++s) {
          const size_t uid = s0 + s;
          draw_inputs(seed, epoch, uid, ci, CDAE_STREAM_CORRUPT, in);
          kept[s] = in;
          hidden(uid, in.data(), in.size(), sc, &Z[s * K]);
          act_deriv(&Z[s * K], &Dv[s * K]);
          if (c.linear_function) input_sum(in.data(), in.size(), &SSUM[s * K]);
        }
        for (size_t s = 0; s < nb; ++s) {                      // dense decode against the block-start rows
          const size_t uid = s0 + s;
          const uint32_t* pos = &col[row_ptr[uid]];
          const size_t n_pos = row_ptr[uid + 1] - row_ptr[uid];
          const double* z = &Z[s * K];
          size_t t = 0;
          for (size_t j = 0; j < I; ++j) {
            while (t < n_pos && pos[t] < j) ++t;
            const double truth = (t < n_pos && pos[t] == j) ? 1. : 0.;
            const double g = loss_grad(output(z, j), truth);
            const double* row = &D[j * K];
            for (size_t k = 0; k < K; ++k) { HG[s * K + k] += g * row[k]; dD[j * K + k] += g * z[k]; }
            dbp[j] += g;
          }
        }
        // EXPERIMENT of round 4 (CDAE_ORACLE_B_SUMMED=1; the HIP path's CDAE_FULL_B_SUMMED): b takes ONE step per block with the summed delta
        // instead of a step per user.  Not the schedule the fixtures hold — DESIGN.md §5c has what it does to the curves.
        static const bool b_summed = std::getenv("CDAE_ORACLE_B_SUMMED") != nullptr;
        std::vector<double> bsum(K, 0.);
        for (size_t s = 0; s < nb; ++s) {                      // hidden layer: delta, b (user order), Wu[u]
          const size_t uid = s0 + s;
          double* delta = &DELTA[s * K];
          for (size_t k = 0; k < K; ++k) delta[k] = HG[s * K + k] * Dv[s * K + k];
          if (b_summed) { for (size_t k = 0; k < K; ++k) bsum[k] += delta[k]; }
          else {
            for (size_t k = 0; k < K; ++k) grad[k] = delta[k] + c.lambda * b[k];
            ada_row(b.data(), b_ag.data(), grad.data());
          }
          if (c.user_factor) {
            double* wu = &Wu[uid * K];
            for (size_t k = 0; k < K; ++k) grad[k] = delta[k] + c.lambda * wu[k];
            ada_row(wu, &Wu_ag[uid * K], grad.data());
          }
          if (c.linear_function) uu_step(uid, delta, &SSUM[s * K], grad.data());
          for (uint32_t j : kept[s]) { has_in[j] = 1; for (size_t k = 0; k < K; ++k) dIn[(size_t)j * K + k] += sc * delta[k]; }
        }
        if (b_summed) {
          for (size_t k = 0; k < K; ++k) grad[k] = bsum[k] + c.lambda * b[k];
          ada_row(b.data(), b_ag.data(), grad.data());
        }
        for (size_t j = 0; j < I; ++j) {                       // one step per row with the block's summed gradient
          ada1(bp[j], bp_ag[j], dbp[j] + c.lambda * bp[j]);
          double* drow = &D[j * K];
          if (!c.asymmetric) {
            for (size_t k = 0; k < K; ++k) grad[k] = dD[j * K + k] + dIn[j * K + k] + c.lambda * drow[k];
            ada_row(drow, &D_ag[j * K], grad.data());
          } else {
            for (size_t k = 0; k < K; ++k) grad[k] = dD[j * K + k] + c.lambda * drow[k];
            ada_row(drow, &D_ag[j * K], grad.data());
            if (has_in[j]) {
              double* wrow = &W[j * K];
              for (size_t k = 0; k < K; ++k) grad[k] = dIn[j * K + k] + c.lambda * wrow[k];
              ada_row(wrow, &W_ag[j * K], grad.data());
            }
          }
        }
      }
    }
  }

  // ---- data_loss, cdae.hpp:78-101 ; penalty_loss, cdae.hpp:103-107 + penalty.hpp:36-39 ----
  double data_loss(uint64_t seed, uint32_t epoch) const {
    double rets = 0.;
    std::vector<uint32_t> in;
    std::vector<double> z(K);
    for (size_t uid = 0; uid < U; ++uid) {
      const uint32_t* pos = &col[row_ptr[uid]];
      size_t n_pos = row_ptr[uid + 1] - row_ptr[uid];
      double user_rets = 0;
      for (uint32_t ci = 0; ci < c.num_corruptions; ++ci) {
        draw_inputs(seed, epoch, uid, ci, CDAE_STREAM_LOSS_CORRUPT, in);     // :87
        hidden(uid, in.data(), in.size(), scale(), z.data());                // :88-92
        for (size_t p = 0; p < n_pos; ++p) user_rets += loss_eval(output(z.data(), pos[p]), 1.);   // :93-96
      }
      rets = rets + user_rets / c.num_corruptions;                           // :98
    }
    return rets;
  }
  static double sqnorm(const std::vector<double>& v) { double s = 0; for (double x : v) s += x * x; return s; }
  double penalty_loss() const {
    return 0.5 * c.lambda * (sqnorm(W) + sqnorm(V) + sqnorm(Wu) + sqnorm(b) + sqnorm(bp));
  }

  // ---- recommend, cdae.hpp:162-196 with Heap (heap.hpp:29-52,66-69) and sort_by_second_desc (utils.hpp:16-19) ----
  void recommend(size_t uid, size_t topk, uint32_t* out, double* out_score) const {
    typedef std::pair<size_t, double> P;
    auto comp = [](const P& a, const P& b) { return a.second > b.second; };
    const uint32_t* pos = &col[row_ptr[uid]];
    size_t n_pos = row_ptr[uid + 1] - row_ptr[uid];
    std::vector<double> z(K);
    if (c.corruption_ratio != 1.) hidden(uid, pos, n_pos, 1.0, z.data());   // :168-169 (default scale 1)
    else hidden(uid, pos, 0, 1.0, z.data());                                 // :170-172
    std::vector<P> heap;
    heap.reserve(topk);
    size_t t = 0;
    for (size_t item = 0; item < I; ++item) {                                // :176
      while (t < n_pos && pos[t] < item) ++t;
      if (t < n_pos && pos[t] == item) continue;                             // :177-179
      P cand(item, output(z.data(), item));                                  // :180
      if (heap.size() < topk) { heap.push_back(cand); std::push_heap(heap.begin(), heap.end(), comp); }   // :181-182
      else if (comp(cand, heap.front())) {                                   // heap.hpp:44-52
        std::pop_heap(heap.begin(), heap.end(), comp); heap.pop_back();
        heap.push_back(cand); std::push_heap(heap.begin(), heap.end(), comp);
      }
    }
    std::sort_heap(heap.begin(), heap.end(), comp);                          // :188, heap.hpp:66-69
    for (size_t i = 0; i < topk; ++i) {
      out[i] = i < heap.size() ? (uint32_t)heap[i].first : 0xFFFFFFFFu;
      if (out_score) out_score[i] = i < heap.size() ? heap[i].second : 0.;
    }
  }
};

// evaluate_rec_list, evaluation.hpp:183-219
void eval_rec_list(const uint32_t* list, size_t n_list, const uint32_t* truth, size_t n_truth, double* rets) {
  for (int i = 0; i < 8; ++i) rets[i] = 0.;
  size_t TOPK = std::min<size_t>(20, n_list);
  double hit = 0., map5 = 0., map10 = 0.;
  for (size_t idx = 0; idx < TOPK; ++idx) {
    if (std::binary_search(truth, truth + n_truth, list[idx])) {
      hit += 1.;
      if (idx < 5) map5 += hit / (idx + 1);
      if (idx < 10) map10 += hit / (idx + 1);
    }
    if (idx == 0) { rets[0] = hit / 1.; rets[3] = hit / n_truth; }
    else if (idx == 4) { rets[1] = hit / 5.; rets[4] = hit / n_truth; }
    else if (idx == 9) { rets[2] = hit / 10.; rets[5] = hit / n_truth; }
  }
  rets[6] = map5 / (double)std::min<size_t>(5, n_truth);
  rets[7] = map10 / (double)std::min<size_t>(10, n_truth);
}

std::vector<double>* param(Oracle* o, uint32_t which) {
  switch (which) {
    case 0: return &o->W; case 1: return &o->W_ag; case 2: return &o->V; case 3: return &o->V_ag;
    case 4: return &o->Wu; case 5: return &o->Wu_ag; case 6: return &o->b; case 7: return &o->b_ag;
    case 8: return &o->bp; case 9: return &o->bp_ag;
    case 10: return &o->Uu; case 11: return &o->Uu_ag;
  }
  return nullptr;
}

}  // namespace

extern "C" {

struct oracle_config {   // mirrors cdae_hip_config minus struct_size / batch_users
  uint32_t num_dim, num_neg, num_corruptions, loss_type;
  uint32_t using_adagrad, asymmetric, user_factor, linear, scaled, tanh_act, linear_function;
  double lambda, learn_rate, corruption_ratio, beta;
};

void* oracle_create(const oracle_config* cfg, uint64_t U, uint64_t I, const int64_t* row_ptr, const uint32_t* col) {
  Oracle* o = new Oracle();
  std::memcpy(&o->c, cfg, sizeof(Cfg));
  o->U = U; o->I = I; o->K = cfg->num_dim;
  o->row_ptr.assign(row_ptr, row_ptr + U + 1);
  o->col.assign(col, col + row_ptr[U]);
  return o;
}
void oracle_destroy(void* h) { delete (Oracle*)h; }

// reset(), cdae.hpp:109-134, with the counter-stream init of include/cdae_rng.h
void oracle_init_params(void* h, uint64_t seed) {
  Oracle* o = (Oracle*)h;
  size_t K = o->K;
  double init_scale = 4. * std::sqrt(6. / (double)(o->I + K));               // :112
  auto fill = [&](std::vector<double>& m, size_t rows, uint32_t id) {
    uint64_t key = cdae_rng_key(seed, 0, id, CDAE_STREAM_INIT);
    m.resize(rows * K);
    for (size_t i = 0; i < rows * K; ++i) m[i] = cdae_init_uniform(key, i) * init_scale;
  };
  fill(o->W, o->I, 0); o->W_ag.assign(o->I * K, 0.0001);                     // :113-114
  if (o->c.asymmetric) { fill(o->V, o->I, 2); o->V_ag.assign(o->I * K, 0.0001); }   // :115-118
  else { o->V.clear(); o->V_ag.clear(); }
  if (o->c.user_factor) { fill(o->Wu, o->U, 4); o->Wu_ag.assign(o->U * K, 0.0001); }   // :119-122
  else { o->Wu.clear(); o->Wu_ag.clear(); }
  o->b.assign(K, 0.); o->b_ag.assign(K, 0.0001);                             // :123-124
  o->bp.assign(o->I, 0.); o->bp_ag.assign(o->I, 0.0001);                     // :125-126
  if (o->c.linear_function) { o->Uu.assign(o->U * K, 1.); o->Uu_ag.assign(o->U * K, 0.0001); }   // :130-133
  else { o->Uu.clear(); o->Uu_ag.clear(); }
}

// ---- reference-sequenced mode (see Oracle::GlibcRand) ----
void oracle_ref_seed(void* h, uint64_t mt_seed, uint32_t rand_seed) { ((Oracle*)h)->ref_seed(mt_seed, rand_seed); }
// the items of every user in DATA order (what decides the reference's hashtable iteration order); NULL = ascending
void oracle_ref_set_insertion_order(void* h, const uint32_t* items) {
  Oracle* o = (Oracle*)h;
  if (items) o->ref_insertion.assign(items, items + o->col.size()); else o->ref_insertion.clear();
}
// reset() with Eigen's Random = rand(): W, [V], [Wu] in that order (cdae.hpp:113,116,120), row-major
void oracle_ref_init_params(void* h) {
  Oracle* o = (Oracle*)h;
  oracle_init_params(h, 0);                                                  // sizes, accumulators, biases
  const double init_scale = 4. * std::sqrt(6. / (double)(o->I + o->K));      // :112
  auto fill = [&](std::vector<double>& m) {
    for (double& v : m) v = (-1. + 2. * (double)o->ref_rand.next() / 2147483647.) * init_scale;
  };
  fill(o->W);
  if (o->c.asymmetric) fill(o->V);
  if (o->c.user_factor) fill(o->Wu);
}
void oracle_train_users_reference_sequenced(void* h, uint64_t u0, uint64_t u1) { ((Oracle*)h)->train_users_reference_sequenced(u0, u1); }
// advance the generators by one user-corruption and return what the step would have used (tests): capacities n_u, n_u, n_u * num_neg
void oracle_ref_draw_user(void* h, uint64_t uid, uint32_t* pos_seq, uint32_t* in_seq, uint64_t* n_in, uint32_t* neg) {
  Oracle* o = (Oracle*)h;
  std::unordered_map<size_t, double> item_set;
  std::vector<uint32_t> p, i, n;
  o->ref_item_set(uid, item_set);
  o->ref_draw_user(item_set, p, i, n);
  std::copy(p.begin(), p.end(), pos_seq); std::copy(i.begin(), i.end(), in_seq); std::copy(n.begin(), n.end(), neg);
  *n_in = i.size();
}
void oracle_step_user_seq(void* h, uint64_t uid, const uint32_t* pos, uint64_t n_pos, const uint32_t* in, uint64_t n_in,
                          const uint32_t* neg, uint64_t n_neg) {
  ((Oracle*)h)->train_user_seq(uid, pos, n_pos, in, n_in, neg, n_neg, nullptr, nullptr, nullptr, nullptr);
}
uint32_t oracle_ref_rand(void* h) { return ((Oracle*)h)->ref_rand.next(); }
uint64_t oracle_ref_mt(void* h) { return ((Oracle*)h)->ref_mt(); }
double oracle_ref_uniform(void* h) { return ((Oracle*)h)->ref_uniform(); }

size_t oracle_param_size(void* h, uint32_t which) { auto* p = param((Oracle*)h, which); return p ? p->size() : 0; }
int oracle_get_param(void* h, uint32_t which, double* out, size_t n) {
  auto* p = param((Oracle*)h, which); if (!p || p->size() != n) return 1;
  std::memcpy(out, p->data(), n * sizeof(double)); return 0;
}
int oracle_set_param(void* h, uint32_t which, const double* in, size_t n) {
  auto* p = param((Oracle*)h, which); if (!p) return 1;
  p->assign(in, in + n); return 0;
}

void oracle_train_users_literal(void* h, uint64_t seed, uint32_t epoch, uint64_t u0, uint64_t u1) {
  ((Oracle*)h)->train_users_literal(seed, epoch, u0, u1);
}
void oracle_train_users_batched(void* h, uint64_t seed, uint32_t epoch, uint64_t u0, uint64_t u1, uint64_t B) {
  ((Oracle*)h)->train_users_batched(seed, epoch, u0, u1, B);
}
void oracle_train_users_full(void* h, uint64_t seed, uint32_t epoch, uint64_t u0, uint64_t u1, uint64_t B) {
  ((Oracle*)h)->train_users_full(seed, epoch, u0, u1, B);
}
// explicit-input single step with taps (known-answer fixtures)
void oracle_step_user(void* h, uint64_t uid, const uint32_t* in, uint64_t n_in, const uint32_t* neg,
                      uint64_t n_neg, double* z, double* y, double* g, double* hg) {
  ((Oracle*)h)->train_user_literal(uid, in, n_in, neg, n_neg, z, y, g, hg);
}
void oracle_draw_inputs(void* h, uint64_t seed, uint32_t epoch, uint64_t uid, uint32_t cidx, uint32_t stream,
                        uint32_t* out, uint64_t* n_out) {
  std::vector<uint32_t> in; ((Oracle*)h)->draw_inputs(seed, epoch, uid, cidx, stream, in);
  std::copy(in.begin(), in.end(), out); *n_out = in.size();
}
void oracle_draw_negatives(void* h, uint64_t seed, uint32_t epoch, uint64_t uid, uint32_t cidx, uint32_t* out) {
  std::vector<uint32_t> neg; ((Oracle*)h)->draw_negatives(seed, epoch, uid, cidx, neg);
  std::copy(neg.begin(), neg.end(), out);
}
// mode 0: inference form (full row, scale 1); mode 1: training corruption 0 with configured scale
void oracle_encode(void* h, uint64_t seed, uint32_t epoch, int mode, const uint32_t* uids, uint64_t n, double* Z) {
  Oracle* o = (Oracle*)h;
  std::vector<uint32_t> in;
  for (uint64_t i = 0; i < n; ++i) {
    size_t uid = uids[i];
    if (mode == 0) {
      const uint32_t* pos = &o->col[o->row_ptr[uid]];
      size_t n_pos = o->row_ptr[uid + 1] - o->row_ptr[uid];
      o->hidden(uid, pos, o->c.corruption_ratio != 1. ? n_pos : 0, 1.0, Z + i * o->K);
    } else {
      o->draw_inputs(seed, epoch, uid, 0, CDAE_STREAM_CORRUPT, in);
      o->hidden(uid, in.data(), in.size(), o->scale(), Z + i * o->K);
    }
  }
}
double oracle_data_loss(void* h, uint64_t seed, uint32_t epoch) { return ((Oracle*)h)->data_loss(seed, epoch); }
double oracle_penalty_loss(void* h) { return ((Oracle*)h)->penalty_loss(); }
double oracle_loss_eval(void* h, double pred, double truth) { return ((Oracle*)h)->loss_eval(pred, truth); }
double oracle_loss_grad(void* h, double pred, double truth) { return ((Oracle*)h)->loss_grad(pred, truth); }

void oracle_recommend(void* h, uint64_t u0, uint64_t u1, uint32_t topk, uint32_t* out, double* scores) {
  Oracle* o = (Oracle*)h;
  for (uint64_t u = u0; u < u1; ++u)
    o->recommend(u, topk, out + (u - u0) * topk, scores ? scores + (u - u0) * topk : nullptr);
}

// TOPN_Evaluation::evaluate, evaluation.hpp:113-181: mean of evaluate_rec_list over users that have test items
void oracle_eval_topn(const uint32_t* rec, uint32_t topk, uint64_t U, const int64_t* test_ptr,
                      const uint32_t* test_col, double* rets8) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double n_test_users = 0;
  for (uint64_t u = 0; u < U; ++u) if (test_ptr[u + 1] > test_ptr[u]) n_test_users += 1.;   // :160
  for (uint64_t u = 0; u < U; ++u) {
    size_t nt = test_ptr[u + 1] - test_ptr[u];
    if (nt == 0) continue;                                                   // :139-140
    double r[8];
    eval_rec_list(rec + u * topk, topk, test_col + test_ptr[u], nt, r);
    for (int i = 0; i < 8; ++i) acc[i] += r[i] / n_test_users;               // :162-166
  }
  for (int i = 0; i < 8; ++i) rets8[i] = acc[i];
}
void oracle_eval_rec_list(const uint32_t* list, uint64_t n_list, const uint32_t* truth, uint64_t n_truth, double* rets8) {
  eval_rec_list(list, n_list, truth, n_truth, rets8);
}

// Heap<pair<size_t,double>> with sort_by_second_desc, as exercised by test/heap_test.hpp
struct OHeap { std::vector<std::pair<size_t, double>> d; };
static bool ocomp(const std::pair<size_t, double>& a, const std::pair<size_t, double>& b) { return a.second > b.second; }
void* oracle_heap_create() { return new OHeap(); }
void oracle_heap_destroy(void* p) { delete (OHeap*)p; }
void oracle_heap_push(void* p, uint64_t id, double v) { auto* h = (OHeap*)p; h->d.emplace_back(id, v); std::push_heap(h->d.begin(), h->d.end(), ocomp); }
void oracle_heap_push_and_pop(void* p, uint64_t id, double v) {            // heap.hpp:44-52
  auto* h = (OHeap*)p; std::pair<size_t, double> t(id, v);
  if (ocomp(t, h->d.front())) { std::pop_heap(h->d.begin(), h->d.end(), ocomp); h->d.pop_back(); h->d.push_back(t); std::push_heap(h->d.begin(), h->d.end(), ocomp); }
}
uint64_t oracle_heap_size(void* p) { return ((OHeap*)p)->d.size(); }
uint64_t oracle_heap_front(void* p) { return ((OHeap*)p)->d.front().first; }
uint64_t oracle_heap_pop(void* p) {                                         // heap.hpp:34-42
  auto* h = (OHeap*)p; std::pop_heap(h->d.begin(), h->d.end(), ocomp);
  uint64_t r = h->d.back().first; h->d.pop_back(); return r;
}
void oracle_heap_sorted(void* p, uint64_t* ids, double* vals) {             // heap.hpp:66-69
  auto* h = (OHeap*)p; auto c = h->d; std::sort_heap(c.begin(), c.end(), ocomp);
  for (size_t i = 0; i < c.size(); ++i) { ids[i] = c[i].first; vals[i] = c[i].second; }
}

}  // extern "C"
