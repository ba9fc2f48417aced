// libcf::CDAE on MI355X: the reference's model class (src/model/recsys/cdae.hpp:13-31 CDAEConfig, :36-454
// CDAE) re-provided with the same names, constructor, public methods and abort-on-error behaviour, so that
// Solver<CDAE>, Evaluation<CDAE> and apps/yelp/yelp.cpp:168-199 compile and run unchanged — but every method of
// the hot path forwards to one entry point of libcdae_hip.so (include/cdae_hip.h), where the per-user
// forward/backward and the AdaGrad/SGD update run as HIP kernels on gfx950.  No CPU fallback: without the
// library or a GPU the first call CHECK-fails.
//
//   method (reference lines)                         -> C ABI
//   reset (109-134, recsys_model_base.hpp:29-34)      cdae_hip_create, _set_interactions, _init_params
//   train_one_iteration (136-146)                     cdae_hip_train_epoch
//   train_one_user_corruption (198-358)               cdae_hip_train_one_user_corruption
//   data_loss (78-101) / penalty_loss (103-107)       cdae_hip_data_loss / cdae_hip_penalty_loss
//   pre_recommend + recommend (162-196)               cdae_hip_recommend_all, then lock-free table reads
//   TOPN_Evaluation::evaluate (evaluation.hpp:113-219) cdae_hip_set_test_rows + cdae_hip_eval_topn (eval_topn_device below)
//   get_user_representations (148-159)                cdae_hip_encode
//
// Environment knobs (not in the reference): CDAE_BATCH_USERS (users per parameter snapshot; 1 = the
// reference's strictly sequential schedule), CDAE_FULL_OUTPUT (1: every unrated item is a negative — dense MFMA
// decode; num_neg ignored), CDAE_SEED (fixes the counter-stream seed; default: one draw of
// libcf::Random, i.e. time-seeded like yelp.cpp:107), CDAE_DEVICE (HIP device index), CDAE_DEVICES (comma list, e.g.
// 0,1,2,3: Solver<CDAE>::train runs data-parallel over these GPUs through cdae_hip_multi_* — users sharded, shared
// parameters exchanged by RCCL inside the library; a repeated id, e.g. 0,0, makes logical shards of one GPU),
// CDAE_EXCHANGE_EVERY (0 = synchronous exchange at every step, the default; k = pipelined every k steps),
// CDAE_LAYOUT (with CDAE_DEVICES): item_rows — THE DEFAULT — the shards cut the ITEM rows, every shard sees every user, the user node
// is sharded by user: the exact single-GPU schedule (sampled decode, or with CDAE_FULL_OUTPUT=1 the full-output one, BASELINE
// configs[4]'s layout), two small all-reduces per batch, no accuracy cost; users — user shards + exchange of the shared parameters'
// deltas (the north star's partitioning), taken only when asked for by name.  Its schedule (cdae_hip_multi_set_schedule):
// CDAE_RELAY_EPOCHS (default 1: the first epoch on the single-GPU schedule, handed from shard to shard — the warm-up the exchanged steps
// need), CDAE_SYNC_BATCH_USERS (users per shard of an exchanged step, default 64), CDAE_COMBINE (global_acc — default — or sum).  Measured
// at ML-10M / Netflix shape on 8 shards: mean-over-seeds Recall@10 within +-0.002 of the sequential reference after the relayed epoch,
// single seeds up to 0.009 / 0.005 — NOT the single-GPU bounds (0.0015 / 0.005).
#ifndef CDAE_HOST_MODEL_RECSYS_CDAE_HPP_
#define CDAE_HOST_MODEL_RECSYS_CDAE_HPP_

#include <cstdlib>
#include <memory>
#include <string>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <cdae_hip.h>

#include <base/data.hpp>
#include <base/instance.hpp>
#include <base/mat.hpp>
#include <base/parallel.hpp>
#include <base/random.hpp>
#include <model/recsys/recsys_model_base.hpp>

#define CDAE_HIP_CHECK(call) CHECK_EQ((call), 0) << "libcdae_hip: " << cdae_hip_last_error() << " "

namespace libcf {

struct CDAEConfig {
  CDAEConfig() = default;
  double lambda = 0.01;
  double learn_rate = 0.1;
  LossType lt = LOGISTIC;
  PenaltyType pt = L2;
  size_t num_dim = 10;
  bool using_adagrad = true;
  double corruption_ratio = 0.5;
  size_t num_corruptions = 1;
  bool asymmetric = false;
  bool user_factor = true;
  bool linear = false;
  size_t num_neg = 5;
  bool scaled = true;
  double beta = 0.;
  bool linear_function = false;
  bool tanh = false;
};

class CDAE : public RecsysModelBase {
 public:
  CDAE(const CDAEConfig& mcfg) : cfg_(mcfg) {
    loss_ = Loss::create(mcfg.lt);
    penalty_ = Penalty::create(mcfg.pt);
    LOG(INFO) << "CDAE Configure (MI355X / HIP): \n"
              << "\t{lambda: " << cfg_.lambda << "}, {Loss: " << loss_->loss_type() << "}, {Penalty: " << penalty_->penalty_type() << "}\n"
              << "\t{Dim: " << cfg_.num_dim << "}, {LearnRate: " << cfg_.learn_rate << "}, {Using AdaGrad: " << cfg_.using_adagrad << "}\n"
              << "\t{Corruption Ratio: " << cfg_.corruption_ratio << "}, {Num Corruptions: " << cfg_.num_corruptions
              << "}, {Asymmetric: " << cfg_.asymmetric << "}\n"
              << "\t{UserFactor: " << cfg_.user_factor << "}, {Linear: " << cfg_.linear << "}, {Num Negative: " << cfg_.num_neg
              << "}, {Scaled: " << cfg_.scaled << "}\n"
              << "\t{Beta: " << cfg_.beta << "}, {LinearFunction: " << cfg_.linear_function << "}, {tanh: " << cfg_.tanh << "}";
  }
  CDAE() : CDAE(CDAEConfig()) {}

  // ---- reset: cdae.hpp:109-134 -------------------------------------------------------------------------
  void reset(const Data& data_set) {
    ModelBase::reset(data_set);
    num_users_ = data_->feature_group_total_dimension(0);
    num_items_ = data_->feature_group_total_dimension(1);
    CHECK(cfg_.pt == L2) << "CDAE's gradient hard-codes the L2 term (cdae.hpp:231)";
    cdae_hip_config c = cdae_hip_config();
    c.struct_size = sizeof(c);
    c.num_dim = static_cast<uint32_t>(cfg_.num_dim);
    c.num_neg = static_cast<uint32_t>(cfg_.num_neg);
    c.num_corruptions = static_cast<uint32_t>(cfg_.num_corruptions);
    c.loss_type = static_cast<uint32_t>(cfg_.lt);          // the C ABI rejects losses CDAE cannot train with
    c.using_adagrad = cfg_.using_adagrad; c.asymmetric = cfg_.asymmetric; c.user_factor = cfg_.user_factor;
    c.linear = cfg_.linear; c.scaled = cfg_.scaled; c.tanh_act = cfg_.tanh;
    c.batch_users = static_cast<uint32_t>(env_u64("CDAE_BATCH_USERS", 0));
    c.full_output = static_cast<uint32_t>(env_u64("CDAE_FULL_OUTPUT", 0));   // north-star extension, not in CDAEConfig
    c.linear_function = cfg_.linear_function;
    c.lambda = cfg_.lambda; c.learn_rate = cfg_.learn_rate; c.corruption_ratio = cfg_.corruption_ratio; c.beta = cfg_.beta;
    std::shared_ptr<Csr> csr = std::make_shared<Csr>();
    data_->to_csr(0, 1, csr->row_ptr, csr->col);           // uid -> sorted {iid}; labels are all 1 (yelp.cpp:66)
    train_csr_ = csr;
    train_generation_ = data_->generation();
    // what the device is about to be given, in a form a test can pin against its own derivation of the same rows
    LOG(INFO) << "CDAE train rows: " << num_users_ << " users x " << num_items_ << " items, " << csr->col.size()
              << " interactions, csr fnv1a64 " << csr_checksum(*csr);
    seed_ = std::getenv("CDAE_SEED") ? env_u64("CDAE_SEED", 0) : Random::next_u64();
    const std::vector<int> devices = env_devices();
    dev_.reset(); multi_.reset();
    if (devices.size() > 1) {                              // data-parallel over several shards (cdae_hip_multi_*)
      cdae_hip_multi_t* raw = nullptr;
      CDAE_HIP_CHECK(cdae_hip_multi_create(&c, static_cast<int>(devices.size()), devices.data(), &raw));
      multi_.reset(raw, [](cdae_hip_multi_t* m) { cdae_hip_multi_destroy(m); });
      // the certified schedule is the default: item rows (exact single-GPU schedule); "users" selects the delta exchange by name
      const char* layout = std::getenv("CDAE_LAYOUT");
      CHECK(!layout || std::string(layout) == "item_rows" || std::string(layout) == "users") << "CDAE_LAYOUT must be item_rows or users";
      item_rows_ = !(layout && std::string(layout) == "users");
      if (item_rows_) {
        CDAE_HIP_CHECK(cdae_hip_multi_set_layout(raw, CDAE_LAYOUT_ITEM_ROWS));
      } else {
        cdae_multi_schedule sc = cdae_multi_schedule();
        sc.period = static_cast<int32_t>(env_u64("CDAE_EXCHANGE_EVERY", 0));
        const char* comb = std::getenv("CDAE_COMBINE");
        CHECK(!comb || std::string(comb) == "sum" || std::string(comb) == "global_acc") << "CDAE_COMBINE must be sum or global_acc";
        sc.combine = comb && std::string(comb) == "sum" ? CDAE_COMBINE_SUM : CDAE_COMBINE_GLOBAL_ACC;
        sc.sync_batch_users = static_cast<uint32_t>(env_u64("CDAE_SYNC_BATCH_USERS", 64));
        sc.relay_epochs = std::getenv("CDAE_RELAY_EPOCHS") ? std::atof(std::getenv("CDAE_RELAY_EPOCHS")) : 1.0;
        CDAE_HIP_CHECK(cdae_hip_multi_set_schedule(raw, &sc));
      }
      CDAE_HIP_CHECK(cdae_hip_multi_set_interactions(raw, num_users_, num_items_, csr->row_ptr.data(), csr->col.data()));
      CDAE_HIP_CHECK(cdae_hip_multi_init_params(raw, seed_));
      if (item_rows_) LOG(INFO) << "CDAE: " << devices.size() << " item-row shards (CDAE_DEVICES; CDAE_LAYOUT=item_rows is the default): the single-GPU schedule over item shards";
      // The sampled decode does NOT get faster in this layout: its two serial chains do not shorten and every batch adds two latency-bound
      // all-reduces — measured 2 x slower and worse than ONE GPU at the BASELINE shapes.  It is the default because it
      // is the single-GPU schedule exactly; say so where the user sees it.
      if (item_rows_ && !c.full_output)
        LOG(WARNING) << "CDAE: CDAE_DEVICES with the SAMPLED decode in the item-rows layout trains the single-GPU schedule exactly but SLOWER than one GPU "
                        "(measured 1.3-1.6 M users/s on 8 GPUs against 2.8 M on one at ML-10M shape, DESIGN.md section 7b).  For throughput use one GPU "
                        "(unset CDAE_DEVICES), or CDAE_LAYOUT=users (relay epoch + exchanged steps: its own, wider accuracy bounds, DESIGN.md section 7); "
                        "the item-rows layout pays for the full-output decode (CDAE_FULL_OUTPUT=1).";
      else LOG(INFO) << "CDAE: " << devices.size() << " user shards (CDAE_DEVICES, CDAE_LAYOUT=users: relay warm-up + exchanged steps, DESIGN.md section 7), exchange every "
                     << env_u64("CDAE_EXCHANGE_EVERY", 0) << " steps";
    } else {
      cdae_hip_t* raw = nullptr;
      CDAE_HIP_CHECK(cdae_hip_create(&c, devices.empty() ? static_cast<int>(env_u64("CDAE_DEVICE", 0)) : devices[0], &raw));
      dev_.reset(raw, [](cdae_hip_t* h) { cdae_hip_destroy(h); });
      CDAE_HIP_CHECK(cdae_hip_set_interactions(raw, num_users_, num_items_, csr->row_ptr.data(), csr->col.data()));
      CDAE_HIP_CHECK(cdae_hip_init_params(raw, seed_));
    }
    epoch_ = 0;
    rec_.reset();
    test_generation_ = std::make_shared<uint64_t>(0);          // a new handle holds no validation rows
  }

  // ---- training: cdae.hpp:136-146 -----------------------------------------------------------------------
  void train_one_iteration(const Data&) {
    CHECK(ready()) << "reset() must be called first";
    cdae_hip_stats st;
    if (multi_) CDAE_HIP_CHECK(cdae_hip_multi_train_epoch(multi_.get(), seed_, epoch_++, &st));
    else CDAE_HIP_CHECK(cdae_hip_train_epoch(dev_.get(), seed_, epoch_++, &st));
    LOG(INFO) << "CDAE epoch " << epoch_ << ": " << st.users << " users in " << st.wall_seconds << " s ("
              << static_cast<double>(st.users) / st.wall_seconds << " users/s, " << st.batches << " batches)";
    rec_.reset();
  }

  // cdae.hpp:198-358 with the caller's corrupted input set; negatives drawn like cdae.hpp:217-220
  void train_one_user_corruption(size_t uid, const std::unordered_map<size_t, double>& input_set,
                                 const std::unordered_map<size_t, double>& output_set) {
    CHECK(ready()) << "reset() must be called first";
    // the device decodes the user's TRAIN row as the positives (that is what train_one_iteration passes, cdae.hpp:143)
    CHECK(is_train_row(uid, output_set)) << "train_one_user_corruption: output_set must be the user's train row";
    std::vector<uint32_t> in, neg(output_set.size() * cfg_.num_neg);
    for (auto& p : input_set) in.push_back(static_cast<uint32_t>(p.first));
    for (auto& n : neg) n = static_cast<uint32_t>(sample_negative_item(output_set));
    CHECK(!multi_) << "train_one_user_corruption steps ONE replica's shared parameters: not available with CDAE_DEVICES";
    CDAE_HIP_CHECK(cdae_hip_train_one_user_corruption(dev_.get(), uid, in.data(), in.size(), neg.data(), neg.size()));
    rec_.reset();
  }

  // ---- reported loss: cdae.hpp:78-107 -------------------------------------------------------------------
  double data_loss(const Data&, size_t = 0) const {
    CHECK(ready()) << "reset() must be called first";
    double v = 0;
    if (multi_) CDAE_HIP_CHECK(cdae_hip_multi_data_loss(multi_.get(), seed_, epoch_, &v));
    else CDAE_HIP_CHECK(cdae_hip_data_loss(dev_.get(), seed_, epoch_, &v));
    return v;
  }
  double penalty_loss() const {
    double v = 0;
    if (multi_) CDAE_HIP_CHECK(cdae_hip_multi_penalty_loss(multi_.get(), &v));
    else CDAE_HIP_CHECK(cdae_hip_penalty_loss(dev_.get(), &v));
    return v;
  }

  // cdae.hpp:148-159
  DMatrix get_user_representations() {
    std::vector<uint32_t> uids(num_users_);
    for (size_t u = 0; u < num_users_; ++u) uids[u] = static_cast<uint32_t>(u);
    std::vector<float> z(num_users_ * cfg_.num_dim);
    if (multi_) {
      CHECK(!item_rows_) << "get_user_representations is not provided in the item-rows layout";
      for (int s = 0; s < cdae_hip_multi_num_shards(multi_.get()); ++s) {       // every shard encodes its own users
        cdae_hip_t* h = nullptr; uint64_t a = 0, b = 0;
        CDAE_HIP_CHECK(cdae_hip_multi_shard(multi_.get(), s, &h, &a, &b));
        std::vector<uint32_t> local(b - a);
        for (uint64_t u = a; u < b; ++u) local[u - a] = static_cast<uint32_t>(u - a);
        CDAE_HIP_CHECK(cdae_hip_encode(h, seed_, epoch_, 0, local.data(), local.size(), z.data() + a * cfg_.num_dim));
      }
    } else {
      CDAE_HIP_CHECK(cdae_hip_encode(dev_.get(), seed_, epoch_, 0, uids.data(), uids.size(), z.data()));
    }
    DMatrix out(num_users_, cfg_.num_dim);
    for (size_t i = 0; i < z.size(); ++i) out.data()[i] = z[i];
    return out;
  }

  // ---- evaluation: cdae.hpp:162-196 ---------------------------------------------------------------------
  // All users are scored and top-k'd on the GPU once (evaluation.hpp:135 calls this before the thread pool
  // starts); recommend() is then a read of an immutable table and safe to call concurrently (evaluation.hpp:137).
  void pre_recommend() { ensure_table(10); }

  // The reference encodes the hidden layer FROM rated_item_set and excludes exactly that set (cdae.hpp:167-179).
  // Evaluation passes the user's train row, which is what the precomputed table holds; any other set takes the
  // explicit device path (cdae_hip_recommend_user), serialised on the handle's mutex.
  std::vector<size_t> recommend(size_t uid, size_t topk, const std::unordered_map<size_t, double>& rated_item_set) const {
    CHECK_LT(uid, num_users_);
    if (is_train_row(uid, rated_item_set)) return recommend_train_row(uid, topk);
    std::vector<uint32_t> rated;
    rated.reserve(rated_item_set.size());
    for (auto& p : rated_item_set) { CHECK_LT(p.first, num_items_); rated.push_back(static_cast<uint32_t>(p.first)); }
    std::vector<uint32_t> ids(topk);
    {
      std::lock_guard<std::mutex> lk(*mu_);
      CHECK(ready()) << "reset() must be called first";
      cdae_hip_t* h = dev_.get();
      uint64_t local = uid;
      CHECK(!(multi_ && item_rows_)) << "recommend() with a foreign rated set is not provided in the item-rows layout";
      if (multi_) {                                          // the shard that owns the user (its Wu row lives there)
        for (int s = 0; s < cdae_hip_multi_num_shards(multi_.get()); ++s) {
          uint64_t a = 0, b = 0;
          CDAE_HIP_CHECK(cdae_hip_multi_shard(multi_.get(), s, &h, &a, &b));
          if (uid >= a && uid < b) { local = uid - a; break; }
        }
      }
      CDAE_HIP_CHECK(cdae_hip_recommend_user(h, local, rated.data(), rated.size(), static_cast<uint32_t>(topk), ids.data()));
    }
    return std::vector<size_t>(ids.begin(), ids.end());
  }

  // recommend() for the user's own train row — what Evaluation asks for — without the caller building a hashtable
  // per user per epoch (evaluation.hpp:118-123): TOPN_Evaluation detects this method and uses it.
  std::vector<size_t> recommend_train_row(size_t uid, size_t topk) const {
    CHECK_LT(uid, num_users_);
    std::shared_ptr<const Table> t = ensure_table(topk);
    std::vector<size_t> out(topk);
    for (size_t i = 0; i < topk; ++i) out[i] = t->ids[uid * topk + i];
    return out;
  }
  // TOPN_Evaluation on the device (evaluation.hpp:113-219 -> cdae_hip_eval_topn): the validation rows go over once per data set
  // (`generation` identifies their contents), the eight means come back; the top-10 table never leaves the GPU.  Sharded models:
  // the lists meet on the host anyway (cdae_hip_multi_eval_topn sums there, in the same order).
  bool eval_topn_device(uint64_t generation, const std::vector<int64_t>& val_ptr, const std::vector<uint32_t>& val_col, size_t topk,
                        double* rets8) const {
    std::lock_guard<std::mutex> lk(*mu_);
    CHECK(ready()) << "reset() must be called first";
    CHECK_EQ(val_ptr.size(), num_users_ + 1);
    if (multi_) {
      CDAE_HIP_CHECK(cdae_hip_multi_eval_topn(multi_.get(), val_ptr.data(), val_col.data(), static_cast<uint32_t>(topk), rets8, nullptr, nullptr));
      return true;
    }
    if (*test_generation_ != generation || generation == 0) {
      CDAE_HIP_CHECK(cdae_hip_set_test_rows(dev_.get(), val_ptr.data(), val_col.data()));
      *test_generation_ = generation;
    }
    CDAE_HIP_CHECK(cdae_hip_eval_topn(dev_.get(), static_cast<uint32_t>(topk), rets8, nullptr, nullptr));
    return true;
  }
  // is `d` the data set this model was reset with?  (Evaluation: recommend_train_row answers for exactly those rows)
  bool trained_on(const Data& d) const { return train_generation_ != 0 && d.generation() == train_generation_; }
  // the train rows the model was reset with, as CSR (sorted item ids per user)
  const std::vector<int64_t>& train_row_ptr() const { CHECK(train_csr_ != nullptr); return train_csr_->row_ptr; }
  const std::vector<uint32_t>& train_col_idx() const { CHECK(train_csr_ != nullptr); return train_csr_->col; }

 private:
  struct Table { size_t topk; std::vector<uint32_t> ids; };
  struct Csr { std::vector<int64_t> row_ptr; std::vector<uint32_t> col; };

  bool is_train_row(size_t uid, const std::unordered_map<size_t, double>& rated) const {
    CHECK(train_csr_ != nullptr) << "reset() must be called first";
    const int64_t a = train_csr_->row_ptr[uid], b = train_csr_->row_ptr[uid + 1];
    if (static_cast<size_t>(b - a) != rated.size()) return false;
    for (int64_t p = a; p < b; ++p) if (!rated.count(train_csr_->col[p])) return false;
    return true;
  }

  std::shared_ptr<const Table> ensure_table(size_t topk) const {
    std::lock_guard<std::mutex> lk(*mu_);
    if (!rec_ || rec_->topk != topk) {
      CHECK(ready()) << "reset() must be called first";
      auto t = std::make_shared<Table>();
      t->topk = topk;
      t->ids.resize(num_users_ * topk);
      if (multi_) CDAE_HIP_CHECK(cdae_hip_multi_recommend_all(multi_.get(), 0, num_users_, static_cast<uint32_t>(topk), t->ids.data()));
      else CDAE_HIP_CHECK(cdae_hip_recommend_all(dev_.get(), 0, num_users_, static_cast<uint32_t>(topk), t->ids.data()));
      rec_ = t;
    }
    return rec_;
  }
  // FNV-1a (64 bit) over the little-endian bytes of row_ptr (int64) then col (uint32)
  static uint64_t csr_checksum(const Csr& c) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto eat = [&h](const void* p, size_t n) {
      const unsigned char* b = static_cast<const unsigned char*>(p);
      for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    };
    eat(c.row_ptr.data(), c.row_ptr.size() * sizeof(int64_t));
    eat(c.col.data(), c.col.size() * sizeof(uint32_t));
    return h;
  }
  bool ready() const { return dev_ != nullptr || multi_ != nullptr; }
  static std::vector<int> env_devices() {                    // CDAE_DEVICES=0,1,2,3
    std::vector<int> out;
    const char* v = std::getenv("CDAE_DEVICES");
    if (!v) return out;
    for (const char* p = v; *p;) {
      char* end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p) break;
      out.push_back(static_cast<int>(d));
      p = (*end == ',') ? end + 1 : end;
    }
    return out;
  }
  static uint64_t env_u64(const char* name, uint64_t dflt) {
    const char* v = std::getenv(name);
    return v ? std::strtoull(v, nullptr, 10) : dflt;
  }

  CDAEConfig cfg_;
  std::shared_ptr<cdae_hip_t> dev_;                  // shared by copies: Solver copies the model (solver.hpp:17)
  std::shared_ptr<cdae_hip_multi_t> multi_;          // instead of dev_ when CDAE_DEVICES names several shards
  bool item_rows_ = false;
  std::shared_ptr<std::mutex> mu_ = std::make_shared<std::mutex>();
  mutable std::shared_ptr<const Table> rec_;
  std::shared_ptr<const Csr> train_csr_;             // host copy of the train rows (recommend: is the caller's set the train row?)
  uint64_t train_generation_ = 0;                    // Data::generation() of the set reset() saw
  std::shared_ptr<uint64_t> test_generation_ = std::make_shared<uint64_t>(0);   // ... of the validation rows the handle holds (shared with copies, like dev_)
  uint64_t seed_ = 0;
  uint32_t epoch_ = 0;
};

}  // namespace libcf

#endif  // CDAE_HOST_MODEL_RECSYS_CDAE_HPP_
