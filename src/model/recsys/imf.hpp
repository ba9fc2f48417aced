// libcf::IMF (implicit-feedback matrix factorisation) on MI355X: the reference's class (src/model/recsys/imf.hpp:12-143) with
// the same config struct, constructor and public methods, forwarding to libcdae_hip.so (cdae_hip_create_mf; kernels in
// cdae_amd/csrc/cdae_mf_kernels.hpp) — SURVEY.md §8(f) rank 4, selected by apps/yelp with --method=MF (yelp.cpp:122-142).
//
//   method (reference lines)                         -> C ABI
//   reset (57-69)                                     cdae_hip_create_mf, _set_interactions, _init_params
//   train_one_iteration (71-86) + train_one_instance  cdae_hip_train_epoch          (default batch_users = 1: the reference's strictly sequential loop, one launch window
//                                                                                    per 256 users.  CDAE_BATCH_USERS=0 opts into the library's block default — 16 users
//                                                                                    (BPR: 8) on BASELINE-sized data sets, certified at ONE shape and hyper-parameter set
//                                                                                    only, DESIGN.md §8b — larger values are throughput settings)
//   predict_user_item_rating (117-119)                host dot product over parameters fetched once (cdae_hip_get_param)
//   recommend (RecsysModelBase, 77-104)               cdae_hip_recommend_all in pre_recommend, then table reads
//   get_user_vecs / get_item_vecs (121-127)           cdae_hip_get_param
// current_loss is 0, as in the reference (IMF does not override ModelBase::data_loss / penalty_loss).
#ifndef CDAE_HOST_MODEL_RECSYS_IMF_HPP_
#define CDAE_HOST_MODEL_RECSYS_IMF_HPP_

#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include <cdae_hip.h>

#include <base/mat.hpp>
#include <base/random.hpp>
#include <model/recsys/recsys_model_base.hpp>

#ifndef CDAE_HIP_CHECK
#define CDAE_HIP_CHECK(call) CHECK_EQ((call), 0) << "libcdae_hip: " << cdae_hip_last_error() << " "
#endif

namespace libcf {

struct IMFConfig {
  IMFConfig() = default;
  double learn_rate = 0.1;
  double beta = 1.;
  double lambda = 0.01;
  LossType lt = SQUARE;
  PenaltyType pt = L2;
  size_t num_dim = 10;
  size_t num_neg = 5;
  bool using_bias_term = true;
  bool using_adagrad = true;
};

class IMF : public RecsysModelBase {
 public:
  explicit IMF(const IMFConfig& mcfg) { configure(mcfg, false, "IMF"); }
  IMF() = default;

  virtual void reset(const Data& data_set) {
    ModelBase::reset(data_set);
    num_users_ = data_->feature_group_total_dimension(0);
    num_items_ = data_->feature_group_total_dimension(1);
    cdae_mf_config c = cdae_mf_config();
    c.struct_size = sizeof(c);
    c.num_dim = static_cast<uint32_t>(num_dim_); c.num_neg = static_cast<uint32_t>(num_neg_);
    c.loss_type = static_cast<uint32_t>(lt_);                  // the C ABI rejects losses the reference's app does not offer here
    c.using_adagrad = using_adagrad_; c.using_bias_term = using_bias_term_; c.pairwise = pairwise_;
    // the drop-in default IS the reference loop (imf.hpp:71-115, bpr.hpp:56-106): a block schedule changes the trajectory and is
    // certified at one shape only, so it has to be asked for (CDAE_BATCH_USERS=0: the library's block default; N > 1: N users)
    c.batch_users = static_cast<uint32_t>(mf_env_u64("CDAE_BATCH_USERS", 1));
    c.lambda = lambda_; c.learn_rate = learn_rate_; c.beta = beta_;
    cdae_hip_t* raw = nullptr;
    CDAE_HIP_CHECK(cdae_hip_create_mf(&c, static_cast<int>(mf_env_u64("CDAE_DEVICE", 0)), &raw));
    dev_.reset(raw, [](cdae_hip_t* h) { cdae_hip_destroy(h); });
    auto csr = std::make_shared<Csr>();
    data_->to_csr(0, 1, csr->row_ptr, csr->col);
    CDAE_HIP_CHECK(cdae_hip_set_interactions(raw, num_users_, num_items_, csr->row_ptr.data(), csr->col.data()));
    csr_ = csr;
    train_generation_ = data_->generation();
    test_generation_ = std::make_shared<uint64_t>(0);          // a new handle holds no validation rows
    seed_ = std::getenv("CDAE_SEED") ? mf_env_u64("CDAE_SEED", 0) : Random::next_u64();
    CDAE_HIP_CHECK(cdae_hip_init_params(raw, seed_));
    epoch_ = 0;
    invalidate();
  }

  virtual void train_one_iteration(const Data&) {
    CHECK(dev_ != nullptr) << "reset() must be called first";
    cdae_hip_stats st;
    CDAE_HIP_CHECK(cdae_hip_train_epoch(dev_.get(), seed_, epoch_++, &st));
    LOG(INFO) << (pairwise_ ? "BPR" : "IMF") << " epoch " << epoch_ << ": " << st.users << " users in " << st.wall_seconds << " s ("
              << static_cast<double>(st.users) / st.wall_seconds << " users/s, " << st.batches << " blocks)";
    invalidate();
  }

  // imf.hpp:117-119
  double predict_user_item_rating(size_t uid, size_t iid) const {
    CHECK(dev_ != nullptr);
    CHECK_LT(uid, num_users_); CHECK_LT(iid, num_items_);
    // the four parameter arrays come from the device ONCE per training epoch (a loop over (uid, iid) pairs would otherwise move
    // the whole model over PCIe per call); train_one_iteration / reset drop the copy
    return score(*host_params(), uid, iid);
  }
  DMatrix get_user_vecs() { return matrix(CDAE_P_WU, num_users_); }      // imf.hpp:121-123
  DMatrix get_item_vecs() { return matrix(CDAE_P_W, num_items_); }       // imf.hpp:125-127

  // all users are scored and top-k'd on the GPU once; recommend() reads the table (evaluation.hpp:135-149)
  void pre_recommend() { ensure_table(10); }
  std::vector<size_t> recommend_train_row(size_t uid, size_t topk) const {
    CHECK_LT(uid, num_users_);
    std::shared_ptr<const Table> t = ensure_table(topk);
    return std::vector<size_t>(t->ids.begin() + uid * topk, t->ids.begin() + (uid + 1) * topk);
  }
  bool trained_on(const Data& d) const { return train_generation_ != 0 && d.generation() == train_generation_; }
  // RecsysModelBase::recommend(uid, topk, rated) excludes exactly `rated` (recsys_model_base.hpp:77-104).  When that is the user's
  // own train row — what Evaluation passes — the GPU table answers; any other set takes the reference's generic scan + heap over
  // predict_user_item_rating (host copy of the parameters, fetched once per epoch).
  std::vector<size_t> recommend(size_t uid, size_t topk, const std::unordered_map<size_t, double>& rated) const {
    CHECK_LT(uid, num_users_);
    if (is_train_row(uid, rated)) return recommend_train_row(uid, topk);
    // the generic scan (recsys_model_base.hpp:77-104) with the parameters fetched ONCE for the call: predict_user_item_rating takes
    // the model's lock and a shared_ptr copy per (user, item) pair, which serialised the evaluation threads on one mutex
    const std::shared_ptr<const HostParams> p = host_params();
    typedef std::pair<size_t, double> P;
    Heap<P> heap(sort_by_second_desc<size_t, double>, topk);
    for (size_t item = 0; item < num_items_; ++item) {
      if (rated.count(item)) continue;
      const P cand(item, score(*p, uid, item));
      if (heap.size() < topk) heap.push(cand); else heap.push_and_pop(cand);
    }
    CHECK_EQ(heap.size(), topk);
    const std::vector<P> sorted = heap.get_sorted_data();
    std::vector<size_t> out(topk);
    for (size_t i = 0; i < topk; ++i) out[i] = sorted[i].first;
    return out;
  }
  // TOPN_Evaluation on the device (evaluation.hpp:113-219 -> cdae_hip_eval_topn): the validation rows go over once per data set
  // (`generation` identifies their contents), the eight means come back; nothing else crosses PCIe
  bool eval_topn_device(uint64_t generation, const std::vector<int64_t>& val_ptr, const std::vector<uint32_t>& val_col, size_t topk,
                        double* rets8) const {
    std::lock_guard<std::mutex> lk(*mu_);
    CHECK(dev_ != nullptr) << "reset() must be called first";
    if (*test_generation_ != generation || generation == 0) {
      CHECK_EQ(val_ptr.size(), num_users_ + 1);
      CDAE_HIP_CHECK(cdae_hip_set_test_rows(dev_.get(), val_ptr.data(), val_col.data()));
      *test_generation_ = generation;
    }
    CDAE_HIP_CHECK(cdae_hip_eval_topn(dev_.get(), static_cast<uint32_t>(topk), rets8, nullptr, nullptr));
    return true;
  }

 protected:
  void configure(const IMFConfig& mcfg, bool pairwise, const char* name) {
    learn_rate_ = mcfg.learn_rate; beta_ = mcfg.beta; lambda_ = mcfg.lambda; num_dim_ = mcfg.num_dim; num_neg_ = mcfg.num_neg;
    using_bias_term_ = mcfg.using_bias_term; using_adagrad_ = mcfg.using_adagrad; lt_ = mcfg.lt; pairwise_ = pairwise;
    loss_ = Loss::create(mcfg.lt);
    penalty_ = Penalty::create(mcfg.pt);
    LOG(INFO) << name << " Model Configure (MI355X / HIP): \n"
              << "\t{lambda: " << lambda_ << "}, {Learn Rate: " << learn_rate_ << "}, {Beta: " << beta_ << "}, {Loss: " << loss_->loss_type()
              << "}, {Penalty: " << penalty_->penalty_type() << "}\n"
              << "\t{Dim: " << num_dim_ << "}, {BiasTerm: " << using_bias_term_ << "}, {Using AdaGrad: " << using_adagrad_
              << "}, {Num Negative: " << num_neg_ << "}";
  }
  struct Csr { std::vector<int64_t> row_ptr; std::vector<uint32_t> col; };
  struct HostParams { std::vector<float> u, v, ub, ib; };
  double score(const HostParams& p, size_t uid, size_t iid) const {           // imf.hpp:117-119
    double s = p.ub[uid] + p.ib[iid];
    for (size_t k = 0; k < num_dim_; ++k) s += static_cast<double>(p.u[uid * num_dim_ + k]) * p.v[iid * num_dim_ + k];
    return s;
  }
  bool is_train_row(size_t uid, const std::unordered_map<size_t, double>& rated) const {
    if (!csr_) return false;
    const int64_t a = csr_->row_ptr[uid], b = csr_->row_ptr[uid + 1];
    if (static_cast<size_t>(b - a) != rated.size()) return false;
    for (int64_t p = a; p < b; ++p) if (!rated.count(csr_->col[p])) return false;
    return true;
  }
  std::shared_ptr<const HostParams> host_params() const {
    std::lock_guard<std::mutex> lk(*mu_);
    if (!host_) {
      auto p = std::make_shared<HostParams>();
      p->u.resize(num_users_ * num_dim_); p->v.resize(num_items_ * num_dim_); p->ub.resize(num_users_); p->ib.resize(num_items_);
      fetch(CDAE_P_WU, p->u); fetch(CDAE_P_W, p->v); fetch(CDAE_P_UB, p->ub); fetch(CDAE_P_BP, p->ib);
      host_ = p;
    }
    return host_;
  }
  void invalidate() {
    std::lock_guard<std::mutex> lk(*mu_);
    rec_.reset();
    host_.reset();
  }
  struct Table { size_t topk; std::vector<uint32_t> ids; };
  std::shared_ptr<const Table> ensure_table(size_t topk) const {
    std::lock_guard<std::mutex> lk(*mu_);
    if (!rec_ || rec_->topk != topk) {
      CHECK(dev_ != nullptr) << "reset() must be called first";
      auto t = std::make_shared<Table>();
      t->topk = topk;
      t->ids.resize(num_users_ * topk);
      CDAE_HIP_CHECK(cdae_hip_recommend_all(dev_.get(), 0, num_users_, static_cast<uint32_t>(topk), t->ids.data()));
      rec_ = t;
    }
    return rec_;
  }
  void fetch(uint32_t which, std::vector<float>& out) const { CDAE_HIP_CHECK(cdae_hip_get_param(dev_.get(), which, out.data(), out.size())); }
  DMatrix matrix(uint32_t which, size_t rows) const {
    std::vector<float> v(rows * num_dim_);
    fetch(which, v);
    DMatrix m(rows, num_dim_);
    for (size_t i = 0; i < v.size(); ++i) m.data()[i] = v[i];
    return m;
  }
  static uint64_t mf_env_u64(const char* name, uint64_t dflt) {
    const char* v = std::getenv(name);
    return v ? std::strtoull(v, nullptr, 10) : dflt;
  }

  double learn_rate_ = 0.1, beta_ = 1., lambda_ = 0.01;
  size_t num_dim_ = 10, num_neg_ = 5;
  bool using_bias_term_ = true, using_adagrad_ = true, pairwise_ = false;
  LossType lt_ = SQUARE;
  std::shared_ptr<cdae_hip_t> dev_;                  // shared by copies: Solver copies the model (solver.hpp:17)
  std::shared_ptr<std::mutex> mu_ = std::make_shared<std::mutex>();
  mutable std::shared_ptr<const Table> rec_;
  mutable std::shared_ptr<const HostParams> host_;   // parameters on the host for predict_user_item_rating; dropped when they change
  std::shared_ptr<const Csr> csr_;                   // the train rows the handle was reset with
  uint64_t train_generation_ = 0;
  std::shared_ptr<uint64_t> test_generation_ = std::make_shared<uint64_t>(0);   // Data::generation() of the validation rows the handle holds (shared with copies, like dev_)
  uint64_t seed_ = 0;
  uint32_t epoch_ = 0;
};

}  // namespace libcf
#endif
