// Popularity baseline: recommend the most-rated unrated items.  apps/yelp always runs it first
// (yelp.cpp:109-113).  Reference: src/model/recsys/popularity.hpp.  CPU, not part of the GPU hot path — but it runs in front
// of every CDAE run, so at Netflix scale it must not build the reference's uid -> {iid -> label} hashtables either: it keeps
// the train rows as one CSR (Data::to_csr) and answers Evaluation through recommend_train_row.
#ifndef CDAE_HOST_MODEL_RECSYS_POPULARITY_HPP_
#define CDAE_HOST_MODEL_RECSYS_POPULARITY_HPP_

#include <algorithm>
#include <memory>
#include <model/recsys/recsys_model_base.hpp>

namespace libcf {

class Popularity : public RecsysModelBase {
 public:
  Popularity() { LOG(INFO) << "Popularity Model"; }
  void reset(const Data& data_set) {
    ModelBase::reset(data_set);
    num_users_ = data_->feature_group_total_dimension(0);
    num_items_ = data_->feature_group_total_dimension(1);
    std::vector<std::pair<size_t, double>> cnt(num_items_);
    for (size_t i = 0; i < num_items_; ++i) cnt[i] = std::make_pair(i, 0.);
    for (uint32_t item : data_set.column(1)) cnt[item].second += 1.;
    std::stable_sort(cnt.begin(), cnt.end(), sort_by_second_desc<size_t, double>);
    ranked_.resize(num_items_);
    for (size_t i = 0; i < num_items_; ++i) ranked_[i] = cnt[i].first;
    auto csr = std::make_shared<Csr>();
    data_set.to_csr(0, 1, csr->row_ptr, csr->col);
    csr_ = csr;
    train_generation_ = data_set.generation();
  }
  void train_one_iteration(const Data&) {}
  std::vector<size_t> recommend(size_t, size_t topk, const std::unordered_map<size_t, double>& rated) const {
    return first_unrated(topk, [&](size_t item) { return rated.count(item) != 0; });
  }
  bool trained_on(const Data& d) const { return train_generation_ != 0 && d.generation() == train_generation_; }
  std::vector<size_t> recommend_train_row(size_t uid, size_t topk) const {
    CHECK(csr_ != nullptr); CHECK_LT(uid, num_users_);
    const uint32_t* a = csr_->col.data() + csr_->row_ptr[uid]; const uint32_t* b = csr_->col.data() + csr_->row_ptr[uid + 1];
    return first_unrated(topk, [&](size_t item) { return std::binary_search(a, b, static_cast<uint32_t>(item)); });
  }
 private:
  struct Csr { std::vector<int64_t> row_ptr; std::vector<uint32_t> col; };
  template <class Rated>
  std::vector<size_t> first_unrated(size_t topk, const Rated& rated) const {
    std::vector<size_t> out;
    out.reserve(topk);
    for (size_t item : ranked_) {
      if (rated(item)) continue;
      out.push_back(item);
      if (out.size() == topk) break;
    }
    CHECK_EQ(out.size(), topk);
    return out;
  }
  std::vector<size_t> ranked_;
  std::shared_ptr<const Csr> csr_;
  uint64_t train_generation_ = 0;
};

}  // namespace libcf
#endif
