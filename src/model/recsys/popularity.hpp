// Popularity baseline: recommend the most-rated unrated items.  apps/yelp always runs it first
// (yelp.cpp:109-113).  Reference: src/model/recsys/popularity.hpp.  CPU, not part of the GPU hot path.
#ifndef CDAE_HOST_MODEL_RECSYS_POPULARITY_HPP_
#define CDAE_HOST_MODEL_RECSYS_POPULARITY_HPP_

#include <algorithm>
#include <model/recsys/recsys_model_base.hpp>

namespace libcf {

class Popularity : public RecsysModelBase {
 public:
  Popularity() { LOG(INFO) << "Popularity Model"; }
  void reset(const Data& data_set) {
    RecsysModelBase::reset(data_set);
    std::vector<std::pair<size_t, double>> cnt(num_items_);
    for (size_t i = 0; i < num_items_; ++i) cnt[i] = std::make_pair(i, 0.);
    for (auto it = data_set.begin(); it != data_set.end(); ++it) cnt[it->get_feature_group_index(1, 0)].second += 1.;
    std::stable_sort(cnt.begin(), cnt.end(), sort_by_second_desc<size_t, double>);
    ranked_.resize(num_items_);
    for (size_t i = 0; i < num_items_; ++i) ranked_[i] = cnt[i].first;
  }
  void train_one_iteration(const Data&) {}
  std::vector<size_t> recommend(size_t, size_t topk, const std::unordered_map<size_t, double>& rated) const {
    std::vector<size_t> out;
    out.reserve(topk);
    for (size_t item : ranked_) {
      if (rated.count(item)) continue;
      out.push_back(item);
      if (out.size() == topk) break;
    }
    CHECK_EQ(out.size(), topk);
    return out;
  }
 private:
  std::vector<size_t> ranked_;
};

}  // namespace libcf
#endif
