// Base of the recommender models (reference: src/model/recsys/recsys_model_base.hpp:18-109): reset() builds
// the uid -> {iid -> label} table, sample_negative_item() is the reference's rand()-rejection sampler (kept
// for the CPU sibling models; the GPU CDAE samples on the device from include/cdae_rng.h), recommend() is
// the generic scan + top-k heap.
#ifndef CDAE_HOST_MODEL_RECSYS_MODEL_BASE_HPP_
#define CDAE_HOST_MODEL_RECSYS_MODEL_BASE_HPP_

#include <cstdlib>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <base/data.hpp>
#include <base/heap.hpp>
#include <base/mat.hpp>
#include <model/loss.hpp>
#include <model/model_base.hpp>
#include <model/penalty.hpp>

namespace libcf {

class RecsysModelBase : public ModelBase {
 public:
  virtual bool is_implicit() const { return true; }
  virtual double rating_converter(double x) const { return x > 3.0 ? 1. : 0.; }

  virtual void reset(const Data& data_set) {
    ModelBase::reset(data_set);
    user_rated_items_ = data_->get_feature_pair_label_hashtable(0, 1);
    num_users_ = data_->feature_group_total_dimension(0);
    num_items_ = data_->feature_group_total_dimension(1);
  }
  virtual double predict_user_item_rating(size_t, size_t) const { return 0.; }
  virtual double predict(const Instance& ins) const {
    return predict_user_item_rating(ins.get_feature_group_index(0, 0), ins.get_feature_group_index(1, 0));
  }
  virtual size_t sample_negative_item(const std::unordered_map<size_t, double>& rated) const {
    for (;;) { const size_t it = static_cast<size_t>(rand()) % num_items_; if (!rated.count(it)) return it; }
  }
  virtual size_t sample_negative_item(const std::unordered_set<size_t>& rated) const {
    for (;;) { const size_t it = static_cast<size_t>(rand()) % num_items_; if (!rated.count(it)) return it; }
  }
  virtual void pre_recommend() {}

  virtual std::vector<size_t> recommend(size_t uid, size_t topk, const std::unordered_map<size_t, double>& rated) const {
    typedef std::pair<size_t, double> P;
    Heap<P> heap(sort_by_second_desc<size_t, double>, topk);
    for (size_t item = 0; item < num_items_; ++item) {
      if (rated.count(item)) continue;
      const P cand(item, predict_user_item_rating(uid, item));
      if (heap.size() < topk) heap.push(cand); else heap.push_and_pop(cand);
    }
    CHECK_EQ(heap.size(), topk);
    std::vector<P> sorted = heap.get_sorted_data();
    std::vector<size_t> out(topk);
    for (size_t i = 0; i < topk; ++i) out[i] = sorted[i].first;
    return out;
  }

 protected:
  size_t num_users_ = 0, num_items_ = 0;
  std::unordered_map<size_t, std::unordered_map<size_t, double>> user_rated_items_;
};

}  // namespace libcf
#endif
