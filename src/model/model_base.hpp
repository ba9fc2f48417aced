// Model concept base (reference: src/model/model_base.hpp:17-84).
#ifndef CDAE_HOST_MODEL_MODEL_BASE_HPP_
#define CDAE_HOST_MODEL_MODEL_BASE_HPP_

#include <memory>
#include <unordered_set>

#include <base/data.hpp>
#include <base/heap.hpp>
#include <base/mat.hpp>
#include <model/loss.hpp>
#include <model/penalty.hpp>

namespace libcf {

class ModelBase {
 public:
  virtual ~ModelBase() {}
  virtual void reset(const Data& data_set) { data_ = &data_set; }
  virtual double current_loss(const Data& data_set, size_t sample_size = 0) const {
    return data_loss(data_set, sample_size) + penalty_loss();
  }
  virtual double data_loss(const Data&, size_t = 0) const { return 0.; }
  virtual double penalty_loss() const { return 0.; }
  virtual double predict(const Instance&) const { LOG(FATAL) << "Unimplemented!"; return 0.; }
  virtual double regularization_coefficent() const { return 0.; }
  virtual void train_one_iteration(const Data&) { LOG(FATAL) << "Unimplemented!"; }

 protected:
  const Data* data_ = nullptr;          // non-owning: the train set must outlive the model (model_base.hpp:63)
  std::shared_ptr<Loss> loss_;
  std::shared_ptr<Penalty> penalty_;
};

class SGDBase {
 public:
  virtual ~SGDBase() {}
  virtual void update_one_sgd_step(const Instance&, double) { LOG(FATAL) << "update_one_sgd_step not implemented!"; }
};

}  // namespace libcf
#endif
