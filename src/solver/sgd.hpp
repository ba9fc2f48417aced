// Per-instance SGD driver of the reference (src/solver/sgd.hpp, sgd-inl.hpp:66-82).  CDAE never uses it
// (its update is inside the model); apps/yelp only includes the header.  Minimal form.
#ifndef CDAE_HOST_SOLVER_SGD_HPP_
#define CDAE_HOST_SOLVER_SGD_HPP_

#include <solver/solver.hpp>

namespace libcf {

template <class Model>
class SGD : public Solver<Model> {
 public:
  SGD(Model& model, size_t max_iteration, double learn_rate = 0.01) : Solver<Model>(model, max_iteration), learn_rate_(learn_rate) {}
  void train_one_iteration(const Data& train_data) {
    for (auto it = train_data.begin(); it != train_data.end(); ++it) this->model_->update_one_sgd_step(*it, learn_rate_);
  }
 private:
  double learn_rate_;
};

}  // namespace libcf
#endif
