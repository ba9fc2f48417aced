// Epoch driver with the reference's interface and log table (src/solver/solver.hpp:11-46,
// solver-inl.hpp:6-112): copies the model, reset(train), then per iteration train_one_iteration,
// current_loss, and every eval_iterations an Evaluation<Model> row.
#ifndef CDAE_HOST_SOLVER_SOLVER_HPP_
#define CDAE_HOST_SOLVER_SOLVER_HPP_

#include <iomanip>
#include <memory>
#include <sstream>
#include <vector>

#include <base/data.hpp>
#include <base/timer.hpp>
#include <model/evaluation.hpp>

namespace libcf {

template <class Model>
class Solver {
 public:
  Solver(Model& model, size_t max_iteration, size_t eval_iterations = 1)
      : max_iteration_(max_iteration), eval_iterations(eval_iterations), model_(std::make_shared<Model>(model)) {}
  explicit Solver(Model& model) : Solver(model, 1) {}
  virtual ~Solver() {}

  std::shared_ptr<Model> get_model() { return model_; }
  virtual void pre_train(const Data&, const Data&) {}
  virtual void train_one_iteration(const Data& train_data) { model_->train_one_iteration(train_data); }

  virtual void train(const Data& train_data, const Data& validation_data = Data(), const std::vector<EvalType>& eval_types = {}) {
    std::vector<std::shared_ptr<Evaluation<Model>>> evals;
    for (EvalType et : eval_types) evals.push_back(Evaluation<Model>::create(et));
    model_->reset(train_data);
    pre_train(train_data, validation_data);
    Timer t;
    const std::string rule(110, '-');
    LOG(INFO) << rule;
    {
      std::stringstream ss;
      ss << std::setfill(' ') << std::setw(5) << "Iters" << "|" << std::setw(8) << "Time" << "|" << std::setw(10) << "Train Loss" << "|";
      if (validation_data.size() > 0) for (auto& e : evals) ss << e->evaluation_type() << "|";
      LOG(INFO) << ss.str();
    }
    auto row = [&](size_t iteration, double train_loss) {
      std::stringstream ss;
      ss << std::setw(5) << iteration << "|" << std::setw(8) << std::setprecision(3) << t.elapsed() << "|" << std::setw(10)
         << std::setprecision(5) << train_loss << "|";
      if (validation_data.size() > 0) for (auto& e : evals) ss << e->evaluate(*model_, validation_data, train_data) << "|";
      LOG(INFO) << ss.str();
    };
    size_t iteration = 0;
    row(iteration, 0.);                                        // untrained model (solver-inl.hpp:37-48)
    while (iteration < max_iteration_) {
      train_one_iteration(train_data);
      const double train_loss = model_->current_loss(train_data);
      ++iteration;
      if (iteration % eval_iterations == 0) row(iteration, train_loss);
    }
    LOG(INFO) << rule;
  }

  virtual void test(const Data& test_data, const std::vector<EvalType>& eval_types = {}) {
    Timer t;
    std::stringstream head, body;
    head << std::setfill(' ') << std::setw(8) << "Time" << "|";
    body << std::setw(8) << std::setprecision(3) << t.elapsed() << "|";
    for (EvalType et : eval_types) {
      auto e = Evaluation<Model>::create(et);
      if (test_data.size() > 0) { head << e->evaluation_type() << "|"; body << e->evaluate(*model_, test_data) << "|"; }
    }
    LOG(INFO) << head.str();
    LOG(INFO) << body.str();
  }

 protected:
  size_t max_iteration_ = 1;
  size_t eval_iterations = 1;
  std::shared_ptr<Model> model_;
};

}  // namespace libcf
#endif
