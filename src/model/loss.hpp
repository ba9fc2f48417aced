// Loss functions with the reference's enum values and interface (src/model/loss.hpp:10-33, 348-367).
// On the GPU path SQUARE and CROSS_ENTROPY reach the CDAE kernels (the two that CDAE's linear output supports,
// SURVEY.md T5) and SQUARE / LOGISTIC / LOG / HINGE / CROSS_ENTROPY the IMF / BPR kernels (cdae_mf_kernels.hpp); the host
// classes exist for Loss::create / loss_type() logging, positive_label() / negative_label() and the CPU-side baselines.
#ifndef CDAE_HOST_MODEL_LOSS_HPP_
#define CDAE_HOST_MODEL_LOSS_HPP_

#include <algorithm>
#include <cmath>
#include <memory>
#include <string>

#include <glog/logging.h>

namespace libcf {

enum LossType { SQUARE = 0, LOGISTIC, LOG, HINGE, SQUARED_HINGE, CROSS_ENTROPY, LOGM };

class Loss {
 public:
  virtual ~Loss() {}
  static std::shared_ptr<Loss> create(const LossType& lt);
  virtual LossType loss() const = 0;
  virtual std::string loss_type() const = 0;
  virtual double evaluate(double pred, double truth) const = 0;
  virtual double gradient(double pred, double truth) const = 0;
  virtual double predict(double x) const { return x; }
  virtual double positive_label() const { return 1.; }
  virtual double negative_label() const { return 0.; }
};

namespace loss_detail {
inline double sigmoid(double x) { return 1. / (1. + std::exp(-x)); }

struct Square : Loss {                     // (t - y)^2
  LossType loss() const { return SQUARE; }
  std::string loss_type() const { return "Square"; }
  double evaluate(double y, double t) const { return (t - y) * (t - y); }
  double gradient(double y, double t) const { return -2. * (t - y); }
};
struct Logistic : Loss {                   // -t log p - (1-t) log(1-p), p must already be a probability
  LossType loss() const { return LOGISTIC; }
  std::string loss_type() const { return "Logistic"; }
  double evaluate(double p, double t) const {
    CHECK(p >= 0. && p <= 1.) << "LOGISTIC expects a probability; CDAE's output is linear — use CE (CROSS_ENTROPY)";
    return t == 0. ? -std::log(std::max(1e-4, 1. - p)) : -std::log(std::max(1e-4, p));
  }
  double gradient(double p, double t) const {
    CHECK(p > 0. && p < 1.) << "LOGISTIC expects a probability; CDAE's output is linear — use CE (CROSS_ENTROPY)";
    return (p - t) / (p * (1. - p));
  }
};
struct CrossEntropy : Loss {               // sigmoid applied inside the loss: (1-t) y + log(1 + e^-y)
  LossType loss() const { return CROSS_ENTROPY; }
  std::string loss_type() const { return "CrossEntropy"; }
  double evaluate(double y, double t) const {
    const double lin = (1. - t) * y;
    if (y > 18.) return lin + std::exp(-y);
    if (y < -18.) return lin - y;
    return lin + std::log1p(std::exp(-y));
  }
  double gradient(double y, double t) const {
    if (y < -18.) return std::exp(y) - t;
    if (y > 18.) return 1. - t;
    return sigmoid(y) - t;
  }
  double predict(double x) const { return sigmoid(x); }
};
struct LogLoss : Loss {                    // log(1 + e^{-y t}), labels +-1
  LossType loss() const { return LOG; }
  std::string loss_type() const { return "Log"; }
  double evaluate(double y, double t) const { const double m = y * t; return m > 18. ? std::exp(-m) : (m < -18. ? -m : std::log1p(std::exp(-m))); }
  double gradient(double y, double t) const { return -t / (1. + std::exp(y * t)); }
  double negative_label() const { return -1.; }
};
struct Hinge : Loss {                      // max(0, 1 - y t), labels +-1
  LossType loss() const { return HINGE; }
  std::string loss_type() const { return "Hinge"; }
  double evaluate(double y, double t) const { return std::max(0., 1. - y * t); }
  double gradient(double y, double t) const { return y * t > 1. ? 0. : -t; }      // loss.hpp:283-288 (z == 1 still steps)
  double negative_label() const { return -1.; }
};
struct SquaredHinge : Loss {
  LossType loss() const { return SQUARED_HINGE; }
  std::string loss_type() const { return "SquaredHinge"; }
  double evaluate(double y, double t) const { const double m = std::max(0., 1. - y * t); return m * m; }
  double gradient(double y, double t) const { const double m = std::max(0., 1. - y * t); return -2. * t * m; }
  double negative_label() const { return -1.; }
};
}  // namespace loss_detail

inline std::shared_ptr<Loss> Loss::create(const LossType& lt) {
  switch (lt) {
    case SQUARE: return std::make_shared<loss_detail::Square>();
    case LOGISTIC: return std::make_shared<loss_detail::Logistic>();
    case LOG: case LOGM: return std::make_shared<loss_detail::LogLoss>();
    case HINGE: return std::make_shared<loss_detail::Hinge>();
    case SQUARED_HINGE: return std::make_shared<loss_detail::SquaredHinge>();
    case CROSS_ENTROPY: return std::make_shared<loss_detail::CrossEntropy>();
  }
  return std::make_shared<loss_detail::Square>();
}

}  // namespace libcf
#endif
