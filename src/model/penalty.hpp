// Regularisers (reference: src/model/penalty.hpp:11-68).  Only used for the reported loss; the training
// gradient hard-codes lambda * param (cdae.hpp:231, 252, ...).
#ifndef CDAE_HOST_MODEL_PENALTY_HPP_
#define CDAE_HOST_MODEL_PENALTY_HPP_

#include <cmath>
#include <memory>
#include <string>

#include <base/mat.hpp>

namespace libcf {

enum PenaltyType { L1 = 0, L2 };

class Penalty {
 public:
  virtual ~Penalty() {}
  static std::shared_ptr<Penalty> create(const PenaltyType& pt);
  virtual std::string penalty_type() const = 0;
  virtual bool is_smooth() const = 0;
  virtual double evaluate(const DMatrix& m) = 0;
  virtual double evaluate(const DVector& v) = 0;
};
struct L2Penalty : Penalty {
  std::string penalty_type() const { return "L2"; }
  bool is_smooth() const { return true; }
  double evaluate(const DMatrix& m) { return m.squaredNorm(); }
  double evaluate(const DVector& v) { return v.squaredNorm(); }
};
struct L1Penalty : Penalty {
  std::string penalty_type() const { return "L1"; }
  bool is_smooth() const { return false; }
  double evaluate(const DMatrix& m) { double s = 0; for (size_t i = 0; i < m.size(); ++i) s += std::fabs(m.data()[i]); return s; }
  double evaluate(const DVector& v) { double s = 0; for (size_t i = 0; i < v.size(); ++i) s += std::fabs(v[i]); return s; }
};
inline std::shared_ptr<Penalty> Penalty::create(const PenaltyType& pt) {
  if (pt == L1) return std::make_shared<L1Penalty>();
  return std::make_shared<L2Penalty>();
}

}  // namespace libcf
#endif
