// Dense types of the host layer.  The reference aliases Eigen (row-major fp64, src/base/mat.hpp:9-22);
// Eigen is not available here and the GPU path keeps all CDAE parameters on the device, so the host
// only needs small owning containers for the out-of-scope sibling models.
#ifndef CDAE_HOST_BASE_MAT_HPP_
#define CDAE_HOST_BASE_MAT_HPP_

#include <cstddef>
#include <vector>

namespace libcf {

class DVector {
 public:
  DVector() = default;
  explicit DVector(size_t n, double v = 0.) : d_(n, v) {}
  static DVector Zero(size_t n) { return DVector(n, 0.); }
  static DVector Ones(size_t n) { return DVector(n, 1.); }
  size_t size() const { return d_.size(); }
  double& operator()(size_t i) { return d_[i]; }
  double operator()(size_t i) const { return d_[i]; }
  double& operator[](size_t i) { return d_[i]; }
  double operator[](size_t i) const { return d_[i]; }
  double dot(const DVector& o) const { double s = 0; for (size_t i = 0; i < d_.size(); ++i) s += d_[i] * o.d_[i]; return s; }
  double squaredNorm() const { return dot(*this); }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
 private:
  std::vector<double> d_;
};

class DMatrix {   // row-major
 public:
  DMatrix() = default;
  DMatrix(size_t r, size_t c, double v = 0.) : r_(r), c_(c), d_(r * c, v) {}
  static DMatrix Constant(size_t r, size_t c, double v) { return DMatrix(r, c, v); }
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  size_t size() const { return d_.size(); }
  double& operator()(size_t i, size_t j) { return d_[i * c_ + j]; }
  double operator()(size_t i, size_t j) const { return d_[i * c_ + j]; }
  double* row(size_t i) { return d_.data() + i * c_; }
  const double* row(size_t i) const { return d_.data() + i * c_; }
  double squaredNorm() const { double s = 0; for (double x : d_) s += x * x; return s; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
 private:
  size_t r_ = 0, c_ = 0;
  std::vector<double> d_;
};

}  // namespace libcf
#endif
