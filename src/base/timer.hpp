// Wall-clock timer, seconds (reference: src/base/timer.hpp:9-38).
#ifndef CDAE_HOST_BASE_TIMER_HPP_
#define CDAE_HOST_BASE_TIMER_HPP_

#include <chrono>
#include <ostream>

namespace libcf {

class Timer {
 public:
  Timer() { reset(); }
  void start() { reset(); }
  void reset() { t0_ = std::chrono::steady_clock::now(); }
  double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
  friend std::ostream& operator<<(std::ostream& o, const Timer& t) { return o << t.elapsed() << " secs"; }
 private:
  std::chrono::steady_clock::time_point t0_;
};

}  // namespace libcf
#endif
