// Process-wide generator with the reference's static interface (src/base/random.hpp:11-82):
// Random::seed / timed_seed / uniform / shuffle.  apps/yelp seeds it before splitting the data
// (yelp.cpp:90) and calls timed_seed() before training (yelp.cpp:107); the GPU CDAE derives the seed of
// its counter-based streams (include/cdae_rng.h) from one draw of this generator at reset().
#ifndef CDAE_HOST_BASE_RANDOM_HPP_
#define CDAE_HOST_BASE_RANDOM_HPP_

#include <algorithm>
#include <cstdint>
#include <ctime>
#include <random>

namespace libcf {

class Random {
 public:
  typedef std::mt19937_64 rng_type;
  static rng_type& engine() { static rng_type e; return e; }
  static void seed() { std::random_device rd; engine().seed(rd()); }
  static void seed(size_t s) { engine().seed(s); }
  static void timed_seed() { engine().seed(static_cast<uint64_t>(std::time(nullptr))); }
  static double uniform(double lo = 0., double hi = 1.) { return std::uniform_real_distribution<double>(lo, hi)(engine()); }
  static double normal(double mean = 0., double stddev = 1.) { return std::normal_distribution<double>(mean, stddev)(engine()); }
  static size_t uniform(size_t begin, size_t end) {
    return begin + static_cast<size_t>(std::uniform_int_distribution<uint64_t>(0, end - begin - 1)(engine()));
  }
  static uint64_t next_u64() { return engine()(); }
  template <class It> static void shuffle(It a, It b) { std::shuffle(a, b, engine()); }
};

}  // namespace libcf
#endif
