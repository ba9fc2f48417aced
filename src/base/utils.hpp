// Small helpers with the reference's names (src/base/utils.hpp:14-91).
#ifndef CDAE_HOST_BASE_UTILS_HPP_
#define CDAE_HOST_BASE_UTILS_HPP_

#include <functional>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

#include <glog/logging.h>

#include <base/timer.hpp>

namespace libcf {

// comparator of the top-k heap: "a before b" when a scores strictly higher (utils.hpp:16-19)
template <typename K, typename V>
inline bool sort_by_second_desc(const std::pair<K, V>& a, const std::pair<K, V>& b) { return a.second > b.second; }
template <typename K, typename V>
inline bool sort_by_second_asc(const std::pair<K, V>& a, const std::pair<K, V>& b) { return a.second < b.second; }

template <class A, class B>
std::ostream& operator<<(std::ostream& o, const std::pair<A, B>& p) { return o << '(' << p.first << ',' << p.second << ')'; }

template <class T>
std::ostream& operator<<(std::ostream& o, const std::vector<T>& v) {
  o << '[';
  const size_t shown = v.size() < 10 ? v.size() : 10;
  for (size_t i = 0; i < shown; ++i) o << (i ? "," : "") << v[i];
  if (shown < v.size()) o << ",...";
  return o << ']';
}

inline void time_function(const std::function<void()>& fn, const std::string& msg = "") {
  Timer t;
  fn();
  LOG(INFO) << "(" << msg << ") took " << t;
}

}  // namespace libcf
#endif
