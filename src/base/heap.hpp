// Bounded top-k heap with the reference's interface and tie behaviour (src/base/heap.hpp:13-88,
// exercised by test/heap_test.hpp): front() is the element every other element is "before" under comp;
// push_and_pop(t) admits t only when comp(t, front()) holds strictly, so on equal scores the earlier
// element stays.
#ifndef CDAE_HOST_BASE_HEAP_HPP_
#define CDAE_HOST_BASE_HEAP_HPP_

#include <algorithm>
#include <functional>
#include <vector>

#include <glog/logging.h>

namespace libcf {

template <class T>
class Heap {
 public:
  typedef std::function<bool(const T&, const T&)> func_type;
  explicit Heap(const func_type& comp, size_t reserve_size = 0) : comp_(comp) { v_.reserve(reserve_size); }
  template <class It>
  Heap(It a, It b, const func_type& comp) : v_(a, b), comp_(comp) { std::make_heap(v_.begin(), v_.end(), comp_); }

  void push(const T& t) { v_.push_back(t); std::push_heap(v_.begin(), v_.end(), comp_); }
  T pop() {
    CHECK(!v_.empty()) << "pop() on an empty heap";
    std::pop_heap(v_.begin(), v_.end(), comp_);
    T out = std::move(v_.back());
    v_.pop_back();
    return out;
  }
  T push_and_pop(const T& t) {
    if (!comp_(t, v_.front())) return t;
    T out = pop();
    push(t);
    return out;
  }
  T& front() { return v_.front(); }
  size_t size() const { return v_.size(); }
  void sort() { std::sort_heap(v_.begin(), v_.end(), comp_); }
  std::vector<T> get_data() { return std::move(v_); }
  std::vector<T> get_sorted_data() { sort(); return std::move(v_); }
  std::vector<T> get_data_copy() const { return v_; }
  std::vector<T> get_sorted_data_copy() const { std::vector<T> c(v_); std::sort_heap(c.begin(), c.end(), comp_); return c; }

 private:
  std::vector<T> v_;
  func_type comp_;
};

}  // namespace libcf
#endif
