// One (user, item, label) record behind the reference's Instance interface
// (src/base/instance.hpp:120-226): feature groups are addressed as (group, position); RECSYS data has two
// single-valued SPARSE_BINARY groups, user = group 0 and item = group 1 (data-inl.hpp:44-63).
#ifndef CDAE_HOST_BASE_INSTANCE_HPP_
#define CDAE_HOST_BASE_INSTANCE_HPP_

#include <cstdint>
#include <ostream>
#include <string>
#include <unordered_map>
#include <vector>

#include <base/io.hpp>
#include <base/utils.hpp>

namespace libcf {

enum LabelType { EMPTY = 0, BINARY, MULTICLASS, CONTINUOUS };
enum FeatureType { DENSE = 0, SPARSE, SPARSE_BINARY };

// string key -> dense id, ids in first-seen order (instance-inl.hpp:22-37)
class FeatureGroupInfo {
 public:
  FeatureGroupInfo() = default;
  explicit FeatureGroupInfo(const FeatureType& ft) : type_(ft) {}
  size_t get_index(const std::string& key, bool allow_new_value = true) {
    auto it = ids_.find(key);
    if (it != ids_.end()) return it->second;
    if (!allow_new_value) return static_cast<size_t>(-1);
    const size_t id = names_.size();
    ids_.emplace(key, id);
    names_.push_back(key);
    return id;
  }
  size_t size() const { return type_ == DENSE ? length_ : names_.size(); }
  void set_length(size_t n) { length_ = n; }
  FeatureType feature_type() const { return type_; }
  const std::string& name(size_t id) const { return names_[id]; }
  void write(std::ostream& o) const {
    io_detail::put<uint64_t>(o, (uint64_t)type_); io_detail::put<uint64_t>(o, length_); io_detail::put<uint64_t>(o, names_.size());
    for (auto& s : names_) io_detail::put_str(o, s);
  }
  void read(std::istream& i) {
    uint64_t t = 0, len = 0, n = 0;
    io_detail::get(i, t); io_detail::get(i, len); io_detail::get(i, n);
    type_ = (FeatureType)t; length_ = len; names_.resize(n); ids_.clear();
    for (uint64_t k = 0; k < n; ++k) { io_detail::get_str(i, names_[k]); ids_.emplace(names_[k], k); }
  }
 private:
  std::unordered_map<std::string, size_t> ids_;
  std::vector<std::string> names_;
  size_t length_ = 0;
  FeatureType type_ = SPARSE_BINARY;
};

// A (user, item, label) record as a VALUE: at most two single-valued feature groups held inline — no heap allocation per
// rating (the reference's Instance owns three std::vectors, instance.hpp:196-226; at 100 M ratings that is 300 M mallocs).
// libcf::Data keeps the ratings as columns and hands these out by value from its iterators.
class Instance {
 public:
  static const size_t kMaxGroups = 2;
  Instance() = default;
  Instance(uint32_t g0, uint32_t g1, double label) : label_(label), n_(2) { idx_[0] = g0; idx_[1] = g1; }
  // append a single-valued categorical group (RECSYS loader)
  void add_feat_group(FeatureGroupInfo& info, const std::string& key) { push(info.get_index(key)); }
  void add_feat_group(const std::vector<size_t>& ids) { for (size_t v : ids) push(v); }
  double label() const { return label_; }
  void set_label(double l) { label_ = l; }
  size_t size() const { return n_; }
  size_t num_feature_groups() const { return n_; }
  size_t feature_group_size(size_t) const { return 1; }
  size_t get_feature_group_index(size_t fg, size_t /*pos*/) const { return idx_[fg]; }
  double get_feature_group_value(size_t /*fg*/, size_t /*pos*/) const { return 1.; }
  friend std::ostream& operator<<(std::ostream& o, const Instance& ins) {
    o << ins.label_ << " |";
    for (size_t k = 0; k < ins.n_; ++k) o << ' ' << k << ':' << ins.idx_[k];
    return o;
  }
 private:
  void push(size_t v) {
    CHECK_LT(static_cast<size_t>(n_), kMaxGroups) << "this build's Instance holds the two single-valued groups of RECSYS data";
    CHECK_LT(v, size_t(1) << 32);
    idx_[n_++] = static_cast<uint32_t>(v);
  }
  double label_ = 0.;
  uint32_t idx_[kMaxGroups] = {0, 0};
  uint32_t n_ = 0;
};

}  // namespace libcf
#endif
