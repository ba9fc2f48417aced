// Dataset container behind the reference's Data interface (src/base/data.hpp:48-181,
// data-inl.hpp): RECSYS text loading with first-seen id dictionaries (data-inl.hpp:44-63), the per-user
// random split (data-inl.hpp:231-272), the uid -> {iid -> label} table the models are reset from
// (data-inl.hpp:414-429), plus a direct CSR export for the GPU path.
#ifndef CDAE_HOST_BASE_DATA_HPP_
#define CDAE_HOST_BASE_DATA_HPP_

#include <algorithm>
#include <functional>
#include <memory>
#include <ostream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <base/instance.hpp>
#include <base/io.hpp>
#include <base/mat.hpp>
#include <base/random.hpp>
#include <base/timer.hpp>
#include <base/utils.hpp>

namespace libcf {

enum DataFormat { VECTOR, LIBSVM, RECSYS };

class DataInfo {
 public:
  std::vector<FeatureGroupInfo> feature_group_infos_;
  size_t total_dimensions_ = 0;
  std::vector<size_t> feature_group_global_idx_;
  LabelType label_type_ = CONTINUOUS;
};

class Data {
 public:
  typedef std::function<std::vector<std::string>(const std::string&)> LineParser;

  Data() = default;
  Data(std::vector<Instance>&& v, const std::shared_ptr<DataInfo>& info) : instances_(std::move(v)), info_(info) {}

  void load(const std::string& filename, const DataFormat& df, const LineParser& parser, bool skip_header = false) {
    CHECK(df == RECSYS) << "only the RECSYS text format is provided by this build";
    if (!info_) info_ = std::make_shared<DataInfo>();
    info_->feature_group_infos_.assign(2, FeatureGroupInfo(SPARSE_BINARY));   // user, item
    info_->label_type_ = CONTINUOUS;
    FileLineReader reader(filename);
    reader.set_line_callback([&](const std::string& line, size_t line_num) {
      if (skip_header && line_num == 0) return;
      const std::vector<std::string> f = parser(line);
      if (f.empty()) return;
      CHECK_GE(f.size(), size_t(3)) << "RECSYS parser must return {user, item, label}";
      Instance ins;
      ins.add_feat_group(info_->feature_group_infos_[0], f[0]);
      ins.add_feat_group(info_->feature_group_infos_[1], f[1]);
      ins.set_label(std::stod(f[2]));
      instances_.push_back(std::move(ins));
    });
    reader.load();
    finalize_dimensions();
    LOG(INFO) << "Data loaded successfully.\n" << *this;
  }

  size_t size() const { return instances_.size(); }
  size_t num_feature_groups() const { CHECK(info_ != nullptr); return info_->feature_group_infos_.size(); }
  size_t total_dimensions() const { CHECK(info_ != nullptr); return info_->total_dimensions_; }
  size_t feature_group_total_dimension(size_t fg) const {
    CHECK_LT(fg, num_feature_groups());
    return info_->feature_group_infos_[fg].size();
  }
  size_t feature_group_start_idx(size_t fg) const { return info_->feature_group_global_idx_[fg]; }
  std::shared_ptr<DataInfo> get_data_info() const { return info_; }

  const Instance* data() const { return instances_.data(); }
  Instance* data() { return instances_.data(); }
  const Instance* begin() const { return data(); }
  Instance* begin() { return data(); }
  const Instance* end() const { return data() + size(); }
  Instance* end() { return data() + size(); }

  void shuffle_data() { Random::shuffle(instances_.begin(), instances_.end()); }

  // instance positions per id of one feature group
  std::unordered_map<size_t, std::vector<size_t>> get_feature_ins_idx_hashtable(size_t fg) const {
    std::unordered_map<size_t, std::vector<size_t>> out;
    out.reserve(feature_group_total_dimension(fg));
    for (size_t i = 0; i < instances_.size(); ++i) out[instances_[i].get_feature_group_index(fg, 0)].push_back(i);
    return out;
  }
  std::unordered_map<size_t, std::unordered_map<size_t, double>> get_feature_pair_label_hashtable(size_t a, size_t b) const {
    std::unordered_map<size_t, std::unordered_map<size_t, double>> out;
    out.reserve(feature_group_total_dimension(a));
    for (const Instance& ins : instances_)
      out[ins.get_feature_group_index(a, 0)].emplace(ins.get_feature_group_index(b, 0), ins.label());
    return out;
  }
  std::unordered_map<size_t, std::unordered_set<size_t>> get_feature_to_set_hashtable(size_t a, size_t b) const {
    std::unordered_map<size_t, std::unordered_set<size_t>> out;
    for (const Instance& ins : instances_) out[ins.get_feature_group_index(a, 0)].insert(ins.get_feature_group_index(b, 0));
    return out;
  }
  std::unordered_map<size_t, std::vector<size_t>> get_feature_to_vec_hashtable(size_t a, size_t b) const {
    std::unordered_map<size_t, std::vector<size_t>> out;
    for (const Instance& ins : instances_) out[ins.get_feature_group_index(a, 0)].push_back(ins.get_feature_group_index(b, 0));
    return out;
  }

  // CSR of group a -> sorted unique ids of group b, every id of a present (possibly empty row)
  void to_csr(size_t a, size_t b, std::vector<int64_t>& row_ptr, std::vector<uint32_t>& col) const {
    const size_t rows = feature_group_total_dimension(a);
    std::vector<std::vector<uint32_t>> tmp(rows);
    for (const Instance& ins : instances_) tmp[ins.get_feature_group_index(a, 0)].push_back((uint32_t)ins.get_feature_group_index(b, 0));
    row_ptr.assign(rows + 1, 0);
    col.clear();
    col.reserve(instances_.size());
    for (size_t r = 0; r < rows; ++r) {
      std::sort(tmp[r].begin(), tmp[r].end());
      tmp[r].erase(std::unique(tmp[r].begin(), tmp[r].end()), tmp[r].end());
      col.insert(col.end(), tmp[r].begin(), tmp[r].end());
      row_ptr[r + 1] = (int64_t)col.size();
    }
  }

  // per id of group fg: shuffle its instances, the first floor(ratio * n) go to test (data-inl.hpp:249-261)
  void random_split_by_feature_group(Data& train, Data& test, size_t fg, double test_ratio) const {
    Timer timer;
    std::vector<Instance> tr, te;
    tr.reserve(size());
    te.reserve(static_cast<size_t>(size() * test_ratio) + 16);
    auto groups = get_feature_ins_idx_hashtable(fg);
    CHECK_EQ(groups.size(), feature_group_total_dimension(fg));
    std::vector<size_t> keys;
    keys.reserve(groups.size());
    for (auto& kv : groups) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());                       // deterministic given Random::seed
    for (size_t key : keys) {
      std::vector<size_t>& idx = groups[key];
      Random::shuffle(idx.begin(), idx.end());
      const size_t n_test = static_cast<size_t>(idx.size() * test_ratio);
      for (size_t k = 0; k < idx.size(); ++k) (k < n_test ? te : tr).push_back(instances_[idx[k]]);
    }
    CHECK_EQ(tr.size() + te.size(), size());
    Random::shuffle(tr.begin(), tr.end());
    Random::shuffle(te.begin(), te.end());
    train = Data(std::move(tr), info_);
    test = Data(std::move(te), info_);
    LOG(INFO) << "Finished splitting data set in " << timer;
  }

  void write(std::ostream& o) const {
    CHECK(info_ != nullptr);
    io_detail::put<uint64_t>(o, info_->feature_group_infos_.size());
    for (auto& g : info_->feature_group_infos_) g.write(o);
    io_detail::put<uint64_t>(o, instances_.size());
    for (auto& ins : instances_) ins.write(o);
  }
  void read(std::istream& i) {
    info_ = std::make_shared<DataInfo>();
    uint64_t ng = 0; io_detail::get(i, ng);
    info_->feature_group_infos_.resize(ng);
    for (auto& g : info_->feature_group_infos_) g.read(i);
    uint64_t n = 0; io_detail::get(i, n);
    instances_.resize(n);
    for (auto& ins : instances_) ins.read(i);
    finalize_dimensions();
  }

  friend std::ostream& operator<<(std::ostream& o, const Data& d) {
    o << "\nData set summary : \n\tNum of Instance: " << d.instances_.size() << "\n";
    if (d.info_) {
      o << "\tNum of feature groups: " << d.info_->feature_group_infos_.size() << "\n\tTotal feature dimensions: "
        << d.info_->total_dimensions_ << "\n";
      for (size_t g = 0; g < d.info_->feature_group_infos_.size(); ++g)
        o << "\tFeature group " << g << " -> size " << d.info_->feature_group_infos_[g].size() << "\n";
    }
    o << "Head of the data set:\n";
    for (size_t k = 0; k < std::min<size_t>(10, d.instances_.size()); ++k) o << "  " << d.instances_[k] << "\n";
    return o;
  }

 private:
  void finalize_dimensions() {
    info_->total_dimensions_ = 0;
    info_->feature_group_global_idx_.assign(info_->feature_group_infos_.size(), 0);
    for (size_t g = 0; g < info_->feature_group_infos_.size(); ++g) {
      info_->feature_group_global_idx_[g] = info_->total_dimensions_;
      info_->total_dimensions_ += info_->feature_group_infos_[g].size();
    }
  }
  std::vector<Instance> instances_;
  std::shared_ptr<DataInfo> info_;
};

}  // namespace libcf
#endif
