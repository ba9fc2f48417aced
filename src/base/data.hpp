// Dataset container behind the reference's Data interface (src/base/data.hpp:48-181, data-inl.hpp): RECSYS text loading
// with first-seen id dictionaries (data-inl.hpp:44-63), the per-user random split (data-inl.hpp:231-272), the
// uid -> {iid -> label} table the CPU models are reset from (data-inl.hpp:414-429) — and the direct CSR export the GPU path
// is fed from.
//
// Layout (round 3): the reference keeps one heap-allocated Instance per rating (three std::vectors each) in a
// std::vector<Instance>; at Netflix scale (100 M ratings) that is 300 M allocations and > 12 GB before the first hashtable.
// Here the ratings are three COLUMNS — user id, item id (uint32) and label (double; not stored at all while every label is
// the same value, which is the implicit-feedback case of apps/yelp, yelp.cpp:66) — 8 bytes per rating.  Text is parsed
// straight into the columns, the cache file is the columns, the split permutes positions and gathers columns, and to_csr is
// a counting sort: nothing on the path from the text file to cdae_hip_set_interactions materialises a per-rating object or
// a per-user container.  begin()/end() hand out Instance VALUES for the code that iterates ratings (Popularity, RMSE/MAE).
#ifndef CDAE_HOST_BASE_DATA_HPP_
#define CDAE_HOST_BASE_DATA_HPP_

#include <algorithm>
#include <atomic>
#include <functional>
#include <iterator>
#include <memory>
#include <numeric>
#include <ostream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <base/instance.hpp>
#include <base/io.hpp>
#include <base/mat.hpp>
#include <base/parallel.hpp>
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

  // ratings by value, in storage order
  class const_iterator {
   public:
    typedef std::random_access_iterator_tag iterator_category;
    typedef Instance value_type;
    typedef std::ptrdiff_t difference_type;
    typedef const Instance* pointer;
    typedef Instance reference;
    const_iterator() = default;
    const_iterator(const Data* d, size_t i) : d_(d), i_(i) {}
    Instance operator*() const { return d_->at(i_); }
    const Instance* operator->() const { cur_ = d_->at(i_); return &cur_; }
    const_iterator& operator++() { ++i_; return *this; }
    const_iterator operator++(int) { const_iterator t(*this); ++i_; return t; }
    const_iterator& operator+=(difference_type n) { i_ += n; return *this; }
    const_iterator operator+(difference_type n) const { return const_iterator(d_, i_ + n); }
    difference_type operator-(const const_iterator& o) const { return (difference_type)i_ - (difference_type)o.i_; }
    bool operator==(const const_iterator& o) const { return i_ == o.i_; }
    bool operator!=(const const_iterator& o) const { return i_ != o.i_; }
    bool operator<(const const_iterator& o) const { return i_ < o.i_; }
   private:
    const Data* d_ = nullptr;
    size_t i_ = 0;
    mutable Instance cur_;
  };

  Data() = default;
  // (user column, item column, labels — empty = every label is `uniform_label`)
  Data(std::vector<uint32_t>&& g0, std::vector<uint32_t>&& g1, std::vector<double>&& labels, double uniform_label,
       const std::shared_ptr<DataInfo>& info)
      : label_(std::move(labels)), uniform_label_(uniform_label), info_(info), generation_(next_generation()) {
    col_[0] = std::move(g0); col_[1] = std::move(g1);
    CHECK_EQ(col_[0].size(), col_[1].size());
    CHECK(label_.empty() || label_.size() == col_[0].size());
  }
  Data(std::vector<Instance>&& v, const std::shared_ptr<DataInfo>& info) : info_(info), generation_(next_generation()) {
    reserve(v.size());
    for (const Instance& ins : v) append(ins);
  }

  // Streaming text -> columns (data-inl.hpp:13-80 for RECSYS): a line goes through the caller's parser (yelp.cpp:60-66), its user
  // and item keys through the first-seen dictionaries, and three scalars are appended.
  void load(const std::string& filename, const DataFormat& df, const LineParser& parser, bool skip_header = false) {
    CHECK(df == RECSYS) << "only the RECSYS text format is provided by this build";
    if (!info_) info_ = std::make_shared<DataInfo>();
    info_->feature_group_infos_.assign(2, FeatureGroupInfo(SPARSE_BINARY));   // user, item
    info_->label_type_ = CONTINUOUS;
    FeatureGroupInfo& users = info_->feature_group_infos_[0];
    FeatureGroupInfo& items = info_->feature_group_infos_[1];
    FileLineReader reader(filename);
    reader.set_line_callback([&](const std::string& line, size_t line_num) {
      if (skip_header && line_num == 0) return;
      const std::vector<std::string> f = parser(line);
      if (f.empty()) return;
      CHECK_GE(f.size(), size_t(3)) << "RECSYS parser must return {user, item, label}";
      const size_t u = users.get_index(f[0]), i = items.get_index(f[1]);
      CHECK_LT(u, size_t(1) << 32); CHECK_LT(i, size_t(1) << 32);
      append(static_cast<uint32_t>(u), static_cast<uint32_t>(i), f[2] == "1" ? 1. : std::stod(f[2]));
    });
    reader.load();
    finalize_dimensions();
    generation_ = next_generation();
    LOG(INFO) << "Data loaded successfully.\n" << *this;
  }

  size_t size() const { return col_[0].size(); }
  size_t num_feature_groups() const { CHECK(info_ != nullptr); return info_->feature_group_infos_.size(); }
  size_t total_dimensions() const { CHECK(info_ != nullptr); return info_->total_dimensions_; }
  size_t feature_group_total_dimension(size_t fg) const {
    CHECK_LT(fg, num_feature_groups());
    return info_->feature_group_infos_[fg].size();
  }
  size_t feature_group_start_idx(size_t fg) const { return info_->feature_group_global_idx_[fg]; }
  std::shared_ptr<DataInfo> get_data_info() const { return info_; }

  // Identity of the CONTENTS: a new value whenever the ratings change (load / read / split result / shuffle); copies of an
  // object share it.  Caches key on this, never on addresses (a Data rebuilt at the same address with the same size is a
  // different data set).
  uint64_t generation() const { return generation_; }

  const_iterator begin() const { return const_iterator(this, 0); }
  const_iterator end() const { return const_iterator(this, size()); }
  Instance at(size_t k) const { return Instance(col_[0][k], col_[1][k], label_of(k)); }
  Instance operator[](size_t k) const { return at(k); }
  // the columns themselves (group 0 = user, 1 = item)
  const std::vector<uint32_t>& column(size_t fg) const { CHECK_LT(fg, size_t(2)); return col_[fg]; }
  double label_of(size_t k) const { return label_.empty() ? uniform_label_ : label_[k]; }

  void shuffle_data() {
    std::vector<size_t> pos(size());
    std::iota(pos.begin(), pos.end(), size_t(0));
    Random::shuffle(pos.begin(), pos.end());
    *this = gather(pos);
  }

  // ---- hashtable views for the CPU models (the GPU models never call these) ----------------------------------------
  // instance positions per id of one feature group
  std::unordered_map<size_t, std::vector<size_t>> get_feature_ins_idx_hashtable(size_t fg) const {
    std::unordered_map<size_t, std::vector<size_t>> out;
    out.reserve(feature_group_total_dimension(fg));
    const std::vector<uint32_t>& c = column(fg);
    for (size_t i = 0; i < c.size(); ++i) out[c[i]].push_back(i);
    return out;
  }
  std::unordered_map<size_t, std::unordered_map<size_t, double>> get_feature_pair_label_hashtable(size_t a, size_t b) const {
    std::unordered_map<size_t, std::unordered_map<size_t, double>> out;
    out.reserve(feature_group_total_dimension(a));
    const std::vector<uint32_t>& ca = column(a); const std::vector<uint32_t>& cb = column(b);
    for (size_t k = 0; k < ca.size(); ++k) out[ca[k]].emplace(cb[k], label_of(k));
    return out;
  }
  std::unordered_map<size_t, std::unordered_set<size_t>> get_feature_to_set_hashtable(size_t a, size_t b) const {
    std::unordered_map<size_t, std::unordered_set<size_t>> out;
    const std::vector<uint32_t>& ca = column(a); const std::vector<uint32_t>& cb = column(b);
    for (size_t k = 0; k < ca.size(); ++k) out[ca[k]].insert(cb[k]);
    return out;
  }
  std::unordered_map<size_t, std::vector<size_t>> get_feature_to_vec_hashtable(size_t a, size_t b) const {
    std::unordered_map<size_t, std::vector<size_t>> out;
    const std::vector<uint32_t>& ca = column(a); const std::vector<uint32_t>& cb = column(b);
    for (size_t k = 0; k < ca.size(); ++k) out[ca[k]].push_back(cb[k]);
    return out;
  }

  // CSR of group a -> sorted unique ids of group b, every id of a present (possibly empty row): what RecsysModelBase::reset's
  // uid -> {iid -> label} table (recsys_model_base.hpp:29-34, data-inl.hpp:414-429) holds for implicit data, without the table.
  // Counting sort by row, then every row sorted and deduplicated in place (rows in parallel on --num_thread threads).
  void to_csr(size_t a, size_t b, std::vector<int64_t>& row_ptr, std::vector<uint32_t>& col) const {
    const size_t rows = feature_group_total_dimension(a), n = size();
    const std::vector<uint32_t>& ca = column(a); const std::vector<uint32_t>& cb = column(b);
    row_ptr.assign(rows + 1, 0);
    for (size_t k = 0; k < n; ++k) { CHECK_LT(ca[k], rows); row_ptr[ca[k] + 1]++; }
    for (size_t r = 0; r < rows; ++r) row_ptr[r + 1] += row_ptr[r];
    col.resize(n);
    {
      std::vector<int64_t> cursor(row_ptr.begin(), row_ptr.end() - 1);
      for (size_t k = 0; k < n; ++k) col[cursor[ca[k]]++] = cb[k];
    }
    std::vector<int64_t> kept(rows, 0);
    const size_t block = 1024, blocks = (rows + block - 1) / block;
    dynamic_parallel_for(0, blocks, [&](size_t blk) {
      for (size_t r = blk * block; r < std::min(rows, (blk + 1) * block); ++r) {
        uint32_t* p = col.data() + row_ptr[r]; uint32_t* q = col.data() + row_ptr[r + 1];
        std::sort(p, q);
        kept[r] = std::unique(p, q) - p;
      }
    });
    int64_t w = 0;
    for (size_t r = 0; r < rows; ++r) {                     // close the gaps duplicates left (none for well-formed data)
      const int64_t src = row_ptr[r];
      if (w != src) std::copy(col.begin() + src, col.begin() + src + kept[r], col.begin() + w);
      row_ptr[r] = w;
      w += kept[r];
    }
    row_ptr[rows] = w;
    col.resize(static_cast<size_t>(w));
  }

  // whole-set split (data-inl.hpp:206-229): one shuffle of the positions; the first floor((1 - ratio) * n) go to train, the rest to test
  void random_split(Data& train, Data& test, double test_ratio) const {
    CHECK_LT(test_ratio, 1.0);
    const size_t n = size(), n_train = static_cast<size_t>((1. - test_ratio) * n);
    std::vector<size_t> pos(n);
    for (size_t k = 0; k < n; ++k) pos[k] = k;
    Random::shuffle(pos.begin(), pos.end());
    train = gather(std::vector<size_t>(pos.begin(), pos.begin() + n_train));
    test = gather(std::vector<size_t>(pos.begin() + n_train, pos.end()));
  }

  // per id of group fg: shuffle its instances, the first floor(ratio * n) go to test (data-inl.hpp:249-261); ids in ascending
  // order and positions in storage order before each shuffle, so the result is a function of Random's state alone.
  void random_split_by_feature_group(Data& train, Data& test, size_t fg, double test_ratio) const {
    Timer timer;
    const size_t n = size(), groups = feature_group_total_dimension(fg);
    const std::vector<uint32_t>& c = column(fg);
    std::vector<size_t> start(groups + 1, 0);
    for (size_t k = 0; k < n; ++k) { CHECK_LT(c[k], groups); start[c[k] + 1]++; }
    for (size_t g = 0; g < groups; ++g) { CHECK_GT(start[g + 1], size_t(0)) << "id " << g << " of group " << fg << " has no instance"; start[g + 1] += start[g]; }
    std::vector<size_t> pos(n);
    {
      std::vector<size_t> cursor(start.begin(), start.end() - 1);
      for (size_t k = 0; k < n; ++k) pos[cursor[c[k]]++] = k;            // stable: storage order inside a group
    }
    std::vector<size_t> tr, te;
    tr.reserve(n);
    te.reserve(static_cast<size_t>(n * test_ratio) + 16);
    for (size_t g = 0; g < groups; ++g) {
      size_t* a = pos.data() + start[g]; size_t* e = pos.data() + start[g + 1];
      Random::shuffle(a, e);
      const size_t n_test = static_cast<size_t>((e - a) * test_ratio);
      te.insert(te.end(), a, a + n_test);
      tr.insert(tr.end(), a + n_test, e);
    }
    CHECK_EQ(tr.size() + te.size(), n);
    std::vector<size_t>().swap(pos);
    Random::shuffle(tr.begin(), tr.end());
    Random::shuffle(te.begin(), te.end());
    train = gather(tr);
    test = gather(te);
    LOG(INFO) << "Finished splitting data set in " << timer;
  }

  // ---- own binary cache (the reference serialises with boost + gzip, serialize.hpp:17-46): the dictionaries, then the columns --
  void write(std::ostream& o) const {
    CHECK(info_ != nullptr);
    io_detail::put<uint64_t>(o, columnar_tag());
    io_detail::put<uint64_t>(o, info_->feature_group_infos_.size());
    for (auto& g : info_->feature_group_infos_) g.write(o);
    io_detail::put<uint64_t>(o, size());
    io_detail::put<uint64_t>(o, label_.empty() ? 1 : 0);
    io_detail::put(o, uniform_label_);
    for (size_t g = 0; g < 2; ++g) o.write(reinterpret_cast<const char*>(col_[g].data()), (std::streamsize)(col_[g].size() * sizeof(uint32_t)));
    if (!label_.empty()) o.write(reinterpret_cast<const char*>(label_.data()), (std::streamsize)(label_.size() * sizeof(double)));
  }
  void read(std::istream& i) {
    uint64_t tag = 0; io_detail::get(i, tag);
    CHECK_EQ(tag, columnar_tag()) << "cache written by an older build of this layer: run --task=prepare again";
    info_ = std::make_shared<DataInfo>();
    uint64_t ng = 0; io_detail::get(i, ng);
    CHECK_EQ(ng, uint64_t(2));
    info_->feature_group_infos_.resize(ng);
    for (auto& g : info_->feature_group_infos_) g.read(i);
    uint64_t n = 0, uniform = 0; io_detail::get(i, n); io_detail::get(i, uniform); io_detail::get(i, uniform_label_);
    CHECK(i.good()) << "truncated cache (header)";
    {
      // a corrupt count must not drive a huge resize, and a truncated file must not leave zero-filled columns behind (user 0 / item 0
      // are valid ids: they would train silently on wrong data)
      const std::istream::pos_type here = i.tellg();
      if (here != std::istream::pos_type(-1)) {
        i.seekg(0, std::ios::end);
        const uint64_t left = static_cast<uint64_t>(i.tellg() - here);
        i.seekg(here);
        const uint64_t need = n * (2 * sizeof(uint32_t) + (uniform ? 0 : sizeof(double)));
        CHECK(n <= left && need <= left) << "cache claims " << n << " ratings but only " << left << " bytes follow: truncated or corrupt";
      }
    }
    for (size_t g = 0; g < 2; ++g) {
      col_[g].resize(n);
      i.read(reinterpret_cast<char*>(col_[g].data()), (std::streamsize)(n * sizeof(uint32_t)));
      CHECK(i.good() && static_cast<uint64_t>(i.gcount()) == n * sizeof(uint32_t)) << "truncated cache (column " << g << ")";
    }
    label_.clear();
    if (!uniform) {
      label_.resize(n);
      i.read(reinterpret_cast<char*>(label_.data()), (std::streamsize)(n * sizeof(double)));
      CHECK(i.good() && static_cast<uint64_t>(i.gcount()) == n * sizeof(double)) << "truncated cache (labels)";
    }
    for (size_t g = 0; g < 2; ++g) {
      const size_t dim = info_->feature_group_infos_[g].size();
      for (uint32_t v : col_[g]) CHECK_LT(static_cast<size_t>(v), dim) << "cache holds an id outside its dictionary (group " << g << ")";
    }
    finalize_dimensions();
    generation_ = next_generation();
  }

  friend std::ostream& operator<<(std::ostream& o, const Data& d) {
    o << "\nData set summary : \n\tNum of Instance: " << d.size() << "\n";
    if (d.info_) {
      o << "\tNum of feature groups: " << d.info_->feature_group_infos_.size() << "\n\tTotal feature dimensions: "
        << d.info_->total_dimensions_ << "\n";
      for (size_t g = 0; g < d.info_->feature_group_infos_.size(); ++g)
        o << "\tFeature group " << g << " -> size " << d.info_->feature_group_infos_[g].size() << "\n";
    }
    o << "Head of the data set:\n";
    for (size_t k = 0; k < std::min<size_t>(10, d.size()); ++k) o << "  " << d.at(k) << "\n";
    return o;
  }

 private:
  static uint64_t columnar_tag() { return 0x324C4F4345414443ull; }   // "CDAECOL2"
  static uint64_t next_generation() { static std::atomic<uint64_t> g(1); return g.fetch_add(1); }
  void reserve(size_t n) { col_[0].reserve(n); col_[1].reserve(n); }
  void append(uint32_t g0, uint32_t g1, double label) {
    const size_t n = col_[0].size();
    if (n == 0) { uniform_label_ = label; label_.clear(); }
    else if (label_.empty() && label != uniform_label_) label_.assign(n, uniform_label_);   // first differing label: materialise
    col_[0].push_back(g0); col_[1].push_back(g1);
    if (!label_.empty()) label_.push_back(label);
  }
  void append(const Instance& ins) {
    CHECK_EQ(ins.num_feature_groups(), size_t(2));
    append(static_cast<uint32_t>(ins.get_feature_group_index(0, 0)), static_cast<uint32_t>(ins.get_feature_group_index(1, 0)), ins.label());
  }
  Data gather(const std::vector<size_t>& pos) const {
    std::vector<uint32_t> g0(pos.size()), g1(pos.size());
    std::vector<double> lab(label_.empty() ? 0 : pos.size());
    for (size_t k = 0; k < pos.size(); ++k) { g0[k] = col_[0][pos[k]]; g1[k] = col_[1][pos[k]]; }
    if (!label_.empty()) for (size_t k = 0; k < pos.size(); ++k) lab[k] = label_[pos[k]];
    return Data(std::move(g0), std::move(g1), std::move(lab), uniform_label_, info_);
  }
  void finalize_dimensions() {
    info_->total_dimensions_ = 0;
    info_->feature_group_global_idx_.assign(info_->feature_group_infos_.size(), 0);
    for (size_t g = 0; g < info_->feature_group_infos_.size(); ++g) {
      info_->feature_group_global_idx_[g] = info_->total_dimensions_;
      info_->total_dimensions_ += info_->feature_group_infos_[g].size();
    }
  }
  std::vector<uint32_t> col_[2];          // user ids, item ids
  std::vector<double> label_;             // empty while every label equals uniform_label_
  double uniform_label_ = 0.;
  std::shared_ptr<DataInfo> info_;
  uint64_t generation_ = 0;               // 0 = empty default object
};

}  // namespace libcf
#endif
