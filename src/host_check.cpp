// Compile-and-link check of the host layer, and a tiny end-to-end driver used by the tests:
//   host_check                      -> exercises the CPU-only pieces (flags, logging, Data, split, cache, heap, metrics)
//   host_check --run_cdae=true ...  -> trains CDAE through Solver<CDAE> on a GPU (needs libcdae_hip.so + a device)
#include <glog/logging.h>
#include <gflags/gflags.h>

#include <base/data.hpp>
#include <base/heap.hpp>
#include <base/io.hpp>
#include <base/random.hpp>
#include <model/recsys/cdae.hpp>
#include <model/recsys/popularity.hpp>
#include <solver/solver.hpp>

DEFINE_string(input_file, "", "user item text file (header line skipped)");
DEFINE_bool(run_cdae, false, "train CDAE on the GPU");
DEFINE_int32(num_dim, 8, "latent dimensions");
DEFINE_int32(iters, 2, "epochs");
DEFINE_string(loss_type, "CE", "SQUARE or CE");
DEFINE_double(cratio, 0.5, "corruption ratio");
DEFINE_string(csr_only, "", "load this Data cache, time Data::to_csr (what a model's reset() derives for the device) and exit");
DEFINE_string(dump_csr, "", "write the train / test rows Data::to_csr produces to <path>.train / <path>.test (int64 rows+1, then uint32 cols)");
DEFINE_string(movielens_sample, "", "a user::item::rating::time file: run the checks of the reference's test/data_test.hpp:17-62 on it and exit");

int main(int argc, char* argv[]) {
  using namespace libcf;
  gflags::ParseCommandLineFlags(&argc, &argv, true);

  // heap semantics (reference test/heap_test.hpp:66-86)
  Heap<std::pair<size_t, double>> h(sort_by_second_desc<size_t, double>);
  const std::pair<size_t, double> items[] = {{10, 10.}, {20, 20.}, {30, 30.}, {5, 5.}, {15, 15.}};
  for (auto& p : items) { if (h.size() < 3) h.push(p); else h.push_and_pop(p); }
  auto sorted = h.get_sorted_data_copy();
  CHECK_EQ(sorted[0].first, size_t(30)); CHECK_EQ(sorted[1].first, size_t(20)); CHECK_EQ(sorted[2].first, size_t(15));
  CHECK_EQ(split_line("a  b c ", " ").size(), size_t(3));

  // metrics (reference evaluation.hpp:183-219)
  std::unordered_map<size_t, double> truth{{9, 1.}, {3, 1.}, {100, 1.}};
  auto r = TOPN_Evaluation<Popularity>::evaluate_rec_list({5, 9, 1, 7, 3, 2, 8, 4, 6, 0}, truth);
  CHECK(r[2] > 0.199 && r[2] < 0.201) << r[2];
  CHECK(r[5] > 0.666 && r[5] < 0.667) << r[5];

  if (!FLAGS_csr_only.empty()) {                  // tools/ingest_bench.py: the host part of reset() at scale
    Timer t_load;
    Data d;
    load(FLAGS_csr_only, d);
    const double load_s = t_load.elapsed();
    Timer t_csr;
    std::vector<int64_t> row_ptr; std::vector<uint32_t> col;
    d.to_csr(0, 1, row_ptr, col);
    LOG(INFO) << "csr_only: " << d.size() << " ratings, " << row_ptr.size() - 1 << " rows, " << col.size() << " unique; load "
              << load_s << " s, to_csr " << t_csr.elapsed() << " s";
    return 0;
  }
  if (!FLAGS_movielens_sample.empty()) {
    // What the reference's own data test asserts (test/data_test.hpp:17-62) on its fixture test/test_data/sample_movielens_data.txt:
    // the "::" line parser gives four fields per line, every instance has two features (user, item), the file holds 200 instances,
    // the cache round-trips, and random_split(0.3) leaves size * 0.7 / size * 0.3 instances.
    auto line_parser = [&](const std::string& line) {
      auto rets = split_line(line, ": ");
      CHECK_EQ(rets.size(), size_t(4));
      return std::vector<std::string>(rets.begin(), rets.begin() + 3);
    };
    Data data;
    data.load(FLAGS_movielens_sample, RECSYS, line_parser);
    const std::string cache = FLAGS_movielens_sample + ".bin";
    save(data, cache);
    Data data1;
    load(cache, data1);
    CHECK_EQ(data1.size(), data.size());
    size_t cnt = 0;
    double labels = 0.;
    for (auto it = data1.begin(); it != data1.end(); ++it) {
      CHECK_EQ((*it).size(), size_t(2));                          // data_test.hpp:48: two features per instance
      CHECK_EQ((*it).get_feature_group_index(0, 0), data.at(cnt).get_feature_group_index(0, 0));
      CHECK_EQ((*it).get_feature_group_index(1, 0), data.at(cnt).get_feature_group_index(1, 0));
      labels += (*it).label();
      ++cnt;
    }
    Data train, test, train2, test2;
    data.random_split_by_feature_group(train, test, 0, 0.3);     // data_test.hpp:56
    CHECK_EQ(train.size() + test.size(), data.size());
    data.random_split(train2, test2, 0.3);                       // data_test.hpp:57-59
    LOG(INFO) << "movielens sample: instances " << cnt << " users " << data.feature_group_total_dimension(0) << " items "
              << data.feature_group_total_dimension(1) << " label_sum " << labels << " by_user_split " << train.size() << " " << test.size()
              << " random_split " << train2.size() << " " << test2.size();
    return 0;
  }
  if (FLAGS_input_file.empty()) { LOG(INFO) << "host layer OK (no input file given)"; return 0; }

  auto parser = [&](const std::string& line) {
    auto f = split_line(line, " ");
    CHECK_EQ(f.size(), size_t(2));
    return std::vector<std::string>{f[0], f[1], "1"};
  };
  Data data;
  data.load(FLAGS_input_file, RECSYS, parser, true);
  const std::string cache = FLAGS_input_file + ".bin";
  save(data, cache);
  Data again;
  load(cache, again);
  CHECK_EQ(again.size(), data.size());
  CHECK_EQ(again.feature_group_total_dimension(1), data.feature_group_total_dimension(1));
  Random::seed(20141119);
  Data train, test;
  again.random_split_by_feature_group(train, test, 0, 0.2);
  CHECK_EQ(train.size() + test.size(), data.size());
  if (!FLAGS_dump_csr.empty()) {
    // the caches the split would write, and the CSR a model's reset() derives from them (tests pin both against numpy)
    save(train, FLAGS_dump_csr + ".train.bin");
    save(test, FLAGS_dump_csr + ".test.bin");
    for (int which = 0; which < 2; ++which) {
      std::vector<int64_t> row_ptr; std::vector<uint32_t> col;
      (which ? test : train).to_csr(0, 1, row_ptr, col);
      std::ofstream out(FLAGS_dump_csr + (which ? ".test" : ".train"), std::ios::binary);
      const uint64_t rows = row_ptr.size() - 1;
      out.write(reinterpret_cast<const char*>(&rows), sizeof rows);
      out.write(reinterpret_cast<const char*>(row_ptr.data()), (std::streamsize)(row_ptr.size() * sizeof(int64_t)));
      out.write(reinterpret_cast<const char*>(col.data()), (std::streamsize)(col.size() * sizeof(uint32_t)));
    }
  }
  {
    Popularity pop;
    Solver<Popularity> solver(pop);
    solver.train(train, test, {TOPN});
  }
  if (FLAGS_run_cdae) {
    CDAEConfig cfg;
    cfg.num_dim = FLAGS_num_dim;
    cfg.corruption_ratio = FLAGS_cratio;
    cfg.scaled = FLAGS_cratio > 0;
    cfg.beta = 1.;
    cfg.lt = FLAGS_loss_type == "SQUARE" ? SQUARE : CROSS_ENTROPY;
    CDAE model(cfg);
    Solver<CDAE> solver(model, FLAGS_iters);
    solver.train(train, test, {TOPN});
    // recommend() with the train row is the precomputed table; with any other rated set the hidden layer is encoded from
    // THAT set and exactly it is excluded (cdae.hpp:167-179)
    std::shared_ptr<CDAE> trained = solver.get_model();
    auto train_sets = train.get_feature_pair_label_hashtable(0, 1);
    const std::unordered_map<size_t, double>& row = train_sets.begin()->second;
    const size_t uid = train_sets.begin()->first;
    CHECK(trained->recommend(uid, 10, row) == trained->recommend_train_row(uid, 10));
    std::unordered_map<size_t, double> fewer(row);
    const size_t dropped = fewer.begin()->first;
    fewer.erase(fewer.begin());
    const std::vector<size_t> rec = trained->recommend(uid, 10, fewer);
    CHECK_EQ(rec.size(), size_t(10));
    for (size_t iid : rec) CHECK(!fewer.count(iid)) << "a rated item was recommended";
    LOG(INFO) << "explicit rated set OK (dropped item " << dropped << ")";
  }
  LOG(INFO) << "host layer OK";
  return 0;
}
