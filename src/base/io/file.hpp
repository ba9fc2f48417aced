// Text-line reading, tokenising and the binary cache used by apps/yelp's prepare/split/test tasks.
// Reference: src/base/io/file.hpp, file_line_reader.hpp, file_utils.hpp:15-25 (split_line on any of the
// delimiter characters, empty tokens dropped), serialize.hpp:17-46 (save/load<T>).  The reference
// serialises with boost + gzip; neither exists here, so save()/load() write the object's own
// write(std::ostream&)/read(std::istream&) image behind a magic word.
#ifndef CDAE_HOST_BASE_IO_FILE_HPP_
#define CDAE_HOST_BASE_IO_FILE_HPP_

#include <cstdint>
#include <fstream>
#include <functional>
#include <string>
#include <vector>

#include <glog/logging.h>

namespace libcf {

inline std::vector<std::string> split_line(const std::string& line, const std::string& delimiters = " ") {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < line.size()) {
    const size_t a = line.find_first_not_of(delimiters, i);
    if (a == std::string::npos) break;
    const size_t b = line.find_first_of(delimiters, a);
    out.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a));
    if (b == std::string::npos) break;
    i = b + 1;
  }
  return out;
}

class FileLineReader {
 public:
  typedef std::function<void(const std::string&, size_t)> Callback;
  explicit FileLineReader(const std::string& filename) : filename_(filename) {}
  void set_line_callback(const Callback& cb) { cb_ = cb; }
  void load() {
    std::ifstream in(filename_);
    CHECK(in.good()) << "cannot open " << filename_;
    std::string line;
    size_t n = 0;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      cb_(line, n++);
    }
  }
 private:
  std::string filename_;
  Callback cb_;
};

namespace io_detail {
const uint64_t kMagic = 0x3145414443464C43ull;   // "CLFCDAE1"
template <class T> inline void put(std::ostream& o, const T& v) { o.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <class T> inline void get(std::istream& i, T& v) { i.read(reinterpret_cast<char*>(&v), sizeof(T)); }
inline void put_str(std::ostream& o, const std::string& s) { put<uint64_t>(o, s.size()); o.write(s.data(), (std::streamsize)s.size()); }
inline void get_str(std::istream& i, std::string& s) { uint64_t n = 0; get(i, n); s.resize(n); i.read(&s[0], (std::streamsize)n); }
}  // namespace io_detail

template <typename T>
inline void save(const T& t, const std::string& filename, bool /*binary_format*/ = true) {
  std::ofstream out(filename, std::ios::binary);
  CHECK(out.good()) << "cannot write " << filename;
  io_detail::put(out, io_detail::kMagic);
  t.write(out);
  CHECK(out.good()) << "write failed: " << filename;
}

template <typename T>
inline void load(const std::string& filename, T& t, bool /*binary_format*/ = true) {
  std::ifstream in(filename, std::ios::binary);
  CHECK(in.good()) << "cannot open " << filename;
  uint64_t magic = 0;
  io_detail::get(in, magic);
  CHECK_EQ(magic, io_detail::kMagic) << filename << " is not a cache written by this build";
  t.read(in);
  CHECK(!in.fail()) << "truncated cache " << filename;
}

}  // namespace libcf
#endif
