// Minimal logging front-end with the glog spellings the libcf-style host layer and apps/yelp use
// (LOG(INFO|WARNING|ERROR|FATAL), CHECK*, FLAGS_log_dir, google::InitGoogleLogging ...).  The image has no
// glog; with the real library installed, drop -Isrc/compat and link -lglog instead.
#ifndef CDAE_COMPAT_GLOG_LOGGING_H_
#define CDAE_COMPAT_GLOG_LOGGING_H_

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

namespace google {
enum LogSeverity { GLOG_INFO = 0, GLOG_WARNING = 1, GLOG_ERROR = 2, GLOG_FATAL = 3 };
typedef int LogSeverityInt;

struct LogState {
  std::string info_file;
  std::ofstream sink;
  static LogState& get() { static LogState s; return s; }
};
inline void InitGoogleLogging(const char*) {}
inline void SetLogDestination(int, const char* base) {
  LogState& s = LogState::get();
  s.info_file = base ? base : "";
  if (!s.info_file.empty()) s.sink.open(s.info_file, std::ios::app);   // silently falls back to stderr
}

class LogMessage {
 public:
  LogMessage(const char* file, int line, int severity) : severity_(severity) {
    static const char tag[] = {'I', 'W', 'E', 'F'};
    const char* slash = file;
    for (const char* p = file; *p; ++p) if (*p == '/') slash = p + 1;
    std::time_t t = std::time(nullptr);
    char buf[32];
    std::strftime(buf, sizeof buf, "%m%d %H:%M:%S", std::localtime(&t));
    stream_ << tag[severity] << buf << ' ' << slash << ':' << line << "] ";
  }
  ~LogMessage() {
    stream_ << '\n';
    const std::string s = stream_.str();
    LogState& st = LogState::get();
    if (st.sink.is_open()) { st.sink << s; st.sink.flush(); }
    if (!st.sink.is_open() || severity_ >= GLOG_WARNING) { std::fputs(s.c_str(), stderr); std::fflush(stderr); }
    if (severity_ == GLOG_FATAL) std::abort();
  }
  std::ostream& stream() { return stream_; }
 private:
  std::ostringstream stream_;
  int severity_;
};
struct Voidify { void operator&(std::ostream&) {} };
}  // namespace google

extern std::string FLAGS_log_dir;
#ifndef CDAE_COMPAT_GLOG_NO_DEFINE
#ifdef __GNUC__
__attribute__((weak))
#endif
std::string FLAGS_log_dir;
#endif

#define CDAE_LOG_INFO ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_INFO).stream()
#define CDAE_LOG_WARNING ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_WARNING).stream()
#define CDAE_LOG_ERROR ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_ERROR).stream()
#define CDAE_LOG_FATAL ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_FATAL).stream()
#define LOG(severity) CDAE_LOG_##severity

#define CHECK(cond) \
  (cond) ? (void)0 : ::google::Voidify() & ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_FATAL).stream() << "Check failed: " #cond " "
#define CDAE_CHECK_OP(a, b, op) \
  ((a)op(b)) ? (void)0      \
             : ::google::Voidify() & ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_FATAL).stream() \
                   << "Check failed: " #a " " #op " " #b " (" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) CDAE_CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CDAE_CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) CDAE_CHECK_OP(a, b, <)
#define CHECK_LE(a, b) CDAE_CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) CDAE_CHECK_OP(a, b, >)
#define CHECK_GE(a, b) CDAE_CHECK_OP(a, b, >=)

#endif  // CDAE_COMPAT_GLOG_LOGGING_H_
