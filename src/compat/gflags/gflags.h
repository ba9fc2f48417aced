// Minimal command-line flags with the gflags spellings apps/yelp uses: DEFINE_{string,int32,double,bool},
// FLAGS_*, gflags::SetUsageMessage, gflags::ParseCommandLineFlags(&argc, &argv, true).  Accepts
// --flag=value, --flag value, --flag (bool), --noflag, and -flag forms.  No gflags in this image; with the
// real library installed, drop -Isrc/compat and link -lgflags instead.
#ifndef CDAE_COMPAT_GFLAGS_GFLAGS_H_
#define CDAE_COMPAT_GFLAGS_GFLAGS_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

namespace gflags {
struct FlagInfo { int kind; void* ptr; std::string help; };   // kind: 0 string, 1 int32, 2 double, 3 bool
inline std::map<std::string, FlagInfo>& registry() { static std::map<std::string, FlagInfo> r; return r; }
inline std::string& usage() { static std::string u; return u; }
struct Registrar {
  Registrar(const char* name, int kind, void* ptr, const char* help) { registry()[name] = FlagInfo{kind, ptr, help}; }
};
inline void SetUsageMessage(const std::string& u) { usage() = u; }

inline bool parse_bool(const std::string& v, bool* out) {
  if (v == "true" || v == "1" || v == "yes" || v == "t" || v == "y") { *out = true; return true; }
  if (v == "false" || v == "0" || v == "no" || v == "f" || v == "n") { *out = false; return true; }
  return false;
}
inline void die(const std::string& msg) { std::fprintf(stderr, "ERROR: %s\n", msg.c_str()); std::exit(1); }

inline void assign(const std::string& name, FlagInfo& f, const std::string& v) {
  char* end = nullptr;
  switch (f.kind) {
    case 0: *static_cast<std::string*>(f.ptr) = v; break;
    case 1: { long x = std::strtol(v.c_str(), &end, 10); if (v.empty() || *end) die("illegal value '" + v + "' for int32 flag --" + name);
              *static_cast<int32_t*>(f.ptr) = (int32_t)x; break; }
    case 2: { double x = std::strtod(v.c_str(), &end); if (v.empty() || *end) die("illegal value '" + v + "' for double flag --" + name);
              *static_cast<double*>(f.ptr) = x; break; }
    default: { bool b; if (!parse_bool(v, &b)) die("illegal value '" + v + "' for bool flag --" + name);
               *static_cast<bool*>(f.ptr) = b; }
  }
}

inline uint32_t ParseCommandLineFlags(int* argc, char*** argv, bool remove_flags) {
  int out = 1;
  char** av = *argv;
  for (int i = 1; i < *argc; ++i) {
    std::string a = av[i];
    if (a == "--") { for (int j = i + 1; j < *argc; ++j) av[out++] = av[j]; break; }
    if (a.size() < 2 || a[0] != '-') { av[out++] = av[i]; continue; }
    std::string body = a.substr(a[1] == '-' ? 2 : 1), value;
    bool has_value = false;
    size_t eq = body.find('=');
    if (eq != std::string::npos) { value = body.substr(eq + 1); body = body.substr(0, eq); has_value = true; }
    if (body == "help" || body == "helpshort") {
      std::printf("%s\n", usage().c_str());
      for (auto& kv : registry()) std::printf("  --%s  %s\n", kv.first.c_str(), kv.second.help.c_str());
      std::exit(0);
    }
    auto it = registry().find(body);
    if (it == registry().end() && body.compare(0, 2, "no") == 0) {
      auto neg = registry().find(body.substr(2));
      if (neg != registry().end() && neg->second.kind == 3 && !has_value) { *static_cast<bool*>(neg->second.ptr) = false; continue; }
    }
    if (it == registry().end()) die("unknown command line flag '" + body + "'");
    if (!has_value) {
      if (it->second.kind == 3) { *static_cast<bool*>(it->second.ptr) = true; continue; }
      if (i + 1 >= *argc) die("flag '--" + body + "' is missing its argument");
      value = av[++i];
    }
    assign(body, it->second, value);
  }
  if (remove_flags) *argc = out;
  return out;
}
}  // namespace gflags
namespace google { using gflags::ParseCommandLineFlags; using gflags::SetUsageMessage; }

#define CDAE_DEFINE_FLAG(type, kind, name, dflt, help) \
  type FLAGS_##name = dflt;                            \
  static ::gflags::Registrar cdae_flag_registrar_##name(#name, kind, &FLAGS_##name, help)
#define DEFINE_string(name, dflt, help) CDAE_DEFINE_FLAG(std::string, 0, name, dflt, help)
#define DEFINE_int32(name, dflt, help) CDAE_DEFINE_FLAG(int32_t, 1, name, dflt, help)
#define DEFINE_double(name, dflt, help) CDAE_DEFINE_FLAG(double, 2, name, dflt, help)
#define DEFINE_bool(name, dflt, help) CDAE_DEFINE_FLAG(bool, 3, name, dflt, help)
#define DECLARE_string(name) extern std::string FLAGS_##name
#define DECLARE_int32(name) extern int32_t FLAGS_##name
#define DECLARE_double(name) extern double FLAGS_##name
#define DECLARE_bool(name) extern bool FLAGS_##name

#endif  // CDAE_COMPAT_GFLAGS_GFLAGS_H_
