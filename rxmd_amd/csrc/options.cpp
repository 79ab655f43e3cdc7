// options.cpp -- parser and printer of the environment switches (options.def).  Host code: compiled without HIP, usable without a GPU.
#include "options.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rxmd {

static bool env_flag(const char *name) {
  const char *v = std::getenv(name);
  return v != nullptr && v[0] != '\0' && std::strcmp(v, "0") != 0;
}
static long long env_int(const char *name, long long def) {
  const char *v = std::getenv(name);
  return (v != nullptr && v[0] != '\0') ? std::strtoll(v, nullptr, 0) : def;
}
static double env_real(const char *name, double def) {
  const char *v = std::getenv(name);
  if (v == nullptr || v[0] == '\0') return def;
  const double x = std::atof(v);
  return x > 0.0 ? x : def;
}

// The RX_EXP_* rows exist -- as names, parsers and text -- only in the object compiled with -DRXMD_EXPERIMENTS (librxmd_hip_exp.so): the product
// library neither reads nor names a switch that skips work (tests/test_host_frontend.py greps its binary for them).
Options Options::from_env() {
  Options o;
#define RX_FLAG(f, env, text) o.f = env_flag(env);
#define RX_INT(f, env, def, text) o.f = env_int(env, def);
#define RX_REAL(f, env, def, text) o.f = env_real(env, def);
#ifdef RXMD_EXPERIMENTS
#define RX_EXP_FLAG(f, env, text) o.f = env_flag(env);
#define RX_EXP_INT(f, env, def, text) o.f = env_int(env, def);
#else
#define RX_EXP_FLAG(f, env, text)
#define RX_EXP_INT(f, env, def, text)
#endif
#include "options.def"
#undef RX_FLAG
#undef RX_INT
#undef RX_REAL
#undef RX_EXP_FLAG
#undef RX_EXP_INT
  return o;
}

std::string Options::describe() {
  std::string out;
  char buf[64];
  (void)buf;
  auto row = [&](const char *env, const std::string &def, const char *text, bool exp) {
    out += std::string("| `") + env + "`" + (exp ? " (exp)" : "") + " | " + def + " | " + text + " |\n";
  };
#define RX_FLAG(f, env, text) row(env, "off", text, false);
#define RX_INT(f, env, def, text) std::snprintf(buf, sizeof buf, "%lld", static_cast<long long>(def)); row(env, buf, text, false);
#define RX_REAL(f, env, def, text) std::snprintf(buf, sizeof buf, "%g", static_cast<double>(def)); row(env, buf, text, false);
#ifdef RXMD_EXPERIMENTS
#define RX_EXP_FLAG(f, env, text) row(env, "off", text, true);
#define RX_EXP_INT(f, env, def, text) std::snprintf(buf, sizeof buf, "%lld", static_cast<long long>(def)); row(env, buf, text, true);
#else
#define RX_EXP_FLAG(f, env, text)
#define RX_EXP_INT(f, env, def, text)
#endif
#include "options.def"
#undef RX_FLAG
#undef RX_INT
#undef RX_REAL
#undef RX_EXP_FLAG
#undef RX_EXP_INT
  return out;
}

}  // namespace rxmd

// C ABI (include/rxmd_hip.h): the table as text; returns the length needed (without the terminating 0), copies at most capacity - 1 characters
extern "C" int rxmd_host_describe_options(char *buf, int capacity) {
  const std::string s = rxmd::Options::describe();
  if (buf != nullptr && capacity > 0) {
    const size_t n = s.size() < static_cast<size_t>(capacity - 1) ? s.size() : static_cast<size_t>(capacity - 1);
    std::memcpy(buf, s.data(), n); buf[n] = '\0';
  }
  return static_cast<int>(s.size());
}
