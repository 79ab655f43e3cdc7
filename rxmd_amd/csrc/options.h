// options.h -- every switch the environment can throw at the library, read ONCE when an engine is created (rxmd_hip_create).
// options.def is the table; README.md lists it from there (rxmd_host_describe_options, tests/test_host_frontend.py).  Host code, no HIP.
#pragma once
#include <string>

namespace rxmd {

struct Options {
#define RX_FLAG(f, env, text) bool f = false;
#define RX_INT(f, env, def, text) long long f = def;
#define RX_REAL(f, env, def, text) double f = def;
#define RX_EXP_FLAG(f, env, text) bool f = false;
#define RX_EXP_INT(f, env, def, text) long long f = def;
#include "options.def"
#undef RX_FLAG
#undef RX_INT
#undef RX_REAL
#undef RX_EXP_FLAG
#undef RX_EXP_INT
  static Options from_env();                     // (the RX_EXP_* rows are read only by the object built with -DRXMD_EXPERIMENTS)
  static std::string describe();                 // the table as markdown rows: | `ENV` | default | meaning |
};

}  // namespace rxmd
