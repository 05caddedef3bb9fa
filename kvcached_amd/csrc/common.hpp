// common.hpp — logging, errors and small helpers shared by the native core.
#pragma once

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace kvc {

using page_id_t = int64_t;
using offset_t = int64_t;

// Same VA hint as the reference (csrc/inc/constants.hpp:18) so tensors land at the same
// addresses in both implementations; nothing depends on the value.
static constexpr uintptr_t kStartAddr = 0x1f000000000ull;
static constexpr size_t kBasePage = 2ull << 20; // 2 MiB: the unit every page size is a multiple of

// ---- logging: env KVCACHED_LOG_LEVEL, default WARNING, to stderr (reference: csrc/inc/gpu_utils.hpp:45-75)
enum LogLevel { LOG_DEBUG = 0, LOG_INFO = 1, LOG_WARNING = 2, LOG_ERROR = 3 };

inline int log_threshold() {
  static const int lvl = []() {
    const char *e = std::getenv("KVCACHED_LOG_LEVEL");
    if (!e) return (int)LOG_WARNING;
    std::string s(e);
    for (auto &c : s) c = (char)toupper(c);
    if (s == "DEBUG") return (int)LOG_DEBUG;
    if (s == "INFO") return (int)LOG_INFO;
    if (s == "ERROR") return (int)LOG_ERROR;
    return (int)LOG_WARNING;
  }();
  return lvl;
}

inline void logf(LogLevel lvl, const char *file, int line, const char *fmt, ...) {
  if ((int)lvl < log_threshold()) return;
  static const char *names[] = {"DEBUG", "INFO", "WARNING", "ERROR"};
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  const char *base = strrchr(file, '/');
  fprintf(stderr, "[kvcached_amd][%s][%s:%d] %s\n", names[lvl], base ? base + 1 : file, line, buf);
}
#define KVC_LOG(lvl, ...) ::kvc::logf(::kvc::lvl, __FILE__, __LINE__, __VA_ARGS__)

// ---- errors
struct GpuError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct NoPagesError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct InvalidError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct NoGpuError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct CallbackError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

inline int64_t env_i64(const char *name, int64_t dflt) {
  const char *e = std::getenv(name);
  return (e && *e) ? std::atoll(e) : dflt;
}
inline bool env_bool(const char *name, bool dflt) {
  const char *e = std::getenv(name);
  if (!e || !*e) return dflt;
  std::string s(e);
  for (auto &c : s) c = (char)tolower(c);
  return s == "1" || s == "true" || s == "yes" || s == "on";
}

// Test hooks: switches that REMOVE a safety step (the TLB invalidation, the rewrite of split mappings, a self test) so that
// the tests can show they would notice. They exist only in the build the tests load for that purpose (-DKVC_TEST_HOOKS,
// kvcached_amd/_testhooks/libkvcached_amd.so); the shipped library contains neither the code nor the names.
#ifdef KVC_TEST_HOOKS
inline bool test_hook_on(const char *name) {
  const bool on = env_bool(name, false);
  if (on) {
    static const char *reported[16] = {};
    for (auto &r : reported) {
      if (r == name) break; // (string literals: one address per call site is enough to keep the log short)
      if (!r) {
        r = name;
        KVC_LOG(LOG_ERROR, "TEST HOOK %s is active: this library build removes safety steps on request and must never be deployed", name);
        break;
      }
    }
  }
  return on;
}
#define KVC_TEST_HOOK(name) ::kvc::test_hook_on("KVCACHED_TEST_" name)
#else
#define KVC_TEST_HOOK(name) false
#endif

// "cuda", "cuda:N", "hip[:N]" -> gpu index (-1 = current device); "cpu" -> is_gpu=false.
struct DeviceSpec {
  bool is_gpu = false;
  int index = -1;
};
inline DeviceSpec parse_device(const std::string &s) {
  DeviceSpec d;
  std::string low = s;
  for (auto &c : low) c = (char)tolower(c);
  auto colon = low.find(':');
  std::string kind = low.substr(0, colon);
  if (kind == "cpu") return d;
  if (kind != "cuda" && kind != "hip") throw InvalidError("Unsupported device string: " + s);
  d.is_gpu = true;
  if (colon != std::string::npos) d.index = std::atoi(low.c_str() + colon + 1);
  return d;
}

} // namespace kvc
