// unmap_trace.cpp — where does hipMemUnmap's time go? Maps 512 pooled 2 MiB pages, then unmaps them; meant to be
// run under `strace -f -T -e trace=ioctl` (time inside each ioctl) next to its own wall-clock numbers.
// build: hipcc --offload-arch=gfx950 -O2 -o unmap_trace unmap_trace.cpp -ldl ; run: LD_PRELOAD=./ioctl_timer.so ./unmap_trace
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/resource.h>
#include <sys/time.h>

#include <map>
#include <string>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));                              \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

// ---- in-process sampling profiler (the image has no perf and ptrace attach is not permitted): SIGPROF every 0.5 ms
// of CPU time while armed, 20 return addresses per sample, symbolised with dladdr at the end.
static const int kMaxSamples = 20000, kDepth = 20;
static void *g_frames[kMaxSamples][kDepth];
static int g_nframes[kMaxSamples];
static volatile int g_samples = 0;
static volatile int g_armed = 0;
static void on_prof(int) {
  if (!g_armed || g_samples >= kMaxSamples) return;
  int i = g_samples;
  g_nframes[i] = backtrace(g_frames[i], kDepth);
  g_samples = i + 1;
}
static void prof_start() {
  struct sigaction sa {};
  sa.sa_handler = on_prof;
  sa.sa_flags = SA_RESTART;
  sigaction(SIGPROF, &sa, nullptr);
  struct itimerval it {};
  it.it_interval.tv_usec = 500;
  it.it_value.tv_usec = 500;
  setitimer(ITIMER_PROF, &it, nullptr);
}
static std::string sym(void *a) {
  Dl_info di{};
  char buf[512];
  if (dladdr(a, &di) && di.dli_fname) {
    const char *base = strrchr(di.dli_fname, '/');
    snprintf(buf, sizeof buf, "%s!%s+0x%lx", base ? base + 1 : di.dli_fname, di.dli_sname ? di.dli_sname : "?",
             (unsigned long)((char *)a - (char *)(di.dli_sname ? di.dli_saddr : di.dli_fbase)));
  } else {
    snprintf(buf, sizeof buf, "%p", a);
  }
  return buf;
}
static void prof_report() {
  std::map<std::string, int> leaf, stack3;
  for (int i = 0; i < g_samples; i++) {
    // frames 0,1 are the handler and the signal trampoline
    std::string l = g_nframes[i] > 2 ? sym(g_frames[i][2]) : "?";
    leaf[l]++;
    std::string s3;
    for (int k = 2; k < g_nframes[i] && k < 9; k++) s3 += sym(g_frames[i][k]) + " <- ";
    stack3[s3]++;
  }
  fprintf(stderr, "--- %d samples while hipMemUnmap ran\n", g_samples);
  std::multimap<int, std::string> byn;
  for (auto &kv : leaf) byn.emplace(-kv.second, kv.first);
  int k = 0;
  for (auto &kv : byn)
    if (k++ < 25) fprintf(stderr, "  leaf %5.1f%%  %s\n", -100.0 * kv.first / g_samples, kv.second.c_str());
  byn.clear();
  for (auto &kv : stack3) byn.emplace(-kv.second, kv.first);
  k = 0;
  for (auto &kv : byn)
    if (k++ < 12) fprintf(stderr, "  stack %5.1f%%  %s\n", -100.0 * kv.first / g_samples, kv.second.c_str());
}

static void dump(const char *label) { // provided by ioctl_timer.so when preloaded
  auto f = (void (*)(const char *))dlsym(RTLD_DEFAULT, "ioctl_timer_dump");
  if (f) f(label);
}

static void cpu_split(const char *label, int n) { // user vs system CPU time of this thread since the last call
  static double lu = 0, ls = 0;
  struct rusage ru;
  getrusage(RUSAGE_THREAD, &ru);
  double u = ru.ru_utime.tv_sec * 1e6 + ru.ru_utime.tv_usec, s = ru.ru_stime.tv_sec * 1e6 + ru.ru_stime.tv_usec;
  fprintf(stderr, "    [%s] per page: user %.2f us, system %.2f us\n", label, (u - lu) / n, (s - ls) / n);
  lu = u;
  ls = s;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int rounds = argc > 2 ? atoi(argv[2]) : 2;
  const bool quiet = argc > 3; // long runs for a sampling profiler: no per-phase dumps
  const size_t PAGE = 2u << 20;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  char *va = nullptr;
  CK(hipMemAddressReserve((void **)&va, (size_t)n * PAGE, PAGE, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (int i = 0; i < n; i++) CK(hipMemCreate(&h[i], PAGE, &prop, 0));
  if (quiet) prof_start();
  dump("setup (reserve + create)");
  cpu_split("setup", n);
  for (int round = 0; round < rounds; round++) {
    double t0 = now_us();
    for (int i = 0; i < n; i++) CK(hipMemMap(va + (size_t)i * PAGE, PAGE, 0, h[i], 0));
    double t1 = now_us();
    if (!quiet) {
      dump("hipMemMap x n");
      cpu_split("map", n);
    }
    for (int i = 0; i < n; i++) CK(hipMemSetAccess(va + (size_t)i * PAGE, PAGE, &acc, 1));
    double t2 = now_us();
    if (!quiet) {
      dump("hipMemSetAccess x n");
      cpu_split("set_access", n);
    }
    g_armed = quiet;
    for (int i = 0; i < n; i++) CK(hipMemUnmap(va + (size_t)i * PAGE, PAGE));
    g_armed = 0;
    double t3 = now_us();
    if (!quiet) {
      dump("hipMemUnmap x n");
      cpu_split("unmap", n);
    }
    if (!quiet || round % 50 == 0) fprintf(stderr, "=== round %d done: map %.2f set_access %.2f unmap %.2f us/page\n", round, (t1 - t0) / n, (t2 - t1) / n,
            (t3 - t2) / n);
  }
  if (quiet) prof_report();
  return 0;
}
