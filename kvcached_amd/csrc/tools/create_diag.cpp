// create_diag: how does hipMemCreate's cost depend on the number and release order of live handles?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("FAIL %s -> %s\n", #e, hipGetErrorString(s_)); exit(1);} } while (0)
using clk = std::chrono::steady_clock;
static double us(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }
static const size_t PAGE = 2u << 20;
int main() {
  CK(hipSetDevice(0));
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
  std::vector<hipMemGenericAllocationHandle_t> h;
  auto create_n = [&](int n, const char *tag) {
    auto t0 = clk::now();
    for (int i = 0; i < n; i++) { hipMemGenericAllocationHandle_t x; CK(hipMemCreate(&x, PAGE, &prop, 0)); h.push_back(x); }
    printf("%-46s live=%6zu  create %.2f us each\n", tag, h.size(), us(t0) / n); fflush(stdout);
  };
  for (int k = 0; k < 10; k++) create_n(2000, "grow");
  // release the 4000 OLDEST, then create
  { auto t0 = clk::now(); for (int i = 0; i < 4000; i++) CK(hipMemRelease(h[i])); printf("release 4000 oldest: %.2f us each\n", us(t0) / 4000);
    h.erase(h.begin(), h.begin() + 4000); }
  create_n(1000, "after releasing 4000 oldest");
  create_n(1000, "  again");
  create_n(2000, "  again (hole used up?)");
  create_n(1000, "  again");
  // release the 4000 NEWEST, then create
  { auto t0 = clk::now(); for (int i = 0; i < 4000; i++) { CK(hipMemRelease(h.back())); h.pop_back(); } printf("release 4000 newest: %.2f us each\n", us(t0) / 4000); }
  create_n(1000, "after releasing 4000 newest");
  create_n(3000, "  again");
  // bigger handles: is the cost per handle or per byte?
  { std::vector<hipMemGenericAllocationHandle_t> big; auto t0 = clk::now();
    for (int i = 0; i < 200; i++) { hipMemGenericAllocationHandle_t x; CK(hipMemCreate(&x, 64 * PAGE, &prop, 0)); big.push_back(x); }
    printf("create 200 x 128 MiB handles with %zu small live: %.2f us each\n", h.size(), us(t0) / 200);
    for (auto x : big) CK(hipMemRelease(x)); }
  // mapped vs unmapped handles: does mapping them change the create cost of later ones?
  { char *va; CK(hipMemAddressReserve((void **)&va, h.size() * PAGE, PAGE, nullptr, 0));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    auto t0 = clk::now();
    for (size_t i = 0; i < h.size(); i++) { CK(hipMemMap(va + i * PAGE, PAGE, 0, h[i], 0)); CK(hipMemSetAccess(va + i * PAGE, PAGE, &acc, 1)); }
    printf("map+access of %zu handles: %.2f us each\n", h.size(), us(t0) / h.size());
    create_n(1000, "create with all of them mapped");
    t0 = clk::now();
    for (size_t i = 0; i + 1000 < h.size(); i++) CK(hipMemUnmap(va + i * PAGE, PAGE));
    printf("unmap: %.2f us each\n", us(t0) / (h.size() - 1000)); }
  auto t0 = clk::now(); size_t n = h.size();
  for (auto x : h) CK(hipMemRelease(x));
  printf("release all %zu: %.2f us each\n", n, us(t0) / n);
  return 0;
}
