// export_diag: which step of "export a MAPPED handle, import it, map the import elsewhere" dies?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#define STEP(e)                                                                                    \
  do {                                                                                             \
    printf("  %s ... ", #e);                                                                       \
    fflush(stdout);                                                                                \
    hipError_t s_ = (e);                                                                           \
    printf("%s\n", hipGetErrorString(s_));                                                         \
    fflush(stdout);                                                                                \
  } while (0)
static const size_t PAGE = 2u << 20;
int main(int argc, char **argv) {
  int variant = argc > 1 ? atoi(argv[1]) : 0;
  hipSetDevice(0);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  prop.location.type = hipMemLocationTypeDevice;
  hipMemAllocationProp plain = prop;
  plain.requestedHandleType = hipMemHandleTypeNone;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  char *va = nullptr;
  STEP(hipMemAddressReserve((void **)&va, 16 * PAGE, PAGE, nullptr, 0));
  hipMemGenericAllocationHandle_t h, hi, z;
  printf("variant %d\n", variant);
  STEP(hipMemCreate(&h, PAGE, &prop, 0));
  STEP(hipMemMap(va, PAGE, 0, h, 0));
  STEP(hipMemSetAccess(va, PAGE, &acc, 1));
  int fd = -1;
  STEP(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
  printf("  fd=%d\n", fd);
  STEP(hipMemImportFromShareableHandle(&hi, (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
  char *dst = va + 4 * PAGE;
  if (variant == 1) { // destination slot used to alias a shared zero page
    STEP(hipMemCreate(&z, PAGE, &plain, 0));
    STEP(hipMemMap(dst, PAGE, 0, z, 0));
    STEP(hipMemMap(dst + PAGE, PAGE, 0, z, 0));
    STEP(hipMemSetAccess(dst, 2 * PAGE, &acc, 1));
    STEP(hipMemUnmap(dst, PAGE));
  }
  STEP(hipMemMap(dst, PAGE, 0, hi, 0));
  STEP(hipMemSetAccess(dst, PAGE, &acc, 1));
  void *p; hipMalloc(&p, PAGE); hipFree(p); // TLB shootdown
  unsigned v = 0xabcd1234, r = 0;
  STEP(hipMemcpy(va + 64, &v, 4, hipMemcpyHostToDevice));
  STEP(hipMemcpy(&r, dst + 64, 4, hipMemcpyDeviceToHost));
  printf("  readback 0x%x shared=%d\n", r, r == v);
  STEP(hipMemUnmap(dst, PAGE));
  STEP(hipMemRelease(hi));
  STEP(hipMemUnmap(va, PAGE));
  STEP(hipMemRelease(h));
  close(fd);
  printf("variant %d done\n", variant);
  return 0;
}
