/* ioctl_timer.c — LD_PRELOAD shim: time spent inside every ioctl, per request code (the image has no strace/perf).
 * ioctl_timer_dump(label) prints and resets the table; unmap_trace.cpp calls it between phases.
 * build: gcc -O2 -shared -fPIC -o ioctl_timer.so ioctl_timer.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

#define SLOTS 64
static struct { unsigned long req; uint64_t n, ns; } tab[SLOTS];
static int (*real_ioctl)(int, unsigned long, ...);

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec;
}

int ioctl(int fd, unsigned long req, ...) {
  va_list ap;
  va_start(ap, req);
  void *arg = va_arg(ap, void *);
  va_end(ap);
  if (!real_ioctl) real_ioctl = (int (*)(int, unsigned long, ...))dlsym(RTLD_NEXT, "ioctl");
  uint64_t t0 = now_ns();
  int rc = real_ioctl(fd, req, arg);
  uint64_t dt = now_ns() - t0;
  for (int i = 0; i < SLOTS; i++) {
    if (tab[i].req == req || tab[i].n == 0) {
      tab[i].req = req;
      __atomic_add_fetch(&tab[i].n, 1, __ATOMIC_RELAXED);
      __atomic_add_fetch(&tab[i].ns, dt, __ATOMIC_RELAXED);
      break;
    }
  }
  return rc;
}

/* other syscalls the VMM path may make: counted under pseudo request codes */
#include <sys/mman.h>
#include <unistd.h>
static void tally(unsigned long req, uint64_t dt) {
  for (int i = 0; i < SLOTS; i++) {
    if (tab[i].req == req || tab[i].n == 0) {
      tab[i].req = req;
      __atomic_add_fetch(&tab[i].n, 1, __ATOMIC_RELAXED);
      __atomic_add_fetch(&tab[i].ns, dt, __ATOMIC_RELAXED);
      break;
    }
  }
}
#define PSEUDO_MMAP 0xffff0001ul
#define PSEUDO_MUNMAP 0xffff0002ul
#define PSEUDO_CLOSE 0xffff0003ul
#define PSEUDO_MPROTECT 0xffff0004ul
#define PSEUDO_MADVISE 0xffff0005ul
void *mmap(void *addr, size_t len, int prot, int flags, int fd, off_t off) {
  static void *(*real)(void *, size_t, int, int, int, off_t);
  if (!real) real = (void *(*)(void *, size_t, int, int, int, off_t))dlsym(RTLD_NEXT, "mmap");
  uint64_t t0 = now_ns();
  void *r = real(addr, len, prot, flags, fd, off);
  tally(PSEUDO_MMAP, now_ns() - t0);
  return r;
}
int munmap(void *addr, size_t len) {
  static int (*real)(void *, size_t);
  if (!real) real = (int (*)(void *, size_t))dlsym(RTLD_NEXT, "munmap");
  uint64_t t0 = now_ns();
  int r = real(addr, len);
  tally(PSEUDO_MUNMAP, now_ns() - t0);
  return r;
}
int close(int fd) {
  static int (*real)(int);
  if (!real) real = (int (*)(int))dlsym(RTLD_NEXT, "close");
  uint64_t t0 = now_ns();
  int r = real(fd);
  tally(PSEUDO_CLOSE, now_ns() - t0);
  return r;
}
int mprotect(void *addr, size_t len, int prot) {
  static int (*real)(void *, size_t, int);
  if (!real) real = (int (*)(void *, size_t, int))dlsym(RTLD_NEXT, "mprotect");
  uint64_t t0 = now_ns();
  int r = real(addr, len, prot);
  tally(PSEUDO_MPROTECT, now_ns() - t0);
  return r;
}
int madvise(void *addr, size_t len, int advice) {
  static int (*real)(void *, size_t, int);
  if (!real) real = (int (*)(void *, size_t, int))dlsym(RTLD_NEXT, "madvise");
  uint64_t t0 = now_ns();
  int r = real(addr, len, advice);
  tally(PSEUDO_MADVISE, now_ns() - t0);
  return r;
}

void ioctl_timer_dump(const char *label) {
  fprintf(stderr, "--- ioctls during: %s\n", label);
  for (int i = 0; i < SLOTS && tab[i].n; i++) {
    /* _IOC decoding: type (magic) in bits 8-15, nr in bits 0-7 */
    static const char *pseudo[] = {"", "mmap", "munmap", "close", "mprotect", "madvise"};
    if ((tab[i].req >> 16) == 0xffff)
      fprintf(stderr, "    %-34s  calls %8llu  total %10.1f us  avg %7.2f us\n", pseudo[tab[i].req & 0xf],
              (unsigned long long)tab[i].n, tab[i].ns / 1e3, tab[i].ns / 1e3 / tab[i].n);
    else
      fprintf(stderr, "    req 0x%08lx (magic '%c' nr 0x%02lx)  calls %8llu  total %10.1f us  avg %7.2f us\n", tab[i].req,
              (char)((tab[i].req >> 8) & 0xff), tab[i].req & 0xff, (unsigned long long)tab[i].n, tab[i].ns / 1e3,
              tab[i].ns / 1e3 / tab[i].n);
    tab[i].n = tab[i].ns = 0;
    tab[i].req = 0;
  }
}
