/* memgetinfo_feed.c — test-input injection for oracle/gen_golden.py, build container only.
 *
 * The reference's PageAllocator::get_avail_physical_pages (csrc/page_allocator.cpp:442-455) reads hipMemGetInfo and does
 * its arithmetic on the reading. This container has the HIP runtime but no GPU, so the call fails before the arithmetic is
 * reached. Preloaded (LD_PRELOAD) into the child process that drives the REAL reference .so, this file answers that ONE call
 * with the (free, total) pair the generator put in the environment - the reading becomes an input of the golden table
 * (tests/golden/avail_physical_pages.json) and everything after it is the reference's own code. Nothing else of HIP is
 * touched or replaced; the product and its tests never load this. */
#include <stddef.h>
#include <stdlib.h>

int hipMemGetInfo(size_t *free_bytes, size_t *total_bytes) {
  const char *f = getenv("KVC_FEED_FREE"), *t = getenv("KVC_FEED_TOTAL");
  if (!f || !t) return 1; /* hipErrorInvalidValue */
  *free_bytes = (size_t)strtoull(f, NULL, 10);
  *total_bytes = (size_t)strtoull(t, NULL, 10);
  return 0; /* hipSuccess */
}
