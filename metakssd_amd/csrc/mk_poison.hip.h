/* mk_poison.hip.h -- the library's device and pinned-host allocations go through these two wrappers so that MK_POISON
 * (host/mk_host_internal.h) can fill them before their first use.  Off (the default) they are hipMalloc / hipHostMalloc. */
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include "host/mk_host_internal.h"

template <class T>
static inline hipError_t mk_dev_alloc(T **p, size_t bytes) {
  hipError_t r = hipMalloc((void **)p, bytes);
  const int pz = mk_poison_byte();
  if (r == hipSuccess && pz >= 0 && bytes) {
    r = hipMemset((void *)*p, pz, bytes);
    if (r == hipSuccess) r = hipDeviceSynchronize(); /* (the engines' queues are not ordered with the default stream) */
  }
  return r;
}
template <class T>
static inline hipError_t mk_pin_alloc(T **p, size_t bytes, unsigned flags) {
  hipError_t r = hipHostMalloc((void **)p, bytes, flags);
  const int pz = mk_poison_byte();
  if (r == hipSuccess && pz >= 0 && bytes) memset((void *)*p, pz, bytes);
  return r;
}
/* a buffer taken back for another use: filled on the stream that will work on it next (device), or here (pinned) */
static inline hipError_t mk_dev_repoison(void *p, size_t bytes, hipStream_t s) {
  const int pz = mk_poison_byte();
  if (pz < 0 || !p || !bytes) return hipSuccess;
  return hipMemsetAsync(p, pz, bytes, s);
}
static inline void mk_pin_repoison(void *p, size_t bytes) {
  const int pz = mk_poison_byte();
  if (pz >= 0 && p && bytes) memset(p, pz, bytes);
}
