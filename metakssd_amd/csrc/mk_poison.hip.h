/* mk_poison.hip.h -- the library's device and pinned-host allocations go through these two wrappers so that MK_POISON
 * (host/mk_host_internal.h) can fill them before their first use.  Off (the default) they are hipMalloc / hipHostMalloc. */
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include "host/mk_host_internal.h"

/* the fill of a fresh allocation runs on a stream of its own (non-blocking, one per device and process, made at the first poisoned allocation) and
 * is waited for there: nobody else knows the buffer yet, so nothing else has to be ordered with it.  NOT the default stream: that one is
 * implicitly ordered with every stream made without hipStreamNonBlocking -- the CU-masked queues of MK_OPT_SPLIT_CUS are such streams -- and a
 * device-wide wait in the middle of an engine's two-queue hand-over is not something an allocation should do. */
#include <mutex>
static inline hipError_t mk_poison_stream(hipStream_t *out) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  int dev = 0;
  hipError_t r = hipGetDevice(&dev);
  if (r != hipSuccess) return r;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> g(mu);
  if (!streams[dev]) {
    r = hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking);
    if (r != hipSuccess) { streams[dev] = nullptr; return r; }
  }
  *out = streams[dev];
  return hipSuccess;
}

template <class T>
static inline hipError_t mk_dev_alloc(T **p, size_t bytes) {
  hipError_t r = hipMalloc((void **)p, bytes);
  const int pz = mk_poison_byte();
  if (r == hipSuccess && pz >= 0 && bytes) {
    hipStream_t ps = nullptr;
    r = mk_poison_stream(&ps);
    if (r == hipSuccess) r = hipMemsetAsync((void *)*p, pz, bytes, ps);
    if (r == hipSuccess) r = hipStreamSynchronize(ps);
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
