// when does hipEventSynchronize return for an event that has MORE work queued behind it on the same stream?
//   stream: [kernel A, 3 ms] [copy D2H 64 B] [event] [kernel B, 3 ms]   -> synchronise the event, then the stream
// with plain, disable-timing and blocking-sync events, and with hipStreamQuery polling instead
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void spin(unsigned long long cycles, int *out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out) *out = 1;
}
int main() {
  hipSetDevice(0);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int *d; hipMalloc(&d, 64); int *h; hipHostMalloc(&h, 64, hipHostMallocDefault);
  const unsigned long long c3ms = 300000ull; /* wall_clock64 ticks at 100 MHz */
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000ull, d); hipStreamSynchronize(s);
  const unsigned flags[3] = {hipEventDefault, hipEventDisableTiming, hipEventBlockingSync | hipEventDisableTiming};
  const char *names[3] = {"default", "disable_timing", "blocking_sync"};
  for (int rep = 0; rep < 2; rep++)
  for (int f = 0; f < 3; f++) {
    hipEvent_t ev; hipEventCreateWithFlags(&ev, flags[f]);
    const double t0 = now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, c3ms, d);
    hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s);
    hipEventRecord(ev, s);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, c3ms, d);
    const double t1 = now();
    hipEventSynchronize(ev);
    const double t2 = now();
    hipStreamSynchronize(s);
    const double t3 = now();
    printf("{\"event\": \"%s\", \"queued_ms\": %.3f, \"event_sync_returned_after_ms\": %.3f, \"stream_done_after_ms\": %.3f}\n", names[f], (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3);
    hipEventDestroy(ev);
  }
  { /* polling hipEventQuery */
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const double t0 = now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, c3ms, d);
    hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s);
    hipEventRecord(ev, s);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, c3ms, d);
    while (hipEventQuery(ev) == hipErrorNotReady) { }
    const double t2 = now();
    hipStreamSynchronize(s);
    printf("{\"event\": \"query_poll\", \"event_done_seen_after_ms\": %.3f, \"stream_done_after_ms\": %.3f}\n", (t2 - t0) * 1e3, (now() - t0) * 1e3);
  }
  fflush(stdout); _exit(0);
}
