// what the first queue of a process costs, by how it is asked for; env variants are set by the calling script
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k(int *p) { if (p) *p = 1; }
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0, n = 0;
  double t0 = now();
  hipGetDeviceCount(&n);
  double t1 = now();
  hipSetDevice(0);
  int cu = 0; hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0);
  double t2 = now();
  hipStream_t s = nullptr;
  if (mode == 0) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  else if (mode == 2) hipStreamCreate(&s);
  else if (mode == 3) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0);
  else if (mode == 4) { hipFuncAttributes a; hipFuncGetAttributes(&a, (const void *)k); }
  double t3 = now();
  if (mode == 4) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double t3b = now();
  int *d; hipMalloc(&d, 4);
  double t4 = now();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  double t5 = now();
  hipStreamSynchronize(s);
  double t6 = now();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  double t7 = now();
  printf("{\"mode\": %d, \"env\": \"%s\", \"init\": %.4f, \"setdev\": %.4f, \"stepA\": %.4f, \"stream_after_A\": %.4f, \"malloc\": %.4f, \"launch1\": %.4f, \"sync1\": %.4f, \"launch2+sync\": %.4f, \"total\": %.4f}\n",
         mode, getenv("PROBE_ENV") ? getenv("PROBE_ENV") : "", t1 - t0, t2 - t1, t3 - t2, t3b - t3, t4 - t3b, t5 - t4, t6 - t5, t7 - t6, t7 - t0);
  fflush(stdout);
  _exit(0);
}
