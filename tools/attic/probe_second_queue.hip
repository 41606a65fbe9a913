// (5) what the first host-to-device copy of a process costs; (6) a second queue made by another thread while the first works
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k(int *p) { if (p) *p = 1; }
static double t_b0, t_b1;
static hipStream_t s2;
static void *mk2(void *) { hipSetDevice(0); t_b0 = now(); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); t_b1 = now(); return nullptr; }
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 5, n = 0;
  hipGetDeviceCount(&n); hipSetDevice(0);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int *d; hipMalloc(&d, 1 << 20);
  void *h; hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s);
  if (mode == 5) {
    double a = now(); hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s); double b = now(); hipStreamSynchronize(s); double c = now();
    hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double e = now();
    hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double f = now();
    hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double g = now();
    printf("{\"mode\": 5, \"env\": \"%s\", \"h2d_first_call\": %.4f, \"h2d_first_sync\": %.4f, \"h2d_second\": %.4f, \"d2h_first\": %.4f, \"d2h_second\": %.4f}\n",
           getenv("PROBE_ENV") ? getenv("PROBE_ENV") : "", b - a, c - b, e - c, f - e, g - f);
  } else {
    pthread_t th; double t0 = now(); pthread_create(&th, nullptr, mk2, nullptr);
    double worst = 0; int iters = 0;
    while (now() - t0 < 0.05) { double a = now(); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s); double b = now(); if (b - a > worst) worst = b - a; iters++; }
    pthread_join(th, nullptr);
    printf("{\"mode\": 6, \"second_stream_create\": %.4f, \"started_after\": %.4f, \"worst_launch_sync_on_first\": %.5f, \"iters\": %d}\n", t_b1 - t_b0, t_b0 - t0, worst, iters);
  }
  fflush(stdout); _exit(0);
}
