// tools/probe_init.hip -- what the start-up of `metakssd dist` is made of on this box: HIP runtime start-up, device and
// pinned allocations of the sizes the engine and the FASTQ stream use, H2D rates by piece size.  Prints one JSON line.
//   hipcc -O2 --offload-arch=gfx950 tools/probe_init.hip -o tools/probe_init && tools/probe_init
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define T(name, expr) do { double a = now(); expr; double b = now(); printf("%s\"%s\": %.5f", first ? "" : ", ", name, b - a); first = 0; } while (0)
int main() {
  int first = 1, n = 0;
  printf("{");
  T("hipGetDeviceCount", hipGetDeviceCount(&n));
  T("hipSetDevice+hipFree0", { hipSetDevice(0); hipFree(0); });
  hipStream_t s, s2;
  T("stream_create_x2", { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); });
  void *d[8];
  T("hipMalloc_537MB", hipMalloc(&d[0], 537u << 20));
  T("hipMalloc_512MB", hipMalloc(&d[1], 512u << 20));
  T("hipMalloc_671MB", hipMalloc(&d[2], 671u << 20));
  T("hipMalloc_134MB", hipMalloc(&d[3], 134u << 20));
  T("hipMalloc_64MB", hipMalloc(&d[4], 64u << 20));
  T("hipMalloc_64MB_b", hipMalloc(&d[5], 64u << 20));
  T("hipMemset_537MB_sync", { hipMemsetAsync(d[0], 0, 537u << 20, s); hipStreamSynchronize(s); });
  void *pg = malloc(64u << 20); memset(pg, 1, 64u << 20);
  T("hipMemcpy_64MB_pageable", hipMemcpy(d[4], pg, 64u << 20, hipMemcpyHostToDevice));
  T("hipMemcpy_64MB_pageable_again", hipMemcpy(d[4], pg, 64u << 20, hipMemcpyHostToDevice));
  std::vector<void *> pins(24);
  double a = now();
  for (auto &p : pins) hipHostMalloc(&p, 9u << 20, hipHostMallocDefault);
  printf(", \"hipHostMalloc_9MB_x24_total\": %.5f", now() - a);
  void *big;
  T("hipHostMalloc_64MB", hipHostMalloc(&big, 64u << 20, hipHostMallocDefault));
  void *big2;
  T("hipHostMalloc_512MB", hipHostMalloc(&big2, 512u << 20, hipHostMallocDefault));
  void *reg = aligned_alloc(4096, 512u << 20); memset(reg, 1, 512u << 20);
  T("hipHostRegister_512MB_touched", hipHostRegister(reg, 512u << 20, hipHostRegisterDefault));
  // H2D by piece size, pieces alternate between two device buffers, one stream
  for (size_t mb : {4, 9, 16, 64}) {
    size_t bytes = mb << 20, total = 0;
    for (auto &p : pins) memset(p, 2, 9u << 20);
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < 64; i++) {
      void *src = mb <= 9 ? pins[i % pins.size()] : (mb == 16 ? (char *)big2 + (size_t)(i % 16) * bytes : big);
      hipMemcpyAsync(d[4 + (i & 1)], src, bytes, hipMemcpyHostToDevice, s);
      total += bytes;
    }
    hipStreamSynchronize(s);
    double dt = now() - t0;
    printf(", \"h2d_%zuMB_pieces_GBps\": %.2f", mb, total / dt / 1e9);
  }
  // two streams concurrently
  {
    double t0 = now(); size_t total = 0;
    for (int i = 0; i < 64; i++) { hipMemcpyAsync(d[4 + (i & 1)], (char *)big2 + (size_t)(i % 16) * (16u << 20), 16u << 20, hipMemcpyHostToDevice, (i & 1) ? s : s2); total += 16u << 20; }
    hipStreamSynchronize(s); hipStreamSynchronize(s2);
    printf(", \"h2d_16MB_two_streams_GBps\": %.2f", total / (now() - t0) / 1e9);
  }
  T("hipHostFree_512MB", hipHostFree(big2));
  T("hipFree_537MB", hipFree(d[0]));
  printf("}\n");
  return 0;
}
