// tools/probe_hostread.hip -- does a kernel read pinned host memory as fast as the copy engine moves it?
// One registered host buffer (as the command line's batch buffers: malloc + hipHostRegister); (a) hipMemcpyAsync to the device,
// (b) a kernel that loads the same bytes through the buffer's device pointer (four 16-byte loads a lane and step, a wave reads
// 4 KiB of consecutive memory: the packed scan kernel's pattern).  Prints GB/s of both and the host time of the first call of each.
//   hipcc --offload-arch=gfx950 -O3 -o probe_hostread tools/probe_hostread.hip && ./probe_hostread [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) read_rows(const uint4 *src, unsigned long long nrows, unsigned long long *out) {
  unsigned long long acc = 0;
  for (unsigned long long r = (unsigned long long)blockIdx.x * 256u + threadIdx.x; r < nrows; r += (unsigned long long)gridDim.x * 256u) {
    const uint4 a = src[4 * r], b = src[4 * r + 1], c = src[4 * r + 2], d = src[4 * r + 3];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x1234567ull) out[0] = acc;
}
int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 256;
  const size_t n = mib << 20;
  const int blocks = argc > 2 ? atoi(argv[2]) : 2048;
  CK(hipSetDevice(0));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t *h = (uint8_t *)aligned_alloc(4096, n);
  memset(h, 0x41, n);
  double t0 = now();
  CK(hipHostRegister(h, n, hipHostRegisterDefault));
  printf("{\"register_ms\": %.3f", (now() - t0) * 1e3);
  void *hd = nullptr; CK(hipHostGetDevicePointer(&hd, h, 0));
  printf(", \"device_pointer_is_host_pointer\": %s", hd == (void *)h ? "true" : "false");
  void *d; CK(hipMalloc(&d, n));
  unsigned long long *out; CK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *first = argc > 3 ? argv[3] : "kernel";
  for (int round = 0; round < 2; round++) {
    const bool kernel = (round == 0) == (strcmp(first, "kernel") == 0);
    for (int rep = 0; rep < 4; rep++) {
      t0 = now();
      CK(hipEventRecord(e0, s));
      if (kernel) hipLaunchKernelGGL(read_rows, dim3(blocks), dim3(256), 0, s, (const uint4 *)hd, (unsigned long long)(n / 64), out);
      else CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
      const double tq = now() - t0;
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf(", \"%s_%d\": {\"queued_ms\": %.3f, \"ms\": %.3f, \"GB_s\": %.1f}", kernel ? "kernel" : "copy", rep, tq * 1e3, ms, n / (ms * 1e-3) / 1e9);
    }
  }
  printf("}\n");
  return 0;
}
