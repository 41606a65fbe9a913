// tools/probe_h2d.hip -- host-to-device rate of pinned row buffers while host threads stream through memory (what the
// FASTQ front end does while the engine copies): SDMA copies (hipMemcpyAsync) vs a copy kernel reading the pinned buffer
// directly, hipHostMalloc vs hipHostRegister memory, hammer threads on either NUMA node.  One JSON line.
//   hipcc -O2 --offload-arch=gfx950 tools/probe_h2d.hip -o tools/probe_h2d -lpthread
#define _GNU_SOURCE
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static std::atomic<int> g_stop{0};
struct hammer { int cpu_lo, cpu_hi; pthread_t th; };
static void *hammer_run(void *arg) {
  hammer *h = (hammer *)arg;
  if (h->cpu_lo >= 0) { cpu_set_t s; CPU_ZERO(&s); for (int c = h->cpu_lo; c < h->cpu_hi; c++) CPU_SET(c, &s); sched_setaffinity(0, sizeof s, &s); }
  const size_t n = 96u << 20;
  char *a = (char *)malloc(n), *b = (char *)malloc(n);
  memset(a, 1, n); memset(b, 2, n);
  while (!g_stop.load(std::memory_order_relaxed)) { memcpy(b, a, n); }
  free(a); free(b);
  return nullptr;
}
__global__ void copy_kernel(const uint4 *src, uint4 *dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  hipSetDevice(0);
  printf("{");
  { FILE *f = fopen("/sys/class/drm/card0/device/numa_node", "r"); int nn = -9; if (f) { if (fscanf(f, "%d", &nn) != 1) nn = -9; fclose(f); }
    char bus[64] = ""; hipDeviceGetPCIBusId(bus, 64, 0);
    char path[128]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus); for (char *p = path; *p; p++) if (*p >= 'A' && *p <= 'F') *p += 32;
    int n2 = -9; f = fopen(path, "r"); if (f) { if (fscanf(f, "%d", &n2) != 1) n2 = -9; fclose(f); }
    printf("\"card0_numa\": %d, \"gpu_pci\": \"%s\", \"gpu_numa\": %d", nn, bus, n2); }
  const size_t total = 2048ull << 20, piece = 16u << 20;
  char *pin = nullptr; hipHostMalloc((void **)&pin, total, hipHostMallocDefault); memset(pin, 3, total);
  char *reg = (char *)aligned_alloc(4096, total); memset(reg, 3, total);
  double t0 = now(); hipHostRegister(reg, total, hipHostRegisterDefault); printf(", \"register_2GiB_s\": %.4f", now() - t0);
  char *d[2]; hipMalloc((void **)&d[0], piece); hipMalloc((void **)&d[1], piece);
  char *dbig; hipMalloc((void **)&dbig, total);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto sdma = [&](char *src) { double a = now(); for (size_t off = 0, i = 0; off < total; off += piece, i++) hipMemcpyAsync(d[i & 1], src + off, piece, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); return total / (now() - a) / 1e9; };
  auto kern = [&](char *src) { double a = now(); hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dbig, total / 16); hipStreamSynchronize(s); return total / (now() - a) / 1e9; };
  sdma(pin); kern(pin);
  struct cfg { const char *name; int k, lo, hi; };
  cfg cfgs[] = {{"idle", 0, -1, -1}, {"hammer8", 8, -1, -1}, {"hammer16", 16, -1, -1}, {"hammer32", 32, -1, -1}, {"hammer64", 64, -1, -1},
                {"hammer16_cpus0_63", 16, 0, 64}, {"hammer16_cpus64_127", 16, 64, 128}, {"hammer32_cpus0_63", 32, 0, 64}, {"hammer32_cpus64_127", 32, 64, 128}};
  for (auto &c : cfgs) {
    g_stop = 0;
    std::vector<hammer> hs(c.k);
    for (auto &h : hs) { h.cpu_lo = c.lo; h.cpu_hi = c.hi; pthread_create(&h.th, nullptr, hammer_run, &h); }
    if (c.k) { timespec ts{0, 300000000}; nanosleep(&ts, nullptr); }
    double a1 = sdma(pin), a2 = sdma(reg), b1 = kern(pin), b2 = kern(reg);
    printf(", \"%s\": {\"sdma_pinned\": %.1f, \"sdma_registered\": %.1f, \"kernel_pinned\": %.1f, \"kernel_registered\": %.1f}", c.name, a1, a2, b1, b2);
    fflush(stdout);
    g_stop = 1;
    for (auto &h : hs) pthread_join(h.th, nullptr);
  }
  printf("}\n");
  return 0;
}
