// tools/ubench_fetch.hip -- calibrates FETCH_SIZE on known byte counts:
//   pattern A: contiguous 16 B/lane stream over the whole buffer (the guide's reference pattern)
//   pattern B: the scan kernel's staging pattern: tiles of 64 rows x 160 B, two passes of 80 B per row,
//              5 x 16 B pieces per row per pass, one wave per tile, second pass one "step" later
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(1024) patA(const uint4 *p, size_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(1024) patB(const uint8_t *rows, size_t nreads, uint32_t *sink) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = nreads / 64, nw = (size_t)gridDim.x * 16, w0 = (size_t)blockIdx.x * 16 + wave;
  uint32_t acc = 0;
  for (size_t t = w0; t < ntiles; t += nw)
    for (int cb = 0; cb < 2; cb++) {
      uint4 v[5];
      for (int i = 0; i < 5; i++) { uint32_t q = lane + 64 * i, r = q / 5, c = q % 5; v[i] = *(const uint4 *)(rows + (t * 64 + r) * 160 + cb * 80 + c * 16); }
      for (int i = 0; i < 5; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
      for (int k = 0; k < 400; k++) acc = acc * 1664525u + 1013904223u; /* ~ one step of work between the passes */
    }
  if (acc == 0x12345678u) sink[0] = acc;
}
// pattern C: same tiles, both 80-byte halves of every row loaded back to back (10 pieces in flight), then the work
__global__ void __launch_bounds__(1024) patC(const uint8_t *rows, size_t nreads, uint32_t *sink) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = nreads / 64, nw = (size_t)gridDim.x * 16, w0 = (size_t)blockIdx.x * 16 + wave;
  uint32_t acc = 0;
  for (size_t t = w0; t < ntiles; t += nw) {
    uint4 v[10];
    for (int i = 0; i < 10; i++) { uint32_t q = lane + 64 * i, r = q / 10, c = q % 10; v[i] = *(const uint4 *)(rows + (t * 64 + r) * 160 + c * 16); }
    for (int i = 0; i < 10; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    for (int k = 0; k < 800; k++) acc = acc * 1664525u + 1013904223u;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// pattern D: rows at a 192-byte pitch... not needed
int main() {
  const size_t nreads = 50000000, bytes = nreads * 160;
  uint8_t *buf; uint32_t *sink;
  (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sink, 64); (void)hipMemset(buf, 1, bytes);
  hipLaunchKernelGGL(patA, dim3(2048), dim3(1024), 0, 0, (const uint4 *)buf, bytes / 16, sink);
  hipLaunchKernelGGL(patB, dim3(256), dim3(1024), 0, 0, buf, nreads, sink);
  hipLaunchKernelGGL(patC, dim3(256), dim3(1024), 0, 0, buf, nreads, sink);
  (void)hipDeviceSynchronize();
  printf("buffer bytes %zu\n", bytes);
  return 0;
}
