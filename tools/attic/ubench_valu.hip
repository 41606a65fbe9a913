// tools/ubench_valu.hip -- issue-rate microbenchmark for the integer VALU ops the scan kernel is made of.
// For each op: one workgroup per CU with W waves per SIMD, each wave runs ITER x 64 instances of the op on
// 8 independent chains; reports cycles per wave-instruction per SIMD (s_memtime) = W-wave aggregate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 2000

template <int OP>
__device__ __forceinline__ void op8(uint32_t (&a)[8], uint32_t (&b)[8], uint64_t (&c)[4], uint32_t s) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (OP == 0) a[i] = (a[i] << 2) | b[i];                                   // v_lshl_or_b32
    if (OP == 1) a[i] = __builtin_amdgcn_alignbit(a[i], b[i], 30);             // v_alignbit_b32
    if (OP == 2) a[i] = (a[i] & 0xfffu) ^ b[i];                                // v_and + v_xor (2 ops)
    if (OP == 3) a[i] = __builtin_amdgcn_ubfe(a[i] + b[i], 8, 2) + a[i];       // add, bfe, add (3 ops)
    if (OP == 4) a[i] = __builtin_amdgcn_perm(a[i], b[i], 0x03020100u ^ a[i]); // xor + v_perm_b32 (2 ops)
    if (OP == 5) a[i] = min(a[i] + 1u, b[i]);                                  // add + v_min_u32 (2 ops)
  }
  if (OP == 6) {
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = (c[i] >> 2) | ((uint64_t)b[i] << 42);   // 64-bit shift form (compiler's choice)
  }
  if (OP == 7) {
#pragma unroll
    for (int i = 0; i < 4; i++) { bool lt = c[i] < c[(i + 1) & 3]; a[i] = lt ? a[i] : b[i]; a[i + 4] = lt ? b[i] : a[i + 4]; }  // cmp_lt_u64 + 2 cndmask
  }
  if (OP == 8) {
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = c[i] >> s;                              // v_lshrrev_b64 by SGPR
  }
}

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, unsigned long long *cyc, uint32_t s) {
  uint32_t a[8], b[8];
  uint64_t c[4];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 17 + i; b[i] = threadIdx.x * 31 + i * 7 + s; }
  for (int i = 0; i < 4; i++) c[i] = ((uint64_t)a[i] << 20) ^ b[i];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
    op8<OP>(a, b, c, s); op8<OP>(a, b, c, s); op8<OP>(a, b, c, s); op8<OP>(a, b, c, s);
    op8<OP>(a, b, c, s); op8<OP>(a, b, c, s); op8<OP>(a, b, c, s); op8<OP>(a, b, c, s);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i] ^ b[i];
  for (int i = 0; i < 4; i++) r ^= (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, int ops_per_call) {
  uint32_t *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  for (int W = 1; W <= 4; W++) {
    int threads = 256 * W;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 3u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    double instr = (double)ITER * 8 * ops_per_call;   // wave-instructions per wave
    printf("%-34s W=%d  cycles/instr/wave %.2f   per SIMD aggregate %.2f cycles/instr\n", name, W, avg / instr, avg / instr / W);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_lshl_or_b32", 8);
  run<1>("v_alignbit_b32", 8);
  run<2>("v_and+v_xor", 16);
  run<3>("add+bfe+add", 24);
  run<4>("xor+v_perm", 16);
  run<5>("add+v_min_u32", 16);
  run<6>("u64 (c>>2)|(b<<42) [4x]", 4);
  run<7>("cmp_lt_u64+2cndmask [4x]", 4);
  run<8>("v_lshrrev_b64 sgpr [4x]", 4);
  return 0;
}
