// tools/ubench_valu2.hip -- inline-asm issue-rate test (nothing can be folded): cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4000
#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned long long *cyc) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 + 11, a6 = a0 ^ 13, a7 = a0 ^ 17, b = a0 * 31 + 1;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
    if (OP == 0) asm volatile(REP16("v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n v_lshl_or_b32 %4, %4, 2, %8\n v_lshl_or_b32 %5, %5, 2, %8\n v_lshl_or_b32 %6, %6, 2, %8\n v_lshl_or_b32 %7, %7, 2, %8\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == 1) asm volatile(REP16("v_alignbit_b32 %0, %0, %8, 30\n v_alignbit_b32 %1, %1, %8, 30\n v_alignbit_b32 %2, %2, %8, 30\n v_alignbit_b32 %3, %3, %8, 30\n v_alignbit_b32 %4, %4, %8, 30\n v_alignbit_b32 %5, %5, %8, 30\n v_alignbit_b32 %6, %6, %8, 30\n v_alignbit_b32 %7, %7, %8, 30\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == 2) asm volatile(REP16("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == 3) asm volatile(REP16("v_lshrrev_b32 %0, 8, %0\n v_lshrrev_b32 %1, 8, %1\n v_lshrrev_b32 %2, 8, %2\n v_lshrrev_b32 %3, 8, %3\n v_lshrrev_b32 %4, 8, %4\n v_lshrrev_b32 %5, 8, %5\n v_lshrrev_b32 %6, 8, %6\n v_lshrrev_b32 %7, 8, %7\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == 4) asm volatile(REP16("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == 5) asm volatile(REP16("v_bfe_u32 %0, %0, 8, 2\n v_bfe_u32 %1, %1, 8, 2\n v_bfe_u32 %2, %2, 8, 2\n v_bfe_u32 %3, %3, 8, 2\n v_bfe_u32 %4, %4, 8, 2\n v_bfe_u32 %5, %5, 8, 2\n v_bfe_u32 %6, %6, 8, 2\n v_bfe_u32 %7, %7, 8, 2\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (OP == 6) asm volatile(REP16("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7\n")
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> void run(const char *name) {
  unsigned *out; unsigned long long *cyc;
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
  for (int W = 1; W <= 4; W++) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * W), 0, 0, out, cyc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 4 * W);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    double instr = (double)ITER * 128;
    printf("%-16s W=%d  per wave %.2f cyc/instr   per SIMD %.2f cyc/instr\n", name, W, avg / instr, avg / instr / W);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_lshl_or_b32"); run<1>("v_alignbit_b32"); run<2>("v_and_b32"); run<3>("v_lshrrev_b32"); run<4>("v_min_u32"); run<5>("v_bfe_u32"); run<6>("v_lshlrev_b32 v");
  return 0;
}
