// tools/ubench_valu3.hip -- issue rate of the instruction forms the scan kernel's probe block uses (inline asm):
// cycles per wave-instruction per SIMD with 1..4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4000
#define REP16(x) x x x x x x x x x x x x x x x x
#define EIGHT(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define OPS(body) asm volatile(REP16(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc))

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned long long *cyc, unsigned sc) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 + 11, a6 = a0 ^ 13, a7 = a0 ^ 17, b = a0 * 31 + 1;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
    if (OP == 0) OPS("v_and_b32_sdwa %0, %0, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %1, %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n");
    if (OP == 1) OPS("v_lshlrev_b32_sdwa %0, %0, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %1, %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %2, %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %3, %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_lshlrev_b32_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n");
    if (OP == 2) OPS("v_bitop3_b32 %0, %0, %8, %1 bitop3:0x32\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0x32\n v_bitop3_b32 %2, %2, %8, %3 bitop3:0x32\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0x32\n v_bitop3_b32 %4, %4, %8, %5 bitop3:0x32\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0x32\n v_bitop3_b32 %6, %6, %8, %7 bitop3:0x32\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0x32\n");
    if (OP == 3) OPS("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n");
    if (OP == 4) OPS("v_perm_b32 %0, %0, %8, %1\n v_perm_b32 %1, %1, %8, %2\n v_perm_b32 %2, %2, %8, %3\n v_perm_b32 %3, %3, %8, %4\n v_perm_b32 %4, %4, %8, %5\n v_perm_b32 %5, %5, %8, %6\n v_perm_b32 %6, %6, %8, %7\n v_perm_b32 %7, %7, %8, %0\n");
    if (OP == 5) OPS("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7\n");
    if (OP == 6) OPS("v_and_or_b32 %0, %0, 3, %8\n v_and_or_b32 %1, %1, 3, %8\n v_and_or_b32 %2, %2, 3, %8\n v_and_or_b32 %3, %3, 3, %8\n v_and_or_b32 %4, %4, 3, %8\n v_and_or_b32 %5, %5, 3, %8\n v_and_or_b32 %6, %6, 3, %8\n v_and_or_b32 %7, %7, 3, %8\n");
    if (OP == 7) OPS("v_lshlrev_b32_e64 %0, %0, 1\n v_lshlrev_b32_e64 %1, %1, 1\n v_lshlrev_b32_e64 %2, %2, 1\n v_lshlrev_b32_e64 %3, %3, 1\n v_lshlrev_b32_e64 %4, %4, 1\n v_lshlrev_b32_e64 %5, %5, 1\n v_lshlrev_b32_e64 %6, %6, 1\n v_lshlrev_b32_e64 %7, %7, 1\n");
    if (OP == 8) OPS("v_and_b32 %0, 0xfffc, %0\n v_and_b32 %1, 0xfffc, %1\n v_and_b32 %2, 0xfffc, %2\n v_and_b32 %3, 0xfffc, %3\n v_and_b32 %4, 0xfffc, %4\n v_and_b32 %5, 0xfffc, %5\n v_and_b32 %6, 0xfffc, %6\n v_and_b32 %7, 0xfffc, %7\n");
    if (OP == 9) OPS("v_or3_b32 %0, %0, %8, %1\n v_or3_b32 %1, %1, %8, %2\n v_or3_b32 %2, %2, %8, %3\n v_or3_b32 %3, %3, %8, %4\n v_or3_b32 %4, %4, %8, %5\n v_or3_b32 %5, %5, %8, %6\n v_or3_b32 %6, %6, %8, %7\n v_or3_b32 %7, %7, %8, %0\n");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> void run(const char *name) {
  unsigned *out; unsigned long long *cyc;
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
  for (int W = 1; W <= 4; W += 3) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * W), 0, 0, out, cyc, 1u);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 4 * W);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    double instr = (double)ITER * 128;
    printf("%-22s W=%d  per wave %.2f cyc/instr   per SIMD %.2f cyc/instr\n", name, W, avg / instr, avg / instr / W);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_and_b32_sdwa W1"); run<1>("v_lshlrev_b32_sdwa B1"); run<2>("v_bitop3_b32"); run<3>("v_or_b32"); run<4>("v_perm_b32");
  run<5>("v_lshlrev_b32 c"); run<6>("v_and_or_b32"); run<7>("v_lshlrev_e64 v,1"); run<8>("v_and_b32 lit"); run<9>("v_or3_b32");
  return 0;
}
