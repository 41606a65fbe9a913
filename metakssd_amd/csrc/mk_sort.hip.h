/*
 * mk_sort.hip.h -- stable LSD radix sort of (u32 key, u32 value) pairs on the device (gfx950, wave64), hand-written.
 *
 * Used by stage II (mk_mco.hip): combco2mco() (co2mco.c:37-59) appends genome after genome to one growable array per k-mer
 * id, i.e. it produces (id, genome) sorted by id with the genomes of an id in input order -- a STABLE sort by the 32-bit id.
 *
 * Four passes of 8 bits.  Every pass is three launches:
 *   mk_rs_hist_kernel     B persistent workgroups, each over a contiguous range of 4096-element tiles: 256-bin histogram in LDS
 *                         -> hist[digit][workgroup]
 *   mk_rs_scan_kernel     one workgroup per digit: exclusive prefix over the workgroups (B <= 1024: one entry per thread), digit totals
 *   mk_rs_scatter_kernel  the same ranges again, tile by tile and inside a tile quarter by quarter (1024 consecutive elements, one
 *                         per thread): a wave finds, with eight ballots, which of its lanes hold the same digit (rank inside the
 *                         wave = lanes below with that digit); the waves' counts per digit go through LDS for the rank of the
 *                         wave; position = digit base + workgroup prefix + what earlier tiles / quarters / waves put there + rank
 *                         inside the wave.  Everything is taken in index order, so equal keys keep their order.
 * A pass in which every key has the same digit is skipped (its histogram says so).  Keys and values ping-pong between two
 * buffer pairs.  Bound: HBM, 16 B per pair and pass (read key + value, write key + value) plus the histogram pass's 4 B.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MK_RS_THREADS 1024u
#define MK_RS_ITEMS 4u
#define MK_RS_TILE (MK_RS_THREADS * MK_RS_ITEMS)
#define MK_RS_MAXB 1024u /* workgroups: the scan kernel takes one entry per thread */

struct mk_rs_plan {
  uint64_t n;
  uint64_t ntiles;
  uint32_t B;        /* workgroups */
  uint64_t per;      /* tiles per workgroup */
};

static inline mk_rs_plan mk_rs_make_plan(uint64_t n, int num_cu) {
  mk_rs_plan p;
  p.n = n;
  p.ntiles = (n + MK_RS_TILE - 1) / MK_RS_TILE;
  uint64_t B = p.ntiles;
  const uint64_t cap = (uint64_t)num_cu * 4u < MK_RS_MAXB ? (uint64_t)num_cu * 4u : MK_RS_MAXB;
  if (B > cap) B = cap;
  if (B == 0) B = 1;
  p.B = (uint32_t)B;
  p.per = (p.ntiles + B - 1) / B;
  return p;
}

__global__ void __launch_bounds__(MK_RS_THREADS) mk_rs_hist_kernel(const uint32_t *key, mk_rs_plan p, uint32_t shift, uint32_t *hist) {
  __shared__ uint32_t h[256];
  if (threadIdx.x < 256u) h[threadIdx.x] = 0u;
  __syncthreads();
  const uint64_t t0 = (uint64_t)blockIdx.x * p.per, t1 = t0 + p.per < p.ntiles ? t0 + p.per : p.ntiles;
  for (uint64_t t = t0; t < t1; t++) {
    const uint64_t base = t * MK_RS_TILE;
    uint32_t k[MK_RS_ITEMS];
    bool ok[MK_RS_ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_ITEMS; i++) {
      const uint64_t idx = base + (uint64_t)i * MK_RS_THREADS + threadIdx.x;
      ok[i] = idx < p.n;
      k[i] = ok[i] ? key[idx] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_ITEMS; i++)
      if (ok[i]) atomicAdd(&h[(k[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256u) hist[(size_t)threadIdx.x * p.B + blockIdx.x] = h[threadIdx.x];
}

/* block d: exclusive prefix of hist[d][0..B) in place, total[d]; flag[0] |= 1 when more than one digit is in use */
__global__ void __launch_bounds__(MK_RS_THREADS) mk_rs_scan_kernel(uint32_t *hist, uint32_t B, unsigned long long *total, uint64_t n, uint32_t *flag) {
  __shared__ unsigned long long wsum[MK_RS_THREADS / 64];
  const uint32_t d = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const unsigned long long v = t < B ? hist[(size_t)d * B + t] : 0ull;
  unsigned long long incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = __shfl_up(incl, o);
    if ((int)lane >= o) incl += u;
  }
  if (lane == 63u) wsum[wave] = incl;
  __syncthreads();
  unsigned long long woff = 0, all = 0;
  for (uint32_t w = 0; w < MK_RS_THREADS / 64; w++) { if (w < wave) woff += wsum[w]; all += wsum[w]; }
  if (t < B) hist[(size_t)d * B + t] = (uint32_t)(woff + incl - v); /* per-digit prefixes stay below 2^32 for n < 2^32 */
  if (t == 0) {
    total[d] = all;
    if (all != 0ull && all != (unsigned long long)n) atomicOr(flag, 1u);
  }
}

__global__ void __launch_bounds__(MK_RS_THREADS) mk_rs_scatter_kernel(const uint32_t *key, const uint32_t *val, uint32_t *okey, uint32_t *oval,
                                                                     mk_rs_plan p, uint32_t shift, const uint32_t *hist,
                                                                     const unsigned long long *total, const uint32_t *flag) {
  if (flag[0] == 0u) return; /* every key has the same digit here: this pass is the identity, the caller does not swap */
  constexpr uint32_t WAVES = MK_RS_THREADS / 64, WP = WAVES + 1u; /* +1: bank spread */
  __shared__ unsigned long long base[256];              /* where the next element with this digit goes */
  __shared__ uint32_t wcount[MK_RS_ITEMS][256][WP];     /* [quarter][digit][wave]: counts, then exclusive prefixes over (quarter, wave) */
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  if (t < 256u) base[t] = total[t]; /* (borrowed: scanned below) */
  __syncthreads();
  if (t == 0) { /* exclusive scan over the 256 digit totals: once per workgroup, cheap */
    unsigned long long run = 0;
    for (uint32_t d = 0; d < 256u; d++) { const unsigned long long c = base[d]; base[d] = run; run += c; }
  }
  __syncthreads();
  if (t < 256u) base[t] += hist[(size_t)t * p.B + blockIdx.x];
  const uint64_t t0 = (uint64_t)blockIdx.x * p.per, t1 = t0 + p.per < p.ntiles ? t0 + p.per : p.ntiles;
  const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
  for (uint64_t tile = t0; tile < t1; tile++) {
    const uint64_t tb = tile * MK_RS_TILE;
    uint32_t k[MK_RS_ITEMS], v[MK_RS_ITEMS], rank[MK_RS_ITEMS];
    bool ok[MK_RS_ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_ITEMS; i++) { /* all loads of the tile in flight */
      const uint64_t idx = tb + (uint64_t)i * MK_RS_THREADS + t;
      ok[i] = idx < p.n;
      k[i] = ok[i] ? key[idx] : 0xFFFFFFFFu;
      v[i] = ok[i] ? val[idx] : 0u;
    }
    for (uint32_t j = t; j < MK_RS_ITEMS * 256u * WP; j += MK_RS_THREADS) ((uint32_t *)wcount)[j] = 0u;
    __syncthreads(); /* (also: the previous tile's readers of wcount and base are done) */
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_ITEMS; i++) { /* quarter i: elements tb + i * 1024 + [0, 1024) in thread order */
      const uint32_t d = (k[i] >> shift) & 255u;
      uint64_t same = __ballot(ok[i]); /* lanes of this wave with the same digit (and an element at all) */
#pragma unroll
      for (uint32_t b = 0; b < 8u; b++) {
        const uint64_t m = __ballot((d >> b) & 1u);
        same &= ((d >> b) & 1u) ? m : ~m;
      }
      rank[i] = (uint32_t)__popcll(same & below);
      if (ok[i] && rank[i] == 0u) wcount[i][d][wave] = (uint32_t)__popcll(same);
    }
    __syncthreads();
    { /* per digit: exclusive prefix over the tile's (quarter, wave) cells in that order; the digit's total moves its base */
      const uint32_t d = t & 255u, q = t >> 8; /* four threads per digit, one quarter each */
      uint32_t run = 0;
#pragma unroll
      for (uint32_t w = 0; w < WAVES; w++) { const uint32_t c = wcount[q][d][w]; wcount[q][d][w] = run; run += c; }
      wcount[q][d][WAVES] = run; /* the quarter's total for this digit */
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_ITEMS; i++) {
      if (!ok[i]) continue;
      const uint32_t d = (k[i] >> shift) & 255u;
      uint32_t before = wcount[i][d][wave] + rank[i];
#pragma unroll
      for (uint32_t q = 0; q < MK_RS_ITEMS; q++) if (q < i) before += wcount[q][d][WAVES];
      const unsigned long long pos = base[d] + before;
      okey[pos] = k[i];
      oval[pos] = v[i];
    }
    __syncthreads();
    if (t < 256u) {
      uint32_t s = 0;
#pragma unroll
      for (uint32_t q = 0; q < MK_RS_ITEMS; q++) s += wcount[q][t][WAVES];
      base[t] += s;
    }
    __syncthreads(); /* the totals have been read: the next tile may zero the cells */
  }
}

/* Sorts n pairs by key, stable.  key[0]/val[0] hold the input; key[1]/val[1] are scratch of the same size.  Returns (in *in_first)
 * which pair of buffers holds the result.  hist: 256 * MK_RS_MAXB u32, total: 256 u64, flag: 4 u32 (one per pass), all device. */
static inline hipError_t mk_radix_sort_pairs_u32(uint32_t *key[2], uint32_t *val[2], uint64_t n, int num_cu, uint32_t *hist,
                                                 unsigned long long *total, uint32_t *flag, uint32_t *h_flag /*pinned, 4*/, hipStream_t st,
                                                 int *result_in) {
  int cur = 0;
  if (n > 1) {
    const mk_rs_plan p = mk_rs_make_plan(n, num_cu);
    hipError_t r = hipMemsetAsync(flag, 0, 4 * sizeof(uint32_t), st);
    if (r != hipSuccess) return r;
    for (uint32_t pass = 0; pass < 4u; pass++) {
      const uint32_t shift = 8u * pass;
      hipLaunchKernelGGL(mk_rs_hist_kernel, dim3(p.B), dim3(MK_RS_THREADS), 0, st, (const uint32_t *)key[cur], p, shift, hist);
      hipLaunchKernelGGL(mk_rs_scan_kernel, dim3(256), dim3(MK_RS_THREADS), 0, st, hist, p.B, total, n, flag + pass);
      hipLaunchKernelGGL(mk_rs_scatter_kernel, dim3(p.B), dim3(MK_RS_THREADS), 0, st, (const uint32_t *)key[cur], (const uint32_t *)val[cur],
                         key[cur ^ 1], val[cur ^ 1], p, shift, (const uint32_t *)hist, (const unsigned long long *)total,
                         (const uint32_t *)(flag + pass));
      r = hipGetLastError();
      if (r != hipSuccess) return r;
      /* whether the pass moved anything decides where the next one reads: the flag comes to the host (4 bytes, once per pass) */
      r = hipMemcpyAsync(h_flag + pass, flag + pass, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
      if (r != hipSuccess) return r;
      r = hipStreamSynchronize(st);
      if (r != hipSuccess) return r;
      if (h_flag[pass]) cur ^= 1;
    }
  }
  *result_in = cur;
  return hipSuccess;
}
