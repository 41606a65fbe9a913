/*
 * mk_sort.hip.h -- stable LSD radix sort of (u32 key, u32 value) pairs on the device (gfx950, wave64), hand-written.
 *
 * Used by stage II (mk_mco.hip): combco2mco() (co2mco.c:37-59) appends genome after genome to one growable array per k-mer
 * id, i.e. it produces (id, genome) sorted by id with the genomes of an id in input order -- a STABLE sort by the 32-bit id.
 *
 * Four passes of 8 bits.  Every pass is three launches:
 *   mk_rs_hist_kernel     B persistent workgroups, each over a contiguous range of 8192-element tiles: 256-bin histogram in LDS
 *                         -> hist[digit][workgroup]
 *   mk_rs_scan_kernel     one workgroup per digit: exclusive prefix over the workgroups (B <= 1024: one entry per thread), digit totals
 *   mk_rs_scatter_kernel  the same ranges again, tile by tile, 512 threads: a wave takes 1024 consecutive elements, 64 at a time; with
 *                         eight ballots it finds which of its lanes hold the same digit (rank = lanes below with that digit + what
 *                         the wave's earlier items hold of it, a running count per wave and digit in LDS); one wave then turns the
 *                         waves' counts into prefixes and digit starts; place in the tile's sorted order = digit start + earlier
 *                         waves + rank.  The tile is put into that order in LDS and stored from there, so that a wave's stores run
 *                         along the digits' runs (position = digit base + workgroup prefix + earlier tiles + place).  Everything
 *                         is taken in index order, so equal keys keep their order.
 * A pass in which every key has the same digit is skipped (its histogram says so).  Keys and values ping-pong between two
 * buffer pairs.  Bound: HBM, 16 B per pair and pass (read key + value, write key + value) plus the histogram pass's 4 B.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MK_RS_THREADS 1024u
/* 8192-element tiles: 256 digits share a tile, so a digit's run in it is 32 elements = one whole 128-byte line on average.  With
 * 4096 (256 threads, four workgroups a CU) the runs are half lines and the scatter takes 2.97 ms a pass for 400 M pairs instead of
 * 2.18-2.38; 16384 (1024 threads, one workgroup a CU) gains nothing more (2.18) */
#define MK_RS_ITEMS 8u
#define MK_RS_SC_THREADS 512u /* the scatter kernel: eight waves, sixteen items per thread, two workgroups per CU */
#define MK_RS_SC_WGS_PER_CU 2u
#define MK_RS_TILE (MK_RS_THREADS * MK_RS_ITEMS)
#define MK_RS_SC_ITEMS 16u
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
  const uint64_t cap = (uint64_t)num_cu * MK_RS_SC_WGS_PER_CU < MK_RS_MAXB ? (uint64_t)num_cu * MK_RS_SC_WGS_PER_CU : MK_RS_MAXB;
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

__global__ void __launch_bounds__(MK_RS_SC_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) mk_rs_scatter_kernel(const uint32_t *key, const uint32_t *val, uint32_t *okey, uint32_t *oval,
                                                                        mk_rs_plan p, uint32_t shift, const uint32_t *hist,
                                                                        const unsigned long long *total, const uint32_t *flag) {
  if (flag[0] == 0u) return; /* every key has the same digit here: this pass is the identity, the caller does not swap */
  constexpr uint32_t WAVES = MK_RS_SC_THREADS / 64, PER_WAVE = 64u * MK_RS_SC_ITEMS;
  static_assert(WAVES * PER_WAVE == MK_RS_TILE, "the scatter kernel's tile is the histogram kernel's");
  __shared__ unsigned long long gofs[256];              /* this tile: output position of an element = gofs[digit] + its place in the tile's sorted order */
  __shared__ uint32_t dstart[256];                      /* this tile: where the digit starts in the tile's sorted order */
  __shared__ uint32_t wc[WAVES][256];                   /* per wave and digit: running count while the wave ranks its items, then the
                                                           exclusive prefix over the waves */
  __shared__ uint32_t lk[MK_RS_TILE], lv[MK_RS_TILE];   /* the tile in sorted order: its stores to HBM run along the digits' runs */
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  unsigned long long base[4] = {0, 0, 0, 0}; /* wave 0, digits 4 * lane + j: where the next tile's first element with the digit goes
                                                (in registers: with it in LDS a fourth workgroup would not fit a CU) */
  if (wave == 0u) { /* exclusive scan over the 256 digit totals (four per lane) + what the workgroups before this one hold */
    unsigned long long c[4], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) { c[j] = total[lane * 4u + j]; sum += c[j]; }
    unsigned long long incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long u = __shfl_up(incl, o);
      if ((int)lane >= o) incl += u;
    }
    unsigned long long run = incl - sum;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
      const uint32_t d = lane * 4u + j;
      base[j] = run + hist[(size_t)d * p.B + blockIdx.x];
      run += c[j];
    }
  }
  const uint64_t t0 = (uint64_t)blockIdx.x * p.per, t1 = t0 + p.per < p.ntiles ? t0 + p.per : p.ntiles;
  for (uint64_t tile = t0; tile < t1; tile++) {
    const uint64_t tb = tile * MK_RS_TILE;
    const uint32_t nvalid = p.n - tb < (uint64_t)MK_RS_TILE ? (uint32_t)(p.n - tb) : MK_RS_TILE;
    uint32_t k[MK_RS_SC_ITEMS], v[MK_RS_SC_ITEMS], rank[MK_RS_SC_ITEMS];
    /* fresh copies per tile: what the items derive from them is one OR each, and is not to be kept (spilled) across the loop */
    uint32_t ln = lane, tt = t;
    asm volatile("" : "+v"(ln), "+v"(tt));
    const uint32_t wbase = (tt >> 6) * PER_WAVE + ln; /* the wave's first element + lane */
    /* a wave takes 1024 consecutive elements, item i = the i-th 64 of them: (wave, item, lane) is index order */
    const uint32_t *kt = key + tb, *vt = val + tb; /* (wave-uniform base + a 32-bit offset per item) */
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_SC_ITEMS; i++) { /* all loads of the tile in flight; past the end: element 0 again, dropped below */
      const uint32_t j = wbase + i * 64u, jj = j < nvalid ? j : 0u;
      k[i] = kt[jj];
      v[i] = vt[jj];
    }
    if (nvalid != MK_RS_TILE) {
#pragma unroll
      for (uint32_t i = 0; i < MK_RS_SC_ITEMS; i++)
        if (wbase + i * 64u >= nvalid) k[i] = 0xFFFFFFFFu;
    }
    /* (the wave's own row: its readers in the other waves are behind the previous tile's barrier C) */
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) wc[wave][j * 64u + ln] = 0u;
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_SC_ITEMS; i++) {
      const bool ok = wbase + i * 64u < nvalid;
      const uint32_t d = (k[i] >> shift) & 255u;
      /* the lanes of this wave with the same digit (and an element at all), as two 32-bit halves.  Per bit: the lane's bit spread
       * over a word (v_bfe_i32: 0 or ~0), the wave's lanes with the bit set (one compare into a scalar pair), and per half
       * same &= ~(set ^ mine) -- four VALU instructions a bit; the plain C of it (select m or ~m per lane, 64 bits wide) compiled
       * to ten */
      const uint64_t has = __builtin_amdgcn_ballot_w64(ok);
      uint32_t lo = (uint32_t)has, hi = (uint32_t)(has >> 32);
#pragma unroll
      for (uint32_t b = 0; b < 8u; b++) {
        const uint32_t mine = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1u);
        const uint64_t m = __builtin_amdgcn_ballot_w64(mine != 0u);
        lo &= ~((uint32_t)m ^ mine);
        hi &= ~((uint32_t)(m >> 32) ^ mine);
      }
      const uint32_t r = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u)); /* lanes below with the same digit */
      const uint32_t prev = wc[wave][d]; /* what the wave's earlier items hold of this digit (LDS is in order inside a wave) */
      rank[i] = prev + r;
      if (ok && r == 0u) wc[wave][d] = prev + (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
      __builtin_amdgcn_sched_barrier(0); /* item by item: interleaved, the sixteen ballot chains do not fit the registers */
    }
    __syncthreads(); /* A */
    if (wave == 0u) { /* four digits per lane: prefix over the waves, where the digit starts inside the tile, where it goes in HBM;
                         the bases move on */
      uint32_t c[4], sum = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t d = lane * 4u + j;
        uint32_t run = 0;
#pragma unroll
        for (uint32_t w = 0; w < WAVES; w++) { const uint32_t x = wc[w][d]; wc[w][d] = run; run += x; }
        c[j] = run;
        sum += run;
      }
      uint32_t incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if ((int)lane >= o) incl += u;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t d = lane * 4u + j;
        dstart[d] = run;
        gofs[d] = base[j] - run; /* (mod 2^64: every use adds a place >= run) */
        base[j] += c[j];
        run += c[j];
      }
    }
    __syncthreads(); /* B */
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_SC_ITEMS; i++) {
      if (wbase + i * 64u >= nvalid) continue;
      const uint32_t d = (k[i] >> shift) & 255u;
      const uint32_t place = dstart[d] + wc[wave][d] + rank[i];
      lk[place] = k[i];
      lv[place] = v[i];
    }
    __syncthreads(); /* C */
#pragma unroll
    for (uint32_t i = 0; i < MK_RS_SC_ITEMS; i++) {
      const uint32_t j = i * MK_RS_SC_THREADS + tt;
      if (j >= nvalid) continue;
      const uint32_t kk = lk[j], vv = lv[j];
      const unsigned long long pos = gofs[(kk >> shift) & 255u] + j;
      okey[pos] = kk;
      oval[pos] = vv;
    }
    /* (the next tile writes gofs / dstart after its barrier A and lk / lv after its barrier B: everyone is through here by then) */
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
      hipLaunchKernelGGL(mk_rs_scatter_kernel, dim3(p.B), dim3(MK_RS_SC_THREADS), 0, st, (const uint32_t *)key[cur], (const uint32_t *)val[cur],
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
