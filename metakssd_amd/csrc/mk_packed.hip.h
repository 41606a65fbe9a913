/*
 * mk_packed.hip.h -- the scan kernel over PACKED rows (MK_ROWS_PACKED, include/metakssd_hip.h): 64 bytes a read -- two bits a base
 * in the scan loop's own coding, one validity bit a base -- made by the host's framer threads (mk_row_pack, mk_frontend.c) so that
 * a 150-base read crosses PCIe as 64 bytes instead of 160.  gfx950, wave64.
 *
 * Same per-read walk as mk_scan_kernel's tuned loop (iseq2comem.c:682-690: a base counts when the TL bytes up to it are bases of
 * one read), same filters, same 16-byte pair records for the same resolve kernel; what falls away is everything that turns text
 * into codes: a lane loads ITS row with four 16-byte loads (a tile of 64 rows is 4 KiB of consecutive memory), the row stays in
 * sixteen registers -- no LDS tile, no transposition, no decode, no newline search.  The 8-base window the loop's funnel shift
 * wants is a half of one of those registers.
 *   A  every lane's read has the same length and only valid bases (the flag in the row header: fixed-length reads without N):
 *      window count, first complete k-mer and the bases that count are scalars; no validity test at all.
 *   B  anything else: per lane, from the window's validity byte -- e = bases in front of the first invalid one, the run length
 *      restarts behind the last invalid one; bases behind a read's end are invalid, so a shorter read simply stops producing.
 */
#pragma once
#include "mk_kernels.hip.h"
#include "mk_batch.hip.h"

#include <utility>

#define MK_PACKED_PITCH_DEV 64u
#define MK_PACKED_WINDOWS 19u

/* the filter test of one 8-base window: fstart = low word of the forward k-mer in front of the window, lo = its eight codes in the
 * top 16 bits.  Returns whether some base (pair) of the window passes; flo := the low word behind the window; pa = what the next
 * window's probes need of this one (see mk_scan_kernel: pair probing for subk 6, one probe per base for subk 5). */
template <int K, int SUBK>
__device__ __forceinline__ bool mk_probe8(const uint32_t fstart, const uint32_t lo, uint32_t (&pa)[3], uint32_t &flo) {
  constexpr uint32_t SH = 2u * (K - SUBK) - 2u;
  constexpr uint32_t D = SUBK == 6 ? (SH - 2u) / 2u : 1u;
  const uint32_t f0 = __builtin_amdgcn_alignbit(fstart, lo, 30), f1 = __builtin_amdgcn_alignbit(fstart, lo, 28);
  const uint32_t f2 = __builtin_amdgcn_alignbit(fstart, lo, 26), f3 = __builtin_amdgcn_alignbit(fstart, lo, 24);
  const uint32_t f4 = __builtin_amdgcn_alignbit(fstart, lo, 22), f5 = __builtin_amdgcn_alignbit(fstart, lo, 20);
  const uint32_t f6 = __builtin_amdgcn_alignbit(fstart, lo, 18), f7 = __builtin_amdgcn_alignbit(fstart, lo, 16);
  const uint32_t bj[9] = {fstart, f0, f1, f2, f3, f4, f5, f6, f7};
  bool fired;
  if constexpr (SUBK == 6) {
    uint32_t mm[4], dd[4];
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
      const uint32_t wsrc = bj[2u * t + 1u];
      dd[t] = *(mk_lds_cu32)(uintptr_t)(((wsrc >> (SH + 8u)) & ((MK_ZF_WORDS - 1u) << 2)) + MK_ZMASK_WORDS * 4u);
      mm[t] = *(mk_lds_cu32)(uintptr_t)(2u * t >= D ? bj[2u * t - D] & 0x3FCu : pa[2u * t]);
    }
#pragma unroll
    for (uint32_t k = 0; k < D; k++) pa[k] = bj[8u - D + k] & 0x3FCu;
    flo = f7;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const uint32_t tt0 = mm[0] & ~dd[0], tt1 = mm[1] & ~dd[1], tt2 = mm[2] & ~dd[2], tt3 = mm[3] & ~dd[3];
    fired = min(min(tt0, tt1), min(tt2, tt3)) == 0u;
  } else {
    uint32_t wd[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      uint32_t addr;
      asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(bj[j + 1u]), "v"(0xFFFCu));
      wd[j] = *(mk_lds_cu32)(uintptr_t)addr;
    }
    const uint32_t bprev = pa[0];
    pa[0] = f6;
    flo = f7;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint32_t sh[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      const uint32_t src = j == 0 ? bprev : bj[j - 1u];
      asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(sh[j]) : "v"(src), "v"(wd[j]));
    }
    fired = (((sh[0] | sh[1] | sh[2]) | (sh[3] | sh[4] | sh[5]) | (sh[6] | sh[7])) & 1u) != 0u;
  }
  return fired;
}

template <typename F, uint32_t... P>
__device__ __forceinline__ void mk_unrolled_windows(F &&body, std::integer_sequence<uint32_t, P...>) {
  (body(std::integral_constant<uint32_t, P>{}), ...);
}

/* WIDE: MK_ROWS_WIDE rows (mk_fasta_pack_rows, mk_frontend.c) -- 240 bases, thirty windows, the codes in dwords 1..15; the validity
 * bytes of the few rows that have a byte that is no base come from the EXTENSION ROW behind them (bytes 16..45 of the next 64 bytes;
 * as a row it holds zero bases), every other row's validity is its length.
 * In a batch (a.batch) a row that lies in no file's range -- what a file left free of its place -- is not loaded at all. */
template <int K, int SUBK, bool WIDE = false>
__global__ void __launch_bounds__(1024) mk_scan_packed_kernel(const mk_scan_args a) {
  extern __shared__ __align__(16) uint32_t lds[];
  constexpr uint32_t WAVES = 16u;
  constexpr uint32_t NWIN = WIDE ? 30u : MK_PACKED_WINDOWS;
  constexpr uint32_t MTW = SUBK == 6 ? MK_ZMASK_WORDS : 0u;
  constexpr uint32_t SH = 2u * (K - SUBK) - 2u;
  constexpr uint32_t D = SUBK == 6 ? (SH - 2u) / 2u : 1u;
  constexpr uint32_t TL = 2u * K;
  uint32_t *bitmap = lds + MTW;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if constexpr (SUBK == 6) mk_build_zfilter(lds, bitmap, a);
  else mk_build_xfilter(bitmap, a);
  const uint32_t wave_global = blockIdx.x * WAVES + wave, nwaves = gridDim.x * WAVES;
  const uint32_t filter_base = (uint32_t)(uintptr_t)(mk_lds_cu32)bitmap;
  if (filter_base != MTW * 4u || a.mt_words != MTW || a.bm_words != 16384u || a.dimmask != (SUBK == 6 ? 0xFFFFFFu : 0xFFFFFu)) {
    if (threadIdx.x == 0) atomicOr(&a.tab.err[0], 4u);
    if (lane == 0) a.cand_count[wave_global] = 0u;
    return;
  }
  const mk_scan_args *ka = (const mk_scan_args *)__builtin_amdgcn_kernarg_segment_ptr();
  const uint64_t nreads = a.nreads;
  const uint32_t ntiles = (uint32_t)((nreads + 63u) >> 6);
  uint4 *const my_cand = a.cand + (size_t)__builtin_amdgcn_readfirstlane(wave_global) * a.cand_cap;
  if (wave_global >= ntiles) {
    if (lane == 0) a.cand_count[wave_global] = 0u;
    return;
  }
  uint32_t qn = 0; /* candidates appended to this wave's buffer so far (wave-uniform) */
  auto push_pair = [&](const bool hit, const uint64_t hm, const uint4 r) {
    const uint32_t cnt = (uint32_t)__popcll(hm);
    if (qn + cnt > a.cand_cap) { mk_resolve_inline(ka, hit, r, nullptr); return; } /* buffer full (dense tables only) */
    if (hit) *(uint4 *)((uint8_t *)my_cand + (qn + mk_mbcnt(hm)) * 16u) = r;
    qn += cnt;
  };
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  auto load_row = [&](uint32_t tile, uint4 (&r)[4]) {
    const uint64_t row = ((uint64_t)tile << 6) + lane;
    bool there = row < nreads;
    if (a.batch && there) { /* (wave-uniform test; the search is a dozen loads out of a table of at most 4 KiB, once per tile) */
      const mk_batch_dev &b = *a.batch;
      const uint32_t grow = (uint32_t)(a.first_ord + row), f = mk_b_find(b.row0s, b.nfiles, grow);
      there = grow - b.row0s[f] < b.files[f].nrow;
    }
    if (there) {
      const uint4 *p = (const uint4 *)(a.rows + row * MK_PACKED_PITCH_DEV);
      r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3];
    } else { r[0] = zero4; r[1] = zero4; r[2] = zero4; r[3] = zero4; } /* no read: zero bases */
  };
  uint4 nx[4];
  load_row(wave_global, nx);
  for (uint32_t tile = wave_global; tile < ntiles; tile += nwaves) {
    const uint4 r0 = nx[0], r1 = nx[1], r2 = nx[2], r3 = nx[3];
    if (tile + nwaves < ntiles) load_row(tile + nwaves, nx); /* the next tile's rows are on their way while this one is walked */
    const uint32_t cw[15] = {r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w}; /* (narrow rows: ten of them) */
    const uint32_t nb = r0.x & 0xFFFFu;
    /* validity bytes, four windows a dword: narrow rows carry them; a wide row has them in its extension row, or needs none */
    uint32_t vw[8] = {r2.w, r3.x, r3.y, r3.z, r3.w, 0u, 0u, 0u};
    if constexpr (WIDE) {
      const bool ext = (r0.x & 0x20000u) != 0u;
      uint4 e0 = zero4, e1 = zero4;
      if (ext) {
        const uint4 *p = (const uint4 *)(a.rows + (((uint64_t)tile << 6) + lane + 1u) * MK_PACKED_PITCH_DEV);
        e0 = p[1]; e1 = p[2];
      }
      const uint32_t ev[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
      for (uint32_t q = 0; q < 8u; q++) {
        /* no extension row: bases 32q .. 32q + 31 are valid as far as the row reaches */
        const uint32_t have = nb > 32u * q ? nb - 32u * q : 0u;
        const uint32_t len = have >= 32u ? 0xFFFFFFFFu : (1u << have) - 1u;
        vw[q] = ext ? ev[q] : len;
      }
    }
    const uint32_t nb0 = __builtin_amdgcn_readfirstlane(nb);
    const bool uniform = __all(nb == nb0 && (r0.x & 0x10000u) != 0u);
    const uint32_t rowidx = (tile << 6) + lane;
    uint32_t flo = 0u, hh = 0u;
    uint32_t pa[3] = {0u, 0u, 0u};
    auto oh_init = [&]() {
      if constexpr (SUBK == 6) {
#pragma unroll
        for (uint32_t k = 0; k < D; k++) pa[k] = (flo >> (2u * (D - k))) & 0x3FCu;
      } else pa[0] = flo >> 2;
    };
    auto window = [&](const uint32_t p, const uint32_t lo, const bool rollonly, const uint32_t jmin, const uint32_t e, const bool live, const uint64_t livem) {
      const uint32_t fstart = flo;
      if (rollonly) {
        flo = __builtin_amdgcn_alignbit(fstart, lo, 16);
        oh_init();
      } else {
        const bool fired = mk_probe8<K, SUBK>(fstart, lo, pa, flo);
        const bool hit = fired && live;
        const uint64_t hm = __builtin_amdgcn_ballot_w64(fired) & livem;
        if (hm) push_pair(hit, hm, make_uint4(fstart, (lo & 0xFFFF0000u) | p | (jmin << 9) | (e << 12), hh, rowidx));
      }
      hh = __builtin_amdgcn_perm(hh, fstart, 0x05040100u); /* hh << 16 | fstart & 0xFFFF */
    };
    /* the nineteen windows written out (the row's registers are indexed by constants): a window past the longest read is skipped */
    auto for_windows = [&](auto &&body) {
      mk_unrolled_windows(body, std::make_integer_sequence<uint32_t, NWIN>{});
    };
    if (uniform) { /* ---- A: one length, every base valid */
      for_windows([&](auto pc) {
        constexpr uint32_t p = decltype(pc)::value;
        if (8u * p < nb0) {
          const uint32_t lo = (p & 1u) ? cw[p >> 1] << 16 : cw[p >> 1];
          const uint32_t e = nb0 - 8u * p < 8u ? nb0 - 8u * p : 8u;
          constexpr uint32_t urun = 8u * p; /* every window in front of this one was full */
          const bool rollonly = urun + e < TL;
          constexpr uint32_t jmin = urun + 1u >= TL ? 0u : TL - 1u - urun;
          window(p, lo, rollonly, jmin & 7u, e, true, __builtin_amdgcn_read_exec());
        }
      });
    } else { /* ---- B: per lane, from the validity bytes */
      /* the longest read of the tile bounds the walk (a wave-wide maximum of sixteen-bit values: two ballots of halves would do;
       * the plain butterfly is a handful of instructions per TILE) */
      uint32_t nbmax = nb;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(nbmax, o); nbmax = t > nbmax ? t : nbmax; }
      nbmax = __builtin_amdgcn_readfirstlane(nbmax);
      uint32_t run = 0;
      for_windows([&](auto pc) {
        constexpr uint32_t p = decltype(pc)::value;
        if (8u * p < nbmax) {
          const uint32_t lo = (p & 1u) ? cw[p >> 1] << 16 : cw[p >> 1];
          const uint32_t v = (vw[p >> 2] >> (8u * (p & 3u))) & 0xFFu;
          const uint32_t inval = v ^ 0xFFu;
          const uint32_t e = inval ? (uint32_t)__builtin_ctz(inval) : 8u;
          const uint32_t jm = run + 1u >= TL ? 0u : TL - 1u - run; /* >= e: nothing of this lane counts here */
          const bool say = jm < e;
          window(p, lo, false, jm & 7u, e, say, __builtin_amdgcn_ballot_w64(say));
          run = inval ? (uint32_t)__builtin_clz(inval << 24) : (run + 8u > 0xFFFFu ? 0xFFFFu : run + 8u); /* bases behind the last invalid one */
        }
      });
    }
  }
  if (lane == 0) a.cand_count[wave_global] = qn;
}
