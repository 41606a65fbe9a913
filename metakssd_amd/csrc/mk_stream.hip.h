/*
 * mk_stream.hip.h -- device side of mk_sketch_push_stream(): FASTA text -> base stream, on the GPU (gfx950, wave64).
 *
 * fasta2co() / uniq_fasta2co() (iseq2comem.c:240-279, :751-790) walk the file byte by byte: '\n' and '\r' are skipped WITHOUT
 * resetting the k-mer window (:257), a '>' skips to the end of its line and resets (:259-274), any other byte that is no base
 * resets (:258, :275-279).  The host front end (mk_fasta_window, mk_frontend.c) does that walk with one branch per character
 * -- 0.3 GB/s per thread, which bounds BASELINE config 5.  Here the raw file bytes go to HBM and three small kernels turn them
 * into the same contiguous BASE STREAM the host walk produces:
 *     kept      : every byte that is neither '\n' nor '\r' nor inside a header line; a header line's '>' stays as ONE byte
 *                 (it is no base: the scan kernel resets the window on it, exactly like the reference's base = 1)
 *     dropped   : '\n', '\r', and everything from behind a line's first '>' up to and including its '\n'
 * "inside a header" is a two-state machine (normal / header) driven by '>' and '\n', i.e. a scan with the monoid of
 * state-transfer functions; the output position of a byte is a prefix count that depends on the state its segment is
 * entered in.  So:
 *   mk_fa_summary_kernel  one wave per segment of MK_FA_SEG bytes: {state after, bytes kept} for BOTH entry states
 *   mk_fa_scan_kernel     one workgroup: composes the segment summaries in order -> entry state and output offset of every
 *                         segment, the new stream length and the state behind the text (carried to the next push)
 *   mk_fa_emit_kernel     the same walk again with the entry state known: kept bytes to stream[offset ..]
 * The scan kernel then reads OVERLAPPING VIRTUAL ROWS straight out of the stream (row i = stream + i * pitch, rowlen =
 * pitch + TL - 1 bytes: every k-mer ends in exactly one row) -- no host byte loop, no second copy of the bases.
 * A byte per lane and ballots: 64 bytes per wave step.  A 4 MB genome is 65 536 steps of about 40 instructions over 256 CUs.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MK_FA_SEG 1024u /* text bytes per wave: 16 steps of 64 bytes (a 4 MB genome gives 4096 waves) */

/* device-resident state of one engine's stream */
struct mk_fa_state {
  unsigned long long len;      /* bytes in the stream buffer */
  unsigned long long nrows;    /* virtual rows the next scan launch may take (mk_fa_scan_kernel) */
  uint32_t in_header;          /* state behind the text pushed so far */
  uint32_t pad;
};

struct mk_fa_sum { /* per segment */
  uint32_t cnt[2]; /* bytes kept when the segment is entered in state 0 (normal) / 1 (header) */
  uint32_t after;  /* bit s: state behind the segment when entered in state s */
  uint32_t off;    /* scan: output offset of the segment (relative to the stream length before this push) | entry state << 31 */
};

/* one 64-byte step: which lanes' bytes are kept, for a step entered in state `st` (wave-uniform); returns the state behind it */
__device__ __forceinline__ uint64_t mk_fa_step(uint8_t ch, bool valid, uint32_t lane, uint32_t &st) {
  const uint64_t gt = __ballot(valid && ch == '>');
  const bool skipch = !valid || ch == '\n' || ch == '\r';
  /* sequence lines only (no '>' in the step, not inside a header): every byte but the line ends stays -- nearly every step of a genome */
  if (gt == 0ull && st == 0u) return __ballot(!skipch);
  const uint64_t nl = __ballot(valid && ch == '\n');
  const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull; /* lanes in front of this one */
  const uint64_t nlb = nl & below;
  /* bytes of this lane's line in front of it: behind the last '\n' below the lane (the whole step when there is none) */
  const uint64_t line = nlb ? below & ~((2ull << (63u - (uint32_t)__builtin_clzll(nlb))) - 1ull) : below;
  /* entered in header state: still inside that header until the first '\n' of the step */
  const bool carried = st && nlb == 0ull;
  const bool in_hdr = carried || (gt & line) != 0ull;
  const uint64_t keep = __ballot(!skipch && !in_hdr);
  /* state behind the step: a '>' behind the last '\n'; without any '\n' the entry state persists unless a '>' raises it */
  if (nl) {
    const uint64_t tail = ~((2ull << (63u - (uint32_t)__builtin_clzll(nl))) - 1ull);
    st = (gt & tail) != 0ull ? 1u : 0u;
  } else if (gt) st = 1u;
  return keep;
}

/* A wave's segment of text, [lo, lo + MK_FA_SEG) with lo a multiple of 16, comes in with ONE 16-byte load per lane and is laid
 * into the wave's 1 KiB of LDS; the walk then reads a byte per lane and step from there (64 consecutive bytes: conflict-free).  A
 * byte load per lane and step from global memory instead is sixteen dependent round trips per segment: the two passes over a
 * batch of 128 MB took 228 + 177 us that way (profiles/r04_a_config5_*).  Bytes from `hi` on are never looked at (the steps mask them);
 * the buffers end in slack, so the load itself may run past `hi`. */
#define MK_FA_WAVES 4u /* waves (segments) per workgroup of the summary and emit kernels */
__device__ __forceinline__ const uint8_t *mk_fa_stage(const uint8_t *text, uint64_t lo, uint32_t lane, uint4 *seg_lds) {
  seg_lds[lane] = *(const uint4 *)(text + lo + 16u * lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return (const uint8_t *)seg_lds;
}

/* summary of the text bytes [lo, hi) (one wave): {state behind, bytes kept} for both entry states */
__device__ __forceinline__ mk_fa_sum mk_fa_summarise(const uint8_t *text, uint64_t lo, uint64_t hi, uint32_t lane, uint4 *seg_lds) {
  uint32_t s0 = 0u, s1 = 1u, c0 = 0u, c1 = 0u;
  if (hi > lo) {
    const uint8_t *seg = mk_fa_stage(text, lo, lane, seg_lds);
    const uint32_t n = (uint32_t)(hi - lo);
#pragma unroll
    for (uint32_t j = 0; j < MK_FA_SEG / 64u; j++) {
      if (64u * j >= n) break;
      const bool valid = 64u * j + lane < n;
      const uint8_t ch = seg[64u * j + lane];
      uint32_t a = s0, b = s1;
      const uint64_t k0 = mk_fa_step(ch, valid, lane, a);
      const uint64_t k1 = s1 == s0 ? k0 : mk_fa_step(ch, valid, lane, b);
      if (s1 == s0) b = a;
      c0 += (uint32_t)__popcll(k0);
      c1 += (uint32_t)__popcll(k1);
      s0 = a; s1 = b;
    }
  }
  mk_fa_sum r;
  r.cnt[0] = c0; r.cnt[1] = c1; r.after = s0 | (s1 << 1); r.off = 0u;
  return r;
}

__global__ void __launch_bounds__(256) mk_fa_summary_kernel(const uint8_t *text, uint64_t n, mk_fa_sum *sum) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t seg = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t lo = seg * MK_FA_SEG;
  if (lo >= n) return;
  const uint64_t hi = lo + MK_FA_SEG < n ? lo + MK_FA_SEG : n;
  __shared__ uint4 seg_lds[MK_FA_WAVES][64];
  const mk_fa_sum r = mk_fa_summarise(text, lo, hi, lane, seg_lds[threadIdx.x >> 6]);
  if (lane == 0) sum[seg] = r;
}

/* Composes nseg summaries in segment order, entered in state `state0` (a workgroup of 1024 threads, all of them call; a
 * wave-shuffle scan of the transfer functions per 1024 segments): sum[i].off := output offset of segment i (relative to the
 * first) | its entry state << 31; kept := bytes kept in all, state := the state behind the last segment (in every thread). */
__device__ __forceinline__ void mk_fa_compose(mk_fa_sum *sum, uint64_t nseg, uint32_t state0, unsigned long long &kept, uint32_t &state) {
  __shared__ uint32_t w_after[16], w_c0[16], w_c1[16];
  __shared__ uint32_t carry_state;
  __shared__ unsigned long long carry_off;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads(); /* (a second call in one kernel must not overwrite what the first one's threads still read) */
  if (threadIdx.x == 0) { carry_state = state0; carry_off = 0ull; }
  __syncthreads();
  for (uint64_t base = 0; base < nseg; base += 1024u) {
    const uint64_t i = base + threadIdx.x;
    /* element = transfer function (after: bit s = state behind when entered in s) + kept bytes for either entry state */
    uint32_t after = 2u, c0 = 0u, c1 = 0u; /* identity: state unchanged, nothing kept */
    if (i < nseg) { after = sum[i].after; c0 = sum[i].cnt[0]; c1 = sum[i].cnt[1]; }
    const uint32_t own_after = after, own_c0 = c0, own_c1 = c1;
    /* inclusive scan over the wave: (earlier) then (this) */
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
      const uint32_t pa = __shfl_up(after, o), p0 = __shfl_up(c0, o), p1 = __shfl_up(c1, o);
      if (lane >= o) {
        /* entered in s: the earlier part leaves state e = pa bit s, having kept p_s; this part adds its count for e */
        const uint32_t e0 = pa & 1u, e1 = (pa >> 1) & 1u;
        const uint32_t n0 = p0 + (e0 ? c1 : c0), n1 = p1 + (e1 ? c1 : c0);
        const uint32_t na = ((after >> e0) & 1u) | (((after >> e1) & 1u) << 1);
        c0 = n0; c1 = n1; after = na;
      }
    }
    if (lane == 63u) { w_after[wave] = after; w_c0[wave] = c0; w_c1[wave] = c1; }
    __syncthreads();
    /* this thread's segment is entered in the state the carry and everything in front of it lead to */
    uint32_t s = carry_state;
    unsigned long long off = carry_off;
    for (uint32_t w = 0; w < wave; w++) {
      off += s ? w_c1[w] : w_c0[w];
      s = (w_after[w] >> s) & 1u;
    }
    /* exclusive part inside the wave: inclusive result without the own element = the previous lane's inclusive result */
    const uint32_t ia = __shfl_up(after, 1), i0 = __shfl_up(c0, 1), i1 = __shfl_up(c1, 1);
    uint32_t es = s;
    unsigned long long eoff = off;
    if (lane) { eoff += s ? i1 : i0; es = (ia >> s) & 1u; }
    if (i < nseg) sum[i].off = (uint32_t)eoff | (es << 31);
    __syncthreads();
    if (threadIdx.x == 1023u) { /* behind the last element of this round */
      carry_off = eoff + (es ? own_c1 : own_c0);
      carry_state = (own_after >> es) & 1u;
    }
    __syncthreads();
  }
  kept = carry_off;
  state = carry_state;
}

/* Composes the summaries in segment order (one workgroup).  Also: stream length behind this push, state behind the text, and
 * the number of virtual rows the scan may take now -- with `final` every row that holds a complete k-mer, otherwise only rows
 * that are complete (their last byte is known). */
__global__ void __launch_bounds__(1024) mk_fa_scan_kernel(mk_fa_sum *sum, uint64_t nseg, mk_fa_state *st, uint8_t *stream, uint32_t pitch,
                                                          uint32_t rowlen, uint32_t TL, int final, uint32_t *err) {
  unsigned long long kept;
  uint32_t state;
  mk_fa_compose(sum, nseg, st->in_header, kept, state);
  __syncthreads(); /* (every thread has read st->in_header before thread 0 writes it) */
  if (threadIdx.x == 0) {
    const unsigned long long len = st->len + kept;
    st->len = len;
    st->in_header = state;
    if (final && state) atomicOr(err, 8u); /* the text ends inside a '>' line: the reference gives up (iseq2comem.c:259-271) */
    unsigned long long rows = 0;
    if (final) {
      if (len >= TL) rows = (len - (TL - 1u) + pitch - 1u) / pitch; /* every row with a complete k-mer in it */
      stream[len] = (uint8_t)'\n'; /* the last row ends here */
    } else if (len >= rowlen) rows = (len - rowlen) / pitch + 1u; /* complete rows only */
    st->nrows = rows;
  }
}

/* the text bytes [lo, hi) once more with the entry state known (off: mk_fa_sum::off of the segment): kept bytes to
 * stream[base + offset ..] (one wave) */
__device__ __forceinline__ void mk_fa_emit_seg(const uint8_t *text, uint64_t lo, uint64_t hi, uint32_t o, uint8_t *stream, uint64_t base, uint32_t lane,
                                               uint4 *seg_lds) {
  uint32_t s = o >> 31;
  uint64_t off = base + (o & 0x7FFFFFFFu);
  if (hi <= lo) return;
  const uint8_t *seg = mk_fa_stage(text, lo, lane, seg_lds);
  const uint32_t n = (uint32_t)(hi - lo);
#pragma unroll
  for (uint32_t j = 0; j < MK_FA_SEG / 64u; j++) {
    if (64u * j >= n) break;
    const bool valid = 64u * j + lane < n;
    const uint8_t ch = seg[64u * j + lane];
    const uint64_t keep = mk_fa_step(ch, valid, lane, s);
    if ((keep >> lane) & 1ull) stream[off + mk_mbcnt(keep)] = ch;
    off += (uint64_t)__popcll(keep);
  }
}

__global__ void __launch_bounds__(256) mk_fa_emit_kernel(const uint8_t *text, uint64_t n, const mk_fa_sum *sum, uint8_t *stream,
                                                         unsigned long long stream_len_before) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t seg = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t lo = seg * MK_FA_SEG;
  if (lo >= n) return;
  const uint64_t hi = lo + MK_FA_SEG < n ? lo + MK_FA_SEG : n;
  __shared__ uint4 seg_lds[MK_FA_WAVES][64];
  mk_fa_emit_seg(text, lo, hi, sum[seg].off, stream, stream_len_before, lane, seg_lds[threadIdx.x >> 6]);
}

/* behind a scan of `rows_taken` rows (device value): the unscanned tail moves to the front of the stream buffer */
__global__ void __launch_bounds__(1024) mk_fa_shift_kernel(uint8_t *stream, uint8_t *tmp, mk_fa_state *st, uint32_t pitch, int phase) {
  const unsigned long long taken = st->nrows * (unsigned long long)pitch;
  const unsigned long long len = st->len;
  const unsigned long long rest = len > taken ? len - taken : 0ull;
  const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, step = (unsigned long long)gridDim.x * blockDim.x;
  if (phase == 0) { for (unsigned long long i = i0; i < rest; i += step) tmp[i] = stream[taken + i]; }
  else {
    for (unsigned long long i = i0; i < rest; i += step) stream[i] = tmp[i];
    if (i0 == 0) { /* (the grid's other threads only read st in this phase: rest was computed before this write by each) */ }
  }
}
__global__ void mk_fa_shift_done_kernel(mk_fa_state *st, uint32_t pitch) {
  const unsigned long long taken = st->nrows * (unsigned long long)pitch;
  st->len = st->len > taken ? st->len - taken : 0ull;
  st->nrows = 0ull;
}
