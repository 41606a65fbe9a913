/*
 * mk_batch.hip.h -- many small inputs in ONE launch sequence (gfx950, wave64): BASELINE config 5, a directory of genomes.
 *
 * The reference sketches one file per OpenMP thread, every thread with a hash table of its own (command_dist.c:344-372:
 * CO[tid], fasta2co(), wrt_co2cmpn_use_inn_subctx()).  A 4 Mbase genome is 4 MB of text and leaves a thousand (L3K10) to
 * sixteen thousand (L2K11) keys: sketched alone it is sixteen commands of 4-50 us each on a chip that wants tens of megabytes per
 * launch (profiles/r03_c_config5_*: 1 % of the HBM roofline per kernel).  Here B files travel together:
 *
 *   text of all files, one copy            -> mk_fab_summary / mk_fab_scan / mk_fab_emit: the FASTA walk of mk_stream.hip.h per file
 *   one base stream, a region per file     -> the scan kernel over ALL virtual rows (unchanged: regions are row-aligned and padded
 *                                             with '\n', so no row and no k-mer spans two files)
 *   candidates                              -> the resolve kernel finds the file of a candidate from its row and upserts into that
 *                                             file's own table (2^tb slots each, side by side)
 *   tables                                  -> mk_b_compact: one key list for the batch {key, first ordinal, count, file}
 *   reference slot order per file           -> mk_b_layout: priority insertion (mk_layout_kernel's algorithm) into a VIRTUAL
 *                                             hashsize-slot table per file, held as an open-addressed map (slot -> key index)
 *                                             of 2^tb entries: the reference's table geometry without its 8 MB .. 2 GB per file
 *   ordered dump                            -> mk_b_bucket / mk_b_bscan / mk_b_scatter / mk_b_emit: (file, component, slot) order
 *                                             by a bucket pass, as the key-list dump of mk_kernels.hip.h
 *
 * A file whose table overflows, holds more keys than the reference admits, or whose layout does not converge is flagged and
 * sketched alone by the engine afterwards (mk_sketch_batch_end): the batch never changes a result, it only groups work.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mk_kernels.hip.h"
#include "mk_stream.hip.h"

#define MK_BF_HEADER_END 1u /* the text ends inside a '>' line: the reference gives up (iseq2comem.c:259-271) */
#define MK_BF_REDO 2u       /* table full / more keys than the batch admits / layout gave up: the engine sketches the file alone */
#define MK_B_PROBES 256u    /* linear probes a key may take in its file's table before the file is flagged */

struct mk_bfile { /* per file, made by the host */
  unsigned long long text_off, text_len; /* bytes into the batch's text buffer (text_off is a multiple of MK_FA_SEG) */
  unsigned long long stream_off;         /* its region of the base stream: row0 * pitch ..  */
  unsigned long long stream_cap;         /* .. of this many bytes: a multiple of the pitch, > text_len (at least one '\n' closes it) */
  uint32_t seg0, nseg;                   /* its text segments [seg0, seg0 + nseg), nseg >= 1 */
  uint32_t row0, nrow;                   /* its virtual rows [row0, row0 + nrow) */
};
struct mk_bstat { /* per file, made by the device */
  unsigned long long kept; /* bytes of its base stream */
  uint32_t flags;          /* MK_BF_* */
  uint32_t D;              /* distinct keys in its table */
  uint32_t pad[2];
};

struct mk_batch_dev {
  const mk_bfile *files;
  const uint32_t *seg0s, *row0s; /* [nfiles + 1]: first segment / first row of every file, and the totals */
  mk_bstat *stat;                /* [nfiles] */
  uint32_t *ctot;                /* [nfiles * comp_num]: ids per (file, component) */
  unsigned long long *misc;      /* [0] keys in the batch's list, [1] ids in the output */
  uint32_t *nout;                /* [nfiles * 16]: ids a file contributes to the output, a 64-byte line per file (every wave of the bucket
                                  * pass adds to it: kept away from the status words everybody reads) */
  uint32_t nfiles, tb;           /* every file's table and map have 1 << tb entries */
  unsigned long long *kc, *ordinv; /* [nfiles << tb]: the accumulation tables, slot format of mk_table */
  unsigned long long *map;       /* [nfiles << tb]: virtual slot << 32 | key index, ~0 = empty */
  /* key list of the batch */
  unsigned long long *key, *ord;
  uint32_t *cnt, *gid;
  uint64_t list_cap;
  uint32_t key_limit;            /* most keys a file may hold here: min(half its table, the reference's hashlimit) */
  uint32_t S;                    /* the reference's hashsize: geometry of the virtual table */
  uint32_t comp_num, comp_code_bits, cnt_hi; /* cnt_hi = 1: keys seen more than once are dropped at the dump (uniq_fasta2co) */
  /* bucket pass */
  uint32_t shift, bpc, bpf;      /* bucket = file * bpf + component * bpc + (slot >> shift); bpf = comp_num * bpc */
  uint32_t *bstart;              /* [nfiles * bpf + 1]: counts, then exclusive starts over the whole batch */
  uint32_t *bcursor;             /* [nfiles * bpf] */
  unsigned long long *tkey;      /* bucket-ordered sort keys: (file * comp_num + component) << 32 | slot */
  uint32_t *tidx;                /* ... and their key indices */
  uint32_t *out_ids;
  uint64_t out_cap;
};

/* index of the last entry <= v in the ascending array a[0..n) (a[0] <= v) */
__device__ __forceinline__ uint32_t mk_b_find(const uint32_t *a, uint32_t n, uint32_t v) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

/* ---- clear: tables to 0, maps to ~0, status to 0, in one launch ------------------------------------------------------- */
__global__ void __launch_bounds__(256) mk_b_clear_kernel(uint4 *zero, unsigned long long nzero16, uint4 *ones, unsigned long long nones16,
                                                         uint4 *stat, unsigned long long nstat16) {
  const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, step = (unsigned long long)gridDim.x * blockDim.x;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u), o = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (unsigned long long i = i0; i < nzero16; i += step) zero[i] = z;
  for (unsigned long long i = i0; i < nones16; i += step) ones[i] = o;
  for (unsigned long long i = i0; i < nstat16; i += step) stat[i] = z;
}

/* ---- FASTA text -> base stream, per file (see mk_stream.hip.h for the walk) ---------------------------------------------- */
__global__ void __launch_bounds__(256) mk_fab_summary_kernel(const uint8_t *text, const mk_batch_dev b, uint32_t nseg_total, mk_fa_sum *sum) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t seg = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (seg >= nseg_total) return;
  const uint32_t f = mk_b_find(b.seg0s, b.nfiles, seg);
  const mk_bfile bf = b.files[f];
  const uint64_t lo = bf.text_off + (uint64_t)(seg - bf.seg0) * MK_FA_SEG, end = bf.text_off + bf.text_len;
  const uint64_t hi = lo + MK_FA_SEG < end ? lo + MK_FA_SEG : end;
  __shared__ uint4 seg_lds[MK_FA_WAVES][64];
  const mk_fa_sum r = mk_fa_summarise(text, lo, hi > lo ? hi : lo, lane, seg_lds[threadIdx.x >> 6]);
  if (lane == 0) sum[seg] = r;
}

/* one workgroup per file: its segments' transfer functions composed from state 0 (a file starts outside a header) */
__global__ void __launch_bounds__(1024) mk_fab_scan_kernel(mk_fa_sum *sum, const mk_batch_dev b, uint32_t TL, uint32_t pitch) {
  const mk_bfile bf = b.files[blockIdx.x];
  unsigned long long kept;
  uint32_t state;
  mk_fa_compose(sum + bf.seg0, bf.nseg, 0u, kept, state);
  if (threadIdx.x == 0) {
    mk_bstat &st = b.stat[blockIdx.x];
    st.kept = kept;
    if (state) atomicOr(&st.flags, MK_BF_HEADER_END);
    (void)TL; (void)pitch;
  }
}

/* kept bytes to the file's region of the stream; the rest of the region -- what the dropped bytes leave free, and the padding up
 * to the next file's first row -- is filled with '\n': a virtual row ends there, so no k-mer reaches into the next file */
__global__ void __launch_bounds__(256) mk_fab_emit_kernel(const uint8_t *text, const mk_batch_dev b, uint32_t nseg_total, const mk_fa_sum *sum,
                                                          uint8_t *stream) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t seg = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (seg >= nseg_total) return;
  const uint32_t f = mk_b_find(b.seg0s, b.nfiles, seg);
  const mk_bfile bf = b.files[f];
  const uint64_t lo = bf.text_off + (uint64_t)(seg - bf.seg0) * MK_FA_SEG, end = bf.text_off + bf.text_len;
  const uint64_t hi = lo + MK_FA_SEG < end ? lo + MK_FA_SEG : end;
  __shared__ uint4 seg_lds[MK_FA_WAVES][64];
  if (hi > lo) mk_fa_emit_seg(text, lo, hi, sum[seg].off, stream, bf.stream_off, lane, seg_lds[threadIdx.x >> 6]);
  /* this segment's share of the fill */
  const uint64_t kept = b.stat[f].kept, gap = bf.stream_cap - kept;
  const uint64_t j = seg - bf.seg0;
  const uint64_t g0 = gap * j / bf.nseg, g1 = gap * (j + 1u) / bf.nseg;
  uint8_t *p = stream + bf.stream_off + kept;
  for (uint64_t i = g0 + lane; i < g1; i += 64u) p[i] = (uint8_t)'\n';
}

/* ---- a candidate's file and its table ---------------------------------------------------------------------------------------
 * counted upsert into the table of the file that row `row` belongs to: linear probing from a multiplicative hash (the order of
 * the key list does not matter here: the layout is made from ordinals), same slot format and the same commutative updates as
 * mk_upsert_big */
__device__ __forceinline__ void mk_b_upsert(const mk_batch_dev &b, uint32_t row, uint64_t key, uint64_t ord) {
  const uint32_t f = mk_b_find(b.row0s, b.nfiles, row);
  const uint32_t mask = (1u << b.tb) - 1u;
  unsigned long long *kc = b.kc + ((size_t)f << b.tb), *ov = b.ordinv + ((size_t)f << b.tb);
  uint32_t n = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64u - b.tb));
  const unsigned long long fresh = ((unsigned long long)key << MK_CNT_BITS) | 1ull;
  for (uint32_t i = 0; i < MK_B_PROBES; i++) {
    unsigned long long cur = __hip_atomic_load(&kc[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool mine = false;
    if (cur == 0ull) {
      const unsigned long long prev = atomicCAS(&kc[n], 0ull, fresh);
      if (prev == 0ull) mine = true;
      else cur = prev;
    }
    if (mine || (cur >> MK_CNT_BITS) == key) {
      if (!mine && (cur & MK_CNT_MASK) < MK_CNT_SAT) atomicAdd(&kc[n], 1ull);
      atomicMax(&ov[n], ~(unsigned long long)ord);
      return;
    }
    n = (n + 1u) & mask;
  }
  atomicOr(&b.stat[f].flags, MK_BF_REDO);
}

/* ---- tables -> one key list ------------------------------------------------------------------------------------------------
 * as mk_compact_kernel: a wave owns 512 consecutive slots (inside one file's table: tb >= 9), a workgroup reserves its output
 * range with one atomicAdd; key 0 is never stored by the FASTA flavours (iseq2comem.c:300-302) */
__global__ void __launch_bounds__(1024) mk_b_compact_kernel(const mk_batch_dev b) {
  __shared__ uint32_t wtotal[16];
  __shared__ unsigned long long block_base;
  constexpr uint32_t CHUNK = 512u, ITER = CHUNK / 64u, WAVES = 16u;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t S = (uint64_t)b.nfiles << b.tb;
  const uint64_t nchunks = S / CHUNK, nblockchunks = (nchunks + WAVES - 1) / WAVES;
  for (uint64_t bc = blockIdx.x; bc < nblockchunks; bc += gridDim.x) {
    const uint64_t ci = bc * WAVES + wave;
    const uint64_t base_slot = ci < nchunks ? ci * CHUNK : S;
    unsigned long long kc[ITER];
#pragma unroll
    for (uint32_t it = 0; it < ITER; it++) {
      const uint64_t n = base_slot + it * 64u + lane;
      kc[it] = n < S ? b.kc[n] : 0ull;
    }
    uint32_t total = 0;
#pragma unroll
    for (uint32_t it = 0; it < ITER; it++) total += (uint32_t)__popcll(__ballot((kc[it] >> MK_CNT_BITS) != 0ull));
    const uint32_t f = (uint32_t)(base_slot >> b.tb);
    if (lane == 0) {
      wtotal[wave] = total;
      if (total) atomicAdd(&b.stat[f].D, total);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t sum = 0;
      for (uint32_t w = 0; w < WAVES; w++) sum += wtotal[w];
      block_base = sum ? atomicAdd(&b.misc[0], (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    unsigned long long base = block_base;
    for (uint32_t w = 0; w < wave; w++) base += wtotal[w];
    if (total) {
#pragma unroll
      for (uint32_t it = 0; it < ITER; it++) {
        const uint64_t n = base_slot + it * 64u + lane;
        const bool occ = (kc[it] >> MK_CNT_BITS) != 0ull;
        const uint64_t m = __ballot(occ);
        if (occ) {
          const uint64_t idx = base + mk_mbcnt(m);
          if (idx < b.list_cap) {
            b.key[idx] = kc[it] >> MK_CNT_BITS;
            b.ord[idx] = ~b.ordinv[n];
            const uint32_t c = (uint32_t)(kc[it] & MK_CNT_MASK);
            b.cnt[idx] = c > 65535u ? 65535u : c;
            b.gid[idx] = f;
          }
        }
        base += (unsigned long long)__popcll(m);
      }
    }
    __syncthreads();
  }
}

/* ---- reference slot order: priority insertion into a virtual table per file --------------------------------------------------
 * mk_layout_kernel's algorithm (a key takes a slot from an occupant with a LATER first ordinal, which restarts its own walk; the
 * fixed point is the table sequential first-come-first-served insertion builds, global_basic.h:282-284, iseq2comem.c:282-302),
 * with slot[n] of the hashsize-slot table replaced by the map entry of (file, n): entries are created on demand -- the first key
 * to probe virtual slot n claims a free entry along the linear probe sequence of hash(n) with ONE 64-bit CAS that also makes
 * it the occupant -- and never move or disappear; an occupant changes by CAS on the same word (slot << 32 | key index). */
__device__ __forceinline__ uint32_t mk_b_maphash(uint32_t n, uint32_t tb) { return (uint32_t)(((uint64_t)n * 0x9E3779B97F4A7C15ull) >> (64u - tb)); }

__global__ void __launch_bounds__(256) mk_b_layout_kernel(const mk_batch_dev b) {
  const uint64_t D = b.misc[0] < b.list_cap ? b.misc[0] : b.list_cap;
  const uint32_t mask = (1u << b.tb) - 1u, S = b.S;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t f = b.gid[i];
    {
      const mk_bstat st = b.stat[f];
      if (st.D > b.key_limit && !(st.flags & MK_BF_REDO)) atomicOr(&b.stat[f].flags, MK_BF_REDO); /* more keys than the batch admits per file */
      if (st.D > b.key_limit || (st.flags & MK_BF_REDO)) continue; /* sketched alone */
    }
    unsigned long long *map = b.map + ((size_t)f << b.tb);
    uint32_t cur = (uint32_t)i;
    unsigned long long ord = b.ord[cur];
    uint32_t n, h2;
    mk_probe_init(b.key[cur], S, n, h2);
    /* every step either moves a key one probe on or evicts a later key; 64 steps per map entry is far beyond any real walk */
    const uint64_t budget = 64ull * ((uint64_t)mask + 1ull);
    uint64_t steps = 0;
    for (;;) {
      if (++steps > budget) { atomicOr(&b.stat[f].flags, MK_BF_REDO); break; }
      /* the map entry of virtual slot n: found, or made (with this key in it) from the first free entry of its probe sequence */
      uint32_t e = mk_b_maphash(n, b.tb);
      unsigned long long v = 0;
      bool taken = false, found = false; /* found: v is the entry of virtual slot n (a legitimate entry may be 0: slot 0 held by key 0) */
      for (uint32_t pr = 0; pr <= mask; pr++) {
        v = __hip_atomic_load(&map[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == ~0ull) {
          const unsigned long long prev = atomicCAS(&map[e], ~0ull, ((unsigned long long)n << 32) | cur);
          if (prev == ~0ull) { taken = true; break; } /* the virtual slot was empty: this key has its place */
          v = prev;
        }
        if ((uint32_t)(v >> 32) == n) { found = true; break; }
        e = (e + 1u) & mask;
      }
      if (taken) break;
      if (!found) { atomicOr(&b.stat[f].flags, MK_BF_REDO); break; } /* (a full map: cannot happen below key_limit) */
      /* occupied: judge the occupant the word names; after a failed CAS the value IT returned (see mk_layout_kernel on stale
       * first looks: occupants only ever get earlier) */
      for (;;) {
        const uint32_t old = (uint32_t)v;
        if (b.ord[old] < ord) { n = mk_probe_next(n, h2, S); break; } /* an earlier key keeps the slot: next probe */
        const unsigned long long prev = atomicCAS(&map[e], v, ((unsigned long long)n << 32) | cur);
        if (prev != v) { v = prev; continue; } /* somebody else changed the slot: judge the real occupant */
        cur = old; /* evicted a later key: it walks its own sequence again */
        ord = b.ord[cur];
        mk_probe_init(b.key[cur], S, n, h2);
        break;
      }
    }
  }
}

/* ---- ordered dump: (file, component, slot) order by a bucket pass over the maps ------------------------------------------------
 * every map entry is one key in its final virtual slot.  Pass 1 counts per bucket and per file, pass 2 (after the scan) deals
 * (sort key, key index) to the buckets, the emit kernel ranks inside a bucket (about four keys) and writes the id. */
__device__ __forceinline__ bool mk_b_entry(const mk_batch_dev &b, uint64_t at, uint32_t &f, uint32_t &bucket, unsigned long long &sk, uint32_t &idx) {
  const unsigned long long v = b.map[at];
  if (v == ~0ull) return false;
  f = (uint32_t)(at >> b.tb);
  if (b.stat[f].flags & MK_BF_REDO) return false;
  idx = (uint32_t)v;
  if (b.cnt[idx] > b.cnt_hi) return false; /* uniq_fasta2co(): repeated keys are not written (iseq2comem.c:819-821) */
  const uint32_t n = (uint32_t)(v >> 32);
  const uint32_t comp = b.comp_num > 1u ? (uint32_t)(b.key[idx] % b.comp_num) : 0u;
  bucket = f * b.bpf + comp * b.bpc + (n >> b.shift);
  sk = ((unsigned long long)(f * b.comp_num + comp) << 32) | n;
  return true;
}

__global__ void __launch_bounds__(256) mk_b_bucket_kernel(const mk_batch_dev b) {
  const uint64_t total = (uint64_t)b.nfiles << b.tb;
  for (uint64_t at = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; at < total; at += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t f = 0, bucket = 0, idx = 0;
    unsigned long long sk = 0;
    const bool on = mk_b_entry(b, at, f, bucket, sk, idx);
    if (on) atomicAdd(&b.bstart[bucket], 1u);
    /* a wave's 64 entries lie in one file's map (tb >= 9): one add per wave for the file's total */
    const uint64_t m = __ballot(on);
    if (m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(&b.nout[(size_t)f * 16u], (uint32_t)__popcll(m));
  }
}

/* one workgroup per file: exclusive scan of its buckets' counts, offset by the ids of the files in front of it; per-component
 * totals; the last file closes the array with the batch's total */
__global__ void __launch_bounds__(1024) mk_b_bscan_kernel(const mk_batch_dev b) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s, base_s;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, f = blockIdx.x;
  /* ids of the files in front of this one */
  uint32_t part = 0;
  for (uint32_t g = threadIdx.x; g < f; g += 1024u) part += b.nout[(size_t)g * 16u];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
    for (uint32_t w = 0; w < 16u; w++) s += wsum[w];
    base_s = s;
    carry_s = 0u;
  }
  __syncthreads();
  const uint32_t base = base_s;
  uint32_t *bc = b.bstart + (size_t)f * b.bpf;
  for (uint32_t r0 = 0; r0 < b.bpf; r0 += 1024u) {
    const uint32_t i = r0 + threadIdx.x;
    const uint32_t v = i < b.bpf ? bc[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if ((int)lane >= o) incl += t;
    }
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
    const uint32_t carry = carry_s;
    if (i < b.bpf) { bc[i] = base + carry + woff + incl - v; b.bcursor[(size_t)f * b.bpf + i] = 0u; }
    __syncthreads();
    if (threadIdx.x == 1023u) carry_s = carry + woff + incl;
    __syncthreads();
  }
  /* component c of this file owns the buckets [c * bpc, (c + 1) * bpc): its size is the difference of two starts */
  if (threadIdx.x < b.comp_num) {
    const uint32_t c = threadIdx.x;
    const uint32_t lo = bc[c * b.bpc] - base;
    const uint32_t hi = c + 1u < b.comp_num ? bc[(c + 1u) * b.bpc] - base : carry_s;
    b.ctot[(size_t)f * b.comp_num + c] = hi - lo;
  }
  if (f + 1u == b.nfiles && threadIdx.x == 0) {
    b.bstart[(size_t)b.nfiles * b.bpf] = base + carry_s;
    b.misc[1] = (unsigned long long)(base + carry_s);
  }
}

__global__ void __launch_bounds__(256) mk_b_scatter_kernel(const mk_batch_dev b) {
  const uint64_t total = (uint64_t)b.nfiles << b.tb;
  for (uint64_t at = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; at < total; at += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t f = 0, bucket = 0, idx = 0;
    unsigned long long sk = 0;
    if (!mk_b_entry(b, at, f, bucket, sk, idx)) continue;
    const uint32_t pos = b.bstart[bucket] + atomicAdd(&b.bcursor[bucket], 1u);
    if (pos < b.out_cap) { b.tkey[pos] = sk; b.tidx[pos] = idx; }
  }
}

__global__ void __launch_bounds__(256) mk_b_emit_kernel(const mk_batch_dev b) {
  const uint32_t n_out = b.bstart[(size_t)b.nfiles * b.bpf];
  if ((uint64_t)n_out > b.out_cap) return; /* the host sees misc[1] > capacity and runs the batch's dump again with larger arrays */
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long sk = b.tkey[j];
    const uint32_t fc = (uint32_t)(sk >> 32), f = fc / b.comp_num, comp = fc - f * b.comp_num;
    const uint32_t bucket = f * b.bpf + comp * b.bpc + ((uint32_t)sk >> b.shift);
    const uint32_t lo = b.bstart[bucket], hi = b.bstart[bucket + 1u];
    uint32_t rank = 0;
    for (uint32_t t = lo; t < hi; t++) rank += b.tkey[t] < sk ? 1u : 0u; /* slots are distinct: no ties */
    b.out_ids[lo + rank] = (uint32_t)(b.key[b.tidx[j]] >> b.comp_code_bits);
  }
}

/* the batch's results to the host's pinned blocks, by the device itself: the per-file block (flags, component sizes, totals) and
 * the ids there are -- min(n_out, ids_cap) of them, 16 bytes a lane (the lists end in slack; ids_cap is a multiple of four) */
/* The last 16 bytes of the status block are not the batch's: they take the ENGINE's error flags home (mk_table::err, set by a scan
 * kernel that bailed out -- a filter that is not where the tuned loop addresses it --, which mk_sketch_batch_end would otherwise
 * never see: it reads this block only). */
__global__ void __launch_bounds__(256) mk_b_home_kernel(const mk_batch_dev b, uint4 *h_stat, uint32_t stat16, uint4 *h_ids, unsigned long long ids_cap,
                                                        const uint32_t *engine_err) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
  const uint4 *st = (const uint4 *)b.stat;
  for (uint64_t i = tid; i < stat16; i += nthreads) {
    uint4 v = st[i];
    if (i == stat16 - 1u) v = make_uint4(engine_err[0], 0u, 0u, 0u);
    h_stat[i] = v;
  }
  unsigned long long n = b.misc[1];
  if (n > b.out_cap) return;
  if (n > ids_cap) n = ids_cap;
  const uint4 *ids = (const uint4 *)b.out_ids;
  for (uint64_t i = tid; i < (n + 3u) / 4u; i += nthreads) h_ids[i] = ids[i];
}
