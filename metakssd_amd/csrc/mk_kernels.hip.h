/*
 * mk_kernels.hip.h -- device code of the MI355X sketch engine (gfx950, wave64).
 *
 * Kernels (DESIGN.md has the data layout and the roofline of each):
 *   mk_scan_kernel      rows of ASCII bases -> rolling forward k-mer -> strand-symmetric LDS filter of the
 *                       accepted inner-substring subspace -> candidates appended to per-wave HBM buffers.
 *   mk_resolve_kernel   candidates -> canonical k-mer -> exact .shuf check -> key -> counted upsert.
 *                       Together they restate the per-read loop of mt_shortreads2koc(), iseq2comem.c:676-720
 *                       (and of fasta2co(), :248-311, on overlapped windows).
 *   mk_import_kernel    fold another shard's {key,count,first ordinal} list into the table (multi-GPU).
 *   mk_compact_kernel   table -> dense list of distinct keys.
 *   mk_layout_kernel    priority insertion: rebuilds the slot layout the reference's SEQUENTIAL
 *                       first-come-first-served double hashing (global_basic.h:282-284,
 *                       iseq2comem.c:701-718) would have produced, from first-occurrence ordinals.
 *   mk_dump_*           slot-order compaction = write_fqkoc2files(), iseq2comem.c:539-553 /
 *                       wrt_co2cmpn_use_inn_subctx(), :638-646.
 *   mk_synth_kernel     synthetic reads (same bytes as mk_synth_rows_host).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "host/mk_host_internal.h"

#define MK_WAVE 64
#define MK_MAX_CB 128                  /* widest column block staged per step, bytes */
#define MK_MAX_PIECES 8                /* 16-byte pieces per lane per step = MK_MAX_CB/16 */
#define MK_EMPTY32 0xFFFFFFFFu
#define MK_ZMASK_WORDS 256u            /* tuned scan kernels: filter masks by the low 8 bits of the pair key, 1 KiB of LDS */
#define MK_CNT_BITS 24                 /* slot word = key << 24 | count (the reference's slot is key << 16 | count16) */
#define MK_CNT_MASK 0xFFFFFFull
#define MK_CNT_SAT 0xF00000ull         /* stop adding long before the count field could carry into the key */

struct mk_keyparams {
  uint64_t tupmask, domask, undomask, lowmask;
  uint32_t TL, crvsaddmove, out2 /*2*half_outctx_len*/, key_lshift /*2*TL-4*out*/, dr4 /*4*drlevel*/;
  int32_t dim_start, dim_end;
  uint32_t S; /* hashsize */
};

struct mk_table {
  unsigned long long *kc;     /* key << 24 | occurrences (>= 1), 0 = empty   [S]  (keys are < 2^39) */
  unsigned long long *ordinv; /* ~(first ordinal), max-combined              [S] */
  uint32_t *err;              /* [0] = table-full flag */
  /* Large tables (S >= 2^26, e.g. L2K11's 537 M slots for a genome's few thousand keys): one bit per block of
   * 1 << dirty_shift slots, set by whoever installs a key, so that clearing, compaction and the dump touch only the
   * blocks that were used.  NULL = dense bookkeeping (every pass walks all S slots). */
  uint32_t *dirty;
  uint32_t dirty_shift;
  /* Front table (dense bookkeeping, tables of 2^20 slots and more): a small accumulation table of the same slot format in
   * front of the S-slot one.  The S slots exist because the reference's hashlimit admits 0.6 S distinct keys; a sketch of
   * 50 M reads leaves 1.6 M in 33.5 M slots, and every pass over the table (clear, compaction) pays for S.  The resolve and
   * import kernels put keys into the front table while it is open (fewer than `limit` keys at the end of the last launch) and
   * a key finds a place within MK_FRONT_PROBES probes; everything else -- a closed or crowded front table, the scan kernel's
   * own overflow path -- goes to the big table and sets state[1].  The two tables are independent partial accumulators
   * (a key may sit in both): when state[1] is set, the compaction first folds the front table into the big one
   * (mk_front_fold_kernel: counts add, first ordinals combine by min, exactly the multi-GPU merge) and lists the big table;
   * otherwise it lists the front table and the big one is neither read nor, at the next begin, cleared. */
  const struct mk_front *fr; /* NULL = no front table.  Behind a pointer: the table descriptor travels in every kernel's
                              * argument block, and 40 more bytes of it cost the scan kernel 56 more spilled SGPRs */
};
struct mk_front {
  unsigned long long *kc1, *ordinv1;
  uint32_t *state;      /* [0] keys installed in the front table, [1] != 0: the big table holds keys, [2] != 0: closed, [3] launch ticket */
  uint32_t shift;       /* home slot in the big table >> shift = first front slot: ceil(log2 S) - log2(front slots) */
  uint32_t mask;        /* front slots - 1 */
  uint32_t limit;
  uint32_t reserved;
};
#define MK_FRONT_PROBES 16u

struct mk_batch_dev; /* mk_batch.hip.h: many small inputs in one launch sequence, a table per file */
__device__ __forceinline__ void mk_b_upsert(const mk_batch_dev &b, uint32_t row, uint64_t key, uint64_t ord);

struct mk_scan_args {
  const uint8_t *rows;
  uint64_t nreads, first_ord;
  uint32_t stride;    /* bytes staged per row (the row's width) */
  uint32_t pitch;     /* address step from one row to the next: == stride for rows that lie side by side, smaller for the
                       * overlapping virtual rows of a base stream (mk_sketch_push_stream) */
  uint32_t rowlen;    /* 0: a row ends at its '\n' or at `stride`; else: bytes from this index on are not part of the row
                       * (virtual rows: the next row's k-mers) -- the staged tile gets a '\n' there */
  const unsigned long long *nreads_dev; /* NULL, or the row count in device memory (nreads is then an upper bound) */
  uint32_t CB, ncb;   /* column block width (bytes, multiple of 4), blocks per row */
  uint32_t ppr;       /* pieces per row per block: CB/16 (vec path) or CB/4 (dword path) */
  uint32_t ppr_inv;   /* floor(2^20/ppr)+1 : q/ppr == (q*ppr_inv)>>20 for q < 2^20/ppr */
  uint32_t rowdw;     /* LDS dwords per staged row (odd => conflict-free row reads) */
  uint32_t wave_lds_dwords;
  uint32_t bm_words;  /* LDS filter words (power of two) */
  uint32_t mt_words;  /* tuned kernels with subk 6: the 256-entry mask table in front of the filter (LDS offset 0), else 0 */
  uint32_t pair_subk; /* 0: every candidate record is a single k-mer (generic kernel); 6 / 5: pair records of the tuned kernels for that
                       * half_subctx_len -- the resolve kernel then holds the matching exact-side filter (mk_build_filter / mk_build_xfilter) */
  uint32_t dimmask;   /* 2^(4*subk)-1: the inner substring after uni >> out2 */
  const uint32_t *accept; /* inner substrings d with dim_start <= shuf[d] < dim_end */
  uint32_t n_accept;
  const int32_t *shuf;
  const uint32_t *accept_bits; /* bit d set <=> dim_start <= shuf[d] < dim_end (2 MiB at subk 6: stays in L2) */
  mk_keyparams kp;
  mk_table tab;
  /* filter hits ("candidates") leave the scan kernel through per-wave append buffers in HBM */
  /* [nslots][cand_cap] 16-byte candidate records:
   *   single k-mer (slow paths, generic kernel): {fwd lo, fwd hi, ord lo, ord hi | 0x80000000}
   *   pair (tuned loop: some base pair of this lane's 8-base window passed the LDS pair filter):
   *       {flo at the window start, packed codes << 16 | e << 12 | jmin << 9 | pos0 >> 3, h2 & 0xFFFF | h3 << 16 (h2, h3 = flo at the two
   *        previous window starts), row index}
   * The resolve kernel holds the same LDS filter, finds the base(s) that passed and rebuilds their k-mers, so the scan
   * kernel's hit path is one ballot and one 16-byte store. */
  uint4 *cand;
  uint32_t *cand_count;         /* [nslots] */
  uint32_t cand_cap;
  /* not NULL: the rows are those of a batch of files (first_ord == 0) and an accepted k-mer goes to the table of the file its row
   * belongs to (mk_b_upsert) instead of a.tab; read by the out-of-line resolve paths only */
  const mk_batch_dev *batch;
};

/* ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ uint32_t mk_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mk_mbcnt(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ void mk_wave_lds_fence() {
  /* same-wave LDS producer -> consumer: DS operations of one wave execute in order; this only stops
   * the compiler from moving LDS accesses across the hand-off */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* probe step of HASH(K,I,S) = (K%S + I*(1+K%(S-1))) % S, incrementally: n_{i+1} = (n_i + h2) mod S */
__device__ __forceinline__ void mk_probe_init(uint64_t key, uint32_t S, uint32_t &n, uint32_t &h2) {
  n = (uint32_t)(key % S);
  h2 = 1u + (uint32_t)(key % (uint64_t)(S - 1));
}
__device__ __forceinline__ uint32_t mk_probe_next(uint32_t n, uint32_t h2, uint32_t S) {
  uint64_t t = (uint64_t)n + h2;
  return (uint32_t)(t >= S ? t - S : t);
}

/* key reduction: iseq2comem.c:696-699 */
__device__ __forceinline__ uint64_t mk_reduce_key(const mk_keyparams &kp, uint64_t uni, uint64_t pf) {
  return (((uni & kp.undomask) + ((uni & kp.lowmask) << kp.key_lshift)) >> kp.dr4) + pf;
}

/* counted upsert into the accumulation table (arrival order is irrelevant: counts add, first ordinals
 * combine by min; the reference-order layout is rebuilt afterwards by mk_layout_kernel).
 * New key: one CAS (installs key with count `add`) + one atomicMax; known key: one atomicAdd + one atomicMax. */
__device__ __forceinline__ void mk_upsert_big(const mk_table &tab, uint32_t S, uint64_t key, uint64_t ord, uint32_t add) {
  if (tab.fr) { /* the compaction has to look at the big table */
    uint32_t *st = tab.fr->state;
    if (__hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(&st[1], 1u);
  }
  uint32_t n, h2;
  mk_probe_init(key, S, n, h2);
  const unsigned long long fresh = ((unsigned long long)key << MK_CNT_BITS) | add;
  for (uint32_t i = 0; i < S; i++) {
    /* load, then CAS only on an empty slot (CAS-first measured the same on all-distinct input and costs a failed
     * atomic per occurrence on repeat-heavy input) */
    unsigned long long cur = __hip_atomic_load(&tab.kc[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool mine = false;
    if (cur == 0ull) {
      const unsigned long long prev = atomicCAS(&tab.kc[n], 0ull, fresh);
      if (prev == 0ull) mine = true; /* installed with its first count */
      else cur = prev;
    }
    if (mine || (cur >> MK_CNT_BITS) == key) {
      if (!mine && (cur & MK_CNT_MASK) < MK_CNT_SAT) atomicAdd(&tab.kc[n], (unsigned long long)add);
      atomicMax(&tab.ordinv[n], ~(unsigned long long)ord);
      if (mine && tab.dirty) { /* the only place a slot ever becomes non-zero */
        const uint32_t b = n >> tab.dirty_shift;
        atomicOr(&tab.dirty[b >> 5], 1u << (b & 31u));
      }
      return;
    }
    n = mk_probe_next(n, h2, S);
  }
  atomicOr(&tab.err[0], 1u); /* every slot taken by other keys */
}
/* the same through the front table when it is open (front_open: state[2] == 0 at the start of the launch).  The front slot
 * of a key is its home slot in the big table scaled down ((key % S) >> shift), so that a pass over the front table meets
 * the keys in the order of their home slots, which is the order the layout kernel and the dump walk the layout table in.
 * Returns true when the key was installed in the front table (the caller counts those into state[0]). */
__device__ __forceinline__ bool mk_upsert(const mk_table &tab, uint32_t S, uint64_t key, uint64_t ord, uint32_t add, bool front_open) {
  if (front_open) {
    const mk_front f = *tab.fr;
    uint32_t n = (uint32_t)(key % S) >> f.shift;
    const uint32_t step = (uint32_t)(key * 0x9E3779B1u) | 1u; /* odd: the sequence visits every slot of the power-of-two table */
    const unsigned long long fresh = ((unsigned long long)key << MK_CNT_BITS) | add;
    for (uint32_t i = 0; i < MK_FRONT_PROBES; i++) {
      unsigned long long cur = __hip_atomic_load(&f.kc1[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool mine = false;
      if (cur == 0ull) {
        const unsigned long long prev = atomicCAS(&f.kc1[n], 0ull, fresh);
        if (prev == 0ull) mine = true;
        else cur = prev;
      }
      if (mine || (cur >> MK_CNT_BITS) == key) {
        if (!mine && (cur & MK_CNT_MASK) < MK_CNT_SAT) atomicAdd(&f.kc1[n], (unsigned long long)add);
        atomicMax(&f.ordinv1[n], ~(unsigned long long)ord);
        return mine;
      }
      n = (n + step) & f.mask;
    }
  }
  mk_upsert_big(tab, S, key, ord, add);
  return false;
}
/* end of a launch that inserts: the workgroup's count of front-table installs goes to front[0]; the last workgroup to get
 * here decides whether the front table is still open for the NEXT launch.  Called by one thread per workgroup. */
__device__ __forceinline__ void mk_front_launch_end(const mk_table &tab, uint32_t installed, uint32_t nworkgroups) {
  if (!tab.fr) return;
  uint32_t *st = tab.fr->state;
  if (installed) atomicAdd(&st[0], installed);
  __threadfence();
  if (atomicAdd(&st[3], 1u) == nworkgroups - 1u) {
    __threadfence();
    const uint32_t total = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[2], total > tab.fr->limit ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ bool mk_front_open(const mk_table &tab) {
  return tab.fr && __hip_atomic_load(&tab.fr->state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

/* reverse complement of a k-mer of `TL` bases held in the low 2*TL bits: reverse the 2-bit groups of the
 * complement.  Equals the reference's incrementally built crvstuple (iseq2comem.c:686). */
__device__ __forceinline__ uint64_t mk_revcomp(uint64_t f, uint32_t TL) {
  uint64_t n = ~f;
  n = ((n >> 2) & 0x3333333333333333ull) | ((n & 0x3333333333333333ull) << 2);
  n = ((n >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((n & 0x0F0F0F0F0F0F0F0Full) << 4);
  n = __builtin_bswap64(n);
  return n >> (64u - 2u * TL);
}

/* The scan kernel's 2-bit code of a base is (byte >> 1) & 3: A=0 C=1 T=2 G=3 -- two VALU ops per four bases
 * cheaper than the reference's A=0 C=1 G=2 T=3 (Basemap, global_basic.c:62-69).  The two codings differ by the
 * Gray map c ^ (c >> 1), which is its own inverse on 2 bits; the LDS filter is built over re-coded substrings
 * and every candidate k-mer is re-coded once here, before anything reference-defined is computed from it. */
__device__ __forceinline__ uint64_t mk_scan_to_ref_codes(uint64_t k) { return k ^ ((k >> 1) & 0x5555555555555555ull); }

/* One candidate = the FORWARD k-mer whose inner substring passed the strand-symmetric LDS filter.  Form the
 * canonical k-mer (iseq2comem.c:691), look it up in the .shuf table (:692-695), reduce to the key (:696-699)
 * and upsert. */
__device__ __forceinline__ void mk_resolve_one(const mk_scan_args &a, uint64_t scan_fwd, uint64_t ord) {
  const uint64_t fwd = mk_scan_to_ref_codes(scan_fwd);
  const uint64_t rc = mk_revcomp(fwd, a.kp.TL);
  const uint64_t uni = fwd < rc ? fwd : rc;
  const uint32_t dim = (uint32_t)((uni & a.kp.domask) >> a.kp.out2);
  /* most candidates are filter false positives or wrong-strand mirrors: settle those on the small bitmap and
   * touch the 4*16^subk-byte .shuf table only for the accepted ones */
  if (!((a.accept_bits[dim >> 5] >> (dim & 31u)) & 1u)) return;
  const int32_t pf = a.shuf[dim];
  if (pf >= a.kp.dim_start && pf < a.kp.dim_end) {
    const uint64_t key = mk_reduce_key(a.kp, uni, (uint64_t)(pf - a.kp.dim_start));
    if (a.batch) mk_b_upsert(*a.batch, (uint32_t)(ord >> 12), key, ord);
    else mk_upsert_big(a.tab, a.kp.S, key, ord, 1u); /* (overflow path of the scan kernel only: straight to the big table) */
  }
}

typedef __attribute__((address_space(3))) const uint32_t *mk_lds_cu32;
__device__ __forceinline__ uint32_t mk_filter_mask(uint32_t x);

/* expands one candidate record; `filter` = the LDS filter (same contents in the scan and the resolve kernel) */
__device__ __forceinline__ void mk_resolve_record(const mk_scan_args &a, const uint4 r, const uint32_t *filter) {
  if (r.w & 0x80000000u) { /* single k-mer */
    mk_resolve_one(a, ((uint64_t)r.y << 32) | r.x, ((uint64_t)(r.w & 0x7FFFFFFFu) << 32) | r.z);
    return;
  }
  const uint32_t SH = a.kp.out2 - 2u; /* the inner substring of the k-mer ending at base j: bits SH.. of the low word in front of base j */
  const uint32_t hm = (uint32_t)((1ull << (2u * a.kp.TL - 32u)) - 1ull); /* mk_kmer_hi<K>::HMASK */
  const uint32_t pos0 = (r.y & 0x1FFu) << 3, jmin = (r.y >> 9) & 7u, e = (r.y >> 12) & 15u;
  const uint32_t lo = r.y & 0xFFFF0000u, h2 = r.z & 0xFFFFu, h3 = r.z >> 16;
  const uint64_t ord0 = ((a.first_ord + (uint64_t)r.w) << 12) | pos0;
  /* first every base of the pair against the LDS filter (no global memory), then only the bases that passed -- one per
   * round, so that the dependent global accesses of mk_resolve_one are not repeated for the seven bases that did not */
  uint32_t before = r.x, hits = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) {
    const uint32_t x = (before >> SH) & a.dimmask;
    if (j >= jmin && j < e && (!filter || (mk_filter_mask(x) & ~filter[(x >> 10) & (a.bm_words - 1u)]) == 0u)) hits |= 1u << j;
    before = __builtin_amdgcn_alignbit(r.x, lo, 30u - 2u * j);
  }
  while (hits) {
    const uint32_t j = (uint32_t)__builtin_ctz(hits);
    hits &= hits - 1u;
    const uint32_t dd = 2u * (j + 1u);
    const uint32_t fl = __builtin_amdgcn_alignbit(r.x, lo, 30u - 2u * j);
    const uint32_t fhi = ((h3 << dd) | (h2 >> (16u - dd))) & hm;
    mk_resolve_one(a, ((uint64_t)fhi << 32) | fl, ord0 + j);
  }
}

/* overflow path of the scan kernel (a wave's append buffer is full: dense accept-everything tables): resolve
 * this lane's record on the spot.  Out of line; reads the argument block through the kernarg pointer so that
 * the call does not push the hot loop's parameters into scratch.  filter == NULL (tuned kernels, whose LDS holds the pair
 * filter): every base of the window goes to the exact test. */
__device__ __noinline__ void mk_resolve_inline(const mk_scan_args *ka, bool hit, uint4 r, const uint32_t *filter) {
  if (hit) mk_resolve_record(*ka, r, filter);
}

__device__ __forceinline__ void mk_build_filter(uint32_t *bitmap, const mk_scan_args &a);
__device__ __forceinline__ void mk_build_xfilter(uint32_t *bitmap, const mk_scan_args &a);

/* ---- resolve kernel ----------------------------------------------------------------------------------------------------
 * Resolves the candidates the scan kernel appended.  A workgroup builds the exact per-base LDS filter once and walks
 * producer slots; per record the work is LDS-only (which of the window's 8 bases pass the filter).  What follows a passing
 * base is a chain of dependent global accesses (accept bitmap in L2 -> .shuf entry -> table slot load -> CAS -> atomicMax),
 * and only about one record in seven has such a base: run lane by lane behind the filter test, every round of that chain
 * would serve a handful of lanes and the kernel is bound by the chain's latency (0.39 ms for 22.8 M records).  So the
 * stages are decoupled through two small per-wave rings in LDS:
 *     records --filter test--> ring 1 {forward k-mer, ordinal} --canonical k-mer + accept bit--> ring 2 {canonical k-mer,
 *     ordinal} --.shuf entry, key, upsert--> table
 * A ring is drained 64 entries at a time, one per lane, so every round of either chain runs with all lanes busy.  Same-wave
 * producer and consumer: no workgroup barrier (DS operations of a wave execute in order).  The table update commutes
 * (mk_upsert), so the order in which candidates arrive is free. */
#define MK_RESOLVE_THREADS 1024
#define MK_RQ_CAP 128u /* ring entries per wave: a push adds at most 64 to fewer than 64 */
#ifndef MK_RESOLVE_UNROLL
#define MK_RESOLVE_UNROLL 4u /* blocks of 64 records a wave takes per round */
#endif


struct mk_wring {
  uint4 *q;
  uint32_t head, tail; /* wave-uniform */
};
__device__ __forceinline__ void mk_wring_push(mk_wring &w, bool has, const uint4 v) {
  const uint64_t b = __ballot(has);
  if (has) w.q[(w.tail + mk_mbcnt(b)) & (MK_RQ_CAP - 1u)] = v;
  w.tail += (uint32_t)__popcll(b);
}
/* takes up to 64 entries: lane l gets entry head + l; returns whether this lane got one */
__device__ __forceinline__ bool mk_wring_pop(mk_wring &w, uint32_t lane, uint4 &v) {
  const uint32_t have = w.tail - w.head, n = have < 64u ? have : 64u;
  v = w.q[(w.head + lane) & (MK_RQ_CAP - 1u)];
  w.head += n;
  return lane < n;
}

/* ring 2 -> table: .shuf entry (iseq2comem.c:692-695), key (:696-699), upsert */
/* Out of line, its operands (key geometry, table and front-table descriptors, .shuf pointer) read through the kernarg
 * pointer: it runs once per 64 accepted candidates, and inlined at every drain site it kept some twenty SGPRs live across
 * the per-record loop, which then spilled over two hundred of them. */
__device__ __noinline__ bool mk_resolve_accepted(const mk_scan_args *ka, bool on, const uint4 v, bool front_open) {
  if (!on) return false;
  const mk_scan_args &a = *ka;
  const uint64_t uni = ((uint64_t)v.y << 32) | v.x;
  const uint32_t dim = (uint32_t)((uni & a.kp.domask) >> a.kp.out2);
  const int32_t pf = a.shuf[dim];
  if (pf >= a.kp.dim_start && pf < a.kp.dim_end) {
    const uint64_t key = mk_reduce_key(a.kp, uni, (uint64_t)(pf - a.kp.dim_start)), ord = ((uint64_t)v.w << 32) | v.z;
    if (a.batch) { mk_b_upsert(*a.batch, (uint32_t)(ord >> 12), key, ord); return false; }
    return mk_upsert(a.tab, a.kp.S, key, ord, 1u, front_open);
  }
  return false;
}
/* ring 1 -> ring 2: canonical k-mer (iseq2comem.c:691) and the accept bit of its inner substring */
__device__ __forceinline__ void mk_resolve_candidate(const mk_scan_args &a, bool on, const uint4 v, mk_wring &r2) {
  bool pass = false;
  uint4 o = v;
  if (on) {
    const uint64_t fwd = mk_scan_to_ref_codes(((uint64_t)v.y << 32) | v.x);
    const uint64_t rc = mk_revcomp(fwd, a.kp.TL);
    const uint64_t uni = fwd < rc ? fwd : rc;
    const uint32_t dim = (uint32_t)((uni & a.kp.domask) >> a.kp.out2);
    pass = (a.accept_bits[dim >> 5] >> (dim & 31u)) & 1u;
    o.x = (uint32_t)uni; o.y = (uint32_t)(uni >> 32);
  }
  mk_wring_push(r2, pass, o);
}

__global__ void __launch_bounds__(MK_RESOLVE_THREADS) mk_resolve_kernel(const mk_scan_args a, uint32_t nslots) {
  extern __shared__ __align__(16) uint32_t rlds[];
  __shared__ uint32_t wg_installed;
  const mk_scan_args *ka = (const mk_scan_args *)__builtin_amdgcn_kernarg_segment_ptr();
  if (threadIdx.x == 0) wg_installed = 0u;
  const bool front_open = mk_front_open(a.tab);
  uint32_t installed = 0; /* wave-uniform: keys this wave put into the front table */
  if (a.pair_subk == 5u) mk_build_xfilter(rlds, a);
  else mk_build_filter(rlds, a);
  const uint32_t lane = mk_lane(), wave = threadIdx.x >> 6;
  uint4 *rings = (uint4 *)(rlds + a.bm_words) + (size_t)wave * 2u * MK_RQ_CAP;
  mk_wring r1{rings, 0u, 0u}, r2{rings + MK_RQ_CAP, 0u, 0u};

  auto drain2 = [&](uint32_t least) {
    while (r2.tail - r2.head >= least) {
      uint4 v;
      const bool on = mk_wring_pop(r2, lane, v);
      mk_wave_lds_fence();
      installed += (uint32_t)__popcll(__ballot(mk_resolve_accepted(ka, on, v, front_open)));
    }
  };
  auto drain1 = [&](uint32_t least) {
    while (r1.tail - r1.head >= least) {
      uint4 v;
      const bool on = mk_wring_pop(r1, lane, v);
      mk_wave_lds_fence();
      mk_resolve_candidate(a, on, v, r2);
      mk_wave_lds_fence();
      drain2(64u);
    }
  };

  /* pair records only come from the tuned kernels (k = 9, 10, 11 with subk 6; k = 11 with subk 5); for any other geometry every record
   * is a single k-mer */
  const uint32_t SH = a.pair_subk ? a.kp.out2 - 2u : 0u;
  const bool x5 = a.pair_subk == 5u;
  const uint32_t hm = a.kp.TL > 16u ? (uint32_t)((1ull << (2u * a.kp.TL - 32u)) - 1ull) : 0u; /* mk_kmer_hi<K>::HMASK */
  const uint32_t wmask = a.bm_words - 1u;

  /* a wave owns whole slots (slot = global wave index + multiples of the number of waves: one slot each when the grid
   * matches the scan's) and takes MK_RESOLVE_UNROLL blocks of 64 records per round; the loads of the next round are issued
   * before the current one is worked on.  (One block per round with the loop body once: 0.224 instead of 0.195 ms -- the
   * slot bookkeeping of `fetch` is then paid per 64 records.) */
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wave, base = 0u, n = slot < nslots ? a.cand_count[slot] : 0u;
  auto fetch = [&](uint4 (&r)[MK_RESOLVE_UNROLL]) -> bool { /* false: nothing left */
    while (slot < nslots && base >= n) {
      slot += nwaves;
      base = 0u;
      n = slot < nslots ? a.cand_count[slot] : 0u;
    }
    if (slot >= nslots) return false;
    const uint4 *c = a.cand + (size_t)slot * a.cand_cap;
#pragma unroll
    for (uint32_t u = 0; u < MK_RESOLVE_UNROLL; u++) {
      const uint32_t i = base + u * 64u + lane;
      r[u] = i < n ? c[i] : make_uint4(0u, 0u, 0u, 0u); /* all-zero pair record: e == 0, no base tested */
    }
    base += 64u * MK_RESOLVE_UNROLL;
    return true;
  };

  auto process = [&](const uint4 r) {
    const bool single = (r.w & 0x80000000u) != 0u;
    if (__any(single)) { /* slow tiers and the generic kernel hand over whole k-mers */
      mk_wring_push(r1, single, make_uint4(r.x, r.y, r.z, r.w & 0x7FFFFFFFu));
      mk_wave_lds_fence();
      drain1(64u);
    }
    /* pair record: which of the window's bases pass the exact filter?  The mask of base j is the OR of three one-hot words,
     * of the low 5 bits of the substrings 0, 1 and 3 bases earlier (mk_filter_mask: fields at offsets 0, 2, 6). */
    const uint32_t jmin = (r.y >> 9) & 7u, e = single ? 0u : (r.y >> 12) & 15u;
    const uint32_t lo = r.y & 0xFFFF0000u;
    uint32_t hits = 0;
    if (x5) { /* subk 5: the 2^20 substrings in 2^19 bits (mk_build_xfilter), the scan kernel's own test */
      uint32_t before = r.x;
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        const uint32_t x = (before >> SH) & 0xFFFFFu;
        if ((rlds[x >> 6] >> (x & 31u)) & 1u) hits |= 1u << j;
        before = __builtin_amdgcn_alignbit(r.x, lo, 30u - 2u * j);
      }
      hits &= (0xFFu << jmin) & ~(0xFFFFFFFFu << e);
    } else {
      uint32_t oh3 = 1u << ((r.x >> (SH + 6u)) & 31u), oh2 = 1u << ((r.x >> (SH + 4u)) & 31u), oh1 = 1u << ((r.x >> (SH + 2u)) & 31u);
      uint32_t before = r.x;
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        const uint32_t oh0 = 1u << ((before >> SH) & 31u);
        const uint32_t word = rlds[(before >> (SH + 10u)) & wmask];
        if (((oh0 | oh1 | oh3) & ~word) == 0u) hits |= 1u << j;
        oh3 = oh2; oh2 = oh1; oh1 = oh0;
        before = __builtin_amdgcn_alignbit(r.x, lo, 30u - 2u * j);
      }
      hits &= (0xFFu << jmin) & ~(0xFFFFFFFFu << e);
    }
    if (__any(hits != 0u)) {
      const uint32_t pos0 = (r.y & 0x1FFu) << 3, h2 = r.z & 0xFFFFu, h3 = r.z >> 16;
      const uint64_t ord0 = ((a.first_ord + (uint64_t)r.w) << 12) | pos0;
      do {
        const bool has = hits != 0u;
        const uint32_t j = has ? (uint32_t)__builtin_ctz(hits) : 0u;
        hits &= hits - 1u;
        const uint32_t dd = 2u * (j + 1u);
        const uint32_t fl = __builtin_amdgcn_alignbit(r.x, lo, 30u - 2u * j);
        const uint32_t fhi = ((h3 << dd) | (h2 >> (16u - dd))) & hm;
        const uint64_t ord = ord0 + j;
        mk_wring_push(r1, has, make_uint4(fl, fhi, (uint32_t)ord, (uint32_t)(ord >> 32)));
        mk_wave_lds_fence();
        drain1(64u);
      } while (__any(hits != 0u));
    }
  };

  uint4 r[MK_RESOLVE_UNROLL], rn[MK_RESOLVE_UNROLL];
  bool more = fetch(r);
  while (more) {
    more = fetch(rn);
#pragma unroll
    for (uint32_t u = 0; u < MK_RESOLVE_UNROLL; u++) process(r[u]);
#pragma unroll
    for (uint32_t u = 0; u < MK_RESOLVE_UNROLL; u++) r[u] = rn[u];
  }
  mk_wave_lds_fence();
  drain1(1u);
  drain2(1u);
  if (a.tab.fr) {
    if (lane == 0 && installed) atomicAdd(&wg_installed, installed);
    __syncthreads();
    if (threadIdx.x == 0) mk_front_launch_end(a.tab, wg_installed, gridDim.x);
  }
}

/* LDS filter: a blocked Bloom filter over B = A u revcomp(A), A = the accepted inner substrings
 * (dim_start <= shuf[d] < dim_end).  The inner substring lies symmetrically inside the k-mer, so the inner
 * substring of the reverse-complement k-mer is the reverse complement of the forward one: a k-mer can only
 * be accepted if its FORWARD inner substring is in B, whichever strand turns out to be canonical.  The hot
 * loop therefore rolls and probes the forward strand only; canonicalisation happens for filter hits, in
 * mk_resolve_one.  One 32-bit word per probe: word = bits 10.. of the substring, three bit positions = its 5-bit
 * fields at bit offsets 0, 2 and 6.  8192 entries in 16384 words: about 0.1 % false positives; correctness never depends on
 * it (every candidate is re-checked against the accept bitmap / .shuf table). */
__device__ __forceinline__ uint32_t mk_filter_mask(uint32_t x) {
  /* bit positions = the 5-bit fields at offsets 0, 2 and 6 of the substring.  EVEN offsets on purpose: the
   * substring slides by one base (2 bits) per step, so field 2k of this base's substring is the low 5 bits of
   * the substring k bases earlier -- in the tuned loop those are already in registers (no extra shifts). */
  /* (a fourth field at offset 4 would be free in the loop -- v_or3_b32 -- but measured MORE false positives:
   * the overlapping fields are too correlated) */
  /* (other even offsets measured worse: (0,4,6) +1 %, (0,2,4) +6 %, anything using offset 8 +13..25 % -- those
   * bits also pick the word) */
  return (1u << (x & 31u)) | (1u << ((x >> 2) & 31u)) | (1u << ((x >> 6) & 31u));
}

/* rolling forward k-mer (iseq2comem.c:685).
 * K == 0: geometry from runtime parameters, 64-bit arithmetic.
 * K  > 8: geometry folded at compile time, the 4K-bit value kept as two 32-bit halves so that every
 *         step is a 32-bit VALU op (v_lshl_or / v_alignbit / v_and). */
template <int K>
struct mk_kmer {
  uint64_t f;
  __device__ __forceinline__ void reset() { f = 0; }
  __device__ __forceinline__ void roll(uint32_t code, const mk_keyparams &kp) { f = ((f << 2) | code) & kp.tupmask; }
  __device__ __forceinline__ uint64_t fwd() const { return f; }
  __device__ __forceinline__ uint32_t dimx(const mk_keyparams &kp, uint32_t dimmask) const { return (uint32_t)(f >> kp.out2) & dimmask; }
  static __device__ __forceinline__ uint32_t TL(const mk_keyparams &kp) { return kp.TL; }
};

template <int K>
struct mk_kmer_hi {
  static_assert(K > 8 && K <= 16, "two-half representation needs 32 < 4K <= 64");
  static constexpr uint32_t HMASK = (uint32_t)((1ull << (4 * K - 32)) - 1ull);
  uint32_t flo, fhi;
  __device__ __forceinline__ void reset() { flo = fhi = 0; }
  __device__ __forceinline__ void roll(uint32_t code, const mk_keyparams &) {
    const uint32_t nfhi = __builtin_amdgcn_alignbit(fhi, flo, 30) & HMASK;
    flo = (flo << 2) | code;
    fhi = nfhi;
  }
  __device__ __forceinline__ uint64_t fwd() const { return ((uint64_t)fhi << 32) | flo; }
  __device__ __forceinline__ uint32_t dimx(const mk_keyparams &kp, uint32_t dimmask) const {
    return (uint32_t)(fwd() >> kp.out2) & dimmask;
  }
  static __device__ __forceinline__ uint32_t TL(const mk_keyparams &) { return 2u * K; }
};
template <> struct mk_kmer<9> : mk_kmer_hi<9> {};
template <> struct mk_kmer<10> : mk_kmer_hi<10> {};
template <> struct mk_kmer<11> : mk_kmer_hi<11> {};
template <> struct mk_kmer<12> : mk_kmer_hi<12> {};



/* scan code of byte J of a raw dword: (byte >> 1) & 3 in one instruction */
template <int J>
__device__ __forceinline__ uint32_t mk_code_of(uint32_t w) {
  uint32_t r;
  asm("v_bfe_u32 %0, %1, %2, 2" : "=v"(r) : "v"(w), "n"(8 * J + 1));
  return r;
}

/* 1 << ((flo >> SH) & 31).  For SH == 8 (k = 11) the shift amount is byte 1 of flo, which SDWA selects for free. */
template <uint32_t SH>
__device__ __forceinline__ uint32_t mk_onehot_at(uint32_t flo) {
  if constexpr (SH == 8u) {
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
        : "=v"(r) : "v"(flo), "s"(1u));
    return r;
  } else {
    return 1u << ((flo >> SH) & 31u);
  }
}

/* SWAR helpers on four bytes */
__device__ __forceinline__ uint32_t mk_nonzero_bytes(uint32_t v) { /* bit 7 of byte j set <=> byte j != 0 */
  return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}

/* ---- pair filter of the tuned scan kernels ---------------------------------------------------------------------------
 * Two consecutive k-mers' inner substrings overlap in 22 of their 24 bits: x_{j+1} = (x_j << 2 | code) mod 2^24, so
 * z = x_j[0..21] = x_{j+1}[2..23].  The tuned loop probes ONE key z per PAIR of bases against a filter over
 *     Z = { d mod 2^22 : d in B } u { d >> 2 : d in B },       B = A u revcomp(A) as above:
 * x_j in B puts z in the first set, x_{j+1} in B in the second, so a pair without a filter hit holds no accepted k-mer.
 * Word = z[8..21] (the same 16384 words), bits = a table lookup by z[0..7] (256 masks of six bits each, 1 KiB at LDS
 * offset 0): the 22 bits of z are all used, none twice.  Half the LDS probes and a third of the instructions of the
 * per-base test; it flags about 1.6 times as many 8-base windows (z drops two bits of either substring), which the resolve
 * kernel, working per base with the filter above, throws out again. */
__device__ __forceinline__ uint32_t mk_zmask(uint32_t b) {
  /* six DISTINCT bit positions per entry (simulated over random accept sets: 1.96 % of the 8-base windows flagged, against
   * 2.46 % with four positions that may coincide; five to seven positions are within 0.03 % of each other, eight is worse) */
  uint32_t m = 0, h = b * 0x9E3779B1u + 0x7F4A7C15u;
  while (__builtin_popcount(m) < 6) {
    m |= 1u << (h >> 27);
    h = h * 0x85EBCA6Bu + 0xC2B2AE35u;
  }
  return m;
}
/* words of the pair filter: 16384 (64 KiB).  (Half of it with two 512-thread workgroups a CU measured slower: profiles/r05_scan_variants.txt) */
#define MK_ZF_WORDS 16384u
__device__ __forceinline__ void mk_build_zfilter(uint32_t *masktab, uint32_t *bitmap, const mk_scan_args &a) {
  for (uint32_t i = threadIdx.x; i < MK_ZMASK_WORDS; i += blockDim.x) masktab[i] = mk_zmask(i);
  for (uint32_t i = threadIdx.x; i < MK_ZF_WORDS; i += blockDim.x) bitmap[i] = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < a.n_accept; i += blockDim.x) {
    const uint32_t d = (uint32_t)mk_scan_to_ref_codes(a.accept[i]); /* reference coding -> scan coding (same map) */
    const uint32_t z1 = d & 0x3FFFFFu, z2 = d >> 2;
    atomicOr(&bitmap[(z1 >> 8) & (MK_ZF_WORDS - 1u)], masktab[z1 & 255u]);
    atomicOr(&bitmap[(z2 >> 8) & (MK_ZF_WORDS - 1u)], masktab[z2 & 255u]);
  }
  __syncthreads();
}

/* subk 5 (L2K11): B has 8192 members among 2^20 substrings -- one in 128, sixteen times as dense as at subk 6, and a pair key
 * (18 bits) would pass one pair in sixteen.  But 2^20 bits are 128 KiB: HALF of that fits the LDS beside the tiles, so the filter
 * is the membership bitmap of B itself with bit 5 of the substring dropped -- word = x[6..19] (16384 words), bit = x[0..4] -- and a
 * substring passes when it or its neighbour x ^ 32 is in B: one base in 64, no hash, no mask table, ONE LDS read per base. */
__device__ __forceinline__ void mk_build_xfilter(uint32_t *bitmap, const mk_scan_args &a) {
  for (uint32_t i = threadIdx.x; i < 16384u; i += blockDim.x) bitmap[i] = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < a.n_accept; i += blockDim.x) {
    const uint32_t d = (uint32_t)mk_scan_to_ref_codes(a.accept[i]) & 0xFFFFFu; /* reference coding -> scan coding (same map) */
    atomicOr(&bitmap[d >> 6], 1u << (d & 31u));
  }
  __syncthreads();
}

__device__ __forceinline__ void mk_build_filter(uint32_t *bitmap, const mk_scan_args &a) {
  for (uint32_t i = threadIdx.x; i < a.bm_words; i += blockDim.x) bitmap[i] = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < a.n_accept; i += blockDim.x) {
    const uint32_t d = (uint32_t)mk_scan_to_ref_codes(a.accept[i]); /* reference coding -> scan coding (same map) */
    atomicOr(&bitmap[(d >> 10) & (a.bm_words - 1u)], mk_filter_mask(d));
  }
  __syncthreads();
}

/* ONEPASS (only with exactly two column blocks per row): the loads for BOTH blocks of a tile are issued together,
 * so every 64-byte sector of the rows is requested once -- two separate 80-byte passes re-fetch the sector the
 * halves share (+37 % HBM reads, measured with a fetch micro-benchmark in round 1).  Costs NPIECES more piece registers. */
template <int K, int SUBK, bool VEC16, int THREADS, int NPIECES, bool ONEPASS>
__global__ void __launch_bounds__(THREADS) mk_scan_kernel(const mk_scan_args a) {
  static_assert(K == 0 || SUBK == 6 || (SUBK == 5 && K == 11), "tuned instantiations: k 9..11 with subk 6, k 11 with subk 5");
  extern __shared__ __align__(16) uint32_t lds[];
  constexpr uint32_t WAVES = THREADS / 64;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  /* LDS: [mask table (tuned kernels)] [filter] [wave tiles] [staging offset table] */
  /* filter words in this kernel's LDS: the pair filter's own count for the tuned subk-6 kernels (== a.bm_words in the product) */
  const uint32_t fwords = (K != 0 && SUBK == 6) ? MK_ZF_WORDS : a.bm_words;
  uint32_t *bitmap = lds + a.mt_words;
  uint32_t *tile = lds + a.mt_words + fwords + wave * a.wave_lds_dwords;

  if constexpr (VEC16) { /* staging offset table (see goff_of below): [2*NPIECES][64] dwords behind the wave tiles */
    if (wave == 0) {
      uint32_t *t = lds + a.mt_words + fwords + WAVES * a.wave_lds_dwords + lane;
#pragma unroll
      for (int i = 0; i < NPIECES; i++) {
        const uint32_t q = lane + 64u * i;
        const uint32_t r = (q * a.ppr_inv) >> 20, c = q - r * a.ppr;
        t[128 * i] = r * a.pitch + c * 16u;
        t[128 * i + 64] = r * a.rowdw + c * 4u;
      }
    }
  }
  /* both end with a barrier: the tables are visible to every wave */
  if constexpr (K != 0 && SUBK == 6) mk_build_zfilter(lds, bitmap, a);
  else if constexpr (K != 0) mk_build_xfilter(bitmap, a);
  else mk_build_filter(bitmap, a);

  /* the tuned loop addresses the mask table (LDS offset 0) and the filter (right behind it) absolutely */
  const uint32_t filter_base = (uint32_t)(uintptr_t)(mk_lds_cu32)bitmap;
  constexpr uint32_t MTW = (K != 0 && SUBK == 6) ? MK_ZMASK_WORDS : 0u; /* mask table in front of the filter: the pair filter only */
  if (K != 0 && (filter_base != MTW * 4u || a.mt_words != MTW || a.bm_words != 16384u || a.dimmask != (SUBK == 6 ? 0xFFFFFFu : 0xFFFFFu))) {
    if (threadIdx.x == 0) atomicOr(&a.tab.err[0], 4u);
    if (lane == 0) a.cand_count[blockIdx.x * WAVES + wave] = 0u; /* nothing for the resolve kernel to pick up */
    return;
  }
  const mk_scan_args *ka = (const mk_scan_args *)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t TL = mk_kmer<K>::TL(a.kp);
  const uint32_t dimmask = a.dimmask;
  const uint32_t wmask4 = (a.bm_words - 1u) << 2;
  /* a launch covers fewer than 2^31 reads (the engine splits larger pushes), so tile and row indices are 32-bit */
  uint64_t nreads = a.nreads;
  if (a.nreads_dev) { const uint64_t nd = *a.nreads_dev; if (nd < nreads) nreads = nd; }
  const uint32_t ntiles = (uint32_t)((nreads + 63u) >> 6);
  const uint32_t wave_global = blockIdx.x * WAVES + wave;
  const uint32_t nwaves = gridDim.x * WAVES;
  /* wave-uniform, and told so: the append below is then one store with a scalar base and a 32-bit offset */
  uint4 *const my_cand = a.cand + (size_t)__builtin_amdgcn_readfirstlane(wave_global) * a.cand_cap;
  if (wave_global >= ntiles) {
    if (lane == 0) a.cand_count[wave_global] = 0u;
    return;
  }

  /* ---- staging: global -> registers (one step ahead) -> LDS tile ------------------------------------------
   * A step is one column block (CB bytes) of one tile (64 rows).  Piece i of this lane is row r_i, 16-byte
   * (or 4-byte) column c_i of the block: both are the same for every step, so the global byte offset and
   * the LDS dword index are computed once. */
  using piece_t = typename std::conditional<VEC16, uint4, uint32_t>::type;
  constexpr int NP = NPIECES; /* pieces per lane per step: >= a.ppr (host picks the smallest instantiation) */
  constexpr uint32_t PW = VEC16 ? 16u : 4u;
  piece_t regs[NP];
  piece_t regs2[ONEPASS ? NP : 1]; /* second column block of the same tile */
  /* row r_i and column c_i of piece i never change; the two offsets derived from them are recomputed where
   * needed (a handful of ops per piece per step) rather than held in 2*NP registers */
  auto goff_calc = [&](int i) { const uint32_t q = lane + 64u * i; const uint32_t r = (q * a.ppr_inv) >> 20, c = q - r * a.ppr; return r * a.pitch + c * PW; };
  auto loff_calc = [&](int i) { const uint32_t q = lane + 64u * i; const uint32_t r = (q * a.ppr_inv) >> 20, c = q - r * a.ppr; return r * a.rowdw + c * (PW / 4u); };
  /* 16-byte path: the 2*NP offsets live in a table in LDS (they depend on the lane only, one table serves every wave).
   * Left to itself the compiler hoists them out of the tile loop, runs out of registers in the 1024-thread builds and
   * reloads them from SCRATCH at every tile: 31 MB of scratch across the grid, i.e. 2.4 GB of extra HBM reads per launch of
   * the benchmark (rocprofv3 FETCH_SIZE).  LDS reads cannot be hoisted across the tile stores and cost no VALU issue. */
  const uint32_t *offtab = lds + a.mt_words + fwords + WAVES * a.wave_lds_dwords + lane;
  auto goff_of = [&](int i) { if constexpr (VEC16) return offtab[128 * i]; else return goff_calc(i); };
  auto loff_of = [&](int i) { if constexpr (VEC16) return offtab[128 * i + 64]; else return loff_calc(i); };
  auto issue_loads = [&](uint32_t tile_id, uint32_t cb) {
    const uint32_t row0 = tile_id << 6;
    const uint8_t *base = a.rows + (uint64_t)row0 * a.pitch + (uint64_t)cb * a.CB;
    const uint32_t cols_here = min(a.CB, a.stride - cb * a.CB);
    if constexpr (ONEPASS) {
      /* host guarantees: ncb == 2, stride == 2*CB.  cb is 0 here. */
      if (row0 + 64u <= nreads) {
#pragma unroll
        for (int i = 0; i < NP; i++)
          if ((uint32_t)i < a.ppr) { regs[i] = *(const piece_t *)(base + goff_of(i)); regs2[i] = *(const piece_t *)(base + a.CB + goff_of(i)); }
      } else {
        const uint32_t rows_here = (uint32_t)(nreads - row0);
#pragma unroll
        for (int i = 0; i < NP; i++) {
          if ((uint32_t)i < a.ppr) {
            const uint32_t q = lane + 64u * i;
            const uint32_t r = (q * a.ppr_inv) >> 20;
            if (r < rows_here) { regs[i] = *(const piece_t *)(base + goff_of(i)); regs2[i] = *(const piece_t *)(base + a.CB + goff_of(i)); }
          }
        }
      }
    } else if (row0 + 64u <= nreads && cols_here == a.CB) { /* full tile, full block: no predicates */
#pragma unroll
      for (int i = 0; i < NP; i++)
        if ((uint32_t)i < a.ppr) regs[i] = *(const piece_t *)(base + goff_of(i));
    } else {
      const uint32_t rows_here = (uint32_t)(nreads - row0 < 64u ? nreads - row0 : 64u);
#pragma unroll
      for (int i = 0; i < NP; i++) {
        if ((uint32_t)i < a.ppr) {
          const uint32_t q = lane + 64u * i;
          const uint32_t r = (q * a.ppr_inv) >> 20, c = q - r * a.ppr;
          /* rows past the last read and columns past the stride are never looked at: do not load them */
          if (r < rows_here && c * PW < cols_here) regs[i] = *(const piece_t *)(base + goff_of(i));
        }
      }
    }
  };
  auto commit = [&](uint32_t cb) {
    (void)cb;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      if ((uint32_t)i < a.ppr) {
        if constexpr (VEC16) {
          uint32_t *p = tile + loff_of(i);
          if constexpr (ONEPASS) {
            const piece_t v = cb ? regs2[i] : regs[i]; /* cb is wave-uniform */
            p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
          } else {
            p[0] = regs[i].x; p[1] = regs[i].y; p[2] = regs[i].z; p[3] = regs[i].w;
          }
        } else {
          tile[loff_of(i)] = regs[i];
        }
      }
    }
  };

  mk_kmer<K> km;
  km.reset();
  uint32_t hh = 0; /* tuned loop: low halves of the forward k-mer's low word at the two previous pair (8-base) boundaries: older << 16 | newer */
  uint32_t run = 0;  /* valid bases since the last reset (the reference's base-1) */
  bool done = true;  /* this lane's row hit its '\n' (or does not exist) */
  uint32_t qn = 0;   /* candidates appended to this wave's buffer so far (wave-uniform) */
  const uint32_t *myrow = tile + lane * a.rowdw;
  uint64_t ord_row = 0;

  /* ---- building blocks ------------------------------------------------------------------------------ */
  struct quad { uint64_t u0, u1, u2, u3; uint32_t m0, m1, m2, m3, w0, w1, w2, w3; };

  auto lookup = [&](uint32_t &mask, uint32_t &word) { /* probe the current forward k-mer's inner substring */
    const uint32_t xx = km.dimx(a.kp, dimmask);
    if constexpr (K != 0) { /* the LDS holds the pair filter: x in B puts its low 22 bits into Z (mk_build_zfilter) */
      const uint32_t z = xx & 0x3FFFFFu;
      word = *(mk_lds_cu32)(uintptr_t)(((z >> 6) & (SUBK == 6 ? (MK_ZF_WORDS - 1u) << 2 : wmask4)) + filter_base);
      mask = lds[z & 255u];
    } else {
      word = *(mk_lds_cu32)(uintptr_t)(((xx >> 8) & wmask4) + filter_base);
      mask = mk_filter_mask(xx);
    }
  };
  /* append the lanes flagged in `hit` (forward k-mer `fwd` ending at row position `pos`) to this wave's
   * candidate buffer: plain stores, nothing to wait for */
  auto push_record = [&](bool hit, const uint4 r) {
    const uint64_t m = __ballot(hit);
    if (m == 0) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    if (qn + cnt > a.cand_cap) { /* buffer full (dense tables only): resolve right here */
      mk_resolve_inline(ka, hit, r, K != 0 ? nullptr : bitmap);
      return;
    }
    if (hit) my_cand[qn + mk_mbcnt(m)] = r;
    qn = __builtin_amdgcn_readfirstlane(qn + cnt);
  };
  /* the tuned loop's append: hm = __ballot(hit), not zero.  The byte offset is 32-bit (a wave's buffer is far below 4 GiB) */
  auto push_pair = [&](const bool hit, const uint64_t hm, const uint4 r) {
    const uint32_t cnt = (uint32_t)__popcll(hm);
    if (qn + cnt > a.cand_cap) { mk_resolve_inline(ka, hit, r, nullptr); return; } /* buffer full (dense tables only) */
    if (hit) *(uint4 *)((uint8_t *)my_cand + (qn + mk_mbcnt(hm)) * 16u) = r;
    qn += cnt;
  };
  auto push = [&](bool hit, uint64_t fwd, uint32_t pos) { /* one k-mer: slow paths, generic kernel */
    const uint64_t ord = ord_row | (uint64_t)pos;
    push_record(hit, make_uint4((uint32_t)fwd, (uint32_t)(fwd >> 32), (uint32_t)ord, (uint32_t)(ord >> 32) | 0x80000000u));
  };

  /* four valid bases, every lane with a full window: roll, canonical k-mer, filter probe */
  auto fast4 = [&](uint32_t codes, quad &q) {
    km.roll(codes & 3u, a.kp);         q.u0 = km.fwd(); lookup(q.m0, q.w0);
    km.roll((codes >> 8) & 3u, a.kp);  q.u1 = km.fwd(); lookup(q.m1, q.w1);
    km.roll((codes >> 16) & 3u, a.kp); q.u2 = km.fwd(); lookup(q.m2, q.w2);
    km.roll(codes >> 24, a.kp);        q.u3 = km.fwd(); lookup(q.m3, q.w3);
  };
  /* any four bytes: newline, N, ragged rows -- byte by byte, exactly iseq2comem.c:682-690 */
  auto general4 = [&](uint32_t w, uint32_t codes, uint32_t x, quad &q) {
    auto one = [&](uint32_t j, uint64_t &pu, uint32_t &pm, uint32_t &pw) {
      const uint32_t ch = (w >> (8u * j)) & 0xffu;
      const bool valid = ((x >> (8u * j)) & 0xffu) == 0u;
      if (!done && ch == '\n') done = true;
      const bool ok = valid && !done;
      if (ok) { km.roll((codes >> (8u * j)) & 3u, a.kp); run++; }
      else if (!done) run = 0; /* any other byte restarts the window (iseq2comem.c:688) */
      pu = km.fwd();
      lookup(pm, pw);
      if (!(ok && run >= TL)) { pm = 1u; pw = 0u; } /* no complete k-mer here: never a hit */
    };
    one(0, q.u0, q.m0, q.w0); one(1, q.u1, q.m1, q.w1); one(2, q.u2, q.m2, q.w2); one(3, q.u3, q.m3, q.w3);
  };
  /* examine four probes; queue the candidates (rare).  Unrolled: a base without a hit costs one compare
   * and one scalar branch */
  auto hits4 = [&](const quad &q, uint32_t pos0) {
    push((q.m0 & ~q.w0) == 0u, q.u0, pos0);
    push((q.m1 & ~q.w1) == 0u, q.u1, pos0 + 1u);
    push((q.m2 & ~q.w2) == 0u, q.u2, pos0 + 2u);
    push((q.m3 & ~q.w3) == 0u, q.u3, pos0 + 3u);
  };
  auto resolve4 = [&](const quad &q, uint32_t pos0) {
    const uint32_t t0 = q.m0 & ~q.w0, t1 = q.m1 & ~q.w1, t2 = q.m2 & ~q.w2, t3 = q.m3 & ~q.w3;
    const uint32_t mn = min(min(t0, t1), min(t2, t3));
    if (__any(mn == 0u)) hits4(q, pos0);
  };
  auto decode = [&](uint32_t w, uint32_t &codes, uint32_t &x) {
    codes = (w >> 1) & 0x03030303u; /* scan coding A0 C1 T2 G3, see mk_scan_to_ref_codes */
    /* expected upper-case letter of each code, compared with the byte folded to upper case */
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, codes);
    x = (w & 0xDFDFDFDFu) ^ expect; /* byte j zero <=> byte j in ACGTacgt */
  };
  /* One dword outside the fast path.
   *  (1) all valid and nobody completes a k-mer        -> roll only (read heads)
   *  (2) every lane in the same situation (same `run`, same valid/newline byte pattern: fixed-length
   *      reads at their head and tail)                 -> scalar control flow, no per-lane predicates
   *  (3) anything else (ragged rows, scattered N)      -> byte-wise predicated general4 */
  auto slow_dword = [&](uint32_t w, uint32_t codes, uint32_t x, uint32_t pos0) {
    if (__all(x == 0u && !done && run + 4u < TL)) {
      km.roll(codes & 3u, a.kp); km.roll((codes >> 8) & 3u, a.kp); km.roll((codes >> 16) & 3u, a.kp); km.roll(codes >> 24, a.kp);
      run += 4u;
      return;
    }
    const uint32_t inval = mk_nonzero_bytes(x);                       /* 0x80 in byte j: not ACGTacgt */
    const uint32_t notnl = mk_nonzero_bytes(w ^ 0x0A0A0A0Au);         /* 0x80 in byte j: not '\n' */
    const uint32_t pat = inval | (notnl >> 1);
    const uint32_t pat0 = __builtin_amdgcn_readfirstlane(pat), run0 = __builtin_amdgcn_readfirstlane(run);
    if (__all(pat == pat0 && run == run0 && !done)) {
      uint32_t urun = run0; /* wave-uniform copy of run */
#pragma unroll 1
      for (uint32_t j = 0; j < 4; j++) {
        if (!(pat0 & (0x40u << (8u * j)))) { done = true; break; }    /* '\n': every lane ends here */
        if (pat0 & (0x80u << (8u * j))) { urun = 0; continue; }       /* invalid byte: window restarts */
        km.roll((codes >> (8u * j)) & 3u, a.kp);
        urun++;
        if (urun >= TL) {
          uint32_t m, wd;
          lookup(m, wd);
          push((m & ~wd) == 0u, km.fwd(), pos0 + j);
        }
      }
      run = urun;
      return;
    }
    quad q;
    general4(w, codes, x, q);
    resolve4(q, pos0);
  };

  uint32_t nt_tile = wave_global; /* next step to load */
  uint32_t nt_cb = 0;
  issue_loads(nt_tile, nt_cb);
  for (uint32_t tile_id = wave_global; tile_id < ntiles; tile_id += nwaves) {
    const uint32_t row0 = tile_id << 6;
    km.reset(); run = 0; hh = 0;
    done = row0 + lane >= nreads;
    ord_row = (a.first_ord + row0 + lane) << 12;
    for (uint32_t cb = 0; cb < a.ncb; cb++) {
      mk_wave_lds_fence();
      commit(cb);
      mk_wave_lds_fence();
      if (a.rowlen) { /* virtual rows: the row stops at rowlen -- a '\n' there and behind it in that dword */
        const uint32_t c0b = cb * a.CB;
        if (a.rowlen >= c0b && a.rowlen < c0b + a.CB) {
          uint32_t *q = tile + lane * a.rowdw + ((a.rowlen - c0b) >> 2);
          const uint32_t sh = (a.rowlen & 3u) * 8u;
          *q = (*q & ((1u << sh) - 1u)) | (0x0A0A0A0Au << sh);
        }
        mk_wave_lds_fence();
      }
      if constexpr (ONEPASS) {
        /* both halves of this tile are in registers; the next tile's loads go out once the second half is in LDS */
        if (cb == 1u) { nt_tile += nwaves; if (nt_tile < ntiles) issue_loads(nt_tile, 0u); }
      } else {
        if (++nt_cb == a.ncb) { nt_cb = 0; nt_tile += nwaves; }
        if (nt_tile < ntiles) issue_loads(nt_tile, nt_cb);
      }
      if (__all(done)) continue;

      const uint32_t col0 = cb * a.CB;
      const uint32_t ndw = min(a.CB, a.stride - col0) >> 2;
      const uint32_t npairs = ndw >> 1;
      if constexpr (K != 0) {
        /* ---- tuned loop (K in 9..11, subk 6: geometry folded at compile time) --------------------------------
         * The k-mer ending at base j has its 24-bit inner substring in bits SH..SH+23 of flo(j-1), the low word
         * of the forward k-mer one base earlier, so the loop rolls ONLY the low word; the high word is rebuilt on
         * demand from hh = the low halves of flo at the two previous pair ends (fhi(j) = flo(j-16) & HMASK): for the rare filter
         * hits and when the slow path takes over.
         * Two instantiations of one pair body (8 bases = 2 dwords of the row):
         *  A  while the LIVE lanes of the wave (rows not past their newline) are IN STEP -- the same run length `urun`, every
         *     run of TL bases and more counting as TL -- and all eight bytes of every live lane are bases: one validity test
         *     and one filter-hit test per pair, run length and the first complete k-mer (jmin) in scalar registers.  Fixed-
         *     length reads without N never leave it before their last pair.
         *  B  any other pair: per lane, e = the bases in front of the first byte that is not one (an N, the newline: k-mers
         *     ending behind it cannot be complete inside this pair), jmin from the lane's own run length, which then restarts
         *     behind the last such byte; a newline finishes the lane.  Costs a dozen VALU more than A.  A lane that met an N
         *     is back in step 22 bases later; finished lanes run along on whatever their row holds, hits dropped.
         * k-mers that are not complete yet (j < jmin) or lie behind e are probed like the others and discarded by the
         * resolve kernel (jmin and e travel in the record).  This is iseq2comem.c:682-690 per lane: a base counts when the
         * TL bytes up to it are bases of one row.  The first version of B was the byte-wise predicated path of the generic
         * kernel, entered for the rest of the row: 50 M reads trimmed to 100..150 bases 4.6 -> 2.4 ms, an N in 1 % of the
         * reads 4.5 -> 2.24 ms, in 5 % 7.6 -> 2.5 ms, and the untouched rows 2.28 -> 2.14 ms on the same box (the kernel
         * lost the byte-wise path, its registers and its spills) (profiles/r02_c_probe_ragged_reads.json). */
        constexpr uint32_t SH = 2u * (K - SUBK) - 2u; /* out2 - 2 */
        constexpr uint32_t HM = mk_kmer<K>::HMASK;
        static_assert(SH + 4u * SUBK <= 32u, "inner substring must lie inside flo(j-1)");
        /* Pair probing (see mk_build_zfilter): bases (2t, 2t+1) of the 8-base window share the key z = x_{2t}[0..21],
         * which sits in bits SH+2.. of before_{2t+1} (the low word in front of base 2t+1).  Filter word z[8..21]:
         * bits SH+10.. of before_{2t+1}; mask-table entry z[0..7]: the low word D = (SH-2)/2 bases earlier holds
         * those bits at 2..9 -- a dword address after one AND.  pa[] carries those addresses out of before_{-D}..before_{-1}
         * (the previous window's last low words) for the pairs whose earlier word lies in front of this window. */
        /* subk 5 (mk_build_xfilter): ONE probe per base.  The substring of base j is x = before_j[10..29]; its filter word x[6..19]
         * sits in the top half of before_{j+1} = before_j << 2 | code (one SDWA AND makes the byte address), its bit x[0..4] in
         * bits 8..12 of before_{j-1} (byte 1: the SDWA shift takes it from there) -- three VALU per base with the funnel shift. */
        constexpr uint32_t D = SUBK == 6 ? (SH - 2u) / 2u : 1u;
        static_assert(SUBK != 6 || (SH >= 4u && SH <= 8u), "pair probing: z[0..7] at bits 2..9 of a low word D bases earlier");
        static_assert(SUBK != 5 || SH == 10u, "per-base probing: word address in WORD_1 of before_{j+1}, bit index in BYTE_1 of before_{j-1}");
        uint32_t urun = 0;
        auto in_step = [&]() -> bool { /* sets urun; needs a live lane */
          const uint32_t rc = min(run, TL);
          urun = __builtin_amdgcn_readlane(rc, (int)__builtin_ctzll(~__ballot(done) & __builtin_amdgcn_read_exec()));
          return __all(done || rc == urun);
        };
        bool step = in_step();
        uint32_t nw0 = myrow[0], nw1 = myrow[1];
        uint32_t p = 0;
        uint32_t flo = km.flo;
        uint32_t pa[3] = {0u, 0u, 0u}; /* mask-table addresses out of before_{-D}..before_{-1}: carried ready for use (no move, no AND at the use) */
        auto oh_init = [&]() { /* before_{k-D} = flo >> 2(D-k): only bits 2..9 matter, and those are exact */
          if constexpr (SUBK == 6) {
#pragma unroll
            for (uint32_t k = 0; k < D; k++) pa[k] = (flo >> (2u * (D - k))) & 0x3FCu;
          } else {
            pa[0] = flo >> 2; /* before_{-1}: its bits 8..12 (= flo[10..14]) are all that is read */
          }
        };
        oh_init();
        uint32_t w0, w1, c0, x0, c1, x1;
        auto next_pair = [&]() {
          w0 = nw0; w1 = nw1;
          nw0 = myrow[2 * p + 2]; /* unconditional prefetch: at most 2 dwords past the row, inside the wave's LDS */
          nw1 = myrow[2 * p + 3];
          decode(w0, c0, x0);
          decode(w1, c1, x1);
        };
        /* roll, probe, hand hits over.  rollonly: nobody completes a k-mer in this pair (wave-uniform); jmin, e: first base
         * with a complete k-mer and bases that count, scalar in A, per lane in B; live: this lane may have a hit, livem: the wave's
         * lanes that may */
        auto pair_body = [&](const bool rollonly, const uint32_t jmin, const uint32_t e, const bool live, const uint64_t livem) {
          const uint32_t fstart = flo;
          /* the eight codes of the pair packed big-endian into the top 16 bits of `lo` (one v_dot4_u32_u8 per dword:
           * weights 64,16,4,1; one v_perm_b32 to place the two bytes): the low word after base j is then ONE funnel
           * shift of {fstart, lo} -- no eight-long dependent roll chain */
          const uint32_t lo = __builtin_amdgcn_perm(__builtin_amdgcn_udot4(c0, 0x01041040u, 0u, false),
                                                    __builtin_amdgcn_udot4(c1, 0x01041040u, 0u, false), 0x04000C0Cu);
          if (rollonly) {
            flo = __builtin_amdgcn_alignbit(fstart, lo, 16);
            oh_init();
          } else {
            const uint32_t f0 = __builtin_amdgcn_alignbit(fstart, lo, 30), f1 = __builtin_amdgcn_alignbit(fstart, lo, 28);
            const uint32_t f2 = __builtin_amdgcn_alignbit(fstart, lo, 26), f3 = __builtin_amdgcn_alignbit(fstart, lo, 24);
            const uint32_t f4 = __builtin_amdgcn_alignbit(fstart, lo, 22), f5 = __builtin_amdgcn_alignbit(fstart, lo, 20);
            const uint32_t f6 = __builtin_amdgcn_alignbit(fstart, lo, 18), f7 = __builtin_amdgcn_alignbit(fstart, lo, 16);
            /* before_0 .. before_7; only the ones a probe names are ever computed (k = 11: f0, f2, f4, f6) */
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
              __builtin_amdgcn_sched_barrier(0); /* all probes in flight before the first result is read */
              __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) once, instead of a staggered wait per probe */
              /* t == 0 <=> every bit of the pair's mask is set in its filter word */
              const uint32_t tt0 = mm[0] & ~dd[0], tt1 = mm[1] & ~dd[1], tt2 = mm[2] & ~dd[2], tt3 = mm[3] & ~dd[3];
              fired = min(min(tt0, tt1), min(tt2, tt3)) == 0u;
            } else {
              uint32_t wd[8];
#pragma unroll
              for (uint32_t j = 0; j < 8; j++) {
                uint32_t addr; /* (before_{j+1} >> 16) & 0xFFFC = ((before_j >> 16) & 0x3FFF) << 2: byte address of word x[6..19] */
                asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
                    : "=v"(addr) : "v"(bj[j + 1u]), "v"(0xFFFCu));
                wd[j] = *(mk_lds_cu32)(uintptr_t)addr;
              }
              const uint32_t bprev = pa[0];
              pa[0] = f6; /* before_7: the next window's before_{-1} */
              flo = f7;
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_s_waitcnt(0xC07F);
              uint32_t sh[8];
#pragma unroll
              for (uint32_t j = 0; j < 8; j++) { /* word >> x[0..4]: the shift amount is byte 1 of before_{j-1} (its low five bits) */
                const uint32_t src = j == 0 ? bprev : bj[j - 1u];
                asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                    : "=v"(sh[j]) : "v"(src), "v"(wd[j]));
              }
              fired = (((sh[0] | sh[1] | sh[2]) | (sh[3] | sh[4] | sh[5]) | (sh[6] | sh[7])) & 1u) != 0u;
            }
            const bool hit = fired && live;
            const uint64_t hm = __builtin_amdgcn_ballot_w64(fired) & livem; /* (the ballot of `hit` itself goes through a 0/1 value and a second compare) */
            if (hm) {
              /* some lane's filter test fired somewhere in this pair: those lanes append one pair record and the
               * resolve kernel, which holds the exact filter, finds the base */
              const uint32_t meta = ((col0 + 8u * p) >> 3) | (jmin << 9) | (e << 12);
              push_pair(hit, hm, make_uint4(fstart, (lo & 0xFFFF0000u) | meta, hh, (uint32_t)(row0 + lane)));
            }
          }
          hh = __builtin_amdgcn_perm(hh, fstart, 0x05040100u); /* hh << 16 | fstart & 0xFFFF */
          ++p;
        };
        uint64_t donem = __ballot(done); /* finished lanes: their bytes do not count in A's validity test */
        for (;;) {
          if (step) { /* ---- A */
            bool left = false, more = true;
            auto pair_is_bases = [&]() -> bool { /* every byte of every live lane's pair is one of ACGTacgt */
              const uint64_t bad = ((uint64_t)x1 << 32) | x0; /* non-zero: a byte of this lane's pair is not ACGTacgt */
              /* tested BEFORE the probes go out: folding this test into the hit test (one branch per pair, probes
               * issued speculatively) measured 4 % slower */
              uint64_t lanes_ok; /* asm: the compiler splits the compare into an OR over a re-derived x0 and a 32-bit compare (two more VALU) */
              asm("v_cmp_eq_u64_e64 %0, 0, %1" : "=s"(lanes_ok) : "v"(bad));
              return (lanes_ok | donem) == __builtin_amdgcn_read_exec();
            };
            while (urun + 1u < TL) { /* the rows' heads: the window fills up (three pairs of a row at k = 11) */
              next_pair();
              if (!pair_is_bases()) { left = true; break; }
              pair_body(urun + 8u < TL, TL - 1u - urun, 8u, !done, ~donem);
              urun += 8u;
              if (p == npairs) { more = false; break; }
            }
            if (!left && more) { /* every base completes a k-mer: no run length to keep, jmin = 0 */
              const uint32_t p0 = p;
              for (;;) {
                next_pair();
                if (!pair_is_bases()) { left = true; break; }
                pair_body(false, 0u, 8u, !done, ~donem);
                if (p == npairs) break;
              }
              urun += 8u * (p - p0);
            }
            run = urun; /* the lanes' own counters take over */
            if (!left) break; /* block exhausted */
          } else {
            next_pair();
          }
          { /* ---- B, on the pair that is loaded */
            const uint64_t inval = ((uint64_t)mk_nonzero_bytes(x1) << 32) | mk_nonzero_bytes(x0); /* 0x80 per byte that is no base */
            const uint64_t nl = ((uint64_t)(mk_nonzero_bytes(w1 ^ 0x0A0A0A0Au) ^ 0x80808080u) << 32) |
                                (mk_nonzero_bytes(w0 ^ 0x0A0A0A0Au) ^ 0x80808080u);             /* 0x80 per '\n' */
            const uint32_t e = inval ? (uint32_t)__builtin_ctzll(inval) >> 3 : 8u;
            const uint32_t jm = run + 1u >= TL ? 0u : TL - 1u - run; /* >= e: nothing of this lane counts here */
            const bool say = !done && jm < e;
            pair_body(false, jm & 7u, e, say, __builtin_amdgcn_ballot_w64(say));
            run = inval ? (uint32_t)__builtin_clzll(inval) >> 3 : min(run + 8u, 0xFFFFu); /* bases behind the last byte that is none */
            done = done || nl != 0ull;
            donem = __ballot(done);
            if (donem == __builtin_amdgcn_read_exec() || p == npairs) break;
            step = in_step();
          }
        }
        km.flo = flo;
        km.fhi = (hh >> 16) & HM; /* = flo(-17) & HMASK at this pair boundary (kept for symmetry with the generic kernel's state) */
      } else {
        uint32_t nw0 = myrow[0], nw1 = ndw > 1 ? myrow[1] : 0x0a0a0a0au;
        for (uint32_t p = 0; p < npairs; p++) {
          const uint32_t w0 = nw0, w1 = nw1;
          if (p + 1 < npairs) { nw0 = myrow[2 * p + 2]; nw1 = myrow[2 * p + 3]; }
          else if (ndw & 1u) nw0 = myrow[ndw - 1];
          uint32_t c0, x0, c1, x1;
          decode(w0, c0, x0);
          decode(w1, c1, x1);
          const uint32_t pos0 = col0 + 8u * p;
          if (__all((x0 | x1) == 0u && !done && run + 1u >= TL)) {
            /* 8 valid bases, every lane with a full window: two batches of four probes in flight */
            quad qa, qb;
            fast4(c0, qa);
            fast4(c1, qb);
            run += 8u;
            resolve4(qa, pos0);
            resolve4(qb, pos0 + 4u);
          } else {
            slow_dword(w0, c0, x0, pos0);
            slow_dword(w1, c1, x1, pos0 + 4u);
            if (__all(done)) break;
          }
        }
        if ((ndw & 1u) && !__all(done)) {
          uint32_t c0, x0;
          decode(nw0, c0, x0);
          slow_dword(nw0, c0, x0, col0 + 4u * (ndw - 1u));
        }
      }
    }
  }
  if (lane == 0) a.cand_count[wave_global] = qn;
}

/* ---- multi-GPU: fold an exported shard into this table -------------------------------------------- */
__global__ void __launch_bounds__(1024) mk_import_kernel(mk_table tab, uint32_t S, const unsigned long long *keys,
                                                         const uint32_t *counts, const unsigned long long *ords, uint64_t n) {
  __shared__ uint32_t wg_installed;
  if (threadIdx.x == 0) wg_installed = 0u;
  __syncthreads();
  const bool front_open = mk_front_open(tab);
  uint32_t installed = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    installed += mk_upsert(tab, S, keys[i], ords[i], counts[i], front_open) ? 1u : 0u;
  if (tab.fr) {
    if (installed) atomicAdd(&wg_installed, installed);
    __syncthreads();
    if (threadIdx.x == 0) mk_front_launch_end(tab, wg_installed, gridDim.x);
  }
}

/* front table -> big table (only when the big table holds keys): the merge of two partial accumulators, as in the import.
 * Counts are clamped to 65535 first: min(sum of clamped, 65535) == min(sum, 65535), and the slot's count field cannot carry. */
__global__ void __launch_bounds__(1024) mk_front_fold_kernel(mk_table tab, uint32_t S) {
  const mk_front f = *tab.fr;
  if (__hip_atomic_load(&f.state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  const uint64_t slots = (uint64_t)f.mask + 1u;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long kc = f.kc1[i];
    if (kc == 0ull) continue;
    const uint32_t c = (uint32_t)(kc & MK_CNT_MASK);
    mk_upsert_big(tab, S, kc >> MK_CNT_BITS, ~f.ordinv1[i], c > 65535u ? 65535u : c);
    f.kc1[i] = 0ull; /* folded once: a later compaction of the same sketch (export, more pushes, finish) must not add it again */
    f.ordinv1[i] = 0ull;
  }
}

/* ---- table -> dense list of distinct keys ------------------------------------------------------------ */
struct mk_dist {
  unsigned long long *key;
  unsigned long long *ord;
  uint32_t *cnt; /* clamped to 65535: min(sum,65535) == min(sum of clamped,65535) */
  uint64_t cap;
};

/* ---- multi-GPU, merge by key slices: the distinct list cut into nparts parts by key % nparts ---------------------------------
 * count: per-part totals (LDS histogram, one global add per part and workgroup); offsets: exclusive prefix -> cursors;
 * scatter: a workgroup reserves room for its entries of part g with one add on cursor g and writes them there. */
#define MK_SPLIT_MAX 16u
__global__ void __launch_bounds__(1024) mk_split_count_kernel(mk_dist d, const unsigned long long *Dp, uint32_t nparts, unsigned long long *hist) {
  __shared__ uint32_t h[MK_SPLIT_MAX];
  if (threadIdx.x < MK_SPLIT_MAX) h[threadIdx.x] = 0u;
  __syncthreads();
  const uint64_t D = *Dp;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&h[(uint32_t)(d.key[i] % nparts)], 1u);
  __syncthreads();
  if (threadIdx.x < nparts && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
__global__ void mk_split_offsets_kernel(const unsigned long long *hist, unsigned long long *cursor, uint32_t nparts) {
  unsigned long long at = 0;
  for (uint32_t g = 0; g < nparts; g++) { cursor[g] = at; at += hist[g]; }
}
__global__ void __launch_bounds__(1024) mk_split_scatter_kernel(mk_dist d, const unsigned long long *Dp, uint32_t nparts, unsigned long long *cursor,
                                                                unsigned long long *okey, uint32_t *ocnt, unsigned long long *oord, uint64_t ocap) {
  __shared__ uint32_t h[MK_SPLIT_MAX];
  __shared__ unsigned long long base[MK_SPLIT_MAX];
  const uint64_t D = *Dp;
  const uint64_t tiles = (D + blockDim.x - 1) / blockDim.x;
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    if (threadIdx.x < MK_SPLIT_MAX) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t i = t * blockDim.x + threadIdx.x;
    unsigned long long k = 0;
    uint32_t g = 0, r = 0;
    if (i < D) { k = d.key[i]; g = (uint32_t)(k % nparts); r = atomicAdd(&h[g], 1u); }
    __syncthreads();
    if (threadIdx.x < nparts) base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)h[threadIdx.x]) : 0ull;
    __syncthreads();
    if (i < D) {
      const unsigned long long o = base[g] + r;
      if (o < ocap) { okey[o] = k; ocnt[o] = d.cnt[i]; oord[o] = d.ord[i]; }
    }
    __syncthreads();
  }
}
__global__ void mk_set_counter_kernel(unsigned long long *counter, unsigned long long v) { counter[0] = v; }


/* ---- sparse bookkeeping for large tables --------------------------------------------------------------------------
 * mk_dirty_list_kernel: bitmap -> list of set block indices (any order), one workgroup.
 * mk_dirty_clear_kernel: re-establishes "all empty" on the listed blocks of the accumulation table (zero) and, with the
 * second list, of the layout table (0xFFFFFFFF). */
__global__ void __launch_bounds__(1024) mk_dirty_list_kernel(uint32_t *bitmap, uint32_t nwords, uint32_t *list, uint32_t *count,
                                                            int clear_bitmap, const uint32_t *big_used) {
  /* behind a front table: nobody has installed a key in the big table (big_used[0] == 0) -- no block is marked */
  if (big_used && __hip_atomic_load(big_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    if (threadIdx.x == 0) *count = 0u;
    return;
  }
  __shared__ uint32_t base_s;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (uint32_t w0 = 0; w0 < nwords; w0 += 1024u) {
    const uint32_t w = w0 + threadIdx.x;
    uint32_t bits = w < nwords ? bitmap[w] : 0u;
    if (bits && clear_bitmap) bitmap[w] = 0u;
    uint32_t off = bits ? atomicAdd(&base_s, (uint32_t)__popc(bits)) : 0u;
    while (bits) {
      list[off++] = (w << 5) + (uint32_t)__builtin_ctz(bits);
      bits &= bits - 1u;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *count = base_s;
}

__global__ void __launch_bounds__(256) mk_dirty_clear_kernel(unsigned long long *kc, unsigned long long *ordinv, uint32_t S,
                                                             const uint32_t *list, const uint32_t *nlist, uint32_t shift,
                                                             uint32_t *slot, const uint32_t *slist, const uint32_t *nslist,
                                                             uint32_t sshift) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t n = *nlist, ns = slot ? *nslist : 0u;
  for (uint64_t i = wave; i < n; i += nwaves) {
    const uint64_t b0 = (uint64_t)list[i] << shift;
    for (uint32_t k = lane; k < (1u << shift); k += 64u)
      if (b0 + k < S) { kc[b0 + k] = 0ull; ordinv[b0 + k] = 0ull; }
  }
  for (uint64_t i = wave; i < ns; i += nwaves) {
    const uint64_t b0 = (uint64_t)slist[i] << sshift;
    for (uint32_t k = lane; k < (1u << sshift); k += 64u)
      if (b0 + k < S) slot[b0 + k] = MK_EMPTY32;
  }
}

/* Each wave owns MK_COMPACT_CHUNK consecutive slots (all loads of the chunk in flight together), a workgroup of
 * 16 waves owns 16 consecutive chunks and reserves its output range with ONE atomicAdd (same-address atomics
 * serialise at ~90 per microsecond on this chip, so there must be few of them). */
#ifndef MK_COMPACT_CHUNK
#define MK_COMPACT_CHUNK 2048u
#endif
#define MK_COMPACT_THREADS 1024
/* CHUNK slots per wave.  Dense: the chunks tile the table.  Sparse (list != NULL): chunk i is the dirty block list[i]
 * (CHUNK == block size), *nlist of them. */
/* what mk_sketch_begin clears in one launch instead of three or four fills (a genome of a directory is 20 launches long, every
 * one of them 3-7 us): the front table, the eight counters, the FASTA stream's state */
__global__ void __launch_bounds__(256) mk_begin_clear_kernel(uint4 *front, unsigned long long n16, unsigned long long *counters,
                                                             uint32_t *fa_state, uint32_t fa_words) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) front[i] = z;
  if (blockIdx.x == 0) {
    if (threadIdx.x < 8u) counters[threadIdx.x] = 0ull;
    if (threadIdx.x < fa_words) fa_state[threadIdx.x] = 0u;
  }
}

template <uint32_t CHUNK>
__global__ void __launch_bounds__(MK_COMPACT_THREADS) mk_compact_kernel(mk_table tab, uint32_t S, mk_dist out,
                                                                        unsigned long long *counter, int drop_key0,
                                                                        const uint32_t *list, const uint32_t *nlist,
                                                                        const uint32_t *big_used, uint32_t run_if) {
  /* with a front table: the pass over it runs when the big table is empty (run_if 0), the pass over the big table when it is
   * not (run_if 1; the front table has been folded into it by then) */
  if (big_used && (__hip_atomic_load(big_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) != (run_if != 0u)) return;
  __shared__ uint32_t wtotal[MK_COMPACT_THREADS / 64];
  __shared__ unsigned long long block_base;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  constexpr uint32_t WAVES = MK_COMPACT_THREADS / 64, ITER = CHUNK / 64u;
  const uint64_t nchunks = list ? (uint64_t)*nlist : ((uint64_t)S + CHUNK - 1) / CHUNK;
  const uint64_t nblockchunks = (nchunks + WAVES - 1) / WAVES;
  for (uint64_t bc = blockIdx.x; bc < nblockchunks; bc += gridDim.x) {
    const uint64_t ci = bc * WAVES + wave;
    const uint64_t base_slot = ci < nchunks ? (list ? (uint64_t)list[ci] : ci) * CHUNK : (uint64_t)S;
    unsigned long long kc[ITER];
#pragma unroll
    for (uint32_t it = 0; it < ITER; it++) {
      const uint64_t n = base_slot + it * 64u + lane;
      kc[it] = n < S ? tab.kc[n] : 0ull;
    }
    uint32_t total = 0;
#pragma unroll
    for (uint32_t it = 0; it < ITER; it++) {
      const bool occ = kc[it] != 0ull && !(drop_key0 && (kc[it] >> MK_CNT_BITS) == 0ull);
      total += (uint32_t)__popcll(__ballot(occ));
    }
    if (lane == 0) wtotal[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t sum = 0;
      for (uint32_t w = 0; w < WAVES; w++) sum += wtotal[w];
      block_base = sum ? atomicAdd(counter, (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    unsigned long long base = block_base;
    for (uint32_t w = 0; w < wave; w++) base += wtotal[w];
    if (total) {
#pragma unroll
      for (uint32_t it = 0; it < ITER; it++) {
        const uint64_t n = base_slot + it * 64u + lane;
        const bool occ = kc[it] != 0ull && !(drop_key0 && (kc[it] >> MK_CNT_BITS) == 0ull);
        const uint64_t m = __ballot(occ);
        if (occ) {
          const uint64_t idx = base + mk_mbcnt(m);
          if (idx < out.cap) {
            out.key[idx] = kc[it] >> MK_CNT_BITS;
            out.ord[idx] = ~tab.ordinv[n];
            const uint32_t c = (uint32_t)(kc[it] & MK_CNT_MASK);
            out.cnt[idx] = c > 65535u ? 65535u : c;
          }
        }
        base += (unsigned long long)__popcll(m);
      }
    }
    __syncthreads(); /* wtotal/block_base are reused by the next block-chunk */
  }
}

/* ---- reference-order slot layout by priority insertion ------------------------------------------------
 * Sequential FCFS insertion puts key K in the first slot of its probe sequence that no EARLIER key holds.
 * That fixed point is unique, so it can be reached in any order: a walking key takes a slot from a later
 * occupant and the evicted key resumes from the start of its own sequence. */
__global__ void __launch_bounds__(256) mk_layout_kernel(mk_dist d, const unsigned long long *Dp, unsigned long long limit,
                                                        uint32_t *slot, uint32_t S, uint32_t *err, uint32_t *dirty_slot,
                                                        uint32_t dirty_slot_shift) {
  /* the number of distinct keys is read from device memory (the compaction kernel in front of this one wrote it): the
   * host does not wait for it.  More keys than the reference admits: the host reports MK_ERR_CROWDED, nothing to lay out. */
  const uint64_t D = *Dp;
  if (D > limit) return;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t cur = (uint32_t)i;
    unsigned long long ord = d.ord[cur];
    uint32_t n, h2;
    mk_probe_init(d.key[cur], S, n, h2);
    uint64_t guard = 0;
    /* the first look at a slot is an L2-served load, which may be stale (per-XCD L2s are not coherent; the CAS
     * itself executes memory-side and is authoritative): after a failed CAS continue from the value IT returned,
     * never from a re-load.  A stale first look is harmless: occupants only ever get EARLIER, so "the occupant I
     * saw is earlier than me" stays true, and "later than me / empty" is re-checked by the CAS. */
    uint32_t old = __hip_atomic_load(&slot[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      if (++guard > 8ull * S) { atomicOr(&err[0], 2u); break; }
      if (old != MK_EMPTY32 && d.ord[old] < ord) { /* an earlier key keeps the slot: next probe */
        n = mk_probe_next(n, h2, S);
        old = __hip_atomic_load(&slot[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      const uint32_t prev = atomicCAS(&slot[n], old, cur);
      if (prev != old) { old = prev; continue; } /* somebody else changed the slot: judge the real occupant */
      if (old == MK_EMPTY32) {                   /* took an empty slot: done */
        if (dirty_slot) {
          const uint32_t b = n >> dirty_slot_shift;
          atomicOr(&dirty_slot[b >> 5], 1u << (b & 31u));
        }
        break;
      }
      cur = old; /* evicted a later key: re-walk its own sequence (everything before this slot is held by earlier keys) */
      ord = d.ord[cur];
      mk_probe_init(d.key[cur], S, n, h2);
      old = __hip_atomic_load(&slot[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

/* ---- slot-order dump, one component at a time ----------------------------------------------------------
 * each wave owns a contiguous chunk of MK_DUMP_CHUNK slots */
#define MK_DUMP_CHUNK 4096u
#define MK_DUMP_SHIFT 12u
#define MK_SPARSE_SHIFT 9u                    /* accumulation-table blocks of 512 slots (8 KiB of kc) */
#define MK_SPARSE_BLOCK (1u << MK_SPARSE_SHIFT)

struct mk_dump_args {
  const uint32_t *slot;
  uint32_t S;
  mk_dist d;
  uint32_t comp_num, comp, comp_code_bits;
  uint32_t cnt_lo, cnt_hi; /* keep keys whose (clamped) count lies in [cnt_lo, cnt_hi] */
  const uint32_t *dirty_slot; /* sparse bookkeeping: one bit per MK_DUMP_CHUNK slots of the layout table, NULL = all chunks */
  uint32_t nchunks;
  uint64_t out_cap; /* entries the output arrays hold: the write kernels do nothing when the dump is larger (host grows, reruns) */
  uint32_t *unset;  /* not NULL (== slot): every occupied slot is handed back EMPTY as it is written out, so that the layout table
                     * is all-empty again behind the dump and the next sketch needs no fill (only where no rerun can be needed) */
};

__device__ __forceinline__ bool mk_dump_pred_idx(const mk_dump_args &a, uint32_t idx);
__device__ __forceinline__ bool mk_dump_pred(const mk_dump_args &a, uint64_t n, uint32_t &idx) {
  idx = n < a.S ? a.slot[n] : MK_EMPTY32;
  return mk_dump_pred_idx(a, idx);
}
__device__ __forceinline__ bool mk_dump_pred_idx(const mk_dump_args &a, uint32_t idx) {
  if (idx == MK_EMPTY32) return false;
  if (a.comp_num > 1 && (uint32_t)(a.d.key[idx] % a.comp_num) != a.comp) return false;
  if (a.cnt_lo > 1u || a.cnt_hi != 0xffffffffu) {
    const uint32_t c = a.d.cnt[idx];
    if (c < a.cnt_lo || c > a.cnt_hi) return false;
  }
  return true;
}

__global__ void __launch_bounds__(256) mk_dump_count_kernel(mk_dump_args a, uint32_t *chunk_count) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= a.nchunks) return;
  if (a.dirty_slot && !((a.dirty_slot[chunk >> 5] >> (chunk & 31u)) & 1u)) { /* untouched chunk of a large table */
    if (lane == 0) chunk_count[chunk] = 0u;
    return;
  }
  uint32_t total = 0;
  /* eight slot words in flight: a wave that walks its chunk one dependent load at a time takes 20 us for 16 KiB */
  for (uint32_t it0 = 0; it0 < MK_DUMP_CHUNK / 64u; it0 += 8u) {
    uint32_t idx[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; u++) {
      const uint64_t n = (uint64_t)chunk * MK_DUMP_CHUNK + (it0 + u) * 64u + lane;
      idx[u] = n < a.S ? a.slot[n] : MK_EMPTY32;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8u; u++) total += (uint32_t)__popcll(__ballot(mk_dump_pred_idx(a, idx[u])));
  }
  if (lane == 0) chunk_count[chunk] = total;
}

/* exclusive scan of chunk counts by one workgroup; total -> *total_out */
__global__ void __launch_bounds__(1024) mk_dump_scan_kernel(uint32_t *chunk_count, uint32_t nchunks, unsigned long long *total_out) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry_s;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nchunks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    unsigned long long v = i < nchunks ? chunk_count[i] : 0ull, incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long t = __shfl_up(incl, o);
      if ((int)lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0;
    for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
    const unsigned long long carry = carry_s;
    if (i < nchunks) chunk_count[i] = (uint32_t)(carry + woff + incl - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry_s;
}

/* out_ids / out_cnt may be pinned host memory mapped into the device (the engine's result arrays): a wave writes
 * consecutive entries, i.e. whole 256-byte / 128-byte segments */
__global__ void __launch_bounds__(256) mk_dump_write_kernel(mk_dump_args a, const uint32_t *chunk_off,
                                                            const unsigned long long *totals, uint32_t *out_ids,
                                                            uint16_t *out_cnt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= a.nchunks) return;
  if (totals[0] > a.out_cap) return;
  if (a.dirty_slot && !((a.dirty_slot[chunk >> 5] >> (chunk & 31u)) & 1u)) return;
  uint32_t off = chunk_off[chunk];
  for (uint32_t it0 = 0; it0 < MK_DUMP_CHUNK / 64u; it0 += 8u) {
    uint32_t idxs[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; u++) {
      const uint64_t n = (uint64_t)chunk * MK_DUMP_CHUNK + (it0 + u) * 64u + lane;
      idxs[u] = n < a.S ? a.slot[n] : MK_EMPTY32;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8u; u++) {
      const uint32_t idx = idxs[u];
      const bool p = mk_dump_pred_idx(a, idx);
      const uint64_t m = __ballot(p);
      if (p) {
        const uint32_t o = off + mk_mbcnt(m);
        out_ids[o] = (uint32_t)(a.d.key[idx] >> a.comp_code_bits);
        if (out_cnt) out_cnt[o] = (uint16_t)a.d.cnt[idx];
      }
      if (a.unset && idx != MK_EMPTY32) a.unset[(uint64_t)chunk * MK_DUMP_CHUNK + (it0 + u) * 64u + lane] = MK_EMPTY32; /* (also keys the predicate drops) */
      off += (uint32_t)__popcll(m);
    }
  }
}

/* ---- slot-order dump for several components in ONE pass over the layout table (component_num is 16 whenever it
 * is not 1: 4(k-drlevel) <= 39 bits caps k-drlevel at 9).  Same chunking as above; chunk_count is [comp][chunk]. */
#define MK_MAX_COMP 16u
__device__ __forceinline__ bool mk_dumpc_pred(const mk_dump_args &a, uint64_t n, uint32_t &idx, uint32_t &comp) {
  idx = n < a.S ? a.slot[n] : MK_EMPTY32;
  comp = 0;
  if (idx == MK_EMPTY32) return false;
  if (a.cnt_lo > 1u || a.cnt_hi != 0xffffffffu) {
    const uint32_t c = a.d.cnt[idx];
    if (c < a.cnt_lo || c > a.cnt_hi) return false;
  }
  comp = (uint32_t)(a.d.key[idx] % a.comp_num);
  return true;
}

__global__ void __launch_bounds__(256) mk_dumpc_count_kernel(mk_dump_args a, uint32_t *chunk_count) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= a.nchunks) return;
  if (a.dirty_slot && !((a.dirty_slot[chunk >> 5] >> (chunk & 31u)) & 1u)) {
    if (lane < a.comp_num) chunk_count[(size_t)lane * a.nchunks + chunk] = 0u;
    return;
  }
  uint32_t mine = 0; /* lane c (< comp_num) accumulates component c */
  for (uint32_t it = 0; it < MK_DUMP_CHUNK / 64u; it++) {
    uint32_t idx, comp;
    const bool p = mk_dumpc_pred(a, (uint64_t)chunk * MK_DUMP_CHUNK + it * 64u + lane, idx, comp);
    uint64_t rest = __ballot(p);
    while (rest) { /* one round per distinct component present among the 64 slots */
      const uint32_t c = __shfl(comp, (int)__builtin_ctzll(rest));
      const uint64_t m = __ballot(p && comp == c);
      if (lane == c) mine += (uint32_t)__popcll(m);
      rest &= ~m;
    }
  }
  if (lane < a.comp_num) chunk_count[(size_t)lane * a.nchunks + chunk] = mine;
}

/* one workgroup per component: exclusive scan of that component's chunk counts; totals[c] = its size */
__global__ void __launch_bounds__(1024) mk_dumpc_scan_kernel(uint32_t *chunk_count, uint32_t nchunks, unsigned long long *totals) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry_s;
  uint32_t *cc = chunk_count + (size_t)blockIdx.x * nchunks;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nchunks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    unsigned long long v = i < nchunks ? cc[i] : 0ull, incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long t = __shfl_up(incl, o);
      if ((int)lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0;
    for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
    const unsigned long long carry = carry_s;
    if (i < nchunks) cc[i] = (uint32_t)(carry + woff + incl - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

/* out_ids/out_cnt hold the components back to back: component c starts at sum(totals[0..c)) */
__global__ void __launch_bounds__(256) mk_dumpc_write_kernel(mk_dump_args a, const uint32_t *chunk_off,
                                                             const unsigned long long *totals, uint32_t *out_ids,
                                                             uint16_t *out_cnt) {
  __shared__ uint32_t wbase[4][MK_MAX_COMP];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= a.nchunks) return;
  if (a.dirty_slot && !((a.dirty_slot[chunk >> 5] >> (chunk & 31u)) & 1u)) return; /* wave-uniform */
  {
    unsigned long long all = 0;
    for (uint32_t c = 0; c < a.comp_num; c++) all += totals[c];
    if (all > a.out_cap) return;
  }
  if (lane < a.comp_num) {
    unsigned long long start = 0;
    for (uint32_t c = 0; c < lane; c++) start += totals[c];
    wbase[wave][lane] = (uint32_t)start + chunk_off[(size_t)lane * a.nchunks + chunk];
  }
  mk_wave_lds_fence();
  for (uint32_t it = 0; it < MK_DUMP_CHUNK / 64u; it++) {
    uint32_t idx, comp;
    const bool p = mk_dumpc_pred(a, (uint64_t)chunk * MK_DUMP_CHUNK + it * 64u + lane, idx, comp);
    uint64_t rest = __ballot(p);
    while (rest) {
      const uint32_t c = __shfl(comp, (int)__builtin_ctzll(rest));
      const bool sel = p && comp == c;
      const uint64_t m = __ballot(sel);
      const uint32_t base = wbase[wave][c];
      if (sel) {
        const uint32_t o = base + mk_mbcnt(m);
        out_ids[o] = (uint32_t)(a.d.key[idx] >> a.comp_code_bits);
        if (out_cnt) out_cnt[o] = (uint16_t)a.d.cnt[idx];
      }
      mk_wave_lds_fence();
      if (lane == 0) wbase[wave][c] = base + (uint32_t)__popcll(m);
      mk_wave_lds_fence();
      rest &= ~m;
    }
    if (a.unset && idx != MK_EMPTY32) a.unset[(uint64_t)chunk * MK_DUMP_CHUNK + it * 64u + lane] = MK_EMPTY32;
  }
}

/* ---- engine start-up: the accepted inner substrings of a .shuf table (iseq2comem.c:693-694).  The host finds them while the
 * runtime creates the engine's queue (a pass over the table on a few threads) and uploads the (d, shuf[d]) pairs: a few
 * thousand of the 16^subk entries.  From them, on the device: the scan filter's source list {d, revcomp(d)}, the accept bitmap,
 * and the ONLY entries of the device's .shuf table anybody reads (the resolve kernel looks at shuf[d] behind a set accept
 * bit).  The 64 MiB table itself never crosses PCIe: 14 ms of every start-up. */
struct mk_accept_pair { uint32_t d; int32_t pf; };
__global__ void __launch_bounds__(256) mk_accept_scatter_kernel(const mk_accept_pair *pairs, uint32_t n, uint32_t dbits, int32_t *shuf,
                                                                uint32_t *bits, uint32_t *accept) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t d = pairs[i].d;
    shuf[d] = pairs[i].pf;
    atomicOr(&bits[d >> 5], 1u << (d & 31u));
    uint32_t c = ~d, r = 0; /* reverse the 2-bit groups of the complement within dbits */
    for (uint32_t k = 0; k < dbits; k += 2) r |= ((c >> k) & 3u) << (dbits - 2u - k);
    accept[2u * i] = d;
    accept[2u * i + 1u] = r;
  }
}
__global__ void mk_front_store_kernel(mk_front *dst, const mk_front v) { *dst = v; }
/* n16 16-byte units at p := v (the engine's own fill: hipMemset would bring the runtime's fill kernels in at start-up) */
__global__ void __launch_bounds__(256) mk_fill16_kernel(uint4 *p, unsigned long long n16, uint32_t v) {
  const uint4 z = make_uint4(v, v, v, v);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = z;
}

/* ---- synthetic reads: 16 bytes of one row per thread ------------------------------------------------- */
__global__ void __launch_bounds__(256) mk_synth_kernel(uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len,
                                                       uint32_t stride, uint8_t *rows) {
  const uint32_t ppr = stride >> 4; /* stride % 16 == 0 */
  const uint64_t total = nreads * ppr;
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = q / ppr;
    const uint32_t c = (uint32_t)(q - r * ppr), b0 = c * 16u;
    uint32_t v[4] = {0, 0, 0, 0};
    if (b0 <= len) {
      const uint64_t w = mk_synth_word(seed, first_read + r, b0 >> 5); /* 16 bases never straddle a 32-base word */
#pragma unroll
      for (uint32_t j = 0; j < 16; j++) {
        const uint32_t b = b0 + j;
        uint32_t ch = 0;
        if (b < len) ch = (0x54474341u >> (8u * (uint32_t)((w >> (2u * (b & 31u))) & 3u))) & 0xffu;
        else if (b == len) ch = '\n';
        v[j >> 2] |= ch << (8u * (j & 3u));
      }
    }
    *(uint4 *)(rows + r * stride + b0) = make_uint4(v[0], v[1], v[2], v[3]);
  }
}

/* ---- key-list-driven dump (large tables / small sketches) ----------------------------------------------------------------
 * write_fqkoc2files() / wrt_co2cmpn_use_inn_subctx() walk all hashsize slots (iseq2comem.c:539-553, :638-646).  A genome leaves
 * some 16 000 keys in L2K11's 537 M slots: sweeping the layout table, even only its touched 16 KiB chunks, reads megabytes per
 * key-kilobyte of output.  Here the dump starts from the distinct-key list instead: once the priority insertion has converged
 * every key finds its own slot again (a walk along its probe sequence), the order of the output is the order of
 * (component, slot), and because slots are hash values that order is produced by a bucket pass -- keys are dealt to buckets of
 * consecutive (component, slot) ranges holding about eight keys each, and a key's rank inside its bucket is counted directly.
 *   mk_kl_find_kernel     key i -> its slot n; sort key (component << 32 | n); bucket counts; the slot is handed back EMPTY
 *                         (the layout table is all-empty again without any sweep)
 *   mk_kl_scan_kernel     exclusive scan of the bucket counts (one workgroup), component totals
 *   mk_kl_scatter_kernel  (sort key, i) into its bucket's range, any order inside
 *   mk_kl_emit_kernel     rank inside the bucket by counting smaller sort keys -> output position; id and count written there
 * Cost is proportional to the number of keys, not to the table. */
struct mk_kl_args {
  mk_dist d;
  const unsigned long long *Dp; /* number of distinct keys (device) */
  unsigned long long limit;     /* more keys than this: the host reports MK_ERR_CROWDED, nothing to do */
  uint32_t *slot;
  uint32_t S;
  uint32_t comp_num, comp_code_bits, cnt_lo, cnt_hi;
  uint32_t shift, bpc;          /* bucket of (component, slot) = component * bpc + (slot >> shift): bpc = (S >> shift) + 1 per component */
  uint32_t nbuckets;            /* comp_num * bpc */
  uint32_t *bcount;             /* [nbuckets + 1]: counts, then exclusive starts */
  uint32_t *bcursor;            /* [nbuckets] */
  unsigned long long *skey;     /* [D] sort key per key index, ~0 = not part of the output */
  unsigned long long *tkey;     /* [D] bucket-ordered sort keys */
  uint32_t *tidx;               /* [D] ... and their key indices */
  unsigned long long *totals;   /* [comp_num] */
  uint32_t *out_ids;
  uint16_t *out_cnt;            /* NULL: no counts */
  uint64_t out_cap;
};

__global__ void __launch_bounds__(256) mk_kl_find_kernel(mk_kl_args a) {
  const uint64_t D = *a.Dp;
  if (D > a.limit) return;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = a.d.key[i];
    uint32_t n, h2;
    mk_probe_init(key, a.S, n, h2);
    /* the layout has converged: the key sits in the first slot of its sequence that no earlier key holds.  Slots that other
     * keys have already handed back read EMPTY: not a stop, only the key's own index is */
    for (uint64_t guard = 0; guard <= a.S; guard++) {
      if (__hip_atomic_load(&a.slot[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)i) break;
      n = mk_probe_next(n, h2, a.S);
    }
    __hip_atomic_store(&a.slot[n], MK_EMPTY32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool keep = true;
    if (a.cnt_lo > 1u || a.cnt_hi != 0xffffffffu) {
      const uint32_t c = a.d.cnt[i];
      keep = c >= a.cnt_lo && c <= a.cnt_hi;
    }
    unsigned long long sk = ~0ull;
    if (keep) {
      const uint32_t comp = a.comp_num > 1u ? (uint32_t)(key % a.comp_num) : 0u;
      sk = ((unsigned long long)comp << 32) | n;
      atomicAdd(&a.bcount[comp * a.bpc + (n >> a.shift)], 1u);
    }
    a.skey[i] = sk;
  }
}

__global__ void __launch_bounds__(1024) mk_kl_scan_kernel(mk_kl_args a) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < a.nbuckets; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < a.nbuckets ? a.bcount[i] : 0u;
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
    if (i < a.nbuckets) { a.bcount[i] = carry + woff + incl - v; a.bcursor[i] = 0u; }
    __syncthreads();
    if (threadIdx.x == 1023u) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) a.bcount[a.nbuckets] = carry_s;
  __syncthreads();
  /* component c owns the buckets [c * bpc, (c + 1) * bpc) */
  if (threadIdx.x < a.comp_num)
    a.totals[threadIdx.x] = (unsigned long long)(a.bcount[(threadIdx.x + 1u) * a.bpc] - a.bcount[threadIdx.x * a.bpc]);
}

__global__ void __launch_bounds__(256) mk_kl_scatter_kernel(mk_kl_args a) {
  const uint64_t D = *a.Dp;
  if (D > a.limit) return;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long sk = a.skey[i];
    if (sk == ~0ull) continue;
    const uint32_t b = (uint32_t)(sk >> 32) * a.bpc + ((uint32_t)sk >> a.shift);
    const uint32_t at = a.bcount[b] + atomicAdd(&a.bcursor[b], 1u);
    a.tkey[at] = sk;
    a.tidx[at] = (uint32_t)i;
  }
}

__global__ void __launch_bounds__(256) mk_kl_emit_kernel(mk_kl_args a) {
  const uint64_t D = *a.Dp;
  if (D > a.limit) return;
  const uint32_t n_out = a.bcount[a.nbuckets];
  if ((uint64_t)n_out > a.out_cap) return; /* the host grows the result arrays and launches this kernel again */
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long sk = a.tkey[j];
    const uint32_t b = (uint32_t)(sk >> 32) * a.bpc + ((uint32_t)sk >> a.shift);
    const uint32_t lo = a.bcount[b], hi = a.bcount[b + 1u];
    uint32_t rank = 0;
    for (uint32_t t = lo; t < hi; t++) rank += a.tkey[t] < sk ? 1u : 0u; /* slots are distinct: no ties */
    const uint32_t i = a.tidx[j];
    a.out_ids[lo + rank] = (uint32_t)(a.d.key[i] >> a.comp_code_bits);
    if (a.out_cnt) a.out_cnt[lo + rank] = (uint16_t)a.d.cnt[i];
  }
}
