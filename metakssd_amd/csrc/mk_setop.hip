/*
 * mk_setop.hip -- `metakssd set -u` / `set -q` on the device.
 *
 * Replaces the dictionary loops of sketch_union() (command_set.c:279-316) and uniq_sketch_union()
 * (:466-509): the reference marks every id of a combined sketch file in a 2^32-bit dictionary (plus a second
 * "still unique" dictionary for -q) and walks the dictionary to emit the ids in ascending order.  Here:
 *
 *   mk_set_mark_kernel    ids -> atomicOr into the `seen` bitmap (and, for -q, into `dup` when the bit was set)
 *   mk_set_count_kernel   popcount of seen (& ~dup) per 1024-word chunk            (one pass over the bitmaps)
 *   mk_set_scan_kernel    exclusive prefix over the 131 072 chunk counts
 *   mk_set_write_kernel   non-empty chunks only: bits -> ascending ids at the chunk's offset
 *
 * All HBM-bound byte/bit work (no contraction, MFMA does not apply): the count pass streams 512 MiB (1 GiB for -q);
 * the mark pass is bound by device-scope atomics.  The bit order inside a word is ours (LSB first); only the
 * ascending id order is observable, and that is what the reference's MSB-first walk produces too.
 */
#include <hip/hip_runtime.h>
#include "mk_poison.hip.h"

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "metakssd_hip.h"

#define MK_SET_WORDS (1ull << 27)          /* 2^32 bits */
#define MK_SET_CHUNK_WORDS 1024u           /* one wave per chunk: 16 consecutive words per lane */
#define MK_SET_NCHUNKS (uint32_t)(MK_SET_WORDS / MK_SET_CHUNK_WORDS)

struct mk_setop {
  int device = 0;
  hipStream_t stream = nullptr;
  uint32_t *d_seen = nullptr, *d_dup = nullptr;
  uint32_t *d_chunk = nullptr;              /* [NCHUNKS] counts, then exclusive offsets (low 32 bits) */
  unsigned long long *d_chunk_off = nullptr; /* [NCHUNKS] 64-bit exclusive offsets */
  unsigned long long *d_total = nullptr, *h_total = nullptr;
  uint32_t *d_stage[2] = {nullptr, nullptr};
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  uint32_t *d_out = nullptr, *h_out = nullptr;
  uint64_t out_cap = 0, h_cap = 0;
  /* mk_setop_filter: the list being filtered, its per-chunk counts / offsets, the boundary positions */
  uint32_t *d_in = nullptr;
  uint64_t in_cap = 0;
  uint32_t *d_fcount = nullptr;
  unsigned long long *d_foff = nullptr;
  uint64_t fchunk_cap = 0;
  uint16_t *d_fflags = nullptr; /* pass 1's keep decisions, 16 bits a lane: [chunks][64] */
  uint64_t fflags_cap = 0;
  unsigned long long *d_bounds = nullptr, *d_bounds_out = nullptr;
  uint64_t bounds_cap = 0;
  /* mk_setop_group: auxiliary first-position table, first-occurrence list, the taxon's slot table */
  unsigned long long *d_aux = nullptr;
  uint64_t aux_cap = 0;
  uint32_t *d_first = nullptr;
  uint64_t first_cap = 0;
  uint32_t *d_slot = nullptr;
  uint64_t slot_cap = 0;
  /* mk_setop_join: the query's ids and counts */
  uint32_t *d_qids = nullptr;
  uint64_t qids_cap = 0;
  uint16_t *d_qab = nullptr;
  uint64_t qab_cap = 0;
  int mode = -1;
  bool begun = false;
  int num_cu = 256;
  /* mk_setop_last_join_ms: events around the last join's kernels (table build + both probe passes), on the handle's stream */
  hipEvent_t ev_join[2] = {nullptr, nullptr};
  bool join_timed = false;
  char err[256] = {0};
};

static thread_local char mk_set_create_err[256];

static int mk_set_fail(mk_setop *s, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(s ? s->err : mk_set_create_err, 256, fmt, ap);
  va_end(ap);
  return code;
}

#define MK_SET_HIP(s, call)                                                                          \
  do {                                                                                               \
    hipError_t _r = (call);                                                                          \
    if (_r != hipSuccess) return mk_set_fail(s, MK_ERR_HIP, "%s: %s", #call, hipGetErrorString(_r)); \
  } while (0)

#define MK_SET_STAGE_IDS ((size_t)16 << 20) /* 64 MiB of ids per staging buffer */

/* ---- kernels ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ void mk_set_mark_one(uint32_t v, uint32_t *seen, uint32_t *dup) {
  const uint32_t w = v >> 5, bit = 1u << (v & 31u);
  if (dup) {
    /* bits are only ever set, so a (possibly stale, the per-XCD L2s are not coherent) read that already shows
     * the duplicate bit is final; otherwise the atomic's return value decides */
    if (dup[w] & bit) return;
    const uint32_t old = atomicOr(&seen[w], bit);
    if (old & bit) atomicOr(&dup[w], bit);
  } else {
    if (seen[w] & bit) return;
    atomicOr(&seen[w], bit);
  }
}

__global__ void __launch_bounds__(256) mk_set_mark_kernel(const uint32_t *ids, uint64_t n, uint32_t *seen, uint32_t *dup) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t n4 = (((uintptr_t)ids & 15u) == 0) ? n >> 2 : 0; /* 16-byte loads when the list is aligned */
  const uint4 *ids4 = reinterpret_cast<const uint4 *>(ids);
  for (uint64_t i = tid; i < n4; i += nthreads) {
    const uint4 v = ids4[i];
    mk_set_mark_one(v.x, seen, dup); mk_set_mark_one(v.y, seen, dup); mk_set_mark_one(v.z, seen, dup); mk_set_mark_one(v.w, seen, dup);
  }
  for (uint64_t i = (n4 << 2) + tid; i < n; i += nthreads) mk_set_mark_one(ids[i], seen, dup);
}

__device__ __forceinline__ uint32_t mk_set_word(const uint32_t *seen, const uint32_t *dup, uint64_t w) {
  return dup ? (seen[w] & ~dup[w]) : seen[w];
}

__global__ void __launch_bounds__(256) mk_set_count_kernel(const uint32_t *seen, const uint32_t *dup, uint32_t *chunk_count) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= MK_SET_NCHUNKS) return;
  /* 1024 words = 4 KiB per chunk: lane reads 4 x 16 bytes, coalesced (word order inside the chunk is irrelevant here) */
  const uint4 *s4 = reinterpret_cast<const uint4 *>(seen + (uint64_t)chunk * MK_SET_CHUNK_WORDS);
  const uint4 *d4 = dup ? reinterpret_cast<const uint4 *>(dup + (uint64_t)chunk * MK_SET_CHUNK_WORDS) : nullptr;
  uint32_t c = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    uint4 v = s4[k * 64u + lane];
    if (d4) { const uint4 d = d4[k * 64u + lane]; v.x &= ~d.x; v.y &= ~d.y; v.z &= ~d.z; v.w &= ~d.w; }
    c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if (lane == 0) chunk_count[chunk] = c;
}

/* exclusive prefix over the chunk counts: one workgroup, 1024 threads x 128 chunks */
__global__ void __launch_bounds__(1024) mk_set_scan_kernel(const uint32_t *chunk_count, unsigned long long *chunk_off,
                                                          unsigned long long *total) {
  __shared__ unsigned long long part[1024];
  constexpr uint32_t PER = MK_SET_NCHUNKS / 1024u;
  const uint32_t t = threadIdx.x;
  unsigned long long sum = 0;
  for (uint32_t k = 0; k < PER; k++) sum += chunk_count[t * PER + k];
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) { /* Hillis-Steele inclusive scan */
    const unsigned long long v = t >= off ? part[t - off] : 0ull;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned long long run = part[t] - sum;
  for (uint32_t k = 0; k < PER; k++) {
    chunk_off[t * PER + k] = run;
    run += chunk_count[t * PER + k];
  }
  if (t == 1023u) *total = part[t];
}

__global__ void __launch_bounds__(256) mk_set_write_kernel(const uint32_t *seen, const uint32_t *dup, const uint32_t *chunk_count,
                                                           const unsigned long long *chunk_off, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= MK_SET_NCHUNKS || chunk_count[chunk] == 0u) return; /* wave-uniform */
  /* ascending order: lane l owns the 16 consecutive words [16 l, 16 l + 16) of the chunk */
  const uint64_t w0 = (uint64_t)chunk * MK_SET_CHUNK_WORDS + 16u * lane;
  uint32_t wd[16];
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) { wd[k] = mk_set_word(seen, dup, w0 + k); mine += __popc(wd[k]); }
  uint32_t incl = mine; /* inclusive scan of the lane totals */
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += v;
  }
  uint32_t *o = out + chunk_off[chunk] + (incl - mine);
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    uint32_t bits = wd[k];
    const uint32_t base = (uint32_t)((w0 + k) << 5);
    while (bits) {
      const uint32_t b = (uint32_t)__builtin_ctz(bits);
      *o++ = base + b;
      bits &= bits - 1u;
    }
  }
}

/* ---- set -i / -s: ordered filter of an id list by dictionary membership (sketch_operate, command_set.c:392-405) ---- */
#define MK_SET_FCHUNK 1024u /* ids per wave: lane l owns the 16 consecutive ids [16 l, 16 l + 16) of the chunk */

/* Ordered stream compaction shared by -i / -s (membership), and by -g (first occurrences; occupied table slots): a
 * predicate P maps an input position to {keep?, value}.  Pass 1 counts per chunk and KEEPS every lane's 16 decisions (2 bytes a lane:
 * n / 8 bytes in all), pass 2 (after the prefix) rewrites from those and asks P only for the values of kept positions -- the predicate
 * (a dictionary probe, a bitmap read) is evaluated once per position, not twice: the join of a MarkerDB keeps 2 % of 100 M positions,
 * its second pass went from a second round of 100 M probes to 2 M (round 6: 5.1 -> see DESIGN.md 4.6).
 * Lane l owns the 16 consecutive positions [16 l, 16 l + 16) of its wave's chunk, so the output keeps the input order. */
struct mk_pred_member { /* ids[i] kept iff its dictionary bit == keep */
  const uint32_t *ids, *seen;
  uint32_t keep;
  __device__ __forceinline__ bool operator()(uint64_t i, uint32_t &v) const {
    v = ids[i];
    return ((seen[v >> 5] >> (v & 31u)) & 1u) == keep;
  }
  __device__ __forceinline__ uint32_t value(uint64_t i) const { return ids[i]; } /* of a position pass 1 has kept */
};

template <class P>
__device__ __forceinline__ uint32_t mk_set_keep16(const P &p, uint64_t i0, uint64_t n, uint32_t v[16]) {
  uint32_t flags = 0;
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    v[k] = 0u;
    if (i0 + k < n && p(i0 + k, v[k])) flags |= 1u << k;
  }
  return flags;
}

template <class P>
__global__ void __launch_bounds__(256) mk_set_fcount_kernel(const P p, uint64_t n, uint64_t nchunks, uint32_t *chunk_count, uint16_t *flags_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= nchunks) return;
  uint32_t v[16];
  const uint32_t flags = mk_set_keep16(p, chunk * MK_SET_FCHUNK + 16u * lane, n, v);
  flags_out[chunk * 64u + lane] = (uint16_t)flags;
  uint32_t c = __popc(flags);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if (lane == 0) chunk_count[chunk] = c;
}

/* exclusive prefix over any number of chunk counts: one workgroup, each thread a contiguous slice */
__global__ void __launch_bounds__(1024) mk_set_scan_n_kernel(const uint32_t *count, uint64_t nchunks, unsigned long long *off,
                                                            unsigned long long *total) {
  __shared__ unsigned long long part[1024];
  const uint32_t t = threadIdx.x;
  const uint64_t per = (nchunks + 1023u) / 1024u, lo = (uint64_t)t * per < nchunks ? (uint64_t)t * per : nchunks,
                 hi = lo + per < nchunks ? lo + per : nchunks;
  unsigned long long sum = 0;
  for (uint64_t k = lo; k < hi; k++) sum += count[k];
  part[t] = sum;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1) {
    const unsigned long long v = t >= o ? part[t - o] : 0ull;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned long long run = part[t] - sum;
  for (uint64_t k = lo; k < hi; k++) { off[k] = run; run += count[k]; }
  if (t == 1023u) *total = part[t];
}

template <class P>
__global__ void __launch_bounds__(256) mk_set_fwrite_kernel(const P p, uint64_t n, uint64_t nchunks, const uint32_t *chunk_count,
                                                            const unsigned long long *chunk_off, uint32_t *out, const uint16_t *flags_in) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= nchunks || chunk_count[chunk] == 0u) return; /* wave-uniform */
  (void)n;
  const uint64_t i0 = chunk * MK_SET_FCHUNK + 16u * lane;
  const uint32_t flags = flags_in[chunk * 64u + lane]; /* pass 1's decisions (positions at and beyond n: 0) */
  const uint32_t mine = __popc(flags);
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += u;
  }
  uint32_t *o = out + chunk_off[chunk] + (incl - mine);
#pragma unroll
  for (uint32_t k = 0; k < 16; k++)
    if ((flags >> k) & 1u) *o++ = p.value(i0 + k);
}

/* ---- set -g: one taxon's table of grouping_genomes() (command_set.c:874-915) ------------------------------------------
 * The reference inserts the taxon's ids one by one (genome order, file order) into an FCFS double-hashing table and
 * dumps the table in slot order, so the bytes encode the order of FIRST occurrences.  On the device:
 *   mk_grp_insert_kernel   id -> (id, smallest position) in an auxiliary table (any order, atomicMin)
 *   ordered compaction     positions that are the first occurrence of their id -> list L in first-occurrence order
 *   mk_grp_layout_kernel   priority insertion of L into the reference's table geometry: rank = index in L; a key takes a
 *                          slot from a later-ranked occupant (which restarts its own walk) and walks past earlier ones;
 *                          the fixed point is the sequential FCFS layout.  Probe arithmetic is the reference's 32-bit
 *                          unsigned HASH(unsigned,int,int) (global_basic.h:282-284); a key that finds no place in
 *                          table_size probes is dropped, id 0 is never stored (:886, :909)
 *   ordered compaction     occupied slots -> ids in slot order */
#define MK_GRP_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ uint32_t mk_grp_mix(uint32_t k) { /* auxiliary table only: any hash will do */
  k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
  return k;
}

__global__ void __launch_bounds__(256) mk_grp_insert_kernel(const uint32_t *ids, uint64_t n, unsigned long long *aux, uint32_t amask,
                                                            int skip_zero) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t key = ids[i];
    if (skip_zero && key == 0u) continue;
    const unsigned long long entry = ((unsigned long long)key << 32) | (uint32_t)i;
    uint32_t h = mk_grp_mix(key) & amask;
    for (;;) {
      unsigned long long cur = __hip_atomic_load(&aux[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == MK_GRP_EMPTY) {
        const unsigned long long prev = atomicCAS(&aux[h], MK_GRP_EMPTY, entry);
        if (prev == MK_GRP_EMPTY) break;
        cur = prev;
      }
      if ((uint32_t)(cur >> 32) == key) { atomicMin(&aux[h], entry); break; }
      h = (h + 1u) & amask;
    }
  }
}

struct mk_pred_first { /* position i kept iff it is the first occurrence of ids[i] (and ids[i] != 0) */
  const uint32_t *ids;
  const unsigned long long *aux;
  uint32_t amask;
  __device__ __forceinline__ bool operator()(uint64_t i, uint32_t &v) const {
    v = ids[i];
    if (v == 0u) return false;
    uint32_t h = mk_grp_mix(v) & amask;
    for (;;) {
      const unsigned long long cur = aux[h];
      if ((uint32_t)(cur >> 32) == v && cur != MK_GRP_EMPTY) return (uint32_t)cur == (uint32_t)i;
      h = (h + 1u) & amask;
    }
  }
  __device__ __forceinline__ uint32_t value(uint64_t i) const { return ids[i]; }
};

struct mk_pred_slot { /* slot s kept iff occupied; value = the id whose rank it holds */
  const uint32_t *slot, *L;
  __device__ __forceinline__ bool operator()(uint64_t s, uint32_t &v) const {
    const uint32_t r = slot[s];
    if (r == 0xFFFFFFFFu) return false;
    v = L[r];
    return true;
  }
  __device__ __forceinline__ uint32_t value(uint64_t s) const { return L[slot[s]]; }
};

/* composite -q (command_composite.c:537-553): reference position i kept iff its id occurs among the query's ids; value =
 * the query's count of that k-mer (of its first occurrence, which is what the reference's dictionary returns) */
struct mk_pred_join {
  const uint32_t *ref_ids;
  const unsigned long long *aux; /* query id -> smallest query position */
  uint32_t amask;
  const uint16_t *qry_abund;
  __device__ __forceinline__ bool operator()(uint64_t i, uint32_t &v) const {
    const uint32_t key = ref_ids[i];
    uint32_t h = mk_grp_mix(key) & amask;
    for (;;) {
      const unsigned long long cur = aux[h];
      if (cur == MK_GRP_EMPTY) return false;
      if ((uint32_t)(cur >> 32) == key) { v = qry_abund[(uint32_t)cur]; return true; }
      h = (h + 1u) & amask;
    }
  }
  __device__ __forceinline__ uint32_t value(uint64_t i) const { uint32_t v = 0; (void)(*this)(i, v); return v; } /* (kept positions only: 2 % of a MarkerDB) */
};

__global__ void __launch_bounds__(256) mk_grp_layout_kernel(const uint32_t *L, uint32_t D, uint32_t *slot, uint32_t S) {
  for (uint32_t r0 = blockIdx.x * blockDim.x + threadIdx.x; r0 < D; r0 += gridDim.x * blockDim.x) {
    uint32_t cur = r0, key = L[cur], h1 = key % S, h2 = 1u + key % (S - 1u), x = 0;
    while (x < S) {
      const uint32_t y = (h1 + x * h2) % S; /* unsigned 32-bit wrap-around included, as the reference computes it */
      uint32_t occ = __hip_atomic_load(&slot[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (occ == 0xFFFFFFFFu) {
        occ = atomicCAS(&slot[y], 0xFFFFFFFFu, cur);
        if (occ == 0xFFFFFFFFu) break; /* settled */
      }
      if (occ > cur) { /* a later first occurrence holds the slot: take it, the evicted key starts over */
        const uint32_t prev = atomicCAS(&slot[y], occ, cur);
        if (prev == occ) { cur = occ; key = L[cur]; h1 = key % S; h2 = 1u + key % (S - 1u); x = 0; }
        /* else: the slot changed under us -- look at it again */
      } else {
        x++; /* an earlier first occurrence: walk on */
      }
    }
  }
}

/* kept entries in front of each boundary position (combco.index.N -> the output's index) */
template <class P>
__global__ void __launch_bounds__(256) mk_set_bounds_kernel(const P p, uint64_t n, uint64_t nchunks, const unsigned long long *chunk_off,
                                                            const unsigned long long *total, const unsigned long long *bounds,
                                                            uint32_t nb, unsigned long long *out, const uint16_t *flags_in) {
  (void)p;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  uint64_t b = bounds[j];
  if (b > n) b = n;
  const uint64_t c = b / MK_SET_FCHUNK;
  unsigned long long acc = c < nchunks ? chunk_off[c] : *total;
  for (uint64_t i = c * MK_SET_FCHUNK; i < b; i++) acc += (flags_in[i >> 4] >> (i & 15u)) & 1u; /* (position i: lane (i % 1024) / 16 of chunk i / 1024, bit i % 16) */
  out[j] = acc;
}

/* ---- host side ------------------------------------------------------------------------------------------ */
extern "C" int mk_setop_create(int device, mk_setop **out) {
  if (!out) return MK_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return mk_set_fail(nullptr, MK_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= ndev) return mk_set_fail(nullptr, MK_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  mk_setop *s = new (std::nothrow) mk_setop();
  if (!s) return MK_ERR_NOMEM;
  s->device = device;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete s;
    return mk_set_fail(nullptr, MK_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
  }
  s->num_cu = prop.multiProcessorCount;
  hipError_t r = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (r == hipSuccess) r = mk_dev_alloc(&s->d_seen, MK_SET_WORDS * 4);
  if (r == hipSuccess) r = mk_dev_alloc(&s->d_dup, MK_SET_WORDS * 4);
  if (r == hipSuccess) r = mk_dev_alloc(&s->d_chunk, (size_t)MK_SET_NCHUNKS * 4);
  if (r == hipSuccess) r = mk_dev_alloc(&s->d_chunk_off, (size_t)MK_SET_NCHUNKS * 8);
  if (r == hipSuccess) r = mk_dev_alloc(&s->d_total, 8);
  if (r == hipSuccess) r = mk_pin_alloc((void **)&s->h_total, 8, hipHostMallocDefault);
  for (int b = 0; b < 2 && r == hipSuccess; b++) {
    r = mk_dev_alloc(&s->d_stage[b], MK_SET_STAGE_IDS * 4);
    if (r == hipSuccess) r = hipEventCreateWithFlags(&s->ev_stage[b], hipEventDisableTiming);
  }
  if (r != hipSuccess) {
    mk_set_fail(nullptr, MK_ERR_NOMEM, "setop allocation: %s", hipGetErrorString(r));
    mk_setop_destroy(s);
    return MK_ERR_NOMEM;
  }
  *out = s;
  return MK_OK;
}

extern "C" int mk_setop_destroy(mk_setop *s) {
  if (!s) return MK_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  (void)hipFree(s->d_seen); (void)hipFree(s->d_dup); (void)hipFree(s->d_chunk); (void)hipFree(s->d_chunk_off);
  (void)hipFree(s->d_total); (void)hipFree(s->d_out);
  (void)hipFree(s->d_aux); (void)hipFree(s->d_first); (void)hipFree(s->d_slot); (void)hipFree(s->d_qids); (void)hipFree(s->d_qab);
  (void)hipFree(s->d_in); (void)hipFree(s->d_fcount); (void)hipFree(s->d_foff); (void)hipFree(s->d_fflags); (void)hipFree(s->d_bounds); (void)hipFree(s->d_bounds_out);
  if (s->h_total) (void)hipHostFree(s->h_total);
  if (s->h_out) (void)hipHostFree(s->h_out);
  for (int b = 0; b < 2; b++) {
    (void)hipFree(s->d_stage[b]);
    if (s->ev_stage[b]) (void)hipEventDestroy(s->ev_stage[b]);
    if (s->ev_join[b]) (void)hipEventDestroy(s->ev_join[b]);
  }
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return MK_OK;
}

extern "C" const char *mk_setop_last_error(const mk_setop *s) { return s ? s->err : mk_set_create_err; }

extern "C" int mk_setop_begin(mk_setop *s, int mode) {
  if (!s || (mode != MK_SET_UNION && mode != MK_SET_UNIQ_UNION)) return MK_ERR_ARG;
  MK_SET_HIP(s, hipSetDevice(s->device));
  /* memset(dict,0,..) command_set.c:284,473; the reference's second dictionary starts all-ones ("unique") and
   * clears bits, ours starts all-zero and sets "duplicate" bits: seen & ~dup is the same set */
  MK_SET_HIP(s, hipMemsetAsync(s->d_seen, 0, MK_SET_WORDS * 4, s->stream));
  if (mode == MK_SET_UNIQ_UNION) MK_SET_HIP(s, hipMemsetAsync(s->d_dup, 0, MK_SET_WORDS * 4, s->stream));
  s->mode = mode;
  s->begun = true;
  return MK_OK;
}

static int mk_set_mark(mk_setop *s, const uint32_t *ids_dev, uint64_t n) {
  uint64_t blocks = (n / 4 + 255) / 256;
  if (blocks > (uint64_t)s->num_cu * 32) blocks = (uint64_t)s->num_cu * 32;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(mk_set_mark_kernel, dim3((unsigned)blocks), dim3(256), 0, s->stream, ids_dev, n, s->d_seen,
                     s->mode == MK_SET_UNIQ_UNION ? s->d_dup : nullptr);
  MK_SET_HIP(s, hipGetLastError());
  return MK_OK;
}

extern "C" int mk_setop_add_device(mk_setop *s, const uint32_t *ids_dev, uint64_t n) {
  if (!s) return MK_ERR_ARG;
  if (!s->begun) return mk_set_fail(s, MK_ERR_STATE, "mk_setop_add before mk_setop_begin");
  if (n == 0) return MK_OK;
  if (!ids_dev) return MK_ERR_ARG;
  MK_SET_HIP(s, hipSetDevice(s->device));
  return mk_set_mark(s, ids_dev, n);
}

extern "C" int mk_setop_add(mk_setop *s, const uint32_t *ids, uint64_t n) {
  if (!s) return MK_ERR_ARG;
  if (!s->begun) return mk_set_fail(s, MK_ERR_STATE, "mk_setop_add before mk_setop_begin");
  if (n == 0) return MK_OK;
  if (!ids) return MK_ERR_ARG;
  MK_SET_HIP(s, hipSetDevice(s->device));
  uint64_t done = 0;
  int b = 0;
  while (done < n) { /* double-buffered: the copy of chunk i+1 overlaps the marking of chunk i */
    const uint64_t m = n - done < MK_SET_STAGE_IDS ? n - done : MK_SET_STAGE_IDS;
    MK_SET_HIP(s, hipEventSynchronize(s->ev_stage[b])); /* previous marking out of this buffer is done */
    MK_SET_HIP(s, hipMemcpyAsync(s->d_stage[b], ids + done, m * 4, hipMemcpyHostToDevice, s->stream));
    int rc = mk_set_mark(s, s->d_stage[b], m);
    if (rc) return rc;
    MK_SET_HIP(s, hipEventRecord(s->ev_stage[b], s->stream));
    done += m;
    b ^= 1;
  }
  MK_SET_HIP(s, hipStreamSynchronize(s->stream)); /* the caller's buffer is free again */
  return MK_OK;
}

extern "C" int mk_setop_finish(mk_setop *s, const uint32_t **ids_out, uint64_t *n_out) {
  if (!s || !ids_out || !n_out) return MK_ERR_ARG;
  if (!s->begun) return mk_set_fail(s, MK_ERR_STATE, "mk_setop_finish before mk_setop_begin");
  MK_SET_HIP(s, hipSetDevice(s->device));
  const uint32_t *dup = s->mode == MK_SET_UNIQ_UNION ? s->d_dup : nullptr;
  const unsigned blocks = MK_SET_NCHUNKS / 4u; /* 4 waves (chunks) per 256-thread block */
  hipLaunchKernelGGL(mk_set_count_kernel, dim3(blocks), dim3(256), 0, s->stream, s->d_seen, dup, s->d_chunk);
  hipLaunchKernelGGL(mk_set_scan_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_chunk, s->d_chunk_off, s->d_total);
  MK_SET_HIP(s, hipGetLastError());
  MK_SET_HIP(s, hipMemcpyAsync(s->h_total, s->d_total, 8, hipMemcpyDeviceToHost, s->stream));
  MK_SET_HIP(s, hipStreamSynchronize(s->stream));
  const uint64_t total = *s->h_total;
  if (total > s->out_cap) {
    (void)hipFree(s->d_out);
    s->d_out = nullptr; s->out_cap = 0;
    const uint64_t cap = total + total / 8 + 1024;
    MK_SET_HIP(s, mk_dev_alloc(&s->d_out, cap * 4));
    s->out_cap = cap;
  }
  if (total > s->h_cap) {
    if (s->h_out) (void)hipHostFree(s->h_out);
    s->h_out = nullptr; s->h_cap = 0;
    const uint64_t cap = total + total / 8 + 1024;
    MK_SET_HIP(s, mk_pin_alloc((void **)&s->h_out, cap * 4, hipHostMallocDefault));
    s->h_cap = cap;
  }
  if (total) {
    hipLaunchKernelGGL(mk_set_write_kernel, dim3(blocks), dim3(256), 0, s->stream, s->d_seen, dup, s->d_chunk, s->d_chunk_off,
                       s->d_out);
    MK_SET_HIP(s, hipGetLastError());
    MK_SET_HIP(s, hipMemcpyAsync(s->h_out, s->d_out, total * 4, hipMemcpyDeviceToHost, s->stream));
    MK_SET_HIP(s, hipStreamSynchronize(s->stream));
  }
  *ids_out = s->h_out;
  *n_out = total;
  s->begun = false;
  return MK_OK;
}

static int mk_set_result_to_host(mk_setop *s, uint64_t total);

static int mk_set_grow(mk_setop *s, void **p, uint64_t *cap, uint64_t need, size_t elem) {
  if (need <= *cap) { /* (MK_POISON: the scratch of the call before, filled on the stream this call's kernels follow on) */
    if (*p) MK_SET_HIP(s, mk_dev_repoison(*p, (size_t)*cap * elem, s->stream));
    return MK_OK;
  }
  (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const uint64_t c = need + need / 8 + 1024;
  MK_SET_HIP(s, mk_dev_alloc(p, c * elem));
  *cap = c;
  return MK_OK;
}

extern "C" int mk_setop_filter(mk_setop *s, int keep_members, const uint32_t *ids, uint64_t n, const uint64_t *bounds, uint32_t nb,
                               const uint32_t **ids_out, uint64_t *n_out, uint64_t *bounds_out) {
  if (!s || !ids_out || !n_out || (n && !ids) || (nb && (!bounds || !bounds_out))) return MK_ERR_ARG;
  if (!s->begun || s->mode != MK_SET_UNION) return mk_set_fail(s, MK_ERR_STATE, "mk_setop_filter needs a union dictionary (begin + add of the pan ids)");
  MK_SET_HIP(s, hipSetDevice(s->device));
  const uint32_t keep = keep_members ? 1u : 0u;
  const uint64_t nchunks = (n + MK_SET_FCHUNK - 1) / MK_SET_FCHUNK;
  int rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_in, &s->in_cap, n, 4))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_out, &s->out_cap, n, 4))) return rc;
  { /* counts and offsets share one capacity */
    uint64_t c1 = s->fchunk_cap, c2 = s->fchunk_cap;
    if ((rc = mk_set_grow(s, (void **)&s->d_fcount, &c1, nchunks, 4))) return rc;
    if ((rc = mk_set_grow(s, (void **)&s->d_foff, &c2, nchunks, 8))) return rc;
    s->fchunk_cap = c1 < c2 ? c1 : c2;
    if ((rc = mk_set_grow(s, (void **)&s->d_fflags, &s->fflags_cap, nchunks * 64u, 2))) return rc;
  }
  {
    uint64_t c1 = s->bounds_cap, c2 = s->bounds_cap;
    if ((rc = mk_set_grow(s, (void **)&s->d_bounds, &c1, nb, 8))) return rc;
    if ((rc = mk_set_grow(s, (void **)&s->d_bounds_out, &c2, nb, 8))) return rc;
    s->bounds_cap = c1 < c2 ? c1 : c2;
  }
  uint64_t total = 0;
  if (n) {
    MK_SET_HIP(s, hipMemcpyAsync(s->d_in, ids, n * 4, hipMemcpyHostToDevice, s->stream));
    const unsigned blocks = (unsigned)((nchunks + 3) / 4);
    const mk_pred_member pm{s->d_in, s->d_seen, keep};
    hipLaunchKernelGGL(mk_set_fcount_kernel<mk_pred_member>, dim3(blocks), dim3(256), 0, s->stream, pm, n, nchunks, s->d_fcount, s->d_fflags);
    hipLaunchKernelGGL(mk_set_scan_n_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_fcount, nchunks, s->d_foff, s->d_total);
    hipLaunchKernelGGL(mk_set_fwrite_kernel<mk_pred_member>, dim3(blocks), dim3(256), 0, s->stream, pm, n, nchunks, s->d_fcount,
                       s->d_foff, s->d_out, (const uint16_t *)s->d_fflags);
    MK_SET_HIP(s, hipGetLastError());
    MK_SET_HIP(s, hipMemcpyAsync(s->h_total, s->d_total, 8, hipMemcpyDeviceToHost, s->stream));
    if (nb) {
      MK_SET_HIP(s, hipMemcpyAsync(s->d_bounds, bounds, (size_t)nb * 8, hipMemcpyHostToDevice, s->stream));
      hipLaunchKernelGGL(mk_set_bounds_kernel<mk_pred_member>, dim3((nb + 255) / 256), dim3(256), 0, s->stream, pm, n, nchunks,
                         s->d_foff, s->d_total, s->d_bounds, nb, s->d_bounds_out, (const uint16_t *)s->d_fflags);
      MK_SET_HIP(s, hipGetLastError());
      MK_SET_HIP(s, hipMemcpyAsync(bounds_out, s->d_bounds_out, (size_t)nb * 8, hipMemcpyDeviceToHost, s->stream));
    }
    MK_SET_HIP(s, hipStreamSynchronize(s->stream));
    total = *s->h_total;
  } else {
    for (uint32_t j = 0; j < nb; j++) bounds_out[j] = 0;
    *s->h_total = 0;
  }
  if ((rc = mk_set_result_to_host(s, total))) return rc;
  *ids_out = s->h_out;
  *n_out = total;
  return MK_OK;
}

static int mk_set_result_to_host(mk_setop *s, uint64_t total) {
  if (total > s->h_cap) {
    if (s->h_out) (void)hipHostFree(s->h_out);
    s->h_out = nullptr; s->h_cap = 0;
    const uint64_t cap = total + total / 8 + 1024;
    MK_SET_HIP(s, mk_pin_alloc((void **)&s->h_out, cap * 4, hipHostMallocDefault));
    s->h_cap = cap;
  }
  if (total) {
    MK_SET_HIP(s, hipMemcpyAsync(s->h_out, s->d_out, total * 4, hipMemcpyDeviceToHost, s->stream));
    MK_SET_HIP(s, hipStreamSynchronize(s->stream));
  }
  return MK_OK;
}

extern "C" int mk_setop_group(mk_setop *s, const uint32_t *ids, uint64_t n, uint32_t table_size, const uint32_t **ids_out,
                              uint64_t *n_out) {
  if (!s || !ids_out || !n_out || (n && !ids) || table_size < 3u) return MK_ERR_ARG;
  if (n >= 0xFFFFFFFFull) return mk_set_fail(s, MK_ERR_ARG, "mk_setop_group: more than 2^32-2 ids in one taxon");
  MK_SET_HIP(s, hipSetDevice(s->device));
  *ids_out = s->h_out;
  *n_out = 0;
  if (n == 0) return MK_OK;
  int rc;
  uint64_t asize = 1024;
  while (asize < 2 * n) asize <<= 1;
  const uint64_t nchunks = (n + MK_SET_FCHUNK - 1) / MK_SET_FCHUNK, schunks = ((uint64_t)table_size + MK_SET_FCHUNK - 1) / MK_SET_FCHUNK;
  const uint64_t maxchunks = nchunks > schunks ? nchunks : schunks;
  if ((rc = mk_set_grow(s, (void **)&s->d_in, &s->in_cap, n, 4))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_first, &s->first_cap, n, 4))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_out, &s->out_cap, n, 4))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_aux, &s->aux_cap, asize, 8))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_slot, &s->slot_cap, table_size, 4))) return rc;
  {
    uint64_t c1 = s->fchunk_cap, c2 = s->fchunk_cap;
    if ((rc = mk_set_grow(s, (void **)&s->d_fcount, &c1, maxchunks, 4))) return rc;
    if ((rc = mk_set_grow(s, (void **)&s->d_foff, &c2, maxchunks, 8))) return rc;
    s->fchunk_cap = c1 < c2 ? c1 : c2;
    if ((rc = mk_set_grow(s, (void **)&s->d_fflags, &s->fflags_cap, maxchunks * 64u, 2))) return rc;
  }
  MK_SET_HIP(s, hipMemcpyAsync(s->d_in, ids, n * 4, hipMemcpyHostToDevice, s->stream));
  MK_SET_HIP(s, hipMemsetAsync(s->d_aux, 0xFF, asize * 8, s->stream));
  MK_SET_HIP(s, hipMemsetAsync(s->d_slot, 0xFF, (size_t)table_size * 4, s->stream));
  uint64_t ib = (n + 255) / 256;
  if (ib > (uint64_t)s->num_cu * 16) ib = (uint64_t)s->num_cu * 16;
  hipLaunchKernelGGL(mk_grp_insert_kernel, dim3((unsigned)ib), dim3(256), 0, s->stream, s->d_in, n, s->d_aux, (uint32_t)(asize - 1), 1);
  /* first occurrences, in input order */
  const mk_pred_first pf{s->d_in, s->d_aux, (uint32_t)(asize - 1)};
  const unsigned fb = (unsigned)((nchunks + 3) / 4);
  hipLaunchKernelGGL(mk_set_fcount_kernel<mk_pred_first>, dim3(fb), dim3(256), 0, s->stream, pf, n, nchunks, s->d_fcount, s->d_fflags);
  hipLaunchKernelGGL(mk_set_scan_n_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_fcount, nchunks, s->d_foff, s->d_total);
  hipLaunchKernelGGL(mk_set_fwrite_kernel<mk_pred_first>, dim3(fb), dim3(256), 0, s->stream, pf, n, nchunks, s->d_fcount, s->d_foff,
                     s->d_first, (const uint16_t *)s->d_fflags);
  MK_SET_HIP(s, hipGetLastError());
  MK_SET_HIP(s, hipMemcpyAsync(s->h_total, s->d_total, 8, hipMemcpyDeviceToHost, s->stream));
  MK_SET_HIP(s, hipStreamSynchronize(s->stream));
  const uint64_t D = *s->h_total;
  if (D) {
    uint64_t lb = (D + 255) / 256;
    if (lb > (uint64_t)s->num_cu * 16) lb = (uint64_t)s->num_cu * 16;
    hipLaunchKernelGGL(mk_grp_layout_kernel, dim3((unsigned)lb), dim3(256), 0, s->stream, s->d_first, (uint32_t)D, s->d_slot, table_size);
    const mk_pred_slot ps{s->d_slot, s->d_first};
    const unsigned sb = (unsigned)((schunks + 3) / 4);
    hipLaunchKernelGGL(mk_set_fcount_kernel<mk_pred_slot>, dim3(sb), dim3(256), 0, s->stream, ps, (uint64_t)table_size, schunks, s->d_fcount, s->d_fflags);
    hipLaunchKernelGGL(mk_set_scan_n_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_fcount, schunks, s->d_foff, s->d_total);
    hipLaunchKernelGGL(mk_set_fwrite_kernel<mk_pred_slot>, dim3(sb), dim3(256), 0, s->stream, ps, (uint64_t)table_size, schunks, s->d_fcount,
                       s->d_foff, s->d_out, (const uint16_t *)s->d_fflags);
    MK_SET_HIP(s, hipGetLastError());
    MK_SET_HIP(s, hipMemcpyAsync(s->h_total, s->d_total, 8, hipMemcpyDeviceToHost, s->stream));
    MK_SET_HIP(s, hipStreamSynchronize(s->stream));
  }
  const uint64_t total = D ? *s->h_total : 0;
  if ((rc = mk_set_result_to_host(s, total))) return rc;
  *ids_out = s->h_out;
  *n_out = total;
  return MK_OK;
}

extern "C" int mk_setop_join(mk_setop *s, const uint32_t *qry_ids, const uint16_t *qry_counts, uint64_t nq, const uint32_t *ref_ids,
                             uint64_t nref, const uint64_t *bounds, uint32_t nb, const uint32_t **counts_out, uint64_t *n_out,
                             uint64_t *bounds_out) {
  if (!s || !counts_out || !n_out || (nq && (!qry_ids || !qry_counts)) || (nref && !ref_ids) || (nb && (!bounds || !bounds_out)))
    return MK_ERR_ARG;
  if (nq >= 0xFFFFFFFFull) return mk_set_fail(s, MK_ERR_ARG, "mk_setop_join: more than 2^32-2 query ids");
  MK_SET_HIP(s, hipSetDevice(s->device));
  *counts_out = s->h_out;
  *n_out = 0;
  for (uint32_t j = 0; j < nb; j++) bounds_out[j] = 0;
  if (nq == 0 || nref == 0) return MK_OK;
  int rc;
  uint64_t asize = 1024;
  while (asize < 2 * nq) asize <<= 1;
  const uint64_t nchunks = (nref + MK_SET_FCHUNK - 1) / MK_SET_FCHUNK;
  if ((rc = mk_set_grow(s, (void **)&s->d_qids, &s->qids_cap, nq, 4))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_qab, &s->qab_cap, nq, 2))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_aux, &s->aux_cap, asize, 8))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_in, &s->in_cap, nref, 4))) return rc;
  if ((rc = mk_set_grow(s, (void **)&s->d_out, &s->out_cap, nref, 4))) return rc;
  {
    uint64_t c1 = s->fchunk_cap, c2 = s->fchunk_cap;
    if ((rc = mk_set_grow(s, (void **)&s->d_fcount, &c1, nchunks, 4))) return rc;
    if ((rc = mk_set_grow(s, (void **)&s->d_foff, &c2, nchunks, 8))) return rc;
    s->fchunk_cap = c1 < c2 ? c1 : c2;
    if ((rc = mk_set_grow(s, (void **)&s->d_fflags, &s->fflags_cap, nchunks * 64u, 2))) return rc;
  }
  {
    uint64_t c1 = s->bounds_cap, c2 = s->bounds_cap;
    if ((rc = mk_set_grow(s, (void **)&s->d_bounds, &c1, nb, 8))) return rc;
    if ((rc = mk_set_grow(s, (void **)&s->d_bounds_out, &c2, nb, 8))) return rc;
    s->bounds_cap = c1 < c2 ? c1 : c2;
  }
  MK_SET_HIP(s, hipMemcpyAsync(s->d_qids, qry_ids, nq * 4, hipMemcpyHostToDevice, s->stream));
  MK_SET_HIP(s, hipMemcpyAsync(s->d_qab, qry_counts, nq * 2, hipMemcpyHostToDevice, s->stream));
  MK_SET_HIP(s, hipMemcpyAsync(s->d_in, ref_ids, nref * 4, hipMemcpyHostToDevice, s->stream));
  for (int b = 0; b < 2; b++) if (!s->ev_join[b]) MK_SET_HIP(s, hipEventCreate(&s->ev_join[b]));
  s->join_timed = false;
  MK_SET_HIP(s, hipEventRecord(s->ev_join[0], s->stream)); /* (behind the uploads: the window below holds kernels and one memset only) */
  MK_SET_HIP(s, hipMemsetAsync(s->d_aux, 0xFF, asize * 8, s->stream));
  uint64_t ib = (nq + 255) / 256;
  if (ib > (uint64_t)s->num_cu * 16) ib = (uint64_t)s->num_cu * 16;
  hipLaunchKernelGGL(mk_grp_insert_kernel, dim3((unsigned)ib), dim3(256), 0, s->stream, s->d_qids, nq, s->d_aux, (uint32_t)(asize - 1), 0);
  const mk_pred_join pj{s->d_in, s->d_aux, (uint32_t)(asize - 1), s->d_qab};
  const unsigned fb = (unsigned)((nchunks + 3) / 4);
  hipLaunchKernelGGL(mk_set_fcount_kernel<mk_pred_join>, dim3(fb), dim3(256), 0, s->stream, pj, nref, nchunks, s->d_fcount, s->d_fflags);
  hipLaunchKernelGGL(mk_set_scan_n_kernel, dim3(1), dim3(1024), 0, s->stream, s->d_fcount, nchunks, s->d_foff, s->d_total);
  hipLaunchKernelGGL(mk_set_fwrite_kernel<mk_pred_join>, dim3(fb), dim3(256), 0, s->stream, pj, nref, nchunks, s->d_fcount, s->d_foff,
                     s->d_out, (const uint16_t *)s->d_fflags);
  MK_SET_HIP(s, hipGetLastError());
  MK_SET_HIP(s, hipEventRecord(s->ev_join[1], s->stream));
  s->join_timed = true;
  MK_SET_HIP(s, hipMemcpyAsync(s->h_total, s->d_total, 8, hipMemcpyDeviceToHost, s->stream));
  if (nb) {
    MK_SET_HIP(s, hipMemcpyAsync(s->d_bounds, bounds, (size_t)nb * 8, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(mk_set_bounds_kernel<mk_pred_join>, dim3((nb + 255) / 256), dim3(256), 0, s->stream, pj, nref, nchunks, s->d_foff,
                       s->d_total, s->d_bounds, nb, s->d_bounds_out, (const uint16_t *)s->d_fflags);
    MK_SET_HIP(s, hipGetLastError());
    MK_SET_HIP(s, hipMemcpyAsync(bounds_out, s->d_bounds_out, (size_t)nb * 8, hipMemcpyDeviceToHost, s->stream));
  }
  MK_SET_HIP(s, hipStreamSynchronize(s->stream));
  const uint64_t total = *s->h_total;
  if ((rc = mk_set_result_to_host(s, total))) return rc;
  *counts_out = s->h_out;
  *n_out = total;
  return MK_OK;
}

/* the last mk_setop_join's device time: dictionary build (memset + insert kernel) and the two passes over the reference ids (count,
 * write), from HIP events on the handle's stream; the uploads in front and the result copies behind are outside the window */
extern "C" int mk_setop_last_join_ms(mk_setop *s, double *ms) {
  if (!s || !ms) return MK_ERR_ARG;
  *ms = 0.0;
  if (!s->join_timed) return mk_set_fail(s, MK_ERR_STATE, "mk_setop_last_join_ms: no join has run on this handle");
  MK_SET_HIP(s, hipSetDevice(s->device));
  float f = 0.f;
  MK_SET_HIP(s, hipEventElapsedTime(&f, s->ev_join[0], s->ev_join[1]));
  *ms = (double)f;
  return MK_OK;
}

/* primer[LOG2(hashsize * 1.5) - 7] (command_set.c:871-872, global_basic.c:75-82): the table a taxon of `total_ids` ids gets */
extern "C" uint32_t mk_setop_group_table_size(uint64_t total_ids) {
  static const uint32_t primes[25] = {251u, 509u, 1021u, 2039u, 4093u, 8191u, 16381u, 32749u, 65521u, 131071u, 262139u, 524287u,
                                      1048573u, 2097143u, 4194301u, 8388593u, 16777213u, 33554393u, 67108859u, 134217689u,
                                      268435399u, 536870909u, 1073741789u, 2147483647u, 4294967291u};
  const unsigned long long v = (unsigned long long)((double)(int)total_ids * 1.5); /* `int hashsize` there */
  if (v == 0) return primes[0];
  const unsigned ind = 63u - (unsigned)__builtin_clzll(v);
  return ind > 7u ? primes[ind - 7u > 24u ? 24u : ind - 7u] : primes[0];
}

/* result left on the device (ascending ids), for callers that keep working there */
extern "C" int mk_setop_result_device(mk_setop *s, const uint32_t **ids_dev, uint64_t *n) {
  if (!s || !ids_dev || !n) return MK_ERR_ARG;
  if (s->begun) return mk_set_fail(s, MK_ERR_STATE, "mk_setop_result_device before mk_setop_finish");
  *ids_dev = s->d_out;
  *n = s->h_total ? *s->h_total : 0;
  return MK_OK;
}

extern "C" void *mk_setop_stream(mk_setop *s) { return s ? (void *)s->stream : nullptr; }
