/*
 * mk_mco.hip -- stage II (inverted index) and the reference-database search on the device.
 *
 * combco2mco() (co2mco.c:12-87) appends, genome after genome, the genome's number to the row of every k-mer id it holds
 * and writes the rows in id order; mco_cbdco_nobin_dist() (command_dist.c:1033-1049) walks, for every id of a query
 * sketch, the id's row and bumps ct[query][genome].  Both are byte/integer work with no contraction: HBM- and
 * atomic-bound, MFMA does not apply.
 *
 *   mk_mco_gid_kernel       position in combco.N -> genome number (binary search in combco.index.N)
 *   mk_radix_sort_pairs_u32 (mk_sort.hip.h)   (id, genome) by id, STABLE: a row keeps the genome order in which the reference
 *                           appends.  Hand-written LSD radix sort, four passes of 8 bits (round 3; rocPRIM before)
 *   mk_mco_rowcount / rowscan / rowwrite   ordered compaction of the row ends: the non-empty rows and their cumulative ends
 *   mk_mco_index_kernel     a slab of the dense 2^32-entry mco.index.N filled from the row table
 *   mk_mco_extent_kernel    query id -> its row's extent (device row table; the CLI takes extents from the mmap'ed index)
 *   mk_mco_count_kernel     the counting loop: one workgroup per slice of a query sketch, counters in LDS (one per
 *                           reference genome, up to 32 768) flushed once per slice; global atomics above that; the
 *                           genome lists are repacked to 16 bits (mk_mco_pack16_kernel) below 65 535 genomes
 */
#include <hip/hip_runtime.h>
#include "mk_poison.hip.h"

#include "mk_sort.hip.h"

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "metakssd_hip.h"

#define MK_MCO_CHUNK 1024u          /* positions per wave in the compaction passes: 16 per lane */
#define MK_MCO_SLAB_ROWS (1ull << 27) /* dense-index rows per mk_mco_index_rows call (1 GiB) */
#define MK_MCO_LDS_REFS 32768u      /* reference genomes whose counters fit in LDS (128 KiB) */
#define MK_MCO_SLICE 16384u         /* query ids per work item */
#define MK_MCO_STAGE_CELLS (16ull << 20) /* cells of the count matrix per pinned staging piece (64 MiB) */

struct mk_mco_item {
  unsigned long long a; /* first position in the component's query id list */
  uint32_t len, k;      /* ids in the slice; query sketch */
};

struct mk_mco {
  int device = 0, num_cu = 256;
  bool opt_global_counters = false, opt_wide_lists = false; /* mk_mco_set_option */
  hipStream_t stream = nullptr;
  /* build */
  uint32_t *d_key[2] = {nullptr, nullptr}, *d_val[2] = {nullptr, nullptr};
  uint64_t pair_cap = 0;
  void *d_tmp = nullptr;
  size_t tmp_cap = 0;
  uint32_t *h_sort_flag = nullptr; /* pinned: which passes of the radix sort moved anything */
  unsigned long long *d_index = nullptr;
  uint64_t index_cap = 0;
  uint32_t *d_chunk = nullptr;
  unsigned long long *d_chunk_off = nullptr, *d_total = nullptr, *h_total = nullptr;
  uint64_t chunk_cap = 0;
  uint32_t *d_row_ids = nullptr;
  unsigned long long *d_row_ends = nullptr;
  uint64_t row_cap = 0, nrows = 0, n = 0;
  bool built = false;
  uint32_t *h_gids = nullptr, *h_row_ids = nullptr;
  unsigned long long *h_row_ends = nullptr;
  uint64_t h_gid_cap = 0, h_row_cap = 0;
  unsigned long long *d_slab = nullptr;
  /* count */
  uint32_t *d_ct = nullptr;
  uint64_t ct_cap = 0;
  uint32_t ref_num = 0, qry_num = 0;
  bool counting = false;
  uint32_t *d_gids = nullptr;
  uint64_t gids_cap = 0;
  uint16_t *d_gids16 = nullptr;
  uint64_t gids16_cap = 0;
  bool lds_configured16 = false;
  uint32_t *d_qids = nullptr;
  unsigned long long *d_es = nullptr, *d_ee = nullptr;
  uint64_t q_cap = 0;
  mk_mco_item *d_items = nullptr;
  uint64_t item_cap = 0;
  bool lds_configured = false;
  uint32_t *h_stage[2] = {nullptr, nullptr};
  hipEvent_t ev_stage[2] = {nullptr, nullptr}; /* a staging piece has crossed to the device */
  /* mk_mco_last_kernel_ms: events around the last build's radix sort and around the last count_add's kernels */
  hipEvent_t ev_sort[2] = {nullptr, nullptr}, ev_count[2] = {nullptr, nullptr};
  bool sort_timed = false, count_timed = false;
  uint32_t sort_passes = 0;
  char err[256] = {0};
};

static thread_local char mk_mco_create_err[256];

static int mk_mco_fail(mk_mco *m, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(m ? m->err : mk_mco_create_err, 256, fmt, ap);
  va_end(ap);
  return code;
}

#define MK_MCO_HIP(m, call)                                                                          \
  do {                                                                                               \
    hipError_t _r = (call);                                                                          \
    if (_r != hipSuccess) return mk_mco_fail(m, MK_ERR_HIP, "%s: %s", #call, hipGetErrorString(_r)); \
  } while (0)

template <class T>
static int mk_mco_grow(mk_mco *m, T **p, uint64_t *cap, uint64_t need) {
  if (need <= *cap && *p) return MK_OK;
  (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const uint64_t c = need + need / 8 + 1024;
  MK_MCO_HIP(m, mk_dev_alloc((void **)p, c * sizeof(T)));
  *cap = c;
  return MK_OK;
}

/* ---- kernels ------------------------------------------------------------------------------------------ */

/* number of entries of a[0..n) that are <= x */
template <class T, class X>
__device__ __forceinline__ uint64_t mk_mco_upper(const T *a, uint64_t lo, uint64_t hi, X x) {
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((X)a[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* co2mco.c:37-39: position k of the combined file belongs to genome j with index[j] <= k < index[j+1] */
__global__ void __launch_bounds__(256) mk_mco_gid_kernel(const unsigned long long *index, uint32_t cofnum, uint64_t n, uint32_t *gid) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    gid[i] = (uint32_t)(mk_mco_upper(index, 0, (uint64_t)cofnum + 1, (unsigned long long)i) - 1);
}

__device__ __forceinline__ uint32_t mk_mco_ends16(const uint32_t *key, uint64_t i0, uint64_t n) {
  uint32_t flags = 0;
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    const uint64_t i = i0 + k;
    if (i < n && (i + 1 == n || key[i] != key[i + 1])) flags |= 1u << k;
  }
  return flags;
}

__global__ void __launch_bounds__(256) mk_mco_rowcount_kernel(const uint32_t *key, uint64_t n, uint64_t nchunks, uint32_t *chunk_count) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= nchunks) return;
  uint32_t c = __popc(mk_mco_ends16(key, chunk * MK_MCO_CHUNK + 16u * lane, n));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if (lane == 0) chunk_count[chunk] = c;
}

/* exclusive prefix over the chunk counts: one workgroup, each thread a contiguous slice */
__global__ void __launch_bounds__(1024) mk_mco_rowscan_kernel(const uint32_t *count, uint64_t nchunks, unsigned long long *off,
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

/* co2mco.c:59: the cumulative row ends, kept only where a row ends */
__global__ void __launch_bounds__(256) mk_mco_rowwrite_kernel(const uint32_t *key, uint64_t n, uint64_t nchunks, const uint32_t *chunk_count,
                                                              const unsigned long long *chunk_off, uint32_t *row_ids,
                                                              unsigned long long *row_ends) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (chunk >= nchunks || chunk_count[chunk] == 0u) return; /* wave-uniform */
  const uint64_t i0 = chunk * MK_MCO_CHUNK + 16u * lane;
  const uint32_t flags = mk_mco_ends16(key, i0, n);
  const uint32_t mine = __popc(flags);
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += u;
  }
  uint64_t o = chunk_off[chunk] + (incl - mine);
#pragma unroll
  for (uint32_t k = 0; k < 16; k++)
    if ((flags >> k) & 1u) { row_ids[o] = key[i0 + k]; row_ends[o] = i0 + k + 1; o++; }
}

/* co2mco.c:59-66: out[r - row0] = cumulative end of the last non-empty row <= r (0 before the first one).  A block owns
 * MK_MCO_INDEX_ROWS consecutive rows; two block-uniform searches bound the table entries that fall into them.  Most blocks of
 * the 2^32-row index hold none: they stream one constant with 16-byte stores (the write roofline); the others search per row
 * inside the block's few entries. */
#define MK_MCO_INDEX_ROWS 2048u
__global__ void __launch_bounds__(256) mk_mco_index_kernel(const uint32_t *row_ids, const unsigned long long *row_ends, uint64_t nrows_tab,
                                                           uint64_t row0, uint64_t nrows, unsigned long long *out) {
  const uint64_t b0 = (uint64_t)blockIdx.x * MK_MCO_INDEX_ROWS;
  if (b0 >= nrows) return;
  const uint64_t last = b0 + MK_MCO_INDEX_ROWS - 1 < nrows ? b0 + MK_MCO_INDEX_ROWS - 1 : nrows - 1;
  const uint64_t lo = mk_mco_upper(row_ids, 0, nrows_tab, (uint64_t)(row0 + b0));     /* table entries <= first row */
  const uint64_t hi = mk_mco_upper(row_ids, lo, nrows_tab, (uint64_t)(row0 + last));  /* table entries <= last row */
  if (lo == hi && last - b0 + 1 == MK_MCO_INDEX_ROWS) { /* no row of the table starts inside a full block: one value */
    const unsigned long long v = lo ? row_ends[lo - 1] : 0ull;
    ulonglong2 *o = (ulonglong2 *)(out + b0); /* out and b0 are 16-byte aligned */
#pragma unroll
    for (uint32_t k = 0; k < MK_MCO_INDEX_ROWS / 512u; k++) o[threadIdx.x + 256u * k] = make_ulonglong2(v, v);
    return;
  }
  for (uint64_t r = b0 + threadIdx.x; r <= last; r += 256u) {
    const uint64_t u = mk_mco_upper(row_ids, lo, hi, (uint64_t)(row0 + r));
    out[r] = u ? row_ends[u - 1] : 0ull;
  }
}

/* command_dist.c:1040-1041 from the row table instead of the dense index */
__global__ void __launch_bounds__(256) mk_mco_extent_kernel(const uint32_t *qids, uint64_t nq, const uint32_t *row_ids,
                                                            const unsigned long long *row_ends, uint64_t nrows_tab,
                                                            unsigned long long *es, unsigned long long *ee) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t id = qids[i];
    const uint64_t u = mk_mco_upper(row_ids, 0, nrows_tab, (uint64_t)id); /* rows <= id */
    unsigned long long s = 0, e = 0;
    if (u && row_ids[u - 1] == id) { e = row_ends[u - 1]; s = u > 1 ? row_ends[u - 2] : 0ull; }
    es[i] = s; ee[i] = e;
  }
}

/* command_dist.c:1038-1045: for every id of the slice, every genome of its row: ct[query][genome]++ */
/* genome numbers as 16-bit values when the database has fewer than 65 536 genomes: the lists are the bytes this kernel moves */
__global__ void __launch_bounds__(256) mk_mco_pack16_kernel(const uint32_t *in, uint64_t n, uint16_t *out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t v = in[i];
    out[i] = v < 0xFFFFu ? (uint16_t)v : (uint16_t)0xFFFFu; /* >= ref_num: ignored by the counting kernel */
  }
}

template <bool LDS, class G>
__global__ void __launch_bounds__(1024) mk_mco_count_kernel(const mk_mco_item *items, uint32_t nitems, const G *gids,
                                                            const unsigned long long *es, const unsigned long long *ee, uint32_t R,
                                                            uint32_t *ct) {
  extern __shared__ uint32_t mk_mco_acc[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
    const mk_mco_item item = items[it];
    uint32_t *dst = ct + (size_t)item.k * R;
    if (LDS) {
      for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) mk_mco_acc[r] = 0u;
      __syncthreads();
    }
    uint32_t *tgt = LDS ? mk_mco_acc : dst;
    const unsigned long long end = item.a + item.len;
    for (unsigned long long base = item.a + (unsigned long long)wave * 64u; base < end; base += (unsigned long long)nwaves * 64u) {
      const unsigned long long i = base + lane;
      unsigned long long s = 0, e = 0;
      if (i < end) { s = es[i]; e = ee[i]; }
      /* Both walks issue every load of a step before the first increment: the kernel is bound by memory latency, not
       * by bytes or by the atomics.  short row (<= 8 genomes): the lane walks its own row */
      const bool wide = e - s > 8ull;
      if (!wide && e > s) {
        uint32_t r[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) r[u] = s + u < e ? (uint32_t)gids[s + u] : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
          if (r[u] < R) atomicAdd(&tgt[r[u]], 1u);
      }
      /* longer rows: the wave walks them together, 512 genomes per step, and has the first step of the next row in
       * flight while it increments for the current one */
      unsigned long long m = __ballot(wide);
      uint32_t cur[8];
      unsigned long long cs = 0, ce = 0;
      bool have = m != 0ull;
      if (have) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        cs = __shfl(s, l, 64); ce = __shfl(e, l, 64);
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) cur[u] = cs + lane + 64u * u < ce ? (uint32_t)gids[cs + lane + 64u * u] : 0xFFFFFFFFu;
      }
      while (have) {
        uint32_t nxt[8];
        unsigned long long ns = 0, ne = 0;
        const bool more = m != 0ull;
        if (more) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1;
          ns = __shfl(s, l, 64); ne = __shfl(e, l, 64);
#pragma unroll
          for (uint32_t u = 0; u < 8; u++) nxt[u] = ns + lane + 64u * u < ne ? (uint32_t)gids[ns + lane + 64u * u] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
          if (cur[u] < R) atomicAdd(&tgt[cur[u]], 1u);
        for (unsigned long long g = cs + 512u + lane; g - lane < ce; g += 512u) { /* rows beyond 512 genomes */
          uint32_t r[8];
#pragma unroll
          for (uint32_t u = 0; u < 8; u++) r[u] = g + 64u * u < ce ? (uint32_t)gids[g + 64u * u] : 0xFFFFFFFFu;
#pragma unroll
          for (uint32_t u = 0; u < 8; u++)
            if (r[u] < R) atomicAdd(&tgt[r[u]], 1u);
        }
        if (more) {
#pragma unroll
          for (uint32_t u = 0; u < 8; u++) cur[u] = nxt[u];
        }
        cs = ns; ce = ne;
        have = more;
      }
    }
    if (LDS) {
      __syncthreads();
      for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) {
        const uint32_t c = mk_mco_acc[r];
        if (c) atomicAdd(&dst[r], c);
      }
      __syncthreads();
    }
  }
}

/* ---- host side ---------------------------------------------------------------------------------------- */

extern "C" int mk_mco_create(int device, mk_mco **out) {
  if (!out) return MK_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return mk_mco_fail(nullptr, MK_ERR_NO_DEVICE, "no HIP device: mk_mco has no CPU path");
  if (device < 0 || device >= ndev) return mk_mco_fail(nullptr, MK_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
  mk_mco *m = new (std::nothrow) mk_mco();
  if (!m) return MK_ERR_NOMEM;
  m->device = device;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete m;
    return mk_mco_fail(nullptr, MK_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
  }
  m->num_cu = prop.multiProcessorCount;
  hipError_t r = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (r == hipSuccess) r = mk_dev_alloc(&m->d_total, 8);
  if (r == hipSuccess) r = mk_pin_alloc((void **)&m->h_total, 8, hipHostMallocDefault);
  if (r != hipSuccess) {
    mk_mco_fail(nullptr, MK_ERR_NOMEM, "mco allocation: %s", hipGetErrorString(r));
    mk_mco_destroy(m);
    return MK_ERR_NOMEM;
  }
  *out = m;
  return MK_OK;
}

extern "C" int mk_mco_destroy(mk_mco *m) {
  if (!m) return MK_OK;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  for (int b = 0; b < 2; b++) { (void)hipFree(m->d_key[b]); (void)hipFree(m->d_val[b]); }
  (void)hipFree(m->d_tmp); (void)hipFree(m->d_index); (void)hipFree(m->d_chunk); (void)hipFree(m->d_chunk_off);
  (void)hipFree(m->d_total); (void)hipFree(m->d_row_ids); (void)hipFree(m->d_row_ends); (void)hipFree(m->d_slab);
  (void)hipFree(m->d_ct); (void)hipFree(m->d_gids); (void)hipFree(m->d_gids16); (void)hipFree(m->d_qids); (void)hipFree(m->d_es); (void)hipFree(m->d_ee);
  (void)hipFree(m->d_items);
  if (m->h_total) (void)hipHostFree(m->h_total);
  if (m->h_sort_flag) (void)hipHostFree(m->h_sort_flag);
  if (m->h_gids) (void)hipHostFree(m->h_gids);
  if (m->h_row_ids) (void)hipHostFree(m->h_row_ids);
  if (m->h_row_ends) (void)hipHostFree(m->h_row_ends);
  for (int b = 0; b < 2; b++) {
    if (m->h_stage[b]) (void)hipHostFree(m->h_stage[b]);
    if (m->ev_stage[b]) (void)hipEventDestroy(m->ev_stage[b]);
    if (m->ev_sort[b]) (void)hipEventDestroy(m->ev_sort[b]);
    if (m->ev_count[b]) (void)hipEventDestroy(m->ev_count[b]);
  }
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return MK_OK;
}

extern "C" const char *mk_mco_last_error(const mk_mco *m) { return m ? m->err : mk_mco_create_err; }

static unsigned mk_mco_blocks(const mk_mco *m, uint64_t n, unsigned per_block) {
  uint64_t b = (n + per_block - 1) / per_block;
  const uint64_t cap = (uint64_t)m->num_cu * 32u;
  if (b > cap) b = cap;
  return b ? (unsigned)b : 1u;
}

/* Host memory into HBM on m->stream.  Pinned memory (mk_host_alloc, a registered arena) goes as it is.  Pageable memory -- what a
 * caller that read combco.N with fread() holds -- would cross through the runtime's bounce buffer at one core's memcpy rate
 * (about 11 GB/s measured); here MK_MCO_COPY_THREADS threads copy slices of a 64 MiB piece into pinned staging while the piece
 * before it crosses PCIe. */
#define MK_MCO_COPY_THREADS 4u
static int mk_mco_upload(mk_mco *m, void *dst, const void *src, size_t bytes) {
  if (!bytes) return MK_OK;
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
  (void)hipGetLastError();
  const size_t piece = MK_MCO_STAGE_CELLS * 4;
  if (pinned || bytes < piece / 8) {
    MK_MCO_HIP(m, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, m->stream));
    return MK_OK;
  }
  for (int b = 0; b < 2; b++) {
    if (!m->h_stage[b]) MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_stage[b], piece, hipHostMallocDefault));
    if (!m->ev_stage[b]) MK_MCO_HIP(m, hipEventCreateWithFlags(&m->ev_stage[b], hipEventDisableTiming));
  }
  /* (the staging pieces also carry mk_mco_count_finish's matrix: that call is over when it returns) */
  uint32_t p = 0;
  for (size_t off = 0; off < bytes; off += piece, p++) {
    const size_t len = bytes - off < piece ? bytes - off : piece;
    const int b = (int)(p & 1u);
    MK_MCO_HIP(m, hipEventSynchronize(m->ev_stage[b])); /* the piece that went through this buffer before (also an earlier call's) */
    uint8_t *to = (uint8_t *)m->h_stage[b];
    const uint8_t *from = (const uint8_t *)src + off;
    const size_t slice = ((len + MK_MCO_COPY_THREADS - 1) / MK_MCO_COPY_THREADS + 4095) & ~(size_t)4095;
    std::thread th[MK_MCO_COPY_THREADS];
    uint32_t started = 0;
    for (uint32_t t = 1; t < MK_MCO_COPY_THREADS; t++) {
      const size_t lo = slice * t;
      if (lo >= len) break;
      const size_t n = len - lo < slice ? len - lo : slice;
      try { th[t] = std::thread([=] { memcpy(to + lo, from + lo, n); }); started |= 1u << t; }
      catch (...) { memcpy(to + lo, from + lo, n); } /* no thread to be had: this one copies the slice */
    }
    memcpy(to, from, len < slice ? len : slice);
    for (uint32_t t = 1; t < MK_MCO_COPY_THREADS; t++)
      if (started & (1u << t)) th[t].join();
    MK_MCO_HIP(m, hipMemcpyAsync((uint8_t *)dst + off, to, len, hipMemcpyHostToDevice, m->stream));
    MK_MCO_HIP(m, hipEventRecord(m->ev_stage[b], m->stream));
  }
  return MK_OK;
}

extern "C" int mk_mco_build(mk_mco *m, const uint32_t *ids, const uint64_t *index, uint32_t cofnum, const uint32_t **gids, uint64_t *n_out,
                            const uint32_t **row_ids, const uint64_t **row_ends, uint64_t *nrows_out) {
  if (!m || !index || !gids || !n_out || !row_ids || !row_ends || !nrows_out) return MK_ERR_ARG;
  const uint64_t n = index[cofnum];
  if (n && !ids) return MK_ERR_ARG;
  /* the sort keeps its per-(digit, workgroup) prefixes in 32 bits (mk_sort.hip.h): as for mk_mco_sort_pairs */
  if (n >= (1ull << 32)) return mk_mco_fail(m, MK_ERR_ARG, "mk_mco_build: a component of %llu ids (fewer than 2^32 are supported)", (unsigned long long)n);
  for (uint32_t j = 0; j < cofnum; j++)
    if (index[j] > index[j + 1]) return mk_mco_fail(m, MK_ERR_ARG, "combco.index not ascending at sketch %u", j);
  MK_MCO_HIP(m, hipSetDevice(m->device));
  m->built = false;
  m->n = n; m->nrows = 0;
  int rc;
  /* the result buffers exist even for an empty component */
  if (n > m->h_gid_cap || !m->h_gids) {
    if (m->h_gids) (void)hipHostFree(m->h_gids);
    m->h_gids = nullptr; m->h_gid_cap = 0;
    const uint64_t c = n + n / 8 + 1024;
    MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_gids, c * 4, hipHostMallocDefault));
    m->h_gid_cap = c;
  }
  uint64_t total = 0;
  if (n) {
    uint64_t cap = m->pair_cap;
    for (int b = 0; b < 2; b++) {
      cap = m->pair_cap;
      if ((rc = mk_mco_grow(m, &m->d_key[b], &cap, n))) return rc;
      cap = m->pair_cap;
      if ((rc = mk_mco_grow(m, &m->d_val[b], &cap, n))) return rc;
    }
    m->pair_cap = cap;
    if ((rc = mk_mco_grow(m, &m->d_index, &m->index_cap, (uint64_t)cofnum + 1))) return rc;
    MK_MCO_HIP(m, hipMemcpyAsync(m->d_index, index, ((size_t)cofnum + 1) * 8, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(mk_mco_gid_kernel, dim3(mk_mco_blocks(m, n, 256)), dim3(256), 0, m->stream, m->d_index, cofnum, n, m->d_val[0]);
    MK_MCO_HIP(m, hipGetLastError());
    if ((rc = mk_mco_upload(m, m->d_key[0], ids, n * 4))) return rc; /* (after the launch: the gid kernel runs under the copy) */
    /* stable sort by id: (key, value) ping-pong between the two buffer pairs; the sorted pair ends up as [1] */
    {
      const size_t tmp_bytes = (size_t)256 * MK_RS_MAXB * 4 + 256 * 8 + 64;
      if (tmp_bytes > m->tmp_cap || !m->d_tmp) {
        (void)hipFree(m->d_tmp);
        m->d_tmp = nullptr; m->tmp_cap = 0;
        MK_MCO_HIP(m, mk_dev_alloc(&m->d_tmp, tmp_bytes));
        m->tmp_cap = tmp_bytes;
      }
      if (!m->h_sort_flag) MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_sort_flag, 4 * sizeof(uint32_t), hipHostMallocDefault));
      uint32_t *hist = (uint32_t *)m->d_tmp;
      unsigned long long *tot = (unsigned long long *)((uint8_t *)m->d_tmp + (size_t)256 * MK_RS_MAXB * 4);
      uint32_t *flag = (uint32_t *)((uint8_t *)m->d_tmp + (size_t)256 * MK_RS_MAXB * 4 + 256 * 8);
      int where = 0;
      for (int b = 0; b < 2; b++) if (!m->ev_sort[b]) MK_MCO_HIP(m, hipEventCreate(&m->ev_sort[b]));
      m->sort_timed = false;
      MK_MCO_HIP(m, hipEventRecord(m->ev_sort[0], m->stream));
      MK_MCO_HIP(m, mk_radix_sort_pairs_u32(m->d_key, m->d_val, n, m->num_cu, hist, tot, flag, m->h_sort_flag, m->stream, &where));
      MK_MCO_HIP(m, hipEventRecord(m->ev_sort[1], m->stream));
      m->sort_timed = true;
      if (where == 0) { /* the passes that moved anything were even in number: the result sits in pair 0 */
        uint32_t *tk = m->d_key[0]; m->d_key[0] = m->d_key[1]; m->d_key[1] = tk;
        uint32_t *tv = m->d_val[0]; m->d_val[0] = m->d_val[1]; m->d_val[1] = tv;
      }
    }
    /* row ends */
    const uint64_t nchunks = (n + MK_MCO_CHUNK - 1) / MK_MCO_CHUNK;
    cap = m->chunk_cap;
    if ((rc = mk_mco_grow(m, &m->d_chunk, &cap, nchunks))) return rc;
    if ((rc = mk_mco_grow(m, &m->d_chunk_off, &m->chunk_cap, nchunks))) return rc;
    const unsigned cblocks = (unsigned)((nchunks + 3) / 4);
    hipLaunchKernelGGL(mk_mco_rowcount_kernel, dim3(cblocks), dim3(256), 0, m->stream, m->d_key[1], n, nchunks, m->d_chunk);
    hipLaunchKernelGGL(mk_mco_rowscan_kernel, dim3(1), dim3(1024), 0, m->stream, m->d_chunk, nchunks, m->d_chunk_off, m->d_total);
    MK_MCO_HIP(m, hipGetLastError());
    MK_MCO_HIP(m, hipMemcpyAsync(m->h_total, m->d_total, 8, hipMemcpyDeviceToHost, m->stream));
    MK_MCO_HIP(m, hipStreamSynchronize(m->stream));
    total = *m->h_total;
    cap = m->row_cap;
    if ((rc = mk_mco_grow(m, &m->d_row_ids, &cap, total))) return rc;
    if ((rc = mk_mco_grow(m, &m->d_row_ends, &m->row_cap, total))) return rc;
    hipLaunchKernelGGL(mk_mco_rowwrite_kernel, dim3(cblocks), dim3(256), 0, m->stream, m->d_key[1], n, nchunks, m->d_chunk,
                       m->d_chunk_off, m->d_row_ids, m->d_row_ends);
    MK_MCO_HIP(m, hipGetLastError());
  }
  if (total > m->h_row_cap || !m->h_row_ids) {
    if (m->h_row_ids) (void)hipHostFree(m->h_row_ids);
    if (m->h_row_ends) (void)hipHostFree(m->h_row_ends);
    m->h_row_ids = nullptr; m->h_row_ends = nullptr; m->h_row_cap = 0;
    const uint64_t c = total + total / 8 + 1024;
    MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_row_ids, c * 4, hipHostMallocDefault));
    MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_row_ends, c * 8, hipHostMallocDefault));
    m->h_row_cap = c;
  }
  if (n) {
    MK_MCO_HIP(m, hipMemcpyAsync(m->h_gids, m->d_val[1], n * 4, hipMemcpyDeviceToHost, m->stream));
    MK_MCO_HIP(m, hipMemcpyAsync(m->h_row_ids, m->d_row_ids, total * 4, hipMemcpyDeviceToHost, m->stream));
    MK_MCO_HIP(m, hipMemcpyAsync(m->h_row_ends, m->d_row_ends, total * 8, hipMemcpyDeviceToHost, m->stream));
    MK_MCO_HIP(m, hipStreamSynchronize(m->stream));
  }
  m->nrows = total;
  m->built = true;
  *gids = m->h_gids; *n_out = n;
  *row_ids = m->h_row_ids; *row_ends = (const uint64_t *)m->h_row_ends; *nrows_out = total;
  return MK_OK;
}

/* the sort of mk_mco_build on its own: n (key, value) pairs in host arrays, sorted by key in place, equal keys in input order */
extern "C" int mk_mco_sort_pairs(mk_mco *m, uint32_t *keys, uint32_t *vals, uint64_t n) {
  if (!m || (n && (!keys || !vals))) return MK_ERR_ARG;
  if (n >= (1ull << 32)) return mk_mco_fail(m, MK_ERR_ARG, "mk_mco_sort_pairs: fewer than 2^32 pairs");
  if (n == 0) return MK_OK;
  MK_MCO_HIP(m, hipSetDevice(m->device));
  int rc;
  uint64_t cap = m->pair_cap;
  for (int b = 0; b < 2; b++) {
    cap = m->pair_cap;
    if ((rc = mk_mco_grow(m, &m->d_key[b], &cap, n))) return rc;
    cap = m->pair_cap;
    if ((rc = mk_mco_grow(m, &m->d_val[b], &cap, n))) return rc;
  }
  m->pair_cap = cap;
  m->built = false;
  const size_t tmp_bytes = (size_t)256 * MK_RS_MAXB * 4 + 256 * 8 + 64;
  if (tmp_bytes > m->tmp_cap || !m->d_tmp) {
    (void)hipFree(m->d_tmp);
    m->d_tmp = nullptr; m->tmp_cap = 0;
    MK_MCO_HIP(m, mk_dev_alloc(&m->d_tmp, tmp_bytes));
    m->tmp_cap = tmp_bytes;
  }
  if (!m->h_sort_flag) MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_sort_flag, 4 * sizeof(uint32_t), hipHostMallocDefault));
  if ((rc = mk_mco_upload(m, m->d_key[0], keys, n * 4)) || (rc = mk_mco_upload(m, m->d_val[0], vals, n * 4))) return rc;
  uint32_t *hist = (uint32_t *)m->d_tmp;
  unsigned long long *tot = (unsigned long long *)((uint8_t *)m->d_tmp + (size_t)256 * MK_RS_MAXB * 4);
  uint32_t *flag = (uint32_t *)((uint8_t *)m->d_tmp + (size_t)256 * MK_RS_MAXB * 4 + 256 * 8);
  int where = 0;
  MK_MCO_HIP(m, mk_radix_sort_pairs_u32(m->d_key, m->d_val, n, m->num_cu, hist, tot, flag, m->h_sort_flag, m->stream, &where));
  MK_MCO_HIP(m, hipMemcpyAsync(keys, m->d_key[where], n * 4, hipMemcpyDeviceToHost, m->stream));
  MK_MCO_HIP(m, hipMemcpyAsync(vals, m->d_val[where], n * 4, hipMemcpyDeviceToHost, m->stream));
  MK_MCO_HIP(m, hipStreamSynchronize(m->stream));
  return MK_OK;
}

extern "C" int mk_mco_index_rows(mk_mco *m, uint64_t row0, uint64_t nrows, uint64_t *out) {
  if (!m || (nrows && !out)) return MK_ERR_ARG;
  if (!m->built) return mk_mco_fail(m, MK_ERR_STATE, "mk_mco_index_rows before mk_mco_build");
  if (nrows > MK_MCO_SLAB_ROWS || row0 + nrows > (1ull << 32)) return mk_mco_fail(m, MK_ERR_ARG, "row range out of bounds");
  if (nrows == 0) return MK_OK;
  MK_MCO_HIP(m, hipSetDevice(m->device));
  if (!m->d_slab) MK_MCO_HIP(m, mk_dev_alloc((void **)&m->d_slab, MK_MCO_SLAB_ROWS * 8));
  hipLaunchKernelGGL(mk_mco_index_kernel, dim3((unsigned)((nrows + MK_MCO_INDEX_ROWS - 1) / MK_MCO_INDEX_ROWS)), dim3(256), 0, m->stream, m->d_row_ids, m->d_row_ends,
                     m->nrows, row0, nrows, m->d_slab);
  MK_MCO_HIP(m, hipGetLastError());
  MK_MCO_HIP(m, hipMemcpyAsync(out, m->d_slab, nrows * 8, hipMemcpyDeviceToHost, m->stream));
  MK_MCO_HIP(m, hipStreamSynchronize(m->stream));
  return MK_OK;
}

extern "C" int mk_mco_set_option(mk_mco *m, int option, int64_t value) {
  if (!m) return MK_ERR_ARG;
  switch (option) {
    case MK_MCO_OPT_GLOBAL_COUNTERS: m->opt_global_counters = value != 0; return MK_OK;
    case MK_MCO_OPT_WIDE_LISTS: m->opt_wide_lists = value != 0; return MK_OK;
    default: return mk_mco_fail(m, MK_ERR_ARG, "unknown mk_mco option %d", option);
  }
}

extern "C" int mk_mco_count_begin(mk_mco *m, uint32_t ref_num, uint32_t qry_num) {
  if (!m) return MK_ERR_ARG;
  MK_MCO_HIP(m, hipSetDevice(m->device));
  const uint64_t cells = (uint64_t)ref_num * qry_num;
  int rc = mk_mco_grow(m, &m->d_ct, &m->ct_cap, cells ? cells : 1);
  if (rc) return rc;
  MK_MCO_HIP(m, hipMemsetAsync(m->d_ct, 0, (cells ? cells : 1) * 4, m->stream));
  m->ref_num = ref_num; m->qry_num = qry_num;
  m->counting = true;
  return MK_OK;
}

extern "C" int mk_mco_count_add(mk_mco *m, const uint32_t *gids, uint64_t ngids, const uint32_t *qry_ids, const uint64_t *ext_start,
                                const uint64_t *ext_end, const uint64_t *qry_index, const uint32_t *qry_ctx_ct) {
  if (!m || !qry_index || !qry_ctx_ct) return MK_ERR_ARG;
  if (!m->counting) return mk_mco_fail(m, MK_ERR_STATE, "mk_mco_count_add before mk_mco_count_begin");
  if ((ext_start == nullptr) != (ext_end == nullptr)) return MK_ERR_ARG;
  if (!gids && !m->built) return mk_mco_fail(m, MK_ERR_STATE, "no gid lists: pass them or call mk_mco_build first");
  if (!ext_start && !m->built) return mk_mco_fail(m, MK_ERR_STATE, "no row table: pass extents or call mk_mco_build first");
  const uint64_t nq = qry_index[m->qry_num];
  if (nq == 0 || m->ref_num == 0) return MK_OK;
  if (!ext_start && !qry_ids) return MK_ERR_ARG;
  MK_MCO_HIP(m, hipSetDevice(m->device));
  int rc;
  const uint32_t *d_lists = m->d_val[1];
  uint64_t nlists = m->n;
  if (gids) {
    if ((rc = mk_mco_grow(m, &m->d_gids, &m->gids_cap, ngids ? ngids : 1))) return rc;
    MK_MCO_HIP(m, hipMemcpyAsync(m->d_gids, gids, ngids * 4, hipMemcpyHostToDevice, m->stream));
    d_lists = m->d_gids;
    nlists = ngids;
  }
  uint64_t cap = m->q_cap;
  if ((rc = mk_mco_grow(m, &m->d_qids, &cap, nq))) return rc;
  cap = m->q_cap;
  if ((rc = mk_mco_grow(m, &m->d_es, &cap, nq))) return rc;
  if ((rc = mk_mco_grow(m, &m->d_ee, &m->q_cap, nq))) return rc;
  if (ext_start) {
    for (uint64_t i = 0; i < nq; i++) /* a corrupt index must not send the kernel out of the lists */
      if (ext_start[i] > ext_end[i] || ext_end[i] > nlists) return mk_mco_fail(m, MK_ERR_ARG, "row extent %llu out of the gid lists", (unsigned long long)i);
    MK_MCO_HIP(m, hipMemcpyAsync(m->d_es, ext_start, nq * 8, hipMemcpyHostToDevice, m->stream));
    MK_MCO_HIP(m, hipMemcpyAsync(m->d_ee, ext_end, nq * 8, hipMemcpyHostToDevice, m->stream));
  } else {
    MK_MCO_HIP(m, hipMemcpyAsync(m->d_qids, qry_ids, nq * 4, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(mk_mco_extent_kernel, dim3(mk_mco_blocks(m, nq, 256)), dim3(256), 0, m->stream, m->d_qids, nq, m->d_row_ids,
                       m->d_row_ends, m->nrows, m->d_es, m->d_ee);
    MK_MCO_HIP(m, hipGetLastError());
  }
  /* work items: slices of the sketches that count (command_dist.c:1035) */
  std::vector<mk_mco_item> items;
  for (uint32_t k = 0; k < m->qry_num; k++) {
    if (qry_index[k] > qry_index[k + 1] || qry_index[k + 1] > nq) return mk_mco_fail(m, MK_ERR_ARG, "query index not ascending at sketch %u", k);
    if (qry_ctx_ct[k] == 0) continue;
    for (uint64_t a = qry_index[k]; a < qry_index[k + 1]; a += MK_MCO_SLICE) {
      const uint64_t len = qry_index[k + 1] - a < MK_MCO_SLICE ? qry_index[k + 1] - a : MK_MCO_SLICE;
      items.push_back(mk_mco_item{a, (uint32_t)len, k});
    }
  }
  if (items.empty()) return MK_OK;
  if (items.size() > 0xFFFFFFFFull) return mk_mco_fail(m, MK_ERR_ARG, "too many query slices");
  if ((rc = mk_mco_grow(m, &m->d_items, &m->item_cap, items.size()))) return rc;
  MK_MCO_HIP(m, hipMemcpyAsync(m->d_items, items.data(), items.size() * sizeof(mk_mco_item), hipMemcpyHostToDevice, m->stream));
  MK_MCO_HIP(m, hipStreamSynchronize(m->stream)); /* `items` leaves scope */
  const uint32_t nitems = (uint32_t)items.size(), R = m->ref_num;
  /* LDS counters pay when a slice brings more increments than the R-counter zero + flush costs */
  const bool lds = R <= MK_MCO_LDS_REFS && nq / nitems >= R / 16u && !m->opt_global_counters;
  const bool narrow = R <= 0xFFFFu && nlists > 0 && !m->opt_wide_lists;
  for (int b = 0; b < 2; b++) if (!m->ev_count[b]) MK_MCO_HIP(m, hipEventCreate(&m->ev_count[b]));
  m->count_timed = false;
  MK_MCO_HIP(m, hipEventRecord(m->ev_count[0], m->stream));
  if (narrow) {
    if ((rc = mk_mco_grow(m, &m->d_gids16, &m->gids16_cap, nlists))) return rc;
    hipLaunchKernelGGL(mk_mco_pack16_kernel, dim3(mk_mco_blocks(m, nlists, 256)), dim3(256), 0, m->stream, d_lists, nlists, m->d_gids16);
  }
  if (lds) {
    bool &configured = narrow ? m->lds_configured16 : m->lds_configured;
    if (!configured) {
      const void *fn = narrow ? (const void *)mk_mco_count_kernel<true, uint16_t> : (const void *)mk_mco_count_kernel<true, uint32_t>;
      MK_MCO_HIP(m, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MK_MCO_LDS_REFS * 4u)));
      configured = true;
    }
    const unsigned grid = nitems < (unsigned)m->num_cu * 2u ? nitems : (unsigned)m->num_cu * 2u;
    if (narrow)
      hipLaunchKernelGGL((mk_mco_count_kernel<true, uint16_t>), dim3(grid), dim3(1024), (size_t)R * 4u, m->stream, m->d_items, nitems,
                         (const uint16_t *)m->d_gids16, m->d_es, m->d_ee, R, m->d_ct);
    else
      hipLaunchKernelGGL((mk_mco_count_kernel<true, uint32_t>), dim3(grid), dim3(1024), (size_t)R * 4u, m->stream, m->d_items, nitems, d_lists,
                         m->d_es, m->d_ee, R, m->d_ct);
  } else {
    const unsigned grid = nitems < (unsigned)m->num_cu * 8u ? nitems : (unsigned)m->num_cu * 8u;
    if (narrow)
      hipLaunchKernelGGL((mk_mco_count_kernel<false, uint16_t>), dim3(grid), dim3(1024), 0, m->stream, m->d_items, nitems,
                         (const uint16_t *)m->d_gids16, m->d_es, m->d_ee, R, m->d_ct);
    else
      hipLaunchKernelGGL((mk_mco_count_kernel<false, uint32_t>), dim3(grid), dim3(1024), 0, m->stream, m->d_items, nitems, d_lists, m->d_es,
                         m->d_ee, R, m->d_ct);
  }
  MK_MCO_HIP(m, hipGetLastError());
  MK_MCO_HIP(m, hipEventRecord(m->ev_count[1], m->stream));
  m->count_timed = true;
  MK_MCO_HIP(m, hipStreamSynchronize(m->stream)); /* the caller's buffers are free again */
  return MK_OK;
}

/* device time of the last mk_mco_build's radix sort (all its passes) and of the last mk_mco_count_add's kernels (16-bit packing of the
 * lists where it applies + the counting kernel), from HIP events on the handle's stream; 0 for a part that has not run */
extern "C" int mk_mco_last_kernel_ms(mk_mco *m, double *sort_ms, double *count_ms) {
  if (!m || !sort_ms || !count_ms) return MK_ERR_ARG;
  *sort_ms = *count_ms = 0.0;
  MK_MCO_HIP(m, hipSetDevice(m->device));
  float f = 0.f;
  if (m->sort_timed) { MK_MCO_HIP(m, hipEventSynchronize(m->ev_sort[1])); MK_MCO_HIP(m, hipEventElapsedTime(&f, m->ev_sort[0], m->ev_sort[1])); *sort_ms = (double)f; }
  if (m->count_timed) { MK_MCO_HIP(m, hipEventSynchronize(m->ev_count[1])); MK_MCO_HIP(m, hipEventElapsedTime(&f, m->ev_count[0], m->ev_count[1])); *count_ms = (double)f; }
  return MK_OK;
}

extern "C" int mk_mco_count_finish(mk_mco *m, uint32_t *ct) {
  if (!m || !ct) return MK_ERR_ARG;
  if (!m->counting) return mk_mco_fail(m, MK_ERR_STATE, "mk_mco_count_finish before mk_mco_count_begin");
  MK_MCO_HIP(m, hipSetDevice(m->device));
  const uint64_t cells = (uint64_t)m->ref_num * m->qry_num;
  m->counting = false;
  if (cells == 0) return MK_OK;
  /* the matrix comes back through two pinned pieces: the copy of piece i+1 runs while piece i is added into ct */
  const uint64_t piece = MK_MCO_STAGE_CELLS;
  if (!m->h_stage[0]) {
    MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_stage[0], piece * 4, hipHostMallocDefault));
    MK_MCO_HIP(m, mk_pin_alloc((void **)&m->h_stage[1], piece * 4, hipHostMallocDefault));
  }
  const uint64_t npieces = (cells + piece - 1) / piece;
  MK_MCO_HIP(m, hipMemcpyAsync(m->h_stage[0], m->d_ct, (cells < piece ? cells : piece) * 4, hipMemcpyDeviceToHost, m->stream));
  for (uint64_t p = 0; p < npieces; p++) {
    MK_MCO_HIP(m, hipStreamSynchronize(m->stream)); /* piece p has arrived */
    if (p + 1 < npieces) {
      const uint64_t off = (p + 1) * piece, len = cells - off < piece ? cells - off : piece;
      MK_MCO_HIP(m, hipMemcpyAsync(m->h_stage[(p + 1) & 1], m->d_ct + off, len * 4, hipMemcpyDeviceToHost, m->stream));
    }
    const uint64_t off = p * piece, len = cells - off < piece ? cells - off : piece;
    const uint32_t *src = m->h_stage[p & 1];
    uint32_t *dst = ct + off;
    for (uint64_t i = 0; i < len; i++) dst[i] += src[i];
  }
  return MK_OK;
}
