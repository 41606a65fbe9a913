/*
 * mk_engine.hip -- C-ABI implementation of the MI355X sketch engine (see include/metakssd_hip.h).
 *
 * One engine owns, on one GPU: the .shuf table and the accepted-subspace list, the accumulation table
 * (hashsize slots: key, first ordinal, count), the layout table (hashsize x u32), the distinct-key
 * list and the output arrays.  There is no CPU implementation behind these entry points.
 */
#include "metakssd_hip.h"
#include "mk_poison.hip.h"
#include "mk_kernels.hip.h"
#include "mk_stream.hip.h"
#include "mk_batch.hip.h"
#include "mk_packed.hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

static thread_local char g_create_error[512] = "";

struct mk_evpair { hipEvent_t a, b; };
#define MK_TICKETS 16
#define MK_REGIONS 3
#define MK_REGION_BYTES ((size_t)256 << 20)

struct mk_engine {
  int device = 0;
  int num_cu = 0;
  mk_params P{};
  mk_keyparams kp{};
  char err[512] = "";
  hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr;
  /* MK_OPT_SPLIT_CUS: the scan kernel on a queue of its own whose CU mask leaves split_cus compute units free, everything else
   * (resolve, compaction, clears) on a queue confined to those -- so that what follows one engine's scan runs BESIDE another
   * engine's scan instead of behind it */
  hipStream_t scan_stream = nullptr, split_stream = nullptr;
  hipEvent_t ev_scan_pre = nullptr, ev_scan_post = nullptr;
  int split_cus = 0, scan_cus = 0;
  bool tail = false, tail_active = false; /* MK_BEGIN_NOTHING_FOLLOWS: this sketch's work behind its first scan runs on own_stream (the whole
                                           * device) instead of split_stream; tail_active: e->stream has been switched, the next begin switches back */
  bool scan_shared = false; /* scan_stream is another engine's (mk_engine_share_scan_queue): not destroyed here */
  mk_engine *scan_owner = nullptr; /* ... that engine */
  int scan_lent = 0;               /* engines that borrow THIS engine's scan queue: it stays while they do */

  int32_t *d_shuf = nullptr;
  uint32_t *d_accept = nullptr;
  uint32_t *d_accept_bits = nullptr;
  uint32_t n_accept = 0;
  uint32_t bm_bits = 0;
  /* candidate append buffers: one slot per scan wave */
  uint4 *d_cand = nullptr; /* 16-byte candidate records (mk_scan_args::cand) */
  uint32_t *d_cand_count = nullptr;
  uint32_t cand_slots = 0, cand_cap = 0;

  void *d_tab = nullptr; /* kc[S] | ordinv[S] */
  size_t tab_bytes = 0;
  /* sparse bookkeeping (large tables): dirty-block bitmaps of the accumulation and the layout table, the lists made of
   * them, and whether both tables are known to be all-empty outside the marked blocks */
  bool sparse = false, tables_tracked = false;
  uint32_t *d_dirty_acc = nullptr, *d_dirty_slot = nullptr, *d_list_acc = nullptr, *d_list_slot = nullptr, *d_nlist = nullptr;
  uint32_t acc_words = 0, slot_words = 0, acc_blocks = 0;
  mk_table tab{};
  uint32_t *d_slot = nullptr;
  mk_dist dist{};
  uint32_t *d_chunk = nullptr; /* [component][chunk] */
  uint32_t nchunks = 0;
  unsigned long long *d_comp_totals = nullptr, *h_comp_totals = nullptr; /* [component] */
  unsigned long long *d_counters = nullptr; /* [0]=distinct, [1]=dump total, [2..3]=err flags (as u32), [4..5]=mk_table::front (as u32[4]) */
  void *d_front = nullptr;          /* front table: kc1[front_slots], ordinv1[front_slots] (mk_table, "Front table") */
  mk_front *d_front_desc = nullptr; /* its descriptor in device memory (mk_table::fr) */
  mk_front front{};                 /* host copy */
  uint64_t front_slots = 0;
  int front_bits_opt = -1;          /* MK_OPT_FRONT_BITS */
  bool big_maybe_dirty = true;      /* the S-slot table may hold keys: begin clears it (false: known to be all zero) */
  unsigned long long *h_counters = nullptr; /* pinned mirror */
  bool tables_ready = false; /* the hashsize-slot tables, the key list and the dump's arrays exist (mk_tables_alloc: at creation, or -- MK_ENGINE_LAZY_TABLES -- at the first mk_sketch_begin) */
  bool init_queued = true; /* mk_engine_create's uploads and kernels may still be running on own_stream (it does not wait for them) */
  mk_accept_pair *d_pairs = nullptr;
  /* result arrays: pinned host memory that the dump kernels write directly (it is mapped into the device's address
   * space), so that a finish needs one host synchronisation and no separate result copy */
  uint32_t *h_ids = nullptr;
  uint16_t *h_cnt = nullptr;
  uint64_t h_cap = 0;
  std::vector<mk_component> comps;

  /* host-push staging: MK_REGIONS device regions filled by the copy stream; consecutive pushes (same stride, consecutive
   * ordinals) land back to back in the open region and are scanned with ONE launch when it is full or at the next
   * flush point -- a scan launch per 4 MiB push would cost more than copying it (launch + filter build + event hops) */
  uint8_t *d_stage[MK_REGIONS] = {};
  size_t stage_bytes = 0;
  hipEvent_t ev_copied[MK_REGIONS] = {}, ev_scanned[MK_REGIONS] = {};
  bool stage_two_streams[MK_REGIONS] = {}; /* the last scan of the region ran on another stream than the copies: ev_scanned is live */
  int stage_cur = 0;
  bool direct_host = false; /* MK_OPT_DIRECT_HOST */
  bool region_open = false;
  size_t region_fill = 0;
  uint32_t region_stride = 0;
  uint64_t region_first_ord = 0, region_rows = 0;
  /* asynchronous pushes: ticket t is done when ev_ticket[t % MK_TICKETS] (recorded behind its last copy) has fired */
  hipEvent_t ev_ticket[MK_TICKETS] = {};
  uint64_t tickets_issued = 0;
  /* dynamic-LDS limit already granted to each scan-kernel instantiation on this engine's device */
  std::vector<std::pair<const void *, size_t>> lds_granted;

  /* mk_sketch_finish_begin / _end: the dump goes to staging arrays in HBM and from there to the pinned result arrays by DMA on a
   * stream of its own, beside the next sketch's kernels */
  uint32_t *d_res_ids = nullptr;
  uint16_t *d_res_cnt = nullptr;
  uint64_t res_cap = 0;
  hipStream_t res_stream = nullptr;
  hipEvent_t ev_res = nullptr;
  bool res_koc = false;
  uint64_t res_D = 0;
  bool res_pending = false, res_side = false; /* res_side: layout and dump of the outstanding result run on res_stream */
  uint64_t res_total = 0;
  unsigned long long *d_snap = nullptr, *h_snap = nullptr; /* key count and flags of the sketch being finished on the side stream */
  std::vector<mk_evpair> ev_finish_side;
  /* key-list-driven dump (sparse bookkeeping): scratch sized to the number of keys, device-side result staging */
  void *d_kl = nullptr;          /* skey[cap] | tkey[cap] | tidx[cap] | out_ids[cap] | out_cnt[cap] */
  uint64_t kl_cap = 0;
  uint32_t *d_kl_buckets = nullptr; /* bcount[nb + 1] | bcursor[nb] */
  uint64_t kl_bucket_cap = 0;
  bool slot_clean = false;       /* the layout table is all-empty (mk_kl_find_kernel hands every slot back) */
  /* mk_sketch_push_stream: raw FASTA text -> base stream on the device (mk_stream.hip.h) */
  uint8_t *d_text = nullptr, *d_stream = nullptr, *d_stream_tmp = nullptr;
  size_t text_cap = 0, stream_cap = 0;
  mk_fa_sum *d_fa_sum = nullptr;
  size_t fa_sum_cap = 0;
  mk_fa_state *d_fa_state = nullptr, *h_fa_state = nullptr; /* device state + pinned mirror */
  unsigned long long *d_split = nullptr; /* mk_partial_export_split: [0..16) part totals, [16..32) cursors */
  bool counter0_used = true; /* d_counters[0] may be non-zero: a compaction has run since the last mk_sketch_begin */
  uint64_t fa_tail = 0;      /* stream bytes carried from the last non-final push (exact: read back) */
  uint64_t fa_rows_done = 0; /* virtual rows scanned so far in this sketch = ordinal of the next one */
  bool fa_used = false, fa_final = false;
  uint32_t fa_pitch = 0;     /* this sketch's row step (0: not chosen yet) */

  /* batches of small inputs (mk_sketch_batch_begin / _end): three contexts, two batches in flight, the third holds the results
   * handed out last */
  struct mk_bctx *bctx[3] = {nullptr, nullptr, nullptr};
  uint64_t batch_begun = 0, batch_ended = 0;
  int batch_tb_opt = 0;                /* MK_OPT_BATCH_TAB_BITS: 0 = by the largest file of the batch */
  const mk_batch_dev *cur_batch = nullptr; /* set around the scan launches of a batch */

  int mode = -1;
  uint32_t min_occ = 1; /* MK_MODE_OCC_SET: dump keys seen at least this often */
  bool begun = false, compacted = false;
  bool count_queued = false; /* mk_partial_count_begin: compaction + counter copy are on the stream, not waited for yet */
  uint64_t D = 0;

  /* launch tuning (fixed in the shipped library; a -DMK_TUNING build reads MK_SCAN_THREADS / MK_SCAN_CB / ..) */
  int tune_threads = 1024;
  bool tune_onepass = true;
  uint32_t tune_cb = MK_MAX_CB;

  bool profiling = false;
  std::vector<mk_evpair> ev_scan, ev_resolve, ev_clear, ev_finish, ev_pool;
  uint64_t prof_rows = 0, prof_bytes = 0;
};

static int mk_fail(mk_engine *e, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) snprintf(e->err, sizeof e->err, "%s", buf);
  else snprintf(g_create_error, sizeof g_create_error, "%s", buf);
  return code;
}

#define MK_HIP(e, call)                                                                                   \
  do {                                                                                                    \
    hipError_t _r = (call);                                                                               \
    if (_r != hipSuccess) return mk_fail((e), _r == hipErrorOutOfMemory ? MK_ERR_NOMEM : MK_ERR_HIP,      \
                                         "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_r)); \
  } while (0)

extern "C" const char *mk_last_error(const mk_engine *e) { return e ? e->err : g_create_error; }

extern "C" int mk_device_count(int *n) {
  if (!n) return MK_ERR_ARG;
  int c = 0;
  hipError_t r = hipGetDeviceCount(&c);
  if (r != hipSuccess) { *n = 0; return mk_fail(nullptr, MK_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(r)); }
  *n = c;
  return MK_OK;
}

extern "C" int mk_host_alloc(void **p, size_t bytes) {
  if (!p) return MK_ERR_ARG;
  hipError_t r = mk_pin_alloc(p, bytes, hipHostMallocDefault);
  return r == hipSuccess ? MK_OK : mk_fail(nullptr, MK_ERR_NOMEM, "mk_pin_alloc(%zu): %s", bytes, hipGetErrorString(r));
}
extern "C" int mk_host_free(void *p) { return hipHostFree(p) == hipSuccess ? MK_OK : MK_ERR_HIP; }
/* A large pinned block made the cheap way: anonymous mapping (huge pages where the kernel gives them), its pages touched by
 * several threads at once, then ONE hipHostRegister.  hipHostMalloc pins at 0.2 ms per MiB on one thread (45 ms for the
 * FASTQ stream's 230 MiB of row buffers, as long as creating the engine); this takes a quarter of that. */
#include <pthread.h>
#include <sys/mman.h>
struct mk_touch_job { uint8_t *p; size_t n; pthread_t th; };
static void *mk_touch_run(void *arg) {
  mk_touch_job *j = (mk_touch_job *)arg;
  const int pz = mk_poison_byte();
  if (pz >= 0) memset(j->p, pz, j->n); /* MK_POISON: the whole block, not just a touch of every page */
  else for (size_t off = 0; off < j->n; off += 4096) j->p[off] = 0;
  return nullptr;
}
extern "C" int mk_host_arena_alloc(void **out, size_t bytes) {
  if (!out || !bytes) return MK_ERR_ARG;
  const size_t huge = (size_t)2 << 20;
  const size_t len = (bytes + huge - 1) & ~(huge - 1);
  void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return mk_fail(nullptr, MK_ERR_NOMEM, "mmap(%zu) failed", len);
#ifdef MADV_HUGEPAGE
  (void)madvise(m, len, MADV_HUGEPAGE);
#endif
  enum { T = 8 };
  mk_touch_job job[T];
  int started = 0;
  for (int t = 0; t < T; t++) {
    const size_t lo = len / T * (size_t)t, hi = t + 1 == T ? len : len / T * (size_t)(t + 1);
    job[t].p = (uint8_t *)m + lo; job[t].n = hi - lo;
    if (t + 1 < T && pthread_create(&job[t].th, nullptr, mk_touch_run, &job[t]) == 0) started |= 1 << t;
    else mk_touch_run(&job[t]);
  }
  for (int t = 0; t < T; t++) if (started & (1 << t)) pthread_join(job[t].th, nullptr);
  hipError_t r = hipHostRegister(m, len, hipHostRegisterDefault);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    munmap(m, len);
    return mk_fail(nullptr, r == hipErrorOutOfMemory ? MK_ERR_NOMEM : MK_ERR_HIP, "hipHostRegister(%zu): %s", len, hipGetErrorString(r));
  }
  *out = m;
  return MK_OK;
}
extern "C" int mk_host_arena_free(void *p, size_t bytes) {
  if (!p) return MK_ERR_ARG;
  const size_t huge = (size_t)2 << 20;
  const size_t len = (bytes + huge - 1) & ~(huge - 1);
  const bool ok = hipHostUnregister(p) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  munmap(p, len);
  return ok ? MK_OK : MK_ERR_HIP;
}

extern "C" int mk_host_register(void *p, size_t bytes) {
  if (!p || !bytes) return MK_ERR_ARG;
  hipError_t r = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (r != hipSuccess) { (void)hipGetLastError(); return mk_fail(nullptr, MK_ERR_HIP, "hipHostRegister(%zu): %s", bytes, hipGetErrorString(r)); }
  return MK_OK;
}
extern "C" int mk_host_register_on(int device, void *p, size_t bytes) {
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return mk_fail(nullptr, MK_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device); }
  return mk_host_register(p, bytes);
}
extern "C" int mk_host_unregister(void *p) {
  if (hipHostUnregister(p) == hipSuccess) return MK_OK;
  (void)hipGetLastError();
  return MK_ERR_HIP;
}

static mk_evpair mk_ev_get(mk_engine *e) {
  mk_evpair p{nullptr, nullptr};
  if (!e->ev_pool.empty()) { p = e->ev_pool.back(); e->ev_pool.pop_back(); return p; }
  hipEventCreate(&p.a);
  hipEventCreate(&p.b);
  return p;
}

/* the host's pass over the .shuf table: the inner substrings d with dim_start <= shuf[d] < dim_end (iseq2comem.c:693-694).  Any
 * table is taken as it is (the reference does not require a permutation).  Runs on a few threads beside the runtime's start-up. */
struct mk_accept_job {
  const int32_t *t;
  uint64_t lo, hi;
  int32_t ds, de;
  std::vector<mk_accept_pair> out;
  pthread_t th;
  bool started;
};
static void *mk_accept_run(void *arg) {
  mk_accept_job *j = (mk_accept_job *)arg;
  const int32_t *t = j->t;
  const uint32_t ds = (uint32_t)j->ds, span = (uint32_t)(j->de - j->ds);
  for (uint64_t d = j->lo; d < j->hi; d++) {
    const int32_t v = t[d];
    if ((uint32_t)v - ds < span) j->out.push_back(mk_accept_pair{(uint32_t)d, v}); /* ds <= v < de in one compare */
  }
  return nullptr;
}

static void mk_bctx_free(struct mk_bctx *c);

extern "C" int mk_engine_destroy(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  if (e->scan_lent > 0) return mk_fail(e, MK_ERR_STATE, "mk_engine_destroy: %d engine(s) still borrow this engine's scan queue (mk_engine_share_scan_queue): destroy them first", e->scan_lent);
  if (e->scan_owner) { e->scan_owner->scan_lent--; e->scan_owner = nullptr; }
  hipSetDevice(e->device);
  /* everything this engine has queued, on the streams it knows: its own, the caller's (mk_engine_set_stream), the queues of MK_OPT_SPLIT_CUS
   * (a borrowed scan queue too: this engine's scans are on it), the side stream.  Not hipDeviceSynchronize(): other engines' work is none of
   * this engine's business */
  (void)hipStreamSynchronize(e->stream); /* (a caller's stream may be the default stream, i.e. NULL: waited for like any other) */
  for (hipStream_t st : {e->own_stream, e->copy_stream, e->scan_stream, e->split_stream, e->res_stream})
    if (st && st != e->stream) (void)hipStreamSynchronize(st);
  hipFree(e->d_pairs);
  hipFree(e->d_cand); hipFree(e->d_cand_count);
  hipFree(e->d_shuf); hipFree(e->d_accept); hipFree(e->d_accept_bits); hipFree(e->d_tab); hipFree(e->d_front); hipFree(e->d_front_desc); hipFree(e->d_slot);
  hipFree(e->d_dirty_acc); hipFree(e->d_dirty_slot); hipFree(e->d_list_acc); hipFree(e->d_list_slot); hipFree(e->d_nlist);
  hipFree(e->dist.key); hipFree(e->dist.ord); hipFree(e->dist.cnt);
  hipFree(e->d_chunk); hipFree(e->d_comp_totals); hipFree(e->d_counters); hipFree(e->d_split);
  hipFree(e->d_kl); hipFree(e->d_kl_buckets);
  for (mk_bctx *c : e->bctx) mk_bctx_free(c);
  hipFree(e->d_res_ids); hipFree(e->d_res_cnt); hipFree(e->d_snap);
  if (e->h_snap) hipHostFree(e->h_snap);
  if (e->res_stream) hipStreamDestroy(e->res_stream);
  if (e->scan_stream && !e->scan_shared) hipStreamDestroy(e->scan_stream);
  if (e->split_stream) hipStreamDestroy(e->split_stream);
  if (e->ev_scan_pre) hipEventDestroy(e->ev_scan_pre);
  if (e->ev_scan_post) hipEventDestroy(e->ev_scan_post);
  if (e->ev_res) hipEventDestroy(e->ev_res);
  hipFree(e->d_text); hipFree(e->d_stream); hipFree(e->d_stream_tmp); hipFree(e->d_fa_sum); hipFree(e->d_fa_state);
  if (e->h_fa_state) hipHostFree(e->h_fa_state);
  for (int i = 0; i < MK_TICKETS; i++) if (e->ev_ticket[i]) hipEventDestroy(e->ev_ticket[i]);
  if (e->h_counters) hipHostFree(e->h_counters);
  if (e->h_ids) hipHostFree(e->h_ids);
  if (e->h_cnt) hipHostFree(e->h_cnt);
  for (int i = 0; i < MK_REGIONS; i++) {
    hipFree(e->d_stage[i]);
    if (e->ev_copied[i]) hipEventDestroy(e->ev_copied[i]);
    if (e->ev_scanned[i]) hipEventDestroy(e->ev_scanned[i]);
  }
  for (auto *v : {&e->ev_scan, &e->ev_resolve, &e->ev_clear, &e->ev_finish, &e->ev_finish_side, &e->ev_pool})
    for (auto &p : *v) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  if (e->own_stream) hipStreamDestroy(e->own_stream);
  delete e;
  return MK_OK;
}

/* allocates / frees the dirty-block bookkeeping of the sparse table passes */
static int mk_config_sparse(mk_engine *e, bool on) {
  hipFree(e->d_dirty_acc); hipFree(e->d_dirty_slot); hipFree(e->d_list_acc); hipFree(e->d_list_slot); hipFree(e->d_nlist);
  e->d_dirty_acc = e->d_dirty_slot = e->d_list_acc = e->d_list_slot = e->d_nlist = nullptr;
  e->tab.dirty = nullptr;
  e->tab.dirty_shift = 0;
  e->sparse = on;
  e->tables_tracked = false;
  if (!on) return MK_OK;
  const uint64_t S = e->P.hashsize;
  e->acc_blocks = (uint32_t)((S + MK_SPARSE_BLOCK - 1) / MK_SPARSE_BLOCK);
  e->acc_words = (e->acc_blocks + 31u) / 32u;
  const uint32_t nch = (uint32_t)((S + MK_DUMP_CHUNK - 1) / MK_DUMP_CHUNK);
  e->slot_words = (nch + 31u) / 32u;
  MK_HIP(e, mk_dev_alloc(&e->d_dirty_acc, (size_t)e->acc_words * 4));
  MK_HIP(e, mk_dev_alloc(&e->d_dirty_slot, (size_t)e->slot_words * 4));
  MK_HIP(e, mk_dev_alloc(&e->d_list_acc, (size_t)e->acc_words * 32 * 4));
  MK_HIP(e, mk_dev_alloc(&e->d_list_slot, (size_t)e->slot_words * 32 * 4));
  MK_HIP(e, mk_dev_alloc(&e->d_nlist, 4 * sizeof(uint32_t)));
  e->tab.dirty = e->d_dirty_acc;
  e->tab.dirty_shift = MK_SPARSE_SHIFT;
  return MK_OK;
}

static int mk_config_cand(mk_engine *e, uint32_t cap) {
  hipFree(e->d_cand); hipFree(e->d_cand_count);
  e->d_cand = nullptr; e->d_cand_count = nullptr;
  e->cand_cap = cap;
  MK_HIP(e, mk_dev_alloc(&e->d_cand, (size_t)e->cand_slots * (e->cand_cap + 1) * sizeof(uint4)));
  MK_HIP(e, mk_dev_alloc(&e->d_cand_count, (size_t)e->cand_slots * sizeof(uint32_t)));
  /* (the scan kernel writes every count its resolve pass reads; the fill only keeps the unused ones defined) */
  hipLaunchKernelGGL(mk_fill16_kernel, dim3(4), dim3(256), 0, e->stream, (uint4 *)e->d_cand_count, (unsigned long long)e->cand_slots / 4ull, 0u);
  MK_HIP(e, hipGetLastError());
  return MK_OK;
}

#include <time.h>
static double mk_tick_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#ifdef MK_TUNING
#define MK_TICK(label) do { if (getenv("MK_DEBUG")) { double n_ = mk_tick_now(); fprintf(stderr, "[engine create] %-22s %.4f s\n", label, n_ - tick_); tick_ = n_; } } while (0)
#else
#define MK_TICK(label) do { } while (0)
#endif

/* front table of 2^bits slots (0: none; -1: by table size -- the power of two at or below a sixth of the slots, from 2^20
 * slots up, with dense bookkeeping) */
static int mk_config_front(mk_engine *e, int bits) {
  hipFree(e->d_front);
  e->d_front = nullptr; e->front_slots = 0;
  e->tab.fr = nullptr;
  e->front = mk_front{};
  e->big_maybe_dirty = true;
  const uint64_t S = e->P.hashsize;
  if (bits < 0) {
    bits = 0;
    if (!e->sparse && S >= (1ull << 20)) while ((2ull << bits) <= S / 6) bits++; /* hashsize is a prime just below a power of two: S/8 would halve it */
    if (e->sparse) {
      /* sparse bookkeeping (tables of 2^26 slots and more: a genome's few thousand keys in half a billion slots): 2^18 front
       * slots take a whole genome-sized sketch (limit 2^16 keys), cost 4 MiB to clear and to list, and the big table's dirty
       * blocks are not touched at all; larger sketches spill into the big table as before */
      uint32_t sb = 0;
      while ((1ull << sb) < S) sb++;
      bits = sb > 21u ? 18 : (sb > 6u ? (int)sb - 3 : 3);
    }
  }
  if (bits == 0) return MK_OK;
  e->front_slots = 1ull << bits;
  MK_HIP(e, mk_dev_alloc(&e->d_front, e->front_slots * 16));
  if (!e->d_front_desc) MK_HIP(e, mk_dev_alloc((void **)&e->d_front_desc, sizeof(mk_front)));
  e->front.kc1 = (unsigned long long *)e->d_front;
  e->front.ordinv1 = e->front.kc1 + e->front_slots;
  e->front.state = (uint32_t *)(e->d_counters + 4);
  { uint32_t sb = 0; while ((1ull << sb) < S) sb++; e->front.shift = sb > (uint32_t)bits ? sb - (uint32_t)bits : 0u; } /* (S-1) >> shift < 2^bits */
  e->front.mask = (uint32_t)(e->front_slots - 1);
  e->front.limit = (uint32_t)(e->front_slots / 4); /* closed to new keys from a quarter full at the start of a launch */
  hipLaunchKernelGGL(mk_front_store_kernel, dim3(1), dim3(1), 0, e->stream, e->d_front_desc, e->front); /* (by value: nothing of the host's is read later) */
  MK_HIP(e, hipGetLastError());
  e->tab.fr = e->d_front_desc;
  return MK_OK;
}

static int mk_dist_reserve(mk_engine *e, uint64_t cap) {
  if (cap <= e->dist.cap && e->dist.key) return MK_OK;
  hipFree(e->dist.key); hipFree(e->dist.ord); hipFree(e->dist.cnt);
  e->dist.key = nullptr; e->dist.ord = nullptr; e->dist.cnt = nullptr; e->dist.cap = 0;
  MK_HIP(e, mk_dev_alloc(&e->dist.key, cap * 8));
  MK_HIP(e, mk_dev_alloc(&e->dist.ord, cap * 8));
  MK_HIP(e, mk_dev_alloc(&e->dist.cnt, cap * 4));
  e->dist.cap = cap;
  return MK_OK;
}

static int mk_tables_alloc(mk_engine *e);
static int mk_engine_init(mk_engine *e, const mk_params *p, bool lazy_tables) {
#ifdef MK_TUNING
  double tick_ = mk_tick_now();
#endif
  /* The accepted inner substrings are found on the host, by up to four threads, WHILE the runtime creates the engine's queue
   * (20 ms of every start-up on this runtime): the pass over the 64 MiB table then costs nothing, and neither the table nor
   * anything made from it has to be waited for -- mk_engine_create returns without a single stream synchronisation. */
  const uint64_t L = p->shuf_len;
  enum { NT = 4 };
  mk_accept_job job[NT];
  const int nt = L >= (1ull << 20) ? NT : 1;
  for (int t = 0; t < nt; t++) {
    job[t].t = p->shuf_table; job[t].lo = L / nt * (uint64_t)t; job[t].hi = t + 1 == nt ? L : L / nt * (uint64_t)(t + 1);
    job[t].ds = p->dim_start; job[t].de = p->dim_end;
    job[t].started = nt > 1 && pthread_create(&job[t].th, nullptr, mk_accept_run, &job[t]) == 0;
  }
  /* ... and one more thread makes the runtime load this library's code object for the device (4 ms at the first launch
   * otherwise), also beside the queue creation */
  struct warm_job { int device; pthread_t th; bool started; } warm{e->device, {}, false};
  warm.started = pthread_create(&warm.th, nullptr, [](void *arg) -> void * {
    warm_job *w = (warm_job *)arg;
    hipFuncAttributes attr;
    if (hipSetDevice(w->device) != hipSuccess || hipFuncGetAttributes(&attr, (const void *)mk_fill16_kernel) != hipSuccess) (void)hipGetLastError();
    return nullptr;
  }, &warm) == 0;
  auto join_accept = [&]() {
    for (int t = 0; t < nt; t++) {
      if (job[t].started) { pthread_join(job[t].th, nullptr); job[t].started = false; }
      else if (job[t].t) mk_accept_run(&job[t]); /* a small table, or a thread that did not start: here */
      job[t].t = nullptr;
    }
  };
  auto join_all = [&]() {
    join_accept();
    if (warm.started) { pthread_join(warm.th, nullptr); warm.started = false; }
  };
  struct joiner { decltype(join_all) &f; ~joiner() { f(); } } join_guard{join_all}; /* no thread outlives an early return */

  MK_HIP(e, hipSetDevice(e->device));
  MK_HIP(e, hipDeviceGetAttribute(&e->num_cu, hipDeviceAttributeMultiprocessorCount, e->device)); /* hipGetDeviceProperties takes 30 ms */
  MK_TICK("device");
#ifdef MK_TUNING
  /* experiment builds: the engine's queue on MK_TUNE_CUS compute units from bit MK_TUNE_CU_FIRST of the queue's CU mask on (the bits go
   * round the XCDs), every grid sized for that many -- what a kernel takes on a part of the chip */
  if (const char *t = getenv("MK_TUNE_CUS")) {
    const int n = atoi(t), first = getenv("MK_TUNE_CU_FIRST") ? atoi(getenv("MK_TUNE_CU_FIRST")) : 0;
    if (n < 1 || first < 0 || first + n > e->num_cu) return mk_fail(e, MK_ERR_ARG, "MK_TUNE_CUS / MK_TUNE_CU_FIRST out of range");
    uint32_t mask[16] = {0};
    for (int i = first; i < first + n; i++) mask[i >> 5] |= 1u << (i & 31);
    MK_HIP(e, hipExtStreamCreateWithCUMask(&e->own_stream, (uint32_t)((e->num_cu + 31) / 32), mask));
    e->num_cu = n;
    fprintf(stderr, "[tuning] queue on CUs %d..%d of the mask\n", first, first + n - 1);
  } else
#endif
  MK_HIP(e, hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
  MK_TICK("stream 1");
  /* host-to-device copies ride on the engine's own stream: a second stream means a second hardware queue (30-60 ms of
   * start-up on this runtime), and the scan of a 128 MiB staging region (45 us) is nothing next to copying it (2.3 ms),
   * so there is no overlap worth a queue.  With a caller stream (mk_engine_set_stream) the scans run there and the copies
   * stay here, ordered by events. */
  e->copy_stream = e->own_stream;
  e->stream = e->own_stream;
  for (int i = 0; i < MK_REGIONS; i++) {
    MK_HIP(e, hipEventCreateWithFlags(&e->ev_copied[i], hipEventDisableTiming));
    MK_HIP(e, hipEventCreateWithFlags(&e->ev_scanned[i], hipEventDisableTiming));
  }

  e->P = *p;
  mk_keyparams &kp = e->kp;
  kp.tupmask = p->tupmask; kp.domask = p->domask; kp.undomask = p->undomask;
  kp.lowmask = (1ull << (2 * p->half_outctx_len)) - 1ull;
  kp.TL = (uint32_t)p->TL; kp.crvsaddmove = (uint32_t)p->crvsaddmove;
  kp.out2 = 2u * (uint32_t)p->half_outctx_len;
  kp.key_lshift = 2u * (uint32_t)p->TL - 4u * (uint32_t)p->half_outctx_len;
  kp.dr4 = 4u * (uint32_t)p->drlevel;
  kp.dim_start = p->dim_start; kp.dim_end = p->dim_end;
  kp.S = p->hashsize;

  /* .shuf table + the filter list B = A u revcomp(A), A = accepted inner substrings (iseq2comem.c:693-694).
   * The inner substring (2*subk bases) sits in the middle of the k-mer, so the reverse-complement k-mer's
   * inner substring is the reverse complement of the forward one: see mk_kernels.hip.h, "LDS filter".
   * Only the accepted (d, shuf[d]) pairs go to the device (mk_accept_scatter_kernel): the resolve kernel reads shuf[d] behind a
   * set accept bit and nowhere else, so the rest of the device's table stays unwritten. */
  const uint32_t dbits = 4u * (uint32_t)p->subk;
  const size_t bit_bytes = ((size_t)((L + 31) / 32) * sizeof(uint32_t) + 15u) & ~(size_t)15u;
  MK_HIP(e, mk_dev_alloc(&e->d_shuf, L * sizeof(int32_t)));
  MK_HIP(e, mk_dev_alloc(&e->d_accept_bits, bit_bytes));
  MK_HIP(e, mk_dev_alloc(&e->d_counters, 8 * sizeof(unsigned long long)));
  MK_TICK("events + first allocs");
  join_accept();
  size_t npairs = 0;
  for (int t = 0; t < nt; t++) npairs += job[t].out.size();
  if (npairs > (1u << 30)) return mk_fail(e, MK_ERR_ARG, "mk_engine_create: %zu accepted inner substrings", npairs);
  e->n_accept = (uint32_t)(2u * npairs);
  /* one pinned block: the counters' mirror, the component sizes, and the accepted pairs on their way up (a pageable source
   * costs the first upload 8 ms of staging set-up) */
  const size_t fixed = (16 + MK_MAX_COMP) * sizeof(unsigned long long);
  MK_HIP(e, mk_pin_alloc((void **)&e->h_counters, fixed + (npairs + 1) * sizeof(mk_accept_pair), hipHostMallocDefault));
  e->h_comp_totals = e->h_counters + 16;
  mk_accept_pair *hp = (mk_accept_pair *)((uint8_t *)e->h_counters + fixed);
  {
    size_t at = 0;
    for (int t = 0; t < nt; t++) { if (!job[t].out.empty()) memcpy(hp + at, job[t].out.data(), job[t].out.size() * sizeof(mk_accept_pair)); at += job[t].out.size(); }
  }
  MK_TICK("accept pass joined + pinned block");
  MK_HIP(e, mk_dev_alloc(&e->d_accept, ((size_t)e->n_accept + 2) * sizeof(uint32_t)));
  if (warm.started) { pthread_join(warm.th, nullptr); warm.started = false; }
  MK_TICK("code object");
  hipLaunchKernelGGL(mk_fill16_kernel, dim3((unsigned)e->num_cu * 2u), dim3(256), 0, e->own_stream, (uint4 *)e->d_accept_bits,
                     (unsigned long long)(bit_bytes / 16u), 0u);
  hipLaunchKernelGGL(mk_fill16_kernel, dim3(1), dim3(64), 0, e->own_stream, (uint4 *)e->d_counters, 4ull, 0u);
  MK_HIP(e, hipGetLastError());
  MK_TICK("first launches");
  if (npairs) {
    /* the kernel reads the pairs out of the pinned block itself (it is mapped into the device): no copy command at start-up */
    uint64_t blocks = (npairs + 255) / 256;
    if (blocks > (uint64_t)e->num_cu * 8u) blocks = (uint64_t)e->num_cu * 8u;
    hipLaunchKernelGGL(mk_accept_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, e->own_stream, (const mk_accept_pair *)hp,
                       (uint32_t)npairs, dbits, e->d_shuf, e->d_accept_bits, e->d_accept);
    MK_HIP(e, hipGetLastError());
  }
  MK_TICK("accept upload + scatter");
  /* LDS filter: 2^bm_bits words of 32 bits indexed by the inner substring's bits 10.. (at most 64 KiB) */
  {
    int wb = 4 * p->subk - 10;
    e->bm_bits = (uint32_t)(wb < 0 ? 0 : wb > 14 ? 14 : wb);
  }

  e->nchunks = (uint32_t)((p->hashsize + MK_DUMP_CHUNK - 1) / MK_DUMP_CHUNK);
  if (p->component_num > (int)MK_MAX_COMP) return mk_fail(e, MK_ERR_ARG, "component_num %d > %u", p->component_num, MK_MAX_COMP);
  e->tab.err = (uint32_t *)(e->d_counters + 2);
  e->comps.resize((size_t)p->component_num);
  e->cand_slots = (uint32_t)e->num_cu * 16u; /* at most 16 waves per workgroup, one workgroup per CU */
  { int rc = mk_config_cand(e, 8192u); if (rc) return rc; }
  MK_TICK("candidate buffers");
  for (int i = 0; i < MK_TICKETS; i++) MK_HIP(e, hipEventCreateWithFlags(&e->ev_ticket[i], hipEventDisableTiming));
#ifdef MK_TUNING /* experiment knobs: compiled only into tools/ builds (make tuning), never into the shipped library */
  if (const char *t = getenv("MK_SCAN_THREADS")) { int v = atoi(t); if (v == 512 || v == 768 || v == 1024) e->tune_threads = v; }
  if (const char *t = getenv("MK_SCAN_ONEPASS")) e->tune_onepass = atoi(t) != 0;
  if (const char *t = getenv("MK_SCAN_CB")) { int v = atoi(t); if (v >= 16 && v <= MK_MAX_CB && v % 16 == 0) e->tune_cb = (uint32_t)v; }
#endif
  return lazy_tables ? MK_OK : mk_tables_alloc(e);
}

/* The hashsize-slot accumulation and layout tables, their sparse bookkeeping, the front table, the key list and the dump's chunk
 * arrays: everything a sketch of ONE input needs and a batch of files (mk_sketch_batch_*: tables of its own per file) never touches
 * -- 21 GB at L2K11.  At creation, or with MK_ENGINE_LAZY_TABLES at the first mk_sketch_begin / mk_engine_set_option. */
static int mk_tables_alloc(mk_engine *e) {
  if (e->tables_ready) return MK_OK;
#ifdef MK_TUNING
  double tick_ = mk_tick_now();
#endif
  const mk_params *p = &e->P;
  MK_HIP(e, hipSetDevice(e->device));
  const uint64_t S = p->hashsize;
  e->tab_bytes = S * (8 + 8);
  MK_HIP(e, mk_dev_alloc(&e->d_tab, e->tab_bytes));
  e->tab.kc = (unsigned long long *)e->d_tab;
  e->tab.ordinv = e->tab.kc + S;
  MK_HIP(e, mk_dev_alloc(&e->d_slot, S * sizeof(uint32_t)));
  MK_TICK("table + layout table");
  /* sparse bookkeeping from 2^26 slots up (L2K11: 537 M slots for a genome's few thousand keys);
   * mk_engine_set_option(MK_OPT_SPARSE) forces it off/on (the tests run the small tables both ways) */
  { int rc = mk_config_sparse(e, S >= (1ull << 26)); if (rc) return rc; }
  { int rc = mk_config_front(e, -1); if (rc) return rc; }
  MK_TICK("sparse + front");
  /* the distinct-key list: hashlimit+1 entries suffice for KOC/SET, MK_MODE_OCC_SET may fill the table (fastq2co never aborts).
   * With sparse bookkeeping (537 M slots at L2K11: 10.7 GB of list for sketches of a few thousand keys) it starts at 32 M
   * entries and grows when a compaction counts more (mk_dist_fit) */
  { int rc = mk_dist_reserve(e, e->sparse && S > (32ull << 20) ? (32ull << 20) : S); if (rc) return rc; }
  MK_HIP(e, mk_dev_alloc(&e->d_chunk, (size_t)e->nchunks * (size_t)p->component_num * sizeof(uint32_t)));
  MK_HIP(e, mk_dev_alloc(&e->d_comp_totals, MK_MAX_COMP * sizeof(unsigned long long)));
  MK_TICK("key list + dump tables");
  e->tables_ready = true;
  return MK_OK;
}

extern "C" int mk_engine_create(const mk_params *p, int device, mk_engine **out) { return mk_engine_create_ex(p, device, 0u, out); }

extern "C" int mk_engine_create_ex(const mk_params *p, int device, unsigned flags, mk_engine **out) {
  if (!p || !out || !p->shuf_table) return mk_fail(nullptr, MK_ERR_ARG, "mk_engine_create: null argument");
  if (p->k < 1 || p->k > 16 || p->hashsize < 251u || p->shuf_len != (1ull << (4 * p->subk)))
    return mk_fail(nullptr, MK_ERR_ARG, "mk_engine_create: inconsistent mk_params (use mk_params_init)");
  int n = 0;
  hipError_t r = hipGetDeviceCount(&n);
  if (r != hipSuccess || n <= 0)
    return mk_fail(nullptr, MK_ERR_NO_DEVICE, "no HIP device (%s); this library has no CPU path",
                   r == hipSuccess ? "device count 0" : hipGetErrorString(r));
  if (device < 0 || device >= n) return mk_fail(nullptr, MK_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
  mk_engine *e = new mk_engine();
  e->device = device;
  int rc = mk_engine_init(e, p, (flags & MK_ENGINE_LAZY_TABLES) != 0);
  if (rc != MK_OK) {
    snprintf(g_create_error, sizeof g_create_error, "%s", e->err);
    mk_engine_destroy(e);
    return rc;
  }
  *out = e;
  return MK_OK;
}

static int mk_flush_region(mk_engine *e);
static int mk_tail_end(mk_engine *e);

/* MK_POISON: what the last sketch left in the engine's scratch -- candidate records, the key list, the dump's arrays, the FASTA
 * stream's buffers, the staging regions -- is overwritten with the pattern at the next mk_sketch_begin, on the stream the new sketch's
 * kernels follow on.  Buffers the side stream may still work on (an outstanding mk_sketch_finish_begin) are left alone; so are the tables,
 * whose clears are the engine's own business (that is what the hook is there to check). */
static int mk_poison_scratch(mk_engine *e) {
  if (mk_poison_byte() < 0) return MK_OK;
  { static const bool off = getenv("MK_POISON_SCRATCH") && atoi(getenv("MK_POISON_SCRATCH")) == 0; if (off) return MK_OK; } /* (bisecting: allocations only) */
  hipStream_t s = e->stream;
  if (e->d_cand) MK_HIP(e, mk_dev_repoison(e->d_cand, (size_t)e->cand_slots * (e->cand_cap + 1) * sizeof(uint4), s));
  if (e->d_kl) MK_HIP(e, mk_dev_repoison(e->d_kl, (size_t)e->kl_cap * (8 + 8 + 4 + 4 + 2) + 64, s));
  if (e->d_kl_buckets) MK_HIP(e, mk_dev_repoison(e->d_kl_buckets, (size_t)e->kl_bucket_cap * 4, s));
  if (e->d_split) MK_HIP(e, mk_dev_repoison(e->d_split, 2 * MK_SPLIT_MAX * sizeof(unsigned long long), s));
  if (!e->res_pending) {
    if (e->dist.key) {
      MK_HIP(e, mk_dev_repoison(e->dist.key, (size_t)e->dist.cap * 8, s));
      MK_HIP(e, mk_dev_repoison(e->dist.ord, (size_t)e->dist.cap * 8, s));
      MK_HIP(e, mk_dev_repoison(e->dist.cnt, (size_t)e->dist.cap * 4, s));
    }
    if (e->d_chunk) MK_HIP(e, mk_dev_repoison(e->d_chunk, (size_t)e->nchunks * (size_t)e->P.component_num * sizeof(uint32_t), s));
    if (e->d_comp_totals) MK_HIP(e, mk_dev_repoison(e->d_comp_totals, MK_MAX_COMP * sizeof(unsigned long long), s));
    if (e->d_res_ids) {
      MK_HIP(e, mk_dev_repoison(e->d_res_ids, (size_t)e->res_cap * 4, s));
      MK_HIP(e, mk_dev_repoison(e->d_res_cnt, (size_t)e->res_cap * 2, s));
    }
  }
  if (e->d_text) MK_HIP(e, mk_dev_repoison(e->d_text, e->text_cap, s));
  if (e->d_stream) MK_HIP(e, mk_dev_repoison(e->d_stream, e->stream_cap, s));
  if (e->d_stream_tmp) MK_HIP(e, mk_dev_repoison(e->d_stream_tmp, 8192, s));
  if (e->d_fa_sum) MK_HIP(e, mk_dev_repoison(e->d_fa_sum, e->fa_sum_cap * sizeof(mk_fa_sum), s));
  if (e->d_stage[0] && e->copy_stream == e->stream) /* (rows staged for a sketch that was never finished are dropped by the begin) */
    for (int i = 0; i < MK_REGIONS; i++) MK_HIP(e, mk_dev_repoison(e->d_stage[i], e->stage_bytes, s));
  return MK_OK;
}

/* MK_OPT_SPLIT_CUS.  The bits of a queue's CU mask go round the XCDs (bit i = compute unit i / 8 of XCD i % 8), so the first
 * num_cu - r bits and the last r are both spread evenly over the eight of them -- and r is a multiple of 32 so that every shader engine
 * (four an XCD) keeps the same number of units on either side: workgroups are dealt to the shader engines in turn whatever units they
 * have left, and with 24, 40 or 48 units on the second queue two of the one-a-CU workgroups of a kernel land on one unit and run one
 * after the other (a pass 3.2 - 4.8 ms instead of 2.33, profiles/r05_split_queues.txt).  The caller has waited for the engine's streams. */
static int mk_config_split(mk_engine *e, int r) {
  bool nomask = false;
#ifdef MK_TUNING
  /* experiment builds: the same two queues WITHOUT CU masks -- the grids alone (num_cu - r scan workgroups, r resolve workgroups, one a CU by
   * their LDS) divide the device, the dispatcher places them where there is room; any multiple of 8 */
  nomask = getenv("MK_TUNE_SPLIT_NOMASK") != nullptr;
#endif
  if (r != 0 && !nomask && (r < 32 || r > e->num_cu / 2 || r % 32 != 0))
    return mk_fail(e, MK_ERR_ARG, "MK_OPT_SPLIT_CUS takes 0 (one queue) or a multiple of 32 up to half the device's %d compute units", e->num_cu);
  if (e->batch_begun != e->batch_ended) return mk_fail(e, MK_ERR_STATE, "MK_OPT_SPLIT_CUS while a batch is in flight");
  if (e->scan_lent > 0) return mk_fail(e, MK_ERR_STATE, "MK_OPT_SPLIT_CUS: %d engine(s) borrow this engine's scan queue (set them back to one queue first)", e->scan_lent);
  if (e->scan_owner) { e->scan_owner->scan_lent--; e->scan_owner = nullptr; }
  if (e->stream == e->split_stream && e->split_stream) e->stream = e->own_stream;
  e->tail = e->tail_active = false;
  if (e->scan_stream) { (void)hipStreamSynchronize(e->scan_stream); if (!e->scan_shared) (void)hipStreamDestroy(e->scan_stream); e->scan_stream = nullptr; }
  if (e->split_stream) { (void)hipStreamSynchronize(e->split_stream); (void)hipStreamDestroy(e->split_stream); e->split_stream = nullptr; }
  /* the side stream carries the scan queue's mask under this option: made anew at the next finish (mk_res_reserve) */
  if (e->res_stream) { (void)hipStreamSynchronize(e->res_stream); (void)hipStreamDestroy(e->res_stream); e->res_stream = nullptr; }
  e->split_cus = 0; e->scan_cus = 0; e->scan_shared = false;
  if (e->init_queued) { MK_HIP(e, hipStreamSynchronize(e->own_stream)); e->init_queued = false; } /* (what creation queued is ordered with no other stream) */
  if (r == 0) return MK_OK;
  const int n = e->num_cu;
  uint32_t scan_mask[16] = {0}, rest_mask[16] = {0};
  if (n > 512) return mk_fail(e, MK_ERR_ARG, "MK_OPT_SPLIT_CUS: %d compute units", n);
  for (int i = 0; i < n; i++) (i < n - r ? scan_mask : rest_mask)[i >> 5] |= 1u << (i & 31);
  if (nomask) {
    MK_HIP(e, hipStreamCreateWithFlags(&e->scan_stream, hipStreamNonBlocking));
    MK_HIP(e, hipStreamCreateWithFlags(&e->split_stream, hipStreamNonBlocking));
  } else {
    MK_HIP(e, hipExtStreamCreateWithCUMask(&e->scan_stream, (uint32_t)((n + 31) / 32), scan_mask));
    MK_HIP(e, hipExtStreamCreateWithCUMask(&e->split_stream, (uint32_t)((n + 31) / 32), rest_mask));
  }
  if (!e->ev_scan_pre) {
    MK_HIP(e, hipEventCreateWithFlags(&e->ev_scan_pre, hipEventDisableTiming));
    MK_HIP(e, hipEventCreateWithFlags(&e->ev_scan_post, hipEventDisableTiming));
  }
  e->split_cus = r; e->scan_cus = n - r;
  if (e->stream == e->own_stream) e->stream = e->split_stream; /* (a caller's stream stays: only the scan moves to its queue) */
  return MK_OK;
}

extern "C" int mk_engine_set_option(mk_engine *e, int option, int64_t value) {
  if (!e) return MK_ERR_ARG;
  if (e->begun) return mk_fail(e, MK_ERR_STATE, "mk_engine_set_option inside a sketch (between begin and finish)");
  /* the side stream still lays out and dumps the last sketch: the key list, the staging arrays and the pinned result arrays it
   * works on are what several options free */
  if (e->res_pending) return mk_fail(e, MK_ERR_STATE, "mk_engine_set_option while a result is outstanding (mk_sketch_finish_end first)");
  MK_HIP(e, hipSetDevice(e->device));
  /* a lazily created engine (MK_ENGINE_LAZY_TABLES): only the options that rebuild pieces of the hashsize-slot tables make them now; the
   * others (the command line's --direct, batch table bits, split queues, ..) leave the 21 GB of an L2K11 engine unmade */
  if (option == MK_OPT_SPARSE || option == MK_OPT_FRONT_BITS || option == MK_OPT_KEYLIST_CAP) { int rc = mk_tables_alloc(e); if (rc) return rc; }
  MK_HIP(e, hipStreamSynchronize(e->stream));
  if (e->res_stream) MK_HIP(e, hipStreamSynchronize(e->res_stream));
  switch (option) {
    case MK_OPT_SPARSE: {
      if (value < -1 || value > 1) return mk_fail(e, MK_ERR_ARG, "MK_OPT_SPARSE takes -1 (by table size), 0 or 1");
      int rc = mk_config_sparse(e, value < 0 ? e->P.hashsize >= (1u << 26) : value != 0);
      if (rc == MK_OK && !e->sparse) rc = mk_dist_reserve(e, e->kp.S); /* the dense passes do not grow the key list */
      return rc ? rc : mk_config_front(e, e->front_bits_opt);
    }
    case MK_OPT_FRONT_BITS:
      if (value < -1 || (value > 0 && value < 3) || value > 28) return mk_fail(e, MK_ERR_ARG, "MK_OPT_FRONT_BITS takes -1 (by table size), 0 (none) or 3..28");
      e->front_bits_opt = (int)value;
      return mk_config_front(e, e->front_bits_opt);
    case MK_OPT_CAND_CAP:
      if (value < 0 || value > (1 << 20)) return mk_fail(e, MK_ERR_ARG, "MK_OPT_CAND_CAP takes 0 .. 2^20 records per scan wave");
      if (e->batch_begun != e->batch_ended) return mk_fail(e, MK_ERR_STATE, "MK_OPT_CAND_CAP while a batch is in flight");
      return mk_config_cand(e, (uint32_t)value);
    case MK_OPT_DIRECT_HOST:
      e->direct_host = value != 0;
      return MK_OK;
    case MK_OPT_SPLIT_CUS: return mk_config_split(e, (int)value);
    case MK_OPT_BATCH_TAB_BITS:
      if (value != 0 && (value < 9 || value > 22)) return mk_fail(e, MK_ERR_ARG, "MK_OPT_BATCH_TAB_BITS takes 0 (by file size) or 9..22");
      if (e->batch_begun != e->batch_ended) return mk_fail(e, MK_ERR_STATE, "MK_OPT_BATCH_TAB_BITS while a batch is in flight");
      e->batch_tb_opt = (int)value;
      return MK_OK;
    case MK_OPT_KEYLIST_CAP: { /* sparse bookkeeping only: the list is grown by the finish / export that needs more */
      if (!e->sparse) return mk_fail(e, MK_ERR_ARG, "MK_OPT_KEYLIST_CAP needs sparse bookkeeping (the dense passes do not grow the key list)");
      if (value < 16 || (uint64_t)value > e->kp.S) return mk_fail(e, MK_ERR_ARG, "MK_OPT_KEYLIST_CAP takes 16 .. hashsize entries");
      hipFree(e->dist.key); hipFree(e->dist.ord); hipFree(e->dist.cnt);
      e->dist.key = nullptr; e->dist.ord = nullptr; e->dist.cnt = nullptr; e->dist.cap = 0;
      return mk_dist_reserve(e, (uint64_t)value);
    }
    case MK_OPT_RESULT_CAP: {
      if (value < 1 || value > (int64_t)e->P.hashsize) return mk_fail(e, MK_ERR_ARG, "MK_OPT_RESULT_CAP takes 1 .. hashsize entries");
      if (e->h_ids) hipHostFree(e->h_ids);
      if (e->h_cnt) hipHostFree(e->h_cnt);
      e->h_ids = nullptr; e->h_cnt = nullptr; e->h_cap = 0;
      MK_HIP(e, mk_pin_alloc((void **)&e->h_ids, (size_t)value * 4, hipHostMallocDefault));
      MK_HIP(e, mk_pin_alloc((void **)&e->h_cnt, (size_t)value * 2, hipHostMallocDefault));
      e->h_cap = (uint64_t)value;
      return MK_OK;
    }
    default: return mk_fail(e, MK_ERR_ARG, "unknown engine option %d", option);
  }
}

extern "C" int mk_engine_set_stream(mk_engine *e, void *hip_stream) {
  if (!e) return MK_ERR_ARG;
  if (e->region_open) { /* staged rows are scanned on the stream they were pushed for */
    MK_HIP(e, hipSetDevice(e->device));
    int rc = mk_flush_region(e);
    if (rc) return rc;
  }
  if (e->init_queued) { /* what mk_engine_create queued on the engine's own stream is not ordered with any other stream */
    MK_HIP(e, hipSetDevice(e->device));
    MK_HIP(e, hipStreamSynchronize(e->own_stream));
    e->init_queued = false;
  }
  /* a sketch begun with MK_BEGIN_NOTHING_FOLLOWS left its tail on the unmasked queue: the second queue waits for it and takes over
   * again first (mk_tail_end), so that what the NEW stream is ordered with below is everything the engine has queued */
  { int rc = mk_tail_end(e); if (rc) return rc; }
  if (e->split_stream && (hipStream_t)hip_stream != e->split_stream) { /* (the caller's stream starts behind the engine's queued work) */
    MK_HIP(e, hipSetDevice(e->device));
    MK_HIP(e, hipEventRecord(e->ev_scan_pre, e->split_stream));
    MK_HIP(e, hipStreamWaitEvent((hipStream_t)hip_stream, e->ev_scan_pre, 0));
  }
  /* NULL is a real stream (HIP's default stream, which is what torch.cuda.current_stream() usually is): it must
   * not mean "keep the engine's own stream", or caller-side ordering silently disappears */
  e->stream = (hipStream_t)hip_stream;
  return MK_OK;
}
/* Two engines with the same MK_OPT_SPLIT_CUS setting: e's scans go to WITH's scan queue from now on, one after the other in the order
 * they are pushed (two queues with the same CU mask would have both scans' workgroups compete for the units, and every kernel's
 * duration would include its wait for them).  WITH owns the queue and keeps it while it is lent (scan_lent): its own
 * MK_OPT_SPLIT_CUS and mk_engine_destroy fail until e has gone back to one queue or been destroyed. */
extern "C" int mk_engine_share_scan_queue(mk_engine *e, mk_engine *with) {
  if (!e || !with || e == with) return MK_ERR_ARG;
  if (e->begun) return mk_fail(e, MK_ERR_STATE, "mk_engine_share_scan_queue inside a sketch");
  if (!e->scan_stream || !with->scan_stream || with->scan_shared || e->split_cus != with->split_cus || e->device != with->device)
    return mk_fail(e, MK_ERR_ARG, "mk_engine_share_scan_queue: both engines need the same MK_OPT_SPLIT_CUS setting on one device, and the owner a queue of its own");
  MK_HIP(e, hipSetDevice(e->device));
  MK_HIP(e, hipStreamSynchronize(e->scan_stream));
  if (e->scan_lent > 0) return mk_fail(e, MK_ERR_STATE, "mk_engine_share_scan_queue: this engine's own queue is lent to another engine");
  if (!e->scan_shared) MK_HIP(e, hipStreamDestroy(e->scan_stream));
  if (e->scan_owner) e->scan_owner->scan_lent--;
  e->scan_stream = with->scan_stream;
  e->scan_shared = true;
  e->scan_owner = with;
  with->scan_lent++;
  return MK_OK;
}
extern "C" int mk_engine_use_own_stream(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  if (e->region_open) {
    MK_HIP(e, hipSetDevice(e->device));
    int rc = mk_flush_region(e);
    if (rc) return rc;
  }
  { int rc = mk_tail_end(e); if (rc) return rc; } /* (MK_BEGIN_NOTHING_FOLLOWS: the second queue waits for the tail on the unmasked one) */
  e->stream = e->split_stream ? e->split_stream : e->own_stream;
  return MK_OK;
}

extern "C" int mk_engine_sync(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  MK_HIP(e, hipSetDevice(e->device));
  { int rc = mk_flush_region(e); if (rc) return rc; }
  if (e->copy_stream != e->stream) MK_HIP(e, hipStreamSynchronize(e->copy_stream));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  return MK_OK;
}

extern "C" int mk_profile_enable(mk_engine *e, int on) { if (!e) return MK_ERR_ARG; e->profiling = on != 0; return MK_OK; }

extern "C" int mk_profile_reset(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  MK_HIP(e, hipSetDevice(e->device));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  if (e->res_stream) MK_HIP(e, hipStreamSynchronize(e->res_stream));
  for (auto *v : {&e->ev_scan, &e->ev_resolve, &e->ev_clear, &e->ev_finish, &e->ev_finish_side}) {
    for (auto &p : *v) e->ev_pool.push_back(p);
    v->clear();
  }
  e->prof_rows = e->prof_bytes = 0;
  return MK_OK;
}

extern "C" int mk_profile_get(mk_engine *e, mk_profile *out) {
  if (!e || !out) return MK_ERR_ARG;
  MK_HIP(e, hipSetDevice(e->device));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  if (e->res_stream) MK_HIP(e, hipStreamSynchronize(e->res_stream));
  memset(out, 0, sizeof *out);
  auto sum = [&](std::vector<mk_evpair> &v) {
    double s = 0;
    for (auto &p : v) { float ms = 0; if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) s += ms; }
    return s;
  };
  out->scan_ms = sum(e->ev_scan);
  out->scan_launches = e->ev_scan.size();
  out->resolve_ms = sum(e->ev_resolve);
  out->clear_ms = sum(e->ev_clear);
  out->finish_ms = sum(e->ev_finish);
  out->finish_side_ms = sum(e->ev_finish_side);
  out->rows_scanned = e->prof_rows;
  out->bases_scanned = e->prof_bytes;
  return MK_OK;
}

/* MK_BEGIN_NOTHING_FOLLOWS is over with its sketch: whatever begins next (a sketch, a batch) starts on the second queue again, behind the work
 * the unmasked queue did for the last one, and without the mark */
static int mk_tail_end(mk_engine *e) {
  e->tail = false;
  if (e->tail_active) {
    if (e->split_stream && e->stream == e->own_stream) {
      MK_HIP(e, hipEventRecord(e->ev_scan_pre, e->own_stream));
      MK_HIP(e, hipStreamWaitEvent(e->split_stream, e->ev_scan_pre, 0));
      e->stream = e->split_stream;
    }
    e->tail_active = false;
  }
  return MK_OK;
}

extern "C" int mk_sketch_begin_occ(mk_engine *e, int min_occurrence) {
  if (!e) return MK_ERR_ARG;
  if (min_occurrence < 1 || min_occurrence >= 15) /* iseq2comem.c:325 */
    return mk_fail(e, MK_ERR_ARG, "fastq2co(): Occurence num should be 1..14 (got %d)", min_occurrence);
  int rc = mk_sketch_begin(e, MK_MODE_OCC_SET);
  if (rc == MK_OK) e->min_occ = (uint32_t)min_occurrence;
  return rc;
}

extern "C" int mk_sketch_begin(mk_engine *e, int mode) {
  if (!e) return MK_ERR_ARG;
  const bool tail = (mode & MK_BEGIN_NOTHING_FOLLOWS) != 0;
  mode &= ~MK_BEGIN_NOTHING_FOLLOWS;
  if (mode < MK_MODE_KOC || mode > MK_MODE_OCC_SET) return MK_ERR_ARG;
  MK_HIP(e, hipSetDevice(e->device));
  { int rc = mk_tables_alloc(e); if (rc) return rc; } /* MK_ENGINE_LAZY_TABLES: the first sketch of one input makes them */
  { int rc = mk_tail_end(e); if (rc) return rc; }
  e->tail = tail && e->scan_stream && e->stream == e->split_stream; /* (means something on split queues only) */
  { int rc = mk_poison_scratch(e); if (rc) return rc; }
  mk_evpair ev{};
  if (e->profiling) { ev = mk_ev_get(e); MK_HIP(e, hipEventRecord(ev.a, e->stream)); }
  /* the table clear the reference does with memset(co,0,..) (iseq2comem.c:223,663) */
  bool clear_front = false;
  if (e->sparse && e->tables_tracked) {
    /* the accumulation table is empty except in the blocks the last sketch marked: clear those, and the marks.  The layout
     * table is empty already (the key-list dump hands every slot back); after a finish that went wrong it is filled anew */
    if (!e->tab.fr || e->big_maybe_dirty) { /* (behind a front table: only when the last sketch may have reached the big table) */
      hipLaunchKernelGGL(mk_dirty_list_kernel, dim3(1), dim3(1024), 0, e->stream, e->d_dirty_acc, e->acc_words, e->d_list_acc, e->d_nlist, 1,
                         (const uint32_t *)nullptr);
      hipLaunchKernelGGL(mk_dirty_clear_kernel, dim3((unsigned)e->num_cu * 8u), dim3(256), 0, e->stream, e->tab.kc, e->tab.ordinv, e->kp.S,
                         (const uint32_t *)e->d_list_acc, (const uint32_t *)e->d_nlist, (uint32_t)MK_SPARSE_SHIFT, (uint32_t *)nullptr,
                         (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t)MK_DUMP_SHIFT);
      MK_HIP(e, hipGetLastError());
    }
    if (e->tab.fr) { clear_front = true; e->big_maybe_dirty = false; }
    if (!e->slot_clean) {
      MK_HIP(e, hipMemsetAsync(e->d_slot, 0xFF, (size_t)e->kp.S * sizeof(uint32_t), e->stream));
      e->slot_clean = true;
    }
  } else {
    /* behind a front table the S-slot table is cleared only when the last sketch may have used it */
    if (!e->tab.fr || e->big_maybe_dirty) MK_HIP(e, hipMemsetAsync(e->d_tab, 0, e->tab_bytes, e->stream));
    if (e->tab.fr) { clear_front = true; e->big_maybe_dirty = false; }
    if (e->sparse) {
      MK_HIP(e, hipMemsetAsync(e->d_slot, 0xFF, (size_t)e->kp.S * sizeof(uint32_t), e->stream));
      MK_HIP(e, hipMemsetAsync(e->d_dirty_acc, 0, (size_t)e->acc_words * 4, e->stream));
      MK_HIP(e, hipMemsetAsync(e->d_dirty_slot, 0, (size_t)e->slot_words * 4, e->stream));
      e->tables_tracked = true;
      e->slot_clean = true;
    }
  }
  { /* front table, counters and the FASTA stream's state in one launch */
    const unsigned long long n16 = clear_front ? (unsigned long long)e->front_slots : 0ull;
    unsigned long long blocks = (n16 + 1023u) / 1024u; /* four 16-byte stores a thread */
    if (blocks > (unsigned long long)e->num_cu * 8u) blocks = (unsigned long long)e->num_cu * 8u;
    if (blocks == 0) blocks = 1;
    const bool fa = e->fa_used && e->d_fa_state;
    hipLaunchKernelGGL(mk_begin_clear_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, (uint4 *)e->d_front, n16, e->d_counters,
                       (uint32_t *)e->d_fa_state, fa ? (uint32_t)(sizeof(mk_fa_state) / 4u) : 0u);
    MK_HIP(e, hipGetLastError());
    e->counter0_used = false;
  }
  if (e->profiling) { MK_HIP(e, hipEventRecord(ev.b, e->stream)); e->ev_clear.push_back(ev); }
  e->region_open = false; /* rows staged for a sketch that was never finished are dropped with it */
  e->mode = mode;
  e->min_occ = 1;
  e->begun = true;
  e->compacted = false;
  e->count_queued = false;
  e->D = 0;
  e->fa_used = false; e->fa_final = false; e->fa_tail = 0; e->fa_rows_done = 0; e->fa_pitch = 0;
  return MK_OK;
}

/* ---- scan launch -------------------------------------------------------------------------------------- */
template <int K, int SK, bool V, int T, int NP, bool OP>
static hipError_t mk_launch_scan_t(mk_engine *e, const mk_scan_args &a, dim3 grid, size_t lds, hipStream_t s) {
  /* the dynamic-LDS limit is a per-device attribute of the kernel.  What this engine has asked for is remembered in the
   * engine (one engine = one device; calls on one engine are serialised by the caller), so engines driven from
   * different host threads share no state here; asking again for another engine on the same device is harmless. */
  const void *fn = (const void *)mk_scan_kernel<K, SK, V, T, NP, OP>;
  size_t *granted = nullptr;
  for (auto &g : e->lds_granted) if (g.first == fn) granted = &g.second;
  if (!granted) { e->lds_granted.emplace_back(fn, 0); granted = &e->lds_granted.back().second; }
  if (lds > *granted) {
    hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (r != hipSuccess) return r;
    *granted = lds;
  }
  hipLaunchKernelGGL((mk_scan_kernel<K, SK, V, T, NP, OP>), grid, dim3(T), lds, s, a);
  return hipGetLastError();
}
template <int K, int SK, bool V>
static hipError_t mk_launch_scan_k(mk_engine *e, int threads, bool onepass, const mk_scan_args &a, dim3 grid, size_t lds, hipStream_t s) {
  /* piece registers: 16-byte path CB <= 80 -> 5 pieces, CB <= 128 -> 8; 4-byte path up to 32 */
  constexpr int NPBIG = V ? MK_MAX_PIECES : MK_MAX_CB / 4;
  constexpr int NPSMALL = V ? 5 : 20;
  const bool small = a.ppr <= (uint32_t)NPSMALL;
  if constexpr (V) {
    if (onepass && small) /* one-pass staging: 2 x 5 pieces live */
      return threads >= 1024 ? mk_launch_scan_t<K, SK, V, 1024, NPSMALL, true>(e, a, grid, lds, s)
             : threads >= 768 ? mk_launch_scan_t<K, SK, V, 768, NPSMALL, true>(e, a, grid, lds, s)
                              : mk_launch_scan_t<K, SK, V, 512, NPSMALL, true>(e, a, grid, lds, s);
  }
  switch (threads) {
    case 1024: return small ? mk_launch_scan_t<K, SK, V, 1024, NPSMALL, false>(e, a, grid, lds, s) : mk_launch_scan_t<K, SK, V, 1024, NPBIG, false>(e, a, grid, lds, s);
    case 768: return small ? mk_launch_scan_t<K, SK, V, 768, NPSMALL, false>(e, a, grid, lds, s) : mk_launch_scan_t<K, SK, V, 768, NPBIG, false>(e, a, grid, lds, s);
    default: return small ? mk_launch_scan_t<K, SK, V, 512, NPSMALL, false>(e, a, grid, lds, s) : mk_launch_scan_t<K, SK, V, 512, NPBIG, false>(e, a, grid, lds, s);
  }
}

/* stride: bytes staged per row.  pitch: address step between rows (== stride for rows side by side; the overlapping virtual
 * rows of a base stream step by less).  rowlen: 0, or the index at which a virtual row stops.  nreads_dev: NULL, or the row
 * count in device memory (nreads is then an upper bound the launch is sized for). */
static int mk_launch_scan_ex(mk_engine *e, const uint8_t *rows_dev, uint32_t stride, uint32_t pitch, uint32_t rowlen, uint64_t nreads,
                             const unsigned long long *nreads_dev, uint64_t first_ord) {
  if (nreads == 0) return MK_OK;
  if (nreads >= (1ull << 31)) return mk_fail(e, MK_ERR_ARG, "scan launch of %llu reads: the caller splits pushes below 2^31", (unsigned long long)nreads);
  const bool wide = (stride & MK_ROWS_WIDE) != 0;     /* wide packed rows (batches of rows only): 240 bases in the 64 bytes */
  const bool packed = (stride & (MK_ROWS_PACKED | MK_ROWS_WIDE)) != 0; /* 64-byte packed rows: mk_scan_packed_kernel, no column blocks, no LDS tiles */
  if (packed) { stride &= ~(MK_ROWS_PACKED | MK_ROWS_WIDE); pitch = stride; }
  mk_scan_args a{};
  a.rows = rows_dev; a.nreads = nreads; a.first_ord = first_ord; a.stride = stride;
  a.pitch = pitch; a.rowlen = rowlen; a.nreads_dev = nreads_dev;
  const bool vec = (stride % 16u == 0) && (pitch % 16u == 0) && (((uintptr_t)rows_dev & 15u) == 0);
  /* column blocks: fewest blocks of at most max_cb bytes, equal width, 16-byte (vec) / 4-byte granular -- 8-byte where the
   * tuned kernels can run (they take whole 8-base windows: a 152-byte row is 80 + 72, not 76 + 76, which fell to the generic
   * kernel at five times the time) */
  const bool tuned_sk = (e->P.subk == 6 && e->P.k >= 9 && e->P.k <= 11) || (e->P.subk == 5 && e->P.k == 11); /* the tuned instantiations */
  const bool tuned_geom = tuned_sk && stride % 8u == 0;
  const uint32_t g = vec ? 16u : (tuned_geom ? 8u : 4u);
  const uint32_t max_cb = e->tune_cb;
  a.ncb = (stride + max_cb - 1) / max_cb;
  a.CB = ((stride + a.ncb - 1) / a.ncb + g - 1) / g * g;
  if (a.CB > max_cb) { a.ncb++; a.CB = ((stride + a.ncb - 1) / a.ncb + g - 1) / g * g; }
  a.ncb = (stride + a.CB - 1) / a.CB;
  a.ppr = a.CB / (vec ? 16u : 4u); /* staging pieces: 16 or 4 bytes */
  a.ppr_inv = (1u << 20) / a.ppr + 1u;
  a.rowdw = (a.CB / 4u) | 1u;
  a.wave_lds_dwords = ((64u * a.rowdw + 1u) & ~1u) + 2u; /* +2: the unconditional word prefetch reads up to 2 dwords past a row */
  a.bm_words = 1u << e->bm_bits;
  /* tuned kernels: 24-bit inner substring (subk 6), k in {9,10,11}, every column block a whole number of 8-base pairs; they
   * keep the pair filter's 256-entry mask table in front of the filter */
  const int tuned_k = (tuned_sk && stride % 8u == 0 && a.CB % 8u == 0) ? e->P.k : 0;
  if (packed && !tuned_k) return mk_fail(e, MK_ERR_ARG, "packed rows need a geometry with a tuned scan kernel");
  a.mt_words = tuned_k && e->P.subk == 6 ? MK_ZMASK_WORDS : 0u;
  a.pair_subk = tuned_k ? (uint32_t)e->P.subk : 0u;
  if (tuned_k && e->P.subk == 5) a.bm_words = 16384u; /* the 2^19-bit membership bitmap of mk_build_xfilter, in both kernels */
  a.dimmask = (uint32_t)((1ull << (4 * e->P.subk)) - 1ull);
  a.accept = e->d_accept; a.n_accept = e->n_accept;
  a.shuf = e->d_shuf;
  a.accept_bits = e->d_accept_bits;
  a.kp = e->kp;
  a.tab = e->tab;
  a.cand = e->d_cand; a.cand_count = e->d_cand_count; a.cand_cap = e->cand_cap;
  a.batch = e->cur_batch;
  if (a.batch) a.tab.fr = nullptr; /* a batch has its own tables: nothing of this launch touches the engine's */
  /* one-pass staging needs exactly two equal column blocks (stride == 2*CB) on the 16-byte path */
  const bool onepass = e->tune_onepass && vec && a.ncb == 2 && stride == 2u * a.CB && a.ppr <= 5u;
  /* workgroup size: as many waves as the LDS budget (filter + per-wave tile and queue) admits */
  int threads = e->tune_threads;
  size_t lds = 0;
  const size_t fwords = tuned_k && e->P.subk == 6 && !packed ? (size_t)MK_ZF_WORDS : (size_t)a.bm_words;
  for (;; threads -= 256) {
    lds = ((size_t)a.mt_words + fwords + (size_t)(threads / 64) * a.wave_lds_dwords) * 4u;
    if (vec) lds += 2u * (a.ppr <= 5u ? 5u : (uint32_t)MK_MAX_PIECES) * 64u * 4u; /* staging offset table of the 16-byte kernels: [2*NPIECES][64] */
    if (packed) { threads = 1024; lds = ((size_t)a.mt_words + (size_t)a.bm_words) * 4u; } /* the filter, nothing else */
    if (lds <= 160u * 1024u || threads <= 512) break;
  }
  if (lds > 160u * 1024u) return mk_fail(e, MK_ERR_ARG, "scan: LDS budget exceeded (%zu bytes)", lds);
  const uint32_t waves = (uint32_t)threads / 64u;
  const uint64_t ntiles = (nreads + 63) / 64;
  uint64_t blocks = (ntiles + waves - 1) / waves;
  uint64_t wgs_per_cu = 1;
#ifdef MK_TUNING
  if (const char *t = getenv("MK_SCAN_WGS_PER_CU")) { const int v = atoi(t); if (v >= 1 && v * waves <= 16u) wgs_per_cu = (uint64_t)v; } /* (cand_slots = 16 waves a CU) */
#endif
  const uint64_t scan_cus = e->scan_stream ? (uint64_t)e->scan_cus : (uint64_t)e->num_cu; /* (MK_OPT_SPLIT_CUS: the scan queue's share) */
  if (blocks > scan_cus * wgs_per_cu) blocks = scan_cus * wgs_per_cu;
  dim3 grid((unsigned)blocks);
#ifdef MK_TUNING
  if (getenv("MK_DEBUG")) {
    static int once = 0;
    if (!once++) fprintf(stderr, "scan cfg: threads %d blocks %llu CB %u ncb %u rowdw %u lds %zu B onepass %d\n", threads,
                         (unsigned long long)blocks, a.CB, a.ncb, a.rowdw, lds, (int)onepass);
  }
#endif

  if (!a.batch) e->big_maybe_dirty = true; /* until a counter copy says otherwise (mk_check_counters) */
  /* the stream the scan kernel goes to: the engine's, or (MK_OPT_SPLIT_CUS) the scan queue, ordered with the engine's on both sides */
  hipStream_t ss = e->stream;
  if (e->scan_stream) {
    MK_HIP(e, hipEventRecord(e->ev_scan_pre, e->stream));
    MK_HIP(e, hipStreamWaitEvent(e->scan_stream, e->ev_scan_pre, 0));
    ss = e->scan_stream;
  }
  mk_evpair ev{};
  if (e->profiling) { ev = mk_ev_get(e); MK_HIP(e, hipEventRecord(ev.a, ss)); }
  hipError_t r;
  if (packed) {
    auto launch_packed = [&](auto kern) -> hipError_t {
      const void *fn = (const void *)kern;
      size_t *granted = nullptr;
      for (auto &g : e->lds_granted) if (g.first == fn) granted = &g.second;
      if (!granted) { e->lds_granted.emplace_back(fn, 0); granted = &e->lds_granted.back().second; }
      if (lds > *granted) {
        const hipError_t rr = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rr != hipSuccess) return rr;
        *granted = lds;
      }
      hipLaunchKernelGGL(kern, grid, dim3(1024), lds, ss, a);
      return hipGetLastError();
    };
    if (wide) switch (tuned_k * 10 + e->P.subk) {
      case 116: r = launch_packed(mk_scan_packed_kernel<11, 6, true>); break;
      case 106: r = launch_packed(mk_scan_packed_kernel<10, 6, true>); break;
      case 96: r = launch_packed(mk_scan_packed_kernel<9, 6, true>); break;
      default: r = launch_packed(mk_scan_packed_kernel<11, 5, true>); break;
    } else switch (tuned_k * 10 + e->P.subk) {
      case 116: r = launch_packed(mk_scan_packed_kernel<11, 6>); break;
      case 106: r = launch_packed(mk_scan_packed_kernel<10, 6>); break;
      case 96: r = launch_packed(mk_scan_packed_kernel<9, 6>); break;
      default: r = launch_packed(mk_scan_packed_kernel<11, 5>); break;
    }
  } else
  switch (tuned_k * 10 + (tuned_k ? e->P.subk : 0)) {
    case 116: r = vec ? mk_launch_scan_k<11, 6, true>(e, threads, onepass, a, grid, lds, ss) : mk_launch_scan_k<11, 6, false>(e, threads, false, a, grid, lds, ss); break;
    case 106: r = vec ? mk_launch_scan_k<10, 6, true>(e, threads, onepass, a, grid, lds, ss) : mk_launch_scan_k<10, 6, false>(e, threads, false, a, grid, lds, ss); break;
    case 96: r = vec ? mk_launch_scan_k<9, 6, true>(e, threads, onepass, a, grid, lds, ss) : mk_launch_scan_k<9, 6, false>(e, threads, false, a, grid, lds, ss); break;
    case 115: r = vec ? mk_launch_scan_k<11, 5, true>(e, threads, onepass, a, grid, lds, ss) : mk_launch_scan_k<11, 5, false>(e, threads, false, a, grid, lds, ss); break;
    default: r = vec ? mk_launch_scan_k<0, 0, true>(e, threads, onepass, a, grid, lds, ss) : mk_launch_scan_k<0, 0, false>(e, threads, false, a, grid, lds, ss); break;
  }
  if (r != hipSuccess) return mk_fail(e, MK_ERR_HIP, "scan launch: %s", hipGetErrorString(r));
  if (e->profiling) {
    MK_HIP(e, hipEventRecord(ev.b, ss));
    e->ev_scan.push_back(ev);
    e->prof_rows += nreads;
    e->prof_bytes += nreads * stride;
  }
  if (e->scan_stream) {
    MK_HIP(e, hipEventRecord(e->ev_scan_post, ss));
    if (e->tail && e->begun && !a.batch && e->stream == e->split_stream) {
      /* MK_BEGIN_NOTHING_FOLLOWS: no other engine's scan will want the device behind this one -- the engine's unmasked queue takes over for the
       * rest of the sketch (resolve with a workgroup on every CU, compaction, ...), ordered behind what the second queue has done so far */
      MK_HIP(e, hipEventRecord(e->ev_scan_pre, e->split_stream));
      MK_HIP(e, hipStreamWaitEvent(e->own_stream, e->ev_scan_pre, 0));
      e->stream = e->own_stream;
      e->tail_active = true;
    }
    MK_HIP(e, hipStreamWaitEvent(e->stream, e->ev_scan_post, 0));
  }
  /* resolve the appended candidates: canonical k-mer, exact .shuf check, upsert */
  {
    const uint32_t used_slots = (uint32_t)blocks * waves;
    mk_evpair ev2{};
    if (e->profiling) { ev2 = mk_ev_get(e); MK_HIP(e, hipEventRecord(ev2.a, e->stream)); }
    /* one resolve workgroup per CU (LDS: the exact filter + two rings per wave); a resolve wave owns whole slots */
    const uint32_t rwgs = (used_slots + MK_RESOLVE_THREADS / 64 - 1) / (MK_RESOLVE_THREADS / 64);
    const uint32_t rcus = e->scan_stream && e->stream == e->split_stream ? (uint32_t)e->split_cus : (uint32_t)e->num_cu; /* one workgroup a CU of ITS queue */
    uint32_t rgrid = rwgs < rcus ? rwgs : rcus;
#ifdef MK_TUNING
    if (const char *t = getenv("MK_RESOLVE_GRID")) { const uint32_t g = (uint32_t)atoi(t); if (g && g <= used_slots) rgrid = g; }
#endif
    const size_t rlds = (size_t)a.bm_words * 4u + (size_t)(MK_RESOLVE_THREADS / 64) * 2u * MK_RQ_CAP * sizeof(uint4);
    {
      const void *fn = (const void *)mk_resolve_kernel;
      size_t *granted = nullptr;
      for (auto &g : e->lds_granted) if (g.first == fn) granted = &g.second;
      if (!granted) { e->lds_granted.emplace_back(fn, 0); granted = &e->lds_granted.back().second; }
      if (rlds > *granted) {
        MK_HIP(e, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
        *granted = rlds;
      }
    }
    hipLaunchKernelGGL(mk_resolve_kernel, dim3(rgrid), dim3(MK_RESOLVE_THREADS), rlds, e->stream, a, used_slots);
    MK_HIP(e, hipGetLastError());
#ifdef MK_TUNING
    if (getenv("MK_COUNT_RECORDS")) { /* experiment builds: the candidate records this launch handed to the resolve kernel (waits for both kernels) */
      static int shown = 0;
      if (shown++ < 3) {
        std::vector<uint32_t> h(used_slots);
        MK_HIP(e, hipStreamSynchronize(e->stream));
        MK_HIP(e, hipMemcpy(h.data(), e->d_cand_count, (size_t)used_slots * 4, hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (uint32_t v : h) sum += v;
        fprintf(stderr, "[scan records] %llu reads -> %llu candidate records (%.4f a read, %.3f %% of the 8-base windows of 150-base reads)\n",
                (unsigned long long)nreads, sum, (double)sum / (double)nreads, 100.0 * (double)sum / ((double)nreads * 19.0));
      }
    }
#endif
    if (e->profiling) { MK_HIP(e, hipEventRecord(ev2.b, e->stream)); e->ev_resolve.push_back(ev2); }
  }
  if (!a.batch) { e->compacted = false; e->count_queued = false; }
  return MK_OK;
}

static int mk_launch_scan(mk_engine *e, const uint8_t *rows_dev, uint32_t stride, uint64_t nreads, uint64_t first_ord) {
  return mk_launch_scan_ex(e, rows_dev, stride, stride, 0u, nreads, nullptr, first_ord);
}

static int mk_check_push(mk_engine *e, const void *rows, uint32_t stride) {
  if (!e) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "push before mk_sketch_begin");
  if (stride & MK_ROWS_PACKED) { /* 64-byte packed rows: the tuned geometries only (their loop takes eight bases in 16 bits) */
    if (stride != (MK_PACKED_PITCH | MK_ROWS_PACKED)) return mk_fail(e, MK_ERR_ARG, "push: packed rows have a pitch of %u bytes", MK_PACKED_PITCH);
    if (!mk_params_packed_ok(&e->P)) return mk_fail(e, MK_ERR_ARG, "push: packed rows need a geometry with a tuned scan kernel (mk_params_packed_ok)");
    if (!rows) return mk_fail(e, MK_ERR_ARG, "push: rows == NULL");
    if ((uintptr_t)rows & 15u) return mk_fail(e, MK_ERR_ARG, "push: packed rows must be 16-byte aligned");
    return MK_OK;
  }
  if (!rows || stride < 4 || stride > 4096 || (stride & 3u)) return mk_fail(e, MK_ERR_ARG, "push: stride must be a multiple of 4 in 4..4096");
  return MK_OK;
}

extern "C" int mk_sketch_push_reads_device(mk_engine *e, const uint8_t *rows_dev, uint32_t stride, uint64_t nreads,
                                           uint64_t first_read_ordinal) {
  int rc = mk_check_push(e, rows_dev, stride);
  if (rc) return rc;
  MK_HIP(e, hipSetDevice(e->device));
  if ((first_read_ordinal + nreads) >> 51) return mk_fail(e, MK_ERR_ARG, "read ordinal too large");
  /* one launch per at most `per` reads: every scan wave then sees few enough tiles that its candidate append
   * buffer (cand_cap entries, ~9-16 candidates per 64-read tile at the usual 1/4096 accept rate) does not
   * overflow into the slow inline path */
  const uint64_t waves = (uint64_t)e->num_cu * (uint64_t)(e->tune_threads / 64);
  uint64_t per = (uint64_t)(e->cand_cap / 32u) * 64u * waves;
  if (per < 64u * waves) per = 64u * waves;
  if (per > (1ull << 30)) per = 1ull << 30; /* 32-bit tile and row indices inside a launch */
  for (uint64_t done = 0; done < nreads; done += per) {
    const uint64_t n = nreads - done < per ? nreads - done : per;
    int rc2 = mk_launch_scan(e, rows_dev + done * (uint64_t)(stride & ~MK_ROWS_PACKED), stride, n, first_read_ordinal + done);
    if (rc2) return rc2;
  }
  return MK_OK;
}

/* scan what has been copied into the open staging region (one launch), and close it */
static int mk_flush_region(mk_engine *e) {
  if (!e->region_open) return MK_OK;
  const int b = e->stage_cur;
  e->region_open = false;
  e->stage_cur = (b + 1) % MK_REGIONS;
  if (e->region_rows == 0) return MK_OK;
  const bool two_streams = e->stream != e->copy_stream; /* same stream: program order is the dependency */
  if (two_streams) {
    MK_HIP(e, hipEventRecord(e->ev_copied[b], e->copy_stream));
    MK_HIP(e, hipStreamWaitEvent(e->stream, e->ev_copied[b], 0));
  }
  int rc = mk_launch_scan(e, e->d_stage[b], e->region_stride, e->region_rows, e->region_first_ord);
  if (rc) return rc;
  if (two_streams) MK_HIP(e, hipEventRecord(e->ev_scanned[b], e->stream));
  e->stage_two_streams[b] = two_streams;
  return MK_OK;
}

extern "C" int mk_sketch_push_reads_async(mk_engine *e, const uint8_t *rows, uint32_t stride, uint64_t nreads,
                                          uint64_t first_read_ordinal, uint64_t *ticket) {
  int rc = mk_check_push(e, rows, stride);
  if (rc) return rc;
  if (!ticket) return MK_ERR_ARG;
  MK_HIP(e, hipSetDevice(e->device));
  if ((first_read_ordinal + nreads) >> 51) return mk_fail(e, MK_ERR_ARG, "read ordinal too large");
#ifdef MK_TUNING
  double tick_ = mk_tick_now();
  const bool first_push_ = !e->d_stage[0];
#endif
  if (e->direct_host) {
    /* rows in pinned (hipHostMalloc / registered) memory are mapped into the device: the scan kernel can read them over
     * PCIe itself, no staging copy.  The ticket then fires when that scan is done with the rows. */
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, rows) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
      rc = mk_flush_region(e);
      if (rc) return rc;
      rc = mk_launch_scan(e, (const uint8_t *)attr.devicePointer, stride, nreads, first_read_ordinal);
      if (rc) return rc;
      const uint64_t t = e->tickets_issued;
      hipEvent_t ev = e->ev_ticket[t % MK_TICKETS];
      if (t >= MK_TICKETS) MK_HIP(e, hipEventSynchronize(ev));
      MK_HIP(e, hipEventRecord(ev, e->stream));
      e->tickets_issued = t + 1;
      *ticket = t;
      return MK_OK;
    }
    (void)hipGetLastError(); /* ordinary memory: the staged path */
  }
  if (!e->d_stage[0]) {
    e->stage_bytes = MK_REGION_BYTES;
    for (int i = 0; i < MK_REGIONS; i++) MK_HIP(e, mk_dev_alloc(&e->d_stage[i], e->stage_bytes));
  }
#ifdef MK_TUNING
  if (first_push_) MK_TICK("push1: staging alloc");
#endif
  const uint32_t pitch = stride & ~MK_ROWS_PACKED; /* bytes per row; `stride` keeps the row format in its top bit */
  for (uint64_t done = 0; done < nreads;) {
    const uint64_t ord = first_read_ordinal + done;
    /* rows that do not continue the open region (another stride, a gap in the ordinals) start a new one */
    if (e->region_open && (stride != e->region_stride || ord != e->region_first_ord + e->region_rows ||
                           e->region_fill + pitch > e->stage_bytes)) {
      rc = mk_flush_region(e);
      if (rc) return rc;
    }
    if (!e->region_open) {
      /* copy stream: the scan that last read this region must be finished before it is overwritten */
      if (e->stage_two_streams[e->stage_cur]) MK_HIP(e, hipStreamWaitEvent(e->copy_stream, e->ev_scanned[e->stage_cur], 0));
      e->region_open = true;
      e->region_fill = 0; e->region_rows = 0;
      e->region_stride = stride; e->region_first_ord = ord;
    }
    const uint64_t room = (e->stage_bytes - e->region_fill) / pitch;
    const uint64_t n = nreads - done < room ? nreads - done : room;
    MK_HIP(e, hipMemcpyAsync(e->d_stage[e->stage_cur] + e->region_fill, rows + done * pitch, n * pitch, hipMemcpyHostToDevice,
                             e->copy_stream));
#ifdef MK_TUNING
    if (first_push_ && done == 0) MK_TICK("push1: first memcpyAsync");
#endif
    e->region_fill += n * pitch;
    e->region_rows += n;
    done += n;
    if (e->region_fill + pitch > e->stage_bytes) { /* full: scan it while the next one fills */
      rc = mk_flush_region(e);
      if (rc) return rc;
    }
  }
  /* the ticket fires when the last copy out of `rows` is done.  A ring slot is reused only after its previous ticket
   * has been waited for here, so a caller may keep up to MK_TICKETS pushes in flight without ever waiting itself. */
  const uint64_t t = e->tickets_issued;
  hipEvent_t ev = e->ev_ticket[t % MK_TICKETS];
  if (t >= MK_TICKETS) MK_HIP(e, hipEventSynchronize(ev));
  MK_HIP(e, hipEventRecord(ev, e->copy_stream));
#ifdef MK_TUNING
  if (first_push_) MK_TICK("push1: ticket record");
#endif
  e->tickets_issued = t + 1;
  *ticket = t;
  return MK_OK;
}

extern "C" int mk_sketch_push_wait(mk_engine *e, uint64_t ticket) {
  if (!e) return MK_ERR_ARG;
  if (ticket >= e->tickets_issued) return mk_fail(e, MK_ERR_ARG, "mk_sketch_push_wait: ticket %llu was never issued", (unsigned long long)ticket);
  if (ticket + MK_TICKETS < e->tickets_issued) return MK_OK; /* its ring slot has been reused: waited for at that point */
  MK_HIP(e, hipSetDevice(e->device));
  MK_HIP(e, hipEventSynchronize(e->ev_ticket[ticket % MK_TICKETS]));
  return MK_OK;
}

extern "C" int mk_sketch_push_reads(mk_engine *e, const uint8_t *rows, uint32_t stride, uint64_t nreads,
                                    uint64_t first_read_ordinal) {
  uint64_t t = 0;
  int rc = mk_sketch_push_reads_async(e, rows, stride, nreads, first_read_ordinal, &t);
  if (rc) return rc;
  return mk_sketch_push_wait(e, t); /* caller's buffer is free again; the last scan may still run */
}

/* ---- FASTA text straight to the device (mk_sketch_push_stream) ------------------------------------------
 * fasta2co() / uniq_fasta2co() read the file through a 64 KiB window and walk it byte by byte (iseq2comem.c:224-279).  Here
 * the raw bytes are copied to HBM, three kernels (mk_stream.hip.h) drop line ends and header lines exactly like that walk,
 * and the scan kernel reads overlapping virtual rows out of the resulting base stream.  Text may arrive in pieces; every
 * piece but the last has final == 0 (rows whose last byte is not known yet wait in the stream buffer; one small
 * synchronisation per non-final piece tells the host how many).  The sketch is identical to the one the host windows
 * (mk_fasta_window + mk_sketch_push_reads) give: the k-mers and their order are the stream's, not the rows'. */
/* address step of the virtual rows: a multiple of 16, not of 128 (see MK_ROW_PITCH).  Long rows cost the least overlap (TL - 1 of
 * 528 bytes are scanned twice); a genome-sized text gets SHORT rows instead: a 4 MB genome is 118 tiles of 528-byte rows -- 118
 * busy waves of the 4096 the chip holds, each walking 35 000 bases one after the other (38 us for the tuned kernel, 168 us for the
 * generic one) -- and 560 tiles of 112-byte rows.  Chosen at the first piece of a sketch and kept for it. */
#define MK_FA_PITCH 528u
#define MK_FA_PITCH_SMALL 112u
#define MK_FA_SMALL_BYTES ((uint64_t)32 << 20)

static int mk_fa_reserve(mk_engine *e, size_t n) {
  if (!e->d_fa_state) {
    MK_HIP(e, mk_dev_alloc(&e->d_fa_state, sizeof(mk_fa_state)));
    MK_HIP(e, hipMemsetAsync(e->d_fa_state, 0, sizeof(mk_fa_state), e->stream)); /* (never the default stream: see mk_poison.hip.h) */
    MK_HIP(e, mk_pin_alloc((void **)&e->h_fa_state, sizeof(mk_fa_state), hipHostMallocDefault));
  }
  /* whole segments: the last wave of a piece loads MK_FA_SEG bytes from its segment's start (mk_fa_stage), so the text buffer holds
   * the piece rounded up to a segment */
  const size_t nround = ((n + MK_FA_SEG - 1) / MK_FA_SEG) * MK_FA_SEG;
  if (nround > e->text_cap || !e->d_stream) { /* (also the first piece of all being empty: the buffers must exist) */
    MK_HIP(e, hipStreamSynchronize(e->stream)); /* kernels of earlier pieces may still read the old buffers */
    const size_t cap = n + n / 4 + ((size_t)1 << 20);
    uint8_t *nt = nullptr, *ns = nullptr, *ntmp = nullptr;
    /* stream: what is carried (less than one row and one step) + the new text + a row's width of slack behind it */
    const size_t scap = cap + 8192;
    MK_HIP(e, mk_dev_alloc(&nt, cap));
    MK_HIP(e, mk_dev_alloc(&ns, scap));
    MK_HIP(e, mk_dev_alloc(&ntmp, 8192));
    MK_HIP(e, hipMemsetAsync(ns, 0, scap, e->stream));
    if (e->d_stream && e->fa_tail) MK_HIP(e, hipMemcpyAsync(ns, e->d_stream, e->fa_tail, hipMemcpyDeviceToDevice, e->stream));
    MK_HIP(e, hipStreamSynchronize(e->stream)); /* the old buffers go now */
    hipFree(e->d_text); hipFree(e->d_stream); hipFree(e->d_stream_tmp);
    e->d_text = nt; e->d_stream = ns; e->d_stream_tmp = ntmp;
    e->text_cap = cap; e->stream_cap = scap;
  }
  const size_t nseg = (n + MK_FA_SEG - 1) / MK_FA_SEG;
  if (nseg > e->fa_sum_cap) {
    MK_HIP(e, hipStreamSynchronize(e->stream));
    hipFree(e->d_fa_sum);
    e->d_fa_sum = nullptr; e->fa_sum_cap = 0;
    MK_HIP(e, mk_dev_alloc(&e->d_fa_sum, (nseg + nseg / 4 + 256) * sizeof(mk_fa_sum)));
    e->fa_sum_cap = nseg + nseg / 4 + 256;
  }
  return MK_OK;
}

extern "C" int mk_sketch_push_stream(mk_engine *e, const uint8_t *text, uint64_t n, int final) {
  if (!e || (!text && n)) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "push before mk_sketch_begin");
  if (e->fa_final) return mk_fail(e, MK_ERR_STATE, "mk_sketch_push_stream: the stream has been closed (final) for this sketch");
  if (n >= (1ull << 31)) return mk_fail(e, MK_ERR_ARG, "mk_sketch_push_stream: at most 2^31 - 1 bytes per call");
  if (e->P.TL + MK_FA_PITCH > 4000u) return mk_fail(e, MK_ERR_ARG, "mk_sketch_push_stream: k-mer too long for the stream rows");
  MK_HIP(e, hipSetDevice(e->device));
  { int rc = mk_flush_region(e); if (rc) return rc; } /* rows pushed the other way keep their place in the order of the pushes */
  int rc = mk_fa_reserve(e, (size_t)n);
  if (rc) return rc;
  if (!e->fa_pitch) e->fa_pitch = (final && n <= MK_FA_SMALL_BYTES) ? MK_FA_PITCH_SMALL : MK_FA_PITCH; /* a whole small text at once */
  const uint32_t TL = (uint32_t)e->P.TL, pitch = e->fa_pitch, rowlen = pitch + TL - 1u;
  const uint32_t width = (rowlen + 1u + 15u) & ~15u; /* staged bytes per row: through the cut, whole 16-byte pieces */
  e->fa_used = true;
  if (n) {
    /* pinned text is copied asynchronously (the caller keeps it untouched until the next waiting call); pageable text may be
     * reused as soon as this call returns, so the copy is waited for */
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, text) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    MK_HIP(e, hipMemcpyAsync(e->d_text, text, (size_t)n, hipMemcpyHostToDevice, e->stream));
    if (!pinned) MK_HIP(e, hipStreamSynchronize(e->stream));
  }
  const uint64_t nseg = (n + MK_FA_SEG - 1) / MK_FA_SEG;
  if (nseg) hipLaunchKernelGGL(mk_fa_summary_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, e->stream, (const uint8_t *)e->d_text, n, e->d_fa_sum);
  hipLaunchKernelGGL(mk_fa_scan_kernel, dim3(1), dim3(1024), 0, e->stream, e->d_fa_sum, nseg, e->d_fa_state, e->d_stream, pitch, rowlen, TL,
                     final ? 1 : 0, e->tab.err);
  if (nseg) hipLaunchKernelGGL(mk_fa_emit_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, e->stream, (const uint8_t *)e->d_text, n,
                               (const mk_fa_sum *)e->d_fa_sum, e->d_stream, (unsigned long long)e->fa_tail);
  MK_HIP(e, hipGetLastError());
  /* rows the launch is sized for: what the stream could hold at most */
  const uint64_t ub_len = e->fa_tail + n;
  uint64_t ub_rows = 0;
  if (final) { if (ub_len >= TL) ub_rows = (ub_len - (TL - 1u) + pitch - 1u) / pitch; }
  else if (ub_len >= rowlen) ub_rows = (ub_len - rowlen) / pitch + 1u;
  if (ub_rows) {
    rc = mk_launch_scan_ex(e, e->d_stream, width, pitch, rowlen, ub_rows, &e->d_fa_state->nrows, e->fa_rows_done);
    if (rc) return rc;
  }
  if (final) { e->fa_final = true; return MK_OK; }
  /* the tail that no complete row covers yet goes to the front of the buffer; how long it is comes back to the host */
  hipLaunchKernelGGL(mk_fa_shift_kernel, dim3(8), dim3(1024), 0, e->stream, e->d_stream, e->d_stream_tmp, e->d_fa_state, pitch, 0);
  hipLaunchKernelGGL(mk_fa_shift_kernel, dim3(8), dim3(1024), 0, e->stream, e->d_stream, e->d_stream_tmp, e->d_fa_state, pitch, 1);
  MK_HIP(e, hipMemcpyAsync(e->h_fa_state, e->d_fa_state, sizeof(mk_fa_state), hipMemcpyDeviceToHost, e->stream));
  hipLaunchKernelGGL(mk_fa_shift_done_kernel, dim3(1), dim3(1), 0, e->stream, e->d_fa_state, pitch);
  MK_HIP(e, hipGetLastError());
  MK_HIP(e, hipStreamSynchronize(e->stream));
  const uint64_t taken = e->h_fa_state->nrows, len = e->h_fa_state->len;
  e->fa_rows_done += taken;
  e->fa_tail = len > taken * pitch ? len - taken * pitch : 0;
  if (e->fa_tail > 8192) return mk_fail(e, MK_ERR_HIP, "mk_sketch_push_stream: stream tail of %llu bytes", (unsigned long long)e->fa_tail);
  return MK_OK;
}

/* ---- compaction / partials --------------------------------------------------------------------------- */
/* table -> distinct-key list on the device; the number of keys lands in d_counters[0].  No host synchronisation. */
static int mk_compact_launch(mk_engine *e) {
  if (e->res_pending) /* the key list of the previous sketch is still being laid out and dumped on the side stream */
    return mk_fail(e, MK_ERR_STATE, "the result of mk_sketch_finish_begin has not been taken yet (mk_sketch_finish_end)");
  { int rc = mk_flush_region(e); if (rc) return rc; } /* rows copied but not scanned yet */
  /* the key counter: zero since mk_sketch_begin unless a compaction has run on this sketch already (partial counts, imports) */
  if (e->counter0_used) MK_HIP(e, hipMemsetAsync(e->d_counters, 0, sizeof(unsigned long long), e->stream));
  e->counter0_used = true;
  /* co[n]=0 stays "empty" in the FASTA set flavours (iseq2comem.c:300-302); the FASTQ slot words carry a count
   * field, so key 0 is an ordinary key there (:398-399, :704-705) */
  const int drop0 = e->mode == MK_MODE_SET || e->mode == MK_MODE_UNIQ_SET;
  const unsigned blocks = (unsigned)(e->num_cu * 2);
  if (e->sparse) { /* only the blocks somebody installed a key in */
    const uint32_t *big_used = nullptr;
    if (e->tab.fr) {
      /* big table untouched (the usual case for a genome): list the front table, the kernels below return at once.  Otherwise
       * the front table is folded into the big one (its dirty blocks are marked by the fold) and that is listed */
      big_used = e->front.state + 1;
      mk_table ft = e->tab;
      ft.kc = e->front.kc1; ft.ordinv = e->front.ordinv1;
      ft.dirty = nullptr;
      hipLaunchKernelGGL(mk_compact_kernel<MK_SPARSE_BLOCK>, dim3(blocks), dim3(MK_COMPACT_THREADS), 0, e->stream, ft, (uint32_t)e->front_slots,
                         e->dist, e->d_counters, drop0, (const uint32_t *)nullptr, (const uint32_t *)nullptr, big_used, 0u);
      hipLaunchKernelGGL(mk_front_fold_kernel, dim3(blocks), dim3(1024), 0, e->stream, e->tab, e->kp.S);
    }
    hipLaunchKernelGGL(mk_dirty_list_kernel, dim3(1), dim3(1024), 0, e->stream, e->d_dirty_acc, e->acc_words, e->d_list_acc, e->d_nlist + 2, 0, big_used);
    hipLaunchKernelGGL(mk_compact_kernel<MK_SPARSE_BLOCK>, dim3(blocks), dim3(MK_COMPACT_THREADS), 0, e->stream, e->tab, e->kp.S, e->dist,
                       e->d_counters, drop0, (const uint32_t *)e->d_list_acc, (const uint32_t *)(e->d_nlist + 2), big_used, 1u);
  } else {
    if (e->tab.fr) {
      /* big table empty (state[1] == 0, the common case): list the front table, the other two kernels return at once.
       * Otherwise fold the front table into the big one and list that. */
      const uint32_t *big_used = e->front.state + 1;
      mk_table ft = e->tab;
      ft.kc = e->front.kc1; ft.ordinv = e->front.ordinv1;
      /* (a front table of 2^18 slots is 128 chunks of 2048: 128 busy waves.  512-slot chunks give the small ones 512) */
      if (e->front_slots <= (1ull << 20))
        hipLaunchKernelGGL(mk_compact_kernel<MK_SPARSE_BLOCK>, dim3(blocks), dim3(MK_COMPACT_THREADS), 0, e->stream, ft, (uint32_t)e->front_slots,
                           e->dist, e->d_counters, drop0, (const uint32_t *)nullptr, (const uint32_t *)nullptr, big_used, 0u);
      else
        hipLaunchKernelGGL(mk_compact_kernel<MK_COMPACT_CHUNK>, dim3(blocks), dim3(MK_COMPACT_THREADS), 0, e->stream, ft, (uint32_t)e->front_slots,
                           e->dist, e->d_counters, drop0, (const uint32_t *)nullptr, (const uint32_t *)nullptr, big_used, 0u);
      hipLaunchKernelGGL(mk_front_fold_kernel, dim3(blocks), dim3(1024), 0, e->stream, e->tab, e->kp.S);
      hipLaunchKernelGGL(mk_compact_kernel<MK_COMPACT_CHUNK>, dim3(blocks), dim3(MK_COMPACT_THREADS), 0, e->stream, e->tab, e->kp.S, e->dist,
                         e->d_counters, drop0, (const uint32_t *)nullptr, (const uint32_t *)nullptr, big_used, 1u);
    } else {
      hipLaunchKernelGGL(mk_compact_kernel<MK_COMPACT_CHUNK>, dim3(blocks), dim3(MK_COMPACT_THREADS), 0, e->stream, e->tab, e->kp.S, e->dist,
                         e->d_counters, drop0, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u);
    }
  }
  MK_HIP(e, hipGetLastError());
  return MK_OK;
}

/* most distinct keys a sketch may hold before the reference gives up */
static uint64_t mk_key_limit(const mk_engine *e) {
  /* fastq2co() tests `keycount > hashlimit` but never advances keycount (:404): only a full table stops it */
  return e->mode == MK_MODE_OCC_SET ? (uint64_t)e->kp.S - 1 : (uint64_t)e->P.hashlimit;
}

/* device error flags + key count -> return code (after the counters have been copied to h_counters) */
static int mk_check_counters(mk_engine *e) {
  const uint32_t errflags = (uint32_t)(e->h_counters[2] & 0xffffffffu);
  e->D = e->h_counters[0];
  if (e->tab.fr && (uint32_t)(e->h_counters[4] >> 32) == 0u && !e->region_open) e->big_maybe_dirty = false; /* state[1]: nobody used the big table */
  if (errflags & 4u) return mk_fail(e, MK_ERR_HIP, "scan kernel: LDS filter not at offset 0");
  if (errflags & 8u) return mk_fail(e, MK_ERR_FORMAT, "fasta2co(): can not find seqences head start from '>' (the stream ends inside a header line)");
  if ((errflags & 1u) || e->D > mk_key_limit(e))
    return mk_fail(e, MK_ERR_CROWDED, "the context space is too crowd (%llu distinct keys > limit %llu), try k=%d",
                   (unsigned long long)e->D, (unsigned long long)mk_key_limit(e), e->P.k + 1);
  if (errflags & 2u) return mk_fail(e, MK_ERR_HIP, "layout kernel did not converge");
  return MK_OK;
}

/* the compaction has counted e->D keys (counters on the host): a key list that starts small (sparse bookkeeping) may hold fewer.
 * Then: a larger list and the compaction once more -- the table is untouched by a compaction, the list is all it writes */
static int mk_dist_fit(mk_engine *e) {
  if (e->D <= e->dist.cap) return MK_OK;
  MK_HIP(e, hipStreamSynchronize(e->stream));
  uint64_t cap = e->D + e->D / 4 + 1024;
  if (cap > e->kp.S) cap = e->kp.S;
  int rc = mk_dist_reserve(e, cap);
  if (rc) return rc;
  rc = mk_compact_launch(e);
  if (rc) return rc;
  MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  return mk_check_counters(e);
}

/* compaction with the key count brought to the host (the multi-GPU export needs it there) */
static int mk_compact(mk_engine *e) {
  if (e->compacted && !e->region_open) return MK_OK;
  int rc = mk_compact_launch(e);
  if (rc) return rc;
  MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  rc = mk_check_counters(e);
  if (rc == MK_OK) rc = mk_dist_fit(e);
  if (rc) return rc;
  e->compacted = true;
  return MK_OK;
}

/* the compaction and the copy of its counters queued, nothing waited for: several engines (one per GPU) compact at the same
 * time when the caller starts them all and only then asks each for its count */
extern "C" int mk_partial_count_begin(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_count_begin before mk_sketch_begin");
  MK_HIP(e, hipSetDevice(e->device));
  if (e->compacted && !e->region_open) return MK_OK;
  int rc = mk_compact_launch(e);
  if (rc) return rc;
  MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  e->count_queued = true;
  return MK_OK;
}

extern "C" int mk_partial_count(mk_engine *e, uint64_t *n) {
  if (!e || !n) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_count before mk_sketch_begin");
  MK_HIP(e, hipSetDevice(e->device));
  int rc;
  if (e->count_queued) { /* mk_partial_count_begin has launched it */
    e->count_queued = false;
    MK_HIP(e, hipStreamSynchronize(e->stream));
    rc = mk_check_counters(e);
    if (rc == MK_OK) rc = mk_dist_fit(e);
    if (rc == MK_OK) e->compacted = true;
  } else {
    rc = mk_compact(e);
  }
  if (rc) return rc;
  *n = e->D;
  return MK_OK;
}

/* mk_partial_export without the final wait: the three copies are queued on the engine's stream (mk_engine_sync waits) */
extern "C" int mk_partial_export_async(mk_engine *e, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev, uint64_t capacity,
                                       uint64_t *n_out) {
  if (!e || !n_out) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_export before mk_sketch_begin");
  MK_HIP(e, hipSetDevice(e->device));
  uint64_t d = 0;
  int rc = mk_partial_count(e, &d);
  if (rc) return rc;
  *n_out = d;
  if (d > capacity) return mk_fail(e, MK_ERR_ARG, "partial_export: capacity %llu < %llu keys", (unsigned long long)capacity, (unsigned long long)d);
  if (d == 0) return MK_OK;
  if (!keys_dev || !counts_dev || !ords_dev) return MK_ERR_ARG;
  MK_HIP(e, hipMemcpyAsync(keys_dev, e->dist.key, d * 8, hipMemcpyDeviceToDevice, e->stream));
  MK_HIP(e, hipMemcpyAsync(counts_dev, e->dist.cnt, d * 4, hipMemcpyDeviceToDevice, e->stream));
  MK_HIP(e, hipMemcpyAsync(ords_dev, e->dist.ord, d * 8, hipMemcpyDeviceToDevice, e->stream));
  return MK_OK;
}

extern "C" int mk_partial_export(mk_engine *e, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev, uint64_t capacity,
                                 uint64_t *n_out) {
  if (!e || !n_out) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_export before mk_sketch_begin");
  int rc = mk_partial_export_async(e, keys_dev, counts_dev, ords_dev, capacity, n_out);
  if (rc) return rc;
  if (*n_out) MK_HIP(e, hipStreamSynchronize(e->stream));
  return MK_OK;
}

extern "C" int mk_partial_import(mk_engine *e, const uint64_t *keys_dev, const uint32_t *counts_dev, const uint64_t *ords_dev,
                                 uint64_t n) {
  if (!e) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_import before mk_sketch_begin");
  if (n == 0) return MK_OK;
  if (!keys_dev || !counts_dev || !ords_dev) return MK_ERR_ARG;
  MK_HIP(e, hipSetDevice(e->device));
  uint64_t blocks = (n + 1023) / 1024;
  if (blocks > (uint64_t)e->num_cu * 2) blocks = (uint64_t)e->num_cu * 2;
  e->big_maybe_dirty = true;
  hipLaunchKernelGGL(mk_import_kernel, dim3((unsigned)blocks), dim3(1024), 0, e->stream, e->tab, e->kp.S,
                     (const unsigned long long *)keys_dev, counts_dev, (const unsigned long long *)ords_dev, n);
  MK_HIP(e, hipGetLastError());
  e->compacted = false; e->count_queued = false;
  return MK_OK;
}

/* ---- the merge by key slices (include/metakssd_hip.h, "the same merge by key slices") --------------------- */
extern "C" int mk_partial_export_split_async(mk_engine *e, uint32_t nparts, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev,
                                             uint64_t capacity, uint64_t *part_counts, uint64_t *n_out) {
  if (!e || !n_out || !part_counts) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_export_split before mk_sketch_begin");
  if (nparts < 1 || nparts > MK_SPLIT_MAX) return mk_fail(e, MK_ERR_ARG, "partial_export_split: 1..%u parts", MK_SPLIT_MAX);
  MK_HIP(e, hipSetDevice(e->device));
  uint64_t d = 0;
  int rc = mk_partial_count(e, &d);
  if (rc) return rc;
  *n_out = d;
  for (uint32_t g = 0; g < nparts; g++) part_counts[g] = 0;
  if (d > capacity) return mk_fail(e, MK_ERR_ARG, "partial_export_split: capacity %llu < %llu keys", (unsigned long long)capacity, (unsigned long long)d);
  if (d == 0) return MK_OK;
  if (!keys_dev || !counts_dev || !ords_dev) return MK_ERR_ARG;
  if (!e->d_split) MK_HIP(e, mk_dev_alloc(&e->d_split, 2 * MK_SPLIT_MAX * sizeof(unsigned long long)));
  MK_HIP(e, hipMemsetAsync(e->d_split, 0, 2 * MK_SPLIT_MAX * sizeof(unsigned long long), e->stream));
  uint64_t blocks = (d + 1023) / 1024;
  if (blocks > (uint64_t)e->num_cu * 2) blocks = (uint64_t)e->num_cu * 2;
  hipLaunchKernelGGL(mk_split_count_kernel, dim3((unsigned)blocks), dim3(1024), 0, e->stream, e->dist, (const unsigned long long *)e->d_counters, nparts, e->d_split);
  hipLaunchKernelGGL(mk_split_offsets_kernel, dim3(1), dim3(1), 0, e->stream, (const unsigned long long *)e->d_split, e->d_split + MK_SPLIT_MAX, nparts);
  hipLaunchKernelGGL(mk_split_scatter_kernel, dim3((unsigned)blocks), dim3(1024), 0, e->stream, e->dist, (const unsigned long long *)e->d_counters, nparts,
                     e->d_split + MK_SPLIT_MAX, (unsigned long long *)keys_dev, counts_dev, (unsigned long long *)ords_dev, capacity);
  MK_HIP(e, hipGetLastError());
  MK_HIP(e, hipMemcpyAsync(part_counts, e->d_split, nparts * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  return MK_OK;
}

extern "C" int mk_partial_export_split(mk_engine *e, uint32_t nparts, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev,
                                       uint64_t capacity, uint64_t *part_counts, uint64_t *n_out) {
  int rc = mk_partial_export_split_async(e, nparts, keys_dev, counts_dev, ords_dev, capacity, part_counts, n_out);
  if (rc) return rc;
  MK_HIP(e, hipStreamSynchronize(e->stream));
  return MK_OK;
}

extern "C" int mk_partial_restart(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_restart before mk_sketch_begin");
  const int mode = e->mode;
  const uint32_t occ = e->min_occ;
  { int rc = mk_flush_region(e); if (rc) return rc; } /* (rows staged and not scanned yet belong to what the caller exported: scanned, then dropped) */
  int rc = mk_sketch_begin(e, mode);
  if (rc == MK_OK) e->min_occ = occ;
  return rc;
}

extern "C" int mk_partial_list_reserve(mk_engine *e, uint64_t n, uint64_t **keys_dev, uint32_t **counts_dev, uint64_t **ords_dev) {
  if (!e || !keys_dev || !counts_dev || !ords_dev) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_list_reserve before mk_sketch_begin");
  if (e->res_pending) return mk_fail(e, MK_ERR_STATE, "partial_list_reserve while the side stream works on the key list (mk_sketch_finish_end first)");
  MK_HIP(e, hipSetDevice(e->device));
  if (n > e->dist.cap || !e->dist.key) { /* grow, keeping what a compaction has listed */
    const uint64_t keep = e->compacted ? (e->D < e->dist.cap ? e->D : e->dist.cap) : 0;
    MK_HIP(e, hipStreamSynchronize(e->stream));
    const uint64_t cap = n + n / 8 + 1024;
    unsigned long long *nk = nullptr, *no = nullptr;
    uint32_t *nc = nullptr;
    MK_HIP(e, mk_dev_alloc(&nk, cap * 8));
    MK_HIP(e, mk_dev_alloc(&no, cap * 8));
    MK_HIP(e, mk_dev_alloc(&nc, cap * 4));
    if (keep) {
      MK_HIP(e, hipMemcpyAsync(nk, e->dist.key, keep * 8, hipMemcpyDeviceToDevice, e->stream));
      MK_HIP(e, hipMemcpyAsync(no, e->dist.ord, keep * 8, hipMemcpyDeviceToDevice, e->stream));
      MK_HIP(e, hipMemcpyAsync(nc, e->dist.cnt, keep * 4, hipMemcpyDeviceToDevice, e->stream));
      MK_HIP(e, hipStreamSynchronize(e->stream)); /* the old list goes now */
    }
    hipFree(e->dist.key); hipFree(e->dist.ord); hipFree(e->dist.cnt);
    e->dist.key = nk; e->dist.ord = no; e->dist.cnt = nc; e->dist.cap = cap;
  }
  *keys_dev = (uint64_t *)e->dist.key; *counts_dev = e->dist.cnt; *ords_dev = (uint64_t *)e->dist.ord;
  return MK_OK;
}

extern "C" int mk_partial_list_adopt(mk_engine *e, const uint64_t *keys_dev, const uint32_t *counts_dev, const uint64_t *ords_dev, uint64_t n,
                                     uint64_t offset) {
  if (!e) return MK_ERR_ARG;
  uint64_t *k = nullptr, *o = nullptr;
  uint32_t *c = nullptr;
  int rc = mk_partial_list_reserve(e, offset + n, &k, &c, &o);
  if (rc || n == 0) return rc;
  if (!keys_dev || !counts_dev || !ords_dev) return MK_ERR_ARG;
  MK_HIP(e, hipMemcpyAsync(k + offset, keys_dev, n * 8, hipMemcpyDeviceToDevice, e->stream));
  MK_HIP(e, hipMemcpyAsync(c + offset, counts_dev, n * 4, hipMemcpyDeviceToDevice, e->stream));
  MK_HIP(e, hipMemcpyAsync(o + offset, ords_dev, n * 8, hipMemcpyDeviceToDevice, e->stream));
  return MK_OK;
}

extern "C" int mk_partial_list_commit(mk_engine *e, uint64_t n) {
  if (!e) return MK_ERR_ARG;
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "partial_list_commit before mk_sketch_begin");
  if (n > e->dist.cap) return mk_fail(e, MK_ERR_ARG, "partial_list_commit: %llu entries, the list holds %llu (mk_partial_list_reserve)", (unsigned long long)n, (unsigned long long)e->dist.cap);
  MK_HIP(e, hipSetDevice(e->device));
  { int rc = mk_flush_region(e); if (rc) return rc; }
  hipLaunchKernelGGL(mk_set_counter_kernel, dim3(1), dim3(1), 0, e->stream, e->d_counters, (unsigned long long)n);
  MK_HIP(e, hipGetLastError());
  e->counter0_used = true;
  e->D = n;
  e->compacted = true; e->count_queued = false; /* the finish starts from this list: no compaction of the table */
  return MK_OK;
}

/* ---- finish: layout + dump ------------------------------------------------------------------------------ */
static int mk_result_capacity(mk_engine *e, uint64_t want) {
  if (want <= e->h_cap && e->h_ids) return MK_OK;
  if (e->h_ids) hipHostFree(e->h_ids);
  if (e->h_cnt) hipHostFree(e->h_cnt);
  e->h_ids = nullptr; e->h_cnt = nullptr; e->h_cap = 0;
  const uint64_t cap = want + want / 8 + 1024;
  MK_HIP(e, mk_pin_alloc((void **)&e->h_ids, cap * 4, hipHostMallocDefault));
  MK_HIP(e, mk_pin_alloc((void **)&e->h_cnt, cap * 2, hipHostMallocDefault));
  e->h_cap = cap;
  return MK_OK;
}

/* the ordered dump of the layout table into the (host-mapped) result arrays; with count_pass == false only the write
 * kernel runs again (after the result arrays have grown; the chunk offsets of the first pass are still valid) */
static int mk_launch_dump(mk_engine *e, bool count_pass, uint32_t *out_ids = nullptr, uint16_t *out_cnt = nullptr, uint64_t out_cap = 0,
                          hipStream_t st = nullptr, bool unset = false) {
  if (!out_ids) { out_ids = e->h_ids; out_cnt = e->h_cnt; out_cap = e->h_cap; } /* straight into the pinned host arrays */
  if (!st) st = e->stream;
  const int C = e->P.component_num;
  const bool koc = e->mode == MK_MODE_KOC;
  mk_dump_args da{};
  da.slot = e->d_slot; da.S = e->kp.S; da.d = e->dist;
  da.dirty_slot = e->sparse ? e->d_dirty_slot : nullptr;
  da.comp_num = (uint32_t)C; da.comp_code_bits = (uint32_t)e->P.comp_code_bits;
  da.cnt_lo = e->mode == MK_MODE_OCC_SET ? e->min_occ : 1u; /* write_fqco2file(): marked keys only (iseq2comem.c:611) */
  da.cnt_hi = e->mode == MK_MODE_UNIQ_SET ? 1u : 0xffffffffu;  /* uniq_fasta2co(): repeated keys dropped */
  da.nchunks = e->nchunks;
  da.out_cap = out_cap;
  da.unset = unset ? e->d_slot : nullptr;
  const unsigned dblocks = (e->nchunks + 3) / 4; /* 4 waves (chunks) per 256-thread block */
  if (C == 1) {
    da.comp = 0;
    if (count_pass) {
      hipLaunchKernelGGL(mk_dump_count_kernel, dim3(dblocks), dim3(256), 0, st, da, e->d_chunk);
      hipLaunchKernelGGL(mk_dump_scan_kernel, dim3(1), dim3(1024), 0, st, e->d_chunk, e->nchunks, e->d_comp_totals);
    }
    hipLaunchKernelGGL(mk_dump_write_kernel, dim3(dblocks), dim3(256), 0, st, da, (const uint32_t *)e->d_chunk,
                       (const unsigned long long *)e->d_comp_totals, out_ids, koc ? out_cnt : nullptr);
  } else {
    /* all components in one count pass and one write pass; components back to back in the output */
    if (count_pass) {
      hipLaunchKernelGGL(mk_dumpc_count_kernel, dim3(dblocks), dim3(256), 0, st, da, e->d_chunk);
      hipLaunchKernelGGL(mk_dumpc_scan_kernel, dim3((unsigned)C), dim3(1024), 0, st, e->d_chunk, e->nchunks, e->d_comp_totals);
    }
    hipLaunchKernelGGL(mk_dumpc_write_kernel, dim3(dblocks), dim3(256), 0, st, da, (const uint32_t *)e->d_chunk,
                       (const unsigned long long *)e->d_comp_totals, out_ids, koc ? out_cnt : nullptr);
  }
  MK_HIP(e, hipGetLastError());
  return MK_OK;
}

/* finish for engines with sparse bookkeeping (tables of 2^26 slots and more; MK_OPT_SPARSE): compaction -> the key count
 * comes to the host (one small synchronisation: the passes below are sized by it, not by the table) -> priority layout ->
 * key-list dump into device staging -> one copy to the pinned result arrays.  The compaction has been launched. */
static int mk_finish_keylist(mk_engine *e, mk_result *out, mk_evpair ev) {
  const uint32_t S = e->kp.S;
  const int C = e->P.component_num;
  const bool koc = e->mode == MK_MODE_KOC;
  auto bail = [&](int rc) { if (e->profiling) e->ev_pool.push_back(ev); return rc; };
  MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  int rc = mk_check_counters(e);
  if (rc == MK_OK) rc = mk_dist_fit(e);
  if (rc) { e->begun = false; return bail(rc); }
  e->compacted = true;
  const uint64_t D = e->D;
  if (D > e->kl_cap) {
    hipFree(e->d_kl);
    e->d_kl = nullptr; e->kl_cap = 0;
    const uint64_t cap = D + D / 4 + 65536;
    MK_HIP(e, mk_dev_alloc(&e->d_kl, cap * (8 + 8 + 4 + 4 + 2) + 64));
    e->kl_cap = cap;
  }
  rc = mk_result_capacity(e, D ? D : 1);
  if (rc) return bail(rc);
  mk_kl_args a{};
  a.d = e->dist; a.Dp = e->d_counters; a.limit = mk_key_limit(e);
  a.slot = e->d_slot; a.S = S;
  a.comp_num = (uint32_t)C; a.comp_code_bits = (uint32_t)e->P.comp_code_bits;
  a.cnt_lo = e->mode == MK_MODE_OCC_SET ? e->min_occ : 1u; /* write_fqco2file(): marked keys only (iseq2comem.c:611) */
  a.cnt_hi = e->mode == MK_MODE_UNIQ_SET ? 1u : 0xffffffffu;  /* uniq_fasta2co(): repeated keys dropped */
  /* buckets of 2^shift consecutive slots per component, about eight keys each */
  {
    const uint64_t want = D / 8 + 256;
    const uint64_t width = ((uint64_t)S * (uint64_t)C) / want;
    uint32_t sh = 0;
    while (sh < 31u && (2ull << sh) <= width) sh++;
    a.shift = sh;
    a.bpc = (S >> sh) + 1u;
    a.nbuckets = (uint32_t)C * a.bpc;
  }
  if ((uint64_t)a.nbuckets * 2 + 2 > e->kl_bucket_cap) {
    hipFree(e->d_kl_buckets);
    e->d_kl_buckets = nullptr; e->kl_bucket_cap = 0;
    const uint64_t cap = (uint64_t)a.nbuckets * 2 + 2 + 4096;
    MK_HIP(e, mk_dev_alloc(&e->d_kl_buckets, cap * 4));
    e->kl_bucket_cap = cap;
  }
  a.bcount = e->d_kl_buckets; a.bcursor = e->d_kl_buckets + a.nbuckets + 1u;
  {
    uint8_t *p = (uint8_t *)e->d_kl;
    a.skey = (unsigned long long *)p; p += e->kl_cap * 8;
    a.tkey = (unsigned long long *)p; p += e->kl_cap * 8;
    a.tidx = (uint32_t *)p; p += e->kl_cap * 4;
    a.out_ids = (uint32_t *)p; p += e->kl_cap * 4;
    a.out_cnt = koc ? (uint16_t *)p : nullptr;
  }
  a.totals = e->d_comp_totals;
  a.out_cap = e->kl_cap;
  uint16_t *stage_cnt = (uint16_t *)((uint8_t *)e->d_kl + e->kl_cap * 24);
  MK_HIP(e, hipMemsetAsync(a.bcount, 0, ((size_t)a.nbuckets + 1) * 4, e->stream));
  uint64_t blocks = (D + 255) / 256;
  if (blocks > (uint64_t)e->num_cu * 16u) blocks = (uint64_t)e->num_cu * 16u;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(mk_layout_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, e->dist, (const unsigned long long *)e->d_counters,
                     (unsigned long long)mk_key_limit(e), e->d_slot, S, e->tab.err, (uint32_t *)nullptr, (uint32_t)MK_DUMP_SHIFT);
  hipLaunchKernelGGL(mk_kl_find_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, a);
  hipLaunchKernelGGL(mk_kl_scan_kernel, dim3(1), dim3(1024), 0, e->stream, a);
  hipLaunchKernelGGL(mk_kl_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, a);
  hipLaunchKernelGGL(mk_kl_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, e->stream, a);
  MK_HIP(e, hipGetLastError());
  MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  MK_HIP(e, hipMemcpyAsync(e->h_comp_totals, e->d_comp_totals, (size_t)C * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  if (D) { /* at most D entries come out: the whole staging prefix in one copy each (the component totals say how much of it counts) */
    MK_HIP(e, hipMemcpyAsync(e->h_ids, a.out_ids, D * 4, hipMemcpyDeviceToHost, e->stream));
    if (koc) MK_HIP(e, hipMemcpyAsync(e->h_cnt, stage_cnt, D * 2, hipMemcpyDeviceToHost, e->stream));
  }
  if (e->profiling) MK_HIP(e, hipEventRecord(ev.b, e->stream));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  rc = mk_check_counters(e);
  if (rc) { e->begun = false; e->slot_clean = false; return bail(rc); } /* (layout did not converge: the table is filled anew) */
  uint64_t total = 0;
  for (int c = 0; c < C; c++) total += e->h_comp_totals[c];
  if (total > D) { e->slot_clean = false; return bail(mk_fail(e, MK_ERR_HIP, "dump produced more entries than distinct keys")); }
  if (e->profiling) e->ev_finish.push_back(ev);
  uint64_t at = 0;
  for (int c = 0; c < C; c++) {
    e->comps[c].n = e->h_comp_totals[c];
    e->comps[c].ids = e->h_ids + at;
    e->comps[c].counts = koc ? e->h_cnt + at : nullptr;
    at += e->h_comp_totals[c];
  }
  out->component_num = C;
  out->total = total;
  out->components = e->comps.data();
  e->begun = false;
  return MK_OK;
}

/* One host synchronisation per finish: compaction, layout and dump take the key count from device memory, the dump writes
 * ids and counts straight into the pinned result arrays, and the counters (key count, component sizes, error flags) come
 * back in one small copy in front of the only hipStreamSynchronize.  Only a result larger than the arrays (first big
 * sketch on this engine) costs a second round: grow, write again. */
static int mk_res_reserve(mk_engine *e, uint64_t want) {
  if (!e->res_stream) {
    if (e->split_cus) {
      /* MK_OPT_SPLIT_CUS: layout, dump and the result copy on the SCAN queue's compute units, beside the scan as they are with one queue
       * (small kernels: they fit next to its workgroups).  On the second queue's few units they make THOSE the bottleneck (2.80 ms a pass
       * against 2.35), and left unmasked they land there whenever the units are idle and stretch the resolve kernel's chain past the next
       * scan's end every other run (2.29 or 2.38 ms a pass, profiles/r05_split_queues.txt) */
      uint32_t scan_mask[16] = {0};
      for (int i = 0; i < e->num_cu - e->split_cus; i++) scan_mask[i >> 5] |= 1u << (i & 31);
      MK_HIP(e, hipExtStreamCreateWithCUMask(&e->res_stream, (uint32_t)((e->num_cu + 31) / 32), scan_mask));
    } else
      MK_HIP(e, hipStreamCreateWithFlags(&e->res_stream, hipStreamNonBlocking));
  }
  if (!e->ev_res) {
    MK_HIP(e, hipEventCreateWithFlags(&e->ev_res, hipEventDisableTiming));
    MK_HIP(e, mk_dev_alloc(&e->d_snap, 8 * sizeof(unsigned long long)));
    MK_HIP(e, mk_pin_alloc((void **)&e->h_snap, 8 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  if (want <= e->res_cap && e->d_res_ids) return MK_OK;
  hipFree(e->d_res_ids); hipFree(e->d_res_cnt);
  e->d_res_ids = nullptr; e->d_res_cnt = nullptr; e->res_cap = 0;
  MK_HIP(e, mk_dev_alloc(&e->d_res_ids, want * 4));
  MK_HIP(e, mk_dev_alloc(&e->d_res_cnt, want * 2));
  e->res_cap = want;
  return MK_OK;
}

static int mk_finish_impl(mk_engine *e, mk_result *out, bool staged);

extern "C" int mk_sketch_finish(mk_engine *e, mk_result *out) {
  if (!e || !out) return MK_ERR_ARG;
  if (e->res_pending) return mk_fail(e, MK_ERR_STATE, "mk_sketch_finish while the result of mk_sketch_finish_begin has not been taken (mk_sketch_finish_end)");
  return mk_finish_impl(e, out, false);
}

/* The same finish in two halves.  _begin: compaction, layout and dump as in mk_sketch_finish, but the dump goes to staging
 * arrays in HBM (a sweep at memory speed instead of 4-byte stores across PCIe), the one synchronisation brings the counters,
 * and the copy to the pinned result arrays is queued on a stream of its own.  The sketch is over when _begin returns: the
 * caller may begin and push the NEXT sketch, whose kernels then run beside that copy.  _end waits for the copy and hands out
 * the result (valid until the next mk_sketch_finish / _finish_begin on the engine).  One result may be outstanding. */
extern "C" int mk_sketch_finish_begin(mk_engine *e) {
  if (!e) return MK_ERR_ARG;
  if (e->res_pending) return mk_fail(e, MK_ERR_STATE, "mk_sketch_finish_begin: take the previous result first (mk_sketch_finish_end)");
  mk_result r;
  return mk_finish_impl(e, &r, true);
}

extern "C" int mk_sketch_finish_end(mk_engine *e, mk_result *out) {
  if (!e || !out) return MK_ERR_ARG;
  if (!e->res_pending) return mk_fail(e, MK_ERR_STATE, "mk_sketch_finish_end without mk_sketch_finish_begin");
  MK_HIP(e, hipSetDevice(e->device));
  MK_HIP(e, hipEventSynchronize(e->ev_res));
  e->res_pending = false;
  if (e->res_side) { /* layout and dump ran on the side stream: their flags and sizes have come back with the result */
    e->res_side = false;
    const int C = e->P.component_num;
    const bool koc = e->res_koc;
    if ((uint32_t)(e->h_snap[2] & 0xffffffffu) & 2u) { e->slot_clean = false; return mk_fail(e, MK_ERR_HIP, "layout kernel did not converge"); }
    uint64_t total = 0, at = 0;
    for (int c = 0; c < C; c++) total += e->h_comp_totals[c];
    if (total > e->res_D) return mk_fail(e, MK_ERR_HIP, "dump produced more entries than distinct keys");
    for (int c = 0; c < C; c++) {
      e->comps[c].n = e->h_comp_totals[c];
      e->comps[c].ids = e->h_ids + at;
      e->comps[c].counts = koc ? e->h_cnt + at : nullptr;
      at += e->h_comp_totals[c];
    }
    e->res_total = total;
  }
  out->component_num = e->P.component_num;
  out->total = e->res_total;
  out->components = e->comps.data();
  return MK_OK;
}

static int mk_finish_impl(mk_engine *e, mk_result *out, bool staged) {
  if (!e->begun) return mk_fail(e, MK_ERR_STATE, "finish before mk_sketch_begin");
  MK_HIP(e, hipSetDevice(e->device));
  if (!e->res_pending) { mk_pin_repoison(e->h_ids, (size_t)e->h_cap * 4); mk_pin_repoison(e->h_cnt, (size_t)e->h_cap * 2); } /* (MK_POISON: the last result's arrays) */
  mk_evpair ev{};
  if (e->profiling) { ev = mk_ev_get(e); MK_HIP(e, hipEventRecord(ev.a, e->stream)); }
  int rc = MK_OK;
  if (!e->compacted || e->region_open) rc = mk_compact_launch(e);
  if (rc) { if (e->profiling) e->ev_pool.push_back(ev); return rc; }
  const uint32_t S = e->kp.S;
  const int C = e->P.component_num;
  const bool koc = e->mode == MK_MODE_KOC;
  if (!e->h_ids) { /* first finish: room for 2 M keys (50 M reads at L3K11 leave 1.6 M), never more than the table admits */
    const uint64_t lim = mk_key_limit(e) + 1;
    rc = mk_result_capacity(e, lim < (2u << 20) ? lim : (2u << 20));
    if (rc) { if (e->profiling) e->ev_pool.push_back(ev); return rc; }
  }

  if (e->sparse) { /* large tables: the dump starts from the key list, not from the table (mk_kl_* kernels) */
    rc = mk_finish_keylist(e, out, ev);
    if (rc == MK_OK && staged) { /* (the key-list dump is staged and copied already: the result is there) */
      MK_HIP(e, mk_res_reserve(e, 1) == MK_OK ? hipSuccess : hipErrorOutOfMemory);
      MK_HIP(e, hipEventRecord(e->ev_res, e->res_stream));
      e->res_pending = true;
      e->res_total = out->total;
    }
    return rc;
  }
  if (staged) {
    /* Two halves, the second on a stream of its own.  On the engine's stream only the compaction and one small copy: the
     * wait brings the key count and the sketch's flags (a crowded table ends here).  Layout, dump and the copy to the host
     * touch nothing but the key list, the layout table and the staging arrays, so they are queued on the side stream and the
     * caller's next mk_sketch_begin / pushes run beside them; the count and flags they need are a snapshot (the next begin
     * clears the counters). */
    rc = mk_res_reserve(e, e->res_cap ? e->res_cap : e->h_cap);
    if (rc) { if (e->profiling) e->ev_pool.push_back(ev); return rc; }
    hipStream_t rs = e->res_stream;
    MK_HIP(e, hipMemcpyAsync(e->d_snap, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, e->stream));
    MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    if (e->profiling) MK_HIP(e, hipEventRecord(ev.b, e->stream));
    MK_HIP(e, hipEventRecord(e->ev_res, e->stream));
    /* the side stream's kernels take the key count from the snapshot: queued now, behind the compaction, while the host still
     * waits for the engine's stream */
    MK_HIP(e, hipStreamWaitEvent(rs, e->ev_res, 0));
    mk_evpair ev2{};
    if (e->profiling) { ev2 = mk_ev_get(e); MK_HIP(e, hipEventRecord(ev2.a, rs)); }
    if (!e->slot_clean) MK_HIP(e, hipMemsetAsync(e->d_slot, 0xFF, (size_t)S * sizeof(uint32_t), rs));
    hipLaunchKernelGGL(mk_layout_kernel, dim3((unsigned)e->num_cu * 16u), dim3(256), 0, rs, e->dist, (const unsigned long long *)e->d_snap,
                       (unsigned long long)mk_key_limit(e), e->d_slot, S, (uint32_t *)(e->d_snap + 2), (uint32_t *)nullptr, (uint32_t)MK_DUMP_SHIFT);
    MK_HIP(e, hipGetLastError());
    /* an error from here on leaves work queued on the side stream: it is waited for, the sketch is over, and the layout table is
     * filled anew by the next finish */
    auto staged_bail = [&](int code) {
      (void)hipStreamSynchronize(e->stream);
      (void)hipStreamSynchronize(rs);
      e->begun = false; e->slot_clean = false; e->compacted = false;
      if (e->profiling) { e->ev_pool.push_back(ev); e->ev_pool.push_back(ev2); }
      return code;
    };
    rc = mk_launch_dump(e, true, e->d_res_ids, e->d_res_cnt, e->res_cap, rs, true);
    if (rc) return staged_bail(rc);
    e->slot_clean = true; /* the write pass hands every slot back (a dump that held back, below, fills the table anew) */
    MK_HIP(e, hipStreamSynchronize(e->stream));
    rc = mk_check_counters(e);
    if (rc) { /* crowded: the layout kernel did nothing, the dump wrote nothing */
      e->begun = false;
      MK_HIP(e, hipStreamSynchronize(rs));
      if (e->profiling) { e->ev_pool.push_back(ev); e->ev_pool.push_back(ev2); }
      return rc;
    }
    const uint64_t D = e->D;
    rc = mk_result_capacity(e, D ? D : 1);
    if (rc) return staged_bail(rc);
    if (D > e->res_cap) { /* the write pass held back (more keys than the staging arrays hold): larger arrays, layout and dump again */
      MK_HIP(e, hipStreamSynchronize(rs));
      rc = mk_res_reserve(e, D + D / 8 + 1024);
      if (rc) return staged_bail(rc);
      MK_HIP(e, hipMemsetAsync(e->d_slot, 0xFF, (size_t)S * sizeof(uint32_t), rs));
      hipLaunchKernelGGL(mk_layout_kernel, dim3((unsigned)e->num_cu * 16u), dim3(256), 0, rs, e->dist, (const unsigned long long *)e->d_snap,
                         (unsigned long long)mk_key_limit(e), e->d_slot, S, (uint32_t *)(e->d_snap + 2), (uint32_t *)nullptr, (uint32_t)MK_DUMP_SHIFT);
      MK_HIP(e, hipGetLastError());
      rc = mk_launch_dump(e, true, e->d_res_ids, e->d_res_cnt, e->res_cap, rs, true);
      if (rc) return staged_bail(rc);
    }
    if (e->profiling) e->ev_finish.push_back(ev);
    if (D) { /* at most D entries come out (all of them with -A): the staging prefix in one copy each */
      MK_HIP(e, hipMemcpyAsync(e->h_ids, e->d_res_ids, D * 4, hipMemcpyDeviceToHost, rs));
      if (koc) MK_HIP(e, hipMemcpyAsync(e->h_cnt, e->d_res_cnt, D * 2, hipMemcpyDeviceToHost, rs));
    }
    MK_HIP(e, hipMemcpyAsync(e->h_comp_totals, e->d_comp_totals, (size_t)C * sizeof(unsigned long long), hipMemcpyDeviceToHost, rs));
    MK_HIP(e, hipMemcpyAsync(e->h_snap, e->d_snap, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, rs));
    if (e->profiling) { MK_HIP(e, hipEventRecord(ev2.b, rs)); e->ev_finish_side.push_back(ev2); }
    MK_HIP(e, hipEventRecord(e->ev_res, rs));
    e->res_pending = true; e->res_side = true;
    e->res_koc = koc; e->res_D = D; /* (the next begin resets mode and key count before this result is taken) */
    e->compacted = false; /* the key list belongs to the side stream until mk_sketch_finish_end */
    e->begun = false;
    return MK_OK;
  }
  MK_HIP(e, hipMemsetAsync(e->d_slot, 0xFF, (size_t)S * sizeof(uint32_t), e->stream));
  e->slot_clean = false; /* (the plain finish leaves the layout in the table: its write pass may have to run again) */
  hipLaunchKernelGGL(mk_layout_kernel, dim3((unsigned)e->num_cu * 16u), dim3(256), 0, e->stream, e->dist,
                     (const unsigned long long *)e->d_counters, (unsigned long long)mk_key_limit(e), e->d_slot, S, e->tab.err,
                     e->sparse ? e->d_dirty_slot : nullptr, (uint32_t)MK_DUMP_SHIFT);
  MK_HIP(e, hipGetLastError());
  rc = mk_launch_dump(e, true);
  if (rc) { if (e->profiling) e->ev_pool.push_back(ev); return rc; }
  MK_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  MK_HIP(e, hipMemcpyAsync(e->h_comp_totals, e->d_comp_totals, (size_t)C * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  if (e->profiling) MK_HIP(e, hipEventRecord(ev.b, e->stream));
  MK_HIP(e, hipStreamSynchronize(e->stream));
  rc = mk_check_counters(e);
  if (rc) { /* the sketch is spent (crowded table, malformed stream): options may be set again, the next call is a begin */
    e->begun = false;
    if (e->profiling) e->ev_pool.push_back(ev);
    return rc;
  }
  e->compacted = true;
  uint64_t total = 0;
  for (int c = 0; c < C; c++) total += e->h_comp_totals[c];
  if (total > e->D) { if (e->profiling) e->ev_pool.push_back(ev); return mk_fail(e, MK_ERR_HIP, "dump produced more entries than distinct keys"); }
  if (total > e->h_cap) { /* the write kernel held back: grow the result arrays and run it again */
    rc = mk_result_capacity(e, total);
    if (rc == MK_OK) rc = mk_launch_dump(e, false);
    if (rc) { if (e->profiling) e->ev_pool.push_back(ev); return rc; }
    if (e->profiling) MK_HIP(e, hipEventRecord(ev.b, e->stream));
    MK_HIP(e, hipStreamSynchronize(e->stream));
  }
  if (e->profiling) e->ev_finish.push_back(ev);
  uint64_t at = 0;
  for (int c = 0; c < C; c++) {
    e->comps[c].n = e->h_comp_totals[c];
    e->comps[c].ids = e->h_ids + at;
    e->comps[c].counts = koc ? e->h_cnt + at : nullptr;
    at += e->h_comp_totals[c];
  }
  out->component_num = C;
  out->total = total;
  out->components = e->comps.data();
  e->begun = false;
  return MK_OK;
}

extern "C" int mk_result_release(mk_engine *e, mk_result *r) {
  if (!e || !r) return MK_ERR_ARG;
  r->components = nullptr;
  r->total = 0;
  return MK_OK;
}

/* ---- synthetic reads on the device --------------------------------------------------------------------- */
extern "C" int mk_synth_rows_device(int device, void *hip_stream, uint64_t seed, uint64_t first_read, uint64_t nreads,
                                    uint32_t len, uint32_t stride, uint8_t *rows_dev) {
  if (!rows_dev || stride < len + 1 || (stride & 15u) || ((uintptr_t)rows_dev & 15u))
    return mk_fail(nullptr, MK_ERR_ARG, "mk_synth_rows_device: stride must be a multiple of 16 and > len");
  if (hipSetDevice(device) != hipSuccess) return mk_fail(nullptr, MK_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
  if (nreads == 0) return MK_OK;
  uint64_t total = nreads * (stride >> 4);
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(mk_synth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, seed, first_read, nreads, len,
                     stride, rows_dev);
  hipError_t r = hipGetLastError();
  return r == hipSuccess ? MK_OK : mk_fail(nullptr, MK_ERR_HIP, "synth launch: %s", hipGetErrorString(r));
}

/* ---- batches of small inputs (mk_batch.hip.h) -------------------------------------------------------------------------------
 * run_stageI()'s team over files (command_dist.c:363-372) as ONE launch sequence per batch of files. */
struct mk_dbuf { void *p = nullptr; size_t cap = 0; };
struct mk_bctx {
  mk_dbuf text, stream, sum, desc, zero, map, list, bcur;
  void *h_desc = nullptr; size_t h_desc_cap = 0;   /* pinned: what goes up in one small copy (descriptor, files, seg0s, row0s) */
  void *h_stat = nullptr; size_t h_stat_cap = 0;   /* pinned: what comes back first (per-file status, per-component sizes, totals) */
  uint32_t *h_ids = nullptr; size_t h_ids_cap = 0; /* pinned: the ids */
  hipEvent_t ev_stat = nullptr;
  mk_batch_dev hb{};
  int mode = 0;
  uint32_t nfiles = 0;
  size_t stat_bytes = 0;
  hipStream_t used_stream = nullptr; /* the queue this batch's launch sequence went to */
  bool rows = false;     /* the files are packed rows (mk_sketch_batch_begin_rows) */
  uint32_t format = 0;   /* MK_ROWS_PACKED / MK_ROWS_WIDE then */
  const uint8_t *rows_src = nullptr; /* where the device reads them: the caller's pinned stretch, or this context's copy */
  std::vector<uint64_t> row_off;     /* per file: bytes from rows_src to its rows */
  uint64_t spec_ids = 0; /* ids the pinned block holds: written there by the batch's own launch sequence */
  std::vector<mk_batch_file> files;
  std::vector<mk_component> comps;              /* [nfiles * component_num] */
  std::vector<std::vector<uint32_t>> alone_ids; /* ids of the files that were sketched alone */
};

static int mk_dbuf_fit(mk_engine *e, mk_dbuf &b, size_t need) {
  if (b.p && need <= b.cap) { MK_HIP(e, mk_dev_repoison(b.p, b.cap, e->stream)); return MK_OK; } /* (MK_POISON: what the context's last batch left) */
  hipFree(b.p);
  b.p = nullptr; b.cap = 0;
  const size_t cap = need + need / 4 + 4096;
  MK_HIP(e, mk_dev_alloc(&b.p, cap));
  b.cap = cap;
  return MK_OK;
}
static int mk_pinned_fit(mk_engine *e, void **p, size_t *cap, size_t need) {
  if (*p && need <= *cap) { mk_pin_repoison(*p, *cap); return MK_OK; }
  if (*p) hipHostFree(*p);
  *p = nullptr; *cap = 0;
  const size_t c = need + need / 4 + 4096;
  MK_HIP(e, mk_pin_alloc(p, c, hipHostMallocDefault));
  *cap = c;
  return MK_OK;
}
static void mk_bctx_free(mk_bctx *c) {
  if (!c) return;
  for (mk_dbuf *b : {&c->text, &c->stream, &c->sum, &c->desc, &c->zero, &c->map, &c->list, &c->bcur}) hipFree(b->p);
  if (c->h_desc) hipHostFree(c->h_desc);
  if (c->h_stat) hipHostFree(c->h_stat);
  if (c->h_ids) hipHostFree(c->h_ids);
  if (c->ev_stat) hipEventDestroy(c->ev_stat);
  delete c;
}
static size_t mk_up16(size_t v) { return (v + 15u) & ~(size_t)15u; }

/* rows one scan launch may take so that a scan wave's candidate buffer is expected to stay below half its capacity: the share
 * of bases whose window passes the LDS filter is the accepted share of the subspace (both strands) plus the filter's false
 * positives (three bits per entry in 32 * bm_words bits; the tuned kernels' pair filter: measured 2 % of the 8-base windows) */
static uint64_t mk_rows_per_launch(const mk_engine *e, uint32_t row_bases, int tuned_subk) {
  const double fill = 1.0 - exp(-3.0 * (double)e->n_accept / (32.0 * (double)(1u << e->bm_bits)));
  /* records per base: the tuned kernels append one record per flagged 8-base window (subk 6: 2 % of them; subk 5: a base in 64
   * passes the half-size bitmap, 12 % of the windows), the generic kernel one per flagged base */
  const double per_base = tuned_subk == 6 ? 0.02 / 8.0 : tuned_subk == 5 ? 0.125 / 8.0 : (double)e->n_accept / (double)e->P.shuf_len + fill * fill * fill;
  const double per_row = per_base * row_bases + 1e-9;
  const uint64_t waves = (uint64_t)e->num_cu * (uint64_t)(e->tune_threads / 64);
  double rows = 0.5 * (double)e->cand_cap / per_row * (double)waves;
  if (rows < 64.0 * (double)waves) rows = 64.0 * (double)waves;
  if (rows > (double)(1ull << 30)) rows = (double)(1ull << 30);
  return (uint64_t)rows / 64u * 64u;
}

/* rows: the files are PACKED ROWS already (mk_fasta_pack_rows on the reader's thread: the FASTA walk done by the host) -- no text,
 * no mk_fab_* kernels; where the rows lie in ONE stretch of pinned memory the scan kernel reads them THERE, through the mapping
 * (profiles/r04_probe_hostread.jsonl: a kernel reads registered host memory at 55.5 GB/s, the copy engine moves it at 57.0 and costs 7.7 ms
 * of set-up at the first copy of a process) */
static int mk_batch_begin_impl(mk_engine *e, int mode, const mk_batch_file *files, uint32_t nfiles, const uint32_t format) {
  if (!e || !files) return MK_ERR_ARG;
  if (format != 0u && format != MK_ROWS_PACKED && format != MK_ROWS_WIDE) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin_rows: format MK_ROWS_PACKED or MK_ROWS_WIDE");
  const bool rows = format != 0u;
  const uint32_t row_bases = format == MK_ROWS_WIDE ? MK_WIDE_MAX_BASES : MK_PACKED_MAX_BASES;
  if (mode != MK_MODE_SET && mode != MK_MODE_UNIQ_SET) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin: MK_MODE_SET or MK_MODE_UNIQ_SET");
  if (nfiles < 1 || nfiles > MK_BATCH_MAX_FILES) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin: 1 .. %u files", MK_BATCH_MAX_FILES);
  if (e->begun) return mk_fail(e, MK_ERR_STATE, "mk_sketch_batch_begin inside a sketch (between begin and finish)");
  if (e->batch_begun - e->batch_ended >= 2) return mk_fail(e, MK_ERR_STATE, "mk_sketch_batch_begin: two batches are in flight (mk_sketch_batch_end first)");
  { int rc0 = mk_tail_end(e); if (rc0) return rc0; } /* (a batch keeps the stream it starts on: no mark of an earlier sketch may move it) */
  if (e->P.TL + MK_FA_PITCH > 4000u) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin: k-mer too long for the stream rows");
  if (rows && !mk_params_packed_ok(&e->P)) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin_rows: no scan kernel for packed rows at k %d, subk %d", e->P.k, e->P.subk);
  uint64_t total = 0, nmax = 0;
  for (uint32_t i = 0; i < nfiles; i++) {
    if (!files[i].text && files[i].n) return MK_ERR_ARG;
    if (rows && ((files[i].n % MK_PACKED_PITCH) || ((uintptr_t)files[i].text & 15u))) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin_rows: file %u: rows of 64 bytes, 16-byte aligned", i);
    if (files[i].n > MK_BATCH_FILE_MAX) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin: file %u has %llu bytes (at most %llu)", i, (unsigned long long)files[i].n, (unsigned long long)MK_BATCH_FILE_MAX);
    total += files[i].n;
    if (files[i].n > nmax) nmax = files[i].n;
  }
  if (total > MK_BATCH_TEXT_MAX) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin: %llu bytes of text (at most %llu)", (unsigned long long)total, (unsigned long long)MK_BATCH_TEXT_MAX);
  MK_HIP(e, hipSetDevice(e->device));
  const int ci = (int)(e->batch_begun % 3u);
  if (!e->bctx[ci]) e->bctx[ci] = new mk_bctx();
  mk_bctx *c = e->bctx[ci];
  if (!c->ev_stat) MK_HIP(e, hipEventCreateWithFlags(&c->ev_stat, hipEventDisableTiming));
  const uint32_t C = (uint32_t)e->P.component_num;
  const uint32_t TL = (uint32_t)e->P.TL, pitch = MK_FA_PITCH_SMALL, rowlen = pitch + TL - 1u;
  const uint32_t width = (rowlen + 1u + 15u) & ~15u;

  /* ---- geometry of the batch: where every file's text, stream region, segments and rows lie */
  /* Texts: one copy for all of them when each starts where the one in front ends, rounded up to 1 KiB (a buffer filled file by file:
   * nothing but the texts and their padding is read); otherwise they are packed file by file.
   * Rows: read where they lie when they are in ascending order, 64 bytes apart, inside memory the device has mapped (both ends are
   * asked for: what lies between two files is never touched, but it must be the caller's ONE pinned stretch); anything else --
   * separate arrays that merely happen to lie in ascending order, pageable memory -- is copied file by file. */
  bool one_copy = true;
  const uint8_t *rows_dev = nullptr;
  for (uint32_t i = 1; i < nfiles && one_copy; i++) {
    if (rows) {
      if (files[i].text < files[i - 1].text + files[i - 1].n || (size_t)(files[i].text - files[0].text) % MK_PACKED_PITCH) one_copy = false;
    } else if (files[i].text != files[i - 1].text + (files[i - 1].n + MK_FA_SEG - 1) / MK_FA_SEG * MK_FA_SEG) one_copy = false;
  }
  if (rows && one_copy) {
    const uint64_t span = (uint64_t)(files[nfiles - 1].text - files[0].text) + files[nfiles - 1].n;
    if (!files[0].text || span == 0 || span > 2 * total + (uint64_t)nfiles * 4096u + ((uint64_t)64 << 20)) one_copy = false;
    else {
      hipPointerAttribute_t pa0, pa1;
      void *dp = nullptr;
      if (hipPointerGetAttributes(&pa0, files[0].text) == hipSuccess && pa0.type == hipMemoryTypeHost &&
          hipPointerGetAttributes(&pa1, files[0].text + span - 1) == hipSuccess && pa1.type == hipMemoryTypeHost &&
          hipHostGetDevicePointer(&dp, (void *)files[0].text, 0) == hipSuccess && dp) rows_dev = (const uint8_t *)dp;
      else { (void)hipGetLastError(); one_copy = false; }
    }
  }
  const size_t desc_bytes = mk_up16(sizeof(mk_batch_dev)) + mk_up16((size_t)nfiles * sizeof(mk_bfile)) + 2 * mk_up16(((size_t)nfiles + 1) * 4);
  int rc = mk_pinned_fit(e, &c->h_desc, &c->h_desc_cap, desc_bytes);
  if (rc) return rc;
  rc = mk_dbuf_fit(e, c->desc, desc_bytes);
  if (rc) return rc;
  uint8_t *hd = (uint8_t *)c->h_desc, *dd = (uint8_t *)c->desc.p;
  mk_batch_dev *hb = (mk_batch_dev *)hd;
  mk_bfile *hf = (mk_bfile *)(hd + mk_up16(sizeof(mk_batch_dev)));
  uint32_t *hseg0 = (uint32_t *)((uint8_t *)hf + mk_up16((size_t)nfiles * sizeof(mk_bfile)));
  uint32_t *hrow0 = (uint32_t *)((uint8_t *)hseg0 + mk_up16(((size_t)nfiles + 1) * 4));
  uint64_t toff = 0, soff = 0;
  uint32_t seg = 0;
  if (rows) c->row_off.assign(nfiles, 0);
  for (uint32_t i = 0; i < nfiles && rows; i++) { /* the files' rows as they lie (one stretch), or side by side */
    mk_bfile &f = hf[i];
    f.text_off = 0; f.text_len = 0; f.seg0 = 0; f.nseg = 0;
    f.stream_off = one_copy ? (uint64_t)(files[i].text - files[0].text) : soff;
    f.stream_cap = files[i].n;
    soff = f.stream_off + f.stream_cap;
    f.row0 = (uint32_t)(f.stream_off / MK_PACKED_PITCH);
    f.nrow = (uint32_t)(f.stream_cap / MK_PACKED_PITCH);
    hseg0[i] = 0;
    hrow0[i] = f.row0;
    c->row_off[i] = f.stream_off;
  }
  for (uint32_t i = 0; i < nfiles && !rows; i++) {
    mk_bfile &f = hf[i];
    f.text_off = one_copy ? (uint64_t)(files[i].text - files[0].text) : toff;
    f.text_len = files[i].n;
    toff = f.text_off + ((f.text_len + MK_FA_SEG - 1) / MK_FA_SEG) * MK_FA_SEG;
    f.stream_off = soff;
    f.stream_cap = (f.text_len + 1u + pitch - 1u) / pitch * pitch; /* > text_len: at least one '\n' closes the region */
    soff += f.stream_cap;
    f.seg0 = seg;
    f.nseg = (uint32_t)((f.text_len + MK_FA_SEG - 1) / MK_FA_SEG);
    if (f.nseg == 0) f.nseg = 1; /* an empty file still has a region to close */
    seg += f.nseg;
    f.row0 = (uint32_t)(f.stream_off / pitch);
    f.nrow = (uint32_t)(f.stream_cap / pitch);
    hseg0[i] = f.seg0;
    hrow0[i] = f.row0;
  }
  const uint64_t text_span = rows ? 0 : one_copy ? (uint64_t)(files[nfiles - 1].text - files[0].text) + files[nfiles - 1].n : toff;
  const uint64_t total_rows = soff / (rows ? MK_PACKED_PITCH : pitch);
  const uint32_t nseg_total = seg;
  hseg0[nfiles] = nseg_total;
  hrow0[nfiles] = (uint32_t)total_rows;
  if (total_rows >= (1ull << 31)) return mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin: too many rows");

  /* ---- tables: 2^tb slots per file, about five times the keys the largest file is expected to leave (its k-mers / 16^drlevel) */
  uint32_t tb = 10;
  uint64_t est_ids = 0; /* ids the whole batch is expected to leave: every file's bases / 16^drlevel */
  {
    /* (bases of the largest file: its text, or its rows' 153 - TL new bases each) */
    const uint64_t nbases = rows ? nmax / MK_PACKED_PITCH * (row_bases + 1u - TL) : nmax;
    est_ids = (rows ? total / MK_PACKED_PITCH * (row_bases + 1u - TL) : total) >> (4u * (uint32_t)e->P.drlevel);
    /* three times the keys the largest file is expected to leave (round 6; five before): every kernel behind a batch's scan sweeps the
     * tables or the maps -- clear, compaction, both bucket passes -- and the link idles while they run.  At L2K11 (15 600 keys a 4 Mbase genome)
     * that is 2^16 instead of 2^17 slots a file, load 0.24: the 32 batches of 1 024 genomes lose 2.3 of 11.3 ms of such kernels
     * (profiles/r06_config5_trace.txt).  A file with more keys than half its table is sketched alone, as before */
    const uint64_t est = (nbases >> (4u * (uint32_t)e->P.drlevel)) * 3u;
    while (tb < 22u && (1ull << tb) < est) tb++;
    if (e->batch_tb_opt) tb = (uint32_t)e->batch_tb_opt;
    while (tb > 9u && ((uint64_t)nfiles << tb) > (1ull << 26)) tb--; /* (files that do not fit then are sketched alone) */
  }
  const uint64_t N = (uint64_t)nfiles << tb;
  /* buckets of the ordered dump: about 2^tb / 32 per file, so that a bucket holds a few keys at the expected load */
  uint32_t shift = 0, bpc = 0;
  {
    uint64_t want = ((1ull << tb) / 32u) / C;
    if (want < 1) want = 1;
    const uint64_t wd = (uint64_t)e->kp.S / want;
    while (shift < 31u && (2ull << shift) <= wd) shift++;
    bpc = (e->kp.S >> shift) + 1u;
  }
  const uint32_t bpf = bpc * C;
  const uint64_t nb = (uint64_t)nfiles * bpf;
  c->stat_bytes = mk_up16((size_t)nfiles * sizeof(mk_bstat)) + mk_up16((size_t)nfiles * C * 4) + 16 + 16; /* .. | misc[2] | the engine's error flags (mk_b_home_kernel) */
  const size_t zero_bytes = (size_t)N * 16 + mk_up16((size_t)(nb + 1) * 4) + c->stat_bytes + (size_t)nfiles * 64;
  static const bool trace = getenv("MK_BATCH_TRACE") != nullptr;
  const double tr0 = trace ? mk_tick_now() : 0.0;
  /* whole segments: a wave stages MK_FA_SEG bytes with one load per lane whatever the segment's last byte is (mk_fa_stage) */
  if (!rows && (rc = mk_dbuf_fit(e, c->text, (((size_t)text_span + MK_FA_SEG - 1) / MK_FA_SEG) * MK_FA_SEG + 256))) return rc;
  if (!rows_dev && (rc = mk_dbuf_fit(e, c->stream, (size_t)soff + 8192))) return rc;
  if (!rows && (rc = mk_dbuf_fit(e, c->sum, (size_t)nseg_total * sizeof(mk_fa_sum)))) return rc;
  if ((rc = mk_dbuf_fit(e, c->zero, zero_bytes))) return rc;
  if ((rc = mk_dbuf_fit(e, c->map, (size_t)N * 8))) return rc;
  if ((rc = mk_dbuf_fit(e, c->list, (size_t)N * 40))) return rc;
  if ((rc = mk_dbuf_fit(e, c->bcur, (size_t)nb * 4 + 16))) return rc;
  if ((rc = mk_pinned_fit(e, &c->h_stat, &c->h_stat_cap, c->stat_bytes))) return rc;
  const double tr1 = trace ? mk_tick_now() : 0.0;
  {
    uint8_t *z = (uint8_t *)c->zero.p;
    hb->kc = (unsigned long long *)z;
    hb->ordinv = hb->kc + N;
    hb->bstart = (uint32_t *)(z + (size_t)N * 16);
    uint8_t *st = z + (size_t)N * 16 + mk_up16((size_t)(nb + 1) * 4);
    hb->stat = (mk_bstat *)st;
    hb->ctot = (uint32_t *)(st + mk_up16((size_t)nfiles * sizeof(mk_bstat)));
    hb->misc = (unsigned long long *)(st + mk_up16((size_t)nfiles * sizeof(mk_bstat)) + mk_up16((size_t)nfiles * C * 4));
    hb->nout = (uint32_t *)(st + c->stat_bytes);
    hb->map = (unsigned long long *)c->map.p;
    uint8_t *l = (uint8_t *)c->list.p;
    hb->key = (unsigned long long *)l;
    hb->ord = hb->key + N;
    hb->tkey = hb->ord + N;
    hb->cnt = (uint32_t *)(hb->tkey + N);
    hb->gid = hb->cnt + N;
    hb->tidx = hb->gid + N;
    hb->out_ids = hb->tidx + N;
    hb->list_cap = N; hb->out_cap = N;
    hb->bcursor = (uint32_t *)c->bcur.p;
    hb->files = (const mk_bfile *)(dd + mk_up16(sizeof(mk_batch_dev)));
    hb->seg0s = (const uint32_t *)(dd + ((uint8_t *)hseg0 - hd));
    hb->row0s = (const uint32_t *)(dd + ((uint8_t *)hrow0 - hd));
    hb->nfiles = nfiles; hb->tb = tb;
    const uint64_t half = 1ull << (tb - 1u);
    hb->key_limit = (uint32_t)(half < e->P.hashlimit ? half : e->P.hashlimit);
    hb->S = e->kp.S;
    hb->comp_num = C; hb->comp_code_bits = (uint32_t)e->P.comp_code_bits;
    hb->cnt_hi = mode == MK_MODE_UNIQ_SET ? 1u : 0xffffffffu;
    hb->shift = shift; hb->bpc = bpc; hb->bpf = bpf;
  }
  c->hb = *hb;
  c->mode = mode; c->nfiles = nfiles; c->rows = rows; c->format = format; c->rows_src = rows ? (rows_dev ? rows_dev : (const uint8_t *)c->stream.p) : nullptr;
  c->files.assign(files, files + nfiles);
  const mk_batch_dev *dbatch = (const mk_batch_dev *)dd;
  hipStream_t s = e->stream;
  c->used_stream = s;

  /* ---- the launch sequence, all of it on the engine's one stream.  (A copy queue of its own for the texts -- made by a thread beside
   * the first batches, the kernels behind an event -- was built and measured: 1 024 genomes in 0.222-0.238 s with it, 0.199-0.213 s
   * without, three runs each on one box: a batch's kernels are 0.4 ms beside 2.3-2.8 ms of copy, and the copy is no faster for
   * running beside them.) */
  double trk = tr1;
  auto tick = [&](const char *what) {
    if (!trace || e->batch_begun) return;
    const double n = mk_tick_now();
    fprintf(stderr, "[mk batch 0]   %-28s %.3f ms\n", what, (n - trk) * 1e3);
    trk = n;
  };
  tick("descriptors built");
  MK_HIP(e, hipMemcpyAsync(c->desc.p, c->h_desc, desc_bytes, hipMemcpyHostToDevice, s));
  tick("descriptor copy queued");
  if (rows) {
    if (!rows_dev) /* (not one mapped stretch: file by file, side by side) */
      for (uint32_t i = 0; i < nfiles; i++)
        if (files[i].n) MK_HIP(e, hipMemcpyAsync((uint8_t *)c->stream.p + hf[i].stream_off, files[i].text, (size_t)files[i].n, hipMemcpyHostToDevice, s));
  } else if (one_copy) {
    if (text_span) MK_HIP(e, hipMemcpyAsync(c->text.p, files[0].text, (size_t)text_span, hipMemcpyHostToDevice, s));
  } else {
    for (uint32_t i = 0; i < nfiles; i++)
      if (files[i].n) MK_HIP(e, hipMemcpyAsync((uint8_t *)c->text.p + hf[i].text_off, files[i].text, (size_t)files[i].n, hipMemcpyHostToDevice, s));
  }
  tick("text copy queued");
  const unsigned wide = (unsigned)e->num_cu * 8u;
  hipLaunchKernelGGL(mk_b_clear_kernel, dim3(wide), dim3(256), 0, s, (uint4 *)c->zero.p, (unsigned long long)(zero_bytes / 16u), (uint4 *)c->map.p,
                     (unsigned long long)(N / 2u), (uint4 *)nullptr, 0ull);
  if (!rows) {
    hipLaunchKernelGGL(mk_fab_summary_kernel, dim3((nseg_total + 3u) / 4u), dim3(256), 0, s, (const uint8_t *)c->text.p, c->hb, nseg_total, (mk_fa_sum *)c->sum.p);
    hipLaunchKernelGGL(mk_fab_scan_kernel, dim3(nfiles), dim3(1024), 0, s, (mk_fa_sum *)c->sum.p, c->hb, TL, pitch);
    hipLaunchKernelGGL(mk_fab_emit_kernel, dim3((nseg_total + 3u) / 4u), dim3(256), 0, s, (const uint8_t *)c->text.p, c->hb, nseg_total,
                       (const mk_fa_sum *)c->sum.p, (uint8_t *)c->stream.p);
  }
  tick("clear + FASTA walk queued");
  MK_HIP(e, hipGetLastError());
  {
    const int tuned = (e->P.subk == 6 && e->P.k >= 9 && e->P.k <= 11) ? 6 : (e->P.subk == 5 && e->P.k == 11) ? 5 : 0;
    const uint64_t per = mk_rows_per_launch(e, rows ? row_bases : rowlen, tuned);
    const uint8_t *src = rows_dev ? rows_dev : (const uint8_t *)c->stream.p;
    e->cur_batch = dbatch;
    for (uint64_t done = 0; done < total_rows && rc == MK_OK; done += per) {
      const uint64_t n = total_rows - done < per ? total_rows - done : per;
      /* (ordinals are rows of the batch: first_ord = the first row of the launch) */
      if (rows) rc = mk_launch_scan_ex(e, src + done * MK_PACKED_PITCH, MK_PACKED_PITCH | format, MK_PACKED_PITCH | format, 0u, n, nullptr, done);
      else rc = mk_launch_scan_ex(e, src + done * pitch, width, pitch, rowlen, n, nullptr, done);
    }
    e->cur_batch = nullptr;
    if (rc) return rc;
  }
  tick("scan queued");
  hipLaunchKernelGGL(mk_b_compact_kernel, dim3((unsigned)e->num_cu * 2u), dim3(1024), 0, s, c->hb);
  hipLaunchKernelGGL(mk_b_layout_kernel, dim3(wide * 2u), dim3(256), 0, s, c->hb);
  hipLaunchKernelGGL(mk_b_bucket_kernel, dim3(wide * 2u), dim3(256), 0, s, c->hb);
  hipLaunchKernelGGL(mk_b_bscan_kernel, dim3(nfiles), dim3(1024), 0, s, c->hb);
  hipLaunchKernelGGL(mk_b_scatter_kernel, dim3(wide * 2u), dim3(256), 0, s, c->hb);
  hipLaunchKernelGGL(mk_b_emit_kernel, dim3(wide * 2u), dim3(256), 0, s, c->hb);
  MK_HIP(e, hipGetLastError());
  tick("finish queued");
  /* the results go home in the same sequence, written by a kernel into the pinned blocks: the counts per file and component and
   * exactly the ids there are (a copy queued by mk_sketch_batch_end would stand behind the NEXT batch's kernels, and the host would
   * hand every result out one batch late with the device idle meanwhile; a copy COMMAND here would have to guess the count, and be
   * the process's first use of the copy engine where the rows are read in place).  The block holds half of every table, 8 M ids at
   * most; a batch with more -- none so far -- gets the rest by a copy in mk_sketch_batch_end. */
  /* (twice the ids the batch is expected to leave, not half of every table: pinned memory costs 0.2 ms a MiB to make, and the first three
   * batches of a run each make a block -- 8.4 MB at L2K11, now 4) */
  c->spec_ids = N / 2u < ((uint64_t)8 << 20) ? N / 2u : ((uint64_t)8 << 20);
  { const uint64_t want = (2u * est_ids + 65536u + 3u) & ~(uint64_t)3u; if (want < c->spec_ids) c->spec_ids = want; }
  if ((rc = mk_pinned_fit(e, (void **)&c->h_ids, &c->h_ids_cap, (size_t)c->spec_ids * 4))) return rc;
  hipLaunchKernelGGL(mk_b_home_kernel, dim3((unsigned)e->num_cu), dim3(256), 0, s, c->hb, (uint4 *)c->h_stat, (uint32_t)(c->stat_bytes / 16u), (uint4 *)c->h_ids,
                     (unsigned long long)c->spec_ids, (const uint32_t *)e->tab.err);
  MK_HIP(e, hipGetLastError());
  MK_HIP(e, hipEventRecord(c->ev_stat, s));
  tick("copies home queued");
  if (trace)
    fprintf(stderr, "[mk batch %llu] device buffers %.3f ms, commands queued in %.3f ms (N %llu, %u ids come home with the batch)\n", (unsigned long long)e->batch_begun,
            (tr1 - tr0) * 1e3, (mk_tick_now() - tr1) * 1e3, (unsigned long long)N, (unsigned)c->spec_ids);
  e->batch_begun++;
  return MK_OK;
}

extern "C" int mk_sketch_batch_begin(mk_engine *e, int mode, const mk_batch_file *files, uint32_t nfiles) { return mk_batch_begin_impl(e, mode, files, nfiles, 0u); }
extern "C" int mk_sketch_batch_begin_rows(mk_engine *e, int mode, uint32_t format, const mk_batch_file *files, uint32_t nfiles) {
  if (format != MK_ROWS_PACKED && format != MK_ROWS_WIDE) return e ? mk_fail(e, MK_ERR_ARG, "mk_sketch_batch_begin_rows: format MK_ROWS_PACKED or MK_ROWS_WIDE") : MK_ERR_ARG;
  return mk_batch_begin_impl(e, mode, files, nfiles, format);
}

extern "C" int mk_sketch_batch_end(mk_engine *e, mk_batch_result *out) {
  if (!e || !out) return MK_ERR_ARG;
  if (e->batch_begun == e->batch_ended) return mk_fail(e, MK_ERR_STATE, "mk_sketch_batch_end without a batch in flight");
  if (e->begun) return mk_fail(e, MK_ERR_STATE, "mk_sketch_batch_end inside a sketch (between begin and finish)");
  MK_HIP(e, hipSetDevice(e->device));
  mk_bctx *c = e->bctx[e->batch_ended % 3u];
  e->batch_ended++; /* whatever happens below, this batch is over */
  MK_HIP(e, hipEventSynchronize(c->ev_stat));
  const uint32_t nfiles = c->nfiles, C = c->hb.comp_num;
  const mk_bstat *st = (const mk_bstat *)c->h_stat;
  const uint32_t *ctot = (const uint32_t *)((const uint8_t *)c->h_stat + mk_up16((size_t)nfiles * sizeof(mk_bstat)));
  const unsigned long long *misc = (const unsigned long long *)((const uint8_t *)ctot + mk_up16((size_t)nfiles * C * 4));
  const uint64_t n_out = misc[1];
  /* a scan kernel of this batch gave up (it sets mk_table::err and returns: the sketches would be empty or partial) */
  if ((uint32_t)misc[2] & 4u) return mk_fail(e, MK_ERR_HIP, "batch: scan kernel: LDS filter not at offset 0");
  if ((uint32_t)misc[2] & ~(1u | 2u | 8u)) return mk_fail(e, MK_ERR_HIP, "batch: a kernel of the batch reported error flags 0x%x", (unsigned)misc[2]);
  if (n_out > c->hb.out_cap || misc[0] > c->hb.list_cap) return mk_fail(e, MK_ERR_HIP, "batch: %llu keys, %llu ids for lists of %llu", misc[0], (unsigned long long)n_out, (unsigned long long)c->hb.list_cap);
  int rc = MK_OK;
  const bool more_ids = n_out > c->spec_ids; /* more than came with the batch's own sequence: the rest now */
  if (more_ids) {
    uint32_t *bigger = nullptr;
    MK_HIP(e, mk_pin_alloc((void **)&bigger, (size_t)n_out * 4 + 4096, hipHostMallocDefault));
    memcpy(bigger, c->h_ids, (size_t)c->spec_ids * 4);
    hipHostFree(c->h_ids);
    c->h_ids = bigger; c->h_ids_cap = (size_t)n_out * 4 + 4096;
    MK_HIP(e, hipMemcpyAsync(c->h_ids + c->spec_ids, c->hb.out_ids + c->spec_ids, (size_t)(n_out - c->spec_ids) * 4, hipMemcpyDeviceToHost, c->used_stream));
    MK_HIP(e, hipEventRecord(c->ev_stat, c->used_stream));
  }
  c->comps.assign((size_t)nfiles * C, mk_component{nullptr, nullptr, 0});
  c->alone_ids.clear();
  /* files the batch could not take: alone, through the ordinary path (their sketches are the same either way) */
  size_t n_alone = 0;
  for (uint32_t i = 0; i < nfiles; i++) if ((st[i].flags & MK_BF_REDO) && !(st[i].flags & MK_BF_HEADER_END)) n_alone++;
  c->alone_ids.resize(n_alone);
  size_t ai = 0;
  for (uint32_t i = 0; i < nfiles; i++) {
    out[i].status = MK_OK; out[i].alone = 0;
    out[i].r.component_num = (int32_t)C; out[i].r.total = 0; out[i].r.components = c->comps.data() + (size_t)i * C;
    if (st[i].flags & MK_BF_HEADER_END) { out[i].status = MK_ERR_FORMAT; continue; }
    if (!(st[i].flags & MK_BF_REDO)) continue;
    out[i].alone = 1;
    mk_result r;
    rc = mk_sketch_begin(e, c->mode);
    if (rc == MK_OK && c->rows) { /* its rows where the batch's scan read them (wide rows keep their extension rows beside them there) */
      const uint64_t nr = c->files[i].n / MK_PACKED_PITCH, per = mk_rows_per_launch(e, c->format == MK_ROWS_WIDE ? MK_WIDE_MAX_BASES : MK_PACKED_MAX_BASES,
                                                                                 e->P.subk == 6 ? 6 : 5);
      for (uint64_t done = 0; done < nr && rc == MK_OK; done += per)
        rc = mk_launch_scan_ex(e, c->rows_src + c->row_off[i] + done * MK_PACKED_PITCH, MK_PACKED_PITCH | c->format, MK_PACKED_PITCH | c->format, 0u,
                               nr - done < per ? nr - done : per, nullptr, done);
    } else if (rc == MK_OK) rc = mk_sketch_push_stream(e, c->files[i].text, c->files[i].n, 1);
    if (rc == MK_OK) rc = mk_sketch_finish(e, &r);
    else if (e->begun) { mk_result dummy; (void)mk_sketch_finish(e, &dummy); }
    if (rc == MK_ERR_CROWDED || rc == MK_ERR_FORMAT) { out[i].status = rc; ai++; continue; }
    if (rc) return rc;
    std::vector<uint32_t> &v = c->alone_ids[ai++];
    v.reserve((size_t)r.total);
    for (uint32_t k = 0; k < C; k++) v.insert(v.end(), r.components[k].ids, r.components[k].ids + r.components[k].n);
    size_t at = 0;
    for (uint32_t k = 0; k < C; k++) {
      mk_component &mc = c->comps[(size_t)i * C + k];
      mc.n = r.components[k].n; mc.ids = v.data() + at; mc.counts = nullptr;
      at += (size_t)r.components[k].n;
    }
    out[i].r.total = r.total;
  }
  if (more_ids) MK_HIP(e, hipEventSynchronize(c->ev_stat));
  /* the batch's ids lie file by file, component by component */
  size_t at = 0;
  for (uint32_t i = 0; i < nfiles; i++) {
    uint64_t tot = 0;
    for (uint32_t k = 0; k < C; k++) tot += ctot[(size_t)i * C + k];
    if (out[i].alone || out[i].status != MK_OK) { at += (size_t)tot; continue; } /* (flagged files have no entries: tot == 0) */
    for (uint32_t k = 0; k < C; k++) {
      mk_component &mc = c->comps[(size_t)i * C + k];
      mc.n = ctot[(size_t)i * C + k]; mc.ids = c->h_ids + at; mc.counts = nullptr;
      at += (size_t)mc.n;
    }
    out[i].r.total = tot;
  }
  if (at != n_out) return mk_fail(e, MK_ERR_HIP, "batch: %zu ids accounted for, %llu written", at, (unsigned long long)n_out);
  return MK_OK;
}
