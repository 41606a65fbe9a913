/*
 * mk_multi.hip -- libmetakssd_multi.so: several engines of libmetakssd_hip.so on the GPUs of one node and the one exchange
 * that merges their partial sketches on GPU 0 (include/metakssd_multi.h).  RCCL point-to-point inside one group call; plain
 * device copies when the same GPU is named twice or RCCL is not usable.
 */
#include "metakssd_multi.h"
#include "mk_poison.hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>


static thread_local char g_multi_error[512] = "";

struct mk_multi {
  int n = 0;
  std::vector<int> dev;
  std::vector<mk_engine *> eng;
  std::vector<hipStream_t> xs;     /* exchange stream per engine, on its device */
  std::vector<ncclComm_t> comm;    /* empty: device copies */
  bool rccl = false;
  mk_params P{};
  char err[512] = "";
  /* exchange buffers: [i] = export of engine i on its own device; recv = everything back to back on device 0 */
  std::vector<unsigned long long *> xk, xo;
  std::vector<uint32_t *> xc;
  std::vector<uint64_t> xcap;
  unsigned long long *rk = nullptr, *ro = nullptr;
  uint32_t *rc = nullptr;
  uint64_t rcap = 0;
  /* merge by key slices: what engine g receives of the others' parts g, on its own device; part sizes in pinned host memory */
  std::vector<unsigned long long *> sk, so;
  std::vector<uint32_t *> sc;
  std::vector<uint64_t> scap;
  uint64_t *h_parts = nullptr; /* [n][16] */
  int merge = 0;               /* MK_MULTI_MERGE_* */
  const char *last_merge = "gather";
  mk_multi_times times{};
};

static int mm_fail(mk_multi *m, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  snprintf(m ? m->err : g_multi_error, 512, "%s", buf);
  return code;
}
#define MM_HIP(m, call)                                                                                        \
  do {                                                                                                         \
    hipError_t _r = (call);                                                                                    \
    if (_r != hipSuccess) return mm_fail((m), _r == hipErrorOutOfMemory ? MK_ERR_NOMEM : MK_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_r)); \
  } while (0)

static double mm_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

extern "C" const char *mk_multi_last_error(const mk_multi *m) { return m ? m->err : g_multi_error; }
extern "C" int mk_multi_count(const mk_multi *m) { return m ? m->n : 0; }
extern "C" mk_engine *mk_multi_engine(mk_multi *m, int i) { return m && i >= 0 && i < m->n ? m->eng[(size_t)i] : nullptr; }
extern "C" const char *mk_multi_transport(const mk_multi *m) { return m && m->rccl ? "rccl" : "device copies"; }

struct mm_create_job { const mk_params *p; int device; mk_engine *e; int rc; char err[512]; pthread_t th; };
static void *mm_create_run(void *arg) {
  mm_create_job *j = (mm_create_job *)arg;
  j->rc = mk_engine_create(j->p, j->device, &j->e);
  if (j->rc != MK_OK) snprintf(j->err, sizeof j->err, "%s", mk_last_error(nullptr));
  return nullptr;
}

extern "C" int mk_multi_destroy(mk_multi *m) {
  if (!m) return MK_ERR_ARG;
  for (int i = 0; i < m->n; i++) {
    if ((size_t)i < m->dev.size()) hipSetDevice(m->dev[(size_t)i]);
    if ((size_t)i < m->xk.size()) { hipFree(m->xk[(size_t)i]); hipFree(m->xc[(size_t)i]); hipFree(m->xo[(size_t)i]); }
    if ((size_t)i < m->xs.size() && m->xs[(size_t)i]) hipStreamDestroy(m->xs[(size_t)i]);
  }
  for (int i = 0; i < m->n; i++)
    if ((size_t)i < m->sk.size() && m->sk[(size_t)i]) { hipSetDevice(m->dev[(size_t)i]); hipFree(m->sk[(size_t)i]); hipFree(m->sc[(size_t)i]); hipFree(m->so[(size_t)i]); }
  if (m->h_parts) hipHostFree(m->h_parts);
  if (!m->dev.empty()) { hipSetDevice(m->dev[0]); hipFree(m->rk); hipFree(m->rc); hipFree(m->ro); }
  for (auto c : m->comm) if (c) ncclCommDestroy(c);
  for (auto e : m->eng) if (e) mk_engine_destroy(e);
  delete m;
  return MK_OK;
}

/* Most engines one sketch may be spread over.  The table slot's count field (24 bits, mk_kernels.hip.h MK_CNT_BITS) stops
 * adding at 0xF00000 and every import adds at most 65535 per key: 16 concurrent imports stay below the key bits with room to
 * spare (16 * 65535 < 2^20 <= 2^24 - 0xF00000), 17 and more could carry. */
#define MK_MULTI_MAX_ENGINES 16
/* the merge by key slices cuts every list into one part per engine: mk_partial_export_split takes at most 16 parts (MK_SPLIT_MAX,
 * mk_kernels.hip.h) and h_parts is laid out [n][16] -- mk_multi_create refuses more engines than that (the command line's --devices list
 * may name 64: the error comes from here, before anything is sketched) */
static_assert(MK_MULTI_MAX_ENGINES <= 16, "mm_finish_slices: one part per engine, at most 16 parts");

extern "C" int mk_multi_create_ex(const mk_params *p, const int *devices, int n, unsigned flags, mk_multi **out) {
  if (!p || !devices || !out || n < 1) return mm_fail(nullptr, MK_ERR_ARG, "mk_multi_create: bad argument");
  if (n > MK_MULTI_MAX_ENGINES) return mm_fail(nullptr, MK_ERR_ARG, "mk_multi_create: %d engines, at most %d", n, MK_MULTI_MAX_ENGINES);
  mk_multi *m = new mk_multi();
  m->n = n;
  m->P = *p;
  m->dev.assign(devices, devices + n);
  m->eng.assign((size_t)n, nullptr);
  /* the engines concurrently: 40 ms each of queue creation, table upload and allocations */
  std::vector<mm_create_job> job((size_t)n);
  for (int i = 0; i < n; i++) {
    job[(size_t)i].p = p; job[(size_t)i].device = devices[i]; job[(size_t)i].e = nullptr; job[(size_t)i].rc = MK_OK; job[(size_t)i].err[0] = 0;
    if (i == 0 || pthread_create(&job[(size_t)i].th, nullptr, mm_create_run, &job[(size_t)i]) != 0) { job[(size_t)i].th = 0; }
  }
  mm_create_run(&job[0]);
  for (int i = 1; i < n; i++) {
    if (job[(size_t)i].th) pthread_join(job[(size_t)i].th, nullptr);
    else mm_create_run(&job[(size_t)i]);
  }
  int rc = MK_OK;
  for (int i = 0; i < n; i++) {
    m->eng[(size_t)i] = job[(size_t)i].e;
    if (job[(size_t)i].rc != MK_OK && rc == MK_OK) { rc = job[(size_t)i].rc; mm_fail(nullptr, rc, "engine on device %d: %s", devices[i], job[(size_t)i].err); }
  }
  if (rc != MK_OK) { mk_multi_destroy(m); return rc; }
  m->xs.assign((size_t)n, nullptr);
  m->xk.assign((size_t)n, nullptr); m->xc.assign((size_t)n, nullptr); m->xo.assign((size_t)n, nullptr);
  m->xcap.assign((size_t)n, 0);
  for (int i = 0; i < n; i++) {
    if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&m->xs[(size_t)i], hipStreamNonBlocking) != hipSuccess) {
      mm_fail(nullptr, MK_ERR_HIP, "exchange stream on device %d", devices[i]);
      mk_multi_destroy(m);
      return MK_ERR_HIP;
    }
  }
  /* RCCL needs every rank on its own device.  A list that names one GPU several times (how this path is tested on a one-GPU
   * box) can only use device copies; on DISTINCT devices a failing RCCL is an error unless the caller asked for copies:
   * a broken RCCL set-up must not pass silently for the xGMI exchange it was supposed to be */
  bool distinct = n > 1;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++)
      if (devices[i] == devices[j]) distinct = false;
  if (distinct && !(flags & MK_MULTI_FORCE_DEVICE_COPIES)) {
    m->comm.assign((size_t)n, nullptr);
    const ncclResult_t r = ncclCommInitAll(m->comm.data(), n, devices);
    if (r == ncclSuccess) m->rccl = true;
    else {
      for (auto &c : m->comm) c = nullptr;
      m->comm.clear();
      if (!(flags & MK_MULTI_ALLOW_DEVICE_COPIES)) {
        mm_fail(nullptr, MK_ERR_HIP, "ncclCommInitAll over %d devices failed: %s (pass MK_MULTI_ALLOW_DEVICE_COPIES / --allow-device-copies "
                                     "to exchange with hipMemcpyPeerAsync instead)", n, ncclGetErrorString(r));
        mk_multi_destroy(m);
        return MK_ERR_HIP;
      }
    }
  }
  if (const char *mg = getenv("MK_MULTI_MERGE")) /* "gather" / "slices": mk_multi_set_merge for callers that cannot call it (the command line's --devices) */
    m->merge = !strcmp(mg, "gather") ? MK_MULTI_MERGE_GATHER : !strcmp(mg, "slices") ? MK_MULTI_MERGE_SLICES : MK_MULTI_MERGE_AUTO;
  if (n > 1)
    fprintf(stderr, "metakssd multi: %d engines, exchange transport: %s, merge: %s\n", n, m->rccl ? "rccl" : "device copies",
            m->merge == MK_MULTI_MERGE_GATHER || (m->merge == MK_MULTI_MERGE_AUTO && n < 4) ? "gather to engine 0" : "key slices (all-to-all, then gather)");
  *out = m;
  return MK_OK;
}

extern "C" int mk_multi_create(const mk_params *p, const int *devices, int n, mk_multi **out) {
  /* MK_MULTI_ALLOW_COPIES=1 in the environment: the opt-in for callers that cannot pass flags */
  const char *env = getenv("MK_MULTI_ALLOW_COPIES");
  return mk_multi_create_ex(p, devices, n, env && env[0] == '1' ? MK_MULTI_ALLOW_DEVICE_COPIES : 0u, out);
}

extern "C" int mk_multi_begin(mk_multi *m, int mode) {
  if (!m) return MK_ERR_ARG;
  for (int i = 0; i < m->n; i++) {
    int rc = mk_sketch_begin(m->eng[(size_t)i], mode);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  return MK_OK;
}
extern "C" int mk_multi_begin_occ(mk_multi *m, int min_occurrence) {
  if (!m) return MK_ERR_ARG;
  for (int i = 0; i < m->n; i++) {
    /* the occurrence threshold applies to the MERGED counts: the partial sketches keep every key (threshold 1 on 1..) */
    int rc = mk_sketch_begin_occ(m->eng[(size_t)i], i == 0 ? min_occurrence : 1);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  return MK_OK;
}

/* ---- moving lists between the engines' devices: RCCL point-to-point inside ONE group call, or device copies ------------------- */
struct mm_xfer { int src, dst; const void *from; void *to; size_t bytes; };
static int mm_move(mk_multi *m, const std::vector<mm_xfer> &x) {
  if (x.empty()) return MK_OK;
  if (m->rccl) {
    ncclResult_t r = ncclGroupStart();
    for (size_t i = 0; i < x.size() && r == ncclSuccess; i++) {
      if (!x[i].bytes) continue;
      /* matched pairwise in order: the k-th send of rank a to rank b meets the k-th receive of b from a */
      r = ncclSend(x[i].from, x[i].bytes, ncclUint8, x[i].dst, m->comm[(size_t)x[i].src], m->xs[(size_t)x[i].src]);
      if (r == ncclSuccess) r = ncclRecv(x[i].to, x[i].bytes, ncclUint8, x[i].src, m->comm[(size_t)x[i].dst], m->xs[(size_t)x[i].dst]);
    }
    const ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) return mm_fail(m, MK_ERR_HIP, "RCCL exchange: %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
  } else {
    for (size_t i = 0; i < x.size(); i++) {
      if (!x[i].bytes) continue;
      MM_HIP(m, hipSetDevice(m->dev[(size_t)x[i].dst]));
      MM_HIP(m, hipMemcpyPeerAsync(x[i].to, m->dev[(size_t)x[i].dst], x[i].from, m->dev[(size_t)x[i].src], x[i].bytes, m->xs[(size_t)x[i].dst]));
    }
  }
  for (int i = 0; i < m->n; i++) {
    MM_HIP(m, hipSetDevice(m->dev[(size_t)i]));
    MM_HIP(m, hipStreamSynchronize(m->xs[(size_t)i]));
  }
  return MK_OK;
}
static int mm_fit3(mk_multi *m, int dev, unsigned long long **k, uint32_t **c, unsigned long long **o, uint64_t *cap, uint64_t need) {
  if (need <= *cap && *k) return MK_OK;
  MM_HIP(m, hipSetDevice(dev));
  hipFree(*k); hipFree(*c); hipFree(*o);
  *k = nullptr; *c = nullptr; *o = nullptr; *cap = 0;
  const uint64_t want = need + need / 8 + 1024;
  MM_HIP(m, mk_dev_alloc(k, want * 8));
  MM_HIP(m, mk_dev_alloc(c, want * 4));
  MM_HIP(m, mk_dev_alloc(o, want * 8));
  *cap = want;
  return MK_OK;
}

extern "C" int mk_multi_set_merge(mk_multi *m, int how) {
  if (!m || how < MK_MULTI_MERGE_AUTO || how > MK_MULTI_MERGE_SLICES) return MK_ERR_ARG;
  m->merge = how;
  return MK_OK;
}
extern "C" const char *mk_multi_last_merge(const mk_multi *m) { return m ? m->last_merge : ""; }
extern "C" int mk_multi_last_times(const mk_multi *m, mk_multi_times *t) {
  if (!m || !t) return MK_ERR_ARG;
  *t = m->times;
  return MK_OK;
}

/* The merge by key slices (SURVEY.md 8e's alternative; include/metakssd_hip.h "the same merge by key slices"): with n engines the
 * gather makes engine 0 fold n - 1 whole lists into its table (14 M random atomics at n = 8 for BASELINE config 4); here every
 * engine folds an n-th of every list at the same time and engine 0 receives lists of keys that are distinct already. */
static int mm_finish_slices(mk_multi *m, mk_result *out, double *gather_ms, double *tail_ms) {
  const double t0 = mm_now();
  const int n = m->n;
  const uint32_t G = (uint32_t)n;
  if (!m->h_parts) MM_HIP(m, mk_pin_alloc((void **)&m->h_parts, (size_t)n * 16 * sizeof(uint64_t), hipHostMallocDefault));
  if (m->sk.empty()) { m->sk.assign((size_t)n, nullptr); m->sc.assign((size_t)n, nullptr); m->so.assign((size_t)n, nullptr); m->scap.assign((size_t)n, 0); }
  /* 1. every engine: compaction (all queued first), then its list cut into n parts by key % n, in its exchange buffers */
  for (int i = 0; i < n; i++) {
    const int rc = mk_partial_count_begin(m->eng[(size_t)i]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  std::vector<uint64_t> D((size_t)n, 0);
  for (int i = 0; i < n; i++) {
    mk_engine *e = m->eng[(size_t)i];
    int rc = mk_partial_count(e, &D[(size_t)i]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(e));
    rc = mm_fit3(m, m->dev[(size_t)i], &m->xk[(size_t)i], &m->xc[(size_t)i], &m->xo[(size_t)i], &m->xcap[(size_t)i], D[(size_t)i]);
    if (rc) return rc;
    uint64_t got = 0;
    rc = mk_partial_export_split_async(e, G, (uint64_t *)m->xk[(size_t)i], m->xc[(size_t)i], (uint64_t *)m->xo[(size_t)i], m->xcap[(size_t)i],
                                       m->h_parts + (size_t)i * 16, &got);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(e));
  }
  for (int i = 0; i < n; i++) {
    const int rc = mk_engine_sync(m->eng[(size_t)i]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  const double t1 = mm_now();
  /* 2. tables cleared for the slices (queued; runs beside the exchange), part g of every engine to engine g */
  for (int i = 0; i < n; i++) {
    const int rc = mk_partial_restart(m->eng[(size_t)i]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  std::vector<uint64_t> incoming((size_t)n, 0);
  std::vector<mm_xfer> xf;
  for (int g = 0; g < n; g++) {
    uint64_t need = 0;
    for (int i = 0; i < n; i++) if (i != g) need += m->h_parts[(size_t)i * 16 + (size_t)g];
    incoming[(size_t)g] = need;
    const int rc = mm_fit3(m, m->dev[(size_t)g], &m->sk[(size_t)g], &m->sc[(size_t)g], &m->so[(size_t)g], &m->scap[(size_t)g], need ? need : 1);
    if (rc) return rc;
  }
  for (int i = 0; i < n; i++) {
    uint64_t src_at = 0;
    for (int g = 0; g < n; g++) {
      const uint64_t c = m->h_parts[(size_t)i * 16 + (size_t)g];
      if (i != g && c) {
        uint64_t dst_at = 0;
        for (int j = 0; j < i; j++) if (j != g) dst_at += m->h_parts[(size_t)j * 16 + (size_t)g];
        xf.push_back({i, g, m->xk[(size_t)i] + src_at, m->sk[(size_t)g] + dst_at, (size_t)c * 8});
        xf.push_back({i, g, m->xc[(size_t)i] + src_at, m->sc[(size_t)g] + dst_at, (size_t)c * 4});
        xf.push_back({i, g, m->xo[(size_t)i] + src_at, m->so[(size_t)g] + dst_at, (size_t)c * 8});
      }
      src_at += c;
    }
  }
  int rc = mm_move(m, xf);
  if (rc) return rc;
  const double t2 = mm_now();
  /* 3. every engine folds its slice: its own part (where the export left it) and what it received; then the slice's list */
  for (int g = 0; g < n; g++) {
    mk_engine *e = m->eng[(size_t)g];
    uint64_t own_at = 0;
    for (int j = 0; j < g; j++) own_at += m->h_parts[(size_t)g * 16 + (size_t)j];
    const uint64_t own = m->h_parts[(size_t)g * 16 + (size_t)g];
    if (own) rc = mk_partial_import(e, (const uint64_t *)(m->xk[(size_t)g] + own_at), m->xc[(size_t)g] + own_at, (const uint64_t *)(m->xo[(size_t)g] + own_at), own);
    if (rc == MK_OK && incoming[(size_t)g])
      rc = mk_partial_import(e, (const uint64_t *)m->sk[(size_t)g], m->sc[(size_t)g], (const uint64_t *)m->so[(size_t)g], incoming[(size_t)g]);
    if (rc == MK_OK) rc = mk_partial_count_begin(e);
    if (rc) return mm_fail(m, rc, "engine %d: %s", g, mk_last_error(e));
  }
  std::vector<uint64_t> R((size_t)n, 0);
  uint64_t total = 0;
  for (int g = 0; g < n; g++) {
    rc = mk_partial_count(m->eng[(size_t)g], &R[(size_t)g]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", g, mk_last_error(m->eng[(size_t)g]));
    total += R[(size_t)g];
  }
  const double t3 = mm_now();
  /* 4. the reduced slices to engine 0, behind its own in its key list: disjoint key sets, so that list is the sketch's */
  mk_engine *e0 = m->eng[0];
  uint64_t *k0 = nullptr, *o0 = nullptr;
  uint32_t *c0 = nullptr;
  rc = mk_partial_list_reserve(e0, total, &k0, &c0, &o0);
  if (rc) return mm_fail(m, rc, "engine 0: %s", mk_last_error(e0));
  xf.clear();
  uint64_t at = R[0];
  for (int g = 1; g < n; g++) {
    uint64_t *kg = nullptr, *og = nullptr;
    uint32_t *cg = nullptr;
    rc = mk_partial_list_reserve(m->eng[(size_t)g], R[(size_t)g], &kg, &cg, &og);
    if (rc) return mm_fail(m, rc, "engine %d: %s", g, mk_last_error(m->eng[(size_t)g]));
    if (R[(size_t)g]) {
      xf.push_back({g, 0, kg, k0 + at, (size_t)R[(size_t)g] * 8});
      xf.push_back({g, 0, cg, c0 + at, (size_t)R[(size_t)g] * 4});
      xf.push_back({g, 0, og, o0 + at, (size_t)R[(size_t)g] * 8});
    }
    at += R[(size_t)g];
  }
  for (int g = 0; g < n; g++) { /* the slices' lists are complete on their engines' streams */
    rc = mk_engine_sync(m->eng[(size_t)g]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", g, mk_last_error(m->eng[(size_t)g]));
  }
  rc = mm_move(m, xf);
  if (rc) return rc;
  const double t4 = mm_now();
  rc = mk_partial_list_commit(e0, total);
  if (rc == MK_OK) rc = mk_sketch_finish(e0, out);
  if (rc) return mm_fail(m, rc, "%s", mk_last_error(e0));
  const double t5 = mm_now();
  m->times = mk_multi_times{t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0};
  m->last_merge = "slices";
  if (gather_ms) *gather_ms = (t2 - t1) + (t4 - t3);
  if (tail_ms) *tail_ms = t5 - t0;
  return MK_OK;
}

extern "C" int mk_multi_finish(mk_multi *m, mk_result *out, double *gather_ms, double *tail_ms) {
  if (!m || !out) return MK_ERR_ARG;
  /* which merge: the slices pay from four engines on (the fold on engine 0 shrinks by n - 1, two more exchange steps come in) */
  if (m->n > 1 && (m->merge == MK_MULTI_MERGE_SLICES || (m->merge == MK_MULTI_MERGE_AUTO && m->n >= 4))) return mm_finish_slices(m, out, gather_ms, tail_ms);
  const double t0 = mm_now();
  const int n = m->n;
  std::vector<uint64_t> cnt((size_t)n, 0), off((size_t)n, 0);
  uint64_t total = 0;
  /* 1. every other engine: compaction, export into a buffer on its own device.  All compactions are queued first, then
   * every engine is asked for its count (the wait) and its three copies are queued, then one wait per device: the GPUs
   * work at the same time, the host does the bookkeeping in between */
  for (int i = 1; i < n; i++) {
    const int rc = mk_partial_count_begin(m->eng[(size_t)i]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  for (int i = 1; i < n; i++) {
    mk_engine *e = m->eng[(size_t)i];
    uint64_t d = 0;
    int rc = mk_partial_count(e, &d);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(e));
    if (d > m->xcap[(size_t)i]) {
      MM_HIP(m, hipSetDevice(m->dev[(size_t)i]));
      hipFree(m->xk[(size_t)i]); hipFree(m->xc[(size_t)i]); hipFree(m->xo[(size_t)i]);
      m->xk[(size_t)i] = nullptr; m->xc[(size_t)i] = nullptr; m->xo[(size_t)i] = nullptr; m->xcap[(size_t)i] = 0;
      const uint64_t cap = d + d / 8 + 1024;
      MM_HIP(m, mk_dev_alloc(&m->xk[(size_t)i], cap * 8));
      MM_HIP(m, mk_dev_alloc(&m->xc[(size_t)i], cap * 4));
      MM_HIP(m, mk_dev_alloc(&m->xo[(size_t)i], cap * 8));
      m->xcap[(size_t)i] = cap;
    }
    uint64_t got = 0;
    rc = mk_partial_export_async(e, (uint64_t *)m->xk[(size_t)i], m->xc[(size_t)i], (uint64_t *)m->xo[(size_t)i], m->xcap[(size_t)i], &got);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(e));
    cnt[(size_t)i] = got;
    off[(size_t)i] = total;
    total += got;
  }
  for (int i = 1; i < n; i++) {
    const int rc = mk_engine_sync(m->eng[(size_t)i]);
    if (rc) return mm_fail(m, rc, "engine %d: %s", i, mk_last_error(m->eng[(size_t)i]));
  }
  const double t1 = mm_now();
  /* 2. the exchange: all lists back to back on device 0 */
  if (total > m->rcap) {
    MM_HIP(m, hipSetDevice(m->dev[0]));
    hipFree(m->rk); hipFree(m->rc); hipFree(m->ro);
    m->rk = nullptr; m->rc = nullptr; m->ro = nullptr; m->rcap = 0;
    const uint64_t cap = total + total / 8 + 1024;
    MM_HIP(m, mk_dev_alloc(&m->rk, cap * 8));
    MM_HIP(m, mk_dev_alloc(&m->rc, cap * 4));
    MM_HIP(m, mk_dev_alloc(&m->ro, cap * 8));
    m->rcap = cap;
  }
  if (total) {
    if (m->rccl) {
      ncclResult_t r = ncclGroupStart();
      for (int i = 1; i < n && r == ncclSuccess; i++) {
        const uint64_t c = cnt[(size_t)i];
        if (!c) continue;
        /* rank i -> rank 0, three arrays; the receives on rank 0 land at the list's offset */
        r = ncclSend(m->xk[(size_t)i], c * 8, ncclUint8, 0, m->comm[(size_t)i], m->xs[(size_t)i]);
        if (r == ncclSuccess) r = ncclRecv(m->rk + off[(size_t)i], c * 8, ncclUint8, i, m->comm[0], m->xs[0]);
        if (r == ncclSuccess) r = ncclSend(m->xc[(size_t)i], c * 4, ncclUint8, 0, m->comm[(size_t)i], m->xs[(size_t)i]);
        if (r == ncclSuccess) r = ncclRecv(m->rc + off[(size_t)i], c * 4, ncclUint8, i, m->comm[0], m->xs[0]);
        if (r == ncclSuccess) r = ncclSend(m->xo[(size_t)i], c * 8, ncclUint8, 0, m->comm[(size_t)i], m->xs[(size_t)i]);
        if (r == ncclSuccess) r = ncclRecv(m->ro + off[(size_t)i], c * 8, ncclUint8, i, m->comm[0], m->xs[0]);
      }
      const ncclResult_t r2 = ncclGroupEnd();
      if (r != ncclSuccess || r2 != ncclSuccess)
        return mm_fail(m, MK_ERR_HIP, "RCCL exchange: %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
      for (int i = 0; i < n; i++) {
        MM_HIP(m, hipSetDevice(m->dev[(size_t)i]));
        MM_HIP(m, hipStreamSynchronize(m->xs[(size_t)i]));
      }
    } else {
      MM_HIP(m, hipSetDevice(m->dev[0]));
      for (int i = 1; i < n; i++) {
        const uint64_t c = cnt[(size_t)i];
        if (!c) continue;
        MM_HIP(m, hipMemcpyPeerAsync(m->rk + off[(size_t)i], m->dev[0], m->xk[(size_t)i], m->dev[(size_t)i], c * 8, m->xs[0]));
        MM_HIP(m, hipMemcpyPeerAsync(m->rc + off[(size_t)i], m->dev[0], m->xc[(size_t)i], m->dev[(size_t)i], c * 4, m->xs[0]));
        MM_HIP(m, hipMemcpyPeerAsync(m->ro + off[(size_t)i], m->dev[0], m->xo[(size_t)i], m->dev[(size_t)i], c * 8, m->xs[0]));
      }
      MM_HIP(m, hipStreamSynchronize(m->xs[0]));
    }
  }
  const double t2 = mm_now();
  /* 3. one import launch over everything, then the reference-order layout and dump on engine 0 */
  mk_engine *e0 = m->eng[0];
  if (total) {
    int rc = mk_partial_import(e0, (const uint64_t *)m->rk, m->rc, (const uint64_t *)m->ro, total);
    if (rc) return mm_fail(m, rc, "import: %s", mk_last_error(e0));
  }
  const double t3 = mm_now();
  int rc = mk_sketch_finish(e0, out);
  if (rc) return mm_fail(m, rc, "%s", mk_last_error(e0));
  /* the other engines' sketches are spent: leave them ready for the next begin */
  const double t4 = mm_now();
  m->times = mk_multi_times{t1 - t0, t2 - t1, t3 - t2, 0.0, t4 - t3, t4 - t0};
  m->last_merge = "gather";
  if (gather_ms) *gather_ms = t2 - t1;
  if (tail_ms) *tail_ms = t4 - t0;
  return MK_OK;
}
