/*
 * metakssd_main.c -- `metakssd dist` command line on top of the C ABI (host C, links libmetakssd_hip.so).
 *
 * Keeps the reference's CLI surface for the sketching path:
 *     metakssd dist -L <file.shuf> [-A] [-u] [-o outdir] [-p N] <fastq/fasta files or directories>...
 * (option table command_dist_wrapper.c:32-59, defaults :68-96, query-only branch of dist_dispatch()
 * command_dist.c:201-248, run_stageI() :341-500) and writes the same sketch directory
 * (cofiles.stat, combco.N, combco.index.N, combco.N.a).
 *
 * Also built, each on the device behind the same C ABI: FASTQ without -A (-n / -Q, fastq2co), `set -u|-q|-i|-s|-g|-c|-P`,
 * `composite -r -q [-b]` / `-d`, stage II (`dist -o <mco> <sketch dir>`, `dist -L .. -r <genomes> -o <db>`) and the
 * database search `dist -r <mco> -o <out> [-M -O -N -D --correction --keepskf -f] <sketch dir>`.
 *
 * Differences, all documented in DESIGN.md: inputs are processed in discovery order (the reference
 * applies a time-seeded shuffle, command_dist.c:215); --byread, composite -i/-s and reverse are not
 * part of this build; -p N sets the number of host threads that read and frame/window input files ahead of the GPU
 * (default 8); --device selects the GPU.
 */
#define _GNU_SOURCE
#include "metakssd_hip.h"
#include "metakssd_multi.h"
#include "mk_host_internal.h" /* mk_poison_byte(): the MK_POISON test hook */

#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <signal.h>
#include <spawn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

extern char **environ;

#define PATHLEN 256 /* global_basic.h:32 */

static void die(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "metakssd: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  /* not exit(): with driver / worker threads still inside HIP calls the runtime's exit handlers can wait for them forever */
  fflush(NULL);
  _exit(1);
}

static int has_suffix(const char *s, const char *suf) {
  size_t a = strlen(s), b = strlen(suf);
  return a >= b && strcmp(s + a - b, suf) == 0;
}
/* isOK_fmt_infile(): one optional .gz/.bz2, then the format suffix (global_basic.h:162-186, global_basic.c:96-128) */
static int fmt_match(const char *name, const char *const *fmts, int n) {
  char tmp[PATHLEN * 2];
  snprintf(tmp, sizeof tmp, "%s", name);
  if (has_suffix(tmp, ".gz")) tmp[strlen(tmp) - 3] = 0;
  else if (has_suffix(tmp, ".bz2")) tmp[strlen(tmp) - 4] = 0;
  for (int i = 0; i < n; i++) {
    char suf[16];
    snprintf(suf, sizeof suf, ".%s", fmts[i]);
    if (has_suffix(tmp, suf)) return 1;
  }
  return 0;
}
static const char *const FASTA_FMT[] = {"fasta", "fna", "fas", "fa"};
static const char *const FASTQ_FMT[] = {"fq", "fastq"};
static int is_fastq(const char *n) { return fmt_match(n, FASTQ_FMT, 2); }
static int is_fasta(const char *n) { return fmt_match(n, FASTA_FMT, 4); }
static int is_compressed(const char *n) { return has_suffix(n, ".gz") || has_suffix(n, ".bz2"); }

typedef struct { char **v; int n, cap; } strlist;
static void sl_push(strlist *l, const char *s) {
  if (strlen(s) >= PATHLEN) die("path: %s exceed maximal path lenth %d", s, PATHLEN);
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 64; l->v = realloc(l->v, sizeof(char *) * l->cap); }
  l->v[l->n++] = strdup(s);
}
static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

/* organize_infile_frm_arg(): global_basic.c:246-325 (directories expanded one level; here sorted by name) */
static void discover(strlist *out, int argc, char **argv) {
  for (int i = 0; i < argc; i++) {
    struct stat st;
    if (stat(argv[i], &st) != 0) die("%dth argument: can't open %s", i + 1, argv[i]);
    if (S_ISDIR(st.st_mode)) {
      DIR *d = opendir(argv[i]);
      if (!d) die("%dth argument: can't open %s", i + 1, argv[i]);
      strlist tmp = {0};
      struct dirent *de;
      while ((de = readdir(d)) != NULL) {
        char full[PATHLEN * 4];
        snprintf(full, sizeof full, "%s/%s", argv[i], de->d_name);
        if (is_fastq(full) || is_fasta(full)) sl_push(&tmp, full);
      }
      closedir(d);
      qsort(tmp.v, tmp.n, sizeof(char *), cmp_str);
      for (int j = 0; j < tmp.n; j++) { sl_push(out, tmp.v[j]); free(tmp.v[j]); }
      free(tmp.v);
    } else if (is_fastq(argv[i]) || is_fasta(argv[i])) {
      sl_push(out, argv[i]);
    } else {
      die("wrong format %dth argument: %s (supported: .fna .fas .fasta .fq .fastq .fa, optionally .gz/.bz2)", i + 1, argv[i]);
    }
  }
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

#define IOBUF ((size_t)64 << 20)
#define ROWBUF ((size_t)64 << 20)

/* The engine is created on a helper thread (HIP start-up, 2 GB of tables, the .shuf upload) while the main thread maps the
 * input and starts framing; whoever needs the engine first waits for it here. */
typedef struct {
  pthread_t th;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int done, rc, device;
  int ndev, devs[64];   /* --devices: more than one GPU -> libmetakssd_multi.so */
  unsigned multi_flags; /* MK_MULTI_ALLOW_DEVICE_COPIES with --allow-device-copies */
  mk_multi *multi;
  int lazy_tables;      /* MK_ENGINE_LAZY_TABLES: the inputs look like a directory of genomes for batches (set before have_params) */
  int have_params;      /* the main thread has read the .shuf file: P may be used */
  const mk_params *P;
  mk_engine *eng;
  char err[512];
  double t_start, t_hip_ready, t_ready; /* seconds since process start */
} engine_future;

static double g_t0; /* process start (monotonic) */
#ifndef MK_DEFAULT_AHEAD
#define MK_DEFAULT_AHEAD 0
#endif
static uint64_t g_pool_bytes = (uint64_t)6 << 30; /* --pool-mib: room for the FASTQ stream's row buffers (mk_fastq_opts::pool_bytes; 0: a few buffers, reused) */
static int g_mmap_input = 0;  /* --mmap-input: the FASTQ file is mapped and the framers read the mapping (rounds 2-4) instead of pread()ing pieces */
static int g_early_chunks = 96; /* --early-chunks: chunks of a FASTQ file framed before the engine is there (mk_fastq_opts::early_chunks) */
static int g_frame_early = 0; /* --frame-early: every chunk of a FASTQ file may be framed at once, also while the runtime and the engine come up (measurement) */
static int g_ahead = MK_DEFAULT_AHEAD; /* --ahead: row buffers the FASTQ stream's framers may run ahead of the pushes by */
static int g_component_sz = 8; /* --component-sz: the reference's compile-time COMPONENT_SZ (global_basic.h:35-37) */

/* libmetakssd_multi.so (it links librccl.so, 570 MB) is loaded only when --devices names several GPUs */
static struct {
  int (*create)(const mk_params *, const int *, int, unsigned, mk_multi **);
  const char *(*last_error)(const mk_multi *);
  mk_engine *(*engine)(mk_multi *, int);
  const char *(*transport)(const mk_multi *);
  int (*begin)(mk_multi *, int);
  int (*begin_occ)(mk_multi *, int);
  int (*finish)(mk_multi *, mk_result *, double *, double *);
} g_multi;

static void load_multi(void) {
  void *h = dlopen("libmetakssd_multi.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) die("--devices with several GPUs needs libmetakssd_multi.so: %s", dlerror());
  *(void **)&g_multi.create = dlsym(h, "mk_multi_create_ex");
  *(void **)&g_multi.last_error = dlsym(h, "mk_multi_last_error");
  *(void **)&g_multi.engine = dlsym(h, "mk_multi_engine");
  *(void **)&g_multi.transport = dlsym(h, "mk_multi_transport");
  *(void **)&g_multi.begin = dlsym(h, "mk_multi_begin");
  *(void **)&g_multi.begin_occ = dlsym(h, "mk_multi_begin_occ");
  *(void **)&g_multi.finish = dlsym(h, "mk_multi_finish");
  if (!g_multi.create || !g_multi.last_error || !g_multi.engine || !g_multi.transport || !g_multi.begin || !g_multi.begin_occ || !g_multi.finish)
    die("libmetakssd_multi.so: missing symbols");
}

typedef struct {
  engine_future *fut;
  mk_engine *eng; /* NULL until engine_get(); with several GPUs: engine 0, where the sketch is finished */
  mk_multi *multi;
  int ndev;
  mk_engine *engs[64];
  uint64_t rr;    /* row buffers dealt so far (round-robin over the engines) */
  double gather_ms, tail_ms;
  uint8_t *io;   /* raw text */
  uint8_t *rows; /* pinned rows */
  uint64_t next_ordinal;
  uint64_t nrows_total;
  int occ;      /* FASTQ without -A: fastq2co()'s reader (quality mask, its record rule) */
  int qmin, TL; /* -Q, k-mer length */
  int nthreads; /* host threads for framing one big FASTQ (mk_fastq_frame_mt) */
  /* the sketch the current file goes into: begun lazily, by whoever pushes first */
  int begun, mode, min_occ;
  uint64_t chunk_bytes;
  mk_fastq_stats fq_stats;
  double t_first_push, t_last_push, t_unmapped, t_begin_s;
  int drop_pages, inflight, direct_host, packed;
  uint8_t *arena; /* row-buffer pool of the FASTQ stream: an anonymous mapping, pinned piece by piece behind the framers (pin_*) */
  size_t arena_bytes;
  struct pinner *pin;
} ctx_t;

#define CHECK(e, call)                                                  \
  do {                                                                  \
    int _rc = (call);                                                   \
    if (_rc != MK_OK) die("%s failed (%d): %s", #call, _rc, mk_last_error(e)); \
  } while (0)

static void *engine_thread(void *arg) {
  engine_future *f = arg;
  int n = 0;
  f->t_start = now_s() - g_t0;
  int rc = mk_device_count(&n); /* first HIP call of the process: runtime start-up, while the main thread reads the .shuf file */
  f->t_hip_ready = now_s() - g_t0;
  pthread_mutex_lock(&f->mu);
  while (!f->have_params) pthread_cond_wait(&f->cv, &f->mu);
  pthread_mutex_unlock(&f->mu);
  if (rc == MK_OK && !f->P) rc = MK_ERR_ARG; /* the main thread gave up */
  if (rc == MK_OK && f->ndev > 1) {
    rc = g_multi.create(f->P, f->devs, f->ndev, f->multi_flags, &f->multi);
    if (rc != MK_OK) snprintf(f->err, sizeof f->err, "%s", g_multi.last_error(NULL));
    else f->eng = g_multi.engine(f->multi, 0);
  } else if (rc == MK_OK) {
    rc = mk_engine_create_ex(f->P, f->device, f->lazy_tables ? MK_ENGINE_LAZY_TABLES : 0u, &f->eng);
    if (rc != MK_OK) snprintf(f->err, sizeof f->err, "%s", mk_last_error(NULL));
  } else snprintf(f->err, sizeof f->err, "%s", mk_last_error(NULL));
  f->t_ready = now_s() - g_t0;
  pthread_mutex_lock(&f->mu);
  f->rc = rc;
  f->done = 1;
  pthread_cond_broadcast(&f->cv);
  pthread_mutex_unlock(&f->mu);
  return NULL;
}

static void engine_start(engine_future *f, int device, const int *devs, int ndev, unsigned multi_flags) {
  memset(f, 0, sizeof *f);
  f->device = device;
  f->ndev = ndev;
  f->multi_flags = multi_flags;
  for (int i = 0; i < ndev && i < 64; i++) f->devs[i] = devs[i];
  if (ndev > 1) load_multi();
  pthread_mutex_init(&f->mu, NULL);
  pthread_cond_init(&f->cv, NULL);
  if (pthread_create(&f->th, NULL, engine_thread, f) != 0) die("cannot start a thread: %s", strerror(errno));
}

static void engine_params(engine_future *f, const mk_params *P) {
  pthread_mutex_lock(&f->mu);
  f->P = P;
  f->have_params = 1;
  pthread_cond_broadcast(&f->cv);
  pthread_mutex_unlock(&f->mu);
}

/* the per-engine options of the command line, for every engine that is attached to a ctx (the start-up engine, the extra engines
 * of --engines, the drivers' engines of --devices) */
static void engine_apply_options(const ctx_t *c, mk_engine *e) {
  if (c->direct_host) mk_engine_set_option(e, MK_OPT_DIRECT_HOST, 1);
  /* test hook: MK_BATCH_TAB_BITS=<9..22> gives every file of a batch a table of that size (MK_OPT_BATCH_TAB_BITS), so that files
   * overflow it and take the sketched-alone path of mk_sketch_batch_end */
  if (getenv("MK_BATCH_TAB_BITS")) mk_engine_set_option(e, MK_OPT_BATCH_TAB_BITS, atoi(getenv("MK_BATCH_TAB_BITS")));
}

static mk_engine *engine_get(ctx_t *c) {
  if (c->eng) return c->eng;
  engine_future *f = c->fut;
  pthread_mutex_lock(&f->mu);
  while (!f->done) pthread_cond_wait(&f->cv, &f->mu);
  pthread_mutex_unlock(&f->mu);
  if (f->rc != MK_OK) die("mk_engine_create failed (%d): %s", f->rc, f->err);
  c->eng = f->eng;
  c->multi = f->multi;
  c->ndev = f->multi ? f->ndev : 1;
  for (int i = 0; i < c->ndev; i++) c->engs[i] = f->multi ? g_multi.engine(f->multi, i) : f->eng;
  for (int i = 0; i < c->ndev; i++) engine_apply_options(c, c->engs[i]);
  return c->eng;
}

/* mk_sketch_begin for the current file, once, at the first push (or at finish for an input without rows) */
static mk_engine *sketch_engine(ctx_t *c) {
  mk_engine *e = engine_get(c);
  if (!c->begun) {
    const double tb = now_s();
    if (c->multi) {
      const int rc = c->mode == MK_MODE_OCC_SET ? g_multi.begin_occ(c->multi, c->min_occ) : g_multi.begin(c->multi, c->mode);
      if (rc != MK_OK) die("mk_multi_begin failed (%d): %s", rc, g_multi.last_error(c->multi));
    } else if (c->mode == MK_MODE_OCC_SET) CHECK(e, mk_sketch_begin_occ(e, c->min_occ)); /* command_dist.c:385-386 */
    else CHECK(e, mk_sketch_begin(e, c->mode));
    c->t_begin_s += now_s() - tb;
    c->begun = 1;
  }
  return e;
}

/* input opened like the reference does: through `zcat -fc` when compressed (iseq2comem.c:216,666-669), directly otherwise
 * (same bytes, no child process).  zcat is started with an argument vector, not through a shell: a file name is never
 * interpreted; its exit status is checked when the input is closed (a corrupt .gz must not yield a silently short sketch). */
typedef struct { FILE *f; pid_t pid; const char *path; } input_t;

static int open_input(const char *path, input_t *in) {
  in->pid = 0; in->path = path; in->f = NULL;
  if (!is_compressed(path)) { in->f = fopen(path, "rb"); return in->f != NULL; }
  int fd[2];
  if (pipe(fd) != 0) return 0;
  /* posix_spawnp, not fork: the process is heavily threaded and holds a live HIP runtime by the time genomes are read (worker
   * threads beside the engines); a spawned child takes none of that state along */
  posix_spawn_file_actions_t fa;
  if (posix_spawn_file_actions_init(&fa) != 0) { close(fd[0]); close(fd[1]); return 0; }
  posix_spawn_file_actions_adddup2(&fa, fd[1], 1);
  posix_spawn_file_actions_addclose(&fa, fd[0]);
  posix_spawn_file_actions_addclose(&fa, fd[1]);
  char *const zargv[] = {(char *)"zcat", (char *)"-fc", (char *)"--", (char *)path, NULL};
  pid_t pid = 0;
  const int src = posix_spawnp(&pid, "zcat", &fa, NULL, zargv, environ);
  posix_spawn_file_actions_destroy(&fa);
  if (src != 0) { close(fd[0]); close(fd[1]); errno = src; return 0; }
  close(fd[1]);
  in->pid = pid;
  in->f = fdopen(fd[0], "rb");
  return in->f != NULL;
}

static void close_input(input_t *in) {
  if (in->f) fclose(in->f);
  if (in->pid > 0) {
    int st = 0;
    if (waitpid(in->pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
      die("%s: `zcat -fc` failed (status %d): the input was not read completely", in->path, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
  }
}

/* buffers of the windowed paths (pipes, FASTA); a mapped FASTQ file does not need them */
/* MK_POISON (mk_host_internal.h): a pool buffer taken back for another file / batch is overwritten with the pattern first */
static void repoison(void *p, size_t n) { const int pz = mk_poison_byte(); if (pz >= 0 && p && n) memset(p, pz, n); }

static void ensure_buffers(ctx_t *c) {
  if (c->io) return;
  c->io = malloc(IOBUF);
  if (!c->io || mk_host_alloc((void **)&c->rows, ROWBUF) != MK_OK) die("out of memory");
}

static void push_rows(ctx_t *c, const uint8_t *rows, uint32_t stride, uint64_t nrows) {
  mk_engine *e = sketch_engine(c);
  CHECK(e, mk_sketch_push_reads(e, rows, stride, nrows, c->next_ordinal));
  c->next_ordinal += nrows;
  c->nrows_total += nrows;
  if (rows == c->rows) repoison(c->rows, ROWBUF); /* (push returned: the rows have been copied) */
}

/* ---- the row-buffer arena is pinned PIECE BY PIECE, behind the framers and in front of the pushes -----------------------------
 * The arena has room for the whole file's rows (mk_fastq_opts::pool_bytes), so that the framers run at their own pace from the first
 * moment -- also through the 80-110 ms in which the HIP runtime and the engine come up and nothing can be pinned or pushed.  Pinning
 * all of it up front would cost more than it saves (fresh pages pin at 5 GB/s, and pinning beside the creation of the engine's queue
 * makes that three times as long); pages the framers have WRITTEN pin at over 100 GB/s.  So: a framer reports a finished buffer
 * (mk_rows_sink::ready); a pinner thread -- idle until the engine is there -- registers runs of finished buffers that lie side by
 * side (up to 256 MiB a call); a push waits until its buffer is pinned.  The pinner is twice as fast as the link, so after its first
 * call it stays in front of the pushes.  A registration always ends at a buffer's end: a copy whose source lies across two
 * registrations is refused by the runtime (invalid argument), so the unit is the buffer, and a new file -- whose buffers may have
 * another size -- starts from nothing pinned (pin_reset). */
#define PIN_RUN_BYTES ((size_t)256 << 20)
typedef struct pinner {
  uint8_t *base;
  size_t bytes, buf_bytes, nbuf;  /* buf_bytes: learnt from the first report after a reset (all buffers of a stream have one size) */
  size_t kept_buf_bytes;          /* the size the standing registrations and states were made for */
  uint8_t *state;                 /* per buffer: 0 fresh, 1 written (may be pinned), 2 pinned */
  size_t state_cap;
  void **regs; int nregs, regs_cap; /* bases of the registrations made (to undo them) */
  int device, go, stop, started, busy;
  pthread_t th;
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_pinned;
  double t_pin_s; int calls; size_t pinned_bytes;
} pinner;
static void *pinner_run(void *arg) {
  pinner *p = arg;
  pthread_mutex_lock(&p->mu);
  for (;;) {
    size_t b0 = p->nbuf;
    if (p->go) for (size_t b = 0; b < p->nbuf; b++) if (p->state[b] == 1) { b0 = b; break; }
    if (p->stop) break;
    if (b0 == p->nbuf) { pthread_cond_wait(&p->cv_work, &p->mu); continue; }
    size_t b1 = b0;
    while (b1 < p->nbuf && (b1 - b0 + 1) * p->buf_bytes <= PIN_RUN_BYTES + p->buf_bytes && p->state[b1] == 1) b1++;
    uint8_t *at = p->base + b0 * p->buf_bytes;
    const size_t len = (b1 - b0) * p->buf_bytes;
    p->busy = 1;
    pthread_mutex_unlock(&p->mu);
    const double t0 = now_s();
    if (mk_host_register_on(p->device, at, len) != MK_OK) die("pinning the row buffers failed: %s", mk_last_error(NULL));
    const double dt = now_s() - t0;
    pthread_mutex_lock(&p->mu);
    p->busy = 0;
    for (size_t b = b0; b < b1; b++) p->state[b] = 2;
    if (p->nregs == p->regs_cap) {
      p->regs_cap = p->regs_cap ? 2 * p->regs_cap : 64;
      p->regs = realloc(p->regs, sizeof(void *) * (size_t)p->regs_cap);
      if (!p->regs) die("out of memory");
    }
    p->regs[p->nregs++] = at;
    p->t_pin_s += dt; p->calls++; p->pinned_bytes += len;
    pthread_cond_broadcast(&p->cv_pinned);
  }
  pthread_mutex_unlock(&p->mu);
  return NULL;
}
/* the buffer at `at` (room `bytes`): its index; the first report after a reset fixes the buffers' size.  Called with the lock held. */
static size_t pin_index(pinner *p, const uint8_t *at, size_t bytes) {
  if (!p->buf_bytes && bytes == p->kept_buf_bytes && p->nbuf) { /* the same layout as the last stream's: its pins stand */
    p->buf_bytes = bytes;
    for (size_t b = 0; b < p->nbuf; b++) if (p->state[b] == 1) p->state[b] = 0; /* (written by the last stream, never pushed: to be written again) */
  }
  if (!p->buf_bytes) {
    /* another layout: a registration must not lie across a buffer's end, so the old ones go (the pinner is idle: go == 0) */
    for (int i = 0; i < p->nregs; i++) mk_host_unregister(p->regs[i]);
    p->nregs = 0;
    p->kept_buf_bytes = bytes;
    p->buf_bytes = bytes;
    p->nbuf = p->bytes / bytes;
    if (p->nbuf > p->state_cap) {
      free(p->state);
      p->state = calloc(p->nbuf, 1);
      if (!p->state) die("out of memory");
      p->state_cap = p->nbuf;
    } else memset(p->state, 0, p->nbuf);
  }
  if (bytes != p->buf_bytes || (size_t)(at - p->base) % p->buf_bytes) die("internal: row buffers of two sizes in one stream");
  return (size_t)(at - p->base) / p->buf_bytes;
}
static void pin_mark(pinner *p, const uint8_t *at, size_t bytes) { /* the buffer at `at` has been written: it may be pinned */
  if (!p || at < p->base || bytes == 0) return;
  pthread_mutex_lock(&p->mu);
  const size_t b = pin_index(p, at, bytes);
  if (b < p->nbuf && p->state[b] == 0) { p->state[b] = 1; if (p->go) pthread_cond_signal(&p->cv_work); }
  pthread_mutex_unlock(&p->mu);
}
static void pin_wait(pinner *p, const uint8_t *at) { /* returns when the buffer at `at` is pinned; lets the pinner loose */
  if (!p || at < p->base) return;
  pthread_mutex_lock(&p->mu);
  if (!p->buf_bytes) die("internal: a push before any row buffer was reported");
  const size_t b = (size_t)(at - p->base) / p->buf_bytes;
  if (b < p->nbuf) {
    if (p->state[b] == 0) p->state[b] = 1;
    p->go = 1;
    pthread_cond_signal(&p->cv_work);
    while (p->state[b] != 2) pthread_cond_wait(&p->cv_pinned, &p->mu);
  }
  pthread_mutex_unlock(&p->mu);
}
static void pin_reset(pinner *p) { /* a new stream into the same arena: its buffers may have another size (pin_index decides) */
  if (!p) return;
  pthread_mutex_lock(&p->mu);
  p->go = 0;
  while (p->busy) pthread_cond_wait(&p->cv_pinned, &p->mu);
  p->buf_bytes = 0;
  pthread_mutex_unlock(&p->mu);
}
static void pin_destroy(pinner *p) { /* registrations undone, thread gone (the mapping is the caller's) */
  if (!p) return;
  pthread_mutex_lock(&p->mu);
  p->stop = 1;
  pthread_cond_broadcast(&p->cv_work);
  pthread_mutex_unlock(&p->mu);
  if (p->started) pthread_join(p->th, NULL);
  for (int i = 0; i < p->nregs; i++) mk_host_unregister(p->regs[i]);
  free(p->regs); free(p->state);
  pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_work); pthread_cond_destroy(&p->cv_pinned);
  free(p);
}
static pinner *pin_create(uint8_t *base, size_t bytes, int device) {
  pinner *p = calloc(1, sizeof *p);
  if (!p) return NULL;
  p->base = base; p->bytes = bytes; p->device = device;
  pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_work, NULL); pthread_cond_init(&p->cv_pinned, NULL);
  if (pthread_create(&p->th, NULL, pinner_run, p) != 0) { free(p); return NULL; }
  p->started = 1;
  return p;
}

/* ---- sink of the whole-file FASTQ stream: pinned buffers, asynchronous pushes, the engine awaited at the first push ---- */
static void cli_sink_ready(void *ctx, const uint8_t *rows, size_t bytes) { pin_mark(((ctx_t *)ctx)->pin, rows, bytes); }
static int cli_sink_push(void *ctx, const uint8_t *rows, uint32_t stride, uint64_t nrows, uint64_t ord, uint64_t *token) {
  ctx_t *c = ctx;
  (void)sketch_engine(c);
  /* the engine is there, i.e. the runtime is up: the pinner may work; this buffer's pages before anything else */
  pin_wait(c->pin, rows);
  if (c->t_first_push == 0) c->t_first_push = now_s() - g_t0;
  /* several GPUs: the row buffers are dealt round-robin, so that every GPU's PCIe link carries a share at any time;
   * ordinals are global, so it does not matter which engine sees which rows */
  const int k = (int)(c->rr++ % (uint64_t)c->ndev);
  uint64_t t = 0;
  const int rc = mk_sketch_push_reads_async(c->engs[k], rows, stride, nrows, ord, &t);
  if (rc != MK_OK && c->ndev > 1) fprintf(stderr, "metakssd: GPU %d: %s\n", k, mk_last_error(c->engs[k]));
  *token = t * 64u + (uint64_t)k;
  return rc;
}
static int cli_sink_wait(void *ctx, uint64_t token) { return mk_sketch_push_wait(((ctx_t *)ctx)->engs[token % 64u], token / 64u); }
/* the row-buffer pool while the HIP runtime is still coming up: the same mapping mk_host_arena_alloc makes (2 MiB granules,
 * huge pages asked for, touched by eight threads), not pinned yet -- the framers fill it while the runtime and the engine start,
 * and the first push pins it (cli_sink_push) */
typedef struct { uint8_t *p; size_t n; } touch_job;
static void *touch_run(void *arg) {
  touch_job *j = arg;
  for (size_t off = 0; off < j->n; off += 4096) j->p[off] = 0;
  return NULL;
}
static uint8_t *arena_map_untouched(size_t bytes, size_t *len_out) {
  const size_t huge = (size_t)2 << 20;
  const size_t len = (bytes + huge - 1) & ~(huge - 1);
  uint8_t *m = mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return NULL;
#ifdef MADV_HUGEPAGE
  (void)madvise(m, len, MADV_HUGEPAGE);
#endif
  *len_out = len;
  return m;
}
static uint8_t *arena_map_unpinned(size_t bytes, size_t *len_out) {
  size_t len = 0;
  uint8_t *m = arena_map_untouched(bytes, &len);
  if (!m) return NULL;
  enum { T = 8 };
  touch_job job[T];
  pthread_t th[T];
  int started = 0;
  for (int t = 0; t < T; t++) {
    const size_t lo = len / T * (size_t)t, hi = t + 1 == T ? len : len / T * (size_t)(t + 1);
    job[t].p = m + lo; job[t].n = hi - lo;
    if (t + 1 < T && pthread_create(&th[t], NULL, touch_run, &job[t]) == 0) started |= 1 << t;
    else touch_run(&job[t]);
  }
  for (int t = 0; t < T; t++) if (started & (1 << t)) pthread_join(th[t], NULL);
  *len_out = len;
  return m;
}
static uint8_t *cli_sink_alloc(void *ctx, size_t bytes) { /* the arena (and what is pinned of it) is kept for the next file and goes with the process */
  ctx_t *c = ctx;
  if (c->arena && c->arena_bytes >= bytes) { pin_reset(c->pin); return c->arena; } /* (MK_POISON: the stream fills every buffer it takes) */
  if (c->pin) { pin_destroy(c->pin); c->pin = NULL; }
  if (c->arena) munmap(c->arena, c->arena_bytes);
  c->arena = NULL; c->arena_bytes = 0;
  /* an untouched anonymous mapping (2 MiB granules, huge pages asked for): the framers' stores bring the pages in, the pinner
   * registers them once they are written -- nothing here waits for the engine or costs anything for room that is never used */
  size_t len = 0;
  uint8_t *m = arena_map_untouched(bytes, &len);
  if (!m) return NULL;
  c->pin = pin_create(m, len, c->fut ? (c->fut->ndev > 1 ? c->fut->devs[0] : c->fut->device) : 0);
  if (!c->pin) { munmap(m, len); return NULL; }
  c->arena = m; c->arena_bytes = len;
  return m;
}
static void cli_sink_release(void *ctx, uint8_t *p, size_t bytes) { (void)ctx; (void)p; (void)bytes; }

/* an uncompressed regular file is framed straight out of its page-cache mapping by the whole-file stream
 * (mk_fastq_stream.c): no copy into an I/O buffer, `-p` threads fault the pages in and frame concurrently, the buffers
 * go to the engine in file order while later chunks are still being framed */
static int sketch_fastq_mapped(ctx_t *c, const char *path) {
  if (is_compressed(path)) return 0;
  int fd = open(path, O_RDONLY);
  if (fd < 0) return 0;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) { close(fd); return 0; }
  const size_t size = (size_t)st.st_size;
  /* the framers pread the file piece by piece (mk_fastq_opts::fd): nothing of it is mapped into this process, so thirty-two threads
   * at memory speed cause no page-table work and no TLB shoot-downs beside the HIP runtime's start-up.  --mmap-input: the mapping of
   * rounds 2-4 (populated and dropped chunk by chunk by the framers), for comparison */
  const uint8_t *map = NULL;
  if (g_mmap_input) {
    map = mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    fd = -1;
    if (map == MAP_FAILED) return 0;
  }
  mk_fastq_opts o;
  memset(&o, 0, sizeof o);
  o.occ = c->occ; o.qmin = c->qmin; o.TL = c->TL;
  o.nthreads = c->nthreads; o.inflight = c->inflight; o.chunk_bytes = c->chunk_bytes; o.ahead = g_ahead;
  o.packed = c->packed; /* reads of up to 152 bases cross PCIe as 64-byte packed rows (geometries with a tuned scan kernel) */
  o.drop_pages = c->drop_pages; /* a private read-only file mapping that is unmapped below */
  o.pool_bytes = g_pool_bytes;
  o.fd = map ? 0 : fd;
  /* while the HIP runtime and the engine come up only a handful of chunks are framed (thirty-two framers at memory speed make that
   * start-up two to eight times as long, profiles/r05_e2e_front_end.txt); the rest from the moment the first buffer has been pushed */
  o.early_chunks = g_frame_early ? 0 : g_early_chunks;
  mk_rows_sink sink = {c, cli_sink_push, cli_sink_wait, cli_sink_alloc, cli_sink_release, cli_sink_ready};
  mk_fastq_stats fs;
  /* the framers start at once, into a pool that is pinned when the engine is there (cli_sink_alloc / cli_sink_push): the
   * engine's queue creation takes three times as long (45 ms instead of 14) when it runs into the driver together with the
   * pinning, so the pinning waits for it -- the framing does not */
  const int rc = mk_fastq_stream(map, size, &o, &sink, c->next_ordinal, &fs);
  if (rc == MK_ERR_ARG || rc == MK_ERR_FORMAT)
    die("%s: FASTQ line longer than the reference's fgets() width (%s): outside the framing contract", path,
        c->occ ? "19998 characters, iseq2comem.c:319,343" : "4094 characters, iseq2comem.c:656,673");
  if (rc != MK_OK) {
    engine_get(c); /* no usable device: say that (pinned buffers cannot be had either), not "out of memory" */
    die("mk_fastq_stream failed (%d): %s", rc, mk_last_error(c->eng));
  }
  c->next_ordinal += fs.rows;
  c->nrows_total += fs.rows;
  c->fq_stats = fs;
  c->t_last_push = now_s() - g_t0;
  if (map) munmap((void *)map, size);
  else close(fd);
  c->t_unmapped = now_s() - g_t0;
  return 1;
}

static void sketch_fastq(ctx_t *c, const char *path) {
  if (sketch_fastq_mapped(c, path)) return;
  ensure_buffers(c);
  input_t in;
  if (!open_input(path, &in)) die("mtfastq2koc():%s: %s", path, strerror(errno));
  FILE *f = in.f;
  uint32_t stride = 160;
  size_t have = 0;
  int eof = 0;
  uint64_t records = 0;
  while (!eof || have) {
    if (!eof) {
      size_t r = fread(c->io + have, 1, IOBUF - have, f);
      have += r;
      if (r == 0) eof = 1;
    }
    size_t off = 0;
    for (;;) {
      uint64_t nrows = 0;
      size_t used = 0;
      uint64_t nrec = 0;
      /* every call starts at a record boundary (what is left of the buffer is moved to its front), which is what the
       * threaded framer needs */
      int rc = mk_fastq_frame_mt(c->io + off, have - off, eof, c->occ, c->qmin, c->TL, records, c->rows, stride, ROWBUF / stride,
                                 c->nthreads, &nrows, &nrec, &used);
      records += nrec;
      if (nrows) push_rows(c, c->rows, stride, nrows);
      off += used;
      if (rc == MK_ERR_ARG && stride < 4096) { stride = stride * 2 > 4096 ? 4096 : MK_ROW_PITCH(stride * 2); continue; } /* longer read: widen rows */
      if (rc == MK_ERR_ARG || rc == MK_ERR_FORMAT)
        die("%s: FASTQ line longer than the reference's fgets() width (%s): outside the framing contract", path,
            c->occ ? "19998 characters, iseq2comem.c:319,343" : "4094 characters, iseq2comem.c:656,673");
      if (rc != MK_OK) die("mk_fastq_frame failed (%d)", rc);
      if (nrows == 0 || off >= have) break;
    }
    memmove(c->io, c->io + off, have - off);
    have -= off;
    if (have == IOBUF) die("%s: a single FASTQ record exceeds the %zu-byte I/O buffer", path, IOBUF);
    if (eof && have && off == 0) break; /* trailing partial record: dropped like the reference does */
  }
  close_input(&in);
}

static int g_host_fasta = 0; /* --host-fasta: window FASTA text on the host (mk_fasta_window) instead of on the device */

static void sketch_fasta(ctx_t *c, const char *path, int TL) {
  ensure_buffers(c);
  input_t in;
  if (!open_input(path, &in)) die("fasta2co():%s: %s", path, strerror(errno));
  FILE *f = in.f;
  if (!g_host_fasta) {
    /* the file's bytes as they are, piece by piece: the device drops line ends and header lines (mk_sketch_push_stream) */
    mk_engine *e = sketch_engine(c);
    int any = 0;
    for (;;) {
      const size_t have = fread(c->io, 1, IOBUF, f);
      if (have == 0) break;
      any = 1;
      CHECK(e, mk_sketch_push_stream(e, c->io, have, 0));
    }
    if (!any) die("fastco():eof or fread error file=%s", path); /* iseq2comem.c:235 */
    CHECK(e, mk_sketch_push_stream(e, NULL, 0, 1));
    close_input(&in);
    return;
  }
  const uint32_t stride = MK_ROW_PITCH(512u); /* 528: not a multiple of 128 */
  mk_fasta_state st;
  if (mk_fasta_window_init(&st, TL) != MK_OK) die("mk_fasta_window_init failed");
  int eof = 0, any = 0;
  while (!eof) {
    size_t have = fread(c->io, 1, IOBUF, f);
    if (have == 0) eof = 1; else any = 1;
    size_t off = 0;
    do {
      uint64_t nrows = 0;
      size_t used = 0;
      const int wrc = mk_fasta_window(&st, c->io + off, have - off, eof, c->rows, stride, ROWBUF / stride, &nrows, &used);
      if (wrc == MK_ERR_FORMAT) die("fasta2co(): can not find seqences head start from '>' 0 (%s ends inside a header line)", path); /* iseq2comem.c:269 */
      if (wrc != MK_OK) die("mk_fasta_window failed (%d)", wrc);
      if (nrows) push_rows(c, c->rows, stride, nrows);
      off += used;
    } while (off < have);
  }
  if (!any) die("fastco():eof or fread error file=%s", path); /* iseq2comem.c:235 */
  close_input(&in);
}

/* ---- parallel front end for many small inputs (genome directories, many small FASTQ files) ---------------
 * The reference parallelises stage I over FILES (command_dist.c:363-366).  Here worker threads read and frame /
 * window whole files into pinned row buffers ahead of the main thread, which pushes and finishes them strictly
 * in input order (the order defines the sketch directory).  A file whose rows do not fit one buffer is left to
 * the main thread's streaming path.  Buffers are handed out in file order, so the pipeline cannot deadlock. */
#define PF_MAX_BUFS 16
typedef struct {
  int ready, too_big, err;   /* err: MK_ERR_* from framing */
  int buf;                   /* pinned buffer index, -1 = none */
  uint64_t nrows;
  uint32_t stride;
  uint64_t text_bytes;       /* FASTA for the device stream: the buffer holds the file's bytes, not rows */
  int is_text;
} pf_slot;
typedef struct {
  strlist *files;
  int TL;
  int qmin;      /* -Q (FASTQ without -A) */
  int koc_until; /* FASTQ files with an index below this are read the -A way (mt_shortreads2koc's reader), the others fastq2co's */
  int nbufs;
  uint8_t *bufs[PF_MAX_BUFS];
  int free_bufs[PF_MAX_BUFS], nfree;
  int next;                  /* next file index to prepare */
  pf_slot *slots;
  pthread_mutex_t mu;
  pthread_cond_t cv_buf, cv_ready;
} pf_t;

static uint8_t *slurp(const char *path, size_t *n_out, size_t limit, int *too_big) {
  input_t in;
  if (!open_input(path, &in)) return NULL;
  size_t cap = (size_t)8 << 20, n = 0;
  uint8_t *b = malloc(cap);
  if (!b) die("out of memory");
  for (;;) {
    if (n == cap) {
      if (cap >= limit) { *too_big = 1; break; }
      cap *= 2;
      uint8_t *nb = realloc(b, cap);
      if (!nb) die("out of memory");
      b = nb;
    }
    size_t r = fread(b + n, 1, cap - n, in.f);
    if (r == 0) break;
    n += r;
  }
  if (*too_big && in.pid > 0) { /* the rest is read again by the streaming path: do not wait for zcat to push it all out */
    fclose(in.f); in.f = NULL;
    kill(in.pid, SIGTERM);
    waitpid(in.pid, NULL, 0);
  } else close_input(&in);
  *n_out = n;
  return b;
}

/* the bytes of an uncompressed regular file straight into `dst` (a pinned row buffer): 1 = done, 0 = not that kind of file or
 * larger than cap */
static int slurp_into(const char *path, uint8_t *dst, size_t cap, size_t *n_out) {
  if (is_compressed(path)) return 0;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return 0;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size > cap) { close(fd); return 0; }
  size_t n = 0;
  while (n < (size_t)st.st_size) {
    const ssize_t r = read(fd, dst + n, (size_t)st.st_size - n);
    if (r <= 0) break;
    n += (size_t)r;
  }
  close(fd);
  if (n != (size_t)st.st_size) return 0;
  *n_out = n;
  return 1;
}

static void *pf_worker(void *arg) {
  pf_t *pf = arg;
  for (;;) {
    pthread_mutex_lock(&pf->mu);
    while (pf->next < pf->files->n && pf->nfree == 0) pthread_cond_wait(&pf->cv_buf, &pf->mu);
    if (pf->next >= pf->files->n) { pthread_mutex_unlock(&pf->mu); return NULL; }
    const int i = pf->next++;
    const int b = pf->free_bufs[--pf->nfree];
    pthread_mutex_unlock(&pf->mu);
    repoison(pf->bufs[b], ROWBUF);

    pf_slot s = {0};
    s.buf = b;
    const char *path = pf->files->v[i];
    size_t n = 0;
    int too_big = 0;
    if (!g_host_fasta && !is_fastq(path)) {
      /* FASTA for the device stream: the file's bytes go into the pinned buffer as they are (no parsing on the host) */
      if (slurp_into(path, pf->bufs[b], ROWBUF, &n)) {
        if (n == 0) s.err = MK_ERR_STATE; /* empty input: the reference's "eof or fread error" (iseq2comem.c:235) */
        s.is_text = 1; s.text_bytes = n;
        pthread_mutex_lock(&pf->mu);
        if (s.err) { pf->free_bufs[pf->nfree++] = b; s.buf = -1; pthread_cond_broadcast(&pf->cv_buf); }
        s.ready = 1;
        pf->slots[i] = s;
        pthread_cond_broadcast(&pf->cv_ready);
        pthread_mutex_unlock(&pf->mu);
        continue;
      }
      /* compressed, a pipe, or larger than a buffer: through a heap copy below */
    }
    uint8_t *text = slurp(path, &n, ROWBUF, &too_big);
    if (!text) s.err = MK_ERR_IO;
    else if (too_big) s.too_big = 1;
    else if (is_fastq(path)) {
      uint32_t stride = 160;
      for (;;) {
        uint64_t nrows = 0;
        size_t used = 0;
        uint64_t nrec = 0;
        int rc = i >= pf->koc_until ? mk_fastq_frame_q(text, n, 1, pf->qmin, pf->TL, 0, pf->bufs[b], stride, ROWBUF / stride, &nrows, &nrec, &used)
                                    : mk_fastq_frame(text, n, 1, pf->bufs[b], stride, ROWBUF / stride, &nrows, &used);
        if (rc == MK_ERR_ARG && stride < 4096) { stride = stride * 2 > 4096 ? 4096 : MK_ROW_PITCH(stride * 2); continue; }
        if (rc != MK_OK) s.err = rc;
        else if (used < n) s.too_big = 1; /* more rows than one buffer holds */
        s.nrows = nrows; s.stride = stride;
        break;
      }
    } else if (!g_host_fasta) {
      if (n == 0) s.err = MK_ERR_STATE;
      else { memcpy(pf->bufs[b], text, n); s.is_text = 1; s.text_bytes = n; } /* (.gz genomes: the text came through zcat) */
    } else {
      mk_fasta_state st;
      const uint32_t stride = MK_ROW_PITCH(512u); /* 528: not a multiple of 128 */
      uint64_t nrows = 0;
      size_t used = 0;
      if (n == 0) s.err = MK_ERR_STATE; /* empty input: the reference's "eof or fread error" (iseq2comem.c:235) */
      else {
        mk_fasta_window_init(&st, pf->TL);
        int rc = mk_fasta_window(&st, text, n, 1, pf->bufs[b], stride, ROWBUF / stride, &nrows, &used);
        if (rc != MK_OK) s.err = rc;
        else if (used < n || st.fresh) s.too_big = 1;
        s.nrows = nrows; s.stride = stride;
      }
    }
    free(text);
    pthread_mutex_lock(&pf->mu);
    if (s.too_big || s.err) { pf->free_bufs[pf->nfree++] = b; s.buf = -1; pthread_cond_broadcast(&pf->cv_buf); }
    s.ready = 1;
    pf->slots[i] = s;
    pthread_cond_broadcast(&pf->cv_ready);
    pthread_mutex_unlock(&pf->mu);
  }
}

/* "0-7", "0,2,5", "0,0": GPU numbers of --devices */
static int parse_devices(const char *s, int *out, int cap) {
  int n = 0;
  while (*s && n < cap) {
    char *end;
    long a = strtol(s, &end, 10);
    if (end == s || a < 0) die("--devices: cannot parse '%s'", s);
    long b = a;
    if (*end == '-') { s = end + 1; b = strtol(s, &end, 10); if (end == s || b < a) die("--devices: bad range"); }
    for (long d = a; d <= b && n < cap; d++) out[n++] = (int)d;
    s = *end == ',' ? end + 1 : end;
    if (*end && *end != ',') die("--devices: cannot parse '%s'", end);
  }
  if (n == 0) die("--devices: empty list");
  return n;
}

static void usage(void) {
  fprintf(stderr,
          "usage: metakssd dist -L <file.shuf> [-A] [-u] [-n minocc] [-Q minqual] [-o outdir] [-p N] [--device D | --devices 0-7] [--engines 1..4] [--no-batch | --batch-text] <fastq|fasta|dir>...\n"
          "       metakssd dist -o <mco dir> <sketch dir>                      (stage II: inverted index)\n"
          "       metakssd dist -L <file.shuf> -r <genomes> -o <db dir>         (stage I + II)\n"
          "       metakssd dist -r <mco dir> -o <outdir> [-M 0|1] [-O 0|1|2] [-N n] [-D d] [--correction 0|1] [--keepskf] [-f skf] <sketch dir>\n"
          "       metakssd set -u|-q|-i <pan dir>|-s <pan dir>|-g <tax.tsv>|-c|-P [-o outdir] [--device D] <sketch dir>\n"
          "       metakssd composite -r <marker db dir> -q <-A sketch dir> [-b] [-o outdir] [--device D]\n"
          "       metakssd composite -d <x.abv>...\n"
          "       metakssd shuffle -k <halfK> -s <halfSubK> -l <level> [--seed N] -o <prefix>\n");
  exit(2);
}

static uint8_t *read_whole(const char *path, size_t *n_out) {
  struct stat st;
  if (stat(path, &st) != 0) return NULL;
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  uint8_t *b = malloc((size_t)st.st_size + 8);
  if (b && fread(b, 1, (size_t)st.st_size, f) != (size_t)st.st_size) { free(b); b = NULL; }
  fclose(f);
  *n_out = (size_t)st.st_size;
  return b;
}

/* `set -P`: print_gnames(), command_set.c:610-631 */
static int set_print_names(const char *in) {
  char path[PATHLEN * 2 + 32];
  snprintf(path, sizeof path, "%s/cofiles.stat", in);
  size_t n = 0;
  uint8_t *st = read_whole(path, &n);
  if (!st || n < 32) die("cannot find cofiles.stat under %s ", in);
  int32_t infile_num;
  memcpy(&infile_num, st + 20, 4);
  if (n < 32 + (size_t)infile_num * (4 + PATHLEN)) die("sketch_union():%s", path);
  for (int i = 0; i < infile_num; i++) {
    uint32_t ct;
    memcpy(&ct, st + 32 + 4 * (size_t)i, 4);
    printf("%d\t%.*s\n", (int)ct, PATHLEN, (const char *)st + 32 + 4 * (size_t)infile_num + (size_t)PATHLEN * i);
  }
  free(st);
  return 0;
}

/* `set -i <pan>` / `set -s <pan>`: sketch_operate(), command_set.c:321-425.  Dictionary of the pan ids and the ordered
 * filter on the device (mk_setop_filter); files, per-file recount and the untouched header fields as in the reference. */
static int set_operate(const char *in, const char *pan, const char *outdir, int intersect, int device) {
  char path[PATHLEN * 2 + 32];
  size_t pn = 0, sn = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", pan);
  uint8_t *pst = read_whole(path, &pn);
  if (!pst || pn < 32) die("cannot find cofiles.stat under %s ", pan);
  snprintf(path, sizeof path, "%s/cofiles.stat", in);
  uint8_t *st = read_whole(path, &sn);
  if (!st || sn < 32) die("cannot find cofiles.stat under %s ", in);
  uint32_t pan_id, in_id;
  int32_t pan_comp, infile_num;
  memcpy(&pan_id, pst, 4); memcpy(&in_id, st, 4);
  memcpy(&pan_comp, pst + 16, 4);
  memcpy(&infile_num, st + 20, 4);
  free(pst);
  if (pan_id != in_id) die("sketcing id not match(%d Vs. %d)", (int)in_id, (int)pan_id);
  if (sn < 32 + 4 * (size_t)infile_num) die("sketch_operate():%s", path);
  uint32_t *ctx_ct = (uint32_t *)(st + 32);
  memset(ctx_ct, 0, 4 * (size_t)infile_num); /* :344-345: recounted below; all_ctx_ct stays as it was */
  mkdir(outdir, 0777);
  mk_setop *so;
  if (mk_setop_create(device, &so) != MK_OK) die("mk_setop_create failed: %s", mk_setop_last_error(NULL));
  uint64_t *post = malloc(8 * ((size_t)infile_num + 1));
  for (int c = 0; c < pan_comp; c++) {
    size_t nb = 0, ib = 0, cb = 0;
    snprintf(path, sizeof path, "%s/pan.%d", pan, c);
    uint8_t *pids = read_whole(path, &nb);
    if (!pids) { snprintf(path, sizeof path, "%s/uniq_pan.%d", pan, c); pids = read_whole(path, &nb); } /* :369-372 */
    if (!pids) die("sketch_operate():%s", path);
    snprintf(path, sizeof path, "%s/combco.index.%d", in, c);
    uint8_t *idx = read_whole(path, &ib);
    if (!idx || ib < 8 * ((size_t)infile_num + 1)) die("sketch_operate():%s", path);
    snprintf(path, sizeof path, "%s/combco.%d", in, c);
    uint8_t *co = read_whole(path, &cb);
    if (!co) die("sketch_operate():%s", path);
    const uint64_t *pos = (const uint64_t *)idx;
    const uint64_t n = pos[infile_num];
    if (n * 4 > cb) die("sketch_operate():%s is shorter than its index says", path);
    const uint32_t *out = NULL;
    uint64_t m = 0;
    if (mk_setop_begin(so, MK_SET_UNION) != MK_OK || mk_setop_add(so, (const uint32_t *)pids, nb / 4) != MK_OK ||
        mk_setop_filter(so, intersect, (const uint32_t *)co, n, pos, (uint32_t)infile_num + 1, &out, &m, post) != MK_OK)
      die("sketch_operate(): %s", mk_setop_last_error(so));
    for (int i = 0; i < infile_num; i++) ctx_ct[i] += (uint32_t)(post[i + 1] - post[i]);
    FILE *f;
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    if (!(f = fopen(path, "wb")) || fwrite(out, 4, m, f) != m) die("sketch_operate():%s", path);
    fclose(f);
    snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
    if (!(f = fopen(path, "wb")) || fwrite(post, 8, (size_t)infile_num + 1, f) != (size_t)infile_num + 1) die("sketch_operate():%s", path);
    fclose(f);
    free(pids); free(idx); free(co);
  }
  free(post);
  mk_setop_destroy(so);
  snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
  FILE *f = fopen(path, "wb");
  if (!f || fwrite(st, 1, sn, f) != sn) die("sketch_operate():%s", path);
  fclose(f);
  free(st);
  return 0;
}

/* `set -g <file.tsv>`: grouping_genomes() (command_set.c:831-974) on top of organize_taxf() (:635-705).
 * Line i of the category file ("<taxid>[TAB<name>]") classifies sketch i.  The taxa are visited in the slot order of the
 * reference's hash of taxids (table of nextPrime(lines / 0.6) slots, double hashing) -- that order is part of the output.
 * Per taxon and component the concatenated id lists of its sketches go through mk_setop_group. */
typedef struct { int taxid; char *name; int *gids; int ng; } taxon_t;

static int next_prime_from(int n) { /* global_basic.c:453-475 */
  for (;; n++) {
    int composite = 0;
    for (int j = 2; (long long)j * j <= n; j++)
      if (n % j == 0) { composite = 1; break; }
    if (!composite) return n;
  }
}

static int set_group(const char *in, const char *taxfile, const char *outdir, int device) {
  size_t tn = 0;
  uint8_t *txt = read_whole(taxfile, &tn);
  if (!txt) die("%s: %s", taxfile, strerror(errno));
  int nlines = 0;
  for (size_t i = 0; i < tn; i++) nlines += txt[i] == '\n';
  const int tsz = next_prime_from((int)((double)nlines / 0.6)); /* LD_FCTR */
  taxon_t *slots = calloc((size_t)tsz, sizeof *slots);
  for (int i = 0; i < tsz; i++) slots[i].taxid = -1;
  int ntax = 0;
  size_t at = 0;
  for (int i = 0; i < nlines; i++) {
    size_t e = at;
    while (txt[e] != '\n') e++;
    if (e - at + 1 >= PATHLEN) die("organize_taxf(): %dth line %.40s is not full read, exceed PATHLEN %d ", i, (char *)txt + at, PATHLEN);
    txt[e] = 0;
    char *save = NULL;
    char *tok = strtok_r((char *)txt + at, "\t", &save);
    at = e + 1;
    if (!tok) die("organize_taxf(): %dth line of %s is empty", i, taxfile);
    const int taxid = atoi(tok);
    const char *name = strtok_r(NULL, "\t", &save);
    if (taxid < 0) die("organize_taxf(): negative taxid %d in %dth line", taxid, i);
    for (int n = 0; n < tsz; n++) {
      const int hv = (taxid % tsz + n * (1 + taxid % (tsz - 1))) % tsz;
      if (hv < 0) die("organize_taxf(): taxid %d overflows the category hash", taxid);
      taxon_t *t = &slots[hv];
      if (t->taxid == -1) {
        t->taxid = taxid; t->name = name ? strdup(name) : NULL;
        t->gids = malloc(sizeof(int)); t->gids[0] = i; t->ng = 1;
        ntax++;
        break;
      }
      if (t->taxid == taxid) {
        if ((t->name == NULL) != (name == NULL) || (name && strcmp(t->name, name) != 0))
          die("organize_taxf() abort!: taxid %d has different taxnames in %dth and %dth lines", taxid, t->gids[0], i);
        t->gids = realloc(t->gids, sizeof(int) * (size_t)(t->ng + 1));
        t->gids[t->ng++] = i;
        break;
      }
    }
  }
  taxon_t *tax = malloc(sizeof *tax * (size_t)(ntax + 1));
  int k = 0;
  for (int i = 0; i < tsz; i++)
    if (slots[i].taxid != -1) tax[k++] = slots[i];
  free(slots);

  char path[PATHLEN * 2 + 32];
  size_t sn = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", in);
  uint8_t *st = read_whole(path, &sn);
  if (!st || sn < 32) die("cannot find cofiles.stat under %s ", in);
  int32_t comp_num, infile_num;
  memcpy(&comp_num, st + 16, 4);
  memcpy(&infile_num, st + 20, 4);
  if (infile_num != nlines)
    die("grouping_genomes():%s's genome number %d not matches %s's genome number %d", path, infile_num, taxfile, nlines);
  mkdir(outdir, 0777);
  mk_setop *so;
  if (mk_setop_create(device, &so) != MK_OK) die("mk_setop_create failed: %s", mk_setop_last_error(NULL));
  uint32_t *ctx_ct = calloc((size_t)ntax + 1, 4);
  uint64_t *outidx = malloc(8 * ((size_t)ntax + 1));
  uint64_t all_ctx_ct = 0;
  int outfn = 0;
  for (int c = 0; c < comp_num; c++) {
    size_t cb = 0, ib = 0;
    snprintf(path, sizeof path, "%s/combco.%d", in, c);
    uint8_t *co = read_whole(path, &cb);
    if (!co) die("grouping_genomes():%s", path);
    snprintf(path, sizeof path, "%s/combco.index.%d", in, c);
    uint8_t *idx = read_whole(path, &ib);
    if (!idx || ib < 8 * ((size_t)infile_num + 1)) die("grouping_genomes():%s", path);
    const uint32_t *ids = (const uint32_t *)co;
    const uint64_t *pos = (const uint64_t *)idx;
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    FILE *f = fopen(path, "wb");
    if (!f) die("grouping_genomes():%s", path);
    uint32_t *cat = NULL;
    uint64_t cat_cap = 0, offset = 0;
    outfn = 0;
    outidx[0] = 0;
    for (int t = 0; t < ntax; t++) {
      if (tax[t].taxid == 0) continue; /* taxid 0 = leave these sketches out (:866) */
      uint64_t total = 0;
      for (int g = 0; g < tax[t].ng; g++) total += pos[tax[t].gids[g] + 1] - pos[tax[t].gids[g]];
      /* a taxon without a k-mer in this component: the reference evaluates LOG2(0 * 1.5) there (command_set.c:878: __builtin_clzll(0),
       * undefined; every build seen so far ends up with primer[0] slots and nothing in them) and writes an EMPTY block: so do we */
      if (total == 0) { outidx[++outfn] = offset; continue; }
      if (total > cat_cap) {
        if (cat) mk_host_free(cat);
        cat_cap = total + total / 4 + 1024;
        if (mk_host_alloc((void **)&cat, cat_cap * 4) != MK_OK) die("out of memory");
      }
      uint64_t w = 0;
      for (int g = 0; g < tax[t].ng; g++) {
        const uint64_t a = pos[tax[t].gids[g]], b = pos[tax[t].gids[g] + 1];
        memcpy(cat + w, ids + a, (b - a) * 4);
        w += b - a;
      }
      const uint32_t *out = NULL;
      uint64_t m = 0;
      if (mk_setop_group(so, cat, total, mk_setop_group_table_size(total), &out, &m) != MK_OK)
        die("grouping_genomes(): %s", mk_setop_last_error(so));
      if (fwrite(out, 4, m, f) != m) die("grouping_genomes():%s", path);
      offset += m; all_ctx_ct += m; ctx_ct[outfn] += (uint32_t)m;
      outidx[++outfn] = offset;
      printf("%d/%d species pangenome grouped\r", t, ntax);
    }
    printf("\n");
    fclose(f);
    if (cat) mk_host_free(cat);
    snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
    if (!(f = fopen(path, "wb")) || fwrite(outidx, 8, (size_t)outfn + 1, f) != (size_t)outfn + 1) die("grouping_genomes():%s", path);
    fclose(f);
    free(co); free(idx);
  }
  mk_setop_destroy(so);
  /* cofiles.stat: the input's header with the new sketch count, koc = 0 and the new total (:929-966) */
  int32_t v = outfn;
  memcpy(st + 20, &v, 4);
  st[4] = 0;
  memcpy(st + 24, &all_ctx_ct, 8);
  snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
  FILE *f = fopen(path, "wb");
  if (!f) die("grouping_genomes():%s", path);
  fwrite(st, 1, 32, f);
  fwrite(ctx_ct, 4, (size_t)outfn, f);
  for (int t = 0; t < ntax; t++) {
    if (tax[t].taxid == 0) continue;
    char name[PATHLEN];
    memset(name, 0, sizeof name);
    if (tax[t].name) snprintf(name, sizeof name, "%d_%s", tax[t].taxid, tax[t].name);
    else snprintf(name, sizeof name, "%d", tax[t].taxid);
    fwrite(name, 1, PATHLEN, f);
  }
  fclose(f);
  for (int t = 0; t < ntax; t++) { free(tax[t].name); free(tax[t].gids); }
  free(tax); free(st); free(ctx_ct); free(outidx); free(txt);
  return 0;
}

/* `set -c <pan dir>...`: combin_pans(), command_set.c:515-608 -- the pan.N (or uniq_pan.N) files of several pan directories
 * become the blocks of one combined sketch directory.  File shuffling only, no device work.  The reference writes 256 bytes
 * starting at each argument string as the name record (whatever follows the NUL in argv); here the rest is zero. */
static int set_combine(int ndirs, char **dirs, const char *outdir) {
  char path[PATHLEN * 2 + 32];
  size_t n0 = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", dirs[0]);
  uint8_t *hdr = read_whole(path, &n0);
  if (!hdr || n0 < 32) die("combin_pans():%s", path);
  uint32_t id0;
  int32_t comp_num;
  memcpy(&id0, hdr, 4);
  memcpy(&comp_num, hdr + 16, 4);
  mkdir(outdir, 0777);
  uint32_t *ctx_ct = calloc((size_t)ndirs, 4);
  uint64_t all_ctx_ct = 0;
  FILE **co = malloc(sizeof(FILE *) * (size_t)comp_num), **ix = malloc(sizeof(FILE *) * (size_t)comp_num);
  uint64_t *off = calloc((size_t)comp_num, 8);
  for (int c = 0; c < comp_num; c++) {
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    if (!(co[c] = fopen(path, "wb"))) die("%s", path);
    snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
    if (!(ix[c] = fopen(path, "wb"))) die("%s", path);
    fwrite(&off[c], 8, 1, ix[c]);
  }
  for (int i = 0; i < ndirs; i++) {
    size_t ni = 0;
    snprintf(path, sizeof path, "%s/cofiles.stat", dirs[i]);
    uint8_t *h = read_whole(path, &ni);
    if (!h || ni < 32) die("combin_pans():%s", path);
    uint32_t idi;
    int32_t ci;
    memcpy(&idi, h, 4);
    memcpy(&ci, h + 16, 4);
    free(h);
    if (idi != id0) die("combin_pans(): %dth shuf_id: %u not match 0th shuf_id: %u", i, idi, id0);
    if (ci != comp_num) die("combin_pans(): %dth comp_num: %u not match 0th comp_num: %u", i, (unsigned)ci, (unsigned)comp_num);
    for (int c = 0; c < comp_num; c++) {
      size_t nb = 0;
      snprintf(path, sizeof path, "%s/pan.%d", dirs[i], c);
      uint8_t *ids = read_whole(path, &nb);
      if (!ids) { snprintf(path, sizeof path, "%s/uniq_pan.%d", dirs[i], c); ids = read_whole(path, &nb); }
      if (!ids) die("%s", path);
      fwrite(ids, 1, nb, co[c]);
      off[c] += nb / 4;
      fwrite(&off[c], 8, 1, ix[c]);
      ctx_ct[i] += (uint32_t)(nb / 4);
      free(ids);
    }
    all_ctx_ct += ctx_ct[i];
  }
  for (int c = 0; c < comp_num; c++) { fclose(co[c]); fclose(ix[c]); }
  int32_t v = ndirs;
  memcpy(hdr + 20, &v, 4);
  memcpy(hdr + 24, &all_ctx_ct, 8);
  snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
  FILE *f = fopen(path, "wb");
  if (!f) die("%s", path);
  fwrite(hdr, 1, 32, f);
  fwrite(ctx_ct, 4, (size_t)ndirs, f);
  for (int i = 0; i < ndirs; i++) {
    char name[PATHLEN];
    memset(name, 0, sizeof name);
    snprintf(name, sizeof name, "%s", dirs[i]);
    fwrite(name, 1, PATHLEN, f);
  }
  fclose(f);
  free(hdr); free(ctx_ct); free(co); free(ix); free(off);
  return 0;
}

/* ---- `metakssd set`: -u / -q (sketch_union / uniq_sketch_union, command_set.c:241-319,427-512), -i / -s <pan>
 * (sketch_operate, :321-425), -P (print_gnames, :610-631) ----
 * The dictionary work runs on the device (mk_setop_*); the directory handling follows the reference: the 32-byte
 * cofiles.stat header is copied as it is, one pan.N / uniq_pan.N per component, and a sketch directory holding a
 * single sketch is offered for renaming in place (:254-267). */
static int cmd_set(int argc, char **argv) {
  int op = -1, device = 0, print = 0; /* 0 subtract, 1 intersect, 2 union, 3 uniq_union (command_set.c:55) */
  const char *outdir = "./", *in = NULL, *panpath = NULL, *taxfile = NULL;
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-u")) { if (op != -1) printf("set operation is already set, -u is ignored.\n"); else op = 2; }
    else if (!strcmp(argv[i], "-q")) { if (op != -1) printf("set operation is already set, -q is ignored.\n"); else op = 3; }
    else if (!strcmp(argv[i], "-s") && i + 1 < argc) { if (op != -1) printf("set operation is already set, -s is ignored.\n"); else { op = 0; panpath = argv[i + 1]; } i++; }
    else if (!strcmp(argv[i], "-i") && i + 1 < argc) { if (op != -1) printf("set operation is already set, -i is ignored.\n"); else { op = 1; panpath = argv[i + 1]; } i++; }
    else if (!strcmp(argv[i], "-c")) { if (op != -1) printf("set operation is already set, -c is ignored.\n"); else op = 4; }
    else if (!strcmp(argv[i], "-P")) print = 1;
    else if (!strcmp(argv[i], "-g") && i + 1 < argc) taxfile = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) outdir = argv[++i];
    else if (!strcmp(argv[i], "-p") && i + 1 < argc) ++i; /* threads: no meaning here */
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (argv[i][0] == '-' && argv[i][1]) die("set option %s is not part of this build (-u -q -i -s -g -c -P are)", argv[i]);
    else if (!in) in = argv[i];
  }
  if (!in) usage();
  if (op == 4) { /* every non-option argument is a pan directory */
    char *dirs[4096];
    int nd = 0;
    for (int i = 0; i < argc && nd < 4096; i++) {
      if (!strcmp(argv[i], "-o") || !strcmp(argv[i], "-p") || !strcmp(argv[i], "--device")) { i++; continue; }
      if (argv[i][0] != '-') dirs[nd++] = argv[i];
    }
    return set_combine(nd, dirs, outdir);
  }
  if (op == 0 || op == 1) return set_operate(in, panpath, outdir, op == 1, device);
  if (op == -1) {
    if (print) return set_print_names(in);
    if (taxfile) return set_group(in, taxfile, outdir, device); /* looked at only without -u/-q/-i/-s (command_set.c:227-231) */
    printf("set operation use : -u, -q, -i or -s\n");
    return 255;
  }
  const char *prefix = op == 2 ? "pan" : "uniq_pan";
  const char *fn = op == 2 ? "sketch_union()" : "uniq_sketch_union()";
  char path[PATHLEN * 2 + 32];
  snprintf(path, sizeof path, "%s/cofiles.stat", in);
  FILE *f = fopen(path, "rb");
  if (!f) die("cannot find cofiles.stat under %s ", in);
  unsigned char hdr[32];
  if (fread(hdr, 1, 32, f) != 32) die("%s:%s", fn, path);
  fclose(f);
  int32_t comp_num, infile_num;
  memcpy(&comp_num, hdr + 16, 4);
  memcpy(&infile_num, hdr + 20, 4);
  if (infile_num == 1) {
    char reply = 0;
    printf("only 1 sketch, use %s as pan-sketch?(Y/N)\n", in);
    if (scanf(" %c", &reply) == 1 && (reply == 'Y' || reply == 'y')) {
      for (int c = 0; c < comp_num; c++) {
        char a[PATHLEN * 2 + 32], b[PATHLEN * 2 + 32];
        snprintf(a, sizeof a, "%s/combco.%d", in, c);
        snprintf(b, sizeof b, "%s/%s.%d", in, prefix, c);
        if (rename(a, b) != 0) die("%s: %s", fn, strerror(errno));
      }
      printf("the union directory: %s created successfully\n", in);
      return 0;
    }
  }
  mkdir(outdir, 0777);
  snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
  if (!(f = fopen(path, "wb"))) die("%s:%s", fn, path);
  fwrite(hdr, 1, 32, f);
  fclose(f);
  mk_setop *so;
  if (mk_setop_create(device, &so) != MK_OK) die("mk_setop_create failed: %s", mk_setop_last_error(NULL));
  for (int c = 0; c < comp_num; c++) {
    snprintf(path, sizeof path, "%s/combco.%d", in, c);
    struct stat st;
    if (stat(path, &st) != 0) die("%s:%s", fn, path);
    const uint64_t n = (uint64_t)st.st_size / 4;
    uint32_t *ids = NULL;
    if (mk_host_alloc((void **)&ids, n ? n * 4 : 4) != MK_OK) die("out of memory");
    if (!(f = fopen(path, "rb")) || fread(ids, 4, n, f) != n) die("%s:%s", fn, path);
    fclose(f);
    const uint32_t *out = NULL;
    uint64_t m = 0;
    if (mk_setop_begin(so, op == 2 ? MK_SET_UNION : MK_SET_UNIQ_UNION) != MK_OK || mk_setop_add(so, ids, n) != MK_OK ||
        mk_setop_finish(so, &out, &m) != MK_OK)
      die("%s: %s", fn, mk_setop_last_error(so));
    mk_host_free(ids);
    snprintf(path, sizeof path, "%s/%s.%d", outdir, prefix, c);
    if (!(f = fopen(path, "wb")) || fwrite(out, 4, m, f) != m) die("%s:%s", fn, path);
    fclose(f);
  }
  mk_setop_destroy(so);
  return 0;
}

/* ---- `metakssd composite -r <ref> -q <qry> [-b] [-o outdir]`: get_species_abundance(), command_composite.c:446-649 ----
 * The join (query k-mer dictionary, lookup of every reference k-mer, :525-553) runs on the device, one mk_setop_join per
 * query and component; the order statistics over each reference's handful of counts and the report stay on the host,
 * in the reference's float arithmetic. */
static int cmp_int_asc(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

/* `composite -d <x.abv>...`: read_abv(), command_composite.c:186-210 -- prints the (reference index, percentage) pairs of
 * the vector files `composite -b` writes.  Host only. */
static int composite_read_abv(int nfiles, char **files) {
  for (int i = 0; i < nfiles; i++) {
    const char *ext = strrchr(files[i], '.');
    if (!ext || strcmp(ext + 1, "abv") != 0) {
      printf("%dth argument %s is not a .abv file, skipped\n", i, files[i]);
      continue;
    }
    size_t n = 0;
    uint8_t *b = read_whole(files[i], &n);
    if (!b) die("read_abv():%s", files[i]);
    for (size_t l = 0; l + 8 <= n || l < n; l += 8) { /* the reference loops while l < st_size */
      int32_t idx = 0;
      float pct = 0;
      if (l + 8 <= n) { memcpy(&idx, b + l, 4); memcpy(&pct, b + l + 4, 4); }
      printf("%d\t%f\n", idx, pct);
    }
    free(b);
  }
  return 1;
}

static int cmd_composite(int argc, char **argv) {
  const char *refdir = NULL, *qrydir = NULL, *outdir = "./";
  int binvec = 0, device = 0;
  for (int i = 0; i < argc; i++)
    if (!strcmp(argv[i], "-d")) { /* cmd_composite(): -d is looked at only without -r (command_composite.c:179-182) */
      int has_r = 0;
      for (int j = 0; j < argc; j++) has_r |= !strcmp(argv[j], "-r");
      if (!has_r) {
        char *files[4096];
        int nf = 0;
        for (int j = 0; j < argc && nf < 4096; j++)
          if (argv[j][0] != '-') files[nf++] = argv[j];
        if (nf < 1) { printf("\vUsage: kssd composite -d <query.abv>\n\v"); return 0; }
        return composite_read_abv(nf, files) == 1 ? 0 : 1;
      }
    }
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-r") && i + 1 < argc) refdir = argv[++i];
    else if (!strcmp(argv[i], "-q") && i + 1 < argc) qrydir = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) outdir = argv[++i];
    else if (!strcmp(argv[i], "-p") && i + 1 < argc) ++i;
    else if (!strcmp(argv[i], "-b")) binvec = 1;
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else die("composite option %s is not part of this build (-r -q -b -o and -d are)", argv[i]);
  }
  if (!refdir || !qrydir || !strcmp(refdir, qrydir)) die("get_species_abundance(): refdir or qrydir is not initialized");
  char path[PATHLEN * 3 + 64];
  size_t rsn = 0, qsn = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", refdir);
  uint8_t *rst = read_whole(path, &rsn);
  if (!rst || rsn < 32) die("cannot find cofiles.stat under %s ", refdir);
  snprintf(path, sizeof path, "%s/cofiles.stat", qrydir);
  uint8_t *qst = read_whole(path, &qsn);
  if (!qst || qsn < 32) die("cannot find cofiles.stat under %s ", qrydir);
  uint32_t ref_id, qry_id;
  int32_t ref_n, qry_n, comp_num;
  memcpy(&ref_id, rst, 4); memcpy(&qry_id, qst, 4);
  memcpy(&comp_num, rst + 16, 4);
  memcpy(&ref_n, rst + 20, 4); memcpy(&qry_n, qst + 20, 4);
  if (!qst[4]) die("get_species_abundance(): query has not abundance");
  if (qry_id != ref_id) printf("get_species_abundance(): qry shuf_id %u not match ref shuf_id: %u\n", qry_id, ref_id);
  if (rsn < 32 + (size_t)ref_n * (4 + PATHLEN) || qsn < 32 + (size_t)qry_n * (4 + PATHLEN)) die("get_species_abundance(): truncated cofiles.stat");
  const char *refname = (const char *)rst + 32 + 4 * (size_t)ref_n, *qryname = (const char *)qst + 32 + 4 * (size_t)qry_n;

  mk_setop *so;
  if (mk_setop_create(device, &so) != MK_OK) die("mk_setop_create failed: %s", mk_setop_last_error(NULL));
  int **vals = calloc((size_t)ref_n, sizeof(int *)); /* per reference sketch: the query's counts of the shared k-mers */
  int *nval = calloc((size_t)ref_n, sizeof(int)), *cap = calloc((size_t)ref_n, sizeof(int));
  uint64_t *seg = malloc(8 * ((size_t)ref_n + 1));
  int *order = malloc(sizeof(int) * (size_t)ref_n), *tmp = malloc(sizeof(int) * (size_t)ref_n);
  struct { int ref_idx; float pct; } *vec = malloc(8 * ((size_t)ref_n + 1));

  for (int q = 0; q < qry_n; q++) {
    for (int r = 0; r < ref_n; r++) nval[r] = 0;
    for (int c = 0; c < comp_num; c++) {
      size_t n1, n2, n3, n4, n5;
      snprintf(path, sizeof path, "%s/combco.%d", refdir, c);
      uint8_t *rco = read_whole(path, &n1);
      if (!rco) die("get_species_abundance():%s", path);
      snprintf(path, sizeof path, "%s/combco.index.%d", refdir, c);
      uint8_t *ridx = read_whole(path, &n2);
      if (!ridx || n2 < 8 * ((size_t)ref_n + 1)) die("get_species_abundance():%s", path);
      snprintf(path, sizeof path, "%s/combco.%d", qrydir, c);
      uint8_t *qco = read_whole(path, &n3);
      if (!qco) die("get_species_abundance():%s", path);
      snprintf(path, sizeof path, "%s/combco.index.%d", qrydir, c);
      uint8_t *qidx = read_whole(path, &n4);
      if (!qidx || n4 < 8 * ((size_t)qry_n + 1)) die("get_species_abundance():%s", path);
      snprintf(path, sizeof path, "%s/combco.%d.a", qrydir, c);
      uint8_t *qab = read_whole(path, &n5);
      if (!qab) die("get_species_abundance():%s", path);
      const uint64_t *rpos = (const uint64_t *)ridx, *qpos = (const uint64_t *)qidx;
      if (rpos[ref_n] * 4 > n1 || qpos[qry_n] * 4 > n3 || qpos[qry_n] * 2 > n5) die("get_species_abundance(): component %d is shorter than its index says", c);
      const uint32_t *counts = NULL;
      uint64_t m = 0;
      if (mk_setop_join(so, (const uint32_t *)qco + qpos[q], (const uint16_t *)qab + qpos[q], qpos[q + 1] - qpos[q], (const uint32_t *)rco,
                        rpos[ref_n], rpos, (uint32_t)ref_n + 1, &counts, &m, seg) != MK_OK)
        die("get_species_abundance(): %s", mk_setop_last_error(so));
      for (int r = 0; r < ref_n; r++) {
        const int k = (int)(seg[r + 1] - seg[r]);
        if (nval[r] + k > cap[r]) {
          cap[r] = (nval[r] + k) * 2 + 8;
          vals[r] = realloc(vals[r], sizeof(int) * (size_t)cap[r]);
        }
        for (int j = 0; j < k; j++) vals[r][nval[r] + j] = (int)counts[seg[r] + (uint64_t)j];
        nval[r] += k;
      }
      free(rco); free(ridx); free(qco); free(qidx); free(qab);
    }
    /* references by decreasing number of shared k-mers; ties keep their index order (what glibc's merge-sort qsort gives
     * the reference, :568-570): bottom-up merge sort */
    for (int i = 0; i < ref_n; i++) order[i] = i;
    for (int w = 1; w < ref_n; w *= 2) {
      for (int lo = 0; lo < ref_n; lo += 2 * w) {
        const int mid = lo + w < ref_n ? lo + w : ref_n, hi = lo + 2 * w < ref_n ? lo + 2 * w : ref_n;
        int a = lo, b = mid, k = lo;
        while (a < mid && b < hi) tmp[k++] = nval[order[b]] > nval[order[a]] ? order[b++] : order[a++];
        while (a < mid) tmp[k++] = order[a++];
        while (b < hi) tmp[k++] = order[b++];
      }
      memcpy(order, tmp, sizeof(int) * (size_t)ref_n);
    }
    FILE *vf = NULL;
    if (binvec) { /* :573-581 */
      char dir[PATHLEN * 2 + 32], qn[PATHLEN + 1];
      if (strlen(outdir) < 3) snprintf(dir, sizeof dir, "%s/abundance_Vec", refdir);
      else snprintf(dir, sizeof dir, "%s", outdir);
      mkdir(dir, 0777);
      snprintf(qn, sizeof qn, "%.*s", PATHLEN, qryname + (size_t)PATHLEN * q);
      const char *base = strrchr(qn, '/');
      snprintf(path, sizeof path, "%s/%s.abv", dir, base ? base + 1 : qn);
      if (!(vf = fopen(path, "wb"))) die("get_species_abundance():%s", path);
    }
    int num_pass = 0;
    float vecsum = 0;
    for (int i = 0; i < ref_n; i++) {
      const int r = order[i], kmer_num = nval[r];
      if (kmer_num < 6) break; /* MIN_KM_S */
      int *v = vals[r] - 1;    /* 1-based like the reference's ref_abund[r][1..kmer_num] */
      qsort(vals[r], (size_t)kmer_num, sizeof(int), cmp_int_asc);
      int sum = 0;
      for (int n = 1; n <= kmer_num; n++) sum += v[n];
      const int median_idx = kmer_num / 2, pct_idx = kmer_num * 0.98; /* ST_PCTL */
      int lastsum = 0, lastn = 0;
      for (int n = pct_idx; n <= kmer_num * 0.99; n++) { lastsum += v[n]; lastn++; } /* ED_PCTL */
      if (binvec) {
        if (v[median_idx] > 1 && kmer_num > 7) {
          vec[num_pass].ref_idx = r;
          vec[num_pass].pct = (float)lastsum / lastn;
          vecsum += vec[num_pass].pct;
          num_pass++;
        }
      } else {
        printf("%.*s\t%.*s\t%d\t%f\t%f\t%d\t%d\n", PATHLEN, qryname + (size_t)PATHLEN * q, PATHLEN, refname + (size_t)PATHLEN * r, kmer_num,
               (float)sum / kmer_num, (float)lastsum / lastn, v[median_idx], v[kmer_num]);
      }
    }
    if (vf) {
      for (int i = 0; i < num_pass; i++) vec[i].pct = (vec[i].pct - 1) * 100 / (vecsum - num_pass);
      fwrite(vec, 8, (size_t)num_pass, vf);
      fclose(vf);
    }
  }
  mk_setop_destroy(so);
  for (int r = 0; r < ref_n; r++) free(vals[r]);
  free(vals); free(nval); free(cap); free(seg); free(order); free(tmp); free(vec); free(rst); free(qst);
  return 0;
}

static int cmd_shuffle(int argc, char **argv) {
  int k = 8, s = 5, l = 2;
  unsigned long long seed = 1;
  const char *out = "default";
  for (int i = 0; i < argc; i++) {
    if (!strcmp(argv[i], "-k") && i + 1 < argc) k = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-s") && i + 1 < argc) s = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-l") && i + 1 < argc) l = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], NULL, 10);
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else usage();
  }
  mk_shuf sh;
  if (mk_shuf_generate(k, s, l, seed, &sh) != MK_OK) die("shuffle: invalid -k/-s/-l (need subk <= k, subk < 8: command_shuffle.c:176-181)");
  char path[PATHLEN * 2];
  snprintf(path, sizeof path, "%s.shuf", out);
  if (mk_shuf_write(&sh, path) != MK_OK) die("write_dim_shuffle_file(): open file %s failed", path);
  printf("kssd shuffle: shuf_id=%d, k = %d, halfCtxLen = %d, level= %d\n", sh.id, sh.k, sh.subk, sh.drlevel);
  mk_shuf_free(&sh);
  return 0;
}


/* ---- stage II and `dist -r` ---------------------------------------------------------------------- */
static int dir_has(const char *dir, const char *name) { /* test_get_fullpath(), command_dist.c:318-338 */
  char path[PATHLEN * 2 + 32];
  struct stat st;
  if (stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return 0;
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE *f = fopen(path, "rb");
  if (!f) return 0;
  fclose(f);
  return 1;
}

/* run_stageII() + combco2mco() (command_dist.c:504-552, co2mco.c:12-87): mcofiles.stat, mco.N, mco.index.N */
static int run_stage2(const char *codir, const char *mcodir, int device, int quiet) {
  char path[PATHLEN * 2 + 32];
  size_t sn = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", codir);
  uint8_t *st = read_whole(path, &sn);
  if (!st || sn < 32) die("run_stageII(():%s", path);
  int32_t comp_num, cofnum;
  memcpy(&comp_num, st + 16, 4); memcpy(&cofnum, st + 20, 4);
  if (sn < 32 + (size_t)cofnum * (4 + PATHLEN)) die("run_stageII(():%s", path);
  if (mkdir(mcodir, 0700)) {
    if (errno == EEXIST) printf("Warning: write mco file to an exists outdir:%s\n", mcodir);
    else die("run_stageII(): mkdir %s error", mcodir);
  }
  snprintf(path, sizeof path, "%s/mcofiles.stat", mcodir);
  FILE *f = fopen(path, "wb");
  if (!f) die("run_stageII(():%s", path);
  fwrite(st, 4, 1, f);                                   /* mco_dstat_t: shuf_id, */
  fwrite(st + 8, 4, 4, f);                               /* kmerlen, dim_rd_len, comp_num, infile_num (command_dist.h:66-75) */
  fwrite(st + 32, 1, (size_t)cofnum * (4 + PATHLEN), f); /* per-sketch k-mer counts, names */
  if (fclose(f)) die("run_stageII(():%s", path);
  mk_mco *m;
  if (mk_mco_create(device, &m) != MK_OK) die("mk_mco_create failed: %s", mk_mco_last_error(NULL));
  const uint64_t slab = 1ull << 27;
  uint64_t *rows = NULL; /* pinned: the slabs come back at PCIe speed instead of through a pageable bounce buffer */
  if (mk_host_alloc((void **)&rows, slab * 8) != MK_OK) die("out of memory");
  for (int c = 0; c < comp_num; c++) {
    size_t nb = 0, ib = 0;
    snprintf(path, sizeof path, "%s/combco.index.%d", codir, c);
    uint8_t *idx = read_whole(path, &ib);
    if (!idx || ib < 8 * ((size_t)cofnum + 1)) die("%s", path);
    snprintf(path, sizeof path, "%s/combco.%d", codir, c);
    uint8_t *ids = read_whole(path, &nb);
    if (!ids || nb / 4 < ((const uint64_t *)idx)[cofnum]) die("%s", path);
    const uint32_t *gids, *row_ids;
    const uint64_t *row_ends;
    uint64_t n, nrows;
    int rc = mk_mco_build(m, (const uint32_t *)ids, (const uint64_t *)idx, (uint32_t)cofnum, &gids, &n, &row_ids, &row_ends, &nrows);
    if (rc != MK_OK) die("mk_mco_build failed (%d): %s", rc, mk_mco_last_error(m));
    free(ids); free(idx);
    snprintf(path, sizeof path, "%s/mco.index.%d", mcodir, c);
    f = fopen(path, "wb");
    if (!f) die("%s", path);
    const uint64_t comp_rows = 1ull << (4 * g_component_sz); /* 1LLU << 4*COMPONENT_SZ rows, co2mco.c:19, 63-66 */
    for (uint64_t r0 = 0; r0 < comp_rows; r0 += slab) {
      const uint64_t nr = comp_rows - r0 < slab ? comp_rows - r0 : slab;
      rc = mk_mco_index_rows(m, r0, nr, rows);
      if (rc != MK_OK) die("mk_mco_index_rows failed (%d): %s", rc, mk_mco_last_error(m));
      if (fwrite(rows, 8, nr, f) != nr) die("%s: write failed", path);
    }
    if (fclose(f)) die("%s: write failed", path);
    snprintf(path, sizeof path, "%s/mco.%d", mcodir, c);
    f = fopen(path, "wb");
    if (!f) die("combco2mco()::%s", path);
    if (n && fwrite(gids, 4, n, f) != n) die("%s: write failed", path);
    if (fclose(f)) die("%s: write failed", path);
    if (!quiet) printf("component %d: %llu ids in %llu rows\n", c, (unsigned long long)n, (unsigned long long)nrows);
  }
  mk_host_free(rows); free(st);
  mk_mco_destroy(m);
  return 0;
}

/* combine_queries() (command_dist.c:1718-1924): `dist -o <out> <sketch dir> <sketch dir>...` strings several sketch directories
 * (batches of queries) together: combco.N appended, combco.index.N continued with the running offset, the cofiles.stat
 * header of the first directory with the summed sketch and k-mer counts, then everybody's count lists and names.  Host only.
 * As there: -A sketches are refused, a directory without cofiles.stat, with another shuf_id or with abundances is skipped
 * with a message. */
static int run_combine(int ndirs, char **dirs, const char *outdir) {
  char path[PATHLEN * 2 + 32];
  size_t n0 = 0;
  mkdir(outdir, 0700);
  snprintf(path, sizeof path, "%s/cofiles.stat", dirs[0]);
  uint8_t *h0 = read_whole(path, &n0);
  if (!h0 || n0 < 32) die("combine_queries():%s", path);
  if (h0[4]) die("combine_queries(): abundance model not supported yet");
  uint32_t id0;
  int32_t comp_num, infile_num;
  uint64_t all_ctx_ct;
  memcpy(&id0, h0, 4); memcpy(&comp_num, h0 + 16, 4); memcpy(&infile_num, h0 + 20, 4); memcpy(&all_ctx_ct, h0 + 24, 8);
  if (n0 < 32 + (size_t)infile_num * (4 + PATHLEN)) die("combine_queries():%s", path);
  /* count lists and name records of every accepted directory, in order */
  size_t ct_bytes = 0, nm_bytes = 0;
  uint8_t *cts = NULL, *nms = NULL;
#define APPEND(buf, len, src, n) do { buf = realloc(buf, len + (n)); if (!buf) die("out of memory"); memcpy(buf + len, src, n); len += (n); } while (0)
  APPEND(cts, ct_bytes, h0 + 32, 4 * (size_t)infile_num);
  APPEND(nms, nm_bytes, h0 + 32 + 4 * (size_t)infile_num, (size_t)PATHLEN * infile_num);
  FILE **co = malloc(sizeof(FILE *) * (size_t)comp_num), **ix = malloc(sizeof(FILE *) * (size_t)comp_num);
  uint64_t *offset = calloc((size_t)comp_num, 8);
  for (int c = 0; c < comp_num; c++) {
    size_t nb = 0, ib = 0;
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    if (!(co[c] = fopen(path, "wb"))) die("%s", path);
    snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
    if (!(ix[c] = fopen(path, "wb"))) die("%s", path);
    snprintf(path, sizeof path, "%s/combco.%d", dirs[0], c);
    uint8_t *ids = read_whole(path, &nb);
    if (!ids) die("%s", path);
    if (nb && fwrite(ids, 1, nb, co[c]) != nb) die("%s: write failed", path);
    free(ids);
    snprintf(path, sizeof path, "%s/combco.index.%d", dirs[0], c);
    uint8_t *idx = read_whole(path, &ib);
    if (!idx || ib < 8) die("%s", path);
    if (fwrite(idx, 1, ib, ix[c]) != ib) die("%s: write failed", path);
    memcpy(&offset[c], idx + ib - 8, 8);
    free(idx);
  }
  for (int i = 1; i < ndirs; i++) {
    size_t ni = 0;
    snprintf(path, sizeof path, "%s/cofiles.stat", dirs[i]);
    uint8_t *h = read_whole(path, &ni);
    if (!h || ni < 32) { printf("%dth query %s is not a valid query: no %s file\n", i, dirs[i], "cofiles.stat"); free(h); continue; }
    uint32_t idi;
    int32_t fi;
    uint64_t cti;
    memcpy(&idi, h, 4); memcpy(&fi, h + 20, 4); memcpy(&cti, h + 24, 8);
    if (idi != id0) { printf("combine_queries(): %dth shuf_id: %u not match 0th shuf_id: %u\n", i, idi, id0); free(h); continue; }
    if (h[4]) { printf("combine_queries(): %dth query abundance model not supported yet \n", i); free(h); continue; }
    if (ni < 32 + (size_t)fi * (4 + PATHLEN)) die("combine_queries():%s", path);
    all_ctx_ct += cti;
    infile_num += fi;
    APPEND(cts, ct_bytes, h + 32, 4 * (size_t)fi);
    APPEND(nms, nm_bytes, h + 32 + 4 * (size_t)fi, (size_t)PATHLEN * fi);
    free(h);
    for (int c = 0; c < comp_num; c++) {
      size_t nb = 0, ib = 0;
      snprintf(path, sizeof path, "%s/combco.%d", dirs[i], c);
      uint8_t *ids = read_whole(path, &nb);
      if (!ids) die("%s", path);
      if (nb && fwrite(ids, 1, nb, co[c]) != nb) die("%s: write failed", path);
      free(ids);
      snprintf(path, sizeof path, "%s/combco.index.%d", dirs[i], c);
      uint64_t *idx = (uint64_t *)read_whole(path, &ib);
      if (!idx || ib < 8) die("%s", path);
      const size_t ne = ib / 8;
      for (size_t k = 1; k < ne; k++) idx[k] += offset[c];
      if (ne > 1 && fwrite(idx + 1, 8, ne - 1, ix[c]) != ne - 1) die("%s: write failed", path);
      offset[c] = idx[ne - 1];
      free(idx);
    }
  }
#undef APPEND
  for (int c = 0; c < comp_num; c++)
    if (fclose(co[c]) || fclose(ix[c])) die("combine_queries(): write failed");
  memcpy(h0 + 20, &infile_num, 4);
  memcpy(h0 + 24, &all_ctx_ct, 8);
  snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
  FILE *f = fopen(path, "wb");
  if (!f || fwrite(h0, 1, 32, f) != 32 || fwrite(cts, 1, ct_bytes, f) != ct_bytes || fwrite(nms, 1, nm_bytes, f) != nm_bytes || fclose(f))
    die("%s: write failed", path);
  free(h0); free(cts); free(nms); free(co); free(ix); free(offset);
  return 0;
}

static void *map_whole(const char *path, size_t *n) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return NULL;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return NULL; }
  *n = (size_t)st.st_size;
  void *p = st.st_size ? mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : NULL;
  close(fd);
  return p == MAP_FAILED ? NULL : p;
}

typedef struct {
  const uint64_t *index;
  const uint32_t *ids;
  uint64_t *es, *ee;
  uint64_t lo, hi;
} extent_job;

static void *extent_worker(void *arg) { /* command_dist.c:1040-1041: the row of k-mer id `ind` in the mmap'ed index */
  extent_job *j = arg;
  for (uint64_t i = j->lo; i < j->hi; i++) {
    const uint32_t ind = j->ids[i];
    j->es[i] = ind > 0 ? j->index[ind - 1] : 0;
    j->ee[i] = j->index[ind];
  }
  return NULL;
}

/* mco_cbdco_nobin_dist() (command_dist.c:902-1079): `dist -r <mco dir> -o <outdir> <sketch dir>` */
static int run_search(const char *refdir, const char *qrydir, const char *outdir, const mk_dist_opts *o, int keep_shared, const char *skf,
                      int device, int nthreads, int quiet) {
  char path[PATHLEN * 2 + 32];
  size_t rn = 0, qn = 0;
  mkdir(outdir, 0700);
  snprintf(path, sizeof path, "%s/mcofiles.stat", refdir);
  uint8_t *rst = read_whole(path, &rn);
  if (!rst || rn < 20) die("need provied mco dir path for mco_co_dist() arg 1. refmco_dstat_fpath");
  snprintf(path, sizeof path, "%s/cofiles.stat", qrydir);
  uint8_t *qst = read_whole(path, &qn);
  if (!qst || qn < 32) die("need provied co dir path for mco_co_dist() arg 2.  qryco_dstat_fpath");
  uint32_t r_shuf, q_shuf;
  int32_t r_comp, ref_num, q_k, q_dr, q_comp, qry_num;
  memcpy(&r_shuf, rst, 4); memcpy(&r_comp, rst + 12, 4); memcpy(&ref_num, rst + 16, 4);
  memcpy(&q_shuf, qst, 4); memcpy(&q_k, qst + 8, 4); memcpy(&q_dr, qst + 12, 4); memcpy(&q_comp, qst + 16, 4); memcpy(&qry_num, qst + 20, 4);
  if (q_shuf != r_shuf) die("qry shuf_id: %d not match ref shuf_id: %d\ntry regenerate .co dir and feed -s the .shuffile used to generated ref database", q_shuf, r_shuf);
  if (q_comp != r_comp) die("qry comp_num: %d not match ref comp_num: %d", q_comp, r_comp);
  if (rn < 20 + (size_t)ref_num * (4 + PATHLEN) || qn < 32 + (size_t)qry_num * (4 + PATHLEN)) die("truncated stat file under %s or %s", refdir, qrydir);
  const uint32_t *ref_ct = (const uint32_t *)(rst + 20), *qry_ct = (const uint32_t *)(qst + 32);
  const char *refnames = (const char *)rst + 20 + 4 * (size_t)ref_num, *qrynames = (const char *)qst + 32 + 4 * (size_t)qry_num;
  const size_t cells = (size_t)ref_num * (size_t)qry_num;
  char skpath[PATHLEN * 2 + 32];
  snprintf(skpath, sizeof skpath, "%s/sharedk_ct.dat", outdir);
  uint32_t *ct = NULL;
  if (skf && skf[0]) { /* -f: print from an earlier run's counts (:987-990) */
    size_t sb = 0;
    snprintf(skpath, sizeof skpath, "%s", skf);
    ct = (uint32_t *)read_whole(skpath, &sb);
    if (!ct || sb < cells * 4) die("open %s failed", skpath);
  } else {
    FILE *probe = fopen(skpath, "rb");
    if (probe) { fclose(probe); die(" mco_cbdco_nobin_dist():%s: File exists", skpath); } /* :954-960 */
    ct = calloc(cells + 1, 4);
    if (!ct) die("out of memory");
    if (!quiet) printf("disf_sz=%lu\trefnum=%d\tqrynum=%d\tnum_mapping_distf=%d\tbatch_qrynum=%d\t%lu\t%lu\n", (unsigned long)(cells * 4), ref_num,
                       qry_num, 0, qry_num, (unsigned long)(cells * 4), 0ul);
    mk_mco *m;
    if (mk_mco_create(device, &m) != MK_OK) die("mk_mco_create failed: %s", mk_mco_last_error(NULL));
    int rc = mk_mco_count_begin(m, (uint32_t)ref_num, (uint32_t)qry_num);
    if (rc != MK_OK) die("mk_mco_count_begin failed (%d): %s", rc, mk_mco_last_error(m));
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 24) nthreads = 24; /* 20-24 framer threads keep PCIe busy; more only take memory bandwidth from the copies */
    for (int c = 0; c < r_comp; c++) {
      size_t gb = 0, xb = 0, ib = 0, cb = 0;
      snprintf(path, sizeof path, "%s/mco.index.%d", refdir, c);
      const uint64_t *index = map_whole(path, &xb);
      if (!index || xb != (8ull << (4 * g_component_sz))) die("%s: not a 16^%d-row index (--component-sz)", path, g_component_sz);
      snprintf(path, sizeof path, "%s/mco.%d", refdir, c);
      const uint32_t *gids = map_whole(path, &gb);
      if (!gids && gb) die("%s", path);
      snprintf(path, sizeof path, "%s/combco.index.%d", qrydir, c);
      uint8_t *qidx = read_whole(path, &ib);
      if (!qidx || ib < 8 * ((size_t)qry_num + 1)) die("%s", path);
      snprintf(path, sizeof path, "%s/combco.%d", qrydir, c);
      uint8_t *qids = read_whole(path, &cb);
      const uint64_t nq = ((const uint64_t *)qidx)[qry_num];
      if (!qids || cb / 4 < nq) die("%s", path);
      uint64_t *es = malloc(8 * (nq + 1)), *ee = malloc(8 * (nq + 1));
      if (!es || !ee) die("out of memory");
      pthread_t th[64];
      extent_job jobs[64];
      int started = 0;
      for (int t = 0; t < nthreads; t++) {
        jobs[t] = (extent_job){index, (const uint32_t *)qids, es, ee, nq * (uint64_t)t / (uint64_t)nthreads, nq * (uint64_t)(t + 1) / (uint64_t)nthreads};
        if (t + 1 == nthreads || pthread_create(&th[started], NULL, extent_worker, &jobs[t]) != 0) extent_worker(&jobs[t]);
        else started++;
      }
      for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
      static const uint32_t no_gids = 0; /* a database of empty sketches has an empty mco.N: still a valid (all-zero) search */
      rc = mk_mco_count_add(m, gids ? gids : &no_gids, gb / 4, NULL, es, ee, (const uint64_t *)qidx, qry_ct);
      if (rc != MK_OK) die("mk_mco_count_add failed (%d): %s", rc, mk_mco_last_error(m));
      free(es); free(ee); free(qids); free(qidx);
      munmap((void *)index, xb);
      if (gids) munmap((void *)gids, gb);
    }
    rc = mk_mco_count_finish(m, ct);
    if (rc != MK_OK) die("mk_mco_count_finish failed (%d): %s", rc, mk_mco_last_error(m));
    mk_mco_destroy(m);
    FILE *sk = fopen(skpath, "wb");
    if (!sk) die(" mco_cbdco_nobin_dist()::%s", skpath);
    if (cells && fwrite(ct, 4, cells, sk) != cells) die("%s: write failed", skpath);
    if (fclose(sk)) die("%s: write failed", skpath);
  }
  snprintf(path, sizeof path, "%s/distance.out", outdir);
  FILE *fp = fopen(path, "w");
  if (!fp) die("dist_print_nobin():%s", path);
  if (o->num_neigb > 1024 || o->num_neigb > ref_num) die("neighborN_max %d should smaller than NREF %d and ref_num %d", o->num_neigb, 1024, ref_num);
  int rc = mk_dist_print(fp, o, q_k, q_dr, (uint32_t)ref_num, (uint32_t)qry_num, ref_ct, qry_ct, refnames, qrynames, ct);
  if (rc != MK_OK) die("mk_dist_print failed (%d)", rc);
  if (fclose(fp)) die("%s: write failed", path);
  if (!keep_shared) remove(skpath); /* :1633 (also a file given with -f) */
  free(ct); free(rst); free(qst);
  return 0;
}

/* ---- one input file -> one sketch on the engine(s) of `c` (run_stageI()'s loop body, command_dist.c:367-404) ------------- */
typedef struct {
  strlist *files;
  const mk_params *P;
  int abundance, uniq, first_nonfq, kmerocrs, kmerqlty, nthreads, quiet;
  pf_t *pf;
  int nworkers;
} job_opts;

/* first half: begin + everything pushed (queued on the engine's stream: a pinned text buffer stays with the caller in *held
 * until the finish has waited); second half: finish + the reference's error messages */
static void sketch_file_push(ctx_t *c, const job_opts *o, int i, int *held_out) {
  const char *path = o->files->v[i];
  c->next_ordinal = 0;
  const int fq = is_fastq(path);
  /* -A holds until the file loop reaches the first input that is no FASTQ (command_dist.c:389-392) */
  const int abundance = o->abundance && i < o->first_nonfq;
  if (fq && abundance && !o->quiet) printf("running mt_shortreads2koc()\n");
  c->begun = 0;
  c->mode = fq ? (abundance ? MK_MODE_KOC : MK_MODE_OCC_SET) : (o->uniq ? MK_MODE_UNIQ_SET : MK_MODE_SET);
  c->min_occ = o->kmerocrs;
  c->occ = fq && !abundance; c->qmin = o->kmerqlty; c->TL = o->P->TL; c->nthreads = o->nthreads;
  int handled = 0, held = -1;
  pf_t *pf = o->pf;
  if (o->nworkers) {
    pthread_mutex_lock(&pf->mu);
    while (!pf->slots[i].ready) pthread_cond_wait(&pf->cv_ready, &pf->mu);
    pf_slot sl = pf->slots[i];
    pthread_mutex_unlock(&pf->mu);
    if (sl.err == MK_ERR_IO) die("%s: cannot open", path);
    if (sl.err == MK_ERR_STATE && !fq) die("fastco():eof or fread error file=%s", path);
    if (sl.err == MK_ERR_FORMAT && !fq) die("fasta2co(): can not find seqences head start from '>' 0 (%s ends inside a header line)", path);
    if (sl.err) die("%s: sequence or header line of 4095+ characters: outside the FASTQ framing contract (iseq2comem.c:656,673)", path);
    if (!sl.too_big) {
      if (sl.is_text) {
        mk_engine *e = sketch_engine(c);
        CHECK(e, mk_sketch_push_stream(e, pf->bufs[sl.buf], sl.text_bytes, 1));
        held = sl.buf; /* pinned text: the copy is queued, the buffer goes back once the finish below has waited */
      } else {
        if (sl.nrows) push_rows(c, pf->bufs[sl.buf], sl.stride, sl.nrows);
        pthread_mutex_lock(&pf->mu); /* push returned: the buffer has been copied to the device */
        pf->free_bufs[pf->nfree++] = sl.buf;
        pthread_cond_broadcast(&pf->cv_buf);
        pthread_mutex_unlock(&pf->mu);
      }
      handled = 1;
    }
  }
  if (!handled) {
    if (fq) sketch_fastq(c, path);
    else sketch_fasta(c, path, o->P->TL);
  }
  (void)sketch_engine(c); /* an input without a single row still gives an (empty) sketch */
  *held_out = held;
}

static void sketch_file_finish(ctx_t *c, const job_opts *o, int i, int held, mk_result *res, double *t_finish) {
  const char *path = o->files->v[i];
  pf_t *pf = o->pf;
  mk_engine *eng = c->eng;
  const double tf = now_s();
  int rc;
  if (c->multi) {
    rc = g_multi.finish(c->multi, res, &c->gather_ms, &c->tail_ms);
    if (rc != MK_OK && rc != MK_ERR_CROWDED) die("mk_multi_finish failed (%d): %s", rc, g_multi.last_error(c->multi));
  } else rc = mk_sketch_finish(eng, res);
  *t_finish += now_s() - tf;
  if (held >= 0) {
    pthread_mutex_lock(&pf->mu);
    pf->free_bufs[pf->nfree++] = held;
    pthread_cond_broadcast(&pf->cv_buf);
    pthread_mutex_unlock(&pf->mu);
  }
  if (rc == MK_ERR_CROWDED) die("the context space is too crowd, try rerun the program using -k%d", o->P->k + 1);
  if (rc == MK_ERR_FORMAT) die("fasta2co(): can not find seqences head start from '>' 0 (%s ends inside a header line)", path); /* iseq2comem.c:269 */
  if (rc != MK_OK) die("mk_sketch_finish failed (%d): %s", rc, mk_last_error(eng));
}

static int sketch_one_file(ctx_t *c, const job_opts *o, int i, mk_result *res, double *t_finish) {
  int held = -1;
  sketch_file_push(c, o, i, &held);
  sketch_file_finish(c, o, i, held, res, t_finish);
  return MK_OK;
}

/* ---- several GPUs, several files: whole files are the unit (SURVEY.md 8e "config 5"; the reference's team over files,
 * command_dist.c:363-372).  One engine and one driver thread per listed GPU; the drivers take the next file from a shared
 * cursor, sketch it whole on their engine and leave a copy of the result; the main thread writes the results in file order.
 * No exchange between GPUs, no RCCL.  Naming one GPU several times gives that many engines on it: one file's finish then runs
 * beside the next file's scan. */
typedef struct {
  int ready;
  mk_result res;       /* a copy: components, ids and counts in malloc'ed memory */
} file_result;
typedef struct {
  pthread_t th;
  ctx_t c;
  engine_future fut_dummy;
  const job_opts *o;
  int device;
  int *cursor;
  file_result *out;
  pthread_mutex_t *mu;
  pthread_cond_t *cv;
  double t_finish;
  int created;
} shard_driver;

static void copy_result(const mk_result *r, mk_result *dst) {
  dst->component_num = r->component_num;
  dst->total = r->total;
  dst->components = malloc(sizeof(mk_component) * (size_t)(r->component_num > 0 ? r->component_num : 1));
  if (!dst->components) die("out of memory");
  for (int k = 0; k < r->component_num; k++) {
    const mk_component *a = &r->components[k];
    mk_component *b = &dst->components[k];
    b->n = a->n;
    b->ids = a->n ? malloc(4 * (size_t)a->n) : NULL;
    b->counts = a->n && a->counts ? malloc(2 * (size_t)a->n) : NULL;
    if (a->n && (!b->ids || (a->counts && !b->counts))) die("out of memory");
    if (a->n) memcpy(b->ids, a->ids, 4 * (size_t)a->n);
    if (a->n && a->counts) memcpy(b->counts, a->counts, 2 * (size_t)a->n);
  }
}
static void free_result(mk_result *r) {
  for (int k = 0; k < r->component_num; k++) { free(r->components[k].ids); free(r->components[k].counts); }
  free(r->components);
  r->components = NULL;
}

static void *shard_driver_run(void *arg) {
  shard_driver *d = arg;
  if (!d->c.eng) { /* engines 1..: created here, all at the same time; engine 0 came through the start-up future */
    mk_engine *e = NULL;
    const int rc = mk_engine_create(d->o->P, d->device, &e);
    if (rc != MK_OK) die("mk_engine_create on GPU %d failed (%d): %s", d->device, rc, mk_last_error(NULL));
    d->c.eng = e;
    d->c.engs[0] = e;
    engine_apply_options(&d->c, e);
  }
  for (;;) {
    pthread_mutex_lock(d->mu);
    const int i = (*d->cursor)++;
    pthread_mutex_unlock(d->mu);
    if (i >= d->o->files->n) return NULL;
    mk_result res;
    sketch_one_file(&d->c, d->o, i, &res, &d->t_finish);
    file_result fr;
    memset(&fr, 0, sizeof fr);
    copy_result(&res, &fr.res);
    mk_result_release(d->c.eng, &res);
    fr.ready = 1;
    pthread_mutex_lock(d->mu);
    d->out[i] = fr;
    pthread_cond_broadcast(d->cv);
    pthread_mutex_unlock(d->mu);
  }
}

/* engines beside the first one on the same GPU, created one after the other by a thread of their own; `ready` counts them */
#define MAX_ENGINES_PER_GPU 4
typedef struct { pthread_t th; const mk_params *P; int device, want; mk_engine *eng[MAX_ENGINES_PER_GPU - 1]; int ready, failed; char err[512]; } extra_engines_t;
static void *extra_engines_run(void *arg) {
  extra_engines_t *s = arg;
  for (int k = 0; k < s->want; k++) {
    if (mk_engine_create(s->P, s->device, &s->eng[k]) != MK_OK) {
      s->eng[k] = NULL;
      snprintf(s->err, sizeof s->err, "%s", mk_last_error(NULL));
      __atomic_store_n(&s->failed, 1, __ATOMIC_RELEASE);
      break;
    }
    __atomic_store_n(&s->ready, k + 1, __ATOMIC_RELEASE);
  }
  return NULL;
}


/* ---- a directory of genomes in batches (mk_sketch_batch_begin / _end) -------------------------------------------------------
 * The reference sketches one file per OpenMP thread (command_dist.c:363-372).  Here consecutive small FASTA files go to the
 * device TOGETHER: the reader threads do the FASTA walk and leave the files of a batch as PACKED ROWS in one pinned buffer
 * (mk_fasta_pack_rows; the scan kernel reads them there), the engine runs ONE launch sequence for all of them, and two batches are
 * in flight while the readers fill the next.  (--batch-text, and every geometry without a scan kernel for packed rows: the files'
 * TEXT at 1 KiB-aligned offsets, one host-to-device copy, the walk on the device.)
 * A file that cannot go that way (FASTQ, compressed, a pipe, larger than BATCH_FILE_MAX) is sketched alone, in its place in the
 * input order.  --no-batch gives the file-by-file driver. */
#define BATCH_FILE_MAX ((size_t)32 << 20)
#define BATCH_BUFS 4      /* buffers of text batches; packed rows: as many as 2 GiB hold, BATCH_BUFS_MAX at most */
#define BATCH_BUFS_MAX 64
static int g_no_batch = 0;
static size_t g_batch_bytes = 0;                 /* --batch-mib: text per batch (0: 256 MiB where the readers pack rows, 128 MiB of text) */
static int g_batch_files = 256;                  /* --batch-files */
static int g_batch_narrow = 0;                   /* --batch-narrow: rows of 152 bases (the FASTQ rows) instead of wide rows of 240 */
static int g_batch_text = 0;                     /* --batch-text: the files' TEXT goes to the device (which then does the FASTA walk too) */

typedef struct { int first, n, batch; } bjob;    /* files [first, first + n); batch: its number among the batches, -1 = one file alone */
typedef struct {
  strlist *files;
  uint64_t *fsize;              /* per file: size when the batches were planned; a reader that meets EOF earlier writes what it got */
  uint8_t *grew;                /* per file: there are bytes behind that size -- the file is sketched alone, from all of its text */
  bjob *jobs; int njobs;
  uint8_t *buf[BATCH_BUFS_MAX]; size_t bufcap; int nbufs;
  uint64_t *foff;               /* offset of every file inside its batch's buffer */
  uint32_t rows_format;         /* MK_ROWS_PACKED / MK_ROWS_WIDE */
  int rows_TL;                  /* != 0: the readers do the FASTA walk and leave PACKED ROWS in the buffer (mk_fasta_pack_rows) ... */
  uint64_t *slot_rows, *nrows;  /* ... per file: rows its place in the buffer holds / rows it got */
  uint8_t **priv;               /* ... per file: rows that did not fit its place (wide rows: more extension rows than the place allows for) */
  int *left;                    /* per job: files not read yet */
  int *failed;                  /* per file: errno of a failed read */
  int released;                 /* batches whose buffer has been handed back */
  double cpu_read, cpu_pack, cpu_blocked; /* summed over the readers: seconds in pread(), in mk_fasta_pack_rows(), waiting for a free buffer */
  int next_job, next_file;      /* reader cursor */
  pthread_mutex_t mu;
  pthread_cond_t cv_ready, cv_free;
} breader;

static void *breader_run(void *arg) {
  breader *r = arg;
  uint8_t *txt = NULL; /* rows: the file's text, here only */
  size_t txt_cap = 0;
  for (;;) {
    pthread_mutex_lock(&r->mu);
    while (r->next_job < r->njobs && (r->jobs[r->next_job].batch < 0 || r->next_file >= r->jobs[r->next_job].n)) { r->next_job++; r->next_file = 0; }
    if (r->next_job >= r->njobs) { pthread_mutex_unlock(&r->mu); free(txt); return NULL; }
    const int j = r->next_job, k = r->next_file++;
    const bjob *job = &r->jobs[j];
    const double tb0 = now_s();
    while (job->batch - r->released >= r->nbufs) pthread_cond_wait(&r->cv_free, &r->mu); /* its buffer still belongs to an older batch */
    r->cpu_blocked += now_s() - tb0;
    pthread_mutex_unlock(&r->mu);
    const double tr0 = now_s();
    double tr1 = tr0;
    const int i = job->first + k;
    uint8_t *const place = r->buf[job->batch % r->nbufs] + r->foff[i];
    uint8_t *dst = place;
    int err = 0;
    if (r->rows_TL) {
      if (txt_cap < r->fsize[i] + 64) {
        free(txt);
        txt_cap = (size_t)r->fsize[i] + ((size_t)1 << 20);
        txt = malloc(txt_cap);
        if (!txt) { txt_cap = 0; err = ENOMEM; }
      }
      dst = txt;
    }
    const int fd = err ? -1 : open(r->files->v[i], O_RDONLY);
    if (err) ;
    else if (fd < 0) err = errno ? errno : EIO;
    else {
      /* to EOF, like the reference and the file-by-file driver (zcat -fc | fread: iseq2comem.c:226-233), within the file's place:
       * a file that SHRANK since the batches were planned is what it is now; one that GREW cannot travel in its place and is
       * sketched alone (the batch carries its first bytes along, their result is dropped) */
      uint64_t got = 0;
      while (got < r->fsize[i]) {
        const ssize_t n = pread(fd, dst + got, (size_t)(r->fsize[i] - got), (off_t)got);
        if (n < 0) { if (errno == EINTR) continue; err = errno ? errno : EIO; break; }
        if (n == 0) break;
        got += (uint64_t)n;
      }
      if (!err) {
        uint8_t probe;
        if (got < r->fsize[i]) r->fsize[i] = got;
        else if (pread(fd, &probe, 1, (off_t)got) > 0) { r->grew[i] = 1; r->fsize[i] = 0; } /* an empty member of its batch */
      }
      close(fd);
    }
    tr1 = now_s();
    if (r->rows_TL && !err && r->grew[i]) r->nrows[i] = 0;
    else if (r->rows_TL && !err) {
      /* the walk and the packing here, on this thread (what the file leaves free of its place is never looked at) */
      uint64_t got_rows = 0;
      int prc = mk_fasta_pack_rows(txt, (size_t)r->fsize[i], r->rows_TL, r->rows_format, place, r->slot_rows[i], &got_rows);
      if (prc == MK_ERR_ARG) { /* more rows than its place holds: into memory of its own (the batch's rows are then copied to the device) */
        const uint64_t full = mk_fasta_pack_bound((size_t)r->fsize[i], r->rows_TL, r->rows_format);
        void *own = NULL;
        if (posix_memalign(&own, 64, (size_t)(full ? full : 1) * MK_PACKED_PITCH) != 0) { prc = MK_ERR_NOMEM; }
        else {
          prc = mk_fasta_pack_rows(txt, (size_t)r->fsize[i], r->rows_TL, r->rows_format, own, full, &got_rows);
          if (prc == MK_OK) r->priv[i] = own; else free(own);
        }
      }
      if (prc == MK_ERR_FORMAT) err = -MK_ERR_FORMAT + 100000; /* (no errno: the text ends inside a '>' line) */
      else if (prc != MK_OK) err = EIO;
      else r->nrows[i] = got_rows;
    }
    pthread_mutex_lock(&r->mu);
    r->cpu_read += tr1 - tr0; r->cpu_pack += now_s() - tr1;
    r->failed[i] = err;
    if (--r->left[j] == 0) pthread_cond_broadcast(&r->cv_ready);
    pthread_mutex_unlock(&r->mu);
  }
}

#ifndef MK_DEFAULT_ENGINES
#define MK_DEFAULT_ENGINES 2
#endif
int main(int argc, char **argv) {
  g_t0 = now_s();
  setvbuf(stdout, NULL, _IOLBF, 0);
  if (argc < 2) usage();
  if (!strcmp(argv[1], "shuffle")) return cmd_shuffle(argc - 2, argv + 2);
  if (!strcmp(argv[1], "set")) return cmd_set(argc - 2, argv + 2);
  if (!strcmp(argv[1], "composite")) return cmd_composite(argc - 2, argv + 2);
  if (strcmp(argv[1], "dist") != 0) die("only the `dist` sketching path, `set`, `composite -q` and `shuffle` are part of this build (got `%s`)", argv[1]);

  const char *shuf_path = NULL, *outdir = ".";
  int abundance = 0, uniq = 0, device = 0, quiet = 0, timing = 0;
  /* -p: host threads of the front end (reference default: every processor, command_dist_wrapper.c:284-293) */
  int nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN), threads_given = 0;
  if (nthreads < 1) nthreads = 1;
  const int ncpu = nthreads;
  if (nthreads > 24) nthreads = 24; /* text rows: 20-24 framer threads keep PCIe busy; more only take memory bandwidth from the copies */
  uint64_t chunk_bytes = (uint64_t)32 << 20; /* text per framing job = about 17 MiB of rows per host-to-device copy */
  int drop_pages = 1, inflight = 3, slow_exit = 0, direct_host = 0, ascii_rows = 0;
  int devs[64], ndev = 0; /* --devices 0-7 / 0,2,5 / 0,0 (the same GPU twice: two engines, for tests) */
  int engines_per_gpu = 0; /* --engines 1..4: engines taking the files of a directory in turn on one GPU (default: MK_DEFAULT_ENGINES from eight files on) */
  int allow_copies = 0;   /* --allow-device-copies: distinct GPUs whose RCCL does not come up exchange with peer copies instead of failing */
  int kmerocrs = 1, kmerqlty = 0; /* command_dist_wrapper.c:79-80 */
  const char *refpath = NULL, *skf = NULL;
  mk_dist_opts dopt = {0, 2, 0, 0, 1.0}; /* command_dist_wrapper.c:83-87 */
  int keep_shared = 0, stage2_after = 0;
  strlist args = {0};
  for (int i = 2; i < argc; i++) {
    if (!strcmp(argv[i], "-L") && i + 1 < argc) shuf_path = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) outdir = argv[++i];
    else if (!strcmp(argv[i], "-p") && i + 1 < argc) { nthreads = atoi(argv[++i]); threads_given = 1; } /* host front-end threads */
    else if (!strcmp(argv[i], "-A")) abundance = 1;
    else if (!strcmp(argv[i], "-u")) uniq = 1;
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) { /* command_dist_wrapper.c:169-180 */
      int v = atoi(argv[++i]);
      if (v > 7) { fprintf(stderr, "metakssd: -n argument is larger than Max, it has been set to 7, ignorned -n %d \n", v); v = 7; }
      else if (v < 1) { fprintf(stderr, "metakssd: -n argument is smaller than Min, it has been set to 1, ignorned -n %d \n", v); v = 1; }
      kmerocrs = v;
    }
    else if (!strcmp(argv[i], "-Q") && i + 1 < argc) kmerqlty = atoi(argv[++i]); /* :182-185 */
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) ndev = parse_devices(argv[++i], devs, 64);
    else if (!strcmp(argv[i], "--engines") && i + 1 < argc) engines_per_gpu = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--allow-device-copies")) allow_copies = 1;
    else if (!strcmp(argv[i], "--no-batch")) g_no_batch = 1; /* genome directories file by file (the driver of round 3) */
    else if (!strcmp(argv[i], "--batch-narrow")) g_batch_narrow = 1; /* rows of 152 bases (the FASTQ framers' format) instead of wide rows of 240 */
    else if (!strcmp(argv[i], "--batch-text")) g_batch_text = 1; /* batches of FASTA TEXT (the device walks it) instead of rows packed by the readers */
    else if (!strcmp(argv[i], "--batch-mib") && i + 1 < argc) g_batch_bytes = (size_t)atoi(argv[++i]) << 20;
    else if (!strcmp(argv[i], "--batch-files") && i + 1 < argc) g_batch_files = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--host-fasta")) g_host_fasta = 1; /* FASTA windows made on the host (mk_fasta_window), not on the device */
    else if (!strcmp(argv[i], "--quiet")) quiet = 1;
    else if (!strcmp(argv[i], "--component-sz") && i + 1 < argc) g_component_sz = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--timing")) timing = 1;
    else if (!strcmp(argv[i], "--chunk-mib") && i + 1 < argc) chunk_bytes = (uint64_t)atoi(argv[++i]) << 20;
    else if (!strcmp(argv[i], "--inflight") && i + 1 < argc) inflight = atoi(argv[++i]); /* row buffers queued for copying */
    else if (!strcmp(argv[i], "--frame-early")) g_frame_early = 1;
    else if (!strcmp(argv[i], "--early-chunks") && i + 1 < argc) g_early_chunks = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--mmap-input")) g_mmap_input = 1;
    else if (!strcmp(argv[i], "--pool-mib") && i + 1 < argc) g_pool_bytes = (uint64_t)atoll(argv[++i]) << 20;
    else if (!strcmp(argv[i], "--ahead") && i + 1 < argc) g_ahead = atoi(argv[++i]); /* row buffers the framers may run ahead by */
    else if (!strcmp(argv[i], "--direct")) direct_host = 1; /* MK_OPT_DIRECT_HOST: scan pinned row buffers in place */
    else if (!strcmp(argv[i], "--ascii-rows")) ascii_rows = 1; /* FASTQ rows as text (160 bytes per 150-base read) instead of packed (64) */
    else if (!strcmp(argv[i], "--slow-exit")) slow_exit = 1; /* destroy the engine and return from main() instead of _exit() */
    else if (!strcmp(argv[i], "--keep-pages")) drop_pages = 0; /* measurement: leave all unmapping to the final munmap */
    else if (!strcmp(argv[i], "-r") && i + 1 < argc) refpath = argv[++i];
    else if (!strcmp(argv[i], "-M") && i + 1 < argc) dopt.metric = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-O") && i + 1 < argc) dopt.outfields = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-N") && i + 1 < argc) dopt.num_neigb = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-D") && i + 1 < argc) dopt.dthreshold = atof(argv[++i]);
    else if (!strcmp(argv[i], "--correction") && i + 1 < argc) dopt.correction = atoi(argv[++i]);
    else if (!strncmp(argv[i], "--correction=", 13)) dopt.correction = atoi(argv[i] + 13);
    else if (!strcmp(argv[i], "--keepskf")) keep_shared = 1;
    else if (!strcmp(argv[i], "-f") && i + 1 < argc) skf = argv[++i];
    else if (argv[i][0] == '-' && argv[i][1]) die("option %s is not part of the sketching path built here", argv[i]);
    else sl_push(&args, argv[i]);
  }
  /* dist_dispatch(), command_dist.c:49-250: what the arguments are decides the mode */
  if (refpath) {
    if (dopt.metric < 0 || dopt.metric > 1 || dopt.outfields < 0 || dopt.outfields > 2) die("-M takes 0/1 and -O 0/1/2");
    struct stat rst;
    if (stat(refpath, &rst) != 0) die("test_get_fullpath()::%s: %s", refpath, strerror(errno)); /* command_dist.c:321-322 */
    const int ref_co = dir_has(refpath, "cofiles.stat"), ref_mco = dir_has(refpath, "mcofiles.stat");
    if (ref_co && !ref_mco) run_stage2(refpath, refpath, device, quiet); /* :119-122: the index goes next to the sketches */
    if (ref_co || ref_mco) {
      if (args.n == 0) return 0;
      if (dir_has(args.v[0], "mcofiles.stat") && !dir_has(args.v[0], "cofiles.stat"))
        die("when -r specified, the query sould not be .mco format, the valid query format shoulde be .fas/.fq file or .co");
      if (!dir_has(args.v[0], "cofiles.stat")) die("please specify valid query genomes seq or .co file for database search");
      return run_search(refpath, args.v[0], outdir, &dopt, keep_shared, skf, device, nthreads, quiet);
    }
    /* raw sequences: stage I into outdir (no abundances, :111), then stage II in place (:112-114) */
    args.n = 0;
    sl_push(&args, refpath);
    abundance = 0;
    stage2_after = 1;
  } else if (args.n >= 1 && dir_has(args.v[0], "cofiles.stat")) {
    if (args.n > 1) return run_combine(args.n, args.v, outdir); /* :191-194 */
    return run_stage2(args.v[0], outdir, device, quiet); /* :187-190 */
  }
  if (!shuf_path) die("-L <file.shuf> is required (numeric levels generate a time-seeded table in the reference; use `metakssd shuffle`)");
  if (args.n == 0) die("please specify the input/query files");
  strlist files = {0};
  discover(&files, args.n, args.v);
  if (files.n == 0) die("not valid raw seq format");

  const double t0 = g_t0;
  /* HIP start-up and then the engine's tables on a helper thread; the main thread reads the .shuf file meanwhile and goes
   * on to map and frame the input */
  /* several GPUs and several files: the files are dealt to the GPUs whole (no exchange, no RCCL); several GPUs and ONE file: its
   * rows are dealt round-robin and the partial sketches merged on GPU 0 (libmetakssd_multi.so) */
  const int shard_files = ndev > 1 && files.n > 1;
  engine_future fut;
  engine_start(&fut, ndev ? devs[0] : device, devs, shard_files ? 0 : ndev, allow_copies ? MK_MULTI_ALLOW_DEVICE_COPIES : 0u);
  mk_shuf sh;
  int rc = mk_shuf_read(shuf_path, &sh);
  if (rc != MK_OK) die("read_dim_shuffle_file(): cannot read %s (%d)", shuf_path, rc);
  mk_params P;
  rc = mk_params_init_csz(&sh, g_component_sz, &P);
  if (rc == MK_ERR_ARG) die("--component-sz %d with k=%d drlevel=%d: more than 16 components (or out of 1..8)", g_component_sz, sh.k, sh.drlevel);
  if (rc != MK_OK) die("get_hashsz(): primer_ind out of range(0 ~ 24) for k=%d drlevel=%d (command_dist.c:291-303)", sh.k, sh.drlevel);
  if (!quiet) printf("rand_id=%d\thalf_ctx_len=%d\thashsize=%u\thashlimit=%u\n", P.shuf_id, P.k, P.hashsize, P.hashlimit);
  const double t_shuf = now_s() - t0;

  { /* a directory of genomes goes to the device in batches of files, each file with a small table of its own: the engine's
     * hashsize-slot tables (21 GB at L2K11) are then made only if a file falls out of its batch (MK_ENGINE_LAZY_TABLES) */
    int plain = 0;
    for (int i = 0; i < files.n; i++) plain += !is_fastq(files.v[i]) && !is_compressed(files.v[i]);
    fut.lazy_tables = plain >= 2 && !g_no_batch && !g_host_fasta && !engines_per_gpu && !shard_files && ndev <= 1;
  }
  engine_params(&fut, &P);
  /* more engines on the same GPU for directories of many files (see the file loop), created beside the first and joining as
   * they come up.  Two by default, four at most: 1024 genomes of 4 Mbases at L3K10 go through at 3 750 genomes/s with one engine,
   * 6 100-6 400 with two, 5 650-6 150 with three, 5 570-5 860 with four (from "engine ready" to "directory written", one box,
   * tools/gpu_session_r3x.sh); at L2K11 (21 GB of tables per engine) at 2 600 / 4 450 / 2 100-4 700 / 1 700-2 000 -- creating 21 GB
   * engines beside a working one takes 0.04 to 1 s each, two pay, more do not.  --engines 1 turns it off */
  if (engines_per_gpu < 0 || engines_per_gpu > MAX_ENGINES_PER_GPU) die("--engines takes 1..%d", MAX_ENGINES_PER_GPU);
  /* which inputs can travel in batches: plain FASTA files of moderate size, when the device parses the text and one engine works */
  uint64_t *fsize = calloc((size_t)files.n, sizeof *fsize);
  uint8_t *elig = calloc((size_t)files.n, 1);
  if (!fsize || !elig) die("out of memory");
  int n_elig = 0;
  if (files.n > 1 && !g_no_batch && !g_host_fasta && !engines_per_gpu && !shard_files && ndev <= 1) {
    if (g_batch_files < 1 || g_batch_files > (int)MK_BATCH_MAX_FILES) die("--batch-files takes 1..%u", MK_BATCH_MAX_FILES);
    /* a batch's tables hold about five times the keys its files are expected to leave (text / 16^drlevel), 2^26 slots at most */
    const size_t geo_cap = (((size_t)1 << 25) / 5u) << (4 * P.drlevel > 20 ? 20 : 4 * P.drlevel);
    /* rows, small sketches (drlevel >= 3: a key in 4 096 k-mers): 256 MiB of text a batch -- the kernels behind a batch's scan (tables,
     * layout, dump: 0.15-0.2 ms during which the link idles) come half as often as with 128 MiB: 1 024 genomes 3 ms sooner at L3K10.
     * Larger sketches (L2K11: a key in 256) keep 128 MiB: their tables and dumps grow with the batch, 256 MiB was 0-5 ms slower there,
     * 512 MiB 15 ms */
    if (!g_batch_bytes) g_batch_bytes = (!g_batch_text && mk_params_packed_ok(&P) && P.drlevel >= 3) ? (size_t)256 << 20 : (size_t)128 << 20;
    if (g_batch_bytes > geo_cap) g_batch_bytes = geo_cap;
    if (g_batch_bytes > ((size_t)512 << 20)) g_batch_bytes = (size_t)512 << 20;
    if (g_batch_bytes < ((size_t)1 << 20)) g_batch_bytes = (size_t)1 << 20;
    for (int i = 0; i < files.n; i++) {
      struct stat fst;
      if (is_fastq(files.v[i]) || is_compressed(files.v[i]) || stat(files.v[i], &fst) != 0 || !S_ISREG(fst.st_mode)) continue;
      if (fst.st_size <= 0 || (size_t)fst.st_size > BATCH_FILE_MAX || (size_t)fst.st_size > g_batch_bytes) continue;
      fsize[i] = (uint64_t)fst.st_size; elig[i] = 1; n_elig++;
    }
    /* test hook (tests/test_golden.py): MK_TEST_PLAN_SKEW="<m>:<d>" plans every m-th file with a size off by d bytes, i.e. as if
     * the file had grown (d < 0) or shrunk (d > 0) between this stat() and the readers' pread() */
    if (getenv("MK_TEST_PLAN_SKEW")) {
      int m = 0; long d = 0;
      if (sscanf(getenv("MK_TEST_PLAN_SKEW"), "%d:%ld", &m, &d) == 2 && m > 0)
        for (int i = 0; i < files.n; i += m)
          if (elig[i] && (d > 0 || fsize[i] > (uint64_t)(-d))) fsize[i] = (uint64_t)((long)fsize[i] + d);
    }
  }
  const int use_batch = n_elig >= 2;
  const int n_engines = (!use_batch && !shard_files && ndev <= 1 && files.n >= 8) ? (engines_per_gpu ? engines_per_gpu : MK_DEFAULT_ENGINES) : 1;
  const int two_engines = n_engines > 1;
  extra_engines_t extra;
  memset(&extra, 0, sizeof extra);
  if (two_engines) {
    extra.P = &P; extra.device = ndev ? devs[0] : device; extra.want = n_engines - 1;
    if (pthread_create(&extra.th, NULL, extra_engines_run, &extra) != 0) die("cannot start a thread: %s", strerror(errno));
  }
  ctx_t c;
  memset(&c, 0, sizeof c);
  c.fut = &fut;
  c.chunk_bytes = chunk_bytes;
  c.drop_pages = drop_pages;
  c.inflight = inflight;
  c.direct_host = direct_host;
  c.packed = !ascii_rows && mk_params_packed_ok(&P);
  /* packed rows are 64 bytes a read on PCIe instead of 160: the framers, not the link, bound a FASTQ file then, and 32 of them did
   * better than 24 (50 M reads: 0.20-0.24 s against 0.22-0.26 from process start; 48 and more are slower again) */
  if (c.packed && !threads_given && ncpu >= 32) nthreads = 32;

  /* -A stays on only if every input is FASTQ: the reference switches it off when its file loop reaches the first
   * non-FASTQ input (command_dist.c:389-392) and then writes no combco.N.a at all (:427-431).  FASTQ files in front of
   * that point have by then been sketched by mt_shortreads2koc() + write_fqkoc2files() (every key, -n / -Q ignored), the
   * ones behind it go through fastq2co(): the same here, in our file order. */
  int first_nonfq = files.n;
  for (int i = files.n - 1; i >= 0; i--)
    if (!is_fastq(files.v[i])) first_nonfq = i;
  const int koc_dir = abundance && first_nonfq == files.n;
  mk_sketchdir *sd;
  rc = mk_sketchdir_open(outdir, &P, koc_dir, files.n, &sd);
  if (rc != MK_OK) die("cannot create sketch directory %s (%d)", outdir, rc);

  /* worker threads prepare files ahead when there are several inputs */
  pf_t pf;
  memset(&pf, 0, sizeof pf);
  pthread_t workers[PF_MAX_BUFS];
  int nworkers = 0;
  if (files.n > 1 && nthreads > 1 && !use_batch) {
    pf.files = &files; pf.TL = P.TL;
    pf.qmin = kmerqlty;
    pf.koc_until = abundance ? first_nonfq : 0; /* files in front of this index are read the mt_shortreads2koc way */
    pf.nbufs = nthreads < PF_MAX_BUFS ? nthreads : PF_MAX_BUFS;
    if (shard_files && pf.nbufs < ndev + 1) pf.nbufs = ndev + 1 < PF_MAX_BUFS ? ndev + 1 : PF_MAX_BUFS; /* every driver may hold one */
    if (pf.nbufs < n_engines + 1) pf.nbufs = n_engines + 1; /* every engine in turn holds one while its file is in flight (4 + 1 <= PF_MAX_BUFS) */
    if (pf.nbufs > files.n) pf.nbufs = files.n;
    pf.slots = calloc(files.n, sizeof(pf_slot));
    pthread_mutex_init(&pf.mu, NULL);
    pthread_cond_init(&pf.cv_buf, NULL);
    pthread_cond_init(&pf.cv_ready, NULL);
    /* one pinned block for all row buffers (an anonymous mapping touched in parallel and registered once): sixteen
     * hipHostMalloc calls of 64 MiB took 0.2 s of a 0.9 s run over 1024 genomes */
    void *arena = NULL;
    if (mk_host_arena_alloc(&arena, (size_t)pf.nbufs * ROWBUF) == MK_OK) {
      for (int b = 0; b < pf.nbufs; b++) {
        pf.bufs[b] = (uint8_t *)arena + (size_t)b * ROWBUF;
        pf.free_bufs[pf.nfree++] = b;
      }
    } else {
      for (int b = 0; b < pf.nbufs; b++) {
        if (mk_host_alloc((void **)&pf.bufs[b], ROWBUF) != MK_OK) { pf.nbufs = b; break; }
        pf.free_bufs[pf.nfree++] = b;
      }
    }
    for (int t = 0; t < pf.nbufs; t++)
      if (pthread_create(&workers[nworkers], NULL, pf_worker, &pf) == 0) nworkers++;
    if (nworkers == 0) { free(pf.slots); pf.slots = NULL; }
  }

  double t_finish = 0, t_batch_wait_read = 0, t_batch_pin = 0, t_batch_begin = 0, t_readers_read = 0, t_readers_pack = 0, t_readers_blocked = 0;
  int nbatches_done = 0;
  job_opts jo = {&files, &P, abundance, uniq, first_nonfq, kmerocrs, kmerqlty, nthreads, quiet, &pf, nworkers};
  if (first_nonfq < files.n && abundance) printf("Warning: close abundance mode (-A) since non-fastq file input.\n");
  if (use_batch) {
    /* jobs in input order: runs of eligible files cut into batches, every other file alone */
    bjob *jobs = calloc((size_t)files.n, sizeof *jobs);
    uint64_t *foff = calloc((size_t)files.n, sizeof *foff);
    int *left = calloc((size_t)files.n, sizeof *left), *failed = calloc((size_t)files.n, sizeof *failed);
    uint8_t *grew = calloc((size_t)files.n + 1, 1);
    /* the readers do the FASTA walk and pack (0.48 bytes a base cross PCIe instead of 1.01, and the scan kernel reads them where the
     * readers left them) wherever there is a scan kernel for packed rows; --batch-text: the text travels and the device walks it */
    const int batch_rows = !g_batch_text && mk_params_packed_ok(&P);
    const uint32_t rows_format = g_batch_narrow ? MK_ROWS_PACKED : MK_ROWS_WIDE;
    uint64_t *slot_rows = calloc((size_t)files.n, sizeof *slot_rows), *nrows_of = calloc((size_t)files.n, sizeof *nrows_of);
    uint8_t **priv = calloc((size_t)files.n, sizeof *priv);
    if (!jobs || !foff || !left || !failed || !slot_rows || !nrows_of || !priv || !grew) die("out of memory");
    int njobs = 0, nbatches = 0;
    size_t bufcap = 0;
    for (int i = 0; i < files.n;) {
      if (!elig[i]) { jobs[njobs].first = i; jobs[njobs].n = 1; jobs[njobs].batch = -1; njobs++; i++; continue; }
      size_t at = 0;
      int n = 0;
      size_t text_at = 0;
      while (i + n < files.n && elig[i + n] && n < g_batch_files && (n == 0 || text_at + fsize[i + n] <= g_batch_bytes)) {
        foff[i + n] = at;
        if (batch_rows) { /* its place: the rows its text can give at most */
          /* (wide rows: an extension row for one row in sixteen -- a genome has them at contig ends and runs of N; a file with
           * more gets memory of its own, breader_run) */
          const uint64_t full = mk_fasta_pack_bound((size_t)fsize[i + n], P.TL, rows_format);
          slot_rows[i + n] = rows_format == MK_ROWS_WIDE ? full / 2u + full / 32u + 8u : full;
          at += (size_t)slot_rows[i + n] * MK_PACKED_PITCH;
        } else at += ((size_t)fsize[i + n] + 1023u) & ~(size_t)1023u;
        text_at += ((size_t)fsize[i + n] + 1023u) & ~(size_t)1023u;
        n++;
      }
      if (at > bufcap) bufcap = at;
      jobs[njobs].first = i; jobs[njobs].n = n; jobs[njobs].batch = nbatches++; left[njobs] = n; njobs++;
      i += n;
    }
    breader br;
    memset(&br, 0, sizeof br);
    br.files = &files; br.fsize = fsize; br.grew = grew; br.jobs = jobs; br.njobs = njobs; br.foff = foff; br.left = left; br.failed = failed;
    br.rows_TL = batch_rows ? P.TL : 0; br.rows_format = rows_format; br.slot_rows = slot_rows; br.nrows = nrows_of; br.priv = priv;
    br.bufcap = (bufcap + 4096 + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1); /* whole 2 MiB granules: every buffer is pinned on its own */
    pthread_mutex_init(&br.mu, NULL);
    pthread_cond_init(&br.cv_ready, NULL);
    pthread_cond_init(&br.cv_free, NULL);
    /* the buffers: mapped and touched now, so that the readers can start while the runtime and the engine come up; pinned when the
     * first batch is handed over (see cli_sink_alloc) */
    /* Rows: the readers may run AHEAD by as many batches as 2 GiB of buffers hold -- the 75 ms the runtime and the engine's queue
     * take to come up are time to walk and pack in (1 024 genomes of 4 Mbases are 2 GB of rows: all of them are packed by then, and
     * what is left is the link's time); the pages are touched by the readers as they write.  Text: four buffers, touched here. */
    int want_bufs = BATCH_BUFS;
    if (batch_rows) {
      want_bufs = (int)((((size_t)2 << 30) + br.bufcap - 1) / br.bufcap);
      if (want_bufs < BATCH_BUFS) want_bufs = BATCH_BUFS;
      if (want_bufs > BATCH_BUFS_MAX) want_bufs = BATCH_BUFS_MAX;
    }
    const int nbufs = nbatches < want_bufs ? nbatches : want_bufs;
    br.nbufs = nbufs;
    size_t arena_len = 0;
    uint8_t *arena = batch_rows ? arena_map_untouched((size_t)nbufs * br.bufcap, &arena_len) : arena_map_unpinned((size_t)nbufs * br.bufcap, &arena_len);
    if (!arena) die("out of memory (%zu bytes of batch buffers)", (size_t)nbufs * br.bufcap);
    repoison(arena, arena_len); /* (MK_POISON: fresh anonymous pages are zeros otherwise) */
    for (int b = 0; b < nbufs; b++) br.buf[b] = arena + (size_t)b * br.bufcap;
    int buf_pinned[BATCH_BUFS_MAX];
    memset(buf_pinned, 0, sizeof buf_pinned);
    /* (rows: 16 readers by default.  On the 64-core host 1 024 genomes of 4 Mbases took 0.108-0.113 s with 16, 0.114-0.118 with 24,
     * 0.120-0.132 with 32 and up to 0.23 with 48: 16 pack the directory inside the runtime's start-up (5-6 GB/s of text each), and every
     * further thread is memory traffic beside that start-up -- the engine is ready at 0.078-0.084 s with 16 readers, 0.088-0.093 with
     * 24, 0.095-0.107 with 32) */
    int nreaders = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
    if (batch_rows && !threads_given && nreaders > 16) nreaders = 16;
    pthread_t readers[64];
    int started = 0;
    for (int t = 0; t < nreaders; t++) if (pthread_create(&readers[started], NULL, breader_run, &br) == 0) started++;
    if (!started) die("cannot start a thread: %s", strerror(errno));
    c.nthreads = nthreads;
    jo.nworkers = 0;
    const int mode = uniq ? MK_MODE_UNIQ_SET : MK_MODE_SET;
    mk_batch_file *bf = calloc((size_t)g_batch_files, sizeof *bf);
    mk_batch_result *bres = calloc((size_t)g_batch_files, sizeof *bres);
    if (!bf || !bres) die("out of memory");
    int fly[2], nfly = 0, done_files = 0;
    const int btrace = getenv("MK_BATCH_TRACE") != NULL;
    #define BATCH_END_OLDEST() do { \
      const bjob *bj_ = &jobs[fly[0]]; \
      const double tf_ = now_s(); \
      rc = mk_sketch_batch_end(c.eng, bres); \
      t_finish += now_s() - tf_; \
      if (btrace) fprintf(stderr, "[batch %d] end: called %.3f returned %.3f ms\n", bj_->batch, (tf_ - g_t0) * 1e3, (now_s() - g_t0) * 1e3); \
      if (rc != MK_OK) die("mk_sketch_batch_end failed (%d): %s", rc, mk_last_error(c.eng)); \
      for (int k_ = 0; k_ < bj_->n; k_++) { \
        const char *path_ = files.v[bj_->first + k_]; \
        if (priv[bj_->first + k_]) { free(priv[bj_->first + k_]); priv[bj_->first + k_] = NULL; } \
        if (grew[bj_->first + k_]) { /* longer than its place: alone, from the whole file, where it stands in the order */ \
          mk_result res_; \
          sketch_one_file(&c, &jo, bj_->first + k_, &res_, &t_finish); \
          rc = mk_sketchdir_add(sd, path_, &res_); \
          if (rc != MK_OK) die("writing sketch for %s failed (%d)", path_, rc); \
          mk_result_release(c.eng, &res_); \
          if (!quiet) printf("%d/%d decomposing %s\r", ++done_files, files.n, path_); \
          continue; \
        } \
        if (bres[k_].status == MK_ERR_CROWDED) die("the context space is too crowd, try rerun the program using -k%d", P.k + 1); \
        if (bres[k_].status == MK_ERR_FORMAT) die("fasta2co(): can not find seqences head start from '>' 0 (%s ends inside a header line)", path_); \
        if (bres[k_].status != MK_OK) die("sketching %s failed (%d)", path_, bres[k_].status); \
        rc = mk_sketchdir_add(sd, path_, &bres[k_].r); \
        if (rc != MK_OK) die("writing sketch for %s failed (%d)", path_, rc); \
        if (!quiet) printf("%d/%d decomposing %s\r", ++done_files, files.n, path_); \
      } \
      repoison(br.buf[bj_->batch % nbufs], br.bufcap); /* (MK_POISON: the next batch in this buffer starts from the pattern) */ \
      pthread_mutex_lock(&br.mu); br.released++; pthread_cond_broadcast(&br.cv_free); pthread_mutex_unlock(&br.mu); \
      fly[0] = fly[1]; nfly--; \
    } while (0)
    for (int j = 0; j < njobs; j++) {
      const bjob *bj = &jobs[j];
      if (bj->batch < 0) { /* one file alone, in its place: everything in front of it comes home first */
        while (nfly) BATCH_END_OLDEST();
        mk_result res;
        sketch_one_file(&c, &jo, bj->first, &res, &t_finish);
        rc = mk_sketchdir_add(sd, files.v[bj->first], &res);
        if (rc != MK_OK) die("writing sketch for %s failed (%d)", files.v[bj->first], rc);
        mk_result_release(c.eng, &res);
        if (!quiet) printf("%d/%d decomposing %s\r", ++done_files, files.n, files.v[bj->first]);
        continue;
      }
      {
        const double tw = now_s();
        pthread_mutex_lock(&br.mu);
        while (left[j] > 0) pthread_cond_wait(&br.cv_ready, &br.mu);
        pthread_mutex_unlock(&br.mu);
        t_batch_wait_read += now_s() - tw;
      }
      for (int k = 0; k < bj->n; k++) {
        const int i = bj->first + k;
        if (failed[i] == -MK_ERR_FORMAT + 100000) die("fasta2co(): can not find seqences head start from '>' 0 (%s ends inside a header line)", files.v[i]);
        if (failed[i]) die("%s: %s", files.v[i], strerror(failed[i]));
        bf[k].text = priv[i] ? priv[i] : br.buf[bj->batch % nbufs] + foff[i];
        bf[k].n = batch_rows ? nrows_of[i] * MK_PACKED_PITCH : fsize[i];
      }
      (void)engine_get(&c);
      const double tp0 = now_s();
      if (!buf_pinned[bj->batch % nbufs]) { /* the runtime is up now.  Buffer by buffer: only the first buffer's pinning lies in front
                                               * of the first batch, every other buffer is pinned behind the begin of the batch in front
                                               * of it (below), while the device works */
        if (mk_host_register(br.buf[bj->batch % nbufs], br.bufcap) != MK_OK) die("pinning the batch buffers failed: %s", mk_last_error(NULL));
        buf_pinned[bj->batch % nbufs] = 1;
      }
      t_batch_pin += now_s() - tp0;
      if (nfly == 2) BATCH_END_OLDEST();
      if (c.t_first_push == 0) c.t_first_push = now_s() - g_t0;
      {
        const double tb = now_s();
        rc = batch_rows ? mk_sketch_batch_begin_rows(c.eng, mode, rows_format, bf, (uint32_t)bj->n) : mk_sketch_batch_begin(c.eng, mode, bf, (uint32_t)bj->n);
        t_batch_begin += now_s() - tb;
        nbatches_done++;
        if (btrace) fprintf(stderr, "[batch %d] begin: called %.3f returned %.3f ms\n", bj->batch, (tb - g_t0) * 1e3, (now_s() - g_t0) * 1e3);
      }
      if (rc != MK_OK) die("mk_sketch_batch_begin failed (%d): %s", rc, mk_last_error(c.eng));
      c.t_last_push = now_s() - g_t0;
      fly[nfly++] = j;
      if (bj->batch + 1 < nbatches && !buf_pinned[(bj->batch + 1) % nbufs]) { /* the next batch's buffer, beside this batch's kernels */
        const double tp1 = now_s();
        if (mk_host_register(br.buf[(bj->batch + 1) % nbufs], br.bufcap) != MK_OK) die("pinning the batch buffers failed: %s", mk_last_error(NULL));
        buf_pinned[(bj->batch + 1) % nbufs] = 1;
        t_batch_pin += now_s() - tp1;
      }
    }
    while (nfly) BATCH_END_OLDEST();
    #undef BATCH_END_OLDEST
    for (int t = 0; t < started; t++) pthread_join(readers[t], NULL);
    t_readers_read = br.cpu_read; t_readers_pack = br.cpu_pack; t_readers_blocked = br.cpu_blocked;
    (void)engine_get(&c);
  } else if (shard_files) {
    /* several engines, whole files each */
    shard_driver *drv = calloc((size_t)ndev, sizeof *drv);
    file_result *results = calloc((size_t)files.n, sizeof *results);
    pthread_mutex_t smu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t scv = PTHREAD_COND_INITIALIZER;
    int cursor = 0;
    if (!drv || !results) die("out of memory");
    c.nthreads = nthreads;
    (void)engine_get(&c); /* engine 0 (devs[0]) from the start-up thread */
    for (int j = 0; j < ndev; j++) {
      shard_driver *d = &drv[j];
      d->o = &jo; d->device = devs[j]; d->cursor = &cursor; d->out = results; d->mu = &smu; d->cv = &scv;
      d->c.chunk_bytes = chunk_bytes; d->c.drop_pages = drop_pages; d->c.inflight = inflight; d->c.direct_host = direct_host;
      d->c.ndev = 1;
      if (j == 0) { d->c.eng = c.eng; d->c.engs[0] = c.eng; }
      if (pthread_create(&d->th, NULL, shard_driver_run, d) != 0) die("cannot start a thread: %s", strerror(errno));
    }
    for (int i = 0; i < files.n; i++) {
      pthread_mutex_lock(&smu);
      while (!results[i].ready) pthread_cond_wait(&scv, &smu);
      file_result fr = results[i];
      pthread_mutex_unlock(&smu);
      rc = mk_sketchdir_add(sd, files.v[i], &fr.res);
      if (rc != MK_OK) die("writing sketch for %s failed (%d)", files.v[i], rc);
      free_result(&fr.res);
      if (!quiet) printf("%d/%d decomposing %s\r", i + 1, files.n, files.v[i]);
    }
    for (int j = 0; j < ndev; j++) { pthread_join(drv[j].th, NULL); t_finish += drv[j].t_finish; c.nrows_total += drv[j].c.nrows_total; }
  } else if (two_engines) {
    /* many files on one GPU: up to four engines take the files in turn and ONE thread drives them all -- a file is begun and
     * pushed on an engine that is free (copy and kernels queued on its stream); when none is free the oldest file in flight is
     * finished on its engine: while the host waits there and writes the result, the GPU already works on the files behind it,
     * and the small kernels of neighbouring files overlap.  (With a single engine the GPU idles through every host round trip:
     * 160 us of kernels and 75 us of copy per 4 Mbase genome against 420 us per genome in steady state.)  Engines join as they
     * come up (21 GB of tables at L2K11 take anything from 0.04 to 1 s each beside a working GPU): until then fewer take turns. */
    ctx_t cx[MAX_ENGINES_PER_GPU];
    cx[0] = c;
    (void)engine_get(&cx[0]);
    int have = 1; /* engines in use */
    struct { int file, eng, held; } fly[MAX_ENGINES_PER_GPU]; /* files in flight, oldest first */
    int nfly = 0, busy[MAX_ENGINES_PER_GPU] = {0, 0, 0, 0}, extra_warned = 0;
    for (int i = 0; i < files.n || nfly; ) {
      if (__atomic_load_n(&extra.failed, __ATOMIC_ACQUIRE) && !extra_warned) {
        /* the extra engines are a speed-up that is on by default: one that does not come up (a smaller or shared GPU: an L2K11
         * engine is 21 GB of tables) must not turn a run that works on one engine into an error -- unless --engines asked for them */
        if (engines_per_gpu) die("mk_engine_create (engine %d of %d, --engines %d) failed: %s", have + 1, n_engines, engines_per_gpu, extra.err);
        fprintf(stderr, "metakssd: warning: engine %d of %d did not come up (%s): going on with %d\n", 1 + __atomic_load_n(&extra.ready, __ATOMIC_ACQUIRE) + 1,
                n_engines, extra.err, 1 + __atomic_load_n(&extra.ready, __ATOMIC_ACQUIRE));
        extra_warned = 1;
      }
      const int ready = 1 + __atomic_load_n(&extra.ready, __ATOMIC_ACQUIRE);
      while (have < ready) { /* a new engine: its own buffers for files that stream */
        cx[have] = cx[0];
        cx[have].io = NULL; cx[have].rows = NULL; cx[have].arena = NULL; cx[have].arena_bytes = 0; cx[have].nrows_total = 0;
        cx[have].eng = extra.eng[have - 1]; cx[have].engs[0] = extra.eng[have - 1]; cx[have].ndev = 1; cx[have].multi = NULL;
        engine_apply_options(&cx[have], cx[have].eng);
        have++;
      }
      int e = -1;
      if (i < files.n)
        for (int k = 0; k < have; k++) if (!busy[k]) { e = k; break; }
      if (e >= 0) {
        fly[nfly].file = i; fly[nfly].eng = e; fly[nfly].held = -1;
        sketch_file_push(&cx[e], &jo, i, &fly[nfly].held);
        busy[e] = 1; nfly++; i++;
        continue;
      }
      /* every engine has a file (or the files are out): the oldest comes home */
      const int j = fly[0].file, ej = fly[0].eng;
      mk_result res;
      sketch_file_finish(&cx[ej], &jo, j, fly[0].held, &res, &t_finish);
      rc = mk_sketchdir_add(sd, files.v[j], &res);
      if (rc != MK_OK) die("writing sketch for %s failed (%d)", files.v[j], rc);
      mk_result_release(cx[ej].eng, &res);
      if (!quiet) printf("%d/%d decomposing %s\r", j + 1, files.n, files.v[j]);
      busy[ej] = 0;
      for (int k = 1; k < nfly; k++) fly[k - 1] = fly[k];
      nfly--;
    }
    /* engines that come up after the last file are not used; the fast exit below does not wait for them */
    if (slow_exit || stage2_after) pthread_join(extra.th, NULL);
    uint64_t rows_all = 0;
    for (int k = 0; k < have; k++) rows_all += cx[k].nrows_total;
    c.nrows_total = rows_all;
    c.eng = cx[0].eng;
  } else {
    for (int i = 0; i < files.n; i++) {
      mk_result res;
      sketch_one_file(&c, &jo, i, &res, &t_finish);
      rc = mk_sketchdir_add(sd, files.v[i], &res);
      if (rc != MK_OK) die("writing sketch for %s failed (%d)", files.v[i], rc);
      mk_result_release(c.eng, &res);
      if (!quiet) printf("%d/%d decomposing %s\r", i + 1, files.n, files.v[i]);
    }
  }
  if (!quiet) printf("\n");
  rc = mk_sketchdir_close(sd);
  if (rc != MK_OK) die("closing sketch directory failed (%d)", rc);
  const double t_written = now_s() - t0;
  if (!quiet) printf("sketched %llu rows from %d file(s) in %.3f s\n", (unsigned long long)c.nrows_total, files.n, t_written);
  if (timing) /* one JSON line for bench.py / tools: seconds since process start unless named *_s */
    printf("{\"timing\": {\"t0_abs\": %.6f, \"exit_abs\": %.6f, \"shuf_read\": %.4f, \"hip_ready\": %.4f, \"engine_ready\": %.4f, \"first_push\": %.4f, \"last_push\": %.4f, \"unmapped\": %.4f, "
           "\"written\": %.4f, \"finish_s\": %.4f, \"begin_s\": %.4f, \"rows\": %llu, \"threads\": %u, \"chunks\": %llu, \"chunks_discarded\": %llu, "
           "\"serial_rows\": %llu, \"stream_setup_s\": %.4f, \"stream_wait_frame_s\": %.4f, \"stream_push_s\": %.4f, \"stream_total_s\": %.4f, "
           "\"push_call_s\": %.4f, \"wait_call_s\": %.4f, \"push_call_max_s\": %.4f, \"first_push_call_s\": %.4f, \"gpus\": %d, \"gather_ms\": %.3f, \"tail_ms\": %.3f, \"transport\": \"%s\", "
           "\"batches\": %d, \"batch_wait_readers_s\": %.4f, \"batch_pin_s\": %.4f, \"batch_begin_s\": %.4f, "
           "\"readers_read_cpu_s\": %.4f, \"readers_pack_cpu_s\": %.4f, \"readers_blocked_s\": %.4f, "
           "\"pin_calls\": %d, \"pin_s\": %.4f, \"pinned_mib\": %.1f, \"pool_mib\": %.1f}}\n",
           g_t0, now_s(), t_shuf, fut.t_hip_ready, fut.t_ready, c.t_first_push, c.t_last_push, c.t_unmapped, t_written, t_finish, c.t_begin_s, (unsigned long long)c.nrows_total,
           c.fq_stats.threads, (unsigned long long)c.fq_stats.chunks, (unsigned long long)c.fq_stats.chunks_discarded,
           (unsigned long long)c.fq_stats.serial_rows, c.fq_stats.t_setup_s, c.fq_stats.t_wait_frame_s, c.fq_stats.t_push_s,
           c.fq_stats.t_total_s, c.fq_stats.t_push_call_s, c.fq_stats.t_wait_call_s, c.fq_stats.t_push_call_max_s,
           c.fq_stats.t_first_push_call_s, c.ndev ? c.ndev : 1, c.gather_ms, c.tail_ms, c.multi ? g_multi.transport(c.multi) : "-",
           nbatches_done, t_batch_wait_read, t_batch_pin, t_batch_begin, t_readers_read, t_readers_pack, t_readers_blocked,
           c.pin ? c.pin->calls : 0, c.pin ? c.pin->t_pin_s : 0.0, c.pin ? (double)c.pin->pinned_bytes / 1048576.0 : 0.0, (double)c.arena_bytes / 1048576.0);
  if (stage2_after) {
    if (c.rows) mk_host_free(c.rows);
    free(c.io);
    if (!c.multi) mk_engine_destroy(engine_get(&c));
    mk_shuf_free(&sh);
    return run_stage2(outdir, outdir, device, quiet);
  }
  if (slow_exit) { /* profilers collect their data in exit handlers */
    if (!c.multi) mk_engine_destroy(engine_get(&c));
    return 0;
  }
  /* everything is on disk: leave without tearing down 2 GB of device tables and the pinned pools page by page */
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}

