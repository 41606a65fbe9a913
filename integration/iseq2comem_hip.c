/*
 * integration/iseq2comem_hip.c -- drop-in replacement for the reference's iseq2comem.c on top of libmetakssd_hip.so.
 *
 * Compiled WITH the reference's own headers (-I<reference>) and linked with every other translation unit of the reference
 * taken unchanged (oracle/Makefile, target ref_hip): the reference's command line, option parsing, input discovery,
 * run_stageI(), the per-component concatenation, cofiles.stat and everything downstream stay the reference's code; only
 * the sequence -> sketch functions declared in iseq2comem.h (:9-25) come from here.  It defines what iseq2comem.c defines:
 *
 *   globals    dim_shuffle, hashsize, hashlimit, component_num            iseq2comem.c:44,50-52 (extern in command_dist.h:108-110)
 *   void   seq2co_global_var_initial(void)                               iseq2comem.c:54-86
 *   llong *mt_shortreads2koc(seqfname, co, pipecmd, p)                   iseq2comem.c:657-727   } -A FASTQ
 *   unsigned int write_fqkoc2files(cofilename, co)                       iseq2comem.c:516-562   }
 *   llong *fastq2co(seqfname, co, pipecmd, Q, M)                         iseq2comem.c:323-419   } FASTQ without -A
 *   llong  write_fqco2file(cofilename, co)                               iseq2comem.c:596-621   }
 *   llong *fasta2co / uniq_fasta2co(seqfname, co, pipecmd)               iseq2comem.c:218-315, 729-828 } FASTA
 *   llong  wrt_co2cmpn_use_inn_subctx(cofilename, co)                    iseq2comem.c:625-652   }
 *   int    reads2mco(...)                                                iseq2comem.c:89-214    --byread: not part of this build
 *
 * The caller's table `co` (CO[tid], command_dist.c:344-348) is never touched: it is the HANDLE under which a sketch function
 * leaves its result for the dump function that run_stageI() calls next with the same pointer (command_dist.c:380-398).
 * Errors keep the reference's behaviour: err() = message + exit.  One engine on GPU MK_DEVICE (default 0); the file-level
 * OpenMP team of run_stageI() is serialised on it.
 */
#include "iseq2comem.h"

#include "command_dist.h"
#include "global_basic.h"

#include "metakssd_hip.h"

#include <err.h>
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

/* ---- what iseq2comem.c defines for the rest of the program (iseq2comem.c:44,50-52) ---- */
dim_shuffle_t *dim_shuffle;
unsigned int hashsize;
unsigned int hashlimit;
int component_num;

static mk_params g_params;
static mk_engine *g_engine;
static pthread_mutex_t g_engine_mu = PTHREAD_MUTEX_INITIALIZER;

/* results parked under the caller's `co` pointer between the sketch call and the dump call */
typedef struct parked {
  const void *key;
  int ncomp, koc;
  unsigned int total;
  uint64_t *n;
  uint32_t **ids;
  uint16_t **counts;
  struct parked *next;
} parked;
static parked *g_parked;
static pthread_mutex_t g_parked_mu = PTHREAD_MUTEX_INITIALIZER;

void seq2co_global_var_initial(void) {
  mk_shuf s;
  s.id = dim_shuffle->dim_shuffle_stat.id;
  s.k = dim_shuffle->dim_shuffle_stat.k;
  s.subk = dim_shuffle->dim_shuffle_stat.subk;
  s.drlevel = dim_shuffle->dim_shuffle_stat.drlevel;
  s.table = dim_shuffle->shuffled_dim;
  s.len = 1ULL << (4 * s.subk);
  if (mk_params_init(&s, &g_params) != MK_OK) err(errno, "seq2co_global_var_initial(): unusable .shuf parameters k=%d subk=%d drlevel=%d", s.k, s.subk, s.drlevel);
  if (g_params.hashsize != hashsize) /* the caller has set hashsize = get_hashsz(dim_shuffle) (command_dist.c:221) */
    err(errno, "seq2co_global_var_initial(): hashsize %u, the engine expects %u", hashsize, g_params.hashsize);
  hashlimit = g_params.hashlimit;
  component_num = g_params.component_num;
  printf("rand_id=%d\thalf_ctx_len=%d\thashsize=%d\thashlimit=%d\n", s.id, s.k, hashsize, hashlimit); /* iseq2comem.c:62 */
}

static mk_engine *engine(void) { /* call with g_engine_mu held */
  if (!g_engine) {
    const char *d = getenv("MK_DEVICE");
    if (mk_engine_create(&g_params, d ? atoi(d) : 0, &g_engine) != MK_OK) err(errno, "mk_engine_create: %s", mk_last_error(NULL));
  }
  return g_engine;
}

static void park(const void *key, const mk_result *r, int koc) {
  parked *p = calloc(1, sizeof *p);
  if (!p) err(errno, "out of memory");
  p->key = key; p->ncomp = r->component_num; p->koc = koc; p->total = (unsigned int)r->total;
  p->n = calloc((size_t)p->ncomp, sizeof *p->n);
  p->ids = calloc((size_t)p->ncomp, sizeof *p->ids);
  p->counts = calloc((size_t)p->ncomp, sizeof *p->counts);
  for (int c = 0; c < p->ncomp; c++) {
    const mk_component *k = &r->components[c];
    p->n[c] = k->n;
    p->ids[c] = malloc(k->n * 4 + 4);
    memcpy(p->ids[c], k->ids, k->n * 4);
    if (koc) { p->counts[c] = malloc(k->n * 2 + 2); memcpy(p->counts[c], k->counts, k->n * 2); }
  }
  pthread_mutex_lock(&g_parked_mu);
  p->next = g_parked;
  g_parked = p;
  pthread_mutex_unlock(&g_parked_mu);
}

static parked *unpark(const void *key, const char *who) {
  pthread_mutex_lock(&g_parked_mu);
  parked **pp = &g_parked, *p = NULL;
  for (; *pp; pp = &(*pp)->next)
    if ((*pp)->key == key) { p = *pp; *pp = p->next; break; }
  pthread_mutex_unlock(&g_parked_mu);
  if (!p) err(errno, "%s: no sketch has been made for this table", who);
  return p;
}

static void unpark_free(parked *p) {
  for (int c = 0; c < p->ncomp; c++) { free(p->ids[c]); free(p->counts[c]); }
  free(p->n); free(p->ids); free(p->counts); free(p);
}

/* ---- input: the reference always reads through popen("zcat -fc <file>") or "<pipecmd> <file>" (iseq2comem.c:216,666-669);
 * an uncompressed regular file without a pipe command is the same bytes straight from its mapping ---- */
static int plain_file(const char *path, const char *pipecmd, const uint8_t **map, size_t *size) {
  if (pipecmd[0] != '\0') return 0;
  const size_t L = strlen(path);
  if ((L > 3 && !strcmp(path + L - 3, ".gz")) || (L > 4 && !strcmp(path + L - 4, ".bz2"))) return 0;
  int fd = open(path, O_RDONLY);
  if (fd < 0) return 0;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) { close(fd); return 0; }
  void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return 0;
  *map = m; *size = (size_t)st.st_size;
  return 1;
}

static FILE *piped(const char *who, const char *path, const char *pipecmd) {
  char cmd[PATHLEN * 2 + 64];
  if (pipecmd[0] != '\0') snprintf(cmd, sizeof cmd, "%s %s", pipecmd, path); /* other pipecmd except decompress cmd */
  else snprintf(cmd, sizeof cmd, "zcat -fc %s", path);
  FILE *f = popen(cmd, "r");
  if (!f) err(errno, "%s:%s", who, cmd);
  return f;
}

#define WINBUF ((size_t)64 << 20)

/* FASTQ through the front end of the library: occ = 0 mt_shortreads2koc's reader, 1 fastq2co's */
static void sketch_fastq(const char *who, mk_engine *e, const char *path, const char *pipecmd, int threads, int occ, int Q) {
  const uint8_t *map;
  size_t size;
  mk_fastq_opts o;
  memset(&o, 0, sizeof o);
  o.occ = occ; o.qmin = Q; o.TL = g_params.TL; o.nthreads = threads < 1 ? 1 : threads; o.inflight = 3;
  o.packed = mk_params_packed_ok(&g_params); /* reads of up to 152 bases cross PCIe as 64-byte packed rows */
  if (plain_file(path, pipecmd, &map, &size)) {
    o.drop_pages = 1;
    const int rc = mk_sketch_push_fastq(e, map, size, &o, 0, NULL);
    munmap((void *)map, size);
    if (rc != MK_OK) err(errno, "%s: %s: %s", who, path, rc == MK_ERR_FORMAT || rc == MK_ERR_ARG ? "a line is longer than the reader's fgets() width" : mk_last_error(e));
    return;
  }
  FILE *f = piped(who, path, pipecmd);
  uint8_t *io = malloc(WINBUF), *rows = NULL;
  if (!io || mk_host_alloc((void **)&rows, WINBUF) != MK_OK) err(errno, "%s: out of memory", who);
  size_t have = 0;
  int eof = 0;
  uint32_t stride = 160;
  uint64_t ord = 0, records = 0;
  while (!eof || have) {
    if (!eof) {
      const size_t r = fread(io + have, 1, WINBUF - have, f);
      have += r;
      if (r == 0) eof = 1;
    }
    size_t off = 0;
    for (;;) {
      uint64_t nrows = 0, nrec = 0;
      size_t used = 0;
      const int rc = mk_fastq_frame_mt(io + off, have - off, eof, occ, Q, g_params.TL, records, rows, stride, WINBUF / stride, o.nthreads,
                                       &nrows, &nrec, &used);
      records += nrec;
      if (nrows && mk_sketch_push_reads(e, rows, stride, nrows, ord) != MK_OK) err(errno, "%s: %s", who, mk_last_error(e));
      ord += nrows;
      off += used;
      if (rc == MK_ERR_ARG && stride < 4096) { stride = stride * 2 > 4096 ? 4096 : MK_ROW_PITCH(stride * 2); continue; }
      if (rc != MK_OK) err(errno, "%s: %s: a line is longer than the reader's fgets() width", who, path);
      if (nrows == 0 || off >= have) break;
    }
    memmove(io, io + off, have - off);
    have -= off;
    if (have == WINBUF) err(errno, "%s: %s: a record larger than the %zu-byte window", who, path, WINBUF);
    if (eof && have && off == 0) break; /* trailing partial record: dropped like the reference does */
  }
  pclose(f);
  free(io);
  mk_host_free(rows);
}

/* fasta2co()'s walk over the file (iseq2comem.c:224-279) happens on the device: the bytes the pipe delivers are pushed as
 * they are, mk_sketch_push_stream drops line ends and header lines there */
static void sketch_fasta(const char *who, mk_engine *e, const char *path, const char *pipecmd) {
  FILE *f = piped(who, path, pipecmd);
  uint8_t *io = malloc(WINBUF);
  if (!io) err(errno, "%s: out of memory", who);
  int any = 0;
  for (;;) {
    const size_t have = fread(io, 1, WINBUF, f);
    if (have == 0) break;
    any = 1;
    if (mk_sketch_push_stream(e, io, have, 0) != MK_OK) err(errno, "%s: %s", who, mk_last_error(e));
  }
  if (!any) err(errno, "fastco():eof or fread error file=%s", path); /* iseq2comem.c:235 */
  if (mk_sketch_push_stream(e, NULL, 0, 1) != MK_OK) err(errno, "%s: %s", who, mk_last_error(e));
  pclose(f);
  free(io);
}

/* begin -> front end -> finish on the one engine, the result parked under `co` */
static llong *sketch(const char *who, int mode, int M, const char *path, llong *co, const char *pipecmd, int threads, int Q) {
  pthread_mutex_lock(&g_engine_mu);
  mk_engine *e = engine();
  const int rc0 = mode == MK_MODE_OCC_SET ? mk_sketch_begin_occ(e, M) : mk_sketch_begin(e, mode);
  if (rc0 != MK_OK) err(errno, "%s: %s", who, mk_last_error(e));
  if (mode == MK_MODE_KOC || mode == MK_MODE_OCC_SET) sketch_fastq(who, e, path, pipecmd, threads, mode == MK_MODE_OCC_SET, Q);
  else sketch_fasta(who, e, path, pipecmd);
  mk_result r;
  const int rc = mk_sketch_finish(e, &r);
  if (rc == MK_ERR_FORMAT) err(errno, "fasta2co(): can not find seqences head start from '>' %d", 0); /* iseq2comem.c:269 */
  if (rc == MK_ERR_CROWDED) /* iseq2comem.c:708-709 (and :303, :811 in the FASTA flavours) */
    err(errno, "the context space is too crowd, try rerun the program using -k%d", g_params.k + 1);
  if (rc != MK_OK) err(errno, "%s: %s", who, mk_last_error(e));
  park(co, &r, mode == MK_MODE_KOC);
  mk_result_release(e, &r);
  pthread_mutex_unlock(&g_engine_mu);
  return co;
}

llong *mt_shortreads2koc(char *seqfname, llong *co, char *pipecmd, int p) {
  printf("running mt_shortreads2koc()\n"); /* iseq2comem.c:658 */
  return sketch("mtfastq2koc()", MK_MODE_KOC, 1, seqfname, co, pipecmd, p, 0);
}

llong *fastq2co(char *seqfname, llong *co, char *pipecmd, int Q, int M) {
  if (M >= 15) err(errno, "fastq2co(): Occurence num should smaller than %d", 15); /* iseq2comem.c:325 */
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  return sketch("fastq2co()", MK_MODE_OCC_SET, M < 1 ? 1 : M, seqfname, co, pipecmd, ncpu > 16 ? 16 : (int)ncpu, Q);
}

llong *fasta2co(char *seqfname, llong *co, char *pipecmd) { return sketch("fasta2co()", MK_MODE_SET, 1, seqfname, co, pipecmd, 1, 0); }

llong *uniq_fasta2co(char *seqfname, llong *co, char *pipecmd) { return sketch("fasta2co()", MK_MODE_UNIQ_SET, 1, seqfname, co, pipecmd, 1, 0); }

/* ---- the dump halves: "<cofilename>.<component>" [+ ".a"], what run_stageI() concatenates afterwards ---- */
static llong dump(const char *who, const char *cofilename, const llong *co, int want_counts) {
  parked *p = unpark(co, who);
  char name[PATHLEN + 32];
  for (int c = 0; c < p->ncomp; c++) {
    snprintf(name, sizeof name, "%s.%d", cofilename, c);
    FILE *f = fopen(name, "wb");
    if (!f) err(errno, "%s", who);
    if (p->n[c] && fwrite(p->ids[c], 4, p->n[c], f) != p->n[c]) err(errno, "%s:%s", who, name);
    fclose(f);
    if (want_counts) {
      snprintf(name, sizeof name, "%s.%d.a", cofilename, c);
      if (!(f = fopen(name, "wb"))) err(errno, "%s", who);
      if (p->n[c] && fwrite(p->counts[c], 2, p->n[c], f) != p->n[c]) err(errno, "%s:%s", who, name);
      fclose(f);
    }
  }
  const llong total = p->total;
  unpark_free(p);
  return total;
}

unsigned int write_fqkoc2files(char *cofilename, llong *co) { return (unsigned int)dump("write_fqkoc2files()", cofilename, co, 1); }
llong write_fqco2file(char *cofilename, llong *co) { return dump("write_fqco2file()", cofilename, co, 0); }
llong wrt_co2cmpn_use_inn_subctx(char *cofilename, llong *co) { return dump("wrt_co2cmpn_use_inn_subctx()", cofilename, co, 0); }

int reads2mco(char *seqfname, const char *co_dir, char *pipecmd) {
  (void)co_dir; (void)pipecmd;
  errx(1, "reads2mco(): --byread sketching of %s is not part of the MI355X build (SURVEY.md section 2: out of scope)", seqfname);
  return 0;
}
