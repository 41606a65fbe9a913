/*
 * metakssd_main.c -- `metakssd dist` command line on top of the C ABI (host C, links libmetakssd_hip.so).
 *
 * Keeps the reference's CLI surface for the sketching path:
 *     metakssd dist -L <file.shuf> [-A] [-u] [-o outdir] [-p N] <fastq/fasta files or directories>...
 * (option table command_dist_wrapper.c:32-59, defaults :68-96, query-only branch of dist_dispatch()
 * command_dist.c:201-248, run_stageI() :341-500) and writes the same sketch directory
 * (cofiles.stat, combco.N, combco.index.N, combco.N.a).
 *
 * Differences, all documented in DESIGN.md: inputs are processed in discovery order (the reference
 * applies a time-seeded shuffle, command_dist.c:215); FASTQ without -A (the 4-bit -n/-Q path,
 * fastq2co) and every non-sketching mode of `dist` are not part of this build; -p is accepted and
 * ignored (the GPU does the work); --device selects the GPU.
 */
#define _GNU_SOURCE
#include "metakssd_hip.h"

#include <dirent.h>
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#define PATHLEN 256 /* global_basic.h:32 */

static void die(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "metakssd: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  exit(1);
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

typedef struct {
  mk_engine *eng;
  uint8_t *io;   /* raw text */
  uint8_t *rows; /* pinned rows */
  uint64_t next_ordinal;
  uint64_t nrows_total;
} ctx_t;

#define CHECK(e, call)                                                  \
  do {                                                                  \
    int _rc = (call);                                                   \
    if (_rc != MK_OK) die("%s failed (%d): %s", #call, _rc, mk_last_error(e)); \
  } while (0)

/* input opened like the reference does: through `zcat -fc` when compressed (iseq2comem.c:216,666-669),
 * directly otherwise (same bytes, no child process) */
static FILE *open_input(const char *path, int *is_pipe) {
  if (is_compressed(path)) {
    char cmd[PATHLEN * 2 + 16];
    snprintf(cmd, sizeof cmd, "zcat -fc '%s'", path);
    *is_pipe = 1;
    return popen(cmd, "r");
  }
  *is_pipe = 0;
  return fopen(path, "rb");
}

static void sketch_fastq(ctx_t *c, const char *path) {
  int is_pipe;
  FILE *f = open_input(path, &is_pipe);
  if (!f) die("mtfastq2koc():%s: %s", path, strerror(errno));
  uint32_t stride = 160;
  size_t have = 0;
  int eof = 0;
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
      int rc = mk_fastq_frame(c->io + off, have - off, eof, c->rows, stride, ROWBUF / stride, &nrows, &used);
      if (nrows) {
        CHECK(c->eng, mk_sketch_push_reads(c->eng, c->rows, stride, nrows, c->next_ordinal));
        c->next_ordinal += nrows;
        c->nrows_total += nrows;
      }
      off += used;
      if (rc == MK_ERR_ARG && stride < 4096) { stride = stride * 2 > 4096 ? 4096 : stride * 2; continue; } /* longer read: widen rows */
      if (rc == MK_ERR_ARG || rc == MK_ERR_FORMAT)
        die("%s: sequence or header line of 4095+ characters: outside the FASTQ framing contract (iseq2comem.c:656,673)", path);
      if (rc != MK_OK) die("mk_fastq_frame failed (%d)", rc);
      if (nrows == 0 || off >= have) break;
    }
    memmove(c->io, c->io + off, have - off);
    have -= off;
    if (have == IOBUF) die("%s: a single FASTQ record exceeds the %zu-byte I/O buffer", path, IOBUF);
    if (eof && have && off == 0) break; /* trailing partial record: dropped like the reference does */
  }
  if (is_pipe) pclose(f); else fclose(f);
}

static void sketch_fasta(ctx_t *c, const char *path, int TL) {
  int is_pipe;
  FILE *f = open_input(path, &is_pipe);
  if (!f) die("fasta2co():%s: %s", path, strerror(errno));
  const uint32_t stride = 512;
  mk_fasta_state st;
  CHECK(c->eng, mk_fasta_window_init(&st, TL));
  int eof = 0, any = 0;
  while (!eof) {
    size_t have = fread(c->io, 1, IOBUF, f);
    if (have == 0) eof = 1; else any = 1;
    size_t off = 0;
    do {
      uint64_t nrows = 0;
      size_t used = 0;
      CHECK(c->eng, mk_fasta_window(&st, c->io + off, have - off, eof, c->rows, stride, ROWBUF / stride, &nrows, &used));
      if (nrows) {
        CHECK(c->eng, mk_sketch_push_reads(c->eng, c->rows, stride, nrows, c->next_ordinal));
        c->next_ordinal += nrows;
        c->nrows_total += nrows;
      }
      off += used;
    } while (off < have);
  }
  if (!any) die("fastco():eof or fread error file=%s", path); /* iseq2comem.c:235 */
  if (is_pipe) pclose(f); else fclose(f);
}

static void usage(void) {
  fprintf(stderr,
          "usage: metakssd dist -L <file.shuf> [-A] [-u] [-o outdir] [-p N] [--device D] <fastq|fasta|dir>...\n"
          "       metakssd shuffle -k <halfK> -s <halfSubK> -l <level> [--seed N] -o <prefix>\n");
  exit(2);
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

int main(int argc, char **argv) {
  setvbuf(stdout, NULL, _IOLBF, 0);
  if (argc < 2) usage();
  if (!strcmp(argv[1], "shuffle")) return cmd_shuffle(argc - 2, argv + 2);
  if (strcmp(argv[1], "dist") != 0) die("only the `dist` sketching path and `shuffle` are part of this build (got `%s`)", argv[1]);

  const char *shuf_path = NULL, *outdir = ".";
  int abundance = 0, uniq = 0, device = 0, quiet = 0;
  strlist args = {0};
  for (int i = 2; i < argc; i++) {
    if (!strcmp(argv[i], "-L") && i + 1 < argc) shuf_path = argv[++i];
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) outdir = argv[++i];
    else if (!strcmp(argv[i], "-p") && i + 1 < argc) ++i; /* accepted, unused */
    else if (!strcmp(argv[i], "-A")) abundance = 1;
    else if (!strcmp(argv[i], "-u")) uniq = 1;
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--quiet")) quiet = 1;
    else if (argv[i][0] == '-' && argv[i][1]) die("option %s is not part of the sketching path built here", argv[i]);
    else sl_push(&args, argv[i]);
  }
  if (!shuf_path) die("-L <file.shuf> is required (numeric levels generate a time-seeded table in the reference; use `metakssd shuffle`)");
  if (args.n == 0) die("please specify the input/query files");
  strlist files = {0};
  discover(&files, args.n, args.v);
  if (files.n == 0) die("not valid raw seq format");

  double t0 = now_s();
  const int dbg = getenv("MK_DEBUG") != NULL;
  mk_shuf sh;
  int rc = mk_shuf_read(shuf_path, &sh);
  if (rc != MK_OK) die("read_dim_shuffle_file(): cannot read %s (%d)", shuf_path, rc);
  mk_params P;
  rc = mk_params_init(&sh, &P);
  if (rc != MK_OK) die("get_hashsz(): primer_ind out of range(0 ~ 24) for k=%d drlevel=%d (command_dist.c:291-303)", sh.k, sh.drlevel);
  if (!quiet) printf("rand_id=%d\thalf_ctx_len=%d\thashsize=%u\thashlimit=%u\n", P.shuf_id, P.k, P.hashsize, P.hashlimit);

  if (dbg) fprintf(stderr, "[t] shuf read + params: %.3f s\n", now_s() - t0);
  ctx_t c = {0};
  rc = mk_engine_create(&P, device, &c.eng);
  if (rc != MK_OK) die("mk_engine_create failed (%d): %s", rc, mk_last_error(NULL));
  if (dbg) fprintf(stderr, "[t] + engine create: %.3f s\n", now_s() - t0);
  c.io = malloc(IOBUF);
  if (!c.io || mk_host_alloc((void **)&c.rows, ROWBUF) != MK_OK) die("out of memory");
  if (dbg) fprintf(stderr, "[t] + host buffers: %.3f s\n", now_s() - t0);

  /* -A is switched off for good by the first non-FASTQ input (command_dist.c:389-392) */
  for (int i = 0; i < files.n; i++)
    if (!is_fastq(files.v[i]) && abundance) {
      abundance = 0;
      printf("Warning: close abundance mode (-A) since non-fastq file input.\n");
    }
  mk_sketchdir *sd;
  rc = mk_sketchdir_open(outdir, &P, abundance, files.n, &sd);
  if (rc != MK_OK) die("cannot create sketch directory %s (%d)", outdir, rc);

  for (int i = 0; i < files.n; i++) {
    const char *path = files.v[i];
    c.next_ordinal = 0;
    if (is_fastq(path)) {
      if (!abundance) die("%s: FASTQ without -A (fastq2co, -n/-Q 4-bit counts) is not built yet", path);
      if (!quiet) printf("running mt_shortreads2koc()\n");
      CHECK(c.eng, mk_sketch_begin(c.eng, MK_MODE_KOC));
      sketch_fastq(&c, path);
    } else {
      CHECK(c.eng, mk_sketch_begin(c.eng, uniq ? MK_MODE_UNIQ_SET : MK_MODE_SET));
      sketch_fasta(&c, path, P.TL);
    }
    if (dbg) fprintf(stderr, "[t] + framing/push of %s: %.3f s\n", path, now_s() - t0);
    mk_result res;
    rc = mk_sketch_finish(c.eng, &res);
    if (rc == MK_ERR_CROWDED) die("the context space is too crowd, try rerun the program using -k%d", P.k + 1);
    if (rc != MK_OK) die("mk_sketch_finish failed (%d): %s", rc, mk_last_error(c.eng));
    rc = mk_sketchdir_add(sd, path, &res);
    if (rc != MK_OK) die("writing sketch for %s failed (%d)", path, rc);
    mk_result_release(c.eng, &res);
    if (!quiet) printf("%d/%d decomposing %s\r", i + 1, files.n, path);
  }
  if (!quiet) printf("\n");
  if (dbg) fprintf(stderr, "[t] + finish/write: %.3f s\n", now_s() - t0);
  rc = mk_sketchdir_close(sd);
  if (rc != MK_OK) die("closing sketch directory failed (%d)", rc);
  if (!quiet) printf("sketched %llu rows from %d file(s) in %.3f s\n", (unsigned long long)c.nrows_total, files.n, now_s() - t0);
  mk_host_free(c.rows);
  free(c.io);
  mk_engine_destroy(c.eng);
  if (dbg) fprintf(stderr, "[t] + teardown: %.3f s\n", now_s() - t0);
  mk_shuf_free(&sh);
  return 0;
}
