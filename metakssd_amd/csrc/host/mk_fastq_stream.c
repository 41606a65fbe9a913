/*
 * mk_fastq_stream.c -- whole-file FASTQ front end: a pool of host threads frames a memory-mapped FASTQ file into
 * fixed-stride row buffers while the calling thread hands the finished buffers to the engine in file order (host C).
 *
 * Replaces the serial reader of mt_shortreads2koc() (iseq2comem.c:657-673: 65536 x {fgets header; fgets sequence into
 * fq_buff[l]; fgets '+'; fgets quality} between the parallel loops -- the reference's Amdahl ceiling)
 * and of fastq2co() (iseq2comem.c:343-363).  Same record rule as there: records are groups of four lines counted from
 * the start of the file, whatever the lines contain.
 *
 * How the threads find record boundaries without a counting pass: the file is cut into chunks of `chunk_bytes`; the
 * framer of chunk c GUESSES the first record start at or behind the chunk's first byte (a line that starts with '@' whose
 * second successor starts with '+': in well-formed FASTQ only a header line qualifies, because the line two behind a
 * quality line is a sequence line) and frames every record that starts in front of the next chunk.  The caller's
 * thread walks the chunks in order and accepts a chunk only if its guessed start is exactly where the previous accepted
 * chunk ended -- by induction from offset 0 that is the position the serial reader would be at.  Anything else (a guess
 * fooled by a malformed file, a chunk that ran out of row space, a chunk without a record start) is framed serially from
 * the true position, so the rows and their order are those of the serial framer on every input; only the speed depends
 * on the guesses.  Row ordinals are assigned when a buffer is pushed, so framing needs no global row index.
 */
#define _GNU_SOURCE
#include "metakssd_hip.h"
#include "mk_host_internal.h"

#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define FS_NONE ((size_t)-1)

typedef struct {
  int ready;
  int buf; /* index into bufs[], -1 = returned already */
  size_t start, end;
  uint64_t nrows, nrec;
  uint32_t stride;
  int rc;
} fs_slot;

/* Where the text comes from.  A mapping (text != NULL): the framers read it in place.  A file descriptor (text == NULL,
 * mk_fastq_opts::fd): every thread preads PIECES of FS_PIECE bytes into a buffer of its own that stays in its core's cache and frames
 * the complete records of each piece -- no page of the file is ever mapped into the process, so there is no page-table work, no
 * madvise() and no TLB shoot-down beside whatever else the process is doing (the HIP runtime coming up: profiles/r05_e2e_front_end.txt). */
#define FS_PIECE ((size_t)1 << 20)
typedef struct { const uint8_t *p; size_t avail; int eof, err; } fs_view;

typedef struct {
  const uint8_t *text;
  int fd;
  size_t piece; /* bytes per pread (FS_PIECE; MK_FS_PIECE in the environment for tests: never below a maximal record) */
  size_t n, chunk;
  uint64_t nchunks;
  int occ, qmin, TL, drop_pages, packed;
  size_t buf_bytes;
  int nbufs;
  uint8_t **bufs;
  int *freelist, nfree;
  fs_slot *slots;
  uint64_t next;
  int stop;
  uint64_t early;  /* chunks that may be framed before the first push has returned (mk_fastq_opts::early_chunks); go: it has */
  int go;
  const mk_rows_sink *sink;
  pthread_mutex_t mu;
  pthread_cond_t cv_buf, cv_ready;
} fs_t;

/* the text from offset `off` on: all of it (mapping), or the next piece (descriptor; `scratch` holds FS_PIECE bytes) */
/* `upto`: the caller frames records that START below this offset -- a piece need not reach further than one maximal record (four lines of
 * fastq2co()'s fgets width) behind it; FS_NONE: a whole piece */
#define FS_RECORD_MAX ((size_t)4 * 20000 + 4096)
static fs_view fs_fetch_upto(const fs_t *f, size_t off, uint8_t *scratch, size_t upto) {
  fs_view v = {NULL, 0, 1, 0};
  if (off >= f->n) return v;
  if (f->text) { v.p = f->text + off; v.avail = f->n - off; return v; }
  size_t want = f->n - off < f->piece ? f->n - off : f->piece;
  if (upto != FS_NONE && upto > off && upto - off + FS_RECORD_MAX < want) want = upto - off + FS_RECORD_MAX;
  size_t got = 0;
  while (got < want) {
    const ssize_t r = pread(f->fd, scratch + got, want - got, (off_t)(off + got));
    if (r < 0) { if (errno == EINTR) continue; v.err = 1; break; }
    if (r == 0) { v.err = 1; break; } /* the file is shorter than the caller said */
    got += (size_t)r;
  }
  v.p = scratch; v.avail = got; v.eof = off + got >= f->n;
  return v;
}
static fs_view fs_fetch(const fs_t *f, size_t off, uint8_t *scratch) { return fs_fetch_upto(f, off, scratch, (size_t)-1); }

static double fs_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static uint32_t fs_round_stride(size_t need) { /* need = bytes of the longest row including its '\n' */
  size_t s = (need + 15u) & ~(size_t)15u;
  if (s < 32) s = 32;
  if (s > 4096) s = 4096;
  if ((s & 127u) == 0 && s < 4096) s += 16; /* a pitch of k * 128 bytes halves the scan rate (L2 channel aliasing): include/metakssd_hip.h */
  return (uint32_t)s;
}

/* first line start at or behind `from` */
static size_t fs_line_start(const uint8_t *t, size_t n, size_t from) {
  if (from == 0 || from >= n || t[from - 1] == '\n') return from < n ? from : FS_NONE;
  const uint8_t *nl = memchr(t + from, '\n', n - from);
  if (!nl || (size_t)(nl - t) + 1 >= n) return FS_NONE;
  return (size_t)(nl - t) + 1;
}

/* guessed record start in [lo, hi): a line starting with '@' whose second successor starts with '+' */
static size_t fs_guess_start(const uint8_t *t, size_t n, size_t lo, size_t hi) {
  size_t p = fs_line_start(t, n, lo);
  while (p != FS_NONE && p < hi) {
    const uint8_t *nl1 = memchr(t + p, '\n', n - p);
    if (!nl1) return FS_NONE;
    const size_t l2 = (size_t)(nl1 - t) + 1;
    if (l2 >= n) return FS_NONE;
    if (t[p] == '@') {
      const uint8_t *nl2 = memchr(t + l2, '\n', n - l2);
      if (!nl2) return FS_NONE;
      const size_t l3 = (size_t)(nl2 - t) + 1;
      if (l3 < n && t[l3] == '+') return p;
    }
    p = l2;
  }
  return FS_NONE;
}

/* longest sequence line (with its '\n') among the first records from `start`: the chunk's first row stride */
static uint32_t fs_sample_stride(const uint8_t *t, size_t n, size_t start, int occ) {
  size_t p = start, longest = 0;
  for (int rec = 0; rec < 16 && p < n; rec++) {
    for (int line = 0; line < 4 && p < n; line++) {
      const uint8_t *nl = memchr(t + p, '\n', n - p);
      const size_t len = nl ? (size_t)(nl - t) + 1 - p : n - p;
      if (line == 1 && len > longest) longest = len;
      p += len;
    }
  }
  (void)occ;
  return fs_round_stride(longest ? longest : 32);
}

/* frames the records starting in [start, stop) into `buf`; grows the stride and starts over when a read does not fit.  The text
 * comes in views (fs_fetch): one for a mapping; for a descriptor piece after piece, the complete records of each (a record cut by a
 * piece's end is framed from the next piece, which starts at it).  `scratch`: the calling thread's piece buffer (descriptor only). */
static void fs_frame_range(const fs_t *f, size_t start, size_t stop, int first_of_file, uint8_t *buf, fs_slot *s, uint8_t *scratch) {
  uint32_t stride;
  /* (descriptor: the piece the stride is sampled from is the first piece the loop below frames -- read once, not twice) */
  fs_view held = fs_fetch_upto(f, start, scratch, stop);
  size_t held_off = start;
  stride = fs_sample_stride(held.p, held.avail, 0, f->occ);
  if (f->occ && stride < 2u * (uint32_t)f->TL + 4u) stride = fs_round_stride(2u * (size_t)f->TL + 4u);
  /* packed rows (64 bytes a read, MK_ROWS_PACKED) when every read of the range has at most 152 bases; a longer one turns up as
   * MK_ERR_ARG and the range is framed again as ASCII rows */
  int packed = f->packed && stride <= MK_PACKED_MAX_BASES + 16u;
  for (;;) {
    const uint32_t sarg = packed ? (MK_PACKED_PITCH | MK_ROWS_PACKED) : stride;
    const uint32_t pitch = packed ? MK_PACKED_PITCH : stride;
    const uint64_t max_rows = f->buf_bytes / pitch;
    uint64_t rows = 0, recs = 0;
    size_t pos = start;
    int rc = MK_OK, again = 0;
    while (pos < stop && rows < max_rows) {
      if (pos != held_off) { held = fs_fetch_upto(f, pos, scratch, stop); held_off = pos; }
      const fs_view v = held;
      if (v.err) { rc = MK_ERR_IO; break; }
      if (!v.avail) break;
      const size_t st = stop - pos < v.avail ? stop - pos : v.avail;
      uint64_t nr = 0, nrec = 0;
      size_t used = 0;
      uint32_t need = 0;
      if (f->occ)
        rc = mk_fastq_frame_q_range(v.p, v.avail, st, v.eof, f->qmin, f->TL, first_of_file && pos == 0 ? 0 : 1, buf + rows * pitch, sarg,
                                    max_rows - rows, &nr, &nrec, &used, &need);
      else {
        rc = mk_fastq_frame_range(v.p, v.avail, st, v.eof, buf + rows * pitch, sarg, max_rows - rows, &nr, &used, &need);
        nrec = nr;
      }
      rows += nr; recs += nrec; pos += used;
      if (rc == MK_ERR_ARG && need && packed) { packed = 0; if (need > stride) stride = fs_round_stride((size_t)need + need / 8u); again = 1; break; }
      if (rc == MK_ERR_ARG && need > stride && stride < 4096 && !packed) { /* a longer read than sampled: wider rows, frame the range again */
        stride = fs_round_stride((size_t)need + need / 8u);
        again = 1;
        break;
      }
      if (rc == MK_ERR_ARG && !need && rows > 0 && nr == 0) { rc = MK_OK; break; } /* the next record's rows do not fit what is left of the buffer: full */
      if (rc != MK_OK) break;
      if (used == 0) { /* no complete record in this view */
        if (!v.eof && rows < max_rows) rc = MK_ERR_FORMAT; /* a record longer than a piece: beyond any line width the reference reads */
        break;
      }
    }
    if (again) continue;
    s->start = start; s->end = pos; s->nrows = rows; s->nrec = recs; s->stride = sarg; s->rc = rc;
    return;
  }
}

/* the first records of the file: mean bytes per record and the longest sequence line (with its '\n'); 0 records: nothing to go by */
static int fs_sample_records(const uint8_t *t, size_t n, size_t *rec_bytes, size_t *seq_max) {
  size_t p = 0, longest = 0;
  int rec = 0;
  for (; rec < 64 && p < n; rec++) {
    size_t q = p;
    int line = 0;
    for (; line < 4 && q < n; line++) {
      const uint8_t *nl = memchr(t + q, '\n', n - q);
      if (!nl) { q = n; break; }
      const size_t len = (size_t)(nl - t) + 1 - q;
      if (line == 1 && len > longest) longest = len;
      q = (size_t)(nl - t) + 1;
    }
    if (line < 4) break; /* an incomplete record at the end */
    p = q;
  }
  *rec_bytes = rec ? p / (size_t)rec : 0;
  *seq_max = longest;
  return rec;
}

/* guessed record start in [lo, hi) through a view (descriptor: within the first piece -- a chunk whose first record start lies
 * further in than that is framed by the calling thread, like every chunk without a usable guess) */
static size_t fs_guess_start_at(const fs_t *f, size_t lo, size_t hi, uint8_t *scratch) {
  if (f->text) return fs_guess_start(f->text, f->n, lo, hi);
  const size_t from = lo ? lo - 1 : 0; /* the byte in front of lo says whether lo is a line start */
  const fs_view v = fs_fetch(f, from, scratch);
  if (!v.avail) return FS_NONE;
  const size_t rel_hi = hi - from < v.avail ? hi - from : v.avail;
  const size_t g = fs_guess_start(v.p, v.avail, lo - from, rel_hi);
  return g == FS_NONE ? FS_NONE : from + g;
}

static void *fs_worker(void *arg) {
  fs_t *f = arg;
  uint8_t *scratch = NULL;
  if (!f->text) { /* this thread's piece buffer, touched here: it lives in the core's cache from the first pread on */
    scratch = malloc(f->piece + 64);
    if (scratch) memset(scratch, 0, f->piece + 64);
  }
  for (;;) {
    pthread_mutex_lock(&f->mu);
    while (!f->stop && f->next < f->nchunks && (f->nfree == 0 || (!f->go && f->next >= f->early))) pthread_cond_wait(&f->cv_buf, &f->mu);
    if (f->stop || f->next >= f->nchunks) { pthread_mutex_unlock(&f->mu); free(scratch); return NULL; }
    const uint64_t c = f->next++;
    const int b = f->freelist[--f->nfree]; /* taken together with the chunk number: the lowest open chunk always has a buffer */
    pthread_mutex_unlock(&f->mu);
    { const int pz = mk_poison_byte(); if (pz >= 0) memset(f->bufs[b], pz, f->buf_bytes); } /* MK_POISON: a fresh or reused row buffer starts from the pattern */

    fs_slot s;
    memset(&s, 0, sizeof s);
    s.buf = b;
    const size_t lo = (size_t)c * f->chunk, hi = lo + f->chunk < f->n ? lo + f->chunk : f->n;
#ifdef MADV_POPULATE_READ
    if (f->drop_pages && f->text) /* a file mapping: map the chunk's pages with one call instead of a fault per 64 KiB (Linux 5.14+; ignored elsewhere) */
      (void)madvise((void *)(f->text + (lo & ~(size_t)4095)), hi - (lo & ~(size_t)4095), MADV_POPULATE_READ);
#endif
    const size_t start = !f->text && !scratch ? FS_NONE : c == 0 ? 0 : fs_guess_start_at(f, lo, hi, scratch);
    if (start == FS_NONE) { s.start = s.end = FS_NONE; }
    else fs_frame_range(f, start, hi, c == 0, f->bufs[b], &s, scratch);
    if (f->drop_pages && f->text && hi - lo > ((size_t)256 << 10)) {
      /* this chunk's text is done with, except its first 128 KiB, which the previous chunk's last record may still reach
       * into (a record is at most 4 lines of 20000 characters).  Dropping a page somebody still reads is harmless -- it
       * faults back in from the page cache -- but costs time. */
      const size_t a = (lo + ((size_t)128 << 10) + 4095) & ~(size_t)4095, b2 = hi & ~(size_t)4095;
      if (b2 > a) madvise((void *)(f->text + a), b2 - a, MADV_DONTNEED);
    }

    if (s.nrows && f->sink && f->sink->ready) f->sink->ready(f->sink->ctx, f->bufs[b], f->buf_bytes);
    pthread_mutex_lock(&f->mu);
    if (s.nrows == 0) { f->freelist[f->nfree++] = b; s.buf = -1; pthread_cond_broadcast(&f->cv_buf); }
    s.ready = 1;
    f->slots[c] = s;
    pthread_cond_broadcast(&f->cv_ready);
    pthread_mutex_unlock(&f->mu);
  }
}

static void fs_release(fs_t *f, int b) {
  if (b < 0) return;
  pthread_mutex_lock(&f->mu);
  f->freelist[f->nfree++] = b;
  pthread_cond_broadcast(&f->cv_buf);
  pthread_mutex_unlock(&f->mu);
}

typedef struct { uint64_t token; int buf; } fs_inflight;

int mk_fastq_stream(const uint8_t *text, size_t n, const mk_fastq_opts *o, const mk_rows_sink *sink, uint64_t first_ordinal,
                    mk_fastq_stats *st) {
  if (!o || !sink || !sink->push || !sink->alloc || !sink->release) return MK_ERR_ARG;
  if (!text && n && o->fd <= 0) return MK_ERR_ARG; /* a mapping, or a descriptor to pread from (mk_fastq_opts::fd) */
  if (o->occ && (o->TL < 2 || o->TL > 32)) return MK_ERR_ARG;
  mk_fastq_stats stats;
  memset(&stats, 0, sizeof stats);
  const double t0 = fs_now();
  fs_t f;
  memset(&f, 0, sizeof f);
  f.text = text; f.n = n;
  f.fd = text ? -1 : o->fd;
  f.early = o->early_chunks > 0 ? (uint64_t)o->early_chunks : 0;
  f.go = o->early_chunks > 0 ? 0 : 1;
  f.piece = FS_PIECE;
  if (getenv("MK_FS_PIECE")) { const long v = atol(getenv("MK_FS_PIECE")); if (v >= 4 * 20000 + 4096) f.piece = (size_t)v; } /* (four lines of fastq2co's fgets width) */
  f.sink = sink;
  f.occ = o->occ != 0; f.qmin = o->qmin; f.TL = o->TL;
  f.packed = o->packed != 0;
  f.drop_pages = o->drop_pages != 0 && ((uintptr_t)text & 4095u) == 0;
  f.chunk = o->chunk_bytes ? (size_t)o->chunk_bytes : (size_t)32 << 20;
  if (f.chunk < 4096) f.chunk = 4096;
  f.nchunks = n ? (n + f.chunk - 1) / f.chunk : 0;
  int T = o->nthreads < 1 ? 1 : o->nthreads > 256 ? 256 : o->nthreads;
  if ((uint64_t)T > f.nchunks) T = (int)(f.nchunks ? f.nchunks : 1);
  const int depth = o->inflight < 1 ? 1 : o->inflight > 8 ? 8 : o->inflight;
  /* rows of a chunk take at most about as many bytes as its text (a record is two lines of the read's length plus a
   * header; the row is one); a chunk that needs more ends early and the pusher frames the rest serially */
  f.buf_bytes = f.chunk + f.chunk / 8 + 8192;
  /* never less than ONE maximal record needs: a read of 19 998 characters (fastq2co()'s fgets width, iseq2comem.c:319,343) is cut
   * into six 4096-byte rows; a buffer below that made the range framer give up on the record with MK_ERR_ARG and the stream
   * reported a line beyond the reference's width */
  if (f.buf_bytes < (size_t)8 * 4096 + 8192) f.buf_bytes = (size_t)8 * 4096 + 8192;
  f.nbufs = T + depth + 1 + (o->ahead < 0 ? 0 : o->ahead > 192 ? 192 : o->ahead);
  if (o->pool_bytes) {
    /* a budget for all buffers: packed rows where the file's first records allow them (buffers a fifth of the text), and as many
     * buffers as the budget holds -- one per chunk at most, then nobody ever waits for one */
    size_t rec_bytes = 0, seq_max = 0;
    uint8_t *first = text ? NULL : malloc(f.piece + 64);
    const fs_view v0 = text || first ? fs_fetch(&f, 0, first) : (fs_view){NULL, 0, 1, 0};
    const int sampled = f.packed && v0.avail ? fs_sample_records(v0.p, v0.avail, &rec_bytes, &seq_max) : 0;
    free(first);
    if (sampled >= 4 && seq_max && seq_max <= MK_PACKED_MAX_BASES + 1u && rec_bytes >= 8) {
      size_t rows = f.chunk / rec_bytes + f.chunk / rec_bytes / 8u + 256u; /* (the header lines grow with the read number: a record does not shrink) */
      { /* three significant bits: files of about the same record length get buffers of the same size (a sink that pins them keeps its pins) */
        size_t g = 1;
        while ((g << 4) <= rows) g <<= 1;
        rows = (rows + g - 1) & ~(g - 1);
      }
      f.buf_bytes = rows * MK_PACKED_PITCH + 8192;
      if (f.buf_bytes < (size_t)8 * 4096 + 8192) f.buf_bytes = (size_t)8 * 4096 + 8192;
    }
    const size_t bb = (f.buf_bytes + 4095) & ~(size_t)4095;
    uint64_t fit = o->pool_bytes / bb;
    fit = fit > 1 ? fit - 1 : 0; /* (one more for the serial fallback) */
    if (fit > f.nchunks) fit = f.nchunks;
    if (fit > (uint64_t)f.nbufs) f.nbufs = (int)(fit > 65536 ? 65536 : fit);
  }
  int rc = MK_OK;
  uint8_t *serial_buf = NULL, *serial_scratch = NULL;
  pthread_t *th = NULL;
  int nth = 0;
  fs_inflight fifo[8];
  int nfifo = 0;
  uint8_t *pool = NULL;
  size_t pool_bytes = 0;
  if (!text && n) { serial_scratch = malloc(f.piece + 64); if (!serial_scratch) { rc = MK_ERR_NOMEM; goto out_nothreads; } }
  f.bufs = calloc((size_t)f.nbufs, sizeof *f.bufs);
  f.freelist = calloc((size_t)f.nbufs, sizeof *f.freelist);
  f.slots = calloc((size_t)(f.nchunks ? f.nchunks : 1), sizeof *f.slots);
  th = calloc((size_t)T, sizeof *th);
  if (!f.bufs || !f.freelist || !f.slots || !th) { rc = MK_ERR_NOMEM; goto out_nothreads; }
  pthread_mutex_init(&f.mu, NULL);
  pthread_cond_init(&f.cv_buf, NULL);
  pthread_cond_init(&f.cv_ready, NULL);
  for (int t = 0; t < T && f.nchunks; t++)
    if (pthread_create(&th[nth], NULL, fs_worker, &f) == 0) nth++;
  if (f.nchunks && nth == 0) { rc = MK_ERR_NOMEM; goto out; }
  /* one block for all row buffers (+ one for the serial fallback): the sink decides how to get pinned memory quickly */
  f.buf_bytes = (f.buf_bytes + 4095) & ~(size_t)4095;
  pool_bytes = f.buf_bytes * (size_t)(f.nbufs + 1);
  pool = sink->alloc(sink->ctx, pool_bytes);
  if (!pool) { rc = MK_ERR_NOMEM; goto out; }
  /* the serial fallback's buffer first, then the framers' in address order; the freelist is popped from its end, so chunk c takes
   * buffer c until buffers come back (mk_fastq_opts::pool_bytes: a sink may pin its block piece by piece in that order) */
  serial_buf = pool;
  pthread_mutex_lock(&f.mu);
  for (int b = 0; b < f.nbufs; b++) f.bufs[b] = pool + (size_t)(b + 1) * f.buf_bytes;
  for (int b = f.nbufs - 1; b >= 0; b--) f.freelist[f.nfree++] = b;
  pthread_cond_broadcast(&f.cv_buf);
  pthread_mutex_unlock(&f.mu);
  stats.t_setup_s = fs_now() - t0;

  {
    size_t pos = 0;
    uint64_t ord = first_ordinal;
    /* rows of [from, to) framed on this thread and pushed synchronously: the fallback for everything the guesses miss */
#define FS_SERIAL(from_, to_)                                                                                      \
    do {                                                                                                           \
      size_t sp_ = (from_);                                                                                        \
      const size_t sto_ = (to_);                                                                                   \
      while (rc == MK_OK && sp_ < sto_) {                                                                          \
        fs_slot ss_;                                                                                               \
        memset(&ss_, 0, sizeof ss_);                                                                               \
        { const int pz_ = mk_poison_byte(); if (pz_ >= 0) memset(serial_buf, pz_, f.buf_bytes); }                  \
        fs_frame_range(&f, sp_, sto_, sp_ == 0, serial_buf, &ss_, serial_scratch);                                 \
        if (ss_.nrows) {                                                                                           \
          uint64_t tok_ = 0;                                                                                       \
          if (sink->ready) sink->ready(sink->ctx, serial_buf, f.buf_bytes);                                        \
          rc = sink->push(sink->ctx, serial_buf, ss_.stride, ss_.nrows, ord, &tok_);                               \
          if (rc == MK_OK && sink->wait) rc = sink->wait(sink->ctx, tok_);                                         \
          ord += ss_.nrows; stats.rows += ss_.nrows; stats.records += ss_.nrec; stats.serial_rows += ss_.nrows;    \
        }                                                                                                          \
        if (rc == MK_OK && ss_.rc != MK_OK) rc = ss_.rc;                                                           \
        if (ss_.end == sp_) break; /* nothing consumed: only possible at the end of the file */                    \
        sp_ = ss_.end;                                                                                             \
      }                                                                                                            \
      pos = sp_;                                                                                                   \
    } while (0)

    for (uint64_t c = 0; c < f.nchunks && rc == MK_OK; c++) {
      const double tw = fs_now();
      pthread_mutex_lock(&f.mu);
      if (!f.go && c >= f.early) { f.go = 1; pthread_cond_broadcast(&f.cv_buf); } /* (every early chunk came to nothing: nobody else would let the framers go) */
      while (!f.slots[c].ready) pthread_cond_wait(&f.cv_ready, &f.mu);
      const fs_slot s = f.slots[c];
      pthread_mutex_unlock(&f.mu);
      stats.t_wait_frame_s += fs_now() - tw;
      stats.chunks++;
      if (s.start == FS_NONE) continue; /* no record starts in this chunk as far as the guess can tell */
      if (s.start > pos) { /* the previous accepted chunk ended early, or the chunks in between had no usable guess */
        FS_SERIAL(pos, s.start);
        if (rc != MK_OK) { fs_release(&f, s.buf); break; }
      }
      if (s.start != pos) { /* the guess was not a record boundary of the serial reader: drop the chunk's rows */
        stats.chunks_discarded++;
        fs_release(&f, s.buf);
        continue;
      }
      if (s.nrows) {
        uint64_t tok = 0;
        const double tp = fs_now();
        rc = sink->push(sink->ctx, f.bufs[s.buf], s.stride, s.nrows, ord, &tok);
        {
          const double d = fs_now() - tp;
          if (stats.t_push_call_s == 0) stats.t_first_push_call_s = d;
          stats.t_push_call_s += d;
          if (d > stats.t_push_call_max_s) stats.t_push_call_max_s = d;
        }
        if (!f.go) { /* the sink has taken its first buffer: whatever it waited for is there, the framers may run freely */
          pthread_mutex_lock(&f.mu);
          f.go = 1;
          pthread_cond_broadcast(&f.cv_buf);
          pthread_mutex_unlock(&f.mu);
        }
        if (rc != MK_OK) { fs_release(&f, s.buf); break; }
        ord += s.nrows; stats.rows += s.nrows; stats.records += s.nrec;
        if (!sink->wait) fs_release(&f, s.buf);
        else {
          if (nfifo == depth) { /* oldest push must be done before its buffer goes back to the framers */
            const double tw2 = fs_now();
            rc = sink->wait(sink->ctx, fifo[0].token);
            stats.t_wait_call_s += fs_now() - tw2;
            fs_release(&f, fifo[0].buf);
            memmove(fifo, fifo + 1, sizeof fifo[0] * (size_t)(--nfifo));
          }
          fifo[nfifo].token = tok; fifo[nfifo].buf = s.buf; nfifo++;
        }
        stats.t_push_s += fs_now() - tp;
      }
      pos = s.end;
      if (s.rc != MK_OK) rc = s.rc; /* a line beyond the reference's fgets() width: the file is outside the contract */
    }
    if (rc == MK_OK && pos < n) FS_SERIAL(pos, n);
#undef FS_SERIAL
  }

out:
  pthread_mutex_lock(&f.mu);
  f.stop = 1;
  pthread_cond_broadcast(&f.cv_buf);
  pthread_mutex_unlock(&f.mu);
  for (int t = 0; t < nth; t++) pthread_join(th[t], NULL);
  for (int i = 0; i < nfifo; i++) { /* the caller's buffers may go only when the copies out of them are done */
    const int wrc = sink->wait ? sink->wait(sink->ctx, fifo[i].token) : MK_OK;
    if (rc == MK_OK) rc = wrc;
  }
  if (pool) sink->release(sink->ctx, pool, pool_bytes);
  pthread_mutex_destroy(&f.mu);
  pthread_cond_destroy(&f.cv_buf);
  pthread_cond_destroy(&f.cv_ready);
out_nothreads:
  free(f.bufs); free(f.freelist); free(f.slots); free(th); free(serial_scratch);
  stats.threads = (uint32_t)nth;
  stats.t_total_s = fs_now() - t0;
  if (st) *st = stats;
  return rc;
}

/* ---- the stream bound to an engine: pinned buffers, asynchronous pushes --------------------------------------------- */
typedef struct { mk_engine *e; } fs_engine_ctx;

static int fs_eng_push(void *ctx, const uint8_t *rows, uint32_t stride, uint64_t nrows, uint64_t ord, uint64_t *token) {
  return mk_sketch_push_reads_async(((fs_engine_ctx *)ctx)->e, rows, stride, nrows, ord, token);
}
static int fs_eng_wait(void *ctx, uint64_t token) { return mk_sketch_push_wait(((fs_engine_ctx *)ctx)->e, token); }
static uint8_t *fs_eng_alloc(void *ctx, size_t bytes) {
  (void)ctx;
  void *p = NULL;
  return mk_host_arena_alloc(&p, bytes) == MK_OK ? (uint8_t *)p : NULL;
}
static void fs_eng_release(void *ctx, uint8_t *p, size_t bytes) { (void)ctx; mk_host_arena_free(p, bytes); }

int mk_sketch_push_fastq(mk_engine *e, const uint8_t *text, size_t n, const mk_fastq_opts *o, uint64_t first_ordinal,
                         mk_fastq_stats *st) {
  if (!e) return MK_ERR_ARG;
  fs_engine_ctx ctx = {e};
  mk_rows_sink sink = {&ctx, fs_eng_push, fs_eng_wait, fs_eng_alloc, fs_eng_release};
  return mk_fastq_stream(text, n, o, &sink, first_ordinal, st);
}
