/*
 * mk_frontend.c -- host front ends that turn input text into fixed-stride rows for the engine,
 * plus the synthetic-read generator (host C, no GPU).
 *
 * Replaces the reader halves of the reference's sketch functions:
 *   FASTQ  : the 4x fgets framing of mt_shortreads2koc(), iseq2comem.c:672-673
 *   FASTA  : the byte loop of fasta2co()/uniq_fasta2co(), iseq2comem.c:240-279 (line breaks skipped
 *            without resetting the k-mer window, '>' header lines skipped with a reset, any other
 *            non-ACGT byte resets)
 */
#define _GNU_SOURCE
#include "metakssd_hip.h"
#include "mk_host_internal.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

/* One row into a row buffer: `len` bytes of text, then zeros up to `stride`.  Row buffers are pinned memory that the GPU's
 * copy engine reads right after the framer is done with them; written with ordinary stores the lines sit dirty in this
 * core's cache and every DMA read has to be snooped out of it (measured: 29 GB/s instead of 55 GB/s host-to-device while
 * 16 threads frame).  Streaming stores put the rows into memory, skip the read-for-ownership and leave the cache to the
 * FASTQ text.  Needs a 16-byte aligned row and a stride that is a multiple of 16; anything else takes memcpy. */
static inline void mk_row_store(uint8_t *row, const uint8_t *src, size_t len, int newline, uint32_t stride) {
  /* newline != 0: a '\n' is appended behind the `len` bytes (len + 1 <= stride) */
#if defined(__SSE2__)
  if ((((uintptr_t)row | stride) & 15u) == 0) {
    size_t off = 0;
    for (; off + 16 <= len; off += 16) _mm_stream_si128((__m128i *)(row + off), _mm_loadu_si128((const __m128i *)(src + off)));
    if (off < len || newline) {
      uint8_t tmp[16] __attribute__((aligned(16))) = {0};
      memcpy(tmp, src + off, len - off);
      if (newline) tmp[len - off] = '\n';
      _mm_stream_si128((__m128i *)(row + off), _mm_load_si128((const __m128i *)tmp));
      off += 16;
    }
    const __m128i z = _mm_setzero_si128();
    for (; off < stride; off += 16) _mm_stream_si128((__m128i *)(row + off), z);
    return;
  }
#endif
  memcpy(row, src, len);
  if (newline) row[len++] = '\n';
  if (len < stride) memset(row + len, 0, stride - len);
}
static inline void mk_rows_done(void) { /* streaming stores are weakly ordered: make them visible before the buffer is handed on */
#if defined(__SSE2__)
  _mm_sfence();
#endif
}

/* ---- packed rows (MK_ROWS_PACKED, include/metakssd_hip.h): 2 bits a base + 1 validity bit a base, 64 bytes a read -------------
 * What mk_scan_kernel's decode() does per dword on the device -- code = (byte >> 1) & 3 (A0 C1 T2 G3), valid <=> the byte folded
 * to upper case is the letter of its code -- done here once, on the framer's thread, so that a 150-base read crosses PCIe as 64
 * bytes instead of 160.  Layout of a row (little-endian dwords):
 *   dword 0        : number of bases (bits 0..15) | flags (bit 16: every base valid)
 *   dwords 1..10   : the codes, eight bases (a "window") in 16 bits, first base in the top two bits; windows 2p and 2p+1 in the high
 *                    and low half of dword 1 + p -- the form the scan loop's funnel shift takes them in
 *   bytes 44..63   : one validity byte per window, bit j = base 8w + j is one of ACGTacgt (bits behind the last base: 0) */
#define MK_PACK_WINDOWS 19u /* 152 bases */

static void mk_pack_block_scalar(const uint8_t *src, uint32_t n, uint32_t *codes2, uint32_t *valid) {
  /* n <= 32 bases -> two code dwords (windows 0..3 of the block) and 32 validity bits */
  uint32_t c[2] = {0, 0}, v = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t b = src[i], code = (b >> 1) & 3u;
    const int ok = (b & 0xDFu) == (uint32_t)"ACTG"[code];
    if (ok) { v |= 1u << i; c[i >> 4] |= code << (30u - 2u * (i & 15u)); }
  }
  codes2[0] = c[0]; codes2[1] = c[1]; *valid = v;
}

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
/* the whole row in one AVX2 function: five blocks of 32 bases written out, loads straight from the text where 32 bytes are readable
 * (`avail`: bytes that may be read from src on; the FASTQ text behind a sequence line is its '+' and quality lines), bases behind the
 * read masked by a sliding window over 32 x 0xFF, 32 x 0x00 */
static const uint8_t mk_pack_lanes[64] __attribute__((aligned(64))) = {
    255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
__attribute__((target("avx2"))) static void mk_row_pack_avx2(uint8_t *row, const uint8_t *src, size_t nb, size_t avail) {
  uint32_t out[16] __attribute__((aligned(16)));
  memset(out, 0, sizeof out);
  uint8_t *vb = (uint8_t *)out + 44;
  const __m256i three = _mm256_set1_epi8(3), fold = _mm256_set1_epi8((char)0xDF), ones = _mm256_set1_epi16(1);
  const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i w = _mm256_setr_epi8(64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1);
  const __m256i gather = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  uint32_t allvalid = 1;
  for (size_t at = 0; at < nb; at += 32u) {
    const uint32_t n = nb - at < 32u ? (uint32_t)(nb - at) : 32u;
    __m256i b;
    if (at + 32u <= avail) b = _mm256_loadu_si256((const __m256i *)(src + at));
    else { uint8_t tmp[32] __attribute__((aligned(32))) = {0}; memcpy(tmp, src + at, n); b = _mm256_load_si256((const __m256i *)tmp); }
    const __m256i codes = _mm256_and_si256(_mm256_srli_epi16(b, 1), three);
    __m256i ok = _mm256_cmpeq_epi8(_mm256_and_si256(b, fold), _mm256_shuffle_epi8(lut, codes));
    ok = _mm256_and_si256(ok, _mm256_loadu_si256((const __m256i *)(mk_pack_lanes + 32u - n))); /* bytes behind the read do not count */
    const uint32_t v = (uint32_t)_mm256_movemask_epi8(ok);
    const __m256i q = _mm256_madd_epi16(_mm256_maddubs_epi16(_mm256_and_si256(codes, ok), w), ones);
    const __m256i g = _mm256_shuffle_epi8(q, gather);
    out[1 + at / 16u] = (uint32_t)_mm256_extract_epi32(g, 0);
    if (at + 16u < nb) out[2 + at / 16u] = (uint32_t)_mm256_extract_epi32(g, 4);
    memcpy(vb + at / 8u, &v, (n + 7u) / 8u);
    if (v != (n == 32u ? 0xFFFFFFFFu : (1u << n) - 1u)) allvalid = 0;
  }
  out[0] = (uint32_t)nb | (allvalid << 16);
  if (((uintptr_t)row & 15u) == 0)
    for (int i = 0; i < 4; i++) _mm_stream_si128((__m128i *)(row + 16 * i), _mm_load_si128((const __m128i *)((const uint8_t *)out + 16 * i)));
  else memcpy(row, out, 64);
}
static int mk_have_avx2(void) {
  static int have = -1;
  if (have < 0) have = __builtin_cpu_supports("avx2") && !getenv("MK_NO_AVX2") ? 1 : 0; /* (MK_NO_AVX2: the tests run the scalar form too) */
  return have;
}
#else
static int mk_have_avx2(void) { return 0; }
#endif

/* nb <= 152 bases at src -> the 64-byte packed row (16-byte aligned: streaming stores, see mk_row_store); avail >= nb: bytes that
 * may be read from src on */
static inline void mk_row_pack(uint8_t *row, const uint8_t *src, size_t nb, size_t avail) {
#if defined(__x86_64__) && defined(__GNUC__)
  if (mk_have_avx2()) { mk_row_pack_avx2(row, src, nb, avail); return; }
#endif
  (void)avail;
  uint32_t out[16] __attribute__((aligned(16)));
  memset(out, 0, sizeof out);
  uint8_t *vb = (uint8_t *)out + 44;
  uint32_t allvalid = 1;
  for (uint32_t at = 0; at < nb; at += 32u) {
    const uint32_t n = nb - at < 32u ? (uint32_t)(nb - at) : 32u;
    uint32_t c2[2], v;
      mk_pack_block_scalar(src + at, n, c2, &v);
    out[1 + at / 16u] = c2[0];
    if (at + 16u < nb) out[2 + at / 16u] = c2[1];
    memcpy(vb + at / 8u, &v, (n + 7u) / 8u);
    if (v != (n == 32u ? 0xFFFFFFFFu : (1u << n) - 1u)) allvalid = 0;
  }
  out[0] = (uint32_t)nb | (allvalid << 16);
#if defined(__SSE2__)
  if (((uintptr_t)row & 15u) == 0) {
    for (int i = 0; i < 4; i++) _mm_stream_si128((__m128i *)(row + 16 * i), _mm_load_si128((const __m128i *)((const uint8_t *)out + 16 * i)));
    return;
  }
#endif
  memcpy(row, out, 64);
}

/* ---- WIDE packed rows (MK_ROWS_WIDE): 240 bases in the same 64 bytes -- the codes take dwords 1..15, and the validity bits, which
 * nearly no row of a genome needs, live in an EXTENSION ROW behind the rows that do: dword 0 = bases | every base valid << 16 | an
 * extension row follows << 17; the extension row has dword 0 = 1 << 18 (zero bases: as a row it is empty) and the validity bytes of
 * the row in front of it at bytes 16..45.  Made by mk_fasta_pack_rows, taken by mk_sketch_batch_begin_rows only. */
#if defined(__x86_64__) && defined(__GNUC__)
/* the whole row in one AVX2 function, as mk_row_pack_avx2: eight blocks of 32 bases, row and extension row built in locals and
 * written out with streaming stores; returns the rows written */
__attribute__((target("avx2"))) static unsigned mk_wide_pack_avx2(uint8_t *row, const uint8_t *src, size_t nb, size_t avail) {
  uint32_t out[32] __attribute__((aligned(32)));
  memset(out, 0, sizeof out);
  uint32_t vv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const __m256i three = _mm256_set1_epi8(3), fold = _mm256_set1_epi8((char)0xDF), ones = _mm256_set1_epi16(1);
  const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i w = _mm256_setr_epi8(64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1, 64, 16, 4, 1);
  const __m256i gather = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  uint32_t bad = 0;
  for (size_t at = 0, k = 0; at < nb; at += 32u, k++) {
    const uint32_t n = nb - at < 32u ? (uint32_t)(nb - at) : 32u;
    __m256i b;
    if (at + 32u <= avail) b = _mm256_loadu_si256((const __m256i *)(src + at));
    else { uint8_t tmp[32] __attribute__((aligned(32))) = {0}; memcpy(tmp, src + at, n); b = _mm256_load_si256((const __m256i *)tmp); }
    const __m256i codes = _mm256_and_si256(_mm256_srli_epi16(b, 1), three);
    __m256i ok = _mm256_cmpeq_epi8(_mm256_and_si256(b, fold), _mm256_shuffle_epi8(lut, codes));
    ok = _mm256_and_si256(ok, _mm256_loadu_si256((const __m256i *)(mk_pack_lanes + 32u - n))); /* bytes behind the row do not count */
    const uint32_t v = (uint32_t)_mm256_movemask_epi8(ok);
    const __m256i q = _mm256_madd_epi16(_mm256_maddubs_epi16(_mm256_and_si256(codes, ok), w), ones);
    const __m256i g = _mm256_shuffle_epi8(q, gather);
    out[1 + 2 * k] = (uint32_t)_mm256_extract_epi32(g, 0);
    if (2 * k + 2 < 16) out[2 + 2 * k] = (uint32_t)_mm256_extract_epi32(g, 4); /* (bases 240..255 do not exist) */
    vv[k] = v;
    bad |= v ^ (n == 32u ? 0xFFFFFFFFu : (1u << n) - 1u);
  }
  const uint32_t allvalid = bad == 0u;
  out[0] = (uint32_t)nb | (allvalid << 16) | ((allvalid ^ 1u) << 17);
  /* ordinary stores: with streaming stores this ran 3 x slower on an EPYC 9575F (115 ns a row against 36: the three store forms in
   * profiles/r04_probe_wide_pack_stores.txt) */
  _mm256_store_si256((__m256i *)row, _mm256_load_si256((const __m256i *)out));
  _mm256_store_si256((__m256i *)(row + 32), _mm256_load_si256((const __m256i *)(out + 8)));
  if (allvalid) return 1u;
  /* the extension row: dword 0 = 1 << 18, the validity bytes at 16..45 */
  _mm256_store_si256((__m256i *)(row + 64), _mm256_setr_epi32(1 << 18, 0, 0, 0, (int)vv[0], (int)vv[1], (int)vv[2], (int)vv[3]));
  _mm256_store_si256((__m256i *)(row + 96), _mm256_setr_epi32((int)vv[4], (int)vv[5], (int)vv[6], (int)vv[7], 0, 0, 0, 0));
  return 2u;
}
#endif
/* nb <= 240 bases at src -> one wide row, or a wide row and its extension row; returns the rows written (row: 32-byte aligned for the
 * vector form's stores, else the scalar form) */
static inline unsigned mk_wide_pack(uint8_t *row, const uint8_t *src, size_t nb, size_t avail) {
#if defined(__x86_64__) && defined(__GNUC__)
  if (mk_have_avx2() && ((uintptr_t)row & 31u) == 0) return mk_wide_pack_avx2(row, src, nb, avail);
#endif
  (void)avail;
  uint32_t out[32] __attribute__((aligned(16)));
  memset(out, 0, sizeof out);
  unsigned allvalid = 1;
  uint8_t *vb = (uint8_t *)out + 64 + 16;
  for (size_t at = 0; at < nb; at += 32u) {
    const uint32_t n = nb - at < 32u ? (uint32_t)(nb - at) : 32u;
    uint32_t c2[2], v;
    mk_pack_block_scalar(src + at, n, c2, &v);
    out[1 + at / 16u] = c2[0];
    if (at + 16u < nb) out[2 + at / 16u] = c2[1];
    memcpy(vb + at / 8u, &v, (n + 7u) / 8u);
    if (v != (n == 32u ? 0xFFFFFFFFu : (1u << n) - 1u)) allvalid = 0;
  }
  out[0] = (uint32_t)nb | (allvalid << 16) | ((allvalid ^ 1u) << 17);
  out[16] = 1u << 18;
  memcpy(row, out, allvalid ? 64u : 128u);
  return allvalid ? 1u : 2u;
}

int mk_params_packed_ok(const mk_params *p) {
  /* the geometries with a tuned scan kernel: their loop takes eight bases in 16 bits (mk_engine.hip) */
  return p && ((p->subk == 6 && p->k >= 9 && p->k <= 11) || (p->subk == 5 && p->k == 11));
}

int mk_pack_rows_host(const uint8_t *rows, uint32_t stride, uint64_t nrows, uint8_t *packed) {
  if ((!rows && nrows) || (!packed && nrows) || stride < 4 || stride > 4096) return MK_ERR_ARG;
  for (uint64_t r = 0; r < nrows; r++) {
    const uint8_t *row = rows + r * (uint64_t)stride;
    const uint8_t *nl = (const uint8_t *)memchr(row, '\n', stride);
    const size_t nb = nl ? (size_t)(nl - row) : stride;
    if (nb > MK_PACKED_MAX_BASES) return MK_ERR_ARG;
    mk_row_pack(packed + r * (uint64_t)MK_PACKED_PITCH, row, nb, (size_t)stride);
  }
  mk_rows_done();
  return MK_OK;
}

#define MK_FQ_LEN 4096 /* iseq2comem.c:656: fgets() never returns more than FQ_LEN-1 characters */

int mk_synth_rows_host(uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len, uint32_t stride, uint8_t *rows) {
  if (!rows || stride < len + 1) return MK_ERR_ARG;
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  for (uint64_t r = 0; r < nreads; r++) {
    uint8_t *row = rows + r * (uint64_t)stride;
    uint64_t w = 0;
    for (uint32_t b = 0; b < len; b++) {
      if ((b & 31) == 0) w = mk_synth_word(seed, first_read + r, b >> 5);
      row[b] = (uint8_t)acgt[(w >> (2 * (b & 31))) & 3];
    }
    row[len] = '\n';
    memset(row + len + 1, 0, stride - len - 1);
  }
  return MK_OK;
}

int mk_synth_fastq_write(const char *path, uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len) {
  if (!path || len == 0 || len > 4094) return MK_ERR_ARG;
  FILE *f = fopen(path, "wb");
  if (!f) return MK_ERR_IO;
  uint8_t *row = (uint8_t *)malloc(len + 8);
  char *qual = (char *)malloc(len + 2);
  if (!row || !qual) { fclose(f); free(row); free(qual); return MK_ERR_NOMEM; }
  memset(qual, 'I', len);
  qual[len] = '\n';
  int ok = 1;
  for (uint64_t r = 0; r < nreads && ok; r++) {
    mk_synth_rows_host(seed, first_read + r, 1, len, len + 1, row);
    ok = fprintf(f, "@r%llu\n", (unsigned long long)(first_read + r)) > 0 && fwrite(row, 1, len + 1, f) == len + 1 &&
         fwrite("+\n", 1, 2, f) == 2 && fwrite(qual, 1, len + 1, f) == len + 1;
  }
  free(row);
  free(qual);
  if (fclose(f) != 0) ok = 0;
  return ok ? MK_OK : MK_ERR_IO;
}

/* ---- the same file from several threads: record i is "@r<i>\n" + len + 1 + 2 + len + 1 bytes, so the offset of every
 * record follows from the number of decimal digits of the read numbers in front of it */
static uint64_t mk_digits_sum(uint64_t a, uint64_t b) { /* sum of the decimal digit counts of a .. b-1 */
  uint64_t sum = 0, lo = 1, d = 1;
  if (a == 0 && b > 0) { sum += 1; a = 1; } /* "0" has one digit */
  while (a < b) {
    const uint64_t hi = lo > UINT64_MAX / 10 ? UINT64_MAX : lo * 10; /* numbers with d digits: [lo, hi) */
    if (a < hi) {
      const uint64_t e = b < hi ? b : hi;
      sum += (e - a) * d;
      a = e;
    }
    lo = hi; d++;
  }
  return sum;
}

typedef struct {
  int fd;
  uint64_t seed, first, lo, hi, off; /* reads [lo, hi) of the file, file offset of read lo */
  uint32_t len;
  int rc;
} mk_fqw_job;

static void *mk_fqw_run(void *arg) {
  mk_fqw_job *j = arg;
  const size_t cap = (size_t)4 << 20, rec_max = 2 * (size_t)j->len + 32;
  uint8_t *b = (uint8_t *)malloc(cap + rec_max);
  if (!b) { j->rc = MK_ERR_NOMEM; return NULL; }
  size_t fill = 0;
  uint64_t off = j->off;
  for (uint64_t r = j->lo; r <= j->hi; r++) {
    if (r == j->hi || fill >= cap) {
      size_t done = 0;
      while (done < fill) {
        ssize_t w = pwrite(j->fd, b + done, fill - done, (off_t)(off + done));
        if (w <= 0) { j->rc = MK_ERR_IO; free(b); return NULL; }
        done += (size_t)w;
      }
      off += fill;
      fill = 0;
      if (r == j->hi) break;
    }
    fill += (size_t)sprintf((char *)b + fill, "@r%llu\n", (unsigned long long)(j->first + r));
    mk_synth_rows_host(j->seed, j->first + r, 1, j->len, j->len + 1, b + fill);
    fill += j->len + 1;
    b[fill++] = '+'; b[fill++] = '\n';
    memset(b + fill, 'I', j->len);
    fill += j->len;
    b[fill++] = '\n';
  }
  free(b);
  return NULL;
}

int mk_synth_fastq_write_mt(const char *path, uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len, int nthreads) {
  if (!path || len == 0 || len > 4094) return MK_ERR_ARG;
  enum { MAXT = 256 };
  int T = nthreads < 1 ? 1 : nthreads > MAXT ? MAXT : nthreads;
  if ((uint64_t)T > nreads) T = nreads ? (int)nreads : 1;
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return MK_ERR_IO;
  const uint64_t fixed = 2ull * len + 7; /* "@r" "\n" seq "\n" "+\n" qual "\n" without the digits */
  const uint64_t total = nreads * fixed + mk_digits_sum(first_read, first_read + nreads);
  if (ftruncate(fd, (off_t)total) != 0) { close(fd); return MK_ERR_IO; }
  mk_fqw_job job[MAXT];
  pthread_t th[MAXT];
  for (int t = 0; t < T; t++) {
    mk_fqw_job *j = &job[t];
    j->fd = fd; j->seed = seed; j->first = first_read; j->len = len; j->rc = MK_OK;
    j->lo = nreads / (uint64_t)T * (uint64_t)t;
    j->hi = t + 1 == T ? nreads : nreads / (uint64_t)T * (uint64_t)(t + 1);
    j->off = j->lo * fixed + mk_digits_sum(first_read, first_read + j->lo);
  }
  int started = 0;
  for (int t = 1; t < T; t++) { if (pthread_create(&th[t], NULL, mk_fqw_run, &job[t]) != 0) break; started = t; }
  int rc = MK_OK;
  for (int t = started + 1; t < T; t++) mk_fqw_run(&job[t]); /* threads that could not be started: their share here */
  mk_fqw_run(&job[0]);
  for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
  for (int t = 0; t < T; t++) if (job[t].rc != MK_OK) rc = job[t].rc;
  if (close(fd) != 0) rc = MK_ERR_IO;
  return rc;
}

/* one text line starting at p: returns its length including the '\n' (or up to `end` when the file ends
 * without one, only if final); 0 = incomplete line, need more data */
/* length of the line at p including its '\n' (0: no '\n' in front of `end` and not final; final: the rest).  The lines of a FASTQ record
 * are short (a header, 150 bases, "+", 150 quality bytes): four memchr() calls a record cost more in call overhead than in bytes,
 * so where the CPU has AVX2 the first '\n' is looked for 32 bytes at a time in place */
static size_t mk_line_generic(const uint8_t *p, const uint8_t *end, int final) {
  const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
  if (nl) return (size_t)(nl - p) + 1;
  return final ? (size_t)(end - p) : 0;
}
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target("avx2"))) static size_t mk_line_avx2(const uint8_t *p, const uint8_t *end, int final) {
  const uint8_t *q = p;
  const __m256i nl = _mm256_set1_epi8('\n');
  while (end - q >= 32) {
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)q), nl));
    if (m) return (size_t)(q - p) + (size_t)__builtin_ctz(m) + 1u;
    q += 32;
  }
  const uint8_t *x = (const uint8_t *)memchr(q, '\n', (size_t)(end - q));
  if (x) return (size_t)(x - p) + 1;
  return final ? (size_t)(end - p) : 0;
}
#endif
static inline size_t mk_line(const uint8_t *p, const uint8_t *end, int final) {
#if defined(__x86_64__) && defined(__GNUC__)
  if (mk_have_avx2()) return mk_line_avx2(p, end, final);
#endif
  return mk_line_generic(p, end, final);
}

/* the framer proper: records that START in front of buf + stop (stop <= n; a record may end behind it).  *need = bytes a
 * row must hold when MK_ERR_ARG reports a sequence line longer than the stride. */
int mk_fastq_frame_range(const uint8_t *buf, size_t n, size_t stop, int final, uint8_t *rows, uint32_t stride, uint64_t max_rows,
                         uint64_t *nrows, size_t *consumed, uint32_t *need) {
  const int packed = (stride & MK_ROWS_PACKED) != 0; /* 64-byte packed rows: a sequence line of more than 152 bases is MK_ERR_ARG */
  if (packed) stride &= ~MK_ROWS_PACKED;
  if ((!buf && n) || !rows || !nrows || !consumed || stride < 4 || stride > 4096 || (stride & 3) || stop > n) return MK_ERR_ARG;
  if (packed && stride != MK_PACKED_PITCH) return MK_ERR_ARG;
  const uint8_t *p = buf, *end = buf + n, *stop_at = buf + stop;
  uint64_t r = 0;
  int rc = MK_OK;
  if (need) *need = 0;
  while (r < max_rows && p < stop_at) {
    /* four lines per record; a record missing any of them is dropped (the && chain at iseq2comem.c:673) */
    size_t l1 = mk_line(p, end, final);
    if (!l1) break;
    const uint8_t *s = p + l1;
    size_t l2 = s < end ? mk_line(s, end, final) : 0;
    if (!l2) { if (final) p = end; break; }
    const uint8_t *q = s + l2;
    size_t l3 = q < end ? mk_line(q, end, final) : 0;
    if (!l3) { if (final) p = end; break; }
    const uint8_t *t = q + l3;
    size_t l4 = t < end ? mk_line(t, end, final) : 0;
    if (!l4) { if (final) p = end; break; }
    /* the reference frames with fgets(…,4096,…): a line of 4095+ characters is split and the record
     * structure is lost (SURVEY.md 8a note 7).  That input is outside the contract: refuse it. */
    if (l1 >= MK_FQ_LEN || l2 >= MK_FQ_LEN || l3 >= MK_FQ_LEN || l4 >= MK_FQ_LEN) { rc = MK_ERR_FORMAT; break; }
    if (s[l2 - 1] != '\n') {
      /* sequence line is the unterminated last line of the file: then lines 3 and 4 cannot exist */
      p = end;
      break;
    }
    if (packed ? l2 - 1 > MK_PACKED_MAX_BASES : l2 > stride) { rc = MK_ERR_ARG; if (need) *need = (uint32_t)l2; break; } /* caller must re-frame from here with a larger stride */
    if (packed) mk_row_pack(rows + r * (uint64_t)stride, s, l2 - 1, (size_t)(end - s));
    else mk_row_store(rows + r * (uint64_t)stride, s, l2, 0, stride);
    r++;
    p = t + l4;
  }
  mk_rows_done();
  *nrows = r;
  *consumed = (size_t)(p - buf);
  return rc;
}

int mk_fastq_frame(const uint8_t *buf, size_t n, int final, uint8_t *rows, uint32_t stride, uint64_t max_rows,
                   uint64_t *nrows, size_t *consumed) {
  return mk_fastq_frame_range(buf, n, n, final, rows, stride, max_rows, nrows, consumed, NULL);
}

#define MK_FQCO_LEN 20000 /* iseq2comem.c:319: fastq2co()'s fgets() width */

/* rows one sequence payload of L bytes needs at this stride (windows overlap by TL-1 bases) */
static uint64_t mk_rows_for(size_t L, uint32_t cap, uint32_t TL) {
  if (L <= cap) return 1;
  const size_t step = cap - (TL - 1);
  return 1 + (L - cap + step - 1) / step;
}

int mk_fastq_frame_q(const uint8_t *buf, size_t n, int final, int32_t qmin, int32_t TL, uint64_t records_before,
                     uint8_t *rows, uint32_t stride, uint64_t max_rows, uint64_t *nrows, uint64_t *nrecords,
                     size_t *consumed) {
  return mk_fastq_frame_q_range(buf, n, n, final, qmin, TL, records_before, rows, stride, max_rows, nrows, nrecords, consumed, NULL);
}

/* records that START in front of buf + stop; *need as in mk_fastq_frame_range */
int mk_fastq_frame_q_range(const uint8_t *buf, size_t n, size_t stop, int final, int32_t qmin, int32_t TL, uint64_t records_before,
                           uint8_t *rows, uint32_t stride, uint64_t max_rows, uint64_t *nrows, uint64_t *nrecords,
                           size_t *consumed, uint32_t *need_stride) {
  const int packed = (stride & MK_ROWS_PACKED) != 0; /* 64-byte packed rows: a read of more than 152 bases is MK_ERR_ARG (no windows) */
  if (packed) stride &= ~MK_ROWS_PACKED;
  if ((!buf && n) || !rows || !nrows || !nrecords || !consumed || stride > 4096 || (stride & 3) || TL < 2 || TL > 32 ||
      stride < 2u * (uint32_t)TL + 4 || stop > n)
    return MK_ERR_ARG;
  if (packed && stride != MK_PACKED_PITCH) return MK_ERR_ARG;
  const uint8_t *p = buf, *end = buf + n, *stop_at = buf + stop;
  const uint32_t cap = packed ? MK_PACKED_MAX_BASES : stride - 1;
  uint64_t r = 0, rec = 0;
  int rc = MK_OK;
  if (need_stride) *need_stride = 0;
  while (p < stop_at) {
    const uint8_t *ln[4];
    size_t len[4] = {0, 0, 0, 0};
    const uint8_t *q = p;
    int have = 0;
    for (; have < 4 && q < end; have++) {
      len[have] = mk_line(q, end, final);
      if (!len[have]) break;
      ln[have] = q;
      q += len[have];
    }
    if (have < 4 && !final) break; /* record still arriving */
    /* fastq2co() reads the NEXT record at the end of the current one and walks it only if those four fgets()
     * did not touch end-of-file (iseq2comem.c:351-362): a record with a missing line, or whose last line has
     * no '\n', is read but never walked.  The first record is read before the loop and always walked (:343-349). */
    const int whole = have == 4 && ln[3][len[3] - 1] == '\n';
    if (!whole && !(records_before + rec == 0 && have >= 2)) { p = end; break; }
    for (int i = 0; i < have; i++)
      if (len[i] >= MK_FQCO_LEN - 1) rc = MK_ERR_FORMAT; /* fgets(…,LEN,…) would split the line */
    if (rc) break;
    size_t L = len[1];
    if (L && ln[1][L - 1] == '\n') L--;
    if (L > cap && (stride < 4096 || packed)) { rc = MK_ERR_ARG; if (need_stride) *need_stride = (uint32_t)(L + 1 > 4096 ? 4096 : L + 1); break; } /* caller re-frames from here with wider rows */
    const uint64_t need = mk_rows_for(L, cap, (uint32_t)TL);
    if (r + need > max_rows) {
      if (r == 0) rc = MK_ERR_ARG;
      break;
    }
    const uint8_t *sq = ln[1], *ql = have == 4 ? ln[3] : NULL;
    const size_t qn = have == 4 ? len[3] : 0;
    size_t at = 0;
    for (uint64_t w = 0; w < need; w++) {
      uint8_t *row = rows + (r + w) * (uint64_t)stride;
      size_t m = L - at < cap ? L - at : cap;
      if (qmin <= -128) { /* every signed quality byte passes */
        if (packed) mk_row_pack(row, sq + at, m, (size_t)(end - (sq + at)));
        else mk_row_store(row, sq + at, m, 1, stride);
      }
      else { /* a base whose quality byte is below -Q resets the window exactly like a non-ACGT byte (:367-379) */
        uint8_t masked[4096];
        for (size_t i = 0; i < m; i++) {
          const int qv = at + i < qn ? (int)(signed char)ql[at + i] : 0;
          masked[i] = qv >= qmin ? sq[at + i] : (uint8_t)'N';
        }
        if (packed) mk_row_pack(row, masked, m, sizeof masked);
        else mk_row_store(row, masked, m, 1, stride);
      }
      at += m;
      if (w + 1 < need) at -= (size_t)(TL - 1);
    }
    r += need;
    rec++;
    p = q;
  }
  mk_rows_done();
  *nrows = r;
  *nrecords = rec;
  *consumed = (size_t)(p - buf);
  return rc;
}

/* ---- the same two framers on several threads -----------------------------------------------------------------------
 * Records are groups of four lines counted from the start of the buffer (which the caller keeps at a record boundary),
 * whatever the lines contain -- exactly what the serial framers (and the reference's 4x fgets) do.  So: count the
 * newlines of T slices in parallel, prefix the counts, cut the buffer at the line starts whose index is a multiple of
 * four, and let every thread run the SERIAL framer over its own whole records, writing at the row index its first
 * record has.  The tail behind the last complete record goes through the serial framer with the caller's `final`. */
typedef struct {
  const uint8_t *buf;
  size_t lo, hi;        /* phase 1: slice; phase 2: whole records [lo, hi) */
  uint64_t lines;       /* phase 1 out */
  int occ, qmin, TL;    /* phase 2 in */
  uint64_t records_before;
  uint8_t *rows;
  uint32_t stride;
  uint64_t want_rows;   /* records in [lo, hi) */
  int rc;
  uint64_t nrows, nrec;
  size_t used;
} mk_frame_job;

static void *mk_count_lines(void *arg) {
  mk_frame_job *j = arg;
  uint64_t c = 0;
  const uint8_t *p = j->buf + j->lo, *end = j->buf + j->hi;
  while (p < end) {
    const uint8_t *nl = memchr(p, '\n', (size_t)(end - p));
    if (!nl) break;
    c++;
    p = nl + 1;
  }
  j->lines = c;
  return NULL;
}

static void *mk_frame_slice(void *arg) {
  mk_frame_job *j = arg;
  if (j->occ)
    j->rc = mk_fastq_frame_q(j->buf + j->lo, j->hi - j->lo, 0, j->qmin, j->TL, j->records_before, j->rows, j->stride, j->want_rows,
                             &j->nrows, &j->nrec, &j->used);
  else {
    j->rc = mk_fastq_frame(j->buf + j->lo, j->hi - j->lo, 0, j->rows, j->stride, j->want_rows, &j->nrows, &j->used);
    j->nrec = j->nrows;
  }
  return NULL;
}

/* byte offset of the start of line number `k` (0-based, counted from `from`, which is a line start) */
static size_t mk_skip_lines(const uint8_t *buf, size_t from, size_t n, uint64_t k) {
  size_t at = from;
  while (k--) {
    const uint8_t *nl = memchr(buf + at, '\n', n - at);
    if (!nl) return n;
    at = (size_t)(nl - buf) + 1;
  }
  return at;
}

int mk_fastq_frame_mt(const uint8_t *buf, size_t n, int final, int occ, int32_t qmin, int32_t TL, uint64_t records_before,
                      uint8_t *rows, uint32_t stride, uint64_t max_rows, int nthreads, uint64_t *nrows, uint64_t *nrecords,
                      size_t *consumed) {
  if ((!buf && n) || !rows || !nrows || !nrecords || !consumed) return MK_ERR_ARG;
  enum { MAXT = 64 };
  int T = nthreads < 1 ? 1 : nthreads > MAXT ? MAXT : nthreads;
  if (n < ((size_t)1 << 20)) T = 1; /* not worth the threads */
  *nrows = *nrecords = 0;
  *consumed = 0;
  uint64_t done_rows = 0, done_rec = 0;
  size_t at = 0;
  if (T > 1) {
    mk_frame_job job[MAXT];
    pthread_t th[MAXT];
    memset(job, 0, sizeof job);
    for (int t = 0; t < T; t++) { job[t].buf = buf; job[t].lo = n / (size_t)T * (size_t)t; job[t].hi = t + 1 == T ? n : n / (size_t)T * (size_t)(t + 1); }
    for (int t = 1; t < T; t++) pthread_create(&th[t], NULL, mk_count_lines, &job[t]);
    mk_count_lines(&job[0]);
    for (int t = 1; t < T; t++) pthread_join(th[t], NULL);
    /* cut points: for every slice the first record boundary (line start whose index is a multiple of four) at or behind
     * its first byte; they come out in non-decreasing order, duplicates are dropped */
    size_t cut[MAXT + 1];
    uint64_t cutline[MAXT + 1];
    uint64_t before = 0; /* newlines in front of slice t */
    int nc = 0;
    for (int t = 0; t < T; t++) {
      size_t ls = job[t].lo;
      uint64_t idx = before;
      before += job[t].lines;
      if (t > 0 && buf[ls - 1] != '\n') { /* slice starts inside a line: the next line start */
        const uint8_t *nl = memchr(buf + ls, '\n', n - ls);
        if (!nl) continue;
        ls = (size_t)(nl - buf) + 1;
        idx++;
      }
      const uint64_t skip = (4u - (idx & 3u)) & 3u;
      const size_t c = mk_skip_lines(buf, ls, n, skip);
      if (c >= n) continue;
      if (nc && c <= cut[nc - 1]) continue;
      cut[nc] = c; cutline[nc] = idx + skip; nc++;
    }
    const uint64_t total_lines = before;
    uint64_t R = total_lines / 4u; /* complete records in the buffer */
    if (R > max_rows) R = max_rows;
    /* a read longer than the row is cut into several rows in the occ flavour at the maximal stride: rows != records there,
     * so such buffers stay serial */
    if (nc >= 2 && cut[0] == 0 && R >= (uint64_t)nc && !(occ && stride == 4096u)) {
      int last = nc - 1;
      while (last > 0 && cutline[last] >= 4u * R) last--;
      const size_t endcut = mk_skip_lines(buf, cut[last], n, 4u * R - cutline[last]); /* start of line 4R */
      int J = 0;
      for (int i = 0; i <= last; i++) {
        const size_t hi = i == last ? endcut : cut[i + 1];
        const uint64_t hiline = i == last ? 4u * R : cutline[i + 1];
        mk_frame_job *j = &job[J++];
        memset(j, 0, sizeof *j);
        j->buf = buf; j->lo = cut[i]; j->hi = hi; j->occ = occ; j->qmin = qmin; j->TL = TL;
        j->records_before = records_before + cutline[i] / 4u;
        j->rows = rows + (cutline[i] / 4u) * (uint64_t)stride; j->stride = stride;
        j->want_rows = (hiline - cutline[i]) / 4u;
      }
      for (int t = 1; t < J; t++) pthread_create(&th[t], NULL, mk_frame_slice, &job[t]);
      mk_frame_slice(&job[0]);
      for (int t = 1; t < J; t++) pthread_join(th[t], NULL);
      for (int t = 0; t < J; t++) {
        if (job[t].rc != MK_OK) return job[t].rc; /* MK_ERR_ARG: widen the rows and call again; nothing reported as consumed */
        if (job[t].nrows != job[t].want_rows || job[t].used != job[t].hi - job[t].lo) return MK_ERR_FORMAT; /* cannot happen */
      }
      done_rows = done_rec = R;
      at = endcut;
    }
  }
  /* the rest (everything when not split): serial, with the caller's `final` */
  if (at < n || n == 0) {
    uint64_t r = 0, rec = 0;
    size_t used = 0;
    int rc;
    if (occ) rc = mk_fastq_frame_q(buf + at, n - at, final, qmin, TL, records_before + done_rec, rows + done_rows * (uint64_t)stride, stride,
                                   max_rows - done_rows, &r, &rec, &used);
    else { rc = mk_fastq_frame(buf + at, n - at, final, rows + done_rows * (uint64_t)stride, stride, max_rows - done_rows, &r, &used); rec = r; }
    if (rc != MK_OK && done_rows == 0) { *nrows = r; *nrecords = rec; *consumed = used; return rc; }
    if (rc == MK_OK || r) { done_rows += r; done_rec += rec; at += used; }
    if (rc != MK_OK && rc != MK_ERR_ARG) return rc;
    /* MK_ERR_ARG after some parallel progress: report the progress; the caller comes back with the remainder */
  }
  *nrows = done_rows;
  *nrecords = done_rec;
  *consumed = at;
  return MK_OK;
}

/* ---- a whole FASTA file -> packed rows (MK_ROWS_PACKED), on the reader's thread -------------------------------------------------
 * The same base stream as mk_fasta_window / the device's mk_fa_* kernels make (kept: every byte that is neither '\n' nor '\r' nor
 * inside a '>' line; a '>' line leaves its '>' as ONE byte, which is no base and so resets the window: iseq2comem.c:240-279), cut
 * into overlapping rows of 152 (MK_ROWS_PACKED) or 240 (MK_ROWS_WIDE) stream bytes at a distance of that + 1 - TL: every TL-byte
 * window of the stream starts in exactly one row, in file order.  Two vector passes over pieces that stay in the cache: (1) text -> stream, 32 bytes a step up to the next
 * '\n' / '\r' / '>'; (2) stream -> rows with the FASTQ framers' packer.  A genome of 4 Mbases crosses PCIe as 1.9 MB
 * (wide rows: 1.2 MB) this way instead of 4.06 MB of text. */
uint64_t mk_fasta_pack_bound(size_t n, int32_t TL, uint32_t format) {
  if (TL < 2 || TL > 32 || (format != MK_ROWS_PACKED && format != MK_ROWS_WIDE)) return 0;
  const size_t cap = format == MK_ROWS_WIDE ? MK_WIDE_MAX_BASES : MK_PACKED_MAX_BASES, step = cap + 1u - (size_t)TL;
  const uint64_t rows = n < (size_t)TL ? 0 : (n - (size_t)TL) / step + 1u; /* the stream is no longer than the text */
  return format == MK_ROWS_WIDE ? 2u * rows : rows;                      /* (every wide row may have an extension row) */
}

/* stream bytes of text[0, n) to out (room: n + 32 bytes); *hdr: inside a '>' line (in: at text[0]; out: behind text[n-1]) */
static size_t mk_fasta_keep_scalar(const uint8_t *p, size_t n, uint8_t *out, int *hdr) {
  size_t o = 0;
  int h = *hdr;
  for (size_t i = 0; i < n; i++) {
    const uint8_t ch = p[i];
    if (h) { if (ch == '\n') h = 0; continue; }
    if (ch == '\n' || ch == '\r') continue;
    if (ch == '>') h = 1;
    out[o++] = ch;
  }
  *hdr = h;
  return o;
}
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target("avx2"))) static size_t mk_fasta_keep_avx2(const uint8_t *p, size_t n, uint8_t *out, int *hdr) {
  const __m256i vnl = _mm256_set1_epi8('\n'), vcr = _mm256_set1_epi8('\r'), vgt = _mm256_set1_epi8('>');
  size_t i = 0, o = 0;
  int h = *hdr;
  while (i < n) {
    if (h) {
      const uint8_t *nl = (const uint8_t *)memchr(p + i, '\n', n - i);
      if (!nl) { i = n; break; }
      i = (size_t)(nl - p) + 1u;
      h = 0;
      continue;
    }
    if (i + 32u > n) break;
    const __m256i b = _mm256_loadu_si256((const __m256i *)(p + i));
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(b, vnl), _mm256_cmpeq_epi8(b, vcr)), _mm256_cmpeq_epi8(b, vgt)));
    _mm256_storeu_si256((__m256i *)(out + o), b); /* (the bytes behind the first special one are overwritten by the next step) */
    if (!m) { i += 32u; o += 32u; continue; }
    const uint32_t k = (uint32_t)__builtin_ctz(m);
    i += k; o += k;
    if (p[i++] == '>') { out[o++] = '>'; h = 1; }
  }
  *hdr = h;
  if (i < n) o += mk_fasta_keep_scalar(p + i, n - i, out + o, hdr);
  return o;
}
#endif

int mk_fasta_pack_rows(const uint8_t *text, size_t n, int32_t TL, uint32_t format, uint8_t *rows, uint64_t max_rows, uint64_t *nrows) {
  if ((!text && n) || !nrows || TL < 2 || TL > 32 || (!rows && max_rows) || ((uintptr_t)rows & 15u)) return MK_ERR_ARG;
  if (format != MK_ROWS_PACKED && format != MK_ROWS_WIDE) return MK_ERR_ARG;
  enum { PIECE = 32768 };
  uint8_t sbuf[PIECE + MK_WIDE_MAX_BASES + 64] __attribute__((aligned(64)));
  const int wide = format == MK_ROWS_WIDE;
  const size_t cap = wide ? MK_WIDE_MAX_BASES : MK_PACKED_MAX_BASES, step = cap + 1u - (size_t)TL;
  size_t fill = 0;
  uint64_t r = 0;
  int hdr = 0;
  *nrows = 0;
  for (size_t at = 0; at < n; at += PIECE) {
    const size_t m = n - at < PIECE ? n - at : PIECE;
#if defined(__x86_64__) && defined(__GNUC__)
    if (mk_have_avx2()) fill += mk_fasta_keep_avx2(text + at, m, sbuf + fill, &hdr);
    else
#endif
      fill += mk_fasta_keep_scalar(text + at, m, sbuf + fill, &hdr);
    size_t s = 0;
    for (; fill - s >= cap; s += step) {
      if (r + (wide ? 2u : 1u) > max_rows) return MK_ERR_ARG;
      if (wide) r += mk_wide_pack(rows + r * (uint64_t)MK_PACKED_PITCH, sbuf + s, cap, fill - s + 32u);
      else mk_row_pack(rows + r++ * (uint64_t)MK_PACKED_PITCH, sbuf + s, cap, fill - s + 32u);
    }
    memmove(sbuf, sbuf + s, fill - s);
    fill -= s;
  }
  if (hdr) return MK_ERR_FORMAT; /* the text ends inside a '>' line: the reference gives up (iseq2comem.c:259-271) */
  if (fill >= (size_t)TL) { /* what is left holds a whole window: one shorter row */
    if (r + (wide ? 2u : 1u) > max_rows) return MK_ERR_ARG;
    memset(sbuf + fill, 0, 32);
    if (wide) r += mk_wide_pack(rows + r * (uint64_t)MK_PACKED_PITCH, sbuf, fill, fill + 32u);
    else mk_row_pack(rows + r++ * (uint64_t)MK_PACKED_PITCH, sbuf, fill, fill + 32u);
  }
  mk_rows_done();
  *nrows = r;
  return MK_OK;
}

int mk_fasta_window_init(mk_fasta_state *st, int32_t TL) {
  if (!st || TL < 2 || TL > 32) return MK_ERR_ARG;
  memset(st, 0, sizeof *st);
  st->TL = (uint32_t)TL;
  return MK_OK;
}

/* close the pending row: terminate, copy out, keep the last TL-1 bytes as the next row's overlap */
static void mk_fasta_emit(mk_fasta_state *st, uint8_t *row, uint32_t stride) {
  const uint32_t keep = st->TL - 1;
  mk_row_store(row, st->pending, st->fill, 1, stride);
  uint32_t nk = st->fill < keep ? st->fill : keep;
  memmove(st->pending, st->pending + st->fill - nk, nk);
  st->fill = nk;
  st->fresh = 0;
}

int mk_fasta_window(mk_fasta_state *st, const uint8_t *buf, size_t n, int final, uint8_t *rows, uint32_t stride,
                    uint64_t max_rows, uint64_t *nrows, size_t *consumed) {
  if (!st || !rows || !nrows || !consumed || stride > 4096 || (stride & 3) || stride < 2 * st->TL + 4) return MK_ERR_ARG;
  const uint32_t cap = stride - 1; /* payload bytes per row; the last byte is the '\n' terminator */
  uint64_t r = 0;
  size_t pos = 0;
  while (pos < n) {
    if (st->fill == cap) { /* row full: every k-mer ending in it is complete, cut here */
      if (r == max_rows) break;
      mk_fasta_emit(st, rows + r++ * (uint64_t)stride, stride);
    }
    uint8_t ch = buf[pos++];
    if (st->in_header) { /* '>' line: skipped up to its '\n' (iseq2comem.c:259-271) */
      if (ch == '\n') st->in_header = 0;
      continue;
    }
    if (ch == '\n' || ch == '\r') continue; /* line breaks do not reset the window (iseq2comem.c:257) */
    if (ch == '>') st->in_header = 1;        /* the '>' itself stays in the stream as one reset byte */
    st->pending[st->fill++] = ch; /* ACGTacgt roll in the kernel; any other byte resets (iseq2comem.c:258,275-279) */
    st->fresh++;
  }
  if (final && pos >= n && st->in_header) { /* a '>' line that the file ends in without a newline: the reference gives up
                                               * ("can not find seqences head start from '>'", iseq2comem.c:259-271) */
    *nrows = r;
    *consumed = pos;
    return MK_ERR_FORMAT;
  }
  if (final && pos >= n && st->fresh > 0 && r < max_rows) mk_fasta_emit(st, rows + r++ * (uint64_t)stride, stride);
  mk_rows_done();
  *nrows = r;
  *consumed = pos;
  return MK_OK;
}
