/*
 * kssd_oracle.c -- TEST INFRASTRUCTURE ONLY (see kssd_oracle.h).
 *
 * Plain-C restatement of the reference's `dist -L <.shuf> [-A]` sketching path at -p 1.
 * Written from the algorithm, not from the text, of /root/reference; each block cites the
 * reference lines whose behaviour it restates.  Pinned against the compiled reference by
 * oracle/check_vs_ref.sh (byte-for-byte on combco.N / combco.N.a / combco.index.N, field-wise on
 * cofiles.stat) and by the golden vectors in tests/golden/.
 */
#define _GNU_SOURCE
#include "kssd_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants of the format (global_basic.h:35-44, command_shuffle.h:20, iseq2comem.h:6-7) ---- */
#define KO_COMPONENT_SZ 8
#define KO_CTX_SPC_USE_L 8
#define KO_LD_FCTR 0.6
#define KO_MIN_SUBCTX_DIM_SMP_SZ 4096
#define KO_OCCRC_BIT 16
#define KO_OCCRC_MAX 0xffffULL
#define KO_HIBIT 0x8000000000000000ULL
#define KO_PATHLEN 256
#define KO_FQ_LEN 4096 /* iseq2comem.c:656 */

/* table sizes: primes just under 2^8 .. 2^32 (global_basic.c:75-82) */
static const unsigned int ko_primes[25] = {
    251u,       509u,       1021u,      2039u,       4093u,       8191u,       16381u,      32749u,     65521u,
    131071u,    262139u,    524287u,    1048573u,    2097143u,    4194301u,    8388593u,    16777213u,  33554393u,
    67108859u,  134217689u, 268435399u, 536870909u,  1073741789u, 2147483647u, 4294967291u};

/* Basemap (global_basic.c:62-69): A/a=0 C/c=1 G/g=2 T/t=3, everything else -1.  Bytes >= 128 index the
 * reference's 128-entry table out of bounds (undefined); the oracle defines them as invalid. */
static inline int ko_code(unsigned char ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

int ko_params_derive(int shuf_id, int k, int subk, int drlevel, ko_params *P) {
  if (!P || k < 1 || k > 16 || subk < 0 || subk > k || subk >= 8 || drlevel < 0 || drlevel > subk) return KO_ERR_ARG;
  memset(P, 0, sizeof *P);
  P->shuf_id = shuf_id;
  P->k = k;
  P->subk = subk;
  P->drlevel = drlevel;
  /* get_hashsz(): command_dist.c:288-305 */
  int pidx = 4 * (k - drlevel) - KO_CTX_SPC_USE_L - 7;
  if (pidx < 0 || pidx > 24) return KO_ERR_PRIMER;
  P->hashsize = ko_primes[pidx];
  /* seq2co_global_var_initial(): iseq2comem.c:56-84 */
  P->half_outctx_len = k - subk;
  P->hashlimit = (unsigned int)(P->hashsize * KO_LD_FCTR); /* double product truncated, :61 */
  P->component_num = (k - drlevel > KO_COMPONENT_SZ) ? (int)(1UL << (4 * (k - drlevel - KO_COMPONENT_SZ))) : 1;
  P->comp_code_bits = (k - drlevel > KO_COMPONENT_SZ) ? 4 * (k - drlevel - KO_COMPONENT_SZ) : 0; /* :518 */
  P->crvsaddmove = 4 * k - 2;
  P->tupmask = 0xffffffffffffffffULL >> (64 - 4 * k);
  P->TL = 2 * k;
  P->domask = ((1ULL << (subk * 4)) - 1) << (2 * P->half_outctx_len);
  P->undomask = ((1ULL << (P->half_outctx_len * 2)) - 1) << (2 * (k + subk));
  P->dim_start = 0;
  ko_llong subspace = 1ULL << (4 * (subk - drlevel));
  P->dim_end = P->dim_start + (int)(subspace > KO_MIN_SUBCTX_DIM_SMP_SZ ? subspace : KO_MIN_SUBCTX_DIM_SMP_SZ);
  return KO_OK;
}

/* probe sequence: HASH(K,I,S) = (K%S + I*(1 + K%(S-1))) % S in 64-bit arithmetic (global_basic.h:282-284) */
static inline ko_llong ko_probe(ko_llong key, ko_llong i, ko_llong S) { return (key % S + i * (1 + key % (S - 1))) % S; }

/* accept test + key reduction (iseq2comem.c:691-699).  returns 1 and sets *key when accepted */
static inline int ko_reduce(const ko_params *P, const int *shuf, ko_llong tuple, ko_llong crvs, ko_llong *key) {
  ko_llong uni = tuple < crvs ? tuple : crvs;
  unsigned int dim_tup = (unsigned int)((uni & P->domask) >> (P->half_outctx_len * 2));
  ko_llong pf = (ko_llong)(long long)shuf[dim_tup]; /* int widened to llong (:693) */
  if (pf >= (ko_llong)(long long)P->dim_end || pf < (ko_llong)(long long)P->dim_start) return 0;
  pf -= (ko_llong)P->dim_start;
  ko_llong lowmask = (1ULL << (P->half_outctx_len * 2)) - 1;
  *key = (((uni & P->undomask) + ((uni & lowmask) << (P->TL * 2 - P->half_outctx_len * 4))) >> (P->drlevel * 4)) + pf;
  return 1;
}

/* counted insert (iseq2comem.c:701-718) */
static inline int ko_insert_koc(const ko_params *P, ko_llong *co, ko_llong key, unsigned int *keycount) {
  ko_llong S = P->hashsize;
  for (ko_llong i = 0; i < S; i++) {
    ko_llong n = ko_probe(key, i, S);
    if (co[n] == 0) {
      co[n] = (key << KO_OCCRC_BIT) + 1;
      if (++*keycount > P->hashlimit) return KO_ERR_CROWDED; /* :708-709 */
      return KO_OK;
    }
    if ((co[n] >> KO_OCCRC_BIT) == key) {
      if ((co[n] & KO_OCCRC_MAX) < KO_OCCRC_MAX) co[n] += 1; /* saturates at 65535, :712-714 */
      return KO_OK;
    }
  }
  return KO_OK; /* table full: the reference falls out of the loop silently */
}

/* one read line: iseq2comem.c:677-720.  `row` must contain a '\n' within `maxlen` bytes. */
static int ko_walk_read_koc(const ko_params *P, const int *shuf, const unsigned char *row, size_t maxlen, ko_llong *co,
                            unsigned int *keycount) {
  int base = 1;
  ko_llong tuple = 0, crvs = 0;
  for (size_t pos = 0; pos < maxlen && row[pos] != '\n'; pos++) {
    int b = ko_code(row[pos]);
    if (b < 0) { base = 1; continue; }
    tuple = ((tuple << 2) | (ko_llong)b) & P->tupmask;
    crvs = (crvs >> 2) + (((ko_llong)b ^ 3ULL) << P->crvsaddmove);
    base++;
    if (base > P->TL) {
      ko_llong key;
      if (!ko_reduce(P, shuf, tuple, crvs, &key)) continue;
      int rc = ko_insert_koc(P, co, key, keycount);
      if (rc) return rc;
    }
  }
  return KO_OK;
}

int ko_koc_from_rows(const ko_params *P, const int *shuf, const unsigned char *rows, size_t stride, size_t nreads,
                     ko_llong *co, int clear_first, unsigned int *keycount_io) {
  unsigned int kc = keycount_io ? *keycount_io : 0;
  if (clear_first) { memset(co, 0, (size_t)P->hashsize * sizeof(ko_llong)); kc = 0; }
  for (size_t t = 0; t < nreads; t++) {
    int rc = ko_walk_read_koc(P, shuf, rows + t * stride, stride, co, &kc);
    if (rc) return rc;
  }
  if (keycount_io) *keycount_io = kc;
  return KO_OK;
}

int ko_koc_from_rows_omp(const ko_params *P, const int *shuf, const unsigned char *rows, size_t stride, size_t nreads,
                         ko_llong *co, int clear_first, int nthreads) {
  if (clear_first) memset(co, 0, (size_t)P->hashsize * sizeof(ko_llong));
  const ko_llong S = P->hashsize;
  (void)nthreads;
  /* same structure as iseq2comem.c:675-721: guided parallel-for over reads, unsynchronised
   * check-then-write insert with atomic write / atomic increment -- benign for timing only. */
#pragma omp parallel for num_threads(nthreads) schedule(guided)
  for (long t = 0; t < (long)nreads; t++) {
    const unsigned char *row = rows + (size_t)t * stride;
    int base = 1;
    ko_llong tuple = 0, crvs = 0;
    for (size_t pos = 0; pos < stride && row[pos] != '\n'; pos++) {
      int b = ko_code(row[pos]);
      if (b < 0) { base = 1; continue; }
      tuple = ((tuple << 2) | (ko_llong)b) & P->tupmask;
      crvs = (crvs >> 2) + (((ko_llong)b ^ 3ULL) << P->crvsaddmove);
      base++;
      if (base <= P->TL) continue;
      ko_llong key;
      if (!ko_reduce(P, shuf, tuple, crvs, &key)) continue;
      for (ko_llong i = 0; i < S; i++) {
        ko_llong n = ko_probe(key, i, S);
        ko_llong cur;
#pragma omp atomic read
        cur = co[n];
        if (cur == 0) {
#pragma omp atomic write
          co[n] = (key << KO_OCCRC_BIT) + 1;
          break;
        } else if ((cur >> KO_OCCRC_BIT) == key) {
          if ((cur & KO_OCCRC_MAX) < KO_OCCRC_MAX) {
#pragma omp atomic
            co[n] += 1;
          }
          break;
        }
      }
    }
  }
  return KO_OK;
}

/* ---- shard-merge model (see kssd_oracle.h) ---- */
int ko_partial_from_rows(const ko_params *P, const int *shuf, const unsigned char *rows, size_t stride, size_t nreads,
                         ko_llong first_read_ordinal, ko_llong *keys, unsigned int *counts, ko_llong *ords, size_t cap,
                         size_t *n_out) {
  ko_llong S = P->hashsize;
  ko_llong *co = calloc((size_t)S, sizeof(ko_llong));  /* key+1 */
  ko_llong *cnt = calloc((size_t)S, sizeof(ko_llong));
  ko_llong *first = calloc((size_t)S, sizeof(ko_llong));
  if (!co || !cnt || !first) { free(co); free(cnt); free(first); return KO_ERR_IO; }
  for (size_t t = 0; t < nreads; t++) {
    const unsigned char *row = rows + t * stride;
    int base = 1;
    ko_llong tuple = 0, crvs = 0;
    for (size_t pos = 0; pos < stride && row[pos] != '\n'; pos++) {
      int b = ko_code(row[pos]);
      if (b < 0) { base = 1; continue; }
      tuple = ((tuple << 2) | (ko_llong)b) & P->tupmask;
      crvs = (crvs >> 2) + (((ko_llong)b ^ 3ULL) << P->crvsaddmove);
      base++;
      if (base <= P->TL) continue;
      ko_llong key;
      if (!ko_reduce(P, shuf, tuple, crvs, &key)) continue;
      for (ko_llong i = 0; i < S; i++) {
        ko_llong n = ko_probe(key, i, S);
        if (co[n] == 0) { co[n] = key + 1; cnt[n] = 1; first[n] = ((first_read_ordinal + t) << 12) | pos; break; }
        if (co[n] == key + 1) { cnt[n]++; break; }
      }
    }
  }
  size_t n = 0;
  int rc = KO_OK;
  for (ko_llong s = 0; s < S; s++) {
    if (!co[s]) continue;
    if (n == cap) { rc = KO_ERR_ARG; break; }
    keys[n] = co[s] - 1;
    counts[n] = cnt[s] > 65535 ? 65535u : (unsigned int)cnt[s];
    ords[n] = first[s];
    n++;
  }
  *n_out = n;
  free(co); free(cnt); free(first);
  return rc;
}

typedef struct { ko_llong key, ord, cnt; } ko_rec;
static int ko_cmp_key(const void *a, const void *b) {
  const ko_rec *x = a, *y = b;
  return x->key < y->key ? -1 : x->key > y->key ? 1 : x->ord < y->ord ? -1 : x->ord > y->ord;
}
static int ko_cmp_ord(const void *a, const void *b) {
  const ko_rec *x = a, *y = b;
  return x->ord < y->ord ? -1 : x->ord > y->ord;
}
int ko_layout_from_partials(const ko_params *P, int nparts, const ko_llong *const *keys, const unsigned int *const *counts,
                            const ko_llong *const *ords, const size_t *n, ko_llong *co) {
  size_t total = 0;
  for (int p = 0; p < nparts; p++) total += n[p];
  ko_rec *r = malloc((total + 1) * sizeof(ko_rec));
  size_t m = 0;
  for (int p = 0; p < nparts; p++)
    for (size_t i = 0; i < n[p]; i++) { r[m].key = keys[p][i]; r[m].ord = ords[p][i]; r[m].cnt = counts[p][i]; m++; }
  qsort(r, m, sizeof(ko_rec), ko_cmp_key);
  size_t d = 0;
  for (size_t i = 0; i < m; i++) { /* same key: counts add, first ordinal = min (sorted: the first one) */
    if (d && r[d - 1].key == r[i].key) r[d - 1].cnt += r[i].cnt;
    else r[d++] = r[i];
  }
  qsort(r, d, sizeof(ko_rec), ko_cmp_ord);
  memset(co, 0, (size_t)P->hashsize * sizeof(ko_llong));
  unsigned int keycount = 0;
  int rc = KO_OK;
  for (size_t i = 0; i < d && rc == KO_OK; i++) {
    rc = ko_insert_koc(P, co, r[i].key, &keycount); /* first sighting: slot = (key<<16)+1 */
    if (rc == KO_OK) {
      ko_llong c = r[i].cnt > 65535 ? 65535 : r[i].cnt;
      for (ko_llong j = 0; j < P->hashsize; j++) { /* set the merged, clamped count on the slot just taken */
        ko_llong s = ko_probe(r[i].key, j, P->hashsize);
        if ((co[s] >> KO_OCCRC_BIT) == r[i].key && co[s] != 0) { co[s] = (r[i].key << KO_OCCRC_BIT) + c; break; }
      }
    }
  }
  free(r);
  return rc;
}

/* ---- fgets() over a memory stream, including the EOF indicator feof() reports ---- */
typedef struct { const unsigned char *p, *end; int eof; } ko_ms;
static unsigned char *ko_ms_gets(ko_ms *s, unsigned char *buf, int size) {
  int n = 0;
  while (n < size - 1) {
    if (s->p == s->end) { s->eof = 1; break; }
    unsigned char c = *s->p++;
    buf[n++] = c;
    if (c == '\n') break;
  }
  if (n == 0) return NULL;
  buf[n] = 0;
  return buf;
}

int ko_koc_from_fastq_bytes(const ko_params *P, const int *shuf, const unsigned char *fq, size_t n, ko_llong *co,
                            ko_llong *nreads_out) {
  /* reader: iseq2comem.c:672-673.  Four fgets(…,FQ_LEN,…) per record, line 2 kept; a record whose
   * 4th fgets fails is dropped.  Batching by 65536 does not change the sequential result. */
  ko_ms s = {fq, fq + n, 0};
  unsigned char tmp[KO_FQ_LEN], seq[KO_FQ_LEN];
  unsigned int keycount = 0;
  ko_llong nreads = 0;
  memset(co, 0, (size_t)P->hashsize * sizeof(ko_llong)); /* :663 */
  while (!s.eof) {
    if (!(ko_ms_gets(&s, tmp, KO_FQ_LEN) && ko_ms_gets(&s, seq, KO_FQ_LEN) && ko_ms_gets(&s, tmp, KO_FQ_LEN) &&
          ko_ms_gets(&s, tmp, KO_FQ_LEN)))
      continue; /* inner for-loop ends; outer while re-tests feof (:672) */
    size_t len = strlen((char *)seq);
    if (len == 0 || seq[len - 1] != '\n') return KO_ERR_CONTRACT; /* line >= 4095 chars: reference walks into stale buffer */
    int rc = ko_walk_read_koc(P, shuf, seq, len, co, &keycount);
    if (rc) return rc;
    nreads++;
  }
  if (nreads_out) *nreads_out = nreads;
  return KO_OK;
}

unsigned int ko_dump_koc(const ko_params *P, const ko_llong *co, uint32_t **ids, uint16_t **cnts, size_t *n_out) {
  /* iseq2comem.c:539-553: slot order; component = key % component_num; id = key >> comp_code_bits */
  unsigned int wr = 0;
  for (int c = 0; c < P->component_num; c++) n_out[c] = 0;
  for (ko_llong s = 0; s < P->hashsize; s++) {
    if (co[s] == 0) continue;
    ko_llong key = co[s] >> KO_OCCRC_BIT;
    int c = (int)(key % (ko_llong)P->component_num);
    if (ids) {
      ids[c][n_out[c]] = (uint32_t)(co[s] >> (P->comp_code_bits + KO_OCCRC_BIT));
      cnts[c][n_out[c]] = (uint16_t)(co[s] & KO_OCCRC_MAX);
    }
    n_out[c]++;
    wr++;
  }
  return wr;
}

/* ---- FASTQ without -A: fastq2co() (iseq2comem.c:323-419) + write_fqco2file() (:596-621) ---- */
#define KO_FQCO_LEN 20000     /* iseq2comem.c:319 */
#define KO_CT_BIT 4           /* :320 */
#define KO_CT_MAX 0xfULL      /* :321 */
int ko_co_from_fastq_bytes(const ko_params *P, const int *shuf, const unsigned char *fq, size_t n, int Q, int M,
                           ko_llong *co, ko_llong *nlines_out) {
  if ((ko_llong)M >= KO_CT_MAX) return KO_ERR_ARG; /* :325 */
  ko_llong S = P->hashsize, tuple = 0, crvs = 0;
  memset(co, 0, (size_t)S * sizeof(ko_llong)); /* :328 */
  ko_ms s = {fq, fq + n, 0};
  /* the reference's two line buffers live across records, so a quality line shorter than its sequence line
   * exposes the previous record's bytes; kept (zero-filled at start where the reference has malloc garbage) */
  unsigned char *seq = calloc(KO_FQCO_LEN + 10, 1), *qual = calloc(KO_FQCO_LEN + 10, 1);
  if (!seq || !qual) { free(seq); free(qual); return KO_ERR_IO; }
  ko_ms_gets(&s, seq, KO_FQCO_LEN); ko_ms_gets(&s, seq, KO_FQCO_LEN);   /* :343 */
  ko_ms_gets(&s, qual, KO_FQCO_LEN); ko_ms_gets(&s, qual, KO_FQCO_LEN); /* :344 */
  ko_llong base = 1, line_num = 0;
  int rc = KO_OK;
  int sl = (int)strlen((char *)seq);
  for (int pos = 0; pos < sl; pos++) {
    if (seq[pos] == '\n') { /* :350-363: next record is read here; it is walked only if that did not touch EOF */
      ko_ms_gets(&s, seq, KO_FQCO_LEN); ko_ms_gets(&s, seq, KO_FQCO_LEN);
      ko_ms_gets(&s, qual, KO_FQCO_LEN); ko_ms_gets(&s, qual, KO_FQCO_LEN);
      sl = (int)strlen((char *)seq);
      line_num += 4;
      if (s.eof) break;
      base = 1;
      pos = -1;
      continue;
    }
    int b = ko_code(seq[pos]);
    if (b < 0 || (int)(signed char)qual[pos] < Q) { base = 1; continue; } /* :367-379 */
    tuple = ((tuple << 2) | (ko_llong)b) & P->tupmask;
    crvs = (crvs >> 2) + (((ko_llong)b ^ 3ULL) << P->crvsaddmove);
    base++;
    if (base <= P->TL) continue; /* :382 */
    ko_llong key;
    if (!ko_reduce(P, shuf, tuple, crvs, &key)) continue;
    for (ko_llong i = 0; i < S; i++) { /* :393-411; keycount is never advanced there, so no crowding abort */
      ko_llong slot = ko_probe(key, i, S);
      if (co[slot] == 0) {
        co[slot] = M == 1 ? ((key << KO_CT_BIT) | KO_CT_MAX) : ((key << KO_CT_BIT) + 1ULL);
        break;
      } else if ((co[slot] >> KO_CT_BIT) == key) {
        if ((co[slot] & KO_CT_MAX) == KO_CT_MAX) break;
        co[slot] += 1ULL;
        if (!((long long)(co[slot] & KO_CT_MAX) < (long long)M)) co[slot] |= KO_CT_MAX;
        break;
      }
    }
  }
  if (sl > 0 && !s.eof && seq[sl - 1] != '\n') rc = KO_ERR_CONTRACT; /* line of LEN-1 chars or more: walk stops short */
  if (nlines_out) *nlines_out = line_num;
  free(seq); free(qual);
  return rc;
}

unsigned int ko_dump_fqco(const ko_params *P, const ko_llong *co, uint32_t **ids, size_t *n_out) {
  /* write_fqco2file(): slots whose 4-bit count field is saturated, in slot order */
  unsigned int wr = 0;
  for (int c = 0; c < P->component_num; c++) n_out[c] = 0;
  for (ko_llong s = 0; s < P->hashsize; s++) {
    if ((co[s] & KO_CT_MAX) != KO_CT_MAX) continue;
    int c = (int)((co[s] >> KO_CT_BIT) % (ko_llong)P->component_num);
    if (ids) ids[c][n_out[c]] = (uint32_t)(co[s] >> (P->comp_code_bits + KO_CT_BIT));
    n_out[c]++;
    wr++;
  }
  return wr;
}

int ko_co_from_fasta_bytes(const ko_params *P, const int *shuf, const unsigned char *fa, size_t n, ko_llong *co, int uniq) {
  /* fasta2co(): iseq2comem.c:218-315; uniq_fasta2co(): :729-828.  The 64 KiB refill window is
   * transparent to the result except for the header-skip loop reading buff[-1] right after a refill
   * (:268, undefined) which the oracle does not model. */
  ko_llong S = P->hashsize;
  memset(co, 0, (size_t)S * sizeof(ko_llong));
  if (n == 0) return KO_ERR_CONTRACT; /* :235 err() on empty input */
  ko_llong tuple = 0, crvs = 0, base = 1;
  unsigned int keycount = 0;
  for (size_t pos = 0; pos < n; pos++) {
    unsigned char ch = fa[pos];
    int b = ko_code(ch);
    if (b >= 0) {
      tuple = ((tuple << 2) | (ko_llong)b) & P->tupmask;
      crvs = (crvs >> 2) + (((ko_llong)b ^ 3ULL) << P->crvsaddmove);
      base++;
    } else if (ch == '\n' || ch == '\r') {
      continue; /* line breaks do not reset the window (:257) */
    } else if (ch == '>') {
      while (pos < n && fa[pos] != '\n') pos++; /* skip header (:259-271) */
      if (pos >= n) return KO_ERR_CONTRACT;     /* header without newline at EOF: reference err()s (:269) */
      base = 1;
      continue;
    } else {
      base = 1; /* any other byte, alphabetic or not (:258,:275-279) */
      continue;
    }
    if (base > (ko_llong)P->TL) {
      ko_llong key;
      if (!ko_reduce(P, shuf, tuple, crvs, &key)) continue;
      for (ko_llong i = 0; i < S; i++) {
        ko_llong s = ko_probe(key, i, S);
        if (co[s] == 0) {
          co[s] = key; /* key 0 leaves the slot empty yet counts (:300-305) */
          if (++keycount > P->hashlimit) return KO_ERR_CROWDED;
          break;
        }
        if (!uniq) {
          if (co[s] == key) break;
        } else if ((co[s] | KO_HIBIT) == (key | KO_HIBIT)) {
          co[s] |= KO_HIBIT; /* second sighting flags the slot (:819-821) */
          break;
        }
      }
    }
  }
  return KO_OK;
}

unsigned int ko_dump_co(const ko_params *P, const ko_llong *co, uint32_t **ids, size_t *n_out) {
  /* iseq2comem.c:638-646: skips empty and HIBIT-flagged slots; file = slot % component_num */
  unsigned int wr = 0;
  for (int c = 0; c < P->component_num; c++) n_out[c] = 0;
  for (ko_llong s = 0; s < P->hashsize; s++) {
    if (co[s] == 0 || co[s] >= KO_HIBIT) continue;
    int c = (int)(co[s] % (ko_llong)P->component_num);
    if (ids) ids[c][n_out[c]] = (uint32_t)(co[s] >> P->comp_code_bits);
    n_out[c]++;
    wr++;
  }
  return wr;
}

int ko_shuf_read(const char *path, int header[4], int **table_out, size_t *len_out) {
  /* command_shuffle.c:215-235: 16-byte header {id,k,subk,drlevel} then int32[16^subk] */
  size_t pl = strlen(path);
  if (pl < 5 || strcmp(path + pl - 5, ".shuf") != 0) return KO_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) return KO_ERR_IO;
  if (fread(header, sizeof(int), 4, f) != 4) { fclose(f); return KO_ERR_IO; }
  if (header[2] < 0 || header[2] >= 8) { fclose(f); return KO_ERR_ARG; }
  size_t len = (size_t)1 << (4 * header[2]);
  int *t = malloc(len * sizeof(int));
  if (!t) { fclose(f); return KO_ERR_IO; }
  if (fread(t, sizeof(int), len, f) != len) { free(t); fclose(f); return KO_ERR_IO; }
  fclose(f);
  *table_out = t;
  *len_out = len;
  return KO_OK;
}

/* ---- stage I driver: command_dist.c:341-500 at -p 1 with the given file order ---- */
static int ko_has_suffix(const char *name, const char *suf) {
  size_t a = strlen(name), b = strlen(suf);
  return a >= b && strcmp(name + a - b, suf) == 0;
}
static int ko_is_fastq_name(const char *name) {
  /* isOK_fmt_infile(name, fastq_fmt): strip one .gz/.bz2 then test .fq/.fastq (global_basic.h:162-186) */
  char tmp[KO_PATHLEN * 2];
  snprintf(tmp, sizeof tmp, "%s", name);
  if (ko_has_suffix(tmp, ".gz")) tmp[strlen(tmp) - 3] = 0;
  else if (ko_has_suffix(tmp, ".bz2")) tmp[strlen(tmp) - 4] = 0;
  return ko_has_suffix(tmp, ".fq") || ko_has_suffix(tmp, ".fastq");
}
static unsigned char *ko_slurp_zcat(const char *path, size_t *n_out) {
  /* every input goes through `zcat -fc` (iseq2comem.c:216,666-669) */
  char cmd[KO_PATHLEN * 2 + 16];
  snprintf(cmd, sizeof cmd, "zcat -fc %s", path);
  FILE *p = popen(cmd, "r");
  if (!p) return NULL;
  size_t cap = 1 << 20, n = 0;
  unsigned char *buf = malloc(cap);
  for (;;) {
    if (n == cap) { cap *= 2; buf = realloc(buf, cap); }
    size_t r = fread(buf + n, 1, cap - n, p);
    if (r == 0) break;
    n += r;
  }
  pclose(p);
  *n_out = n;
  return buf;
}

int ko_dist_stage1(const char *shuf_path, int abundance, int uniq, const char *outdir, int nfiles, const char **files) {
  return ko_dist_stage1_ex(shuf_path, abundance, uniq, 0, 1, outdir, nfiles, files); /* defaults: command_dist_wrapper.c:79-80 */
}
int ko_dist_stage1_ex(const char *shuf_path, int abundance, int uniq, int Q, int M, const char *outdir, int nfiles,
                      const char **files) {
  int header[4], *shuf = NULL;
  size_t shuf_len = 0;
  int rc = ko_shuf_read(shuf_path, header, &shuf, &shuf_len);
  if (rc) return rc;
  ko_params P;
  rc = ko_params_derive(header[0], header[1], header[2], header[3], &P);
  if (rc) { free(shuf); return rc; }
  mkdir(outdir, 0777);
  ko_llong *co = malloc((size_t)P.hashsize * sizeof(ko_llong));
  if (!co) { free(shuf); return KO_ERR_IO; }

  int C = P.component_num;
  FILE **fid = calloc(C, sizeof(FILE *)), **fab = calloc(C, sizeof(FILE *));
  size_t **index = calloc(C, sizeof(size_t *));
  char path[KO_PATHLEN * 2];
  for (int c = 0; c < C; c++) {
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    fid[c] = fopen(path, "wb");
    index[c] = calloc(nfiles + 1, sizeof(size_t));
    if (!fid[c]) return KO_ERR_IO;
  }
  uint32_t **ids = calloc(C, sizeof(uint32_t *));
  uint16_t **cnts = calloc(C, sizeof(uint16_t *));
  size_t *nout = calloc(C, sizeof(size_t));
  unsigned int *ctx_ct = calloc(nfiles, sizeof(unsigned int));
  ko_llong all_ctx_ct = 0;
  int abundance_at_start = abundance;

  for (int i = 0; i < nfiles && rc == KO_OK; i++) {
    size_t n = 0;
    unsigned char *bytes = ko_slurp_zcat(files[i], &n);
    if (!bytes) { rc = KO_ERR_IO; break; }
    int is_fq = ko_is_fastq_name(files[i]);
    if (is_fq && abundance) {
      rc = ko_koc_from_fastq_bytes(&P, shuf, bytes, n, co, NULL);
      if (rc == KO_OK) {
        ko_dump_koc(&P, co, NULL, NULL, nout);
        for (int c = 0; c < C; c++) { ids[c] = malloc(4 * (nout[c] + 1)); cnts[c] = malloc(2 * (nout[c] + 1)); }
        ctx_ct[i] = ko_dump_koc(&P, co, ids, cnts, nout);
      }
    } else if (is_fq) {
      rc = ko_co_from_fastq_bytes(&P, shuf, bytes, n, Q, M, co, NULL); /* command_dist.c:385-386 */
      if (rc == KO_OK) {
        ko_dump_fqco(&P, co, NULL, nout);
        for (int c = 0; c < C; c++) { ids[c] = malloc(4 * (nout[c] + 1)); cnts[c] = NULL; }
        ctx_ct[i] = ko_dump_fqco(&P, co, ids, nout);
      }
    } else {
      if (abundance) abundance = 0; /* command_dist.c:389-392: -A is switched off for good by the first FASTA */
      rc = ko_co_from_fasta_bytes(&P, shuf, bytes, n, co, uniq);
      if (rc == KO_OK) {
        ko_dump_co(&P, co, NULL, nout);
        for (int c = 0; c < C; c++) { ids[c] = malloc(4 * (nout[c] + 1)); cnts[c] = NULL; }
        ctx_ct[i] = ko_dump_co(&P, co, ids, nout);
      }
    }
    free(bytes);
    if (rc) break;
    all_ctx_ct += ctx_ct[i];
    for (int c = 0; c < C; c++) {
      fwrite(ids[c], 4, nout[c], fid[c]);
      index[c][i + 1] = index[c][i] + nout[c];
      if (cnts[c]) {
        if (!fab[c]) {
          snprintf(path, sizeof path, "%s/combco.%d.a", outdir, c);
          fab[c] = fopen(path, "wb");
        }
        fwrite(cnts[c], 2, nout[c], fab[c]);
      }
      free(ids[c]);
      free(cnts[c]);
      ids[c] = NULL;
      cnts[c] = NULL;
    }
  }
  (void)abundance_at_start;
  for (int c = 0; c < C; c++) {
    if (fid[c]) fclose(fid[c]);
    if (fab[c]) fclose(fab[c]);
    if (rc == KO_OK) {
      snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
      FILE *fi = fopen(path, "wb");
      if (!fi) { rc = KO_ERR_IO; break; }
      fwrite(index[c], sizeof(size_t), nfiles + 1, fi); /* command_dist.c:464 */
      fclose(fi);
    }
    free(index[c]);
  }
  if (rc == KO_OK) {
    /* cofiles.stat: co_dstat_t (global_basic.h:116-126) 32 bytes, padding zeroed here */
    unsigned char hdr[32];
    memset(hdr, 0, sizeof hdr);
    uint32_t u32;
    int32_t i32;
    u32 = (uint32_t)P.shuf_id; memcpy(hdr + 0, &u32, 4);
    hdr[4] = abundance ? 1 : 0;
    i32 = P.k * 2; memcpy(hdr + 8, &i32, 4);
    i32 = P.drlevel * 2; memcpy(hdr + 12, &i32, 4);
    i32 = C; memcpy(hdr + 16, &i32, 4);
    i32 = nfiles; memcpy(hdr + 20, &i32, 4);
    memcpy(hdr + 24, &all_ctx_ct, 8);
    snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
    FILE *fs = fopen(path, "wb");
    if (!fs) rc = KO_ERR_IO;
    else {
      fwrite(hdr, 1, 32, fs);
      fwrite(ctx_ct, sizeof(unsigned int), nfiles, fs);
      for (int i = 0; i < nfiles; i++) {
        char name[KO_PATHLEN];
        memset(name, 0, sizeof name);
        strncpy(name, files[i], KO_PATHLEN - 1);
        fwrite(name, 1, KO_PATHLEN, fs);
      }
      fclose(fs);
    }
  }
  free(fid); free(fab); free(index); free(ids); free(cnts); free(nout); free(ctx_ct); free(co); free(shuf);
  return rc;
}


/* ---- `set -u` / `set -q`: sketch_union() (command_set.c:241-319) / uniq_sketch_union() (:427-512) ---- */
#define KO_COMPONENT_SZ 8 /* global_basic.h:36: ids of one component live in [0, 16^8) */
size_t ko_set_union(const uint32_t *ids, size_t n, int uniq, uint32_t *out) {
  const size_t comp_sz = (size_t)1 << (4 * KO_COMPONENT_SZ);
  ko_llong *dict = calloc(comp_sz / 64, sizeof(ko_llong));                 /* :280 / :467: k-mer present */
  ko_llong *dict2 = uniq ? malloc(comp_sz / 8) : NULL;                     /* :468: 1 = seen once so far, 0 = repeated */
  if (!dict || (uniq && !dict2)) { free(dict); free(dict2); return (size_t)-1; }
  if (uniq) memset(dict2, 0xff, comp_sz / 8);                              /* :474 */
  for (size_t i = 0; i < n; i++) {
    const uint32_t v = ids[i];
    const ko_llong bit = 0x8000000000000000ULL >> (v % 64);
    if (uniq && (dict[v / 64] & bit)) dict2[v / 64] &= ~bit;               /* :486-487 */
    dict[v / 64] |= bit;                                                   /* :295 / :489 */
  }
  size_t m = 0;
  for (size_t w = 0; w < comp_sz / 64; w++) {                              /* :303-312 / :496-505: ascending ids */
    const ko_llong word = uniq ? (dict[w] & dict2[w]) : dict[w];
    if (!word) continue;
    for (int b = 0; b < 64; b++)
      if ((0x8000000000000000ULL >> b) & word) { if (out) out[m] = (uint32_t)(64 * w + b); m++; }
  }
  free(dict); free(dict2);
  return m;
}

int ko_set_stage(const char *indir, const char *outdir, int uniq, int answer_yes) {
  char path[KO_PATHLEN * 2];
  unsigned char hdr[32]; /* co_dstat_t (global_basic.h:116-126) is copied as it is, padding bytes included (:274) */
  snprintf(path, sizeof path, "%s/cofiles.stat", indir);
  FILE *f = fopen(path, "rb");
  if (!f) return KO_ERR_IO;
  if (fread(hdr, 1, 32, f) != 32) { fclose(f); return KO_ERR_IO; }
  fclose(f);
  int32_t comp_num, infile_num;
  memcpy(&comp_num, hdr + 16, 4);
  memcpy(&infile_num, hdr + 20, 4);
  const char *prefix = uniq ? "uniq_pan" : "pan";
  if (infile_num == 1 && answer_yes) { /* :254-267 / :441-455: the single sketch is renamed in place */
    for (int c = 0; c < comp_num; c++) {
      char a[KO_PATHLEN * 2], b[KO_PATHLEN * 2];
      snprintf(a, sizeof a, "%s/combco.%d", indir, c);
      snprintf(b, sizeof b, "%s/%s.%d", indir, prefix, c);
      if (rename(a, b) != 0) return KO_ERR_IO;
    }
    return KO_OK;
  }
  mkdir(outdir, 0777);
  snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
  f = fopen(path, "wb");
  if (!f) return KO_ERR_IO;
  fwrite(hdr, 1, 32, f);
  fclose(f);
  for (int c = 0; c < comp_num; c++) {
    snprintf(path, sizeof path, "%s/combco.%d", indir, c);
    struct stat st;
    if (stat(path, &st) != 0) return KO_ERR_IO;
    const size_t n = (size_t)st.st_size / 4;
    uint32_t *ids = malloc(4 * (n + 1));
    f = fopen(path, "rb");
    if (!f || !ids) { free(ids); return KO_ERR_IO; }
    if (fread(ids, 4, n, f) != n) { fclose(f); free(ids); return KO_ERR_IO; }
    fclose(f);
    uint32_t *out = malloc(4 * (n + 1));
    const size_t m = ko_set_union(ids, n, uniq, out);
    free(ids);
    if (m == (size_t)-1) { free(out); return KO_ERR_IO; }
    snprintf(path, sizeof path, "%s/%s.%d", outdir, prefix, c);
    f = fopen(path, "wb");
    if (!f) { free(out); return KO_ERR_IO; }
    fwrite(out, 4, m, f);
    fclose(f);
    free(out);
  }
  return KO_OK;
}


/* ---- `set -i <pan>` / `set -s <pan>`: sketch_operate() (command_set.c:321-425) ---- */
size_t ko_set_filter(const uint32_t *pan, size_t npan, int keep_members, const uint32_t *ids, size_t n, uint32_t *out) {
  const size_t comp_sz = (size_t)1 << (4 * KO_COMPONENT_SZ);
  ko_llong *dict = calloc(comp_sz / 64, sizeof(ko_llong)); /* :361,364 */
  if (!dict) return (size_t)-1;
  for (size_t i = 0; i < npan; i++) dict[pan[i] / 64] |= 0x8000000000000000ULL >> (pan[i] % 64); /* :374-377 */
  size_t m = 0;
  for (size_t i = 0; i < n; i++) { /* :394-404: order kept */
    const int member = (dict[ids[i] / 64] & (0x8000000000000000ULL >> (ids[i] % 64))) > 0;
    if (keep_members == member) out[m++] = ids[i];
  }
  free(dict);
  return m;
}

static unsigned char *ko_slurp(const char *path, size_t *n_out) {
  struct stat st;
  if (stat(path, &st) != 0) return NULL;
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  unsigned char *b = malloc((size_t)st.st_size + 8);
  if (b && fread(b, 1, (size_t)st.st_size, f) != (size_t)st.st_size) { free(b); b = NULL; }
  fclose(f);
  *n_out = (size_t)st.st_size;
  return b;
}

int ko_set_operate(const char *indir, const char *pandir, const char *outdir, int intersect) {
  char path[KO_PATHLEN * 2];
  size_t pan_stat_n = 0, stat_n = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", pandir);
  unsigned char *pan_stat = ko_slurp(path, &pan_stat_n);
  snprintf(path, sizeof path, "%s/cofiles.stat", indir);
  unsigned char *stat_mem = ko_slurp(path, &stat_n); /* :336-340: the whole file, rewritten at the end */
  if (!pan_stat || !stat_mem || pan_stat_n < 32 || stat_n < 32) { free(pan_stat); free(stat_mem); return KO_ERR_IO; }
  uint32_t pan_id, in_id;
  int32_t pan_comp, infile_num;
  memcpy(&pan_id, pan_stat, 4); memcpy(&in_id, stat_mem, 4);
  memcpy(&pan_comp, pan_stat + 16, 4);
  memcpy(&infile_num, stat_mem + 20, 4);
  free(pan_stat);
  if (pan_id != in_id) { free(stat_mem); return KO_ERR_ARG; } /* :341 */
  uint32_t *ctx_ct = (uint32_t *)(stat_mem + 32);
  memset(ctx_ct, 0, (size_t)infile_num * 4); /* :344-345; all_ctx_ct in the header is left as it was */
  mkdir(outdir, 0777);
  size_t *pos = malloc(sizeof(size_t) * ((size_t)infile_num + 1)), *post = malloc(sizeof(size_t) * ((size_t)infile_num + 1));
  int rc = KO_OK;
  for (int c = 0; c < pan_comp && rc == KO_OK; c++) { /* :365: the PAN directory's component count drives the loop */
    size_t nb = 0, ib = 0, cb = 0;
    snprintf(path, sizeof path, "%s/pan.%d", pandir, c);
    unsigned char *pan = ko_slurp(path, &nb);
    if (!pan) { snprintf(path, sizeof path, "%s/uniq_pan.%d", pandir, c); pan = ko_slurp(path, &nb); } /* :369-372 */
    snprintf(path, sizeof path, "%s/combco.index.%d", indir, c);
    unsigned char *idx = ko_slurp(path, &ib);
    snprintf(path, sizeof path, "%s/combco.%d", indir, c);
    unsigned char *co = ko_slurp(path, &cb);
    if (!pan || !idx || !co || ib < sizeof(size_t) * ((size_t)infile_num + 1)) { free(pan); free(idx); free(co); rc = KO_ERR_IO; break; }
    memcpy(pos, idx, sizeof(size_t) * ((size_t)infile_num + 1));
    const uint32_t *ids = (const uint32_t *)co;
    uint32_t *out = malloc(4 * (pos[infile_num] + 1));
    size_t m = 0;
    post[0] = 0;
    for (int i = 0; i < infile_num; i++) { /* :392-405, one file at a time through the same dictionary */
      const size_t k = ko_set_filter((const uint32_t *)pan, nb / 4, intersect, ids + pos[i], pos[i + 1] - pos[i], out + m);
      if (k == (size_t)-1) { rc = KO_ERR_IO; break; }
      m += k;
      post[i + 1] = m;
      ctx_ct[i] += (uint32_t)k;
    }
    if (rc == KO_OK) {
      snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
      FILE *f = fopen(path, "wb");
      if (f) { fwrite(out, 4, m, f); fclose(f); } else rc = KO_ERR_IO;
      snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
      f = fopen(path, "wb");
      if (f) { fwrite(post, sizeof(size_t), (size_t)infile_num + 1, f); fclose(f); } else rc = KO_ERR_IO;
    }
    free(out); free(pan); free(idx); free(co);
  }
  if (rc == KO_OK) {
    snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
    FILE *f = fopen(path, "wb");
    if (f) { fwrite(stat_mem, 1, stat_n, f); fclose(f); } else rc = KO_ERR_IO; /* :415-418 */
  }
  free(pos); free(post); free(stat_mem);
  return rc;
}


/* ---- `set -g <file.tsv>`: organize_taxf() (command_set.c:635-705) + grouping_genomes() (:831-974) ---- */
static const unsigned int ko_primer[25] = { /* global_basic.c:75-82 */
  251, 509, 1021, 2039, 4093, 8191, 16381, 32749, 65521, 131071, 262139, 524287, 1048573, 2097143, 4194301, 8388593,
  16777213, 33554393, 67108859, 134217689, 268435399, 536870909, 1073741789, 2147483647, 4294967291u};

static int ko_next_prime(int n) { /* global_basic.c:453-475 */
  for (;;) {
    int composite = 0;
    for (int j = 2; j <= (int)sqrt((double)n); j++)
      if (n % j == 0) { composite = 1; break; }
    if (!composite) return n;
    n++;
  }
}

/* the per-taxon table of grouping_genomes() (:874-897): ids in the given order, FCFS double hashing in 32-bit
 * unsigned arithmetic exactly as HASH(unsigned, int, int) evaluates (global_basic.h:282-284), id 0 never stored,
 * an id that finds no place in `table_size` probes is dropped; dump in slot order (:907-915) */
size_t ko_group_layout(const uint32_t *ids, size_t n, uint32_t table_size, uint32_t *out) {
  uint32_t *tab = calloc(table_size, 4);
  if (!tab) return (size_t)-1;
  for (size_t i = 0; i < n; i++) {
    const uint32_t key = ids[i];
    for (int x = 0; x < (int)table_size; x++) {
      const uint32_t y = (key % table_size + (uint32_t)x * (1u + key % (table_size - 1u))) % table_size;
      if (tab[y] == 0) { tab[y] = key; break; }
      if (tab[y] == key) break;
    }
  }
  size_t m = 0;
  for (uint32_t x = 0; x < table_size; x++)
    if (tab[x] != 0) out[m++] = tab[x];
  free(tab);
  return m;
}

uint32_t ko_group_table_size(uint64_t total_ids) { /* :867-872: int hashsize; LOG2(hashsize * 1.5); primer[ind - 7] */
  const unsigned long long v = (unsigned long long)((double)(int)total_ids * 1.5);
  const unsigned ind = (unsigned)(63 - __builtin_clzll(v));
  return ind > 7 ? ko_primer[ind - 7] : ko_primer[0];
}

typedef struct { int taxid; char *name; int *gids; int ng; } ko_taxon;

int ko_set_group(const char *indir, const char *taxfile, const char *outdir) {
  /* ---- organize_taxf(): line i = genome i: "<taxid>[\t<name>]"; taxa come out in the slot order of a hash of taxids ---- */
  size_t tn = 0;
  unsigned char *txt = ko_slurp(taxfile, &tn);
  if (!txt) return KO_ERR_IO;
  int ln = 0;
  for (size_t i = 0; i < tn; i++) ln += txt[i] == '\n';
  const int hashsz = ko_next_prime((int)((double)ln / 0.6)); /* LD_FCTR, global_basic.h:44 */
  ko_taxon *hs = calloc((size_t)hashsz, sizeof(ko_taxon));
  for (int i = 0; i < hashsz; i++) hs[i].taxid = -1;
  int tax_count = 0, rc = KO_OK;
  size_t at = 0;
  for (int i = 0; i < ln && rc == KO_OK; i++) {
    size_t e = at;
    while (txt[e] != '\n') e++;
    if (e - at + 1 >= KO_PATHLEN) { rc = KO_ERR_CONTRACT; break; } /* fgets(…,PATHLEN,…) would split the line (:657-659) */
    txt[e] = 0;
    char *line = (char *)txt + at;
    at = e + 1;
    char *tok = strtok(line, "\t");
    if (!tok) { rc = KO_ERR_CONTRACT; break; } /* the reference dereferences NULL here */
    const int taxid = atoi(tok);
    char *name = strtok(NULL, "\t");
    for (int n = 0; n < hashsz; n++) {
      const int hv = (taxid % hashsz + n * (1 + taxid % (hashsz - 1))) % hashsz; /* HASH() in int arithmetic */
      if (hv < 0) { rc = KO_ERR_CONTRACT; break; }
      if (hs[hv].taxid == -1) {
        hs[hv].taxid = taxid;
        hs[hv].name = name ? strdup(name) : NULL;
        hs[hv].gids = malloc(sizeof(int));
        hs[hv].gids[0] = i;
        hs[hv].ng = 1;
        tax_count++;
        break;
      } else if (hs[hv].taxid == taxid) {
        if ((hs[hv].name == NULL) != (name == NULL) || (name && strcmp(hs[hv].name, name) != 0)) { rc = KO_ERR_ARG; break; } /* :680-683 */
        hs[hv].gids = realloc(hs[hv].gids, sizeof(int) * (size_t)(hs[hv].ng + 1));
        hs[hv].gids[hs[hv].ng++] = i;
        break;
      }
    }
  }
  free(txt);
  ko_taxon *tax = malloc(sizeof(ko_taxon) * (size_t)(tax_count + 1));
  int taxn = 0;
  for (int n = 0; n < hashsz; n++)
    if (hs[n].taxid != -1) tax[taxn++] = hs[n];
  free(hs);
  /* ---- grouping_genomes() ---- */
  char path[KO_PATHLEN * 2];
  size_t sn = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", indir);
  unsigned char *st = rc == KO_OK ? ko_slurp(path, &sn) : NULL;
  if (rc == KO_OK && (!st || sn < 32)) rc = KO_ERR_IO;
  int32_t comp_num = 0, infile_num = 0;
  if (rc == KO_OK) {
    memcpy(&comp_num, st + 16, 4);
    memcpy(&infile_num, st + 20, 4);
    if (infile_num != ln) rc = KO_ERR_ARG; /* :845-846 */
  }
  if (rc == KO_OK) mkdir(outdir, 0777);
  uint32_t *ctx_ct = calloc((size_t)taxn + 1, 4);
  size_t *outidx = malloc(sizeof(size_t) * ((size_t)taxn + 1));
  ko_llong all_ctx_ct = 0;
  int outfn = 0;
  for (int c = 0; c < comp_num && rc == KO_OK; c++) {
    size_t cb = 0, ib = 0;
    snprintf(path, sizeof path, "%s/combco.%d", indir, c);
    unsigned char *co = ko_slurp(path, &cb);
    snprintf(path, sizeof path, "%s/combco.index.%d", indir, c);
    unsigned char *idx = ko_slurp(path, &ib);
    if (!co || !idx) { free(co); free(idx); rc = KO_ERR_IO; break; }
    const uint32_t *ids = (const uint32_t *)co;
    const size_t *pos = (const size_t *)idx;
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    FILE *f = fopen(path, "wb");
    if (!f) { free(co); free(idx); rc = KO_ERR_IO; break; }
    outfn = 0;
    size_t offset = 0;
    outidx[0] = 0;
    for (int t = 0; t < taxn; t++) {
      if (tax[t].taxid == 0) continue; /* :866, :906 */
      size_t total = 0;
      for (int g = 0; g < tax[t].ng; g++) total += pos[tax[t].gids[g] + 1] - pos[tax[t].gids[g]];
      /* no k-mer of this taxon in this component: the reference evaluates LOG2(0 * 1.5) = __builtin_clzll(0) here (:878, undefined), ends up
       * with a table of primer[0] slots in every build seen so far, inserts nothing and writes an EMPTY block (checked against the compiled
       * reference: oracle/check_vs_ref.py, case set_g_empty_component) */
      if (total == 0) { outfn++; outidx[outfn] = offset; continue; }
      uint32_t *cat = malloc(4 * total), *out = malloc(4 * total);
      size_t k = 0;
      for (int g = 0; g < tax[t].ng; g++)
        for (size_t i = pos[tax[t].gids[g]]; i < pos[tax[t].gids[g] + 1]; i++) cat[k++] = ids[i];
      const size_t m = ko_group_layout(cat, total, ko_group_table_size(total), out);
      fwrite(out, 4, m, f);
      offset += m; all_ctx_ct += m; ctx_ct[outfn] += (uint32_t)m;
      outfn++;
      outidx[outfn] = offset;
      free(cat); free(out);
    }
    fclose(f);
    if (rc == KO_OK) {
      snprintf(path, sizeof path, "%s/combco.index.%d", outdir, c);
      f = fopen(path, "wb");
      if (f) { fwrite(outidx, sizeof(size_t), (size_t)outfn + 1, f); fclose(f); } else rc = KO_ERR_IO;
    }
    free(co); free(idx);
  }
  if (rc == KO_OK) { /* :935-966: header with the new count, koc = 0, the new total; names "<taxid>_<name>" or "<taxid>" */
    int32_t v = outfn;
    memcpy(st + 20, &v, 4);
    st[4] = 0;
    memcpy(st + 24, &all_ctx_ct, 8);
    snprintf(path, sizeof path, "%s/cofiles.stat", outdir);
    FILE *f = fopen(path, "wb");
    if (!f) rc = KO_ERR_IO;
    else {
      fwrite(st, 1, 32, f);
      fwrite(ctx_ct, 4, (size_t)outfn, f);
      for (int t = 0; t < taxn; t++) {
        if (tax[t].taxid == 0) continue;
        char name[KO_PATHLEN];
        memset(name, 0, sizeof name); /* the reference leaves the bytes after the NUL uninitialised */
        if (tax[t].name) snprintf(name, sizeof name, "%d_%s", tax[t].taxid, tax[t].name);
        else snprintf(name, sizeof name, "%d", tax[t].taxid);
        fwrite(name, 1, KO_PATHLEN, f);
      }
      fclose(f);
    }
  }
  for (int t = 0; t < taxn; t++) { free(tax[t].name); free(tax[t].gids); }
  free(tax); free(st); free(ctx_ct); free(outidx);
  return rc;
}


/* ---- `composite -r <ref> -q <qry> [-b]`: get_species_abundance() (command_composite.c:446-649) ---- */
static int ko_cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; } /* :657-659 */

int ko_composite(const char *refdir, const char *qrydir, const char *outdir, int binvec, FILE *out) {
  char path[KO_PATHLEN * 2];
  size_t rn_bytes = 0, qn_bytes = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", refdir);
  unsigned char *rst = ko_slurp(path, &rn_bytes);
  snprintf(path, sizeof path, "%s/cofiles.stat", qrydir);
  unsigned char *qst = ko_slurp(path, &qn_bytes);
  if (!rst || !qst || rn_bytes < 32 || qn_bytes < 32) { free(rst); free(qst); return KO_ERR_IO; }
  int32_t ref_n, qry_n, comp_num;
  memcpy(&ref_n, rst + 20, 4); memcpy(&qry_n, qst + 20, 4); memcpy(&comp_num, rst + 16, 4);
  if (!qst[4]) { free(rst); free(qst); return KO_ERR_ARG; } /* :467 "query has not abundance" */
  const char *refname = (const char *)rst + 32 + 4 * (size_t)ref_n, *qryname = (const char *)qst + 32 + 4 * (size_t)qry_n;
  int **ab = malloc(sizeof(int *) * (size_t)ref_n); /* ab[r][0] = matches, ab[r][1..] = the query's counts of them (:491-492) */
  size_t *cap = malloc(sizeof(size_t) * (size_t)ref_n);
  for (int r = 0; r < ref_n; r++) { cap[r] = 8; ab[r] = malloc(sizeof(int) * 8); }
  int rc = KO_OK;
  for (int q = 0; q < qry_n && rc == KO_OK; q++) {
    for (int r = 0; r < ref_n; r++) ab[r][0] = 0;
    for (int c = 0; c < comp_num && rc == KO_OK; c++) {
      size_t a1, a2, a3, a4, a5;
      snprintf(path, sizeof path, "%s/combco.%d", refdir, c);
      uint32_t *rco = (uint32_t *)ko_slurp(path, &a1);
      snprintf(path, sizeof path, "%s/combco.index.%d", refdir, c);
      size_t *ridx = (size_t *)ko_slurp(path, &a2);
      snprintf(path, sizeof path, "%s/combco.%d", qrydir, c);
      uint32_t *qco = (uint32_t *)ko_slurp(path, &a3);
      snprintf(path, sizeof path, "%s/combco.index.%d", qrydir, c);
      size_t *qidx = (size_t *)ko_slurp(path, &a4);
      snprintf(path, sizeof path, "%s/combco.%d.a", qrydir, c);
      uint16_t *qab = (uint16_t *)ko_slurp(path, &a5);
      if (!rco || !ridx || !qco || !qidx || !qab) rc = KO_ERR_IO;
      if (rc == KO_OK) {
        /* :525-536: k-mer -> position dictionary of this query: double hashing in 32-bit unsigned arithmetic, no duplicate test */
        const int hash_sz = ko_next_prime((int)((double)(qidx[q + 1] - qidx[q]) / 0.6));
        size_t *dict = calloc((size_t)hash_sz + 1, sizeof(size_t));
        for (size_t idx = qidx[q]; idx < qidx[q + 1]; idx++)
          for (int i = 0; i < hash_sz; i++) {
            const unsigned hv = (qco[idx] % (unsigned)hash_sz + (unsigned)i * (1u + qco[idx] % (unsigned)(hash_sz - 1))) % (unsigned)hash_sz;
            if (dict[hv] == 0) { dict[hv] = idx + 1; break; }
          }
        for (int r = 0; r < ref_n; r++) /* :537-553 */
          for (size_t ri = ridx[r]; ri < ridx[r + 1]; ri++)
            for (int i = 0; i < hash_sz; i++) {
              const unsigned hv = (rco[ri] % (unsigned)hash_sz + (unsigned)i * (1u + rco[ri] % (unsigned)(hash_sz - 1))) % (unsigned)hash_sz;
              if (dict[hv] == 0) break;
              if (qco[dict[hv] - 1] == rco[ri]) {
                if ((size_t)ab[r][0] + 2 > cap[r]) { cap[r] *= 2; ab[r] = realloc(ab[r], sizeof(int) * cap[r]); }
                ab[r][++ab[r][0]] = qab[dict[hv] - 1];
                break;
              }
            }
        free(dict);
      }
      free(rco); free(ridx); free(qco); free(qidx); free(qab);
    }
    if (rc) break;
    /* :568-570: references by decreasing number of matches; qsort() of glibc <= 2.36 is a stable merge sort, i.e. ties
     * stay in index order -- restated as an insertion sort with that property */
    int *order = malloc(sizeof(int) * (size_t)ref_n);
    for (int i = 0; i < ref_n; i++) {
      int j = i;
      while (j > 0 && ab[order[j - 1]][0] < ab[i][0]) { order[j] = order[j - 1]; j--; }
      order[j] = i;
    }
    FILE *vf = NULL;
    struct { int ref_idx; float pct; } *vec = malloc(8 * ((size_t)ref_n + 1));
    int num_pass = 0;
    float vecsum = 0;
    if (binvec) { /* :573-581 */
      char dir[KO_PATHLEN * 2], qn[KO_PATHLEN + 1], vpath[KO_PATHLEN * 4];
      if (strlen(outdir) < 3) snprintf(dir, sizeof dir, "%s/abundance_Vec", refdir);
      else snprintf(dir, sizeof dir, "%s", outdir);
      mkdir(dir, 0777);
      snprintf(qn, sizeof qn, "%.*s", KO_PATHLEN, qryname + (size_t)KO_PATHLEN * q);
      const char *base = strrchr(qn, '/');
      snprintf(vpath, sizeof vpath, "%s/%s.abv", dir, base ? base + 1 : qn);
      vf = fopen(vpath, "wb");
      if (!vf) rc = KO_ERR_IO;
    }
    for (int i = 0; i < ref_n && rc == KO_OK; i++) {
      int *a = ab[order[i]];
      const int kmer_num = a[0];
      if (kmer_num < 6) break; /* MIN_KM_S */
      qsort(a + 1, (size_t)kmer_num, sizeof(int), ko_cmp_int);
      int sum = 0;
      for (int n = 1; n <= kmer_num; n++) sum += a[n];
      const int median_idx = kmer_num / 2, pct_idx = kmer_num * 0.98;
      int lastsum = 0, lastn = 0;
      for (int n = pct_idx; n <= kmer_num * 0.99; n++) { lastsum += a[n]; lastn++; }
      if (binvec) {
        if (a[median_idx] > 1 && kmer_num > 7) {
          vec[num_pass].ref_idx = order[i];
          vec[num_pass].pct = (float)lastsum / lastn;
          vecsum += vec[num_pass].pct;
          num_pass++;
        }
      } else {
        fprintf(out, "%s\t%s\t%d\t%f\t%f\t%d\t%d\n", qryname + (size_t)KO_PATHLEN * q, refname + (size_t)KO_PATHLEN * order[i], kmer_num,
                (float)sum / kmer_num, (float)lastsum / lastn, a[median_idx], a[kmer_num]);
      }
    }
    if (binvec && vf) {
      for (int i = 0; i < num_pass; i++) vec[i].pct = (vec[i].pct - 1) * 100 / (vecsum - num_pass); /* :623 */
      fwrite(vec, 8, (size_t)num_pass, vf);
      fclose(vf);
    }
    free(vec); free(order);
  }
  for (int r = 0; r < ref_n; r++) free(ab[r]);
  free(ab); free(cap); free(rst); free(qst);
  return rc;
}


/* ======== SURVEY.md 8f N4: stage II (`combco2mco`, co2mco.c:12-87) and `dist -r <mco> <co>` (command_dist.c:902-1079,
 * 1531-1690) ======== */
#include <fcntl.h>
#include <math.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct { uint32_t id, gid; } ko_pair;

/* stable merge sort by id: a row's genome numbers come out in the order the reference appends them (co2mco.c:37-56:
 * genomes ascending, positions ascending inside a genome) */
static void ko_pair_msort(ko_pair *a, ko_pair *tmp, size_t n) {
  if (n < 2) return;
  size_t h = n / 2;
  ko_pair_msort(a, tmp, h);
  ko_pair_msort(a + h, tmp, n - h);
  size_t i = 0, j = h, o = 0;
  while (i < h && j < n) tmp[o++] = a[j].id < a[i].id ? a[j++] : a[i++];
  while (i < h) tmp[o++] = a[i++];
  while (j < n) tmp[o++] = a[j++];
  memcpy(a, tmp, n * sizeof *a);
}

int ko_mco_build(const uint32_t *ids, const uint64_t *index, int cofnum, uint32_t *gids_out, uint32_t **row_ids,
                 uint64_t **row_ends, size_t *nrows) {
  size_t n = cofnum > 0 ? (size_t)index[cofnum] : 0, o = 0;
  ko_pair *p = malloc(sizeof *p * (n + 1)), *t = malloc(sizeof *t * (n + 1));
  if (!p || !t) { free(p); free(t); return KO_ERR_ARG; }
  for (int j = 0; j < cofnum; j++)                                   /* co2mco.c:37 */
    for (size_t k = (size_t)index[j]; k < (size_t)index[j + 1]; k++) /* :39 */
      p[o++] = (ko_pair){ids[k], (uint32_t)j};                       /* :41, :55 */
  ko_pair_msort(p, t, n);
  size_t d = 0;
  for (size_t i = 0; i < n; i++) d += i + 1 == n || p[i].id != p[i + 1].id;
  uint32_t *ri = malloc(4 * (d + 1));
  uint64_t *re = malloc(8 * (d + 1));
  d = 0;
  for (size_t i = 0; i < n; i++) {
    gids_out[i] = p[i].gid;                                          /* :74-80: rows ascending, each row as appended */
    if (i + 1 == n || p[i].id != p[i + 1].id) { ri[d] = p[i].id; re[d++] = i + 1; } /* :59: cumulative row ends */
  }
  free(p); free(t);
  *row_ids = ri; *row_ends = re; *nrows = d;
  return KO_OK;
}

/* mco.index.N as the reference writes it (:59-66): 2^32 cumulative row ends, 32 GiB */
static int ko_write_dense_index(const char *path, const uint32_t *row_ids, const uint64_t *row_ends, size_t nrows) {
  FILE *f = fopen(path, "wb");
  if (!f) return KO_ERR_IO;
  const size_t B = 1u << 20;
  uint64_t *buf = malloc(8 * B);
  size_t d = 0;
  uint64_t cur = 0;
  for (uint64_t r0 = 0; r0 < (1ull << 32); r0 += B) {
    for (uint64_t r = r0; r < r0 + B; r++) {
      if (d < nrows && row_ids[d] == r) cur = row_ends[d++];
      buf[r - r0] = cur;
    }
    if (fwrite(buf, 8, B, f) != B) { fclose(f); free(buf); return KO_ERR_IO; }
  }
  free(buf);
  return fclose(f) ? KO_ERR_IO : KO_OK;
}

/* run_stageII() (command_dist.c:504-552) + combco2mco(); dense_index = 0 leaves mco.index.N out (tests that cannot
 * afford 32 GiB per component) */
int ko_stage2(const char *codir, const char *mcodir, int dense_index) {
  char path[KO_PATHLEN * 2];
  size_t sn = 0;
  snprintf(path, sizeof path, "%s/cofiles.stat", codir);
  unsigned char *st = ko_slurp(path, &sn);
  if (!st || sn < 32) { free(st); return KO_ERR_IO; }
  int32_t comp_num, cofnum;
  memcpy(&comp_num, st + 16, 4); memcpy(&cofnum, st + 20, 4);
  mkdir(mcodir, 0700); /* :506 (an existing directory only earns a warning) */
  snprintf(path, sizeof path, "%s/mcofiles.stat", mcodir);
  FILE *f = fopen(path, "wb");
  if (!f) { free(st); return KO_ERR_IO; }
  fwrite(st, 4, 1, f);                                      /* shuf_id :527 */
  fwrite(st + 8, 4, 4, f);                                  /* kmerlen, dim_rd_len, comp_num, infile_num :528-531 */
  fwrite(st + 32, 1, (size_t)cofnum * (4 + KO_PATHLEN), f); /* ctx_ct list and names :534-539 */
  fclose(f);
  int rc = KO_OK;
  for (int c = 0; c < comp_num && rc == KO_OK; c++) {
    size_t nb = 0, ib = 0;
    snprintf(path, sizeof path, "%s/combco.%d", codir, c);
    uint32_t *ids = (uint32_t *)ko_slurp(path, &nb);
    snprintf(path, sizeof path, "%s/combco.index.%d", codir, c);
    uint64_t *idx = (uint64_t *)ko_slurp(path, &ib);
    if (!ids || !idx || ib < 8 * ((size_t)cofnum + 1)) { free(ids); free(idx); rc = KO_ERR_IO; break; }
    uint32_t *gids = malloc(4 * (nb / 4 + 1)), *ri = NULL;
    uint64_t *re = NULL;
    size_t nr = 0;
    rc = ko_mco_build(ids, idx, cofnum, gids, &ri, &re, &nr);
    if (rc == KO_OK) {
      snprintf(path, sizeof path, "%s/mco.%d", mcodir, c);
      f = fopen(path, "wb");
      if (!f) rc = KO_ERR_IO;
      else { fwrite(gids, 4, (size_t)idx[cofnum], f); fclose(f); }
      snprintf(path, sizeof path, "%s/mco.index.%d", mcodir, c);
      if (rc == KO_OK && dense_index) rc = ko_write_dense_index(path, ri, re, nr);
    }
    free(ids); free(idx); free(gids); free(ri); free(re);
  }
  free(st);
  return rc;
}

/* the counting loop of mco_cbdco_nobin_dist() (command_dist.c:1033-1049) for one component; row extents either from the
 * reference's dense index (dense != NULL: `s = ind > 0 ? index[ind-1] : 0 .. index[ind]`) or from the sparse row table */
void ko_mco_count(const uint32_t *gids, const uint64_t *dense, const uint32_t *row_ids, const uint64_t *row_ends, size_t nrows,
                  const uint32_t *qry_ids, const uint64_t *qry_index, int qry_num, const uint32_t *qry_ctx_ct, int ref_num,
                  uint32_t *ct) {
  for (int k = 0; k < qry_num; k++) {
    if (qry_ctx_ct[k] == 0) continue;                            /* :1035 */
    uint32_t *row = ct + (size_t)k * (size_t)ref_num;            /* :1037 */
    for (size_t n = (size_t)qry_index[k]; n < (size_t)qry_index[k + 1]; n++) {
      uint32_t ind = qry_ids[n];
      uint64_t s, e;
      if (dense) { s = ind > 0 ? dense[ind - 1] : 0; e = dense[ind]; } /* :1040-1041 */
      else {
        size_t lo = 0, hi = nrows;
        while (lo < hi) { size_t m = (lo + hi) / 2; if (row_ids[m] < ind) lo = m + 1; else hi = m; }
        if (lo == nrows || row_ids[lo] != ind) continue;
        s = lo ? row_ends[lo - 1] : 0; e = row_ends[lo];
      }
      for (uint64_t g = s; g < e; g++) row[gids[g]]++;           /* :1043-1044 */
    }
  }
}

/* output_ctrl() (command_dist.c:1637-1680): one line of distance.out, or nothing when the distance is above the
 * threshold.  Returns the line length (0 = suppressed). */
static int ko_dist_line(char *line, size_t cap, const ko_dist_opts *o, int kmerlen, int dim_rd_len, long long cmprsn,
                        const char *qname, const char *rname, unsigned X, unsigned Y, unsigned XnY) {
  double rs = 0;
  if (o->correction) {                                                  /* :1639-1646 */
    unsigned xr = X - XnY, yr = Y - XnY;
    double base = 1 - 1 / pow(4, (kmerlen - dim_rd_len));
    double px = 1 - pow(base, xr), py = 1 - pow(base, yr);
    rs = px * py * (xr + yr) / (px + py - 2 * px * py);
  }
  unsigned denom = o->metric == 0 ? X + Y - XnY : (X < Y ? X : Y);      /* :1648-1649 */
  double metric = ((double)XnY - rs) / denom;
  double dist = log(o->metric == 0 ? 1 / (2 * metric) + 0.5 : 1 / metric) / kmerlen; /* :1636, :1651 */
  if (dist > 1) dist = 1;
  if (dist > o->dthreshold) return 0;                                   /* :1653 */
  int len = snprintf(line, cap, "%s\t%s\t%u-%u|%u|%u\t%.6lf\t%.6lf", qname, rname, XnY, (unsigned)rs, X, Y, metric, dist);
  if (o->outfields > 0) {                                               /* :1657-1670 */
    double sd = pow(metric * (1 - metric) / denom, 0.5);
    double pv = 0.5 * erfc(metric / sd * pow(0.5, 0.5));
    len += snprintf(line + len, cap - (size_t)len, "\t%E\t%E", pv, pv * cmprsn);
    if (o->outfields > 1) {
      double m1 = metric - 1.96 * sd, m2 = metric + 1.96 * sd;
      double d1 = log(o->metric == 0 ? 1 / (2 * m2) + 0.5 : 1 / m2) / kmerlen;
      double d2 = log(o->metric == 0 ? 1 / (2 * m1) + 0.5 : 1 / m1) / kmerlen;
      len += snprintf(line + len, cap - (size_t)len, "\t[%.6lf,%.6lf]\t[%.6lf,%.6lf]", m1, m2, d1, d2);
    }
  }
  len += snprintf(line + len, cap - (size_t)len, "\n");
  return len;
}

/* dist_print_nobin() (command_dist.c:1531-1634) */
int ko_dist_print(FILE *fp, const ko_dist_opts *o, int kmerlen, int dim_rd_len, int ref_num, int qry_num,
                  const uint32_t *ref_ctx_ct, const uint32_t *qry_ctx_ct, const char *refnames, const char *qrynames,
                  const uint32_t *ct) {
  static const char *hdr[2][3] = {{"Jaccard\tMashD", "P-value(J)\tFDR(J)", "Jaccard_CI\tMashD_CI"},
                                  {"ContainmentM\tAafD", "P-value(C)\tFDR(C)", "ContainmentM_CI\tAafD_CI"}};
  if (o->metric < 0 || o->metric > 1 || o->outfields < 0 || o->outfields > 2) return KO_ERR_ARG;
  fprintf(fp, "Qry\tRef\tShared_k|Ref_s|Qry_s");                        /* :1567-1570 */
  for (int i = 0; i <= o->outfields; i++) fprintf(fp, "\t%s", hdr[o->metric][i]);
  fprintf(fp, "\n");
  int N = o->num_neigb;
  if (N > 1024 || N > ref_num) return KO_ERR_ARG;                       /* :1574 */
  long long cmprsn = (long long)((unsigned)ref_num * (unsigned)qry_num); /* :1562: unsigned product */
  char line[1024];
  struct { double m; int rid; } best[1025];
  for (int q = 0; q < qry_num; q++) {
    unsigned Y = qry_ctx_ct[q];
    const uint32_t *row = ct + (size_t)q * (size_t)ref_num;
    const char *qn = qrynames + (size_t)q * KO_PATHLEN;
    if (N) {                                                            /* :1591-1618 */
      for (int i = 0; i < N; i++) { best[i].m = 0; best[i].rid = -1; }
      for (int r = 0; r < ref_num; r++) {
        unsigned X = ref_ctx_ct[r], XnY = row[r];
        double m = o->metric == 1 ? (double)XnY / (X < Y ? X : Y) : (double)XnY / (X + Y - XnY);
        for (int i = N - 1; i >= 0; i--) {
          if (m > best[i].m) { best[i + 1] = best[i]; best[i].m = m; best[i].rid = r; }
          else break;
        }
      }
      for (int i = 0; i < N; i++) {
        if (best[i].rid < 0) continue;
        int len = ko_dist_line(line, sizeof line, o, kmerlen, dim_rd_len, cmprsn, qn, refnames + (size_t)best[i].rid * KO_PATHLEN,
                               ref_ctx_ct[best[i].rid], Y, row[best[i].rid]);
        if (len > 1) fwrite(line, 1, (size_t)len, fp);
      }
    } else {
      for (int r = 0; r < ref_num; r++) {                               /* :1620-1626 */
        int len = ko_dist_line(line, sizeof line, o, kmerlen, dim_rd_len, cmprsn, qn, refnames + (size_t)r * KO_PATHLEN,
                               ref_ctx_ct[r], Y, row[r]);
        if (len > 1) fwrite(line, 1, (size_t)len, fp);
      }
    }
  }
  return KO_OK;
}

static void *ko_map_file(const char *path, size_t *n) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return NULL;
  struct stat s;
  fstat(fd, &s);
  *n = (size_t)s.st_size;
  void *p = s.st_size ? mmap(NULL, (size_t)s.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : NULL;
  close(fd);
  return p == MAP_FAILED ? NULL : p;
}

/* mco_cbdco_nobin_dist(): `dist -r <refdir> -o <outdir> <qrydir>`.  refdir holds mcofiles.stat + mco.N + the dense
 * mco.index.N (ref_is_co = 0), or is a sketch directory whose inverted index is built in memory (ref_is_co = 1: same
 * numbers without the 32 GiB files).  Writes distance.out and, with keep_shared, sharedk_ct.dat. */
int ko_dist_search(const char *refdir, int ref_is_co, const char *qrydir, const char *outdir, const ko_dist_opts *o) {
  char path[KO_PATHLEN * 2];
  size_t rn = 0, qn = 0;
  snprintf(path, sizeof path, "%s/%s", refdir, ref_is_co ? "cofiles.stat" : "mcofiles.stat");
  unsigned char *rst = ko_slurp(path, &rn);
  snprintf(path, sizeof path, "%s/cofiles.stat", qrydir);
  unsigned char *qst = ko_slurp(path, &qn);
  if (!rst || !qst || qn < 32 || rn < 32) { free(rst); free(qst); return KO_ERR_IO; }
  const size_t rh = ref_is_co ? 32 : 20; /* sizeof(co_dstat_t) / sizeof(mco_dstat_t) */
  int32_t r_shuf, r_k, r_dr, r_comp, ref_num, q_shuf, q_k, q_dr, q_comp, qry_num;
  memcpy(&r_shuf, rst, 4);
  memcpy(&r_k, rst + (ref_is_co ? 8 : 4), 4); memcpy(&r_dr, rst + (ref_is_co ? 12 : 8), 4);
  memcpy(&r_comp, rst + (ref_is_co ? 16 : 12), 4); memcpy(&ref_num, rst + (ref_is_co ? 20 : 16), 4);
  memcpy(&q_shuf, qst, 4); memcpy(&q_k, qst + 8, 4); memcpy(&q_dr, qst + 12, 4); memcpy(&q_comp, qst + 16, 4);
  memcpy(&qry_num, qst + 20, 4);
  (void)r_k; (void)r_dr;
  if (r_comp != q_comp || r_shuf != q_shuf) { free(rst); free(qst); return KO_ERR_ARG; } /* :944-949 */
  const uint32_t *ref_ct = (const uint32_t *)(rst + rh), *qry_ct = (const uint32_t *)(qst + 32);
  const char *refnames = (const char *)rst + rh + 4 * (size_t)ref_num, *qrynames = (const char *)qst + 32 + 4 * (size_t)qry_num;
  mkdir(outdir, 0700);
  uint32_t *ct = calloc((size_t)ref_num * (size_t)qry_num + 1, 4);
  int rc = KO_OK;
  for (int c = 0; c < r_comp && rc == KO_OK; c++) {
    size_t a = 0, b = 0, g = 0, x = 0;
    snprintf(path, sizeof path, "%s/combco.%d", qrydir, c);
    uint32_t *qids = (uint32_t *)ko_slurp(path, &a);
    snprintf(path, sizeof path, "%s/combco.index.%d", qrydir, c);
    uint64_t *qidx = (uint64_t *)ko_slurp(path, &b);
    if (!qids || !qidx) { free(qids); free(qidx); rc = KO_ERR_IO; break; }
    if (ref_is_co) {
      snprintf(path, sizeof path, "%s/combco.%d", refdir, c);
      uint32_t *rids = (uint32_t *)ko_slurp(path, &g);
      snprintf(path, sizeof path, "%s/combco.index.%d", refdir, c);
      uint64_t *ridx = (uint64_t *)ko_slurp(path, &x);
      if (!rids || !ridx) { free(rids); free(ridx); free(qids); free(qidx); rc = KO_ERR_IO; break; }
      uint32_t *gids = malloc(g + 4), *ri = NULL;
      uint64_t *re = NULL;
      size_t nr = 0;
      rc = ko_mco_build(rids, ridx, ref_num, gids, &ri, &re, &nr);
      if (rc == KO_OK) ko_mco_count(gids, NULL, ri, re, nr, qids, qidx, qry_num, qry_ct, ref_num, ct);
      free(rids); free(ridx); free(gids); free(ri); free(re);
    } else {
      snprintf(path, sizeof path, "%s/mco.%d", refdir, c);
      const uint32_t *gids = ko_map_file(path, &g);
      snprintf(path, sizeof path, "%s/mco.index.%d", refdir, c);
      const uint64_t *dense = ko_map_file(path, &x);
      if (!dense || x != (8ull << 32)) rc = KO_ERR_IO;
      else ko_mco_count(gids, dense, NULL, NULL, 0, qids, qidx, qry_num, qry_ct, ref_num, ct);
      if (gids) munmap((void *)gids, g);
      if (dense) munmap((void *)dense, x);
    }
    free(qids); free(qidx);
  }
  if (rc == KO_OK) {
    snprintf(path, sizeof path, "%s/distance.out", outdir);
    FILE *fp = fopen(path, "w");
    if (!fp) rc = KO_ERR_IO;
    else {
      rc = ko_dist_print(fp, o, q_k, q_dr, ref_num, qry_num, ref_ct, qry_ct, refnames, qrynames, ct); /* :972-973: the QUERY's k */
      fclose(fp);
    }
    if (rc == KO_OK && o->keep_shared) {                                /* :1633 removes the file otherwise */
      snprintf(path, sizeof path, "%s/sharedk_ct.dat", outdir);
      fp = fopen(path, "wb");
      if (!fp) rc = KO_ERR_IO;
      else { fwrite(ct, 4, (size_t)ref_num * (size_t)qry_num, fp); fclose(fp); }
    }
  }
  free(ct); free(rst); free(qst);
  return rc;
}
