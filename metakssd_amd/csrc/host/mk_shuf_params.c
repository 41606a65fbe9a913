/*
 * mk_shuf_params.c -- .shuf format and derived parameters (host C, no GPU).
 *
 * Replaces: read_dim_shuffle_file() command_shuffle.c:215-235, write_dim_shuffle_file()
 * command_shuffle.c:174-213 (with a seeded generator instead of srand(time)), get_hashsz()
 * command_dist.c:286-315 and seq2co_global_var_initial() iseq2comem.c:54-86.
 */
#include "metakssd_hip.h"
#include "mk_host_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* format constants: global_basic.h:35-44, command_shuffle.h:20 */
#define MK_COMPONENT_SZ 8
#define MK_CTX_SPC_USE_L 8
#define MK_LD_FCTR 0.6
#define MK_MIN_SUBCTX_DIM_SMP_SZ 4096ULL

/* hash-table sizes: largest primes below 2^8..2^32 (global_basic.c:75-82) */
static const uint32_t mk_primes[25] = {
    251u,       509u,       1021u,      2039u,       4093u,       8191u,       16381u,      32749u,     65521u,
    131071u,    262139u,    524287u,    1048573u,    2097143u,    4194301u,    8388593u,    16777213u,  33554393u,
    67108859u,  134217689u, 268435399u, 536870909u,  1073741789u, 2147483647u, 4294967291u};

int mk_shuf_read(const char *path, mk_shuf *out) {
  if (!path || !out) return MK_ERR_ARG;
  size_t pl = strlen(path);
  if (pl < 5 || strcmp(path + pl - 5, ".shuf") != 0) return MK_ERR_FORMAT; /* command_shuffle.c:217-219 */
  FILE *f = fopen(path, "rb");
  if (!f) return MK_ERR_IO;
  int32_t hdr[4];
  if (fread(hdr, sizeof(int32_t), 4, f) != 4) { fclose(f); return MK_ERR_FORMAT; }
  if (hdr[2] < 0 || hdr[2] >= 8 || hdr[1] < hdr[2] || hdr[1] > 16) { fclose(f); return MK_ERR_FORMAT; }
  uint64_t len = 1ULL << (4 * hdr[2]);
  int32_t *t = (int32_t *)malloc(len * sizeof(int32_t));
  if (!t) { fclose(f); return MK_ERR_NOMEM; }
  if (fread(t, sizeof(int32_t), len, f) != len) { free(t); fclose(f); return MK_ERR_FORMAT; }
  fclose(f);
  out->id = hdr[0]; out->k = hdr[1]; out->subk = hdr[2]; out->drlevel = hdr[3];
  out->table = t;
  out->len = len;
  return MK_OK;
}

int mk_shuf_generate(int32_t k, int32_t subk, int32_t drlevel, uint64_t seed, mk_shuf *out) {
  /* constraints of the reference generator: k >= subk, subk < 8 (command_shuffle.c:176-181) */
  if (!out || subk < 1 || subk >= 8 || k < subk || k > 16 || drlevel < 0 || drlevel > subk) return MK_ERR_ARG;
  uint64_t len = 1ULL << (4 * subk);
  int32_t *t = (int32_t *)malloc(len * sizeof(int32_t));
  if (!t) return MK_ERR_NOMEM;
  for (uint64_t i = 0; i < len; i++) t[i] = (int32_t)i;
  /* Fisher-Yates, j drawn from a counter-based stream so the table depends only on (seed, subk) */
  for (uint64_t i = len - 1; i > 0; i--) {
    uint64_t j = mk_mix64(mk_mix64(seed) + i) % (i + 1);
    int32_t tmp = t[i]; t[i] = t[j]; t[j] = tmp;
  }
  out->id = (int32_t)(mk_mix64(seed ^ 0x6b737364ULL) & 0x7fffffffULL);
  out->k = k; out->subk = subk; out->drlevel = drlevel;
  out->table = t;
  out->len = len;
  return MK_OK;
}

int mk_shuf_write(const mk_shuf *s, const char *path) {
  if (!s || !s->table || !path) return MK_ERR_ARG;
  FILE *f = fopen(path, "wb");
  if (!f) return MK_ERR_IO;
  int32_t hdr[4] = {s->id, s->k, s->subk, s->drlevel};
  int ok = fwrite(hdr, sizeof(int32_t), 4, f) == 4 && fwrite(s->table, sizeof(int32_t), s->len, f) == s->len;
  if (fclose(f) != 0) ok = 0;
  return ok ? MK_OK : MK_ERR_IO;
}

void mk_shuf_free(mk_shuf *s) {
  if (s && s->table) { free(s->table); s->table = NULL; s->len = 0; }
}

int mk_params_init(const mk_shuf *s, mk_params *P) { return mk_params_init_csz(s, MK_COMPONENT_SZ, P); }

/* the same with the reference's compile-time COMPONENT_SZ (global_basic.h:35-37, `make alert` builds with -DCOMPONENT_SZ=8)
 * as a parameter: ids of one component live in [0, 16^component_sz) */
int mk_params_init_csz(const mk_shuf *s, int32_t component_sz, mk_params *P) {
  if (!s || !P || !s->table) return MK_ERR_ARG;
  if (component_sz < 1 || component_sz > 8) return MK_ERR_ARG; /* ids are 32-bit */
  int k = s->k, subk = s->subk, drl = s->drlevel;
  if (k < 1 || k > 16 || subk < 0 || subk >= 8 || subk > k || drl < 0 || drl > subk) return MK_ERR_FORMAT;
  if (s->len != (1ULL << (4 * subk))) return MK_ERR_FORMAT;
  memset(P, 0, sizeof *P);
  P->shuf_id = s->id; P->k = k; P->subk = subk; P->drlevel = drl;
  int pidx = 4 * (k - drl) - MK_CTX_SPC_USE_L - 7; /* command_dist.c:289 */
  if (pidx < 0 || pidx > 24) return MK_ERR_FORMAT;  /* reference aborts: command_dist.c:291-303 */
  P->hashsize = mk_primes[pidx];
  P->hashlimit = (uint32_t)((double)P->hashsize * MK_LD_FCTR); /* truncation toward zero, iseq2comem.c:61 */
  P->half_outctx_len = k - subk;
  P->TL = 2 * k;
  P->crvsaddmove = 4 * k - 2;
  P->component_sz = component_sz;
  if (k - drl - component_sz > 1) return MK_ERR_ARG; /* more than 16 components: the engine dumps at most 16 in one pass */
  P->component_num = (k - drl > component_sz) ? (int32_t)(1UL << (4 * (k - drl - component_sz))) : 1;
  P->comp_code_bits = (k - drl > component_sz) ? 4 * (k - drl - component_sz) : 0;
  P->tupmask = 0xffffffffffffffffULL >> (64 - 4 * k);
  P->domask = ((1ULL << (4 * subk)) - 1) << (2 * P->half_outctx_len);
  P->undomask = ((1ULL << (2 * P->half_outctx_len)) - 1) << (2 * (k + subk));
  P->dim_start = 0;
  uint64_t subspace = 1ULL << (4 * (subk - drl));
  P->dim_end = P->dim_start + (int32_t)(subspace > MK_MIN_SUBCTX_DIM_SMP_SZ ? subspace : MK_MIN_SUBCTX_DIM_SMP_SZ);
  P->shuf_table = s->table;
  P->shuf_len = s->len;
  return MK_OK;
}
