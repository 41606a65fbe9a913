/*
 * mk_sketchdir.c -- on-disk sketch directory writer (host C, no GPU).
 *
 * Replaces the file side of run_stageI(): the per-component concatenation loop
 * (command_dist.c:408-470: combco.<c>, combco.index.<c> = size_t[nfiles+1] cumulative element
 * counts, combco.<c>.a) and the cofiles.stat record (command_dist.c:477-500; co_dstat_t
 * global_basic.h:116-126; file-name prefixes command_set.c:236-237).  The reference first writes
 * per-file temporaries "<i>.co.<c>[.a]" and concatenates them; the bytes that end up in the
 * directory are the same when appended directly, which is what happens here.  The three padding
 * bytes after `bool koc` and the bytes after each path's NUL -- uninitialised in the reference --
 * are written as zero.
 */
#include "metakssd_hip.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>

#define MK_PATHLEN 256 /* global_basic.h:32 */

struct mk_sketchdir {
  char outdir[1024];
  mk_params P;
  int koc, nfiles, added;
  FILE **fid, **fab;
  uint64_t **index;   /* [component][nfiles+1] */
  uint32_t *ctx_ct;   /* per-file distinct-key count (ctx_ct_list) */
  char (*names)[MK_PATHLEN];
  uint64_t all_ctx_ct;
};

/* returns MK_ERR_IO when closing a sketch file fails (a full disk shows up at fclose, not at fwrite) */
static int mk_sd_free(mk_sketchdir *d) {
  if (!d) return MK_OK;
  int rc = MK_OK;
  int C = d->P.component_num;
  for (int c = 0; c < C; c++) {
    if (d->fid && d->fid[c] && fclose(d->fid[c]) != 0) rc = MK_ERR_IO;
    if (d->fab && d->fab[c] && fclose(d->fab[c]) != 0) rc = MK_ERR_IO;
    if (d->index) free(d->index[c]);
  }
  free(d->fid); free(d->fab); free(d->index); free(d->ctx_ct); free(d->names);
  free(d);
  return rc;
}

int mk_sketchdir_open(const char *outdir, const mk_params *p, int koc, int nfiles, mk_sketchdir **out) {
  if (!outdir || !p || !out || nfiles < 1 || strlen(outdir) >= sizeof(((mk_sketchdir *)0)->outdir) - 64) return MK_ERR_ARG;
  mk_sketchdir *d = (mk_sketchdir *)calloc(1, sizeof *d);
  if (!d) return MK_ERR_NOMEM;
  strcpy(d->outdir, outdir);
  d->P = *p;
  d->koc = koc ? 1 : 0;
  d->nfiles = nfiles;
  int C = p->component_num;
  d->fid = (FILE **)calloc(C, sizeof(FILE *));
  d->fab = (FILE **)calloc(C, sizeof(FILE *));
  d->index = (uint64_t **)calloc(C, sizeof(uint64_t *));
  d->ctx_ct = (uint32_t *)calloc(nfiles, sizeof(uint32_t));
  d->names = calloc(nfiles, MK_PATHLEN);
  if (!d->fid || !d->fab || !d->index || !d->ctx_ct || !d->names) { mk_sd_free(d); return MK_ERR_NOMEM; }
  if (mkdir(outdir, 0777) != 0 && errno != EEXIST) { mk_sd_free(d); return MK_ERR_IO; } /* command_dist.c:213 */
  char path[1200];
  for (int c = 0; c < C; c++) {
    d->index[c] = (uint64_t *)calloc((size_t)nfiles + 1, sizeof(uint64_t));
    snprintf(path, sizeof path, "%s/combco.%d", outdir, c);
    d->fid[c] = fopen(path, "wb");
    if (d->koc) {
      snprintf(path, sizeof path, "%s/combco.%d.a", outdir, c);
      d->fab[c] = fopen(path, "wb");
    }
    if (!d->index[c] || !d->fid[c] || (d->koc && !d->fab[c])) { mk_sd_free(d); return MK_ERR_IO; }
  }
  *out = d;
  return MK_OK;
}

int mk_sketchdir_add(mk_sketchdir *d, const char *input_path, const mk_result *r) {
  if (!d || !input_path || !r || d->added >= d->nfiles || r->component_num != d->P.component_num) return MK_ERR_ARG;
  if (strlen(input_path) >= MK_PATHLEN) return MK_ERR_ARG;
  int i = d->added;
  for (int c = 0; c < r->component_num; c++) {
    const mk_component *k = &r->components[c];
    if (k->n && fwrite(k->ids, sizeof(uint32_t), k->n, d->fid[c]) != k->n) return MK_ERR_IO;
    if (d->koc) {
      if (k->n && !k->counts) return MK_ERR_ARG;
      if (k->n && fwrite(k->counts, sizeof(uint16_t), k->n, d->fab[c]) != k->n) return MK_ERR_IO;
    }
    d->index[c][i + 1] = d->index[c][i] + k->n;
  }
  d->ctx_ct[i] = (uint32_t)r->total;
  d->all_ctx_ct += r->total;
  strcpy(d->names[i], input_path);
  d->added++;
  return MK_OK;
}

int mk_sketchdir_close(mk_sketchdir *d) {
  if (!d) return MK_ERR_ARG;
  int rc = MK_OK;
  char path[1200];
  if (d->added != d->nfiles) rc = MK_ERR_STATE;
  for (int c = 0; c < d->P.component_num && rc == MK_OK; c++) {
    snprintf(path, sizeof path, "%s/combco.index.%d", d->outdir, c);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(d->index[c], sizeof(uint64_t), (size_t)d->nfiles + 1, f) != (size_t)d->nfiles + 1) rc = MK_ERR_IO;
    if (f && fclose(f) != 0) rc = MK_ERR_IO;
  }
  if (rc == MK_OK) {
    unsigned char hdr[32]; /* co_dstat_t laid out by hand: offsets 0,4,8,12,16,20,24 */
    memset(hdr, 0, sizeof hdr);
    uint32_t u32 = (uint32_t)d->P.shuf_id;
    int32_t kmerlen = d->P.k * 2, dim_rd_len = d->P.drlevel * 2, comp_num = d->P.component_num, infile_num = d->nfiles;
    memcpy(hdr + 0, &u32, 4);
    hdr[4] = (unsigned char)d->koc;
    memcpy(hdr + 8, &kmerlen, 4);
    memcpy(hdr + 12, &dim_rd_len, 4);
    memcpy(hdr + 16, &comp_num, 4);
    memcpy(hdr + 20, &infile_num, 4);
    memcpy(hdr + 24, &d->all_ctx_ct, 8);
    snprintf(path, sizeof path, "%s/cofiles.stat", d->outdir);
    FILE *f = fopen(path, "wb");
    if (!f) rc = MK_ERR_IO;
    else {
      if (fwrite(hdr, 1, 32, f) != 32 || fwrite(d->ctx_ct, sizeof(uint32_t), d->nfiles, f) != (size_t)d->nfiles ||
          fwrite(d->names, MK_PATHLEN, d->nfiles, f) != (size_t)d->nfiles)
        rc = MK_ERR_IO;
      if (fclose(f) != 0) rc = MK_ERR_IO;
    }
  }
  const int crc = mk_sd_free(d);
  return rc != MK_OK ? rc : crc;
}
