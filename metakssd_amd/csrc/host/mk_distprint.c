/*
 * mk_distprint.c -- distance.out of the reference-database search (host side of SURVEY.md 8f N4).
 *
 * Follows dist_print_nobin() (command_dist.c:1531-1634) and output_ctrl() (:1636-1690): one line per (query, reference)
 * pair -- or per query its N best references -- with the shared / reference / query sketch sizes, Jaccard or containment,
 * the Mash / Aaf distance derived from it, and optionally a normal-approximation P-value, its Bonferroni product and 95 %
 * intervals.  double arithmetic through libm, printf formats as there, so the text is the same byte for byte.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "metakssd_hip.h"

#define MK_NAME_BYTES 256  /* PATHLEN: one name record of cofiles.stat / mcofiles.stat */
#define MK_MAX_NEIGB 1024  /* NREF, command_dist.c:1572 */

typedef struct {
  const mk_dist_opts *o;
  int kmerlen;
  double space_miss; /* 1 - 1/4^(kmerlen - dim_rd_len): the chance that one random k-mer misses a given one (:1642) */
  double pairs;      /* ref_num * qry_num as the reference computes it (:1562) */
} mk_dp_ctx;

/* GET_MATRIC (:1636): similarity -> evolutionary distance before the division by the k-mer length */
static double mk_dp_distance(const mk_dp_ctx *c, double sim) {
  return log(c->o->metric == 0 ? 1 / (2 * sim) + 0.5 : 1 / sim) / c->kmerlen;
}

/* the similarity the N-best selection sorts by (:1596-1598) */
static double mk_dp_similarity(const mk_dist_opts *o, unsigned x, unsigned y, unsigned shared) {
  if (o->metric == 1) return (double)shared / (x < y ? x : y);
  return (double)shared / (x + y - shared);
}

static void mk_dp_emit(FILE *fp, const mk_dp_ctx *c, const char *qname, const char *rname, unsigned x, unsigned y, unsigned shared) {
  double expected_by_chance = 0;
  if (c->o->correction) { /* :1639-1646 */
    const unsigned only_x = x - shared, only_y = y - shared;
    const double px = 1 - pow(c->space_miss, only_x), py = 1 - pow(c->space_miss, only_y);
    expected_by_chance = px * py * (only_x + only_y) / (px + py - 2 * px * py);
  }
  const unsigned denom = c->o->metric == 0 ? x + y - shared : (x < y ? x : y);
  const double sim = ((double)shared - expected_by_chance) / denom;
  double d = mk_dp_distance(c, sim);
  if (d > 1) d = 1;
  if (d > c->o->dthreshold) return; /* :1653 */
  char line[1024];
  size_t len = (size_t)snprintf(line, sizeof line, "%s\t%s\t%u-%u|%u|%u\t%.6lf\t%.6lf", qname, rname, shared,
                                (unsigned)expected_by_chance, x, y, sim, d);
  if (len >= sizeof line) len = sizeof line - 1;
  if (c->o->outfields >= 1) {
    const double sd = pow(sim * (1 - sim) / denom, 0.5);
    const double p = 0.5 * erfc(sim / sd * pow(0.5, 0.5));
    len += (size_t)snprintf(line + len, sizeof line - len, "\t%E\t%E", p, p * c->pairs);
    if (len >= sizeof line) len = sizeof line - 1;
    if (c->o->outfields >= 2) {
      const double lo = sim - 1.96 * sd, hi = sim + 1.96 * sd;
      len += (size_t)snprintf(line + len, sizeof line - len, "\t[%.6lf,%.6lf]\t[%.6lf,%.6lf]", lo, hi, mk_dp_distance(c, hi),
                              mk_dp_distance(c, lo));
      if (len >= sizeof line) len = sizeof line - 1;
    }
  }
  if (len + 1 < sizeof line) line[len++] = '\n';
  fwrite(line, 1, len, fp);
}

int mk_dist_print(void *out, const mk_dist_opts *o, int32_t kmerlen, int32_t dim_rd_len, uint32_t ref_num, uint32_t qry_num,
                  const uint32_t *ref_ctx_ct, const uint32_t *qry_ctx_ct, const char *refnames, const char *qrynames,
                  const uint32_t *ct) {
  static const char *const columns[2][3] = {{"Jaccard\tMashD", "P-value(J)\tFDR(J)", "Jaccard_CI\tMashD_CI"},
                                            {"ContainmentM\tAafD", "P-value(C)\tFDR(C)", "ContainmentM_CI\tAafD_CI"}};
  FILE *fp = (FILE *)out;
  if (!fp || !o || o->metric < 0 || o->metric > 1 || o->outfields < 0 || o->outfields > 2) return MK_ERR_ARG;
  if ((ref_num && (!ref_ctx_ct || !refnames)) || (qry_num && (!qry_ctx_ct || !qrynames)) || (ref_num && qry_num && !ct)) return MK_ERR_ARG;
  fputs("Qry\tRef\tShared_k|Ref_s|Qry_s", fp);
  for (int f = 0; f <= o->outfields; f++) fprintf(fp, "\t%s", columns[o->metric][f]);
  fputc('\n', fp);
  const int nbest = o->num_neigb;
  if (nbest < 0 || nbest > MK_MAX_NEIGB || (uint32_t)nbest > ref_num) return MK_ERR_ARG; /* :1574 */
  mk_dp_ctx c = {o, kmerlen, 1 - 1 / pow(4, kmerlen - dim_rd_len), (double)(long long)(ref_num * qry_num)};
  struct { double sim; long rid; } best[MK_MAX_NEIGB + 1];
  for (uint32_t q = 0; q < qry_num; q++) {
    const uint32_t *row = ct + (size_t)q * ref_num;
    const char *qname = qrynames + (size_t)q * MK_NAME_BYTES;
    const unsigned y = qry_ctx_ct[q];
    if (nbest == 0) {
      for (uint32_t r = 0; r < ref_num; r++) mk_dp_emit(fp, &c, qname, refnames + (size_t)r * MK_NAME_BYTES, ref_ctx_ct[r], y, row[r]);
      continue;
    }
    /* the N most similar references, ties to the earlier one, similarity 0 (and NaN) never listed (:1592-1611) */
    for (int i = 0; i < nbest; i++) { best[i].sim = 0; best[i].rid = -1; }
    for (uint32_t r = 0; r < ref_num; r++) {
      const double s = mk_dp_similarity(o, ref_ctx_ct[r], y, row[r]);
      int at = nbest;
      while (at > 0 && s > best[at - 1].sim) { best[at] = best[at - 1]; at--; }
      if (at < nbest) { best[at].sim = s; best[at].rid = (long)r; }
    }
    for (int i = 0; i < nbest; i++)
      if (best[i].rid >= 0)
        mk_dp_emit(fp, &c, qname, refnames + (size_t)best[i].rid * MK_NAME_BYTES, ref_ctx_ct[best[i].rid], y, row[best[i].rid]);
  }
  return ferror(fp) ? MK_ERR_IO : MK_OK;
}
