/*
 * kssd_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the MetaKSSD `dist -L <.shuf> [-A]` sketching path and of the rows widened into after it
 * (`set`, `composite -q`, stage II and the `dist -r` search).  It is the parity checker for the HIP engine; nothing in
 * the product (metakssd_amd/, include/) may link, import or execute it.  Only tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg and the measurement / fuzz scripts under tools/ (as the checker and the timed CPU baseline) use it.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Parity is PINNED: oracle/check_vs_ref.py runs this restatement and the compiled reference
 * (oracle/_ref/metakssd, built from the reference's own sources by oracle/Makefile) on the same
 * inputs and compares the payload files byte for byte; tests/golden/ holds vectors produced by
 * the compiled reference (tests/golden/make_golden.py).
 */
#ifndef KSSD_ORACLE_H
#define KSSD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned long long ko_llong; /* global_basic.h:51 */

/* everything seq2co_global_var_initial() + get_hashsz() derive (iseq2comem.c:54-86, command_dist.c:286-315) */
typedef struct ko_params {
  int shuf_id, k, subk, drlevel; /* dim_shuffle_stat_t, command_shuffle.h:4-10 */
  int half_outctx_len;           /* iseq2comem.c:59 */
  int TL;                        /* iseq2comem.c:70 */
  int crvsaddmove;               /* iseq2comem.c:68 */
  int component_num;             /* iseq2comem.c:64-65 */
  int comp_code_bits;            /* iseq2comem.c:518 */
  int dim_start, dim_end;        /* iseq2comem.c:80-84 */
  unsigned int hashsize;         /* command_dist.c:288-305 */
  unsigned int hashlimit;        /* iseq2comem.c:61 */
  ko_llong tupmask, domask, undomask; /* iseq2comem.c:69,74-76 */
} ko_params;

enum { KO_OK = 0, KO_ERR_PRIMER = -1, KO_ERR_CROWDED = -2, KO_ERR_ARG = -3, KO_ERR_IO = -4, KO_ERR_CONTRACT = -5 };

int ko_params_derive(int shuf_id, int k, int subk, int drlevel, ko_params *out);

/* -A FASTQ path: mt_shortreads2koc() at -p 1 (iseq2comem.c:657-727).  co[] has hashsize slots and
 * is cleared here like the reference does (:663). */
int ko_koc_from_fastq_bytes(const ko_params *P, const int *shuf, const unsigned char *fq, size_t n, ko_llong *co,
                            ko_llong *nreads_out);
/* same per-read loop (iseq2comem.c:676-720) over already framed fixed-stride rows ('\n' terminated);
 * clear_first=0 continues filling the same table (the reference's batches do exactly that, :672-722). */
int ko_koc_from_rows(const ko_params *P, const int *shuf, const unsigned char *rows, size_t stride, size_t nreads,
                     ko_llong *co, int clear_first, unsigned int *keycount_io);
/* OpenMP form of the same loop, racy exactly like the reference (:675,:703-714); only for timing. */
int ko_koc_from_rows_omp(const ko_params *P, const int *shuf, const unsigned char *rows, size_t stride, size_t nreads,
                         ko_llong *co, int clear_first, int nthreads);

/* write_fqkoc2files() (iseq2comem.c:516-562) into memory: ids[c], cnts[c] must hold n_out[c] entries;
 * call once with ids==NULL to size (n_out filled), then again. */
unsigned int ko_dump_koc(const ko_params *P, const ko_llong *co, uint32_t **ids, uint16_t **cnts, size_t *n_out);

/* FASTQ without -A: fastq2co() (iseq2comem.c:323-419): 4-bit occurrence field, a key is marked (field = 0xf) once it
 * was seen M times (M = -n, 1..7) over bases whose quality byte is >= Q (-Q, raw character code).  The last record of a
 * file whose final line lacks its newline is NOT walked (:357: the record is read, feof() is already set). */
int ko_co_from_fastq_bytes(const ko_params *P, const int *shuf, const unsigned char *fq, size_t n, int Q, int M,
                           ko_llong *co, ko_llong *nlines_out);
/* write_fqco2file() (iseq2comem.c:596-621) into memory; two-call protocol like ko_dump_koc. */
unsigned int ko_dump_fqco(const ko_params *P, const ko_llong *co, uint32_t **ids, size_t *n_out);

/* FASTA path: fasta2co() (iseq2comem.c:218-315) / uniq_fasta2co() (:729-828) over a byte stream. */
int ko_co_from_fasta_bytes(const ko_params *P, const int *shuf, const unsigned char *fa, size_t n, ko_llong *co, int uniq);
/* wrt_co2cmpn_use_inn_subctx() (iseq2comem.c:625-652) into memory; same two-call protocol. */
unsigned int ko_dump_co(const ko_params *P, const ko_llong *co, uint32_t **ids, size_t *n_out);

/* ---- model of the multi-GPU shard merge (no reference counterpart: it is single-process) ----
 * ko_partial_from_rows: the distinct keys of one contiguous read range as {key, min(count,65535), ordinal of the
 *   first occurrence}, ordinal = (first_read_ordinal + row) << 12 | position of the k-mer's last base.
 * ko_layout_from_partials: merges such lists (counts add then clamp, first ordinals take min) and inserts the keys
 *   into co[] in first-ordinal order with the reference's insert (iseq2comem.c:701-718).  The claim the tests
 *   check: this equals ko_koc_from_rows over the whole input, i.e. sharding does not change a byte. */
int ko_partial_from_rows(const ko_params *P, const int *shuf, const unsigned char *rows, size_t stride, size_t nreads,
                         ko_llong first_read_ordinal, ko_llong *keys, unsigned int *counts, ko_llong *ords, size_t cap,
                         size_t *n_out);
int ko_layout_from_partials(const ko_params *P, int nparts, const ko_llong *const *keys, const unsigned int *const *counts,
                            const ko_llong *const *ords, const size_t *n, ko_llong *co);

/* read_dim_shuffle_file() (command_shuffle.c:215-235) */
int ko_shuf_read(const char *path, int header[4], int **table_out, size_t *len_out);

/* whole `dist -L shuf [-A] [-u] -o outdir files...` stage I at -p 1, files in the GIVEN order
 * (the reference permutes them with a time-seeded shuffle, command_dist.c:215) :
 * run_stageI() command_dist.c:341-500.  cofiles.stat padding and post-NUL path bytes are zeroed. */
int ko_dist_stage1(const char *shuf_path, int abundance, int uniq, const char *outdir, int nfiles, const char **files);
/* same with -Q / -n (used by the FASTQ-without--A path only; defaults 0 / 1, command_dist_wrapper.c:79-80,169-185) */
int ko_dist_stage1_ex(const char *shuf_path, int abundance, int uniq, int Q, int M, const char *outdir, int nfiles,
                      const char **files);

/* ---- `set -u` / `set -q` ----
 * sketch_union() (command_set.c:241-319) / uniq_sketch_union() (:427-512) for one component: the ids of all the
 * sketches of a combined sketch file go through a 2^32-bit dictionary and come out ascending; uniq keeps the ids
 * that occur exactly once in the whole file.  out may be NULL (count only); returns the count, (size_t)-1 = no memory. */
size_t ko_set_union(const uint32_t *ids, size_t n, int uniq, uint32_t *out);
/* the whole command on a sketch directory: cofiles.stat header copied as it is, pan.N / uniq_pan.N written.
 * answer_yes: the reply to "only 1 sketch, use ... as pan-sketch?(Y/N)" when infile_num == 1 (renames in place). */
int ko_set_stage(const char *indir, const char *outdir, int uniq, int answer_yes);
/* `set -i <pan>` / `set -s <pan>`: sketch_operate() (command_set.c:321-425).  ko_set_filter: the ids that are
 * (keep_members=1) / are not (0) in the dictionary built from `pan`, input order kept; returns the count.
 * ko_set_operate: the whole command (combco.N, combco.index.N, cofiles.stat with recounted per-file sizes; the header's
 * all_ctx_ct and koc are left untouched and no .a files are written, exactly like the reference). */
size_t ko_set_filter(const uint32_t *pan, size_t npan, int keep_members, const uint32_t *ids, size_t n, uint32_t *out);
int ko_set_operate(const char *indir, const char *pandir, const char *outdir, int intersect);
/* `set -g <file.tsv>`: grouping_genomes() (command_set.c:831-974) with organize_taxf() (:635-705).  Per taxon the ids of its
 * genomes (file order) go through an FCFS double-hashing table of ko_group_table_size(total ids) slots and come out in
 * slot order; ko_group_layout is that table (32-bit unsigned probe arithmetic, id 0 never stored, ids that find no
 * place within table_size probes are dropped -- the table can be SMALLER than the id count, see :871). */
size_t ko_group_layout(const uint32_t *ids, size_t n, uint32_t table_size, uint32_t *out);
uint32_t ko_group_table_size(uint64_t total_ids);
int ko_set_group(const char *indir, const char *taxfile, const char *outdir);

/* ---- `composite -r <ref> -q <qry> [-b] [-o dir]`: get_species_abundance() (command_composite.c:446-649).
 * For every query sketch and reference sketch: the query's counts of the k-mers they share, then per reference (by
 * decreasing number of shared k-mers, at least 6) the line "<qry>\t<ref>\t<n>\t<mean>\t<98-99 percentile mean>\t<median>\t<max>"
 * on `out`, or with binvec the normalised .abv vector file.  float arithmetic as in the reference. */
#include <stdio.h>
int ko_composite(const char *refdir, const char *qrydir, const char *outdir, int binvec, FILE *out);

/* ---- stage II (combco2mco, co2mco.c:12-87) and the `dist -r <mco> <co>` search (command_dist.c:902-1079,
 * output :1531-1690).  ko_mco_build: the inverted index of one component -- gids_out[index[cofnum]] = the genome numbers
 * row by row (rows = k-mer ids ascending, a row in genome order), row_ids / row_ends (malloc'ed) = the non-empty rows and
 * their cumulative ends, i.e. the places where the reference's dense 2^32-entry mco.index.N changes value. */
int ko_mco_build(const uint32_t *ids, const uint64_t *index, int cofnum, uint32_t *gids_out, uint32_t **row_ids,
                 uint64_t **row_ends, size_t *nrows);
/* run_stageII(): mcofiles.stat, mco.N and (dense_index != 0) the 32 GiB mco.index.N per component */
int ko_stage2(const char *codir, const char *mcodir, int dense_index);
/* ct[qry_num x ref_num] += shared k-mer counts of one component; `dense` (the reference's index) or the sparse rows */
void ko_mco_count(const uint32_t *gids, const uint64_t *dense, const uint32_t *row_ids, const uint64_t *row_ends, size_t nrows,
                  const uint32_t *qry_ids, const uint64_t *qry_index, int qry_num, const uint32_t *qry_ctx_ct, int ref_num,
                  uint32_t *ct);
typedef struct ko_dist_opts {
  int metric;        /* -M 0 Jaccard / 1 containment            (command_dist_wrapper.c:85) */
  int outfields;     /* -O 0 distance / 1 + p,q values / 2 + CI (:86, default 2) */
  int correction;    /* --correction                            (:87) */
  double dthreshold; /* -D, default 1                           (:84) */
  int num_neigb;     /* -N, 0 = all                             (:83) */
  int keep_shared;   /* --keepskf: keep sharedk_ct.dat          (:92) */
} ko_dist_opts;
int ko_dist_print(FILE *fp, const ko_dist_opts *o, int kmerlen, int dim_rd_len, int ref_num, int qry_num,
                  const uint32_t *ref_ctx_ct, const uint32_t *qry_ctx_ct, const char *refnames, const char *qrynames,
                  const uint32_t *ct);
int ko_dist_search(const char *refdir, int ref_is_co, const char *qrydir, const char *outdir, const ko_dist_opts *o);

#ifdef __cplusplus
}
#endif
#endif
