/*
 * metakssd_hip.h -- C ABI of libmetakssd_hip.so, the MI355X (gfx950) engine for MetaKSSD's
 * k-mer sketching hot path (`metakssd dist -L <.shuf> [-A]`, FASTQ/FASTA -> sketch).
 *
 * The reference has no FFI layer: the seam is a pair of C functions over file-static globals
 * (SURVEY.md 8b).  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout):
 *
 *   reference                                               this ABI
 *   ------------------------------------------------------  ---------------------------------------
 *   read_dim_shuffle_file()      command_shuffle.c:215-235   mk_shuf_read / mk_shuf_free
 *   write_dim_shuffle_file()     command_shuffle.c:174-213   mk_shuf_generate / mk_shuf_write (seeded)
 *   get_hashsz()                 command_dist.c:286-315   }  mk_params_init
 *   seq2co_global_var_initial()  iseq2comem.c:54-86       }
 *   CO[tid]=malloc(hashsize*8)   command_dist.c:344-348      mk_engine_create / mk_engine_destroy
 *   memset(co,0,..)              iseq2comem.c:663            mk_sketch_begin
 *   mt_shortreads2koc() loop     iseq2comem.c:675-721        mk_sketch_push_reads[_device]  (MK_MODE_KOC)
 *   fastq2co() loop              iseq2comem.c:349-413        mk_sketch_begin_occ + same push (MK_MODE_OCC_SET)
 *   fasta2co()/uniq_fasta2co()   iseq2comem.c:218-315,729-828  same entry points on overlapped windows
 *                                                            (MK_MODE_SET / MK_MODE_UNIQ_SET)
 *   write_fqkoc2files()          iseq2comem.c:516-562     }  mk_sketch_finish -> mk_result
 *   wrt_co2cmpn_use_inn_subctx() iseq2comem.c:625-652     }
 *   write_fqco2file()            iseq2comem.c:596-621     }
 *   err(errno,"...too crowd")    iseq2comem.c:708-709        MK_ERR_CROWDED (never exit() in here)
 *   -- (no counterpart: single process; the thread team of mt_shortreads2koc, iseq2comem.c:675-720, spread over GPUs) --
 *                                                            mk_partial_count/export/import (merge by gather),
 *                                                            mk_partial_export_split/restart/list_reserve/adopt/commit (merge by key slices)
 *   sketch_union()/uniq_sketch_union() dictionaries  command_set.c:279-316,466-509   mk_setop_begin/add/finish
 *   sketch_operate() membership filter               command_set.c:361-405           mk_setop_filter
 *   grouping_genomes() per-taxon table               command_set.c:866-915           mk_setop_group
 *   get_species_abundance() dictionary + lookups     command_composite.c:525-553     mk_setop_join
 *   combco2mco() inverted index                      co2mco.c:37-66                  mk_mco_build / mk_mco_index_rows
 *   mco_cbdco_nobin_dist() counting loop             command_dist.c:1033-1049        mk_mco_count_begin/add/finish
 *   dist_print_nobin() / output_ctrl()               command_dist.c:1531-1690        mk_dist_print
 *
 * Conventions: plain pointers and sizes only; every function returns MK_OK (0) or a negative
 * MK_ERR_* code and never calls exit(); mk_last_error() gives the text.  One engine per GPU;
 * calls on one engine must be serialised by the caller.  Nothing in this library falls back to a
 * CPU implementation: without a usable HIP device mk_engine_create fails with MK_ERR_NO_DEVICE.
 */
#ifndef METAKSSD_HIP_H
#define METAKSSD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_ABI_VERSION 1

enum {
  MK_OK = 0,
  MK_ERR_ARG = -1,       /* bad argument / unsupported parameter combination */
  MK_ERR_NO_DEVICE = -2, /* no HIP device, or device index out of range */
  MK_ERR_HIP = -3,       /* a HIP runtime call failed (text in mk_last_error) */
  MK_ERR_CROWDED = -4,   /* more than hashlimit distinct keys: iseq2comem.c:708-709 */
  MK_ERR_STATE = -5,     /* call out of order (push before begin, ...) */
  MK_ERR_IO = -6,
  MK_ERR_FORMAT = -7, /* malformed .shuf / primer index out of 0..24 (command_dist.c:291-303) */
  MK_ERR_NOMEM = -8
};

/* sketch flavours (which reference function pair is being replaced) */
enum {
  MK_MODE_KOC = 0,     /* -A FASTQ: mt_shortreads2koc + write_fqkoc2files: ids + 16-bit counts      */
  MK_MODE_SET = 1,     /* FASTA:    fasta2co + wrt_co2cmpn_use_inn_subctx: ids, key 0 never stored  */
  MK_MODE_UNIQ_SET = 2, /* FASTA -u: uniq_fasta2co: keys seen more than once are dropped at the dump */
  MK_MODE_OCC_SET = 3   /* FASTQ without -A: fastq2co + write_fqco2file: ids of the keys seen at least
                           min_occurrence times (-n), key 0 kept; see mk_sketch_begin_occ            */
};

/* ---- .shuf (command_shuffle.h:4-16) -------------------------------------------------------- */
typedef struct mk_shuf {
  int32_t id, k, subk, drlevel; /* dim_shuffle_stat_t, 16-byte file header */
  int32_t *table;               /* permutation of 0 .. 16^subk-1 */
  uint64_t len;                 /* 16^subk */
} mk_shuf;

int mk_shuf_read(const char *path, mk_shuf *out);
/* Format-compatible deterministic generator (the reference's is srand(time)): Fisher-Yates driven by
 * a counter-based splitmix64 stream; id = low 31 bits of mix(seed). */
int mk_shuf_generate(int32_t k, int32_t subk, int32_t drlevel, uint64_t seed, mk_shuf *out);
int mk_shuf_write(const mk_shuf *s, const char *path);
void mk_shuf_free(mk_shuf *s);

/* ---- derived parameters ------------------------------------------- */
typedef struct mk_params {
  int32_t shuf_id, k, subk, drlevel;
  int32_t half_outctx_len; /* k - subk                         iseq2comem.c:59 */
  int32_t TL;              /* 2k                               iseq2comem.c:70 */
  int32_t crvsaddmove;     /* 4k-2                             iseq2comem.c:68 */
  int32_t component_num;   /* iseq2comem.c:64-65 */
  int32_t comp_code_bits;  /* iseq2comem.c:518 */
  int32_t dim_start, dim_end; /* iseq2comem.c:80-84 */
  uint32_t hashsize;       /* primer[4(k-drlevel)-15]          command_dist.c:288-305 */
  uint32_t hashlimit;      /* (uint)(hashsize*0.6)             iseq2comem.c:61 */
  uint64_t tupmask, domask, undomask; /* iseq2comem.c:69,74-76 */
  const int32_t *shuf_table; /* host pointer, 16^subk entries; must outlive mk_engine_create */
  uint64_t shuf_len;
  int32_t component_sz;      /* COMPONENT_SZ, global_basic.h:35-37: 8 unless the reference was built with -DCOMPONENT_SZ=n */
  int32_t reserved;
} mk_params;

int mk_params_init(const mk_shuf *shuf, mk_params *out);
/* for sketch directories of a reference build with another -DCOMPONENT_SZ (1..8): component_num = 16^(k-drlevel-component_sz),
 * ids below 16^component_sz, stage II index files of 16^component_sz rows.  MK_ERR_ARG above 16 components. */
int mk_params_init_csz(const mk_shuf *shuf, int32_t component_sz, mk_params *out);

/* ---- engine ---------------------------------------------------------------------------------- */
typedef struct mk_engine mk_engine;

typedef struct mk_component {
  uint32_t *ids;    /* host memory, engine-owned, reference slot order */
  uint16_t *counts; /* NULL unless MK_MODE_KOC */
  uint64_t n;
} mk_component;

typedef struct mk_result {
  int32_t component_num;
  uint64_t total;           /* what write_fqkoc2files()/wrt_co2cmpn_use_inn_subctx() return */
  mk_component *components; /* component_num entries; valid until mk_result_release / next begin */
} mk_result;

/* kernel-time accounting (hipEvent pairs on the engine's stream), filled while profiling is on */
typedef struct mk_profile {
  double scan_ms;        /* sum over scan-kernel launches since mk_profile_reset */
  uint64_t scan_launches;
  double resolve_ms;     /* candidate resolution launches (canonical k-mer, exact .shuf check, upsert) */
  double clear_ms;       /* table clear in mk_sketch_begin */
  double finish_ms;      /* compaction + priority layout + ordered dump (device part of finish); for mk_sketch_finish_begin: the
                          * part on the engine's stream (compaction) */
  uint64_t bases_scanned; /* sum of nreads*stride handed to scan launches (row bytes, not bases) */
  uint64_t rows_scanned;
  double finish_side_ms; /* mk_sketch_finish_begin: layout + dump + copy to the host on the side stream, beside the next sketch */
} mk_profile;

int mk_device_count(int *n);
int mk_engine_create(const mk_params *p, int device, mk_engine **out);
/* MK_ENGINE_LAZY_TABLES: the hashsize-slot tables, the key list and the dump's arrays (2.6 GB at L3K11, 21 GB at L2K11) are made by the
 * first mk_sketch_begin instead of here.  For an engine that mostly sketches BATCHES of files (mk_sketch_batch_*: a small table per
 * file, nothing of the big ones is touched): a process that only runs batches then starts 30 ms sooner at L2K11 and leaves 21 GB less
 * for the driver to take back when it exits -- which the next process would wait for.  A file that falls out of its batch is sketched
 * alone through mk_sketch_begin and pays for the tables then. */
enum { MK_ENGINE_LAZY_TABLES = 1u };
int mk_engine_create_ex(const mk_params *p, int device, unsigned flags, mk_engine **out);
int mk_engine_destroy(mk_engine *e);
/* Run all engine work on a caller-owned hipStream_t (e.g. torch's current stream) so that it is ordered with the
 * caller's own work on that stream.  NULL selects HIP's default stream (that is what torch.cuda.current_stream()
 * normally is), NOT the engine's own stream: mk_engine_use_own_stream() goes back to that (the initial state). */
/* Engine options, settable between sketches (not between begin and finish):
 *   MK_OPT_SPARSE   -1 / 0 / 1: dirty-block bookkeeping of the table passes by table size (default: on from 2^26 slots) / off / on
 *   MK_OPT_CAND_CAP records per scan wave in the candidate append buffers (default 8192; 0 = resolve every filter hit inline)
 *   MK_OPT_RESULT_CAP entries the pinned result arrays hold now (default: 2 M at the first finish; a larger sketch grows them
 *                   and writes its result a second time).  Setting it, like every later mk_sketch_finish on the engine,
 *                   INVALIDATES the arrays an earlier mk_result pointed to (they are the engine's: copy what must outlive
 *                   the next sketch)
 *   MK_OPT_DIRECT_HOST 0 / 1: host pushes of rows in pinned memory are scanned in place over PCIe (one scan launch per push,
 *                   no staging copy) instead of being copied into staging regions first (default 0)
 *   MK_OPT_FRONT_BITS -1 / 0 / 3..28: a small accumulation table of 2^n slots in front of the hashsize-slot one, so that the
 *                   per-sketch passes (clear, compaction) cost what the sketch holds, not what the reference's table could
 *                   hold; -1 (default): about an eighth of hashsize, from 2^20 slots up (dense bookkeeping only), 0: none
 *   MK_OPT_KEYLIST_CAP entries the distinct-key list holds now (sparse bookkeeping only: there the list starts at 32 M entries instead
 *                   of hashsize -- 10.7 GB at L2K11 -- and the finish / export that counts more keys grows it and compacts again)
 *   MK_OPT_BATCH_TAB_BITS 0 / 9..22: log2 of the slots every file's table gets in mk_sketch_batch_begin (default 0: about five times
 *                   the keys the batch's largest file is expected to leave); a file whose table is too small is sketched alone
 *   MK_OPT_SPLIT_CUS 0 / 32, 64 .. half the device's compute units: two hardware queues with complementary CU masks -- the scan kernel on
 *                   all but this many compute units, everything that follows a scan (candidate resolution, compaction, clears) on
 *                   these, ordered by events.  One engine gains nothing (its kernels depend on each other); TWO engines taking
 *                   sketches in turn do: what follows engine A's scan runs beside engine B's scan instead of in front of it
 *                   (device-resident rows only, DESIGN.md 4.4: 2.25-2.35 instead of 2.38-2.42 ms a pass of 50 M reads with 32 over hundreds of
 *                   sketches, 0-3 % over twenty, a loss on a throttled GPU -- an option; bench.py times it as a side leg, never as its headline).
 *                   Layout, dump and the result copy of mk_sketch_finish_begin run on the scan queue's units under this option.
 *                   Default 0: one queue, every kernel on the whole device
 *   (8 and 9 were MK_OPT_ROWS160 and MK_OPT_BATCH_QUEUES in round 4: a scan kernel that kept a lane's text row in registers and a
 *   second queue for every other batch.  Both measured slower than what they replaced -- profiles/r04_b_kernel_stats_rows160_variant.csv,
 *   profiles/r04_d_config5_two_queues.txt -- and were removed in round 5; the numbers stay retired.)
 * Results are bit-identical for every setting; the tests run both. */
enum { MK_OPT_SPARSE = 1, MK_OPT_CAND_CAP = 2, MK_OPT_RESULT_CAP = 3, MK_OPT_DIRECT_HOST = 4, MK_OPT_FRONT_BITS = 5, MK_OPT_KEYLIST_CAP = 6,
       MK_OPT_BATCH_TAB_BITS = 7, MK_OPT_SPLIT_CUS = 10 };
int mk_engine_set_option(mk_engine *e, int option, int64_t value);
int mk_engine_set_stream(mk_engine *e, void *hip_stream);
int mk_engine_use_own_stream(mk_engine *e);
/* MK_OPT_SPLIT_CUS on two engines of one device: e's scan kernels go to WITH's scan queue, one after the other in the order they are
 * pushed, instead of to a second queue with the same CU mask (where the workgroups of both scans compete for the units and a kernel's
 * duration includes its wait for them).  WITH owns the queue and keeps it while it is lent: setting WITH back to one queue or destroying
 * it fails with MK_ERR_STATE until e has been destroyed or set back to one queue. */
int mk_engine_share_scan_queue(mk_engine *e, mk_engine *with);
const char *mk_last_error(const mk_engine *e); /* e may be NULL: last error of a failed create */

/* mode | MK_BEGIN_NOTHING_FOLLOWS (engines with MK_OPT_SPLIT_CUS; ignored elsewhere): no other engine's scan is queued behind this sketch --
 * the last input of a run.  What follows its scan (candidate resolution, compaction, ..) then runs on the engine's unmasked queue, i.e. on the
 * whole device, instead of on the second queue's few compute units: the run's tail is 0.8 ms instead of 2.3 ms.  Same result. */
enum { MK_BEGIN_NOTHING_FOLLOWS = 0x100 };
int mk_sketch_begin(mk_engine *e, int mode); /* MK_MODE_OCC_SET here means min_occurrence 1 */
/* fastq2co(…, Q, M) (iseq2comem.c:323-419): MK_MODE_OCC_SET with M = min_occurrence, 1 <= M < 15 (:325; the CLI
 * clamps -n to 1..7, command_dist_wrapper.c:169-180).  The quality threshold Q is applied by the front end
 * (mk_fastq_frame_q).  fastq2co() never advances its key counter (:404), so it does not abort at hashlimit: this
 * flavour reports MK_ERR_CROWDED only when the table itself is full (distinct keys >= hashsize). */
int mk_sketch_begin_occ(mk_engine *e, int min_occurrence);
/* Row pitch.  Any multiple of 4 in 4..4096 is accepted; multiples of 16 take the 16-byte staging path.  AVOID MULTIPLES OF 128:
 * the 64 rows of a tile then start in the same few L2 channels and the scan runs at half its rate (7.5 Gbases of 250-base
 * reads: 4.48 ms at a pitch of 256, 2.69 ms at 272; the same at 128, 384, 512, 640 -- profiles/r02_c_probe_read_length.json).
 * MK_ROW_PITCH(bytes incl. the newline) is what this library's own front ends use. */
#define MK_ROW_PITCH(need) ((((need) + 15u) & ~15u) + (((((need) + 15u) & ~15u) & 127u) == 0u && (((need) + 15u) & ~15u) < 4096u ? 16u : 0u))
/* Fixed-stride rows, each an ASCII sequence line terminated by '\n' (the layout of the reference's
 * fq_buff[l][FQ_LEN], iseq2comem.c:659,673); a row without '\n' ends at `stride`.  stride % 4 == 0,
 * 4 <= stride <= 4096.  Read i of this call has global ordinal first_read_ordinal + i: ordinals define
 * the sequential order whose table layout the result reproduces, so they must increase in file order.
 * Host variant: returns once `rows` may be reused (device copy done; the scan may still be running). */
int mk_sketch_push_reads(mk_engine *e, const uint8_t *rows, uint32_t stride, uint64_t nreads, uint64_t first_read_ordinal);
/* PACKED rows: `stride` = MK_PACKED_PITCH | MK_ROWS_PACKED in any of the push calls.  A row is then 64 bytes for a read of up to
 * MK_PACKED_MAX_BASES bases: 2 bits a base in the scan kernel's own coding ((byte >> 1) & 3: A0 C1 T2 G3) and one validity bit a
 * base (is it one of ACGTacgt) -- what the scan kernel's first instructions make of an ASCII row, done by the framer thread
 * instead, so that a 150-base read crosses PCIe as 64 bytes, not 160.  Little-endian dwords: [0] = bases | (every base valid) << 16;
 * [1 + p] = the codes of bases 16p .. 16p + 15, first base in the top two bits; bytes 44 + w = validity of bases 8w .. 8w + 7 (bit j:
 * base 8w + j).  The sketch is the one the ASCII rows give (a byte that is no base resets the k-mer window either way,
 * iseq2comem.c:682-690).  Only the geometries with a tuned scan kernel take packed rows (mk_params_packed_ok); the library's FASTQ
 * stream makes them with mk_fastq_opts.packed, mk_pack_rows_host converts ASCII rows. */
#define MK_ROWS_PACKED 0x80000000u
#define MK_PACKED_PITCH 64u
#define MK_PACKED_MAX_BASES 152u
int mk_params_packed_ok(const mk_params *p);
int mk_pack_rows_host(const uint8_t *rows, uint32_t stride, uint64_t nrows, uint8_t *packed /* nrows * MK_PACKED_PITCH bytes */);
/* A whole FASTA file -> packed rows on the host: the reference's walk (iseq2comem.c:240-279: '\n' and '\r' skipped without a reset,
 * a '>' line skipped and a reset, any other byte that is no base a reset) gives the base stream, which is cut into overlapping rows
 * of CAP stream bytes at a distance of CAP + 1 - TL, so that every TL-byte window starts in exactly one row, in file order.
 *   format MK_ROWS_PACKED: CAP = MK_PACKED_MAX_BASES; the rows of the push calls (pushed with ascending ordinals they give the sketch
 *                          mk_sketch_push_stream gives for the text).
 *   format MK_ROWS_WIDE:   CAP = MK_WIDE_MAX_BASES (240) in the same 64 bytes: dword 0 = bases | every base valid << 16 | an extension
 *                          row follows << 17, dwords 1..15 the codes; a row with a byte that is no base is followed by its EXTENSION
 *                          ROW (dword 0 = 1 << 18, i.e. zero bases; bytes 16..45 = the validity bytes of the row in front).  0.31
 *                          bytes a base on PCIe.  Taken by mk_sketch_batch_begin_rows only.
 * mk_fasta_pack_bound(n, TL, format) rows always suffice for n bytes of text.  MK_ERR_FORMAT: the text ends inside a '>' line (the
 * reference's abort, :259-271).  `rows` 16-byte aligned.  Host code only. */
#define MK_ROWS_WIDE 0x40000000u
#define MK_WIDE_MAX_BASES 240u
uint64_t mk_fasta_pack_bound(size_t n, int32_t TL, uint32_t format);
int mk_fasta_pack_rows(const uint8_t *text, size_t n, int32_t TL, uint32_t format, uint8_t *rows, uint64_t max_rows, uint64_t *nrows);
int mk_sketch_push_reads_device(mk_engine *e, const uint8_t *rows_dev, uint32_t stride, uint64_t nreads,
                                uint64_t first_read_ordinal);
/* Asynchronous host variant: returns once the copies are queued; `rows` must stay untouched until
 * mk_sketch_push_wait(e, *ticket) has returned (mk_sketch_finish does not wait for tickets).  Up to 16 pushes may be in
 * flight; the oldest is waited for inside the call beyond that.  mk_sketch_push_reads = async + wait. */
int mk_sketch_push_reads_async(mk_engine *e, const uint8_t *rows, uint32_t stride, uint64_t nreads, uint64_t first_read_ordinal,
                               uint64_t *ticket);
int mk_sketch_push_wait(mk_engine *e, uint64_t ticket);
/* FASTA text as it is in the file (config 5; what SURVEY.md 8b calls mk_sketch_push_stream): the engine copies the bytes to
 * the device and does there what fasta2co() / uniq_fasta2co() do while they walk their 64 KiB window (iseq2comem.c:240-279,
 * :751-790): '\n' and '\r' are skipped without resetting the k-mer window, a '>' skips to the end of its line and resets, any
 * other byte that is no base resets.  The k-mers (which may span line breaks) are then taken from overlapping rows of the
 * resulting base stream in HBM -- no host byte loop (mk_fasta_window remains as the host-side equivalent; both give the same
 * sketch).  Text may be pushed in pieces (each < 2^31 bytes): final == 0 for every piece but the last of the sketch, final != 0
 * for the last (n == 0 is allowed).  `text` may be reused when the call returns for pageable memory; pinned memory must stay
 * untouched until the next call on this engine that waits (a non-final push, mk_sketch_finish, mk_engine_sync).  A stream that
 * ends inside a header line is the reference's "can not find seqences head start from '>'" abort: mk_sketch_finish returns
 * MK_ERR_FORMAT.  May be mixed with mk_sketch_push_reads* in one sketch; ordinals follow the order of the calls. */
int mk_sketch_push_stream(mk_engine *e, const uint8_t *text, uint64_t n, int final);
int mk_sketch_finish(mk_engine *e, mk_result *out);
/* mk_sketch_finish in two halves, for callers that sketch one input after another: _begin runs compaction, layout and dump (into
 * staging arrays in HBM), waits once for the counters and queues the copy of the result to the host on a stream of its own;
 * when it returns the sketch is over and the NEXT one may be begun and pushed -- its kernels run beside that copy.  With dense
 * bookkeeping (tables below 2^26 slots) _begin keeps only the compaction on the engine's stream: priority layout, dump and copy are
 * queued on the side stream, so a pass over resident reads costs clear + scan + resolve + compaction.  _end waits for the side
 * stream and hands out the result.  One result may be outstanding: _end comes before the next mk_sketch_finish /
 * mk_sketch_finish_begin / mk_partial_* on the engine.  Errors of the sketch (MK_ERR_CROWDED, MK_ERR_FORMAT) are returned by
 * _begin. */
int mk_sketch_finish_begin(mk_engine *e);
int mk_sketch_finish_end(mk_engine *e, mk_result *out);
int mk_result_release(mk_engine *e, mk_result *r);
int mk_engine_sync(mk_engine *e);

/* ---- many small inputs in one launch sequence (BASELINE config 5: a directory of genomes) ------------------------------------
 * run_stageI() sketches one file per OpenMP thread, each thread with a table of its own (command_dist.c:344-348, :363-372:
 * CO[tid], fasta2co() / uniq_fasta2co(), wrt_co2cmpn_use_inn_subctx()).  On the device a 4 Mbase genome alone is a dozen
 * launches of a few microseconds each; here up to MK_BATCH_MAX_FILES files travel together: their text in one copy, the FASTA
 * walk, the scan, the tables (one per file, side by side), the reference's slot order (a virtual hashsize-slot table per file)
 * and the ordered dump in ONE launch sequence for all of them.  Every file's sketch is exactly what mk_sketch_begin(mode) +
 * mk_sketch_push_stream(text, n, 1) + mk_sketch_finish give for it alone (a file the batch's tables cannot take is sketched that
 * way by mk_sketch_batch_end itself).
 *   mode: MK_MODE_SET or MK_MODE_UNIQ_SET.  files[i].text: the file's bytes (FASTA text as in the file), 1 <= nfiles <=
 *   MK_BATCH_MAX_FILES, each at most MK_BATCH_FILE_MAX bytes, MK_BATCH_TEXT_MAX in all.  The texts must stay untouched until
 *   the matching mk_sketch_batch_end has returned.  When every text starts where the one in front of it ends, rounded up to
 *   a multiple of 1024 bytes (one pinned buffer filled file by file), the texts cross PCIe in ONE copy.
 *   Two batches may be in flight: _begin queues everything and returns, _end waits for the OLDEST batch and hands out one
 *   result per file -- status MK_OK, MK_ERR_FORMAT (the text ends inside a '>' line: the reference's abort, iseq2comem.c:259-271)
 *   or MK_ERR_CROWDED; the arrays are the engine's and stay valid until the next mk_sketch_batch_end on it.
 *   Not between mk_sketch_begin and mk_sketch_finish.
 *   The slow path, for the record: a file that falls out of its batch (its table too small, more keys than half of it, a layout that
 *   runs out of steps) is sketched alone INSIDE mk_sketch_batch_end, on the engine's queue -- behind the next batch when one has
 *   been begun already, so that call then returns a batch's time later than it would have; and with MK_ENGINE_LAZY_TABLES the first
 *   such file also pays for the engine's big tables.  Results do not change (tests: 512-slot tables per file with several batches
 *   in flight, through the ABI and through the command line). */
#define MK_BATCH_MAX_FILES 1024u
#define MK_BATCH_FILE_MAX ((uint64_t)64 << 20)
#define MK_BATCH_TEXT_MAX ((uint64_t)1 << 30)
typedef struct mk_batch_file {
  const uint8_t *text;
  uint64_t n;
} mk_batch_file;
typedef struct mk_batch_result {
  int32_t status;
  int32_t alone; /* != 0: the file was sketched alone (its table in the batch was too small) */
  mk_result r;
} mk_batch_result;
int mk_sketch_batch_begin(mk_engine *e, int mode, const mk_batch_file *files, uint32_t nfiles);
/* The same for files whose FASTA walk the HOST has done already: files[i].text = the file's rows as mk_fasta_pack_rows made them
 * (format MK_ROWS_PACKED or MK_ROWS_WIDE; 64 bytes a row, 16-byte aligned), files[i].n = their bytes.  No text crosses PCIe (0.48 /
 * 0.31 bytes a base instead of 1.01) and no mk_fab_* kernel runs.  When the files' rows lie in ONE stretch of pinned memory
 * (hipHostMalloc / mk_host_register / mk_host_arena_alloc) in ascending order -- with room between them or not -- the scan kernel
 * reads them WHERE THEY LIE, through the mapping: no copy command at all, and what lies between two files is never touched.  Anywhere
 * else the rows are copied first.  Only for geometries with mk_params_packed_ok().  Results: mk_sketch_batch_end as above (a file the
 * batch's tables cannot take is sketched alone from its rows). */
int mk_sketch_batch_begin_rows(mk_engine *e, int mode, uint32_t format, const mk_batch_file *files, uint32_t nfiles);
int mk_sketch_batch_end(mk_engine *e, mk_batch_result *out /* [nfiles of the oldest batch] */);

/* pinned host memory for the caller's read batches (hipHostMalloc) */
int mk_host_alloc(void **p, size_t bytes);
int mk_host_free(void *p);
/* pin memory the caller already owns (hipHostRegister / hipHostUnregister): row buffers, or the .shuf table before
 * mk_engine_create (its upload then takes 1 ms instead of 20).  Whole mappings of the caller's own (mmap, aligned_alloc of whole
 * pages) that stay mapped while registered -- not pieces of the malloc heap: registering and unregistering those left this runtime
 * with stale pinned ranges, and a later copy out of ordinary memory faulted on the device. */
int mk_host_register(void *p, size_t bytes);
int mk_host_unregister(void *p);
int mk_host_register_on(int device, void *p, size_t bytes); /* the same from a thread that has not used `device` yet */
/* a large pinned block the quick way (anonymous mapping touched by several threads, then one hipHostRegister): a quarter of
 * mk_host_alloc's time for hundreds of MiB; free with the same `bytes` */
int mk_host_arena_alloc(void **p, size_t bytes);
int mk_host_arena_free(void *p, size_t bytes);

/* ---- multi-GPU merge: distinct keys of this engine's shard ---------------------- */
/* number of distinct keys currently held (runs the compaction kernel) */
int mk_partial_count(mk_engine *e, uint64_t *n);
/* copy {key, min(count,65535), first ordinal} of every distinct key into caller DEVICE buffers */
int mk_partial_export(mk_engine *e, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev, uint64_t capacity,
                      uint64_t *n_out);
/* The same in two steps for a caller that drives several engines (one per GPU): mk_partial_count_begin queues the compaction
 * and returns, the next mk_partial_count / mk_partial_export[_async] waits for it; mk_partial_export_async queues the three
 * copies on the engine's stream and returns (mk_engine_sync waits).  Engines on different GPUs then compact and copy at the
 * same time instead of one after the other (libmetakssd_multi.so does exactly that). */
int mk_partial_count_begin(mk_engine *e);
int mk_partial_export_async(mk_engine *e, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev, uint64_t capacity,
                            uint64_t *n_out);
/* fold another shard's export (DEVICE buffers) into this engine: counts add, first ordinals take min */
int mk_partial_import(mk_engine *e, const uint64_t *keys_dev, const uint32_t *counts_dev, const uint64_t *ords_dev,
                      uint64_t n);

/* ---- the same merge by key slices ("all-to-all by key % G so each GPU reduces a key slice, then gather reduced
 * slices"): with G shards the gather above makes ONE engine fold G - 1 whole lists into its table; here every engine folds a G-th
 * of every list, and the one that finishes receives G - 1 lists of DISTINCT keys, which need no folding at all.
 *   every shard:     mk_partial_export_split   its list, cut into G parts by key % G (part sizes to the host)
 *                    mk_partial_restart        the same sketch goes on with empty tables
 *                    [exchange: part g of every shard goes to shard g]
 *                    mk_partial_import         the parts it received and its own part g  ->  its table holds slice g, reduced
 *                    mk_partial_count / mk_partial_list_reserve(e, n_g, ..)  the reduced slice: n_g keys and where they are
 *   finishing shard: mk_partial_list_reserve(e, sum of n_g, ..)  room behind its own slice; [gather: the other slices land there]
 *                    mk_partial_list_commit(e, sum)  /  mk_partial_list_adopt for lists that arrived somewhere else
 *                    mk_sketch_finish          layout + dump straight from the list: the slices are disjoint, so their
 *                                              concatenation IS the sketch's list of distinct keys (the table is not consulted)
 * Results are those of the gather, of one engine scanning everything, and of the reference's sequential run. */
/* mk_partial_export with the list cut into nparts (1..16) parts by key % nparts: part g starts at the sum of the parts in front of
 * it; order inside a part is arbitrary.  Everything is queued on the engine's stream (mk_engine_sync waits): the buffers are
 * complete and part_counts[0..nparts) -- HOST memory, pinned preferred -- are valid after it.  n_out = all parts together. */
int mk_partial_export_split_async(mk_engine *e, uint32_t nparts, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev,
                                  uint64_t capacity, uint64_t *part_counts, uint64_t *n_out);
int mk_partial_export_split(mk_engine *e, uint32_t nparts, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t *ords_dev,
                            uint64_t capacity, uint64_t *part_counts, uint64_t *n_out);
/* the sketch in progress continues with empty tables (flavour and occurrence threshold kept): what was pushed so far lives on only
 * in what the caller exported */
int mk_partial_restart(mk_engine *e);
/* the engine's own key list as DEVICE arrays with room for n entries.  After mk_partial_count said D keys, entries [0, D) are
 * those keys and stay (also when the arrays have to grow); a caller fills [D, n) -- with keys that are DISTINCT from those and
 * from one another -- and commits. */
int mk_partial_list_reserve(mk_engine *e, uint64_t n, uint64_t **keys_dev, uint32_t **counts_dev, uint64_t **ords_dev);
/* the first n entries of the key list are this sketch's distinct keys: mk_sketch_finish lays them out and dumps them */
int mk_partial_list_commit(mk_engine *e, uint64_t n);
/* copies n entries from the caller's DEVICE arrays to [offset, offset + n) of the key list (reserve + copy; commit separately) */
int mk_partial_list_adopt(mk_engine *e, const uint64_t *keys_dev, const uint32_t *counts_dev, const uint64_t *ords_dev, uint64_t n,
                          uint64_t offset);

/* ---- profiling --------------------------------------------------------------------------------- */
int mk_profile_enable(mk_engine *e, int on);
int mk_profile_reset(mk_engine *e);
int mk_profile_get(mk_engine *e, mk_profile *out); /* synchronises the stream */

/* ---- synthetic input: read i = len bases i.i.d. uniform from a counter-based PRNG --
 * word(i,j) = mix64(mix64(seed ^ i) + j);  base b of read i = "ACGT"[(word(i,b/32) >> 2(b%32)) & 3]
 * rows[i*stride .. ] = bases, '\n', zero padding.  Host and device versions produce identical bytes. */
int mk_synth_rows_host(uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len, uint32_t stride, uint8_t *rows);
int mk_synth_rows_device(int device, void *hip_stream, uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len,
                         uint32_t stride, uint8_t *rows_dev);
/* FASTQ text of the same reads: "@r<i>\n<bases>\n+\n<'I'*len>\n" */
int mk_synth_fastq_write(const char *path, uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len);
/* the same file written by `nthreads` threads (record offsets are computable: pwrite into place) */
int mk_synth_fastq_write_mt(const char *path, uint64_t seed, uint64_t first_read, uint64_t nreads, uint32_t len, int nthreads);

/* ---- host-side stage I helpers (run_stageI bookkeeping, command_dist.c:341-500) ----------------- */
/* FASTQ framing of mt_shortreads2koc's reader (iseq2comem.c:672-673) over a memory buffer: copies
 * each record's sequence line (with its '\n') into rows[i*stride]; records whose 4th line is missing
 * are dropped.  Returns the number of rows written through *nrows and the bytes consumed through
 * *consumed (so a caller can stream).  final!=0 means `buf` ends the file. */
int mk_fastq_frame(const uint8_t *buf, size_t n, int final, uint8_t *rows, uint32_t stride, uint64_t max_rows,
                   uint64_t *nrows, size_t *consumed);
/* FASTQ framing of fastq2co's reader (iseq2comem.c:343-363, fgets width 20000) for MK_MODE_OCC_SET: like
 * mk_fastq_frame, plus (a) bases whose quality byte (signed char, raw code) is below qmin (-Q) are written as 'N',
 * which resets the k-mer window exactly like the reference's test at :367; a quality line shorter than its
 * sequence counts as quality 0 there (the reference reads the previous record's bytes); (b) the record rule of that
 * reader: a record missing a line, or whose 4th line lacks its '\n', is not walked, except the first record of
 * the file (records_before == 0), which is walked whenever its sequence line exists; (c) a sequence longer than
 * stride-1 is cut into rows overlapping by TL-1 bases once stride is 4096 (MK_ERR_ARG below that: widen and
 * call again from *consumed).  *nrecords counts records, *nrows rows. */
int mk_fastq_frame_q(const uint8_t *buf, size_t n, int final, int32_t qmin, int32_t TL, uint64_t records_before,
                     uint8_t *rows, uint32_t stride, uint64_t max_rows, uint64_t *nrows, uint64_t *nrecords,
                     size_t *consumed);
/* mk_fastq_frame (occ == 0) or mk_fastq_frame_q (occ != 0) on `nthreads` host threads: the buffer is cut at record
 * boundaries (every fourth line start, counted from the start of `buf`, which must be a record boundary) and the slices
 * are framed concurrently into their places in `rows`; same rows, counts and *consumed as the serial call. */
int mk_fastq_frame_mt(const uint8_t *buf, size_t n, int final, int occ, int32_t qmin, int32_t TL, uint64_t records_before,
                      uint8_t *rows, uint32_t stride, uint64_t max_rows, int nthreads, uint64_t *nrows, uint64_t *nrecords,
                      size_t *consumed);
/* ---- whole-file FASTQ stream (mk_fastq_stream.c): replaces the serial 4 x fgets reader between the parallel loops of
 * mt_shortreads2koc() (iseq2comem.c:672-673) / fastq2co() (:343-363) for a file that is in memory (mmap).  `nthreads` host
 * threads frame chunks of the text into row buffers, the calling thread hands the buffers to `sink` in file order with
 * consecutive row ordinals.  Rows, their order and the error behaviour are those of mk_fastq_frame / mk_fastq_frame_q over
 * the whole text with final != 0, whatever the file contains; row strides are chosen per buffer (multiples of 16). */
typedef struct mk_fastq_opts {
  int32_t occ;          /* 0: mt_shortreads2koc's reader; 1: fastq2co's (quality mask, its record rule, long-read windows) */
  int32_t qmin, TL;     /* occ only: -Q, k-mer length in bases */
  int32_t nthreads;     /* framer threads (1..256) */
  int32_t inflight;     /* pushes kept in flight before the oldest is waited for (1..8); ignored without sink.wait */
  uint64_t chunk_bytes; /* text bytes per framing job = one row buffer = one host-to-device copy, 0 = 32 MiB */
  int32_t drop_pages;   /* != 0 ONLY for a read-only FILE mapping the caller is done with afterwards: a framer gives the
                           pages of its chunk back (madvise MADV_DONTNEED) when it has framed them, so that the page-table
                           work of unmapping a multi-GB file is spread over the threads instead of landing in one munmap.
                           Never set it for anonymous memory (the text would read back as zeros). */
  int32_t ahead;        /* row buffers beyond threads + inflight + 1 (0..192): how far the framers may run ahead of the pushes, e.g. while
                           the engine is still being created */
  int32_t packed;       /* != 0: buffers whose reads all have at most MK_PACKED_MAX_BASES bases are framed as PACKED rows (the sink
                           gets stride = MK_PACKED_PITCH | MK_ROWS_PACKED for those), the others as ASCII rows */
  int32_t fd;           /* > 0 and text == NULL: the text is the first `n` bytes of this file descriptor, read with pread() -- every framer
                           thread preads pieces of 1 MiB into a buffer of its own and frames them there.  Nothing of the file is mapped
                           into the process: no page-table work for 15 GB of text, no madvise(), no TLB shoot-downs beside the rest of
                           the process (the HIP runtime's start-up takes 3-8 times as long beside framers that populate and drop the
                           pages of a mapping, profiles/r05_e2e_front_end.txt).  drop_pages is ignored.  Rows, order, errors: as from a mapping */
  int32_t early_chunks; /* 0: every chunk is framed as soon as a thread and a buffer are free.  n > 0: only the first n chunks are; the others
                           wait until the sink's first push has returned -- for a sink whose first push waits for something that the
                           framers would disturb: thirty-two threads framing at memory speed make the HIP runtime's start-up and the
                           creation of the engine two to eight times as long (profiles/r05_e2e_front_end.txt), which costs more than the framing
                           done meanwhile saves */
  int32_t reserved2;
  uint64_t pool_bytes;  /* 0: threads + inflight + 1 + ahead buffers, each with room for a chunk's text rows.  Otherwise a budget for all
                           row buffers together: with `packed` and a file whose first records are short enough for packed rows the
                           buffers are sized for PACKED rows (a fifth of the text), and there are as many as the budget holds, at most
                           one per chunk -- with one per chunk no framer ever waits for a buffer, so a file is framed at the threads'
                           pace from the first moment on, e.g. while the HIP runtime and the engine are still coming up.  Buffers are
                           handed to the chunks in address order (chunk c frames into buffer c until buffers come back), which lets a
                           sink pin its block piece by piece in front of the pushes.  A chunk whose rows do not fit its buffer ends
                           early and the rest is framed on the calling thread, as ever: sizes only ever cost speed */
} mk_fastq_opts;
typedef struct mk_fastq_stats {
  uint64_t rows, records, chunks, chunks_discarded, serial_rows; /* discarded / serial: work redone on the calling thread */
  uint32_t threads;
  double t_setup_s, t_wait_frame_s, t_push_s, t_total_s; /* calling thread: buffer pool, waiting for framers, in push/wait */
  double t_push_call_s, t_wait_call_s, t_push_call_max_s, t_first_push_call_s; /* t_push_s split: sink.push / sink.wait calls */
} mk_fastq_stats;
typedef struct mk_rows_sink {
  void *ctx;
  int (*push)(void *ctx, const uint8_t *rows, uint32_t stride, uint64_t nrows, uint64_t first_row_ordinal, uint64_t *token);
  int (*wait)(void *ctx, uint64_t token);       /* NULL: push is synchronous, the buffer is free when it returns */
  uint8_t *(*alloc)(void *ctx, size_t bytes);   /* called once: one block for all row buffers (pinned for an engine) */
  void (*release)(void *ctx, uint8_t *p, size_t bytes);
  /* NULL, or: called by a FRAMER thread (any of them, in any order) when the buffer at `rows` has been written and will not be
   * touched again before it is pushed -- a sink that pins its block piece by piece (pages that have been written pin fast, fresh
   * ones slowly) learns here what may be pinned; `bytes` = the room of the buffer, not what was used of it */
  void (*ready)(void *ctx, const uint8_t *rows, size_t bytes);
} mk_rows_sink;
int mk_fastq_stream(const uint8_t *text, size_t n, const mk_fastq_opts *o, const mk_rows_sink *sink, uint64_t first_ordinal,
                    mk_fastq_stats *st);
/* the stream bound to an engine: pinned row buffers, mk_sketch_push_reads_async / mk_sketch_push_wait */
int mk_sketch_push_fastq(mk_engine *e, const uint8_t *text, size_t n, const mk_fastq_opts *o, uint64_t first_ordinal,
                         mk_fastq_stats *st);

/* FASTA front end of fasta2co (iseq2comem.c:240-279): strips line breaks, maps headers/invalid bytes to
 * window resets and cuts the base stream into rows of `stride` bytes overlapping by TL-1 bases so that
 * every k-mer lies in exactly one row.  Call mk_fasta_window_init once per file, then feed the file in
 * any chunking; when *consumed < n (max_rows reached) call again with the rest; pass final!=0 with
 * the last chunk (n may be 0) to flush the pending row.  MK_ERR_FORMAT: the input ends inside a '>' line (no newline
 * behind the last header), where the reference aborts (iseq2comem.c:259-271). */
typedef struct mk_fasta_state {
  uint32_t TL;        /* k-mer length in bases (2k) */
  uint32_t in_header; /* inside a '>' line */
  uint32_t fill;      /* bytes in pending[] */
  uint32_t fresh;     /* of which not yet part of any emitted row */
  uint8_t pending[4096];
} mk_fasta_state;
int mk_fasta_window_init(mk_fasta_state *st, int32_t TL);
int mk_fasta_window(mk_fasta_state *st, const uint8_t *buf, size_t n, int final, uint8_t *rows, uint32_t stride,
                    uint64_t max_rows, uint64_t *nrows, size_t *consumed);

/* sketch directory writer: combco.N / combco.index.N / combco.N.a / cofiles.stat
 * (command_dist.c:408-500; co_dstat_t global_basic.h:116-126, padding zeroed) */
typedef struct mk_sketchdir mk_sketchdir;
int mk_sketchdir_open(const char *outdir, const mk_params *p, int koc, int nfiles, mk_sketchdir **out);
int mk_sketchdir_add(mk_sketchdir *d, const char *input_path, const mk_result *r);
int mk_sketchdir_close(mk_sketchdir *d);

/* ---- `metakssd set -u` / `set -q` ---------------------------------------------------
 * sketch_union() (command_set.c:241-319) and uniq_sketch_union() (:427-512) push every id of a component's combined
 * sketch file through a 2^32-bit dictionary and write the marked ids in ascending order (pan.N); -q keeps the ids
 * that occur exactly once in the whole file (uniq_pan.N).  mk_setop holds the dictionaries on the device:
 *   mk_setop_begin  = memset(dict..)            :284 / :473-474
 *   mk_setop_add    = the marking loop          :293-296 / :482-490   (any number of calls, host or device lists)
 *   mk_setop_finish = the ascending walk        :303-312 / :496-505   -> ids in library-owned pinned host memory,
 *                     valid until the next begin/finish/destroy on this handle */
enum { MK_SET_UNION = 0, MK_SET_UNIQ_UNION = 1 };
typedef struct mk_setop mk_setop;
int mk_setop_create(int device, mk_setop **out); /* MK_ERR_NO_DEVICE without a HIP device: no CPU path */
int mk_setop_destroy(mk_setop *s);
const char *mk_setop_last_error(const mk_setop *s); /* s may be NULL: last error of a failed create */
int mk_setop_begin(mk_setop *s, int mode);
int mk_setop_add(mk_setop *s, const uint32_t *ids, uint64_t n);            /* host list; returns when `ids` may be reused */
int mk_setop_add_device(mk_setop *s, const uint32_t *ids_dev, uint64_t n); /* device list, asynchronous on mk_setop_stream() */
int mk_setop_finish(mk_setop *s, const uint32_t **ids_out, uint64_t *n_out);
int mk_setop_result_device(mk_setop *s, const uint32_t **ids_dev, uint64_t *n); /* the same result, left in HBM */
/* `set -i <pan>` / `set -s <pan>`: sketch_operate() (command_set.c:321-425).  With the pan ids in the dictionary
 * (mk_setop_begin(MK_SET_UNION) + mk_setop_add: the indexing loop :374-377), keep -- in input order -- the ids of a
 * combined sketch file that are (keep_members != 0, -i) / are not (-s) in it: the loop :392-405.  bounds[0..nb) are
 * positions into `ids` (combco.index.N); bounds_out[j] receives the number of kept ids in front of position bounds[j]
 * (the output's combco.index.N).  The dictionary stays valid for further calls.  Result memory as for mk_setop_finish. */
int mk_setop_filter(mk_setop *s, int keep_members, const uint32_t *ids, uint64_t n, const uint64_t *bounds, uint32_t nb,
                    const uint32_t **ids_out, uint64_t *n_out, uint64_t *bounds_out);
void *mk_setop_stream(mk_setop *s); /* hipStream_t the handle works on */

/* `set -g <file.tsv>`: the per-taxon table of grouping_genomes() (command_set.c:866-915).  `ids` = the id lists of the
 * taxon's genomes concatenated in the order of the category file; they go through an FCFS double-hashing table of
 * `table_size` slots (HASH() in 32-bit unsigned arithmetic, id 0 never stored, an id without a place after table_size
 * probes dropped) and come back in slot order.  mk_setop_group_table_size(total ids) is the size the reference picks
 * (:867-872).  Independent of the dictionary state; result memory as for mk_setop_finish. */
int mk_setop_group(mk_setop *s, const uint32_t *ids, uint64_t n, uint32_t table_size, const uint32_t **ids_out, uint64_t *n_out);
uint32_t mk_setop_group_table_size(uint64_t total_ids);

/* ---- `metakssd composite -r <ref> -q <qry>`: the join of get_species_abundance() ----------------------
 * command_composite.c:525-553 builds a k-mer -> position dictionary of one query sketch (one component) and, for every
 * reference sketch, collects the query's counts (combco.N.a) of the k-mers the two share.  mk_setop_join does both for a
 * whole reference component: counts_out = the query's count for every reference position whose id occurs in the query, in
 * reference order; bounds / bounds_out as in mk_setop_filter (combco.index.N of the reference -> one segment of
 * counts_out per reference sketch).  Independent of the dictionary state; result memory as for mk_setop_finish. */
int mk_setop_join(mk_setop *s, const uint32_t *qry_ids, const uint16_t *qry_counts, uint64_t nq, const uint32_t *ref_ids,
                  uint64_t nref, const uint64_t *bounds, uint32_t nb, const uint32_t **counts_out, uint64_t *n_out,
                  uint64_t *bounds_out);
/* measurement (bench.py's `next_rows` leg): device time of the last join -- dictionary build and the two passes over the reference
 * ids -- from HIP events on the handle's stream (the loop it replaces: command_composite.c:525-553) */
int mk_setop_last_join_ms(mk_setop *s, double *ms);

/* ---- stage II and the reference-database search ---------------------------------------------
 * combco2mco() (co2mco.c:12-87) turns a component's combined sketch file (genome-major: combco.N + combco.index.N) into
 * the inverted index (k-mer-id-major): mco.N = for every k-mer id in ascending order the numbers of the genomes that
 * hold it, in genome order; mco.index.N = 2^32 cumulative row ends (32 GiB).  mco_cbdco_nobin_dist()
 * (command_dist.c:902-1079) then counts, for every query sketch and reference genome, the k-mer ids they share
 * (loop :1033-1049) into sharedk_ct.dat, and dist_print_nobin()/output_ctrl() (:1531-1690) print distance.out.
 *   mk_mco_build        = the append loop + prefix sum    co2mco.c:37-59     (stable sort of (id, genome) by id)
 *   mk_mco_index_rows   = the dense index, a slab of rows co2mco.c:59-66     (filled on the device from the row table)
 *   mk_mco_count_*      = the shared-k-mer counting loop  command_dist.c:1033-1049
 *   mk_dist_print       = dist_print_nobin + output_ctrl  command_dist.c:1531-1690  (host: libm + snprintf)
 * Result pointers are library-owned pinned host memory, valid until the next build / destroy on the handle. */
typedef struct mk_mco mk_mco;
int mk_mco_create(int device, mk_mco **out); /* MK_ERR_NO_DEVICE without a HIP device: no CPU path */
int mk_mco_destroy(mk_mco *m);
const char *mk_mco_last_error(const mk_mco *m); /* m may be NULL: last error of a failed create */
/* ids[index[cofnum]], index[cofnum + 1] = one component of a sketch directory.  gids[*n]: mco.N; row_ids / row_ends
 * [*nrows]: the non-empty rows (ascending ids) and their cumulative ends.  The row table also stays on the device. */
int mk_mco_build(mk_mco *m, const uint32_t *ids, const uint64_t *index, uint32_t cofnum, const uint32_t **gids, uint64_t *n,
                 const uint32_t **row_ids, const uint64_t **row_ends, uint64_t *nrows);
/* rows [row0, row0 + nrows) of the dense mco.index.N of the last build (nrows <= 2^27 per call) into out[] (host) */
int mk_mco_index_rows(mk_mco *m, uint64_t row0, uint64_t nrows, uint64_t *out);
/* the stable sort of mk_mco_build on its own (hand-written LSD radix sort, mk_sort.hip.h): n < 2^32 (key, value) pairs in host
 * arrays, sorted by key in place, equal keys in input order -- what combco2mco()'s append loop amounts to (co2mco.c:37-59) */
int mk_mco_sort_pairs(mk_mco *m, uint32_t *keys, uint32_t *vals, uint64_t n);
/* counting: begin zeroes a qry_num x ref_num matrix in HBM; every add handles one component; finish adds the matrix
 * into ct[qry_num * ref_num] (host).  For an add, gids[ngids] is the component's mco.N (NULL: the lists of the last
 * mk_mco_build, still on the device); the row of query id i is gids[ext_start[i] .. ext_end[i]) -- what the reference
 * reads from the mmap'ed index (:1040-1041) -- or, with ext_start == NULL, is looked up on the device in the row table
 * of the last build from qry_ids[i].  Sketches with qry_ctx_ct[k] == 0 are skipped (:1035). */
/* which instantiation of the counting kernel runs (same matrix either way; the tests run all):
 *   MK_MCO_OPT_GLOBAL_COUNTERS 1: global atomics straight into the matrix even where per-workgroup LDS counters would be chosen
 *   MK_MCO_OPT_WIDE_LISTS      1: genome lists stay 32-bit even below 65 535 genomes */
enum { MK_MCO_OPT_GLOBAL_COUNTERS = 1, MK_MCO_OPT_WIDE_LISTS = 2 };
int mk_mco_set_option(mk_mco *m, int option, int64_t value);
int mk_mco_count_begin(mk_mco *m, uint32_t ref_num, uint32_t qry_num);
int mk_mco_count_add(mk_mco *m, const uint32_t *gids, uint64_t ngids, const uint32_t *qry_ids, const uint64_t *ext_start,
                     const uint64_t *ext_end, const uint64_t *qry_index, const uint32_t *qry_ctx_ct);
int mk_mco_count_finish(mk_mco *m, uint32_t *ct);
/* measurement (bench.py's `next_rows` leg): device time of the last build's radix sort and of the last count_add's kernels, from HIP
 * events on the handle's stream -- the loops they replace are co2mco.c:37-59 and command_dist.c:1035-1048 */
int mk_mco_last_kernel_ms(mk_mco *m, double *sort_ms, double *count_ms);

/* distance.out (host).  Options as command_dist_wrapper.c:83-92. */
typedef struct mk_dist_opts {
  int32_t metric;     /* -M: 0 Jaccard / MashD, 1 containment / AafD */
  int32_t outfields;  /* -O: 0 distance, 1 + P-value and FDR, 2 + confidence intervals (default) */
  int32_t correction; /* --correction */
  int32_t num_neigb;  /* -N: the N best references per query, 0 = all */
  double dthreshold;  /* -D: lines with a larger distance are dropped (default 1) */
} mk_dist_opts;
/* names: 256-byte records as in cofiles.stat / mcofiles.stat.  `out` is a FILE*.  MK_ERR_ARG: -N above 1024 or above
 * ref_num (the reference gives up there, :1574), metric / outfields out of range. */
int mk_dist_print(void *out, const mk_dist_opts *o, int32_t kmerlen, int32_t dim_rd_len, uint32_t ref_num, uint32_t qry_num,
                  const uint32_t *ref_ctx_ct, const uint32_t *qry_ctx_ct, const char *refnames, const char *qrynames,
                  const uint32_t *ct);

#ifdef __cplusplus
}
#endif
#endif
