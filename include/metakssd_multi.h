/*
 * metakssd_multi.h -- C ABI of libmetakssd_multi.so: one sketch over several GPUs of one node (BASELINE config 4).
 *
 * The reference's parallel axis is one process with OpenMP threads over one shared table (iseq2comem.c:675-720) or over
 * files (command_dist.c:363-372).  Here: one engine per GPU (libmetakssd_hip.so), the caller deals fixed-stride row buffers
 * with GLOBAL read ordinals to the engines in any pattern (the command line deals the FASTQ stream's buffers round-robin so
 * that every GPU's PCIe link is busy), every GPU scans into its own table with no traffic between GPUs, and
 * mk_multi_finish is the one exchange: the distinct-key lists {key, count, first ordinal} of engines 1.. go to engine 0's
 * GPU with grouped ncclSend / ncclRecv (RCCL over xGMI: every sender has its own link to GPU 0), engine 0 folds them in with
 * ONE import launch and lays out and dumps the result.  Counts add and first ordinals take the minimum, both commutative, so
 * the result is bit-identical to one engine scanning everything -- and to the reference's sequential run.
 *
 * A separate library because it links librccl.so (570 MB): the single-GPU command line never loads it.
 * When the device list names one GPU several times (tests on a one-GPU box) the lists move with plain device copies
 * (hipMemcpyPeerAsync) instead; on distinct GPUs a failing RCCL is an error unless the caller opts in to copies
 * (mk_multi_create_ex).  mk_multi_transport() says which transport is in use.
 */
#ifndef METAKSSD_MULTI_H
#define METAKSSD_MULTI_H
#include "metakssd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mk_multi mk_multi;

/* engines on devices[0..n) (created concurrently, one host thread each), engine 0 is where results are finished.
 * n <= 16: the table slot's count field has room for that many concurrent imports of 65535, not for 17 (mk_multi.hip).
 * Transport: RCCL when the devices are distinct.  If RCCL cannot be initialised there, mk_multi_create FAILS (MK_ERR_HIP, the
 * RCCL error in mk_multi_last_error(NULL)) unless the caller opted in to device copies -- MK_MULTI_ALLOW_DEVICE_COPIES here,
 * MK_MULTI_ALLOW_COPIES=1 in the environment for mk_multi_create, --allow-device-copies on the command line.  A list that
 * names one GPU several times can only use device copies and does.  The transport chosen is printed on stderr for n > 1. */
enum { MK_MULTI_ALLOW_DEVICE_COPIES = 1u, /* RCCL failed to initialise: fall back to hipMemcpyPeerAsync instead of failing */
       MK_MULTI_FORCE_DEVICE_COPIES = 2u  /* do not try RCCL at all */ };
int mk_multi_create(const mk_params *p, const int *devices, int n, mk_multi **out);
int mk_multi_create_ex(const mk_params *p, const int *devices, int n, unsigned flags, mk_multi **out);
int mk_multi_destroy(mk_multi *m);
const char *mk_multi_last_error(const mk_multi *m); /* m may be NULL: last error of a failed create */
int mk_multi_count(const mk_multi *m);
mk_engine *mk_multi_engine(mk_multi *m, int i); /* push rows with mk_sketch_push_reads[_async|_device] on these */
const char *mk_multi_transport(const mk_multi *m); /* "rccl" or "device copies" */

int mk_multi_begin(mk_multi *m, int mode);           /* mk_sketch_begin on every engine */
int mk_multi_begin_occ(mk_multi *m, int min_occurrence);
/* gather + import + finish; the result is engine 0's (valid until its next begin).  gather_ms / tail_ms (may be NULL):
 * wall time of the exchange alone and of everything in this call */
int mk_multi_finish(mk_multi *m, mk_result *out, double *gather_ms, double *tail_ms);

/* How mk_multi_finish merges (same result either way):
 *   GATHER  the lists of engines 1.. go to engine 0, which folds them into its table with one import launch (above)
 *   SLICES  SURVEY.md 8e's alternative: every engine cuts its list into n parts by key % n, part g goes to engine g (all-to-all),
 *           every engine folds its slice at the same time, the reduced slices -- disjoint key sets -- go to engine 0 and are its key
 *           list as they stand: engine 0 folds an n-th of what it folds under GATHER and runs layout + dump
 *   AUTO    (default) GATHER below four engines, SLICES from four on */
enum { MK_MULTI_MERGE_AUTO = 0, MK_MULTI_MERGE_GATHER = 1, MK_MULTI_MERGE_SLICES = 2 };
int mk_multi_set_merge(mk_multi *m, int how);
const char *mk_multi_last_merge(const mk_multi *m); /* "gather" or "slices": what the last mk_multi_finish did */
/* wall-clock phases of the last mk_multi_finish, milliseconds.  GATHER: export (compaction + copies on engines 1..), exchange,
 * import (queued), 0, finish.  SLICES: export (compaction + split on every engine), exchange (all-to-all), import (fold +
 * compaction of the slices, waited for), gather (reduced slices to engine 0), finish (layout + dump on engine 0) */
typedef struct { double export_ms, exchange_ms, import_ms, gather_ms, finish_ms, total_ms; } mk_multi_times;
int mk_multi_last_times(const mk_multi *m, mk_multi_times *t);

#ifdef __cplusplus
}
#endif
#endif
