/* mk_host_internal.h -- small helpers shared by the host C files and the HIP translation units */
#ifndef MK_HOST_INTERNAL_H
#define MK_HOST_INTERNAL_H
#include <stdint.h>

#if defined(__HIPCC__)
#define MK_HD __host__ __device__
#else
#define MK_HD
#endif

/* splitmix64 finaliser used as a counter-based PRNG (synthetic reads, .shuf generator) */
static inline MK_HD uint64_t mk_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* word j (32 bases) of synthetic read i: SURVEY.md 8d / metakssd_hip.h */
static inline MK_HD uint64_t mk_synth_word(uint64_t seed, uint64_t read, uint64_t j) {
  return mk_mix64(mk_mix64(seed ^ read) + j);
}

/* MK_POISON=<byte> (environment, read once; "1" means 0xA5): every device and pinned allocation of the library and of the command
 * line, and every buffer they take back for another batch / sketch / file, is filled with that byte before use -- the substitute for
 * a GPU address sanitizer on hosts that have none: a kernel that reads what nobody wrote in THIS use of the buffer then reads the
 * pattern instead of a previous process's zeros or a previous batch's bytes.  Returns -1 when the hook is off.  0x43 ('C') makes
 * stale TEXT a run of valid bases, 0xA5 makes stale table slots and indices garbage. */
#include <stdlib.h>
static inline int mk_poison_byte(void) {
  static int v = -2;
  if (v == -2) {
    const char *t = getenv("MK_POISON");
    if (!t || !*t) v = -1;
    else { long x = strtol(t, NULL, 0); v = x <= 0 ? -1 : (x == 1 ? 0xA5 : (int)(x & 0xff)); }
  }
  return v;
}
#if !defined(__HIPCC__)
#include <stddef.h>
/* range forms of the two FASTQ framers (mk_frontend.c), shared with the whole-file stream (mk_fastq_stream.c) */
int mk_fastq_frame_range(const uint8_t *buf, size_t n, size_t stop, int final, uint8_t *rows, uint32_t stride, uint64_t max_rows,
                         uint64_t *nrows, size_t *consumed, uint32_t *need);
int mk_fastq_frame_q_range(const uint8_t *buf, size_t n, size_t stop, int final, int32_t qmin, int32_t TL, uint64_t records_before,
                           uint8_t *rows, uint32_t stride, uint64_t max_rows, uint64_t *nrows, uint64_t *nrecords,
                           size_t *consumed, uint32_t *need);
#endif
#endif
