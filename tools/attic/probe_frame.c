// how fast the whole-file FASTQ stream frames when nothing is pushed anywhere: threads x {text rows, packed rows} x {drop pages or not}
#define _GNU_SOURCE
#include "metakssd_hip.h"
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int push(void *c, const uint8_t *r, uint32_t s, uint64_t n, uint64_t o, uint64_t *t) { (void)c; (void)r; (void)s; (void)o; *(uint64_t *)c += n; *t = 0; return 0; }
static uint8_t *al(void *c, size_t b) { (void)c; uint8_t *p = mmap(NULL, b, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(p, b, MADV_HUGEPAGE); for (size_t i = 0; i < b; i += 4096) p[i] = 0; return p; }
static void rel(void *c, uint8_t *p, size_t b) { (void)c; munmap(p, b); }
int main(int argc, char **argv) {
  const char *path = argv[1];
  int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st);
  for (int pass = 0; pass < 2; pass++)
  for (int packed = 0; packed < 2; packed++)
    for (int drop = 1; drop >= 0; drop--)
      for (int T = 24; T <= 96; T *= 2) {
        const uint8_t *m = mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        mk_fastq_opts o; memset(&o, 0, sizeof o);
        o.nthreads = T; o.inflight = 3; o.chunk_bytes = 32u << 20; o.drop_pages = drop; o.packed = packed;
        uint64_t rows = 0;
        mk_rows_sink sink = {&rows, push, NULL, al, rel};
        mk_fastq_stats fs;
        double t0 = now();
        int rc = mk_fastq_stream(m, st.st_size, &o, &sink, 0, &fs);
        double t1 = now();
        munmap((void *)m, st.st_size);
        double t2 = now();
        printf("{\"pass\": %d, \"packed\": %d, \"drop_pages\": %d, \"threads\": %d, \"rc\": %d, \"rows\": %llu, \"stream_s\": %.4f, \"setup_s\": %.4f, \"munmap_s\": %.4f, \"text_GBps\": %.1f}\n", pass, packed, drop, T, rc,
               (unsigned long long)rows, t1 - t0, fs.t_setup_s, t2 - t1, st.st_size / (t1 - t0 - fs.t_setup_s) / 1e9);
        fflush(stdout);
      }
  return 0;
}
