// how fast T threads get a 15 GB file in /dev/shm into their hands: (a) pread into a private buffer that is reused, by chunk size,
// (b) a shared mapping with MADV_POPULATE_READ per chunk and a touch of every page, (c) the same without populate
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int fd; static size_t size, chunk; static int mode; static const uint8_t *map; static volatile uint64_t sink;
static size_t next_chunk; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static void *run(void *arg) {
  (void)arg;
  uint8_t *buf = mode == 0 ? malloc(chunk + 4096) : NULL;
  uint64_t acc = 0;
  for (;;) {
    pthread_mutex_lock(&mu); size_t c = next_chunk++; pthread_mutex_unlock(&mu);
    size_t off = c * chunk; if (off >= size) break;
    size_t len = size - off < chunk ? size - off : chunk;
    if (mode == 0) {
      size_t got = 0; while (got < len) { ssize_t r = pread(fd, buf + got, len - got, off + got); if (r <= 0) break; got += r; }
      for (size_t i = 0; i < len; i += 64) acc += buf[i]; /* touch every line: what a framer's memchr does at least */
    } else {
      if (mode == 1) madvise((void *)(map + off), len, MADV_POPULATE_READ);
      for (size_t i = 0; i < len; i += 64) acc += map[off + i];
      madvise((void *)(map + off), len, MADV_DONTNEED);
    }
  }
  sink += acc; free(buf); return NULL;
}
int main(int argc, char **argv) {
  fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st); size = st.st_size;
  for (int pass = 0; pass < 2; pass++)
  for (mode = 0; mode < 3; mode++)
    for (int T = 12; T <= 96; T *= 2)
      for (chunk = mode == 0 ? (1u << 20) : (32u << 20); chunk <= (32u << 20); chunk *= 8) {
        if (mode) map = mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
        next_chunk = 0;
        pthread_t th[96]; double t0 = now();
        for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, run, NULL);
        for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
        double t1 = now();
        if (mode) munmap((void *)map, size);
        printf("{\"pass\": %d, \"mode\": \"%s\", \"threads\": %d, \"chunk_mib\": %zu, \"seconds\": %.4f, \"GBps\": %.1f}\n", pass,
               mode == 0 ? "pread" : mode == 1 ? "mmap+populate" : "mmap", T, chunk >> 20, t1 - t0, size / (t1 - t0) / 1e9);
        fflush(stdout);
      }
  return 0;
}
