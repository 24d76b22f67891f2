// shm_write_probe.c -- how fast can N GB of output reach ONE file on tmpfs?  (the end-to-end route of `rb` is bound by this step)
//   gcc -O2 -pthread -o tools/shm_write_probe tools/shm_write_probe.c && tools/shm_write_probe /dev/shm/probe.bin 16 32
// Modes: pwrite from T threads (16 MB segments; what rb does), mmap + memcpy, mmap + MADV_POPULATE_WRITE per thread + memcpy,
// the same with MADV_HUGEPAGE first, fallocate + pwrite.
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
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
typedef struct {
    int mode, fd, t, T;
    size_t bytes;
    const char *src;
    char *map;
} job;
#define SEG ((size_t)16 << 20)
static void *worker(void *a) {
    job *j = (job *)a;
    const size_t nseg = (j->bytes + SEG - 1) / SEG;
    if (j->mode == 0 || j->mode == 4) { // pwrite, segments round-robin
        for (size_t s = (size_t)j->t; s < nseg; s += (size_t)j->T) {
            const size_t off = s * SEG, n = off + SEG <= j->bytes ? SEG : j->bytes - off;
            size_t done = 0;
            while (done < n) {
                ssize_t w = pwrite(j->fd, j->src + (off + done) % ((size_t)1 << 30), n - done, (off_t)(off + done));
                if (w <= 0) { perror("pwrite"); exit(1); }
                done += (size_t)w;
            }
        }
    } else { // mapped: each thread owns a contiguous range
        const size_t per = ((j->bytes / (size_t)j->T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        size_t lo = per * (size_t)j->t, hi = lo + per;
        if (lo > j->bytes) lo = j->bytes;
        if (hi > j->bytes) hi = j->bytes;
        if (hi > lo && (j->mode == 2 || j->mode == 3))
            if (madvise(j->map + lo, hi - lo, MADV_POPULATE_WRITE)) perror("MADV_POPULATE_WRITE");
        for (size_t off = lo; off < hi; off += SEG) {
            const size_t n = off + SEG <= hi ? SEG : hi - off;
            memcpy(j->map + off, j->src + off % ((size_t)1 << 30), n);
        }
    }
    return NULL;
}
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/dev/shm/rb_probe.bin";
    const size_t gb = argc > 2 ? (size_t)atol(argv[2]) : 8;
    const int T = argc > 3 ? atoi(argv[3]) : 32;
    const size_t bytes = gb << 30;
    char *src = (char *)malloc(((size_t)1 << 30) + SEG);
    for (size_t i = 0; i < ((size_t)1 << 30) + SEG; i += 4096) src[i] = (char)i;
    memset(src, 'x', ((size_t)1 << 30) + SEG);
    {
        FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/shmem_enabled", "r");
        char b[256] = "";
        if (f) { if (fgets(b, sizeof b, f)) printf("shmem_enabled: %s", b); fclose(f); }
    }
    static const char *names[] = {"pwrite, 16 MB segments", "mmap + memcpy", "mmap + POPULATE_WRITE + memcpy", "mmap + HUGEPAGE + POPULATE_WRITE + memcpy", "fallocate + pwrite"};
    const int only = argc > 4 ? atoi(argv[4]) : -1; // one mode only
    for (int mode = 0; mode < 5; mode++) {
        if (only >= 0 && mode != only) continue;
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        const double t0 = now();
        char *map = NULL;
        if (mode >= 1 && mode <= 3) {
            if (ftruncate(fd, (off_t)bytes)) { perror("ftruncate"); return 1; }
            map = (char *)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { perror("mmap"); return 1; }
            if (mode == 3 && madvise(map, bytes, MADV_HUGEPAGE)) perror("MADV_HUGEPAGE");
        }
        if (mode == 4 && fallocate(fd, 0, 0, (off_t)bytes)) perror("fallocate");
        const double t1 = now();
        pthread_t th[256];
        job jobs[256];
        for (int t = 0; t < T; t++) {
            jobs[t] = (job){mode, fd, t, T, bytes, src, map};
            pthread_create(&th[t], NULL, worker, &jobs[t]);
        }
        for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
        const double t2 = now();
        if (map) munmap(map, bytes);
        close(fd);
        const double t3 = now();
        printf("%-45s %2d threads  setup %.2f s  write %.2f s  close %.2f s  -> %.2f GB/s\n", names[mode], T, t1 - t0, t2 - t1, t3 - t2, (double)bytes / 1e9 / (t3 - t0));
        fflush(stdout);
    }
    unlink(path);
    return 0;
}
