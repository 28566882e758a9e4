/* Compressor micro-benchmark (round 4): csrc/sct_fdeflate.c over a file in 1 MB gzip members, best of 29 passes, every member checked through zlib once.
 *   gcc -O2 -o /tmp/fd_bench tools/history/fd_bench.c sctagger_amd/csrc/sct_fdeflate.c sctagger_amd/csrc/sct_pinflate.c sctagger_amd/csrc/sct_inflate.c -lz -lpthread && /tmp/fd_bench out.tsv
 * (out.tsv: the text the command writes; e.g. the first 400 000 rows of cfg3 rendered from the oracle) */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
int64_t sctfd_gzip_member(const uint8_t *in, int64_t n, uint8_t *out, void *scratch);
int64_t sctfd_scratch_bytes(void);
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *in = malloc(n); if (fread(in, 1, n, f) != (size_t)n) return 1;
    const long piece = 1 << 20;
    uint8_t *out = malloc(piece + piece / 8 + 4096), *back = malloc(piece);
    void *scratch = malloc(sctfd_scratch_bytes());
    double best = 1e9; long btot = 0;
    for (int rep = 0; rep < 30; rep++) {
        double t0 = now(); long tot = 0;
        for (long a = 0; a < n; a += piece) {
            long m = n - a < piece ? n - a : piece;
            long z = sctfd_gzip_member(in + a, m, out, scratch);
            if (z <= 0) { printf("fail\n"); return 1; }
            tot += z;
            if (rep == 0) {
                z_stream s; memset(&s, 0, sizeof s); inflateInit2(&s, 31); s.next_in = out; s.avail_in = z; s.next_out = back; s.avail_out = piece;
                int r = inflate(&s, Z_FINISH); if (r != Z_STREAM_END || (long)s.total_out != m || memcmp(back, in + a, m)) { printf("ROUND TRIP BROKEN at %ld\n", a); return 2; }
                inflateEnd(&s);
            }
        }
        double dt = now() - t0;
        if (rep > 0 && dt < best) best = dt;
        btot = tot;
    }
    printf("best of 29: %.1f MB/s  ratio %.3f\n", n / best / 1e6, (double)n / btot);
    return 0;
}
