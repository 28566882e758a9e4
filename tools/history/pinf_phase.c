/* Phases of the parallel gzip decoder on one thread (round 4): plain decoding, marker decoding from a block start found by search, markers -> bytes + CRC.
 *   gcc -O2 -o /tmp/pinf_phase tools/history/pinf_phase.c sctagger_amd/csrc/sct_inflate.c -lpthread && /tmp/pinf_phase file.gz   (a gzip file with a plain 10-byte header) */
#include "../../sctagger_amd/csrc/sct_pinflate.c"
#include <stdio.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *in = malloc(n + 64); if (fread(in, 1, n, f) != (size_t)n) return 1;
    const uint8_t *df = in + 10; const int64_t dn = n - 10 - 8;   /* plain 10-byte header assumed */
    /* 1. serial direct decode */
    uint8_t *out = malloc((size_t)1 << 30);
    for (int rep = 0; rep < 2; rep++) {
        sct_inf *s = (sct_inf *)sctio_inflate_new();
        sctinf_start_at(s, df, dn, 0, NULL, 0);
        s->stop_bit = INT64_MAX;
        double t0 = now(); int64_t tot = 0; int32_t st = 0; int64_t used = 0;
        while (st == 0) { int64_t got = sctio_inflate_run(s, out + WINDOW + tot - WINDOW + WINDOW, ((int64_t)1 << 30) - tot - 2*WINDOW, &st, &used); tot += got; }
        double dt = now() - t0;
        printf("serial direct: %ld bytes, status %d, %.3f s, %.2f GB/s\n", (long)tot, st, dt, tot / dt / 1e9);
        sctio_inflate_free(s);
    }
    /* 2. marker decode from a block start near the middle, to the end */
    for (int rep = 0; rep < 2; rep++) {
        sct_inf *s = (sct_inf *)sctio_inflate_new();
        symbuf b = {0};
        double t0 = now();
        int64_t bit = (dn / 4) * 8, found = -1; long tried = 0;
        for (; bit < dn * 8; bit++) {
            const uint32_t x = (uint32_t)(load64(df + (bit >> 3)) >> (bit & 7));
            if ((x & 7u) != 4u || ((x >> 3) & 31u) > 29u || ((x >> 8) & 31u) > 29u) continue;
            if (!precode_complete(df, dn, bit + 13)) continue;
            tried++;
            sctinf_start_at(s, df, dn, bit, NULL, 0);
            if (sctinf_read_block_header(s) != 0) continue;
            b.n = 0; int ended = 0, blocks = 0;
            int rc = decode_markers(s, &b, INT64_MAX, (int64_t)1 << 40, &ended, &blocks);
            if (rc == 0) { found = bit; printf("  blocks %d ended %d\n", blocks, ended); break; }
        }
        double dt = now() - t0;
        long marks = 0; for (int64_t i = 0; i < b.n; i++) marks += (b.v[i] & MARK) != 0;
        printf("marker decode from bit %ld (%ld full header attempts): %ld symbols, %.3f s, %.2f GB/s; markers %.1f %%\n", (long)found, tried, (long)b.n, dt, b.n / dt / 1e9, 100.0 * marks / (b.n ? b.n : 1));
        /* 3. resolve */
        item_t it; memset(&it, 0, sizeof it);
        it.sym = b; it.n = b.n; it.win = malloc(WINDOW); memset(it.win, 'A', WINDOW); it.win_fill_at = WINDOW;
        t0 = now();
        resolve_item(&it);
        dt = now() - t0;
        printf("resolve + crc: %.3f s, %.2f GB/s\n", dt, b.n / dt / 1e9);
        t0 = now(); uint32_t c = sctio_crc32(0, it.buf + SCTIO_PIECE_HEAD, it.n); dt = now() - t0;
        printf("crc alone: %.3f s (%08x)\n", dt, c);
        free(it.buf); free(b.v);
        sctio_inflate_free(s);
    }
    return 0;
}
