/*
 * fuzz_inflate.c — differential fuzzing of csrc/sct_inflate.c against zlib, CPU only, meant to run under ASan + UBSan.
 *
 * Build and run (tools/fuzz_inflate.sh does both and writes the log kept under profiles/):
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/fuzz_inflate tools/fuzz_inflate.c \
 *       sctagger_amd/csrc/sct_inflate.c sctagger_amd/csrc/sct_pinflate.c -lz -lpthread
 *   /tmp/fuzz_inflate <iterations> <seed> [0 = without the parallel driver]
 *
 * Every iteration: take a raw DEFLATE stream made by zlib from structured data (level, strategy, window and memory level
 * drawn at random), mutate it (bit flips, byte writes, truncation, slice duplication / removal / swap, splices of two
 * streams, random block headers at byte boundaries), decode it with zlib (the judge: gzip.open in the reference is zlib) and
 * with the decoder under test, in output blocks of awkward sizes, and compare:
 *   zlib finishes the stream  -> the decoder must finish it too, with the same output and the same number of input bytes used
 *   zlib reports an error or runs out of input -> the decoder must not report success with DIFFERENT output ... and any
 *     "success" where zlib fails is counted and shown (accepting what zlib rejects would let the command read garbage the
 *     reference refuses)
 * Out-of-bounds reads or writes, signed overflow, misaligned access etc. end the run through the sanitizers.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

void *sctio_inflate_new(void);
void sctio_inflate_free(void *p);
void sctio_inflate_set_input(void *p, const uint8_t *in, int64_t len);
int64_t sctio_inflate_run(void *p, uint8_t *out, int64_t cap, int32_t *status, int64_t *in_used);
void sctio_inflate_set_stop(void *p, int64_t stop_bit);
void sctio_inflate_get_state(void *p, int64_t *bitpos, uint8_t *window, uint32_t *win_fill);
/* the parallel driver (csrc/sct_pinflate.c) */
void *sctio_pinflate_new(const uint8_t *in, int64_t len, int32_t threads, int64_t chunk_bytes, int64_t start_bit, const uint8_t *window, uint32_t win_fill);
int64_t sctio_pinflate_next(void *p, uint8_t **piece, uint32_t *crc, int32_t *status, int64_t *in_used);
void sctio_pinflate_stats(void *p, int64_t *speculative, int64_t *redone);
void sctio_pinflate_free(void *p);
void sctio_piece_free(uint8_t *piece);
int64_t sctio_piece_head(void);
uint32_t sctio_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);
/* the fast compressor of the command's output (csrc/sct_fdeflate.c) */
int64_t sctfd_gzip_member(const uint8_t *in, int64_t n, uint8_t *out, void *scratch_mem);
int64_t sctfd_scratch_bytes(void);

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static uint32_t below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0; }

#define MAX_PLAIN (1 << 20)
#define MAX_COMP (MAX_PLAIN + (MAX_PLAIN >> 2) + 4096)
#define OUT_CAP (4 << 20) /* what either decoder may produce from one mutated stream */

static size_t make_plain(uint8_t *p) {
    const uint32_t kind = below(8);
    size_t n = 1 + below(kind == 0 ? 2000 : 200000);
    if (below(40) == 0) n = below(3);
    switch (kind) {
    case 0: /* random bytes */
        for (size_t i = 0; i < n; i++) p[i] = (uint8_t)rnd();
        break;
    case 1: { /* the -lr file: read id, two numbers, a 22-character segment */
        size_t at = 0;
        uint32_t id = below(1000000);
        while (at + 64 < n) at += (size_t)sprintf((char *)p + at, "r%09u\t%u\t%u\t", id++, below(4), 20 + below(40));
        for (size_t i = 0; i < at; i++)
            if (p[i] == 0) p[i] = 'A';
        for (size_t i = at; i < n; i++) p[i] = (uint8_t)"ACGT\n"[below(5)];
        break;
    }
    case 2: /* runs */
        for (size_t i = 0; i < n;) {
            const uint8_t c = (uint8_t)rnd();
            size_t k = 1 + below(600);
            while (k-- && i < n) p[i++] = c;
        }
        break;
    case 3: /* copies at every distance up to the window */
        for (size_t i = 0; i < n;) {
            if (i > 4 && below(10) < 6) {
                size_t st = below((uint32_t)i), k = 3 + below(400);
                while (k-- && i < n) p[i++] = p[st++];
            } else {
                size_t k = 1 + below(40);
                while (k-- && i < n) p[i++] = (uint8_t)rnd();
            }
        }
        break;
    case 4: /* two symbols */
        for (size_t i = 0; i < n; i++) p[i] = (uint8_t)("ab"[below(2)]);
        break;
    case 5: /* every byte value, repeated: long codes */
        for (size_t i = 0; i < n; i++) p[i] = (uint8_t)(i * 7 + (i >> 8));
        break;
    case 6: /* skewed alphabet: very uneven code lengths */
        for (size_t i = 0; i < n; i++) {
            uint32_t r = below(1 << 14), c = 0;
            while ((r & 1) && c < 200) { r >>= 1; c++; }
            p[i] = (uint8_t)(c * 3 + below(3));
        }
        break;
    default: /* DNA text */
        for (size_t i = 0; i < n; i++) p[i] = (uint8_t)"ACGT"[below(4)];
        break;
    }
    return n;
}

static size_t deflate_raw(const uint8_t *p, size_t n, uint8_t *c) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    if (deflateInit2(&z, (int)below(10), Z_DEFLATED, -(9 + (int)below(7)), 1 + (int)below(9), strategies[below(5)]) != Z_OK) abort();
    z.next_in = (Bytef *)p;
    z.avail_in = (uInt)n;
    z.next_out = c;
    z.avail_out = MAX_COMP;
    /* now and then a flush in the middle: empty stored blocks, byte-aligned block starts */
    if (n > 10 && below(4) == 0) {
        z.avail_in = (uInt)(n / 2);
        if (deflate(&z, below(2) ? Z_FULL_FLUSH : Z_SYNC_FLUSH) != Z_OK) abort();
        z.avail_in = (uInt)(n - n / 2);
    }
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    const size_t k = z.total_out;
    deflateEnd(&z);
    return k;
}

static size_t mutate(uint8_t *c, size_t n, const uint8_t *other, size_t other_n) {
    const uint32_t rounds = 1 + below(4);
    for (uint32_t r = 0; r < rounds && n > 0; r++) {
        switch (below(9)) {
        case 0: c[below((uint32_t)n)] ^= (uint8_t)(1u << below(8)); break;
        case 1: c[below((uint32_t)n)] = (uint8_t)rnd(); break;
        case 2: n = below((uint32_t)n + 1); break; /* truncate */
        case 3: { /* duplicate a slice */
            const size_t a = below((uint32_t)n), k = 1 + below(64);
            if (n + k < MAX_COMP && a + k <= n) {
                memmove(c + a + k, c + a, n - a);
                n += k;
            }
            break;
        }
        case 4: { /* remove a slice */
            const size_t a = below((uint32_t)n), k = 1 + below(64);
            if (a + k <= n) {
                memmove(c + a, c + a + k, n - a - k);
                n -= k;
            }
            break;
        }
        case 5: { /* splice the tail of another stream */
            if (other_n > 2) {
                const size_t a = below((uint32_t)n), b = below((uint32_t)other_n), k = other_n - b;
                if (a + k < MAX_COMP) {
                    memcpy(c + a, other + b, k);
                    n = a + k;
                }
            }
            break;
        }
        case 6: { /* a few flips close together (inside one Huffman table description, with luck) */
            const size_t a = below((uint32_t)n);
            for (int i = 0; i < 4; i++) c[a + below((uint32_t)(n - a < 8 ? n - a : 8))] ^= (uint8_t)(1u << below(8));
            break;
        }
        case 7: { /* random bytes from some point on */
            const size_t a = below((uint32_t)n);
            for (size_t i = a; i < n && i < a + 32; i++) c[i] = (uint8_t)rnd();
            break;
        }
        default: break; /* leave it valid */
        }
    }
    return n;
}

/* -> 0 finished, 1 error, 2 input ended too early, 3 output limit */
static char zlib_msg[128];
static int run_zlib(const uint8_t *c, size_t n, uint8_t *out, size_t *out_n, size_t *used) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) abort();
    z.next_in = (Bytef *)c;
    z.avail_in = (uInt)n;
    z.next_out = out;
    z.avail_out = OUT_CAP;
    const int rc = inflate(&z, Z_FINISH);
    *out_n = z.total_out;
    *used = z.total_in;
    snprintf(zlib_msg, sizeof(zlib_msg), "%s", z.msg ? z.msg : "-");
    inflateEnd(&z);
    if (rc == Z_STREAM_END) return 0;
    if (rc == Z_DATA_ERROR || rc == Z_NEED_DICT) return 1;
    if (z.avail_out == 0) return 3;
    return 2; /* Z_BUF_ERROR / Z_OK with input exhausted */
}

static int run_ours(const uint8_t *c, size_t n, uint8_t *blk, size_t cap, uint8_t *out, size_t *out_n, size_t *used) {
    void *st = sctio_inflate_new();
    /* the input exactly sized on the heap: a read past its end is an ASan report */
    uint8_t *in = (uint8_t *)malloc(n ? n : 1);
    memcpy(in, c, n);
    sctio_inflate_set_input(st, in, (int64_t)n);
    int32_t status = 0;
    int64_t in_used = 0;
    size_t total = 0;
    int verdict = 3;
    for (;;) {
        const int64_t k = sctio_inflate_run(st, blk, (int64_t)cap, &status, &in_used);
        if (k < 0 || (size_t)k > cap) {
            fprintf(stderr, "decoder returned %lld bytes for a block of %zu\n", (long long)k, cap);
            abort();
        }
        if (total + (size_t)k > OUT_CAP) { verdict = 3; break; }
        memcpy(out + total, blk + 32768, (size_t)k);
        total += (size_t)k;
        if (status == 1) { verdict = 0; break; }
        if (status == -3) { verdict = 2; break; }
        if (status < 0) { verdict = 1; break; }
        if (total == OUT_CAP) { verdict = 3; break; }
    }
    *out_n = total;
    *used = (size_t)in_used;
    free(in);
    sctio_inflate_free(st);
    return verdict;
}

/* The same stream through the parallel driver: a random number of threads, chunks of a few hundred bytes to a few KB (so that a
 * stream of some size is cut many times), now and then after the serial decoder has done the beginning (the hand-over the command
 * makes: stop at a block boundary, pass position and window on).  Also checks the CRC-32 the pieces come with. */
static long par_pieces_spec = 0, par_pieces_redo = 0, par_handover = 0, par_crc_bad = 0;
static int run_parallel(const uint8_t *c, size_t n, uint8_t *out, size_t *out_n, size_t *used) {
    uint8_t *in = (uint8_t *)malloc(n ? n : 1);   /* exactly sized: a read past its end is an ASan report */
    memcpy(in, c, n);
    size_t total = 0;
    int verdict = 3;
    int64_t start_bit = 0;
    uint8_t window[32768];
    uint32_t win_fill = 0;
    uint32_t crc = 0;
    int32_t status = 0;
    int64_t in_used = 0;
    if (n > 64 && below(3) == 0) {
        /* the serial decoder first, to the first block boundary behind a random position */
        void *st = sctio_inflate_new();
        sctio_inflate_set_input(st, in, (int64_t)n);
        sctio_inflate_set_stop(st, (int64_t)below((uint32_t)(8 * n)));
        uint8_t *blk = (uint8_t *)malloc(32768 + 65536);
        for (;;) {
            const int64_t k = sctio_inflate_run(st, blk, 65536, &status, &in_used);
            if (total + (size_t)k > OUT_CAP) { status = 99; break; }
            memcpy(out + total, blk + 32768, (size_t)k);
            crc = (uint32_t)crc32(crc, blk + 32768, (uInt)k);
            total += (size_t)k;
            if (status != 0) break;
        }
        free(blk);
        if (status == 2) {
            sctio_inflate_get_state(st, &start_bit, window, &win_fill);
            par_handover++;
        }
        sctio_inflate_free(st);
        if (status != 2) {   /* the stream ended, broke or overflowed before the hand-over: the serial decoder's verdict */
            *out_n = total;
            *used = (size_t)in_used;
            free(in);
            return status == 1 ? 0 : status == -3 ? 2 : status == 99 ? 3 : 1;
        }
    }
    static const int64_t chunks[] = {64, 100, 257, 1000, 4096, 20000, 0};
    void *h = sctio_pinflate_new(in, (int64_t)n, 1 + (int32_t)below(4), chunks[below(7)], start_bit, window, win_fill);
    if (!h) abort();
    const int64_t head = sctio_piece_head();
    for (;;) {
        uint8_t *piece = NULL;
        uint32_t pcrc = 0;
        const int64_t k = sctio_pinflate_next(h, &piece, &pcrc, &status, &in_used);
        if (status != 0) {
            verdict = status == 1 ? 0 : status == -3 ? 2 : 1;
            break;
        }
        if (total + (size_t)k > OUT_CAP) {
            sctio_piece_free(piece);
            verdict = 3;
            break;
        }
        memcpy(out + total, piece + head, (size_t)k);
        if ((uint32_t)crc32(0, piece + head, (uInt)k) != pcrc) par_crc_bad++;
        crc = sctio_crc32_combine(crc, pcrc, k);
        total += (size_t)k;
        sctio_piece_free(piece);
    }
    if (verdict == 0 && crc != (uint32_t)crc32(0, out, (uInt)total)) par_crc_bad++;
    int64_t sp = 0, rd = 0;
    sctio_pinflate_stats(h, &sp, &rd);
    par_pieces_spec += sp;
    par_pieces_redo += rd;
    sctio_pinflate_free(h);
    *out_n = total;
    *used = (size_t)in_used;
    free(in);
    return verdict;
}

/* The other direction: the structured data through the fast compressor (one gzip member, output buffer exactly as large as the
 * contract says: a write past it is an ASan report), decoded by zlib: must be the data, CRC and length included. */
static long fd_ok = 0, fd_stored_like = 0;
static int check_fast_compressor(const uint8_t *plain, size_t pn, uint8_t *back) {
    static void *scratch = NULL;
    if (!scratch) scratch = malloc((size_t)sctfd_scratch_bytes());
    uint8_t *in = (uint8_t *)malloc(pn ? pn : 1);
    memcpy(in, plain, pn);
    const size_t cap = pn + pn / 1000 + 64;
    uint8_t *z = (uint8_t *)malloc(cap);
    const int64_t zn = sctfd_gzip_member(in, (int64_t)pn, z, scratch);
    int ok = zn > 0 && (size_t)zn <= cap;
    if (ok) {
        z_stream s;
        memset(&s, 0, sizeof(s));
        if (inflateInit2(&s, 31) != Z_OK) abort();
        s.next_in = z;
        s.avail_in = (uInt)zn;
        s.next_out = back;
        s.avail_out = OUT_CAP;
        const int rc = inflate(&s, Z_FINISH);
        ok = rc == Z_STREAM_END && s.total_out == pn && s.avail_in == 0 && memcmp(back, plain, pn) == 0;
        inflateEnd(&s);
        if ((size_t)zn > pn) fd_stored_like++;
    }
    free(z);
    free(in);
    if (ok) fd_ok++;
    return ok;
}

int main(int argc, char **argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 100000;
    rng_state ^= (uint64_t)(argc > 2 ? atoll(argv[2]) : 1) * 0x9E3779B97F4A7C15ull;
    uint8_t *plain = (uint8_t *)malloc(MAX_PLAIN), *comp = (uint8_t *)malloc(MAX_COMP), *other = (uint8_t *)malloc(MAX_COMP);
    uint8_t *oz = (uint8_t *)malloc(OUT_CAP), *oo = (uint8_t *)malloc(OUT_CAP + 16), *op = (uint8_t *)malloc(OUT_CAP + 16);
    const int with_parallel = argc > 3 ? atoi(argv[3]) : 1;
    long par_ok = 0, par_bad = 0, par_vs_serial = 0, par_limit = 0;
    size_t other_n = 0;
    long agree_ok = 0, agree_bad = 0, limit = 0, ours_ok_zlib_bad = 0, ours_bad_zlib_ok = 0, class_differs = 0, valid = 0;
    static const size_t caps[] = {65536, 65537, 70001, 131072, 1 << 20};
    for (long it = 0; it < iters; it++) {
        const size_t pn = make_plain(plain);
        if (!check_fast_compressor(plain, pn, op)) {
            fprintf(stderr, "FAST COMPRESSOR: zlib does not get the data back at iteration %ld (%zu bytes)\n", it, pn);
            return 2;
        }
        size_t cn = deflate_raw(plain, pn, comp);
        const int keep_valid = below(8) == 0;
        if (!keep_valid) cn = mutate(comp, cn, other, other_n);
        else valid++;
        size_t zn = 0, zu = 0, on = 0, ou = 0;
        const int vz = run_zlib(comp, cn, oz, &zn, &zu);
        const size_t cap = caps[below(5)];
        uint8_t *blk = (uint8_t *)malloc(32768 + cap); /* exactly sized: a write past the block is an ASan report */
        const int vo = run_ours(comp, cn, blk, cap, oo, &on, &ou);
        free(blk);
        if (vz == 3 || vo == 3) {
            limit++;
        } else if (vz == 0 && vo == 0) {
            if (zn != on || memcmp(oz, oo, zn) != 0 || zu != ou) {
                fprintf(stderr, "MISMATCH at iteration %ld: zlib %zu bytes out / %zu in, decoder %zu / %zu\n", it, zn, zu, on, ou);
                return 2;
            }
            if (keep_valid && (zn != pn || memcmp(oz, plain, pn) != 0)) {
                fprintf(stderr, "round trip broken at iteration %ld\n", it);
                return 2;
            }
            agree_ok++;
        } else if (vz != 0 && vo != 0) {
            agree_bad++;
            if (vz != vo) class_differs++; /* "corrupt" vs "ends too early": which one is seen first depends on look-ahead */
        } else if (vo == 0) {
            if (ours_ok_zlib_bad < 20) fprintf(stderr, "decoder accepts what zlib rejects (zlib verdict %d: %s) at iteration %ld, %zu bytes in\n", vz, zlib_msg, it, cn);
            ours_ok_zlib_bad++;
        } else {
            if (ours_bad_zlib_ok < 5) fprintf(stderr, "decoder rejects (%d) what zlib accepts at iteration %ld, %zu bytes in\n", vo, it, cn);
            ours_bad_zlib_ok++;
        }
        if (with_parallel) {
            /* the parallel driver: the same verdict as the serial decoder (finished / refused), and on success the same bytes and
             * the same count of input bytes — which the block above has just compared with zlib's */
            size_t pn2 = 0, pu = 0;
            const int vp = run_parallel(comp, cn, op, &pn2, &pu);
            if (vp == 3 || vo == 3) par_limit++;
            else if ((vp == 0) != (vo == 0)) {
                if (par_vs_serial < 20) fprintf(stderr, "parallel verdict %d, serial verdict %d (zlib %d) at iteration %ld, %zu bytes in\n", vp, vo, vz, it, cn);
                par_vs_serial++;
            } else if (vp == 0) {
                if (pn2 != on || memcmp(op, oo, on) != 0 || pu != ou) {
                    fprintf(stderr, "PARALLEL MISMATCH at iteration %ld: serial %zu bytes out / %zu in, parallel %zu / %zu\n", it, on, ou, pn2, pu);
                    return 2;
                }
                par_ok++;
            } else par_bad++;
        }
        /* keep this stream around as splice material */
        if (cn && below(4) == 0) {
            memcpy(other, comp, cn);
            other_n = cn;
        }
        if ((it + 1) % 100000 == 0) {
            printf("%ld streams: both finish %ld (identical output), both refuse %ld (%ld with a different reason), output limit %ld, "
                   "decoder ok / zlib not %ld, zlib ok / decoder not %ld\n", it + 1, agree_ok, agree_bad, class_differs, limit, ours_ok_zlib_bad, ours_bad_zlib_ok);
            fflush(stdout);
        }
    }
    printf("DONE %ld streams (%ld left valid): both finish %ld, both refuse %ld, output limit %ld, decoder ok / zlib not %ld, zlib ok / decoder not %ld\n",
           iters, valid, agree_ok, agree_bad, limit, ours_ok_zlib_bad, ours_bad_zlib_ok);
    if (with_parallel)
        printf("PARALLEL %ld streams: finishes like the serial decoder with identical output %ld, refuses like it %ld, verdict differs %ld, output limit %ld; "
               "pieces accepted from speculative decoding %ld, decoded again serially %ld, hand-overs from the serial decoder %ld, piece CRCs wrong %ld\n",
               iters, par_ok, par_bad, par_vs_serial, par_limit, par_pieces_spec, par_pieces_redo, par_handover, par_crc_bad);
    printf("COMPRESSOR %ld inputs through csrc/sct_fdeflate.c and back through zlib: identical %ld (%ld of them not smaller than the input: stored blocks)\n", iters, fd_ok, fd_stored_like);
    fflush(stdout);
    free(plain); free(comp); free(other); free(oz); free(oo); free(op);
    return (ours_ok_zlib_bad || ours_bad_zlib_ok || par_vs_serial || par_crc_bad) ? 3 : 0;
}
