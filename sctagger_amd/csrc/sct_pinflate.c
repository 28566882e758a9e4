/*
 * sct_pinflate.c — ONE DEFLATE stream decoded by several threads (part of libsctagger_io.so).
 *
 * Why: the reference reads its -lr file through gzip.open (/root/reference/scTagger.py:602-628), one serial stream.  The streaming
 * command's own serial decoder (sct_inflate.c) inflates 10^7 rows in 0.43 s — and 10^8 rows (BASELINE configs[3] through the
 * command) in 4.3 s, for 50 ms of GPU work.  A DEFLATE stream has no index, but it can be decoded from the middle in two passes
 * (the idea of pugz / rapidgzip; nothing of theirs is used here):
 *
 *   cut       the compressed bytes into chunks of equal size
 *   find      in every chunk but the first, the first bit position at which a dynamic-Huffman block starts: try every bit
 *             offset — BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a COMPLETE code-length code, complete literal / distance
 *             codes (sctinf_read_block_header) — and decode from there; a position that does not decode cleanly through its
 *             first blocks was not one
 *   pass 1    decode the chunk from that position to the first block boundary at or behind the next chunk's first byte, into
 *             16-bit SYMBOLS: a literal byte, or 0x8000 | k = "byte k of the 32 KiB that precede this chunk's output", which
 *             nobody knows yet.  A match that reaches back behind the chunk's start copies such markers; markers are copied like
 *             any other symbol.  (In the -lr file they never die out: "\t1\t30\t" is copied from row to row for ever.)
 *   chain     in stream order: a chunk whose start is where its predecessor stopped is accepted; its window = the last 32 KiB
 *             before it, known once the predecessor's last 32 KiB are resolved (32 K table look-ups, serial but cheap)
 *   pass 2    resolve: markers -> bytes through the window, symbols -> bytes, CRC-32 of the piece; any thread, any order
 *
 * A chunk whose start was not found, was found wrongly (it does not begin where the predecessor ended) or that fails to decode is
 * simply decoded again from the position the chain has reached, with the known window, by the serial decoder (REDO): speculation
 * never decides what the output is.  Errors of the stream itself surface there, at the position the serial decoder would report
 * them.  tools/fuzz_inflate.c runs this path against zlib AND against the serial decoder under ASan + UBSan.
 *
 * The consumer (sctio_pinflate_next) gets the pieces in stream order, each in a malloc'ed buffer of its own with SCTIO_PIECE_HEAD
 * bytes of room in front (the reader puts the previous piece's unfinished line there), and owns them (sctio_piece_free).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sct_inflate_impl.h"

#define SCTIO_PIECE_HEAD (1 << 20)
#define MARK 0x8000u

void *sctio_inflate_new(void);
void sctio_inflate_free(void *p);
int64_t sctio_inflate_run(void *p, uint8_t *out, int64_t cap, int32_t *status, int64_t *in_used);

/* ---------------------------------------------------------------------------------------------------------------------------
 * CRC-32 (the gzip polynomial), slice by 8, and the combination of two CRCs (what zlib calls crc32_combine)
 * ------------------------------------------------------------------------------------------------------------------------- */
static uint32_t crc_tab[8][256];
static int crc_can_fold = 0;   /* the CPU has carry-less multiplication (crc_fold below); set once, in crc_init */
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void) {
#if defined(__x86_64__)
    crc_can_fold = (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) ? 1 : 0;
#endif
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xFF];
}
static uint32_t crc_tables(uint32_t c, const uint8_t *p, int64_t n) {   /* c = the running (inverted) state */
    while (n > 0 && ((uintptr_t)p & 7)) {
        c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFF];
        n--;
    }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = crc_tab[7][v & 0xFF] ^ crc_tab[6][(v >> 8) & 0xFF] ^ crc_tab[5][(v >> 16) & 0xFF] ^ crc_tab[4][(v >> 24) & 0xFF] ^
            crc_tab[3][(v >> 32) & 0xFF] ^ crc_tab[2][(v >> 40) & 0xFF] ^ crc_tab[1][(v >> 48) & 0xFF] ^ crc_tab[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFF];
    return c;
}

#if defined(__x86_64__)
#include <immintrin.h>
/* Folding with carry-less multiplications (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Intel 2009):
 * four 128-bit accumulators are each multiplied by x^512 mod P (k1, k2: the two 64-bit halves) and added to the next 64 bytes; the
 * four are then folded into one (x^128: k3, k4), that into 64 bits (k5) and reduced to 32 by Barrett's method (P and its inverse).
 * Bit-reflected constants for the gzip polynomial.  n >= 64 and a multiple of 16; the command sums 3.8 + 3.0 GB per 10^8 rows (the
 * decoded input, the members of its output): 1.9 GB/s with the tables, 4 of its 31 CPU-seconds. */
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc_fold(uint32_t c, const uint8_t *p, int64_t n) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64;
    n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        n -= 64;
    }
    __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x2);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x3);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x4);
    while (n >= 16) {
        a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), _mm_loadu_si128((const __m128i *)p));
        p += 16;
        n -= 16;
    }
    const __m128i lo32 = _mm_setr_epi32(~0, 0, ~0, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);          /* 128 -> 96 bits */
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);                         /* -> 64 */
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, lo32), k5, 0x00), x2);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, lo32), poly, 0x10);   /* Barrett */
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, lo32), poly, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t sctio_crc32(uint32_t crc, const uint8_t *p, int64_t n) {
    pthread_once(&crc_once, crc_init);
    uint32_t c = ~crc;
#if defined(__x86_64__)
    if (n >= 256) {
        if (crc_can_fold) {
            const int64_t m = n & ~(int64_t)15;
            c = crc_fold(c, p, m);
            p += m;
            n -= m;
        }
    }
#endif
    return ~crc_tables(c, p, n);
}
/* the table version alone (tests compare the two) */
uint32_t sctio_crc32_tables(uint32_t crc, const uint8_t *p, int64_t n) {
    pthread_once(&crc_once, crc_init);
    return ~crc_tables(~crc, p, n);
}
/* GF(2) 32 x 32 matrices as 32 words: mat[i] = image of bit i */
static uint32_t gf2_times(const uint32_t *mat, uint32_t vec) {
    uint32_t sum = 0;
    for (int i = 0; vec; vec >>= 1, i++)
        if (vec & 1u) sum ^= mat[i];
    return sum;
}
static void gf2_square(uint32_t *sq, const uint32_t *mat) {
    for (int i = 0; i < 32; i++) sq[i] = gf2_times(mat, mat[i]);
}
/* CRC of A || B from crc(A), crc(B), len(B): advance crc(A) over len(B) zero bytes (repeated squaring of the "one zero bit"
 * operator), then xor */
uint32_t sctio_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2) {
    if (len2 <= 0) return crc1;
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u;
    uint32_t row = 1;
    for (int i = 1; i < 32; i++) {
        odd[i] = row;
        row <<= 1;
    }
    gf2_square(even, odd);   /* two zero bits */
    gf2_square(odd, even);   /* four */
    do {                     /* first squaring below: one zero byte */
        gf2_square(even, odd);
        if (len2 & 1) crc1 = gf2_times(even, crc1);
        len2 >>= 1;
        if (!len2) break;
        gf2_square(odd, even);
        if (len2 & 1) crc1 = gf2_times(odd, crc1);
        len2 >>= 1;
    } while (len2);
    return crc1 ^ crc2;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * pass 1: the decoder of sct_inflate.c once more, writing 16-bit symbols
 * ------------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint16_t *v;
    int64_t n, cap;
} symbuf;

static int sym_room(symbuf *b, int64_t extra) {
    if (b->n + extra <= b->cap) return 0;
    int64_t want = b->cap ? b->cap * 2 : (1 << 20);
    while (want < b->n + extra) want *= 2;
    uint16_t *p = (uint16_t *)realloc(b->v, (size_t)want * sizeof(uint16_t));
    if (!p) return -1;
    b->v = p;
    b->cap = want;
    return 0;
}

/* copy `len` symbols from `dist` back; positions before the chunk's start are markers into the unknown window */
static inline void sym_copy(uint16_t *v, int64_t p, uint32_t dist, uint32_t len) {
    int64_t src = p - (int64_t)dist;
    uint32_t j = 0;
    for (; src < 0 && j < len; src++, j++) v[p + j] = (uint16_t)(MARK | (uint32_t)(WINDOW + src));
    for (; j < len; j++, src++) v[p + j] = v[src];
}

/*
 * Decode blocks into b until (a) a block boundary at or behind bit `stop_bit`, (b) the end of the stream (*ended = 1), or (c) more
 * than `max_out` symbols (stops at the next block boundary).  *blocks counts the blocks that decoded completely.
 * 0 = stopped as asked; -1 corrupt; -3 the input ends; -4 out of memory.
 */
static int decode_markers(sct_inf *s, symbuf *b, int64_t stop_bit, int64_t max_out, int *ended, int *blocks) {
    *ended = 0;
    for (;;) {
        if (s->phase == 3) {
            *ended = 1;
            return 0;
        }
        if (s->phase == 0) {
            if (sctinf_bitpos(s) >= stop_bit || b->n > max_out) return 0;
            const int rc = sctinf_read_block_header(s);
            if (rc) return rc;
            continue;
        }
        if (s->phase == 2) {
            const uint32_t k = s->stored_left;
            if ((int64_t)k > s->in_end - s->in_next) return -3;
            if (sym_room(b, (int64_t)k + 8)) return -4;
            for (uint32_t i = 0; i < k; i++) b->v[b->n + i] = s->in_next[i];
            b->n += k;
            s->in_next += k;
            s->stored_left = 0;
            s->phase = s->final_block ? 3 : 0;
            ++*blocks;
            continue;
        }
        /* ---- a Huffman block ---- */
        int done = 0;
        while (!done) {
            if (sym_room(b, 1 << 16)) return -4;
            uint16_t *const v = b->v;
            int64_t p = b->n;
            const int64_t p_end = b->cap - 320;
            uint64_t bb = s->bitbuf;
            int bc = s->bitcount;
            const uint8_t *in = s->in_next;
            const uint8_t *const in_fast_end = s->in_end - 16;
            int rc = 0;
            /* fast loop: >= 16 bytes of input and >= 320 symbols of room ahead */
            while (in <= in_fast_end && p < p_end) {
                bb |= load64(in) << bc;
                in += (63 - bc) >> 3;
                bc |= 56;
                uint32_t e = s->ll[bb & ((1u << LL_BITS) - 1)];
                if (e & K_LIT) {
                    bb >>= (e & 0xFF);
                    bc -= (int)(e & 0xFF);
                    v[p++] = (uint16_t)((e >> 16) & 0xFF);
                    e = s->ll[bb & ((1u << LL_BITS) - 1)];
                    if (e & K_LIT) {
                        bb >>= (e & 0xFF);
                        bc -= (int)(e & 0xFF);
                        v[p++] = (uint16_t)((e >> 16) & 0xFF);
                        e = s->ll[bb & ((1u << LL_BITS) - 1)];
                        if (e & K_LIT) {
                            bb >>= (e & 0xFF);
                            bc -= (int)(e & 0xFF);
                            v[p++] = (uint16_t)((e >> 16) & 0xFF);
                            continue;
                        }
                    }
                    if (bc < 48) continue;
                }
                if (e & K_SUB) {
                    bb >>= LL_BITS;
                    bc -= LL_BITS;
                    e = s->ll[(e >> 16) + (bb & ((1u << ((e >> 8) & 0xF)) - 1))];
                    if (e & K_LIT) {
                        bb >>= (e & 0xFF);
                        bc -= (int)(e & 0xFF);
                        v[p++] = (uint16_t)((e >> 16) & 0xFF);
                        continue;
                    }
                }
                if (e & (K_EOB | K_BAD)) {
                    if (e & K_BAD) {
                        rc = -1;
                        break;
                    }
                    bb >>= (e & 0xFF);
                    bc -= (int)(e & 0xFF);
                    done = 1;
                    break;
                }
                bb >>= (e & 0xFF);
                bc -= (int)(e & 0xFF);
                const int xb = (int)((e >> 8) & 0xF);
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xb) - 1));
                bb >>= xb;
                bc -= xb;
                uint32_t d = s->dd[bb & ((1u << D_BITS) - 1)];
                if (d & K_SUB) {
                    bb >>= D_BITS;
                    bc -= D_BITS;
                    d = s->dd[(d >> 16) + (bb & ((1u << ((d >> 8) & 0xF)) - 1))];
                }
                if (d & (K_BAD | K_LIT | K_EOB)) {
                    rc = -1;
                    break;
                }
                bb >>= (d & 0xFF);
                bc -= (int)(d & 0xFF);
                const int db = (int)((d >> 8) & 0xF);
                const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << db) - 1));
                bb >>= db;
                bc -= db;
                if ((int64_t)dist > p + WINDOW) {
                    rc = -1;
                    break;
                }
                if ((int64_t)dist <= p && dist >= 4) {   /* the usual case: whole source inside the chunk, four symbols per step */
                    const uint16_t *src = v + p - dist;
                    uint16_t *dst = v + p, *const end = v + p + len;
                    do {   /* may write up to 3 symbols past the match: 320 of room */
                        memcpy(dst, src, 8);
                        dst += 4;
                        src += 4;
                    } while (dst < end);
                } else {
                    sym_copy(v, p, dist, len);
                }
                p += len;
            }
            {   /* give the whole bytes that were only looked at back to the input */
                const int back = bc >> 3;
                in -= back;
                bc -= back * 8;
                bb &= bc ? (((uint64_t)1 << bc) - 1) : 0;
            }
            s->bitbuf = bb;
            s->bitcount = bc;
            s->in_next = in;
            b->n = p;
            if (rc) return rc;
            if (done) break;
            if (in <= in_fast_end) continue;   /* out of room only: grow and go on */
            /* careful loop near the end of the input: one symbol at a time, every access checked */
            while (!done) {
                if (sym_room(b, 600)) return -4;
                need_bits(s, 15);
                uint32_t e = s->ll[s->bitbuf & ((1u << LL_BITS) - 1)];
                if (e & K_SUB) {
                    if (s->bitcount < LL_BITS) return -3;
                    take_bits(s, LL_BITS);
                    need_bits(s, 15);
                    e = s->ll[(e >> 16) + (s->bitbuf & ((1u << ((e >> 8) & 0xF)) - 1))];
                }
                if (e & K_BAD) return s->in_next >= s->in_end && s->bitcount < 15 ? -3 : -1;
                if ((int)(e & 0xFF) > s->bitcount) return -3;
                take_bits(s, (int)(e & 0xFF));
                if (e & K_LIT) {
                    b->v[b->n++] = (uint16_t)((e >> 16) & 0xFF);
                    continue;
                }
                if (e & K_EOB) {
                    done = 1;
                    break;
                }
                const int xb = (int)((e >> 8) & 0xF);
                if (!need_bits(s, xb)) return -3;
                const uint32_t len = (e >> 16) + take_bits(s, xb);
                need_bits(s, 15);
                uint32_t d = s->dd[s->bitbuf & ((1u << D_BITS) - 1)];
                if (d & K_SUB) {
                    if (s->bitcount < D_BITS) return -3;
                    take_bits(s, D_BITS);
                    need_bits(s, 15);
                    d = s->dd[(d >> 16) + (s->bitbuf & ((1u << ((d >> 8) & 0xF)) - 1))];
                }
                if (d & (K_BAD | K_LIT | K_EOB)) return (d & K_BAD) && s->in_next >= s->in_end && s->bitcount < 15 ? -3 : -1;
                if ((int)(d & 0xFF) > s->bitcount) return -3;
                take_bits(s, (int)(d & 0xFF));
                const int db = (int)((d >> 8) & 0xF);
                if (!need_bits(s, db)) return -3;
                const uint32_t dist = (d >> 16) + take_bits(s, db);
                if ((int64_t)dist > b->n + WINDOW) return -1;
                sym_copy(b->v, b->n, dist, len);
                b->n += len;
            }
        }
        s->phase = s->final_block ? 3 : 0;
        ++*blocks;
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * the driver
 * ------------------------------------------------------------------------------------------------------------------------- */
enum { C_IDLE = 0, C_RUNNING = 1, C_DONE = 2, C_TAKEN = 3 };

typedef struct {
    int state;
    int found;                 /* pass 1 produced something that may be used */
    int ended;                 /* the stream's final block lies in this chunk */
    int64_t start_bit, end_bit;
    symbuf sym;                /* pass 1's symbols (every chunk but the first) */
    uint8_t *bytes;            /* ... or bytes already (the first chunk, redone stretches): malloc base, data at + SCTIO_PIECE_HEAD */
    int64_t n_bytes;
    uint32_t bytes_crc;        /* CRC-32 of those bytes, summed by pass 1 (not under the lock: up to 128 MB) */
} chunk_t;

typedef struct {
    int ready;                 /* resolved: buf / n / crc are final */
    int resolving;
    uint8_t *buf;              /* malloc base; the piece's bytes start at buf + SCTIO_PIECE_HEAD */
    int64_t n;
    uint32_t crc;
    symbuf sym;                /* still to be resolved (pass 2) through win[] */
    uint8_t *win;              /* the WINDOW bytes that precede this piece ... */
    uint32_t win_fill_at;      /* ... of which only the last win_fill_at exist (the stream's first 32 KiB) */
} item_t;

typedef struct {
    const uint8_t *in;
    int64_t in_len, chunk_bytes;
    int64_t start_bit, origin_bit;   /* where the first chunk begins (a block boundary the caller has decoded up to), rounded down to a byte */
    int n_chunks, n_threads, lookahead;
    chunk_t *chunks;
    item_t **items;            /* in stream order; entries are allocated one by one (workers hold pointers to them) */
    int n_items, cap_items, cons;
    pthread_t *threads;
    int threads_started;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int stop;
    int next_p1;               /* next chunk whose pass 1 nobody has taken */
    /* the chain: everything before cur_bit is decided */
    int chain_k;               /* next chunk the chain looks at */
    int64_t cur_bit;
    uint8_t win[WINDOW];
    uint32_t win_fill;
    int chain_done, chain_error;
    int64_t end_bit;           /* where the stream ended (chain_done without error) */
    int redo_wanted, redo_running;
    int64_t redo_target;
    int64_t speculative, redone;   /* pieces by origin (statistics) */
    /* symbol buffers go round: a chunk's pass 1 takes one, the piece's pass 2 gives it back (fresh memory costs a page fault per
     * 4 KiB, and the faults of many threads queue up behind one lock of the kernel: pass 1 ran five times slower on four threads) */
    symbuf pool[64];
    int n_pool;
    int64_t sym_hint;              /* symbols a chunk usually produces (the last chunk's count) */
    uint8_t win0[WINDOW];          /* the output that precedes start_bit (its last win0_fill bytes) */
    uint32_t win0_fill;
} pinf_t;

static symbuf pool_take(pinf_t *h) {   /* mutex held */
    symbuf b = {0, 0, 0};
    if (h->n_pool > 0) b = h->pool[--h->n_pool];
    b.n = 0;
    return b;
}
static void pool_give(pinf_t *h, symbuf b) {   /* mutex held */
    if (!b.v) return;
    if (h->n_pool < 64) h->pool[h->n_pool++] = b;
    else free(b.v);
}

static int64_t chunk_first_bit(const pinf_t *h, int k) { return k == 0 ? h->start_bit : h->origin_bit + (int64_t)k * h->chunk_bytes * 8; }

/* new window after `n` more output bytes `tail` (the last min(n, WINDOW) of them are given) */
static void window_push(uint8_t *win, uint32_t *fill, const uint8_t *data, int64_t n) {
    if (n >= WINDOW) {
        memcpy(win, data + n - WINDOW, WINDOW);
        *fill = WINDOW;
        return;
    }
    const uint32_t keep = (uint32_t)((int64_t)*fill + n > WINDOW ? WINDOW - n : *fill);
    memmove(win + WINDOW - keep - n, win + WINDOW - keep, keep);
    memcpy(win + WINDOW - n, data, (size_t)n);
    *fill = keep + (uint32_t)n;
}

static item_t *new_item(pinf_t *h) {
    if (h->n_items == h->cap_items) {
        const int cap = h->cap_items * 2;
        item_t **p = (item_t **)realloc(h->items, (size_t)cap * sizeof(item_t *));
        if (!p) return NULL;
        h->items = p;
        h->cap_items = cap;
    }
    item_t *it = (item_t *)calloc(1, sizeof(item_t));
    if (!it) return NULL;
    h->items[h->n_items++] = it;
    return it;
}

/* the chain (mutex held): accept what fits, discard what does not, ask for a REDO where there is a gap */
static void chain_advance(pinf_t *h) {
    while (!h->chain_done && !h->redo_wanted && !h->redo_running) {
        if (h->chain_k >= h->n_chunks) {   /* behind the last chunk and the stream has not ended: the rest, serially */
            h->redo_wanted = 1;
            h->redo_target = INT64_MAX;
            break;
        }
        chunk_t *c = &h->chunks[h->chain_k];
        if (c->state != C_DONE) break;
        if (!c->found || c->start_bit < h->cur_bit) {   /* nothing usable / begins inside what is already decided */
            pool_give(h, c->sym);
            free(c->bytes);
            memset(&c->sym, 0, sizeof(c->sym));
            c->bytes = NULL;
            c->state = C_TAKEN;
            h->chain_k++;
            continue;
        }
        if (c->start_bit > h->cur_bit) {   /* a gap: the predecessor stopped early, or its successor's start was missed */
            h->redo_wanted = 1;
            h->redo_target = c->start_bit;
            break;
        }
        item_t *it = new_item(h);
        if (!it) {
            h->chain_error = -4;
            h->chain_done = 1;
            break;
        }
        if (c->bytes) {   /* bytes already */
            it->buf = c->bytes;
            it->n = c->n_bytes;
            it->crc = c->bytes_crc;
            it->ready = 1;
            window_push(h->win, &h->win_fill, it->buf + SCTIO_PIECE_HEAD + (it->n > WINDOW ? it->n - WINDOW : 0), it->n > WINDOW ? WINDOW : it->n);
        } else {
            it->sym = c->sym;
            it->n = c->sym.n;
            it->win = (uint8_t *)malloc(WINDOW);
            if (!it->win) {
                it->ready = 1;   /* (nothing to hand out: the consumer reports the error when it gets here) */
                h->chain_error = -4;
                h->chain_done = 1;
                break;
            }
            memcpy(it->win, h->win, WINDOW);
            it->win_fill_at = h->win_fill;
            /* the window behind this piece: its last 32 KiB resolved now (a marker may point at any byte of the old window) */
            const int64_t t = it->n > WINDOW ? WINDOW : it->n;
            uint8_t tail[WINDOW];
            const uint16_t *sv = it->sym.v + it->n - t;
            for (int64_t i = 0; i < t; i++) {
                const uint16_t x = sv[i];
                tail[i] = (x & MARK) ? it->win[x & (MARK - 1)] : (uint8_t)x;   /* (a marker before the stream's start: pass 2 reports it) */
            }
            window_push(h->win, &h->win_fill, tail, t);
        }
        c->bytes = NULL;
        memset(&c->sym, 0, sizeof(c->sym));
        c->state = C_TAKEN;
        h->cur_bit = c->end_bit;
        h->speculative++;
        if (c->ended) {
            h->chain_done = 1;
            h->end_bit = c->end_bit;
        }
        h->chain_k++;
    }
}

/* pass 2 of one item (no lock held): symbols -> bytes through the window, CRC */
static int resolve_item(item_t *it) {
    const uint32_t win_fill_at = it->win_fill_at;
    uint8_t *buf = (uint8_t *)malloc((size_t)(SCTIO_PIECE_HEAD + it->n + 64));
    if (!buf) return -4;
    uint8_t *o = buf + SCTIO_PIECE_HEAD;
    const uint16_t *v = it->sym.v;
    const uint8_t *w = it->win;
    /* Two thirds of the symbols of a -lr file are markers (every row copies its first columns from the row before, and so on back
     * to the unknown window), in runs the branch predictor cannot follow: with a branch per symbol this pass cost as much as decoding
     * the bytes directly does.  Without one: the window byte is always loaded (a literal's low bits index it harmlessly) and selected. */
    const int64_t n = it->n;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const uint32_t x0 = v[i], x1 = v[i + 1], x2 = v[i + 2], x3 = v[i + 3];
        const uint8_t w0 = w[x0 & (MARK - 1)], w1 = w[x1 & (MARK - 1)], w2 = w[x2 & (MARK - 1)], w3 = w[x3 & (MARK - 1)];
        const uint32_t m0 = 0u - (x0 >> 15), m1 = 0u - (x1 >> 15), m2 = 0u - (x2 >> 15), m3 = 0u - (x3 >> 15);
        o[i] = (uint8_t)((w0 & m0) | (x0 & ~m0));
        o[i + 1] = (uint8_t)((w1 & m1) | (x1 & ~m1));
        o[i + 2] = (uint8_t)((w2 & m2) | (x2 & ~m2));
        o[i + 3] = (uint8_t)((w3 & m3) | (x3 & ~m3));
    }
    for (; i < n; i++) {
        const uint32_t x = v[i];
        o[i] = (x & MARK) ? w[x & (MARK - 1)] : (uint8_t)x;
    }
    /* the marker that reaches back furthest: as signed 16-bit numbers the markers are the negative ones, the smallest the furthest */
    int lowest16 = 0;
    i = 0;
#if defined(__x86_64__)
    if (n >= 8) {
        __m128i mn = _mm_setzero_si128();
        for (; i + 8 <= n; i += 8) mn = _mm_min_epi16(mn, _mm_loadu_si128((const __m128i *)(v + i)));
        int16_t t[8];
        _mm_storeu_si128((__m128i *)t, mn);
        for (int k = 0; k < 8; k++)
            if (t[k] < lowest16) lowest16 = t[k];
    }
#endif
    for (; i < n; i++)
        if ((int16_t)v[i] < lowest16) lowest16 = (int16_t)v[i];
    const uint32_t lowest = lowest16 < 0 ? (uint32_t)(lowest16 & (MARK - 1)) : WINDOW;
    it->buf = buf;
    it->crc = sctio_crc32(0, o, it->n);
    free(it->win);
    it->win = NULL;
    return lowest < WINDOW - win_fill_at ? -1 : 0;   /* a match that reaches back before the first byte of the stream */
}

/* HCLEN and the code-length code's lengths start at bit `at`: is that code complete (zlib accepts no other)? */
static int precode_complete(const uint8_t *in, int64_t in_len, int64_t at) {
    if ((at >> 3) + 16 > in_len) return 1;   /* near the end of the input: let the header parser decide */
    const uint64_t a = load64(in + (at >> 3)) >> (at & 7);   /* >= 57 bits */
    const int n = (int)(a & 15u) + 4;
    uint32_t kraft = 0;
    uint64_t v = a >> 4;     /* at least 53 bits left: 16 lengths from here, the others from a second load */
    for (int i = 0; i < n; i++) {
        if (i == 16) v = load64(in + ((at + 4 + 48) >> 3)) >> ((at + 4 + 48) & 7);
        const uint32_t l = (uint32_t)(v & 7u);
        v >>= 3;
        if (l) kraft += 128u >> l;
    }
    return kraft == 128u;
}

/* Symbols (bytes) a piece may hold before it ends at the next block boundary: sixteen times the chunk's compressed bytes — a
 * -lr file compresses 4 : 1, so an ordinary chunk is decoded whole — and never more than 128 M.  (Until round 5: 256 times + 64 M —
 * a stream of zeros, or a crafted one, made pieces of gigabytes, several of them in flight, where the serial decoder works in
 * constant memory.  What a piece does not take, the chain hands to its serial decoder or to the next piece.) */
static int64_t piece_limit(const pinf_t *h) {
    int64_t lim = h->chunk_bytes * 16;
    if (lim < ((int64_t)8 << 20)) lim = (int64_t)8 << 20;
    if (lim > ((int64_t)128 << 20)) lim = (int64_t)128 << 20;
    return lim;
}

#define PENDING_MAX ((int64_t)3 << 30)   /* bytes of decoded pieces nobody has fetched yet, beyond which no further chunk is started */

/* pass 1 of chunk k (no lock held) */
static void pass1(pinf_t *h, int k, sct_inf *s, symbuf b, int64_t hint) {
    chunk_t *c = &h->chunks[k];
    const int64_t first = chunk_first_bit(h, k), stop = k + 1 < h->n_chunks ? chunk_first_bit(h, k + 1) : INT64_MAX;
    const int64_t max_out = piece_limit(h);
    c->found = 0;
    c->ended = 0;
    if (k == 0) {
        /* the stream's first chunk: plain bytes, nothing is unknown */
        sctinf_start_at(s, h->in, h->in_len, h->start_bit, h->win0, h->win0_fill);
        s->stop_bit = stop;
        int64_t cap = 8 << 20, n = 0;
        uint8_t *buf = (uint8_t *)malloc((size_t)(SCTIO_PIECE_HEAD + cap + WINDOW));
        int32_t st = 0;
        int64_t used = 0;
        while (buf) {
            /* sctio_inflate_run wants WINDOW bytes of scratch in front of where it writes: the bytes decoded so far serve */
            const int64_t room = cap - n;
            if (room < (1 << 20)) {
                cap *= 2;
                uint8_t *nb = (uint8_t *)realloc(buf, (size_t)(SCTIO_PIECE_HEAD + cap + WINDOW));
                if (!nb) {
                    free(buf);
                    buf = NULL;
                    break;
                }
                buf = nb;
                continue;
            }
            /* (the decoder copies its own history in front of the output: it overwrites bytes we already hold with the same bytes) */
            if (n > max_out) s->stop_bit = 0;   /* pieces are handed over whole: end this one at the next block boundary */
            const int64_t got = sctio_inflate_run(s, buf + SCTIO_PIECE_HEAD + n - WINDOW, room, &st, &used);
            n += got;
            if (st != 0) break;
        }
        if (buf && (st == 1 || st == 2)) {
            c->found = 1;
            c->start_bit = h->start_bit;
            c->end_bit = sctinf_bitpos(s);
            if (st == 1) {
                c->ended = 1;
                c->end_bit = used * 8;
            }
            c->bytes = buf;
            c->n_bytes = n;
            c->bytes_crc = sctio_crc32(0, buf + SCTIO_PIECE_HEAD, n);
        } else {
            free(buf);   /* an error in the first chunk: the REDO from bit 0 reports it */
        }
        free(b.v);
        return;
    }
    /* find a block start at or behind the chunk's first bit, before the next chunk's */
    if (hint > 0 && sym_room(&b, hint + hint / 8 + (1 << 16))) {   /* one allocation instead of doublings */
        free(b.v);
        return;
    }
    const int64_t last = (stop == INT64_MAX ? h->in_len * 8 : stop) - 1;
    for (int64_t bit = first; bit <= last && !__atomic_load_n(&h->stop, __ATOMIC_RELAXED); bit++) {
        /* 13 bits: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29 */
        const int64_t byte = bit >> 3;
        if (byte + 8 > h->in_len) break;
        const uint32_t x = (uint32_t)(load64(h->in + byte) >> (bit & 7));
        if ((x & 7u) != 4u || ((x >> 3) & 31u) > 29u || ((x >> 8) & 31u) > 29u) continue;
        if (!precode_complete(h->in, h->in_len, bit + 13)) continue;
        sctinf_start_at(s, h->in, h->in_len, bit, NULL, 0);
        if (sctinf_read_block_header(s) != 0) continue;
        b.n = 0;
        int ended = 0, blocks = 0;
        const int rc = decode_markers(s, &b, stop, max_out, &ended, &blocks);
        if (rc == -4) break;
        if (rc != 0) {
            /* not a block start after all (or the stream is broken here: then the chain finds out).  A candidate that got through
             * several blocks before it failed is not worth a second search: the chain will redo this stretch. */
            if (blocks >= 2) break;
            continue;
        }
        c->found = 1;
        c->start_bit = bit;
        c->end_bit = sctinf_bitpos(s);
        c->ended = ended;
        if (ended) c->end_bit = (c->end_bit + 7) & ~(int64_t)7;
        c->sym = b;
        return;
    }
    free(b.v);
}

/* a stretch decoded serially from h->cur_bit with the known window, to the first block boundary at or behind `target` */
typedef struct {
    uint8_t *buf;
    int64_t n, end_bit;
    int rc, ended;
} redo_t;

static void redo(pinf_t *h, int64_t from_bit, int64_t target, const uint8_t *win, uint32_t win_fill, sct_inf *s, redo_t *r) {
    memset(r, 0, sizeof(*r));
    sctinf_start_at(s, h->in, h->in_len, from_bit, win, win_fill);
    s->stop_bit = target;
    const int64_t limit = piece_limit(h);   /* pieces are handed over whole: not unboundedly large */
    int64_t cap = 8 << 20, n = 0;
    uint8_t *buf = (uint8_t *)malloc((size_t)(SCTIO_PIECE_HEAD + cap + WINDOW));
    int32_t st = 0;
    int64_t used = 0;
    if (!buf) {
        r->rc = -4;
        return;
    }
    /* the decoder keeps its own copy of the history (s->window) and lays it in front of every output block: the first WINDOW
     * bytes before buf + HEAD + n are scratch for it — they lie inside the head room / the bytes already written (same bytes) */
    for (;;) {
        if (cap - n < (1 << 20)) {
            cap *= 2;
            uint8_t *nb = (uint8_t *)realloc(buf, (size_t)(SCTIO_PIECE_HEAD + cap + WINDOW));
            if (!nb) {
                free(buf);
                r->rc = -4;
                return;
            }
            buf = nb;
        }
        if (n > limit && s->stop_bit != 0) s->stop_bit = 0;   /* the next block boundary will do */
        const int64_t got = sctio_inflate_run(s, buf + SCTIO_PIECE_HEAD + n - WINDOW, cap - n, &st, &used);
        n += got;
        if (st != 0) break;
    }
    if (st < 0) {
        free(buf);
        r->rc = st;
        return;
    }
    r->buf = buf;
    r->n = n;
    r->ended = st == 1;
    r->end_bit = st == 1 ? used * 8 : sctinf_bitpos(s);
}

static void *worker(void *arg) {
    pinf_t *h = (pinf_t *)arg;
    sct_inf *s = (sct_inf *)sctio_inflate_new();
    pthread_mutex_lock(&h->mu);
    if (!s && !h->chain_done) {
        /* no memory for this thread's decoder state: if every worker ended up here nobody would ever finish the chain and the
         * consumer would wait for ever — the stream fails instead, like any other allocation on the way */
        h->chain_error = -4;
        h->chain_done = 1;
        pthread_cond_broadcast(&h->cv);
    }
    for (;;) {
        if (h->stop || !s) break;
        chain_advance(h);
        /* 1. a stretch the chain needs decoded serially */
        if (h->redo_wanted && !h->redo_running) {
            h->redo_wanted = 0;
            h->redo_running = 1;
            const int64_t from = h->cur_bit, target = h->redo_target;
            uint8_t win[WINDOW];
            memcpy(win, h->win, WINDOW);
            const uint32_t wf = h->win_fill;
            pthread_mutex_unlock(&h->mu);
            redo_t r;
            redo(h, from, target, win, wf, s, &r);
            uint32_t crc = r.rc == 0 ? sctio_crc32(0, r.buf + SCTIO_PIECE_HEAD, r.n) : 0;
            pthread_mutex_lock(&h->mu);
            h->redo_running = 0;
            if (r.rc != 0) {
                h->chain_error = r.rc;
                h->chain_done = 1;
            } else {
                item_t *it = new_item(h);
                if (!it) {
                    free(r.buf);
                    h->chain_error = -4;
                    h->chain_done = 1;
                } else {
                    it->buf = r.buf;
                    it->n = r.n;
                    it->crc = crc;
                    it->ready = 1;
                    window_push(h->win, &h->win_fill, r.buf + SCTIO_PIECE_HEAD + (r.n > WINDOW ? r.n - WINDOW : 0), r.n > WINDOW ? WINDOW : r.n);
                    h->cur_bit = r.end_bit;
                    h->redone++;
                    if (r.ended) {
                        h->chain_done = 1;
                        h->end_bit = r.end_bit;
                    }
                }
            }
            pthread_cond_broadcast(&h->cv);
            continue;
        }
        /* 2. an accepted piece that waits for its pass 2 (lowest first: the consumer wants them in order) */
        item_t *it = NULL;
        for (int i = h->cons; i < h->n_items; i++)
            if (!h->items[i]->ready && !h->items[i]->resolving) {
                it = h->items[i];
                break;
            }
        if (it) {
            it->resolving = 1;
            pthread_mutex_unlock(&h->mu);
            const int rc = resolve_item(it);
            pthread_mutex_lock(&h->mu);
            pool_give(h, it->sym);
            memset(&it->sym, 0, sizeof(it->sym));
            it->resolving = 0;
            it->ready = 1;
            if (rc != 0) {
                free(it->buf);   /* nothing of this piece is handed out */
                it->buf = NULL;
            }
            if (rc != 0 && !h->chain_error) {
                /* a speculative piece that cannot be right (out of memory, or a reference before the start of the stream): the
                 * stream is broken or memory is; say so rather than hand out bytes */
                h->chain_error = rc;
                h->chain_done = 1;
            }
            pthread_cond_broadcast(&h->cv);
            continue;
        }
        /* 3. pass 1 of the next chunk, not too far ahead of the chain */
        /* (... nor of the consumer: the pieces nobody has fetched yet are whole buffers; without this bound a reader that waits for
         * its own consumers let the decoders run through the file, 3.8 GB of pieces for 10^8 rows) */
        /* (... nor by more BYTES than a pipeline needs: pieces are bounded — piece_limit — but a stream that inflates 16 : 1 makes every
         * one of them that large; what is decoded and not yet fetched, plus what the running passes will add, stays under PENDING_MAX) */
        int64_t pending = 0;
        for (int i = h->cons; i < h->n_items; i++) pending += h->items[i]->n * (h->items[i]->ready ? 1 : 3);   /* symbols: 2 bytes + the bytes to come */
        for (int k = h->chain_k; k < h->next_p1; k++)
            if (h->chunks[k].state == C_RUNNING || h->chunks[k].state == C_DONE) pending += 3 * (h->chunks[k].state == C_DONE ? h->chunks[k].sym.n + h->chunks[k].n_bytes : h->sym_hint);
        if (!h->chain_done && h->next_p1 < h->n_chunks && h->next_p1 < h->chain_k + h->lookahead && h->n_items - h->cons < h->lookahead + 8 &&
            (pending < PENDING_MAX || h->next_p1 == h->chain_k)) {
            const int k = h->next_p1++;
            h->chunks[k].state = C_RUNNING;
            symbuf b = {0, 0, 0};
            if (k > 0) b = pool_take(h);
            const int64_t hint = h->sym_hint;
            pthread_mutex_unlock(&h->mu);
            pass1(h, k, s, b, hint);
            pthread_mutex_lock(&h->mu);
            h->chunks[k].state = C_DONE;
            if (h->chunks[k].found && h->chunks[k].sym.n > 0) h->sym_hint = h->chunks[k].sym.n;
            pthread_cond_broadcast(&h->cv);
            continue;
        }
        if (h->chain_done) {
            int pending = 0;
            for (int i = h->cons; i < h->n_items; i++)
                if (!h->items[i]->ready) pending = 1;
            if (!pending) break;
        }
        pthread_cond_wait(&h->cv, &h->mu);
    }
    pthread_mutex_unlock(&h->mu);
    sctio_inflate_free(s);
    return NULL;
}

/*
 * Start decoding the raw DEFLATE stream in[0 .. len) (it may hold more than the stream: a gzip trailer, further members) with
 * `threads` threads and chunks of `chunk_bytes` compressed bytes (0 = chosen from len and threads), from bit `start_bit` on — a
 * block boundary up to which the caller has decoded the stream itself (0 = the beginning) — with `window` = the WINDOW-byte
 * history buffer of that decoding, of which the last `win_fill` bytes are output.  NULL = out of memory.
 */
void *sctio_pinflate_new(const uint8_t *in, int64_t len, int32_t threads, int64_t chunk_bytes, int64_t start_bit, const uint8_t *window,
                         uint32_t win_fill) {
    pinf_t *h = (pinf_t *)calloc(1, sizeof(pinf_t));
    if (!h) return NULL;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (start_bit < 0 || start_bit > len * 8) start_bit = 0;
    if (win_fill > WINDOW || !window) win_fill = 0;
    h->start_bit = start_bit;
    h->origin_bit = start_bit & ~(int64_t)7;
    if (win_fill) memcpy(h->win0 + WINDOW - win_fill, window + WINDOW - win_fill, win_fill);
    h->win0_fill = win_fill;
    memcpy(h->win, h->win0, WINDOW);
    h->win_fill = win_fill;
    h->cur_bit = start_bit;
    const int64_t rest = len - (h->origin_bit >> 3);
    if (chunk_bytes <= 0) {
        chunk_bytes = rest / ((int64_t)threads * 4) + 1;
        if (chunk_bytes < (256 << 10)) chunk_bytes = 256 << 10;
        if (chunk_bytes > (8 << 20)) chunk_bytes = 8 << 20;
    }
    if (chunk_bytes < 64) chunk_bytes = 64;
    h->in = in;
    h->in_len = len;
    h->chunk_bytes = chunk_bytes;
    h->n_chunks = (int)((rest + chunk_bytes - 1) / chunk_bytes);
    if (h->n_chunks < 1) h->n_chunks = 1;
    h->n_threads = threads;
    h->lookahead = 2 * threads + 2;
    h->chunks = (chunk_t *)calloc((size_t)h->n_chunks, sizeof(chunk_t));
    h->cap_items = 2 * h->n_chunks + 4;
    h->items = (item_t **)calloc((size_t)h->cap_items, sizeof(item_t *));
    h->threads = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!h->chunks || !h->items || !h->threads) {
        free(h->chunks);
        free(h->items);
        free(h->threads);
        free(h);
        return NULL;
    }
    pthread_mutex_init(&h->mu, NULL);
    pthread_cond_init(&h->cv, NULL);
    for (int i = 0; i < threads; i++) {
        if (pthread_create(&h->threads[i], NULL, worker, h) != 0) break;
        h->threads_started++;
    }
    if (!h->threads_started) {
        pthread_mutex_destroy(&h->mu);
        pthread_cond_destroy(&h->cv);
        free(h->chunks);
        free(h->items);
        free(h->threads);
        free(h);
        return NULL;
    }
    return h;
}

/*
 * The next piece of the output, in stream order: *piece = malloc base (the bytes start at *piece + SCTIO_PIECE_HEAD; the caller owns
 * it: sctio_piece_free), returns its length, *crc its CRC-32.  *status = 0: a piece; 1: the stream has ended, no piece (*in_used =
 * bytes of the input the stream took, rounded up to whole bytes); < 0: -1 corrupt stream, -3 the input ends too early, -4 out of
 * memory — reported only after every piece in front of the error has been handed out, as the serial decoder would.
 */
int64_t sctio_pinflate_next(void *p, uint8_t **piece, uint32_t *crc, int32_t *status, int64_t *in_used) {
    pinf_t *h = (pinf_t *)p;
    *piece = NULL;
    pthread_mutex_lock(&h->mu);
    for (;;) {
        chain_advance(h);
        if (h->redo_wanted) pthread_cond_broadcast(&h->cv);
        if (h->cons < h->n_items) {
            item_t *it = h->items[h->cons];
            if (it->ready) {
                if (h->chain_error && !it->buf) break;
                *piece = it->buf;
                *crc = it->crc;
                const int64_t n = it->n;
                it->buf = NULL;
                h->cons++;
                *status = 0;
                pthread_cond_broadcast(&h->cv);   /* pass 1 may move on */
                pthread_mutex_unlock(&h->mu);
                return n;
            }
        } else if (h->chain_done) {
            break;
        }
        pthread_cond_wait(&h->cv, &h->mu);
    }
    if (h->chain_error) *status = h->chain_error;
    else {
        *status = 1;
        *in_used = (h->end_bit + 7) >> 3;
    }
    pthread_mutex_unlock(&h->mu);
    return 0;
}

/* pieces by origin so far: accepted from a speculative pass 1 / decoded again serially */
void sctio_pinflate_stats(void *p, int64_t *speculative, int64_t *redone) {
    pinf_t *h = (pinf_t *)p;
    pthread_mutex_lock(&h->mu);
    *speculative = h->speculative;
    *redone = h->redone;
    pthread_mutex_unlock(&h->mu);
}

void sctio_pinflate_free(void *p) {
    pinf_t *h = (pinf_t *)p;
    if (!h) return;
    pthread_mutex_lock(&h->mu);
    __atomic_store_n(&h->stop, 1, __ATOMIC_RELAXED);
    pthread_cond_broadcast(&h->cv);
    pthread_mutex_unlock(&h->mu);
    for (int i = 0; i < h->threads_started; i++) pthread_join(h->threads[i], NULL);
    for (int i = 0; i < h->n_chunks; i++) {
        free(h->chunks[i].sym.v);
        free(h->chunks[i].bytes);
    }
    for (int i = 0; i < h->n_pool; i++) free(h->pool[i].v);
    for (int i = 0; i < h->n_items; i++) {
        free(h->items[i]->buf);
        free(h->items[i]->sym.v);
        free(h->items[i]->win);
        free(h->items[i]);
    }
    pthread_mutex_destroy(&h->mu);
    pthread_cond_destroy(&h->cv);
    free(h->chunks);
    free(h->items);
    free(h->threads);
    free(h);
}

void sctio_piece_free(uint8_t *piece) { free(piece); }
int64_t sctio_piece_head(void) { return SCTIO_PIECE_HEAD; }
