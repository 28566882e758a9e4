/*
 * sct_fdeflate.c — a fast DEFLATE (RFC 1951) compressor for the output of the streaming command (part of libsctagger_io.so).
 *
 * Why: the reference writes its TSV through gzip.open at level 9 (/root/reference/scTagger.py:770-772): 870 CPU-seconds for the
 * 10^8-row output.  zlib at level 4 — what this command used through round 3 — still takes 40 of the command's 56 CPU-seconds at
 * that size; on the 16 CPUs the GPU box grants, deflating the output is what the command waits for, not the device (50 ms) and not
 * the input (8 s of inflate).  The contract is the DECOMPRESSED bytes; so the default compressor is this one: a greedy LZ77 parse
 * with one hash probe per position (4-byte hash, matches taken from 6 bytes on, 32 KiB window), dynamic Huffman codes per block of 64 K
 * symbols, a stored block where that is not smaller.  About three times zlib level 4's speed on the output text and a slightly
 * SMALLER output (figures: DESIGN.md §10).  `--gzip-level 1..9` still goes through zlib.
 *
 * Every block is checked for size before it is written (dynamic Huffman vs stored), so a piece of n bytes never needs more than
 * n + n / 1000 + 64 bytes.  tests/test_fastio.py and tools/fuzz_inflate.c decode what this writes with zlib and with the
 * command's own decoders.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

uint32_t sctio_crc32(uint32_t crc, const uint8_t *p, int64_t n);   /* sct_pinflate.c */

#ifndef FD_HASH_BITS
#define FD_HASH_BITS 15
#endif
#define FD_HASH_SIZE (1 << FD_HASH_BITS)
#define FD_WINDOW 32768
#define FD_BLOCK_SYMS 65536
#define FD_MAX_MATCH 258

static const uint16_t fd_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t fd_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t fd_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t fd_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t fd_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static uint8_t fd_len_code[256];     /* length - 3 -> length code 0..28 */
static uint8_t fd_dist_code[512];    /* distance - 1 below 256 -> code; 256 + ((distance - 1) >> 7) for the others */
static pthread_once_t fd_tables_once = PTHREAD_ONCE_INIT;

static void fd_fill_tables(void) {
    for (int c = 0; c < 29; c++) {
        const int hi = c == 28 ? 258 : fd_len_base[c] + (1 << fd_len_extra[c]) - 1;
        for (int l = fd_len_base[c]; l <= hi && l <= 258; l++) fd_len_code[l - 3] = (uint8_t)c;
    }
    fd_len_code[255] = 28;   /* length 258 has a code of its own */
    for (int c = 0; c < 30; c++) {
        const int lo = fd_dist_base[c] - 1, hi = lo + (1 << fd_dist_extra[c]) - 1;
        for (int d = lo; d <= hi; d++) {
            if (d < 256) fd_dist_code[d] = (uint8_t)c;
            else fd_dist_code[256 + (d >> 7)] = (uint8_t)c;
        }
    }
}
/* (every compressing thread calls this first; pthread_once orders the table writes before every reader) */
static void fd_init_tables(void) { pthread_once(&fd_tables_once, fd_fill_tables); }
static inline int fd_dcode(uint32_t dist_minus_1) { return dist_minus_1 < 256 ? fd_dist_code[dist_minus_1] : fd_dist_code[256 + (dist_minus_1 >> 7)]; }

/* ---- Huffman code lengths (limited to `maxlen` bits) from frequencies ------------------------------------------------------- */
typedef struct {
    uint32_t freq;
    uint16_t sym;
} fd_node;

static int fd_cmp_node(const void *a, const void *b) {
    const fd_node *x = (const fd_node *)a, *y = (const fd_node *)b;
    if (x->freq != y->freq) return x->freq < y->freq ? -1 : 1;
    return (int)x->sym - (int)y->sym;
}

/* lens[0..n) <- code lengths, 0 for unused symbols; always a COMPLETE code with at least two codes (zlib's inflate refuses
 * incomplete sets except the single-code distance case; two dummy codes cost nothing) */
static void fd_code_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *lens) {
    fd_node nodes[288];
    int used = 0;
    memset(lens, 0, (size_t)n);
    for (int i = 0; i < n; i++)
        if (freq[i]) {
            nodes[used].freq = freq[i];
            nodes[used].sym = (uint16_t)i;
            used++;
        }
    if (used < 2) {   /* nothing or one symbol: that one (or symbol 0) and one more, one bit each */
        const int a = used ? nodes[0].sym : 0, b = a == 0 ? 1 : 0;
        lens[a] = 1;
        lens[b] = 1;
        return;
    }
    qsort(nodes, (size_t)used, sizeof(fd_node), fd_cmp_node);
    /* Huffman by two queues: leaves in ascending order, internal nodes are created in ascending order too */
    uint32_t wt[2 * 288];
    int16_t parent[2 * 288];
    for (int i = 0; i < used; i++) wt[i] = nodes[i].freq;
    int leaf = 0, inner = used, next = used;
    while (next < 2 * used - 1) {
        int a, b;
        a = (leaf < used && (inner >= next || wt[leaf] <= wt[inner])) ? leaf++ : inner++;
        b = (leaf < used && (inner >= next || wt[leaf] <= wt[inner])) ? leaf++ : inner++;
        wt[next] = wt[a] + wt[b];
        parent[a] = (int16_t)next;
        parent[b] = (int16_t)next;
        next++;
    }
    const int root = 2 * used - 2;
    int16_t depth[2 * 288];   /* (a degenerate frequency table gives depths up to the number of symbols) */
    depth[root] = 0;
    for (int i = root - 1; i >= 0; i--) depth[i] = (int16_t)(depth[parent[i]] + 1);
    /* how many codes of each length; lengths beyond maxlen are folded back (miniz's enforcement of a maximum code size) */
    int count[64] = {0};
    for (int i = 0; i < used; i++) count[depth[i] > 63 ? 63 : depth[i]]++;
    for (int l = maxlen + 1; l < 64; l++) {
        count[maxlen] += count[l];
        count[l] = 0;
    }
    uint32_t total = 0;
    for (int l = maxlen; l >= 1; l--) total += (uint32_t)count[l] << (maxlen - l);
    while (total != (1u << maxlen)) {   /* over-subscribed: lengthen the cheapest way until Kraft's sum is exactly one */
        count[maxlen]--;
        for (int l = maxlen - 1; l >= 1; l--)
            if (count[l]) {
                count[l]--;
                count[l + 1] += 2;
                break;
            }
        total--;
    }
    /* the most frequent symbols get the shortest codes */
    int at = used - 1;
    for (int l = 1; l <= maxlen; l++)
        for (int k = 0; k < count[l]; k++) lens[nodes[at--].sym] = (uint8_t)l;
}

/* canonical codes, bit-reversed (DEFLATE writes Huffman codes most significant bit first into an LSB-first stream) */
static void fd_codes(const uint8_t *lens, int n, uint16_t *codes) {
    int count[16] = {0};
    uint32_t next[16], code = 0;
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (!l) {
            codes[i] = 0;
            continue;
        }
        uint32_t c = next[l]++, r = 0;
        for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);
        codes[i] = (uint16_t)r;
    }
}

/* ---- bit writer ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t bits;
    int n;
    uint8_t *p;
} fd_bw;
static inline void fd_put(fd_bw *w, uint32_t v, int len) {   /* len <= 32 in all; the buffer is flushed to below 32 bits */
    w->bits |= (uint64_t)v << w->n;
    w->n += len;
    if (w->n >= 32) {
        const uint32_t x = (uint32_t)w->bits;
        memcpy(w->p, &x, 4);
        w->p += 4;
        w->bits >>= 32;
        w->n -= 32;
    }
}
static inline void fd_flush_bytes(fd_bw *w) {   /* to a byte boundary */
    while (w->n > 0) {
        *w->p++ = (uint8_t)w->bits;
        w->bits >>= 8;
        w->n -= 8;
    }
    w->bits = 0;
    w->n = 0;
}

/* ---- one block: the symbols sym[0..ns) covering raw[0..raw_len) --------------------------------------------------------------- */
typedef struct {
    uint32_t lit[286], dist[30];
} fd_freqs;

static void fd_write_block(fd_bw *w, const uint32_t *sym, int ns, const fd_freqs *f, const uint8_t *raw, int64_t raw_len, int final) {
    uint8_t ll[288], dl[32];
    uint16_t lc[288], dc[32];
    uint32_t lf[286];
    memcpy(lf, f->lit, sizeof(lf));
    lf[256] = 1;   /* end of block */
    fd_code_lengths(lf, 286, 15, ll);
    fd_code_lengths(f->dist, 30, 15, dl);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) hlit--;
    while (hdist > 1 && dl[hdist - 1] == 0) hdist--;
    /* the code lengths, run-length coded with the code-length alphabet */
    uint8_t all[286 + 30], rl_sym[286 + 30], rl_extra[286 + 30];
    memcpy(all, ll, (size_t)hlit);
    memcpy(all + hlit, dl, (size_t)hdist);
    const int na = hlit + hdist;
    int nr = 0;
    uint32_t cf[19] = {0};
    for (int i = 0; i < na;) {
        int run = 1;
        while (i + run < na && all[i + run] == all[i]) run++;
        const int v = all[i];
        i += run;
        while (run > 0) {
            if (v == 0 && run >= 11) {
                const int k = run > 138 ? 138 : run;
                rl_sym[nr] = 18;
                rl_extra[nr++] = (uint8_t)(k - 11);
                run -= k;
            } else if (v == 0 && run >= 3) {
                rl_sym[nr] = 17;
                rl_extra[nr++] = (uint8_t)(run - 3);
                run = 0;
            } else if (v != 0 && run >= 4) {   /* the value itself, then "repeat the previous" 3..6 times */
                rl_sym[nr] = (uint8_t)v;
                rl_extra[nr++] = 0;
                run--;
                while (run >= 3) {
                    const int k = run > 6 ? 6 : run;
                    rl_sym[nr] = 16;
                    rl_extra[nr++] = (uint8_t)(k - 3);
                    run -= k;
                }
            } else {
                rl_sym[nr] = (uint8_t)v;
                rl_extra[nr++] = 0;
                run--;
            }
        }
    }
    for (int i = 0; i < nr; i++) cf[rl_sym[i]]++;
    uint8_t cl[19];
    uint16_t cc[19];
    fd_code_lengths(cf, 19, 7, cl);
    fd_codes(cl, 19, cc);
    int hclen = 19;
    while (hclen > 4 && cl[fd_clen_order[hclen - 1]] == 0) hclen--;
    fd_codes(ll, 286, lc);
    fd_codes(dl, 30, dc);
    /* size of the block as dynamic Huffman, against stored */
    uint64_t bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
    for (int i = 0; i < nr; i++) bits += cl[rl_sym[i]] + (rl_sym[i] == 16 ? 2 : rl_sym[i] == 17 ? 3 : rl_sym[i] == 18 ? 7 : 0);
    for (int i = 0; i < 286; i++) bits += (uint64_t)lf[i] * ll[i];
    for (int i = 257; i < 286; i++) bits += (uint64_t)f->lit[i] * fd_len_extra[i - 257];
    for (int i = 0; i < 30; i++) bits += (uint64_t)f->dist[i] * (dl[i] + fd_dist_extra[i]);
    const uint64_t stored_bytes = (uint64_t)raw_len + 5 * (uint64_t)((raw_len + 65534) / 65535 + (raw_len == 0)) + 1;
    if ((bits + 7) / 8 >= stored_bytes) {
        /* stored blocks of at most 65 535 bytes */
        int64_t at = 0;
        do {
            const int64_t k = raw_len - at > 65535 ? 65535 : raw_len - at;
            fd_put(w, (uint32_t)(final && at + k == raw_len), 1);
            fd_put(w, 0, 2);
            fd_flush_bytes(w);
            *w->p++ = (uint8_t)k;
            *w->p++ = (uint8_t)(k >> 8);
            *w->p++ = (uint8_t)~k;
            *w->p++ = (uint8_t)(~k >> 8);
            memcpy(w->p, raw + at, (size_t)k);
            w->p += k;
            at += k;
        } while (at < raw_len);
        return;
    }
    fd_put(w, (uint32_t)final, 1);
    fd_put(w, 2, 2);
    fd_put(w, (uint32_t)(hlit - 257), 5);
    fd_put(w, (uint32_t)(hdist - 1), 5);
    fd_put(w, (uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; i++) fd_put(w, cl[fd_clen_order[i]], 3);
    for (int i = 0; i < nr; i++) {
        fd_put(w, cc[rl_sym[i]], cl[rl_sym[i]]);
        if (rl_sym[i] == 16) fd_put(w, rl_extra[i], 2);
        else if (rl_sym[i] == 17) fd_put(w, rl_extra[i], 3);
        else if (rl_sym[i] == 18) fd_put(w, rl_extra[i], 7);
    }
    /* The symbols.  More than half of the compressor's time was here with one fd_put per code (two table reads and a conditional
     * flush each); the rows are mostly literals (DNA), so: code and length of a literal in one table word, those of a match length
     * with its extra bits folded in, up to three literals (45 bits) or one match (48) between two unconditional 8-byte stores. */
    uint32_t lit_e[256], len_e[256];   /* bits 0..23 the code (+ extra bits), 24..31 its width */
    for (int i = 0; i < 256; i++) lit_e[i] = (uint32_t)lc[i] | ((uint32_t)ll[i] << 24);
    for (int l3 = 0; l3 < 256; l3++) {
        const int c = fd_len_code[l3];
        len_e[l3] = ((uint32_t)lc[257 + c] | ((uint32_t)(l3 + 3 - fd_len_base[c]) << ll[257 + c])) | ((uint32_t)(ll[257 + c] + fd_len_extra[c]) << 24);
    }
    while (w->n >= 8) {   /* whole bytes out; up to 7 bits stay */
        *w->p++ = (uint8_t)w->bits;
        w->bits >>= 8;
        w->n -= 8;
    }
    uint64_t acc = w->bits;
    int nb = w->n;
    uint8_t *p = w->p;
#define FD_ADD(v, l) (acc |= (uint64_t)(v) << nb, nb += (int)(l))
#define FD_STORE() (memcpy(p, &acc, 8), p += nb >> 3, acc >>= (nb & ~7), nb &= 7)
    for (int i = 0; i < ns;) {
        uint32_t s = sym[i++];
        if (!(s & 0x80000000u)) {
            uint32_t e = lit_e[s];
            FD_ADD(e & 0xFFFFFFu, e >> 24);
            if (i < ns && !((s = sym[i]) & 0x80000000u)) {
                e = lit_e[s];
                FD_ADD(e & 0xFFFFFFu, e >> 24);
                i++;
                if (i < ns && !((s = sym[i]) & 0x80000000u)) {
                    e = lit_e[s];
                    FD_ADD(e & 0xFFFFFFu, e >> 24);
                    i++;
                }
            }
        } else {
            const uint32_t d1 = s & 0x7FFFu, e = len_e[(s >> 15) & 0xFFu];
            const int d = fd_dcode(d1);
            FD_ADD(e & 0xFFFFFFu, e >> 24);
            FD_ADD(dc[d], dl[d]);
            FD_ADD(d1 + 1 - fd_dist_base[d], fd_dist_extra[d]);
        }
        FD_STORE();
    }
    FD_ADD(lc[256], ll[256]);
    FD_STORE();
#undef FD_ADD
#undef FD_STORE
    w->bits = acc;
    w->n = nb;
    w->p = p;
}

static inline uint64_t fd_load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
/* Candidates are found by the FOUR bytes at a position (the table keeps the latest position of every 4-gram), but a match is taken
 * only from FD_MIN_ACCEPT = 6 bytes on: the rows' segments are DNA, where four equal bytes turn up by chance all the time, and a
 * short match costs more bits than the literals it replaces.  Measured on the command's output text (ratio; speed is the same):
 * accept from 4: 3.43, from 6: 3.79, from 7: 3.75, from 8: 3.65; hashing 5 / 6 bytes instead: 3.58 / 3.46; zlib level 1 / 4: 3.28 / 3.61. */
#define FD_MIN_ACCEPT 6
/* ... and only if it is cheaper than its bytes as literals (fd_worth): a match costs about FD_MATCH_BASE_BITS (length code + distance
 * code) + the distance's extra bits, a letter of a segment 2.25 bits, another character about 5.  Ratio 3.80 -> 3.90 (base 10; 9: 3.89,
 * 11: 3.88, 12: 3.89) at the same speed.  Hashing more bytes finds a candidate of six letters at almost every position of a segment
 * (4^6 six-grams, 16 K of them in the window) and spends its time rejecting them: 5 bytes 140 MB/s, 6 bytes 112 against 226. */
#ifndef FD_MATCH_BASE_BITS
#define FD_MATCH_BASE_BITS 10
#endif
#ifndef FD_HASH_BYTES
#define FD_HASH_BYTES 4
#endif
static inline uint32_t fd_hash64(uint64_t v) { return (uint32_t)(((v << (64 - 8 * FD_HASH_BYTES)) * 0x9E3779B185EBCA87ull) >> (64 - FD_HASH_BITS)); }
static inline int fd_worth(uint64_t v8, int64_t len, int64_t dist) {
    /* literal cost of the matched bytes, in eighths of a bit: 18 for a letter, 40 for the other characters of a row (digits, tabs,
     * commas, line ends: bit 6 clear), counted among the first eight bytes — the rest are priced as letters */
    const uint64_t m = ~v8 & 0x4040404040404040ull & (len >= 8 ? ~0ull : ((1ull << (8 * len)) - 1));
    const int cost = 18 * (int)len + 22 * __builtin_popcountll(m);
    const int dbits = dist < 5 ? 0 : (63 - __builtin_clzll((uint64_t)dist - 1)) - 1;   /* extra bits of the distance code */
    return cost > 8 * (FD_MATCH_BASE_BITS + dbits);
}

/* scratch of one compression (per thread): the hash table and a block's symbols */
typedef struct {
    int32_t head[FD_HASH_SIZE];
    uint32_t sym[FD_BLOCK_SYMS];
} fd_scratch;

int64_t sctfd_scratch_bytes(void) { return (int64_t)sizeof(fd_scratch); }

/*
 * in[0 .. n) as ONE gzip member at out[0 .. return value); out must hold n + n / 1000 + 64 bytes.  scratch: sctfd_scratch_bytes()
 * bytes, contents irrelevant.  n < 2^31.
 */
int64_t sctfd_gzip_member(const uint8_t *in, int64_t n, uint8_t *out, void *scratch_mem) {
    fd_init_tables();
    fd_scratch *S = (fd_scratch *)scratch_mem;
    static const uint8_t hdr[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 4, 255};   /* XFL 4 = "fastest algorithm", OS unknown */
    memcpy(out, hdr, 10);
    fd_bw w;
    w.bits = 0;
    w.n = 0;
    w.p = out + 10;
    for (int i = 0; i < FD_HASH_SIZE; i++) S->head[i] = -FD_WINDOW - 1;
    fd_freqs f;
    memset(&f, 0, sizeof(f));
    uint32_t lit2[2][256];   /* literal counts by the parity of the symbol's index: neighbours never wait for each other's increment */
    memset(lit2, 0, sizeof(lit2));
    int ns = 0;
    int64_t i = 0, block_start = 0;
    const int64_t last_hash = n - 8;   /* positions from which 8 bytes can be loaded */
    while (i < n) {
        if (ns >= FD_BLOCK_SYMS - 1) {
            for (int c = 0; c < 256; c++) f.lit[c] = lit2[0][c] + lit2[1][c];
            fd_write_block(&w, S->sym, ns, &f, in + block_start, i - block_start, 0);
            memset(&f, 0, sizeof(f));
            memset(lit2, 0, sizeof(lit2));
            ns = 0;
            block_start = i;
        }
        if (i <= last_hash) {
            const uint64_t v8 = fd_load64(in + i);
            const uint32_t h = fd_hash64(v8);
            const int64_t cand = S->head[h];
            S->head[h] = (int32_t)i;
            uint64_t x0;
            if (i - cand <= FD_WINDOW && (((x0 = fd_load64(in + cand) ^ v8) << (64 - 8 * FD_HASH_BYTES)) == 0)) {
                /* extend, eight bytes at a time */
                int64_t len;
                const int64_t max = n - i < FD_MAX_MATCH ? n - i : FD_MAX_MATCH;
                if (x0) {
                    len = __builtin_ctzll(x0) >> 3;
                    goto matched;
                }
                len = 8;
                while (len + 8 <= max) {
                    const uint64_t x = fd_load64(in + i + len) ^ fd_load64(in + cand + len);
                    if (x) {
                        len += __builtin_ctzll(x) >> 3;
                        goto matched;
                    }
                    len += 8;
                }
                while (len < max && in[i + len] == in[cand + len]) len++;
            matched:
                if (len < FD_MIN_ACCEPT || !fd_worth(v8, len, i - cand)) goto literal;
                S->sym[ns++] = 0x80000000u | ((uint32_t)(len - 3) << 15) | (uint32_t)(i - cand - 1);
                f.lit[257 + fd_len_code[len - 3]]++;
                f.dist[fd_dcode((uint32_t)(i - cand - 1))]++;
                /* a few positions of the match go into the table too (the ends: where the next match is likeliest to start) */
                if (i + len <= last_hash) {
                    S->head[fd_hash64(fd_load64(in + i + 1))] = (int32_t)(i + 1);
                    S->head[fd_hash64(fd_load64(in + i + len - 2))] = (int32_t)(i + len - 2);
                    S->head[fd_hash64(fd_load64(in + i + len - 1))] = (int32_t)(i + len - 1);
                }
                i += len;
                continue;
            }
        }
    literal:
        lit2[ns & 1][in[i]]++;
        S->sym[ns++] = in[i];
        i++;
    }
    for (int c = 0; c < 256; c++) f.lit[c] = lit2[0][c] + lit2[1][c];
    fd_write_block(&w, S->sym, ns, &f, in + block_start, n - block_start, 1);
    fd_flush_bytes(&w);
    const uint32_t crc = sctio_crc32(0, in, n), isize = (uint32_t)n;
    memcpy(w.p, &crc, 4);
    memcpy(w.p + 4, &isize, 4);
    w.p += 8;
    return w.p - out;
}
