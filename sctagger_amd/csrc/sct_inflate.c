/*
 * sct_inflate.c — a DEFLATE (RFC 1951) decoder for the streaming match_trie command (part of libsctagger_io.so).
 *
 * Why: the command's wall time at 10^7 rows is the serial inflate of the -lr file (zlib: ~0.39 GB/s of output); everything
 * else hides behind it (sctagger_amd/streamcli.py).  This decoder is written for that one job — the whole compressed file is
 * in memory, the output goes into large caller-provided blocks — with the usual means of the fast decoders: a 64-bit bit
 * buffer refilled eight bytes at a time, one table look-up per symbol (11-bit primary table for literals / lengths, 8-bit
 * for distances, sub-tables for longer codes), up to three literals per refill, matches copied eight bytes at a time.
 * Near the end of the input or of an output block a careful byte-wise loop takes over.
 *
 * Output is chunked: sctio_inflate_run fills the caller's block and stops at a symbol boundary when it is (nearly) full;
 * the decoder keeps the last 32 KiB it wrote and puts them in front of the next block, where the next matches find them.
 * Test infrastructure compares it with zlib on every kind of stream (tests/test_fastio.py); corrupt input is reported, never
 * read or written out of bounds.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sct_inflate_impl.h"

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint32_t rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

/* what a symbol's entry looks like, without the step width in the low byte */
static uint32_t ll_payload(int sym) {
    if (sym < 256) return K_LIT | ((uint32_t)sym << 16);
    if (sym == 256) return K_EOB;
    if (sym > 285) return K_BAD;
    return ((uint32_t)len_base[sym - 257] << 16) | ((uint32_t)len_extra[sym - 257] << 8);
}
static uint32_t d_payload(int sym) {
    if (sym > 29) return K_BAD;
    return ((uint32_t)dist_base[sym] << 16) | ((uint32_t)dist_extra[sym] << 8);
}

/*
 * Canonical Huffman code -> look-up table.  lens[0..n) code lengths (0 = unused, <= 15).  kind: 0 = literal/length, 1 = distance,
 * 2 = code-length code (payload = symbol << 16).  Returns 0, or -1 if the lengths over-subscribe the code space, the table does
 * not fit, or the code is incomplete — zlib's rule (inftrees.c), so that the command refuses exactly the streams gzip.open in the
 * reference refuses: an incomplete literal/length or distance code is accepted only if it consists of ONE code of ONE bit (or of
 * no code at all: a block without matches), a code-length code never.  Unassigned slots decode to K_BAD.
 */
static int build_table(uint32_t *table, int table_size, int tbits, const uint8_t *lens, int n, int kind) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    uint32_t code = 0, next[16];
    int64_t space = 1;
    int max_len = 0;
    for (int l = 1; l <= 15; l++) {
        space = space * 2 - count[l];
        if (space < 0) return -1;
        if (count[l]) max_len = l;
        code = (code + (l > 1 ? (uint32_t)count[l - 1] : 0u)) << 1;
        next[l] = code;
    }
    if (space > 0 && max_len != 0 && (kind == 2 || max_len != 1)) return -1;   /* incomplete set */
    const int psize = 1 << tbits;
    for (int i = 0; i < psize; i++) table[i] = K_BAD | 1u;
    /* pass 1: sub-table widths per primary prefix */
    uint8_t *subw = (uint8_t *)calloc((size_t)psize, 1);
    uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    if (!subw || !rev) {
        free(subw);
        free(rev);
        return -1;
    }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        rev[s] = rev_bits(next[l]++, l);
        if (l > tbits) {
            const uint32_t p = rev[s] & (uint32_t)(psize - 1);
            if (l - tbits > subw[p]) subw[p] = (uint8_t)(l - tbits);
        }
    }
    int used = psize;
    for (int p = 0; p < psize; p++) {
        if (!subw[p]) continue;
        const int ssize = 1 << subw[p];
        if (used + ssize > table_size) {
            free(subw);
            free(rev);
            return -1;
        }
        table[p] = K_SUB | ((uint32_t)used << 16) | ((uint32_t)subw[p] << 8) | (uint32_t)tbits;
        for (int i = 0; i < ssize; i++) table[used + i] = K_BAD | 1u;
        used += ssize;
    }
    /* pass 2: entries */
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t pay = kind == 0 ? ll_payload(s) : kind == 1 ? d_payload(s) : ((uint32_t)s << 16);
        if (l <= tbits) {
            for (uint32_t i = rev[s]; i < (uint32_t)psize; i += 1u << l) table[i] = pay | (uint32_t)l;
        } else {
            const uint32_t p = rev[s] & (uint32_t)(psize - 1);
            const uint32_t start = table[p] >> 16, w = subw[p];
            for (uint32_t i = rev[s] >> tbits; i < (1u << w); i += 1u << (l - tbits)) table[start + i] = pay | (uint32_t)(l - tbits);
        }
    }
    free(subw);
    free(rev);
    return 0;
}

void *sctio_inflate_new(void) { return calloc(1, sizeof(sct_inf)); }
void sctio_inflate_free(void *p) { free(p); }

/* start a raw DEFLATE stream at in[0] (what follows a gzip member's header) */
void sctio_inflate_set_input(void *p, const uint8_t *in, int64_t len) {
    sct_inf *s = (sct_inf *)p;
    s->in = s->in_next = in;
    s->in_end = in + len;
    s->bitbuf = 0;
    s->bitcount = 0;
    s->phase = 0;
    s->final_block = 0;
    s->stored_left = 0;
    s->win_fill = 0;
    s->stop_bit = -1;
}

/* stop at the first block boundary at or behind bit `stop_bit` of the input (status 2 of sctio_inflate_run); -1 = never */
void sctio_inflate_set_stop(void *p, int64_t stop_bit) { ((sct_inf *)p)->stop_bit = stop_bit; }
/* where the decoder stands: bit position of the next unread bit, and its history (window[WINDOW], the last *win_fill bytes valid) */
void sctio_inflate_get_state(void *p, int64_t *bitpos, uint8_t *window, uint32_t *win_fill) {
    const sct_inf *s = (const sct_inf *)p;
    *bitpos = sctinf_bitpos(s);
    memcpy(window, s->window, WINDOW);
    *win_fill = s->win_fill;
}

void sctinf_start_at(sct_inf *s, const uint8_t *in, int64_t len, int64_t bitpos, const uint8_t *window, uint32_t win_fill) {
    sctio_inflate_set_input(s, in, len);
    s->in_next = in + (bitpos >> 3);
    if (s->in_next > s->in_end) s->in_next = s->in_end;
    if (bitpos & 7) {
        if (need_bits(s, (int)(bitpos & 7))) take_bits(s, (int)(bitpos & 7));
    }
    if (win_fill) memcpy(s->window + WINDOW - win_fill, window + WINDOW - win_fill, win_fill);
    s->win_fill = win_fill;
}

int sctinf_read_block_header(sct_inf *s) {
    if (!need_bits(s, 3)) return -3;
    s->final_block = (int)take_bits(s, 1);
    const uint32_t type = take_bits(s, 2);
    if (type == 0) {
        /* stored: to the next byte boundary; the bit buffer holds whole bytes read ahead — give them back */
        take_bits(s, s->bitcount & 7);
        s->in_next -= s->bitcount >> 3;
        s->bitbuf = 0;
        s->bitcount = 0;
        if (s->in_end - s->in_next < 4) return -3;
        const uint32_t len = (uint32_t)s->in_next[0] | ((uint32_t)s->in_next[1] << 8), nlen = (uint32_t)s->in_next[2] | ((uint32_t)s->in_next[3] << 8);
        if ((len ^ 0xFFFFu) != nlen) return -1;
        s->in_next += 4;
        s->stored_left = len;
        s->phase = 2;
        return 0;
    }
    if (type == 3) return -1;
    uint8_t lens[288 + 32];
    int nlit, ndist;
    if (type == 1) {
        nlit = 288;
        ndist = 32;
        for (int i = 0; i < 144; i++) lens[i] = 8;
        for (int i = 144; i < 256; i++) lens[i] = 9;
        for (int i = 256; i < 280; i++) lens[i] = 7;
        for (int i = 280; i < 288; i++) lens[i] = 8;
        for (int i = 0; i < 32; i++) lens[288 + i] = 5;
    } else {
        if (!need_bits(s, 14)) return -3;
        nlit = (int)take_bits(s, 5) + 257;
        ndist = (int)take_bits(s, 5) + 1;
        const int npre = (int)take_bits(s, 4) + 4;
        if (nlit > 286 || ndist > 30) return -1;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t pl[19] = {0};
        for (int i = 0; i < npre; i++) {
            if (!need_bits(s, 3)) return -3;
            pl[order[i]] = (uint8_t)take_bits(s, 3);
        }
        uint32_t pre[PRE_SIZE];
        if (build_table(pre, PRE_SIZE, PRE_BITS, pl, 19, 2)) return -1;
        int i = 0;
        while (i < nlit + ndist) {
            if (!need_bits(s, 7 + 7) && s->bitcount < 1) return -3;   /* the code (<= 7) plus its repeat count (<= 7) */
            const uint32_t e = pre[s->bitbuf & (PRE_SIZE - 1)];
            if (e & K_BAD) return -1;
            if ((int)(e & 0xFF) > s->bitcount) return -3;
            take_bits(s, (int)(e & 0xFF));
            const int sym = (int)(e >> 16);
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
                continue;
            }
            int rep, val = 0;
            if (sym == 16) {
                if (i == 0) return -1;
                if (!need_bits(s, 2)) return -3;
                val = lens[i - 1];
                rep = 3 + (int)take_bits(s, 2);
            } else if (sym == 17) {
                if (!need_bits(s, 3)) return -3;
                rep = 3 + (int)take_bits(s, 3);
            } else {
                if (!need_bits(s, 7)) return -3;
                rep = 11 + (int)take_bits(s, 7);
            }
            if (i + rep > nlit + ndist) return -1;
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) return -1;  /* no end-of-block code */
        /* the distance lengths follow the literal/length ones directly: move them to where the fixed layout has them */
        memmove(lens + 288, lens + nlit, (size_t)ndist);
        memset(lens + nlit, 0, (size_t)(288 - nlit));
        memset(lens + 288 + ndist, 0, (size_t)(32 - ndist));
    }
    if (build_table(s->ll, LL_SIZE, LL_BITS, lens, 288, 0)) return -1;
    if (build_table(s->dd, D_SIZE, D_BITS, lens + 288, 32, 1)) return -1;
    s->phase = 1;
    return 0;
}

/*
 * Decode into out[WINDOW .. WINDOW + cap): the first WINDOW bytes of `out` receive the previous 32 KiB of output (history for the
 * matches).  Returns the number of bytes produced; *status = 0: the block is full, call again with the next one; 1: the stream
 * ended (*in_used = bytes of the input consumed, rounded up to whole bytes); < 0: -1 corrupt stream, -3 the input ends too early.
 * cap must be at least 64 KiB.
 */
int64_t sctio_inflate_run(void *p, uint8_t *out, int64_t cap, int32_t *status, int64_t *in_used) {
    sct_inf *s = (sct_inf *)p;
    uint8_t *const hist = out + WINDOW - s->win_fill;     /* oldest byte a match may reach */
    memcpy(hist, s->window + WINDOW - s->win_fill, s->win_fill);
    uint8_t *o = out + WINDOW, *const o_begin = o, *const o_end = o + cap;
    int rc = 0, stopped = 0;
    *status = 0;
    if (cap < 65536) {
        *status = -1;
        return 0;
    }
    while (s->phase != 3) {
        if (s->phase == 0) {
            rc = sctinf_read_block_header(s);
            if (rc) break;
            continue;
        }
        if (s->phase == 2) {
            uint32_t k = s->stored_left;
            if ((int64_t)k > o_end - o) k = (uint32_t)(o_end - o);
            if ((int64_t)k > s->in_end - s->in_next) {
                rc = -3;
                break;
            }
            memcpy(o, s->in_next, k);
            o += k;
            s->in_next += k;
            s->stored_left -= k;
            if (s->stored_left) break;  /* block full */
            s->phase = s->final_block ? 3 : 0;
            if (s->phase == 0 && s->stop_bit >= 0 && sctinf_bitpos(s) >= s->stop_bit) {
                stopped = 1;
                break;
            }
            continue;
        }
        /* ---- a Huffman block ---- */
        uint64_t bb = s->bitbuf;
        int bc = s->bitcount;
        const uint8_t *in = s->in_next;
        const uint8_t *const in_fast_end = s->in_end - 16;
        int done = 0;   /* 1 = end of block, 2 = output (nearly) full */
        /* fast loop: >= 16 bytes of input and >= 320 bytes of output room ahead */
        while (in <= in_fast_end && o_end - o >= 320) {
            bb |= load64(in) << bc;   /* bytes already in the buffer are OR-ed onto themselves */
            in += (63 - bc) >> 3;
            bc |= 56;
            uint32_t e = s->ll[bb & ((1u << LL_BITS) - 1)];
            if (e & K_LIT) {          /* up to three literals on one refill (3 x 15 bits) */
                bb >>= (e & 0xFF);
                bc -= (int)(e & 0xFF);
                *o++ = (uint8_t)(e >> 16);
                e = s->ll[bb & ((1u << LL_BITS) - 1)];
                if (e & K_LIT) {
                    bb >>= (e & 0xFF);
                    bc -= (int)(e & 0xFF);
                    *o++ = (uint8_t)(e >> 16);
                    e = s->ll[bb & ((1u << LL_BITS) - 1)];
                    if (e & K_LIT) {
                        bb >>= (e & 0xFF);
                        bc -= (int)(e & 0xFF);
                        *o++ = (uint8_t)(e >> 16);
                        continue;
                    }
                }
                if (bc < 48) continue;   /* not enough left for a length + distance: refill first */
            }
            if (e & K_SUB) {
                bb >>= LL_BITS;
                bc -= LL_BITS;
                e = s->ll[(e >> 16) + (bb & ((1u << ((e >> 8) & 0xF)) - 1))];
                if (e & K_LIT) {
                    bb >>= (e & 0xFF);
                    bc -= (int)(e & 0xFF);
                    *o++ = (uint8_t)(e >> 16);
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
            /* a length */
            bb >>= (e & 0xFF);
            bc -= (int)(e & 0xFF);
            const int xb = (int)((e >> 8) & 0xF);
            uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xb) - 1));
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
            if ((int64_t)dist > o - hist) {
                rc = -1;
                break;
            }
            const uint8_t *src = o - dist;
            uint8_t *dst = o;
            o += len;
            if (dist >= 8) {
                do {   /* may write up to 7 bytes past the match: the fast loop keeps 320 bytes of room */
                    memcpy(dst, src, 8);
                    dst += 8;
                    src += 8;
                } while (dst < o);
            } else if (dist == 1) {
                memset(dst, *src, len);
            } else {
                while (dst < o) *dst++ = *src++;
            }
        }
        /* give the whole bytes that were only looked at back to the input, then go on carefully */
        {
            const int back = bc >> 3;
            in -= back;
            bc -= back * 8;
            bb &= bc ? (((uint64_t)1 << bc) - 1) : 0;
        }
        s->bitbuf = bb;
        s->bitcount = bc;
        s->in_next = in;
        if (rc) break;
        /* careful loop: one symbol at a time, every access checked */
        while (!done) {
            if (o_end - o < 258 + 8) {
                done = 2;
                break;
            }
            need_bits(s, 15);
            uint32_t e = s->ll[s->bitbuf & ((1u << LL_BITS) - 1)];
            if (e & K_SUB) {
                if (s->bitcount < LL_BITS) {
                    rc = -3;
                    break;
                }
                take_bits(s, LL_BITS);
                need_bits(s, 15);
                e = s->ll[(e >> 16) + (s->bitbuf & ((1u << ((e >> 8) & 0xF)) - 1))];
            }
            if (e & K_BAD) {
                rc = s->in_next >= s->in_end && s->bitcount < 15 ? -3 : -1;
                break;
            }
            if ((int)(e & 0xFF) > s->bitcount) {
                rc = -3;
                break;
            }
            take_bits(s, (int)(e & 0xFF));
            if (e & K_LIT) {
                *o++ = (uint8_t)(e >> 16);
                continue;
            }
            if (e & K_EOB) {
                done = 1;
                break;
            }
            const int xb = (int)((e >> 8) & 0xF);
            if (!need_bits(s, xb)) {
                rc = -3;
                break;
            }
            const uint32_t len = (e >> 16) + take_bits(s, xb);
            need_bits(s, 15);
            uint32_t d = s->dd[s->bitbuf & ((1u << D_BITS) - 1)];
            if (d & K_SUB) {
                if (s->bitcount < D_BITS) {
                    rc = -3;
                    break;
                }
                take_bits(s, D_BITS);
                need_bits(s, 15);
                d = s->dd[(d >> 16) + (s->bitbuf & ((1u << ((d >> 8) & 0xF)) - 1))];
            }
            if (d & (K_BAD | K_LIT | K_EOB)) {
                rc = (d & K_BAD) && s->in_next >= s->in_end && s->bitcount < 15 ? -3 : -1;
                break;
            }
            if ((int)(d & 0xFF) > s->bitcount) {
                rc = -3;
                break;
            }
            take_bits(s, (int)(d & 0xFF));
            const int db = (int)((d >> 8) & 0xF);
            if (!need_bits(s, db)) {
                rc = -3;
                break;
            }
            const uint32_t dist = (d >> 16) + take_bits(s, db);
            if ((int64_t)dist > o - hist) {
                rc = -1;
                break;
            }
            const uint8_t *src = o - dist;
            for (uint32_t i = 0; i < len; i++) o[i] = src[i];
            o += len;
        }
        if (rc) break;
        if (done == 1) s->phase = s->final_block ? 3 : 0;
        if (done == 2) break;
        if (s->phase == 0 && s->stop_bit >= 0 && sctinf_bitpos(s) >= s->stop_bit) {
            stopped = 1;
            break;
        }
    }
    /* remember the last 32 KiB for the next block's matches */
    {
        const int64_t have = (o - o_begin) + (int64_t)s->win_fill;
        const uint32_t keep = have > WINDOW ? WINDOW : (uint32_t)have;
        memmove(s->window + WINDOW - keep, o - keep, keep);
        s->win_fill = keep;
    }
    if (rc) *status = rc;
    else if (stopped) *status = 2;   /* at a block boundary at or behind stop_bit (sctinf_bitpos says where) */
    else if (s->phase == 3) {
        *status = 1;
        /* whole bytes still in the bit buffer were not consumed */
        *in_used = (s->in_next - s->in) - (s->bitcount >> 3);
    }
    return o - o_begin;
}
