/*
 * sct_inflate_impl.h — what sct_inflate.c (the serial DEFLATE decoder) and sct_pinflate.c (its parallel driver) share: the decoder
 * state, the table layout, bit-buffer helpers.  Internal to libsctagger_io.so.
 */
#pragma once
#include <stdint.h>
#include <string.h>

#define WINDOW 32768
#define LL_BITS 11
#define D_BITS 8
#define PRE_BITS 7
/* primary table + sub-tables: every sub-table belongs to one primary slot with at least one code longer than the primary
 * width, so there are at most as many as symbols, each of at most 2^(15 - width) entries */
#define LL_SIZE (2048 + 288 * 16)
#define D_SIZE (256 + 32 * 128)
#define PRE_SIZE 128

/* table entries: bits 0..7 = bits to take from the stream for this step, bits 8..11 = extra bits that follow a length / distance
 * code (or the index width of a sub-table), bits 12..15 = kind, bits 16..31 = literal, base value, or sub-table index */
#define K_LIT 0x1000u
#define K_EOB 0x2000u
#define K_SUB 0x4000u
#define K_BAD 0x8000u          /* unassigned code */

typedef struct {
    const uint8_t *in, *in_end, *in_next;
    uint64_t bitbuf;
    int bitcount;
    int phase;        /* 0 = at a block header, 1 = inside a Huffman block, 2 = inside a stored block, 3 = stream finished */
    int final_block;
    uint32_t stored_left;
    uint32_t win_fill; /* valid bytes in window[] */
    int64_t stop_bit;  /* >= 0: stop at the first block boundary at or behind this bit position of the input (status 2) */
    uint32_t ll[LL_SIZE], dd[D_SIZE];
    uint8_t window[WINDOW];
} sct_inf;


/* bits: make sure `n` (<= 32) are there; returns 0 if the input ran out */
static inline int need_bits(sct_inf *s, int n) {
    while (s->bitcount < n) {
        if (s->in_next >= s->in_end) return 0;
        s->bitbuf |= (uint64_t)*s->in_next++ << s->bitcount;
        s->bitcount += 8;
    }
    return 1;
}
static inline uint32_t take_bits(sct_inf *s, int n) {
    const uint32_t v = (uint32_t)(s->bitbuf & (((uint64_t)1 << n) - 1));
    s->bitbuf >>= n;
    s->bitcount -= n;
    return v;
}
static inline uint64_t load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
/* bit position of the next unread bit, counted from the start of the input */
static inline int64_t sctinf_bitpos(const sct_inf *s) { return (int64_t)(s->in_next - s->in) * 8 - s->bitcount; }

/* reads a block header at the current position: phase becomes 1 (tables built) or 2 (stored); 0, -1 corrupt, -3 input ends */
int sctinf_read_block_header(sct_inf *s);
/* start decoding at bit `bitpos` of in[0..len) with `win_fill` bytes of history (the last win_fill bytes of window[WINDOW]) */
void sctinf_start_at(sct_inf *s, const uint8_t *in, int64_t len, int64_t bitpos, const uint8_t *window, uint32_t win_fill);
