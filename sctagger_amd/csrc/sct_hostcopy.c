/* sct_hostcopy.c — the loops the copy threads of the page-locked ring run (sct_stream.hip.h): plain C, built with gcc, one clone per
 * instruction set (AVX-512, AVX2, baseline), picked when the library is loaded (ifunc).  The reference hands its seam 32-bit offsets
 * (scTagger.py:618: RawArray('i')) and takes a dict; this library's sct_match takes 64-bit offsets and returns 64-bit CSR offsets:
 * what crosses the PCIe link in between is one byte per row either way, and these loops are the two conversions. */
#include <stddef.h>
#include <stdint.h>

#ifndef SCT_CLONES   /* (tests/test_hostcopy.py builds every clone on its own) */
#define SCT_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif

/* offsets -> one-byte row lengths; returns the OR of all differences: a bit above bit 7 means a row of more than 255 characters
 * (or, top bit, offsets that decrease) — the caller then looks again */
SCT_CLONES uint64_t sct_narrow_offs64(const int64_t *offs, int64_t cnt, uint8_t *out) {
    uint64_t acc = 0;
    for (int64_t i = 0; i < cnt; i++) {
        const uint64_t d = (uint64_t)offs[i + 1] - (uint64_t)offs[i];
        out[i] = (uint8_t)d;
        acc |= d;
    }
    return acc;
}

SCT_CLONES uint64_t sct_narrow_offs32(const int32_t *offs, int64_t cnt, uint8_t *out) {
    uint32_t acc = 0;
    for (int64_t i = 0; i < cnt; i++) {
        const uint32_t d = (uint32_t)offs[i + 1] - (uint32_t)offs[i];
        out[i] = (uint8_t)d;
        acc |= d;
    }
    /* (the sign bit of a 32-bit difference moves to bit 63: "decreasing" reads the same for both widths) */
    return (uint64_t)(acc & 0x7FFFFFFFu) | ((uint64_t)(acc >> 31) << 63);
}

/* one-byte hit counts -> 64-bit CSR offsets: dst[i] = base + c8[0] + ... + c8[i-1], i = 0 .. n-1 */
void sct_counts8_to_offs64(const uint8_t *c8, int64_t n, int64_t base, int64_t *dst) {
    int64_t run = base;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const int64_t a = c8[i], b = c8[i + 1], c = c8[i + 2], d = c8[i + 3];
        dst[i] = run;
        dst[i + 1] = run + a;
        dst[i + 2] = run + a + b;
        dst[i + 3] = run + a + b + c;
        run += a + b + c + d;
    }
    for (; i < n; i++) {
        dst[i] = run;
        run += c8[i];
    }
}

SCT_CLONES void sct_counts8_to_u32(const uint8_t *c8, int64_t n, uint32_t *dst) {
    for (int64_t i = 0; i < n; i++) dst[i] = c8[i];
}

/* 64-bit offsets relative to a chunk -> the caller's, and the counts they imply */
SCT_CLONES void sct_rebase_offs64(const int64_t *rel, int64_t n, int64_t base, int64_t *dst) {
    for (int64_t i = 0; i < n; i++) dst[i] = base + rel[i];
}

SCT_CLONES void sct_offs64_to_u32(const int64_t *rel, int64_t n, uint32_t *dst) {
    for (int64_t i = 0; i < n; i++) dst[i] = (uint32_t)(rel[i + 1] - rel[i]);
}
