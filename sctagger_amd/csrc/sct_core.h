// sct_core.h — bit-parallel primitives shared by the device kernels and the host-side table builder.
//
// Everything here is plain integer arithmetic written once and compiled twice: by hipcc into the gfx950
// kernels, and by the host compiler for the seed-table construction in sct_set_barcodes (and, in
// tests/host_core_check.cpp, for a CPU check of the very same arithmetic against a brute-force DP).
//
// Encoding (reference scTagger.py:512-516 map_char): A,C,G,T -> 0..3, any other byte -> 0.  Sequences
// are held as two bit-planes: lo bit i = low bit of the code of character i, hi bit i = its high bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SCT_HD __host__ __device__ __forceinline__
#define SCT_HDM __host__ __device__ __forceinline__
#else
#define SCT_HD static inline
#define SCT_HDM inline
#endif

SCT_HD uint32_t sct_code(uint8_t b) { return b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u; }
SCT_HD bool sct_is_acgt(uint8_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }

// ---------------------------------------------------------------------------------------------------
// Myers / Hyyro bit-vector column step, pattern vertical (bit i = pattern row i+1), text streamed.
// Semi-global: free start in the text (horizontal delta 0 enters row 0), the pattern is matched whole.
// After the step `score` is D[m][j], the best distance of the pattern to any substring ending at j.
// Bits above m-1 never influence lower bits (carries and shifts only move upward), so Eq need not be
// masked to m bits.
// ---------------------------------------------------------------------------------------------------
template <typename W>
struct MyersState {
    W Pv, Mv;
    int score;
};

template <typename W>
SCT_HD void myers_init(MyersState<W> &s, int m) {
    s.Pv = ~(W)0;
    s.Mv = 0;
    s.score = m;
}

// lo/hi: pattern bit-planes; c: text code 0..3; m: pattern length (1..bits of W)
template <typename W>
SCT_HD void myers_step(MyersState<W> &s, W lo, W hi, uint32_t c, int m) {
    const W ml = (W)0 - (W)(c & 1u);
    const W mh = (W)0 - (W)(c >> 1);
    const W Eq = ~(lo ^ ml) & ~(hi ^ mh);
    const W Xv = Eq | s.Mv;
    const W Xh = ((((Eq & s.Pv) + s.Pv)) ^ s.Pv) | Eq;
    W Ph = s.Mv | ~(Xh | s.Pv);
    W Mh = s.Pv & Xh;
    s.score += (int)((Ph >> (m - 1)) & 1);
    s.score -= (int)((Mh >> (m - 1)) & 1);
    Ph <<= 1;
    Mh <<= 1;
    s.Pv = Mh | ~(Xv | Ph);
    s.Mv = Ph & Xv;
}

// ---------------------------------------------------------------------------------------------------
// Transposed Wu-Manber automaton for short texts (L <= 31): bit i of R_d = "some substring of the text
// ending at position i matches the pattern prefix consumed so far with <= d edits".  The pattern is the
// streamed dimension (bl steps), so no per-column score tracking is needed: after the last pattern
// character the answer is the smallest d with R_d != 0.
//   tls, ths  text bit-planes shifted left by one (bit i+1 = plane bit of text[i])
//   ve        bits 1..L (positions that have a character before them), va = bits 0..L
// Transitions per pattern character p: match (advance both, text char == p), substitution (advance
// both), deletion of p (advance the pattern only), insertion (advance the text only).
// ---------------------------------------------------------------------------------------------------
template <int E>
SCT_HD int wm_verify(uint32_t tls, uint32_t ths, uint32_t ve, uint32_t va, uint32_t plo, uint32_t phi, int bl) {
    uint32_t R0 = va, R1 = va, R2 = va;
    for (int j = 0; j < bl; j++) {
        const uint32_t ml = 0u - ((plo >> j) & 1u);
        const uint32_t mh = 0u - ((phi >> j) & 1u);
        const uint32_t Em = ~(tls ^ ml) & ~(ths ^ mh) & ve;
        const uint32_t s0 = R0 << 1;
        const uint32_t n0 = s0 & Em;
        if (E >= 1) {
            const uint32_t s1 = R1 << 1;
            const uint32_t n1 = (s1 & Em) | s0 | R0 | (n0 << 1);
            if (E >= 2) {
                const uint32_t s2 = R2 << 1;
                R2 = (s2 & Em) | s1 | R1 | (n1 << 1);
            }
            R1 = n1;
        }
        R0 = n0;
    }
    if (R0 & va) return 0;
    if (E >= 1 && (R1 & va)) return 1;
    if (E >= 2 && (R2 & va)) return 2;
    return E + 1;
}

// ---------------------------------------------------------------------------------------------------
// Seed families of the filter kernel (bl = 16, max_err <= 2).  The barcode is cut into four quarters
// q0..q3 of 4 characters.  With <= 2 edits at least two quarters are copied unedited; attributing an
// insertion to the quarter on its left, two untouched quarters a < b sit in the text 4(b-a)+delta apart,
// delta = insertions - deletions in the quarters between them.  Enumerating the cases:
//   adjacent pairs (0,1) (1,2) (2,3): delta = 0 -> an exact 8-mer at barcode offset 0 / 4 / 8
//   (0,2) and (1,3): exactly one edit in between (two would leave an adjacent untouched pair) -> |delta| <= 1
//   (0,3): one edit in q1 and one in q2 -> |delta| <= 2
// so every barcode within 2 edits of some substring shares one of these six 16-bit keys with the text.
// With <= 1 edit one half is untouched (families H1, H2); with 0 edits H1 alone suffices.
// Keys are built from the bit-planes: low byte from the lo plane, high byte from the hi plane.
// ---------------------------------------------------------------------------------------------------
enum { SCT_F_H1 = 0, SCT_F_H2 = 1, SCT_F_MID = 2, SCT_F_02 = 3, SCT_F_13 = 4, SCT_F_03 = 5, SCT_NFAM = 6 };
#define SCT_KEYS 65536

// W = uint32_t (texts up to 31 characters) or uint64_t (up to 63)
template <typename W>
SCT_HD uint32_t sct_key8(W lo, W hi, int t) {  // 8 consecutive characters starting at t
    return ((uint32_t)(lo >> t) & 0xFFu) | (((uint32_t)(hi >> t) & 0xFFu) << 8);
}
template <typename W>
SCT_HD uint32_t sct_key44(W lo, W hi, int t, int t2) {  // 4 characters at t and 4 at t2
    const uint32_t l = ((uint32_t)(lo >> t) & 0xFu) | (((uint32_t)(lo >> t2) & 0xFu) << 4);
    const uint32_t h = ((uint32_t)(hi >> t) & 0xFu) | (((uint32_t)(hi >> t2) & 0xFu) << 4);
    return l | (h << 8);
}
// key of a 16-character barcode for family f
SCT_HD uint32_t sct_pattern_key(int f, uint32_t plo, uint32_t phi) {
    switch (f) {
    case SCT_F_H1: return sct_key8(plo, phi, 0);
    case SCT_F_H2: return sct_key8(plo, phi, 8);
    case SCT_F_MID: return sct_key8(plo, phi, 4);
    case SCT_F_02: return sct_key44(plo, phi, 0, 8);
    case SCT_F_13: return sct_key44(plo, phi, 4, 12);
    default: return sct_key44(plo, phi, 0, 12);
    }
}

// Seed placements.  A family F fixes which two quarters (A < B) are assumed untouched; a placement is the pair
// of text positions (t, t2) at which those quarters are looked for:
//   adjacent quarters (H1, MID, H2): one exact 8-mer at t (t2 = t + 4)
//   (0,2), (1,3): t2 = t + 8 + d, |d| <= 1;   (0,3): t2 = t + 12 + d, |d| <= 2
// A placement is valid for a text of length L iff the second quarter ends inside it: t2 + 4 <= L.
// Kernels walk the placements wave-uniformly up to the longest length in the wave and test validity per lane;
// the host-side completeness test (tests/host_core_check.cpp) goes through the same enumeration.
template <int F>
struct SctFamily {
    static constexpr int A = F == SCT_F_H1 ? 0 : F == SCT_F_H2 ? 2 : F == SCT_F_MID ? 1 : F == SCT_F_02 ? 0 : F == SCT_F_13 ? 1 : 0;
    static constexpr int B = F == SCT_F_H1 ? 1 : F == SCT_F_H2 ? 3 : F == SCT_F_MID ? 2 : F == SCT_F_02 ? 2 : F == SCT_F_13 ? 3 : 3;
    static constexpr int GAP = 4 * (B - A);               // nominal distance of the two quarters in the text
    static constexpr int SLACK = B - A - 1;               // |d| <= number of quarters in between (each holds one edit)
    static constexpr int MIN_ERR = B - A == 1 ? (F == SCT_F_H1 ? 0 : F == SCT_F_H2 ? 1 : 2) : 2;  // smallest E that needs F
};

SCT_HD bool sct_placement_valid(int t2, int len) { return t2 + 4 <= len; }
template <int F, typename W>
SCT_HD uint32_t sct_placement_key(W lo, W hi, int t, int t2) {
    return SctFamily<F>::GAP == 4 ? sct_key8(lo, hi, t) : sct_key44(lo, hi, t, t2);
}

template <int F, typename Fn>
SCT_HD void sct_for_each_placement(int len, Fn &&fn) {
    for (int d = -SctFamily<F>::SLACK; d <= SctFamily<F>::SLACK; d++)
        for (int t = 0; t + SctFamily<F>::GAP + d + 4 <= len; t++) fn(t, t + SctFamily<F>::GAP + d);
}

// Every (family, key) lookup for one text (planes lo/hi), family by family.
template <int F, typename W, typename Fn>
struct SctSeedExpand {
    W lo, hi;
    Fn &fn;
    SCT_HDM void operator()(int t, int t2) { fn(F, sct_placement_key<F, W>(lo, hi, t, t2)); }
};
template <int E, typename W, typename Fn>
SCT_HD void sct_for_each_seed(W lo, W hi, int len, Fn &&fn) {
    { SctSeedExpand<SCT_F_H1, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_H1>(len, x); }
    if (E >= 1) { SctSeedExpand<SCT_F_H2, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_H2>(len, x); }
    if (E >= 2) {
        { SctSeedExpand<SCT_F_MID, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_MID>(len, x); }
        { SctSeedExpand<SCT_F_02, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_02>(len, x); }
        { SctSeedExpand<SCT_F_13, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_13>(len, x); }
        { SctSeedExpand<SCT_F_03, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_03>(len, x); }
    }
}

// ---------------------------------------------------------------------------------------------------
// Anchored verification.  A candidate found through family F at placement (t, t2) only has to be checked
// under the hypothesis that produced it: quarter A of the barcode sits exactly at text position t and
// quarter B exactly at t2.  (If the barcode matches the text within E edits, some optimal alignment leaves
// two quarters untouched, that placement is enumerated, and the hypothesis holds for it; under any
// hypothesis the value computed below is the cost of a real alignment, hence >= the true distance.  So the
// minimum over all enumerated placements whose key matches is the true distance whenever that is <= E.)
// Only the 8 characters outside the two quarters are run through the Wu-Manber automaton:
//   left   characters before quarter A: free start, must arrive exactly at t
//   middle characters between the quarters: from t+4 exactly to t2
//   right  characters after quarter B: from t2+4, free end
// em(c) returns the match mask of character code c (bit i+1 set iff text[i] == c, bits 1..len);
// code holds the barcode as 2-bit codes, character j at bits 2j..2j+1; va = bits 0..len.
// ---------------------------------------------------------------------------------------------------
template <int E, typename W>
struct WmState {
    W R0, R1, R2;
    SCT_HDM void set(W v) { R0 = R1 = R2 = v; }
    SCT_HDM void step(W Em) {
        const W s0 = R0 << 1;
        const W n0 = s0 & Em;
        if (E >= 1) {
            const W s1 = R1 << 1;
            const W n1 = (s1 & Em) | s0 | R0 | (n0 << 1);
            if (E >= 2) {
                const W s2 = R2 << 1;
                R2 = (s2 & Em) | s1 | R1 | (n1 << 1);
            }
            R1 = n1;
        }
        R0 = n0;
    }
    // keep only the states that sit exactly at position `at`, then skip `skip` exactly matched characters;
    // text characters inserted right after them are accounted for by re-closing under insertion
    SCT_HDM void anchor(int at, int skip) {
        const W m = (W)1 << at;
        R0 = (R0 & m) << skip;
        if (E >= 1) R1 = ((R1 & m) << skip) | (R0 << 1);
        if (E >= 2) R2 = ((R2 & m) << skip) | (R1 << 1);
    }
    SCT_HDM int result(W va) const {
        if (R0 & va) return 0;
        if (E >= 1 && (R1 & va)) return 1;
        if (E >= 2 && (R2 & va)) return 2;
        return E + 1;
    }
};

// bits 0..len of a W (positions that exist in a text of `len` characters)
template <typename W>
SCT_HD W sct_valid_mask(int len) {
    return len >= (int)(8 * sizeof(W)) - 1 ? ~(W)0 : (((W)1 << (len + 1)) - 1);
}

template <int E, int F, typename W, typename EmFetch>
SCT_HD int wm_anchored(EmFetch &&em, W va, uint32_t code, int t, int t2) {
    constexpr int A = SctFamily<F>::A, B = SctFamily<F>::B;
    WmState<E, W> s;
    s.set(A == 0 ? ((W)1 << t) : va);
    for (int j = 0; j < 4 * A; j++) s.step(em((code >> (2 * j)) & 3u));
    if (B == A + 1) {
        s.anchor(t, 8);
    } else {
        s.anchor(t, 4);
        for (int j = 4 * (A + 1); j < 4 * B; j++) s.step(em((code >> (2 * j)) & 3u));
        s.anchor(t2, 4);
    }
    for (int j = 4 * (B + 1); j < 16; j++) s.step(em((code >> (2 * j)) & 3u));
    return s.result(va);
}

// ---------------------------------------------------------------------------------------------------
// Whitelist dictionary of the extract_sr_bc_from_lr stage (reference scTagger.py:801-809): keys are whole
// barcodes of 1..16 raw bytes, zero-padded into four little-endian words.  One open-addressing table holds
// all keys (the length is part of the key); host build and device lookup share hash and slot layout.
// ---------------------------------------------------------------------------------------------------
struct SctWlSlot {
    uint32_t k[4];
    int32_t value;
    uint32_t len;  // 0 = empty slot
};

SCT_HD uint32_t sct_wl_hash(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len) {
    uint32_t h = len * 0x9E3779B1u;
    h = (h ^ k0) * 0x85EBCA6Bu;
    h = ((h << 13) | (h >> 19)) ^ k1;
    h *= 0xC2B2AE35u;
    h = ((h << 17) | (h >> 15)) ^ k2;
    h *= 0x27D4EB2Fu;
    h = ((h << 11) | (h >> 21)) ^ k3;
    h *= 0x165667B1u;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    return h;
}
