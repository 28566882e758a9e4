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

#include <type_traits>

#if defined(__HIPCC__)
#define SCT_HD __host__ __device__ __forceinline__
#define SCT_HDM __host__ __device__ __forceinline__
#else
#define SCT_HD static inline
#define SCT_HDM inline
#endif

SCT_HD uint32_t sct_code(uint8_t b) { return b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u; }
SCT_HD bool sct_is_acgt(uint8_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }

// Does the reference ever read byte j of a row (row = the row's `len` bytes)?  It reads a byte when it inserts a window that
// contains it (scTagger.py:677-682: sub = segment[i : i + bl + e], inserted iff it starts with the task's prefix — over all
// 4^k tasks: iff its first k raw bytes are all in ACGT — and holds at least bl - e characters), decodes and maps every byte of
// such a window (:682, :555) and raises on one >= 128.  A byte no inserted window covers is never looked at.
SCT_HD bool sct_byte_is_read(const uint8_t *row, int64_t len, int64_t j, int bl, int max_err, int k) {
    const int64_t last = len - (bl - max_err);          // last window start that is long enough
    int64_t i = j - (bl + max_err) + 1;
    if (i < 0) i = 0;
    for (; i <= j && i <= last; i++) {
        bool ok = true;
        for (int q = 0; q < k; q++) ok = ok && sct_is_acgt(row[i + q]);   // (i + k <= i + bl - e <= len)
        if (ok) return true;
    }
    return false;
}
SCT_HD uint32_t sct_low_bits(int k) { return k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u); }

// ---------------------------------------------------------------------------------------------------
// Four ASCII bytes at once (one aligned 32-bit word of a segment) -> three nibbles, bit j = byte j of the word:
//   lo4, hi4  the two bits of map_char's code (scTagger.py:512-516), 0 where the byte is not one of A C G T
//   bad4      1 where the byte is not exactly 'A', 'C', 'G' or 'T'
// A = 0x41, C = 0x43, G = 0x47, T = 0x54: bit 2 of the byte is the high code bit, bit 1 ^ bit 2 the low one.  With
// d = byte ^ 0x41 the byte is valid iff bits 7, 6, 5, 3 of d are clear and bits 4 and 0 both equal (lo & hi) — only T
// (d = 0x15) has them set; bits 1 and 2 carry the code.  The bits of interest sit 8 apart; a multiplication whose
// partial products all land on distinct bit positions moves them next to each other.
// ---------------------------------------------------------------------------------------------------
SCT_HD void sct_decode4(uint32_t w, uint32_t &lo4, uint32_t &hi4, uint32_t &bad4) {
    const uint32_t x = w ^ (w >> 1);                                  // bit 1 of each byte: low code bit
    const uint32_t d = w ^ 0x41414141u;
    const uint32_t t = (x >> 1) & (w >> 2) & 0x01010101u;              // both code bits set
    const uint32_t bad = (d & 0xF9F9F9F9u) ^ (t | (t << 4));           // zero byte <=> valid byte
    const uint32_t nz = (((bad & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | bad) & 0x80808080u;
    bad4 = (nz * 0x00204081u) >> 28;
    lo4 = (((x & 0x02020202u) * 0x00810204u) >> 24) & ~bad4;
    hi4 = (((w & 0x04040404u) * 0x00408102u) >> 24) & ~bad4;
}

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

// The same step with the match mask Eq supplied by the caller (bit i = pattern[i] == text character), as used
// for arbitrary byte alphabets.  GLOBAL_START = true charges the characters before the match (horizontal delta +1
// enters row 0): the pattern must then start at the first column ("SHW" / prefix mode).
template <typename W, bool GLOBAL_START>
SCT_HD void myers_step_eq(MyersState<W> &s, W Eq, int m) {
    const W Xv = Eq | s.Mv;
    const W Xh = ((((Eq & s.Pv) + s.Pv)) ^ s.Pv) | Eq;
    W Ph = s.Mv | ~(Xh | s.Pv);
    W Mh = s.Pv & Xh;
    s.score += (int)((Ph >> (m - 1)) & 1);
    s.score -= (int)((Mh >> (m - 1)) & 1);
    Ph = (Ph << 1) | (GLOBAL_START ? (W)1 : (W)0);
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
//
// For very large barcode sets (the 3.6 M-entry whitelist scale) 16-bit keys leave ~100 barcodes per bucket; when
// max_err <= 1 a second scheme cuts the barcode into THIRDS of 5, 5 and 6 characters: with <= 1 edit two thirds are
// untouched, giving the families T01 (10-mer at offset 0), T12 (11-mer at offset 5) and T02 (5-mer + 6-mer,
// |delta| <= 1) with 20- and 22-bit keys.
// ---------------------------------------------------------------------------------------------------
enum {
    SCT_F_H1 = 0, SCT_F_H2 = 1, SCT_F_MID = 2, SCT_F_02 = 3, SCT_F_13 = 4, SCT_F_03 = 5, SCT_NFAM = 6,  // quarters
    SCT_F_T01 = 6, SCT_F_T12 = 7, SCT_F_T02 = 8, SCT_NFAM_ALL = 9                                       // thirds
};
#define SCT_KEYS 65536

// A family F fixes which two pieces (A < B) of the barcode are assumed untouched.
//   PA, LA / PB, LB  start and length of the two pieces in the barcode
//   GAP              nominal distance of the two piece starts in the text, SLACK = pieces in between: the second
//                    piece is looked for at t2 = t + GAP + d, |d| <= SLACK (each piece in between holds one edit)
//   KEYCH            characters in the key (2 bits each)
template <int F>
struct SctFamily {
    static constexpr bool THIRDS = F >= SCT_F_T01;
    static constexpr int A = F == SCT_F_H1 ? 0 : F == SCT_F_H2 ? 2 : F == SCT_F_MID ? 1 : F == SCT_F_02 ? 0 : F == SCT_F_13 ? 1
                             : F == SCT_F_03 ? 0 : F == SCT_F_T01 ? 0 : F == SCT_F_T12 ? 1 : 0;
    static constexpr int B = F == SCT_F_H1 ? 1 : F == SCT_F_H2 ? 3 : F == SCT_F_MID ? 2 : F == SCT_F_02 ? 2 : F == SCT_F_13 ? 3
                             : F == SCT_F_03 ? 3 : F == SCT_F_T01 ? 1 : F == SCT_F_T12 ? 2 : 2;
    static constexpr int piece_start(int i) { return THIRDS ? (i == 0 ? 0 : i == 1 ? 5 : i == 2 ? 10 : 16) : 4 * i; }
    static constexpr int PA = piece_start(A), LA = piece_start(A + 1) - piece_start(A);
    static constexpr int PB = piece_start(B), LB = piece_start(B + 1) - piece_start(B);
    static constexpr int GAP = PB - PA;
    static constexpr int SLACK = B - A - 1;
    static constexpr bool ADJ = B == A + 1;
    static constexpr int KEYCH = LA + LB;
    static constexpr uint32_t NKEYS = 1u << (2 * KEYCH);
    static constexpr int INDEX = F;
    static constexpr int BL = 16;
};

// The same description with run-time piece boundaries, for barcode lengths 8..16 other than 16: the barcode is cut
// into four pieces at floor(i * bl / 4) (sizes differ by at most one); the completeness argument does not depend on
// the piece sizes.  Only the quarter families exist at run time.  Keys have at most 8 characters (16 bits).
struct SctFamilyRT {
    int A, B, PA, LA, PB, LB, GAP, SLACK, KEYCH, INDEX, BL;
    bool ADJ;
};
SCT_HD SctFamilyRT sct_family_rt(int f, int bl) {
    SctFamilyRT r;
    r.A = f == SCT_F_H1 ? 0 : f == SCT_F_H2 ? 2 : f == SCT_F_MID ? 1 : f == SCT_F_02 ? 0 : f == SCT_F_13 ? 1 : 0;
    r.B = f == SCT_F_H1 ? 1 : f == SCT_F_H2 ? 3 : f == SCT_F_MID ? 2 : f == SCT_F_02 ? 2 : f == SCT_F_13 ? 3 : 3;
    r.PA = r.A * bl / 4;
    r.LA = (r.A + 1) * bl / 4 - r.PA;
    r.PB = r.B * bl / 4;
    r.LB = (r.B + 1) * bl / 4 - r.PB;
    r.GAP = r.PB - r.PA;
    r.SLACK = r.B - r.A - 1;
    r.ADJ = r.B == r.A + 1;
    r.KEYCH = r.LA + r.LB;
    r.INDEX = f;
    r.BL = bl;
    return r;
}

// generic forms over a family description `fam` (SctFamily<F>{} folds to constants, SctFamilyRT does not)
template <typename W, typename Fam>
SCT_HD uint32_t sct_key_of(const Fam &fam, W lo, W hi, int t, int t2) {
    const int LA = fam.LA, LB = fam.LB, K = fam.KEYCH;
    if (fam.ADJ) {
        return ((uint32_t)(lo >> t) & ((1u << K) - 1u)) | (((uint32_t)(hi >> t) & ((1u << K) - 1u)) << K);
    } else {
        const uint32_t l = ((uint32_t)(lo >> t) & ((1u << LA) - 1u)) | (((uint32_t)(lo >> t2) & ((1u << LB) - 1u)) << LA);
        const uint32_t h = ((uint32_t)(hi >> t) & ((1u << LA) - 1u)) | (((uint32_t)(hi >> t2) & ((1u << LB) - 1u)) << LA);
        return l | (h << K);
    }
}
template <typename Fam>
SCT_HD bool sct_valid_of(const Fam &fam, int t2, int len) { return t2 + fam.LB <= len; }
// Which placements are worth a lookup.  Piece A at t and piece B at t2 = t + GAP + d pin the barcode in the text: the PA
// characters before piece A must end at t, so at least (PA - t)+ of them are deleted; the stretch between the pieces absorbs
// |d| edits; of the n2 = BL - PB - LB characters behind piece B at least (n2 - (len - t2 - LB))+ are deleted.  A placement
// whose three lower bounds add up to more than E cannot verify within E edits (the anchored automaton computes the cost of an
// alignment with exactly these anchors) and is skipped: for 22-character texts that is 32 of the 146 placements of the six
// quarter families at E = 2, 6 of the 15 of H1.  sct_placement_need = the shortest text for which the placement can still verify.
static constexpr int SCT_NEED_NEVER = 1 << 20;
static constexpr int SCT_E_UNPRUNED = 99;   // "every placement whose pieces lie in the text" (tests of the automaton itself)
template <typename Fam>
SCT_HD int sct_placement_need(const Fam &fam, int E, int t, int t2) {
    const int d = t2 - t - fam.GAP, ad = d < 0 ? -d : d;
    const int left = fam.PA > t ? fam.PA - t : 0;
    const int rb = E - ad - left;                 // edits left for the characters behind piece B
    if (rb < 0) return SCT_NEED_NEVER;
    const int n2 = fam.BL - fam.PB - fam.LB;
    return t2 + fam.LB + (n2 > rb ? n2 - rb : 0);
}
template <typename Fam, typename Fn>
SCT_HD void sct_for_each_placement_of(const Fam &fam, int len, Fn &&fn, int E = SCT_E_UNPRUNED) {
    for (int d = -fam.SLACK; d <= fam.SLACK; d++)
        for (int t = 0; t + fam.GAP + d + fam.LB <= len; t++)
            if (sct_placement_need(fam, E, t, t + fam.GAP + d) <= len) fn(t, t + fam.GAP + d);
}

// key of the text (planes lo/hi, W = uint32_t up to 31 characters, uint64_t up to 63) for pieces at t and t2:
// the low KEYCH bits come from the lo plane, the next KEYCH bits from the hi plane
template <int F, typename W>
SCT_HD uint32_t sct_placement_key(W lo, W hi, int t, int t2) {
    constexpr int LA = SctFamily<F>::LA, LB = SctFamily<F>::LB, K = SctFamily<F>::KEYCH;
    if (SctFamily<F>::ADJ) {
        return ((uint32_t)(lo >> t) & ((1u << K) - 1u)) | (((uint32_t)(hi >> t) & ((1u << K) - 1u)) << K);
    } else {
        const uint32_t l = ((uint32_t)(lo >> t) & ((1u << LA) - 1u)) | (((uint32_t)(lo >> t2) & ((1u << LB) - 1u)) << LA);
        const uint32_t h = ((uint32_t)(hi >> t) & ((1u << LA) - 1u)) | (((uint32_t)(hi >> t2) & ((1u << LB) - 1u)) << LA);
        return l | (h << K);
    }
}
template <int F>
SCT_HD uint32_t sct_pattern_key_f(uint32_t plo, uint32_t phi) {
    return sct_placement_key<F, uint32_t>(plo, phi, SctFamily<F>::PA, SctFamily<F>::PB);
}
// key of a 16-character barcode for family f (run-time f)
SCT_HD uint32_t sct_pattern_key(int f, uint32_t plo, uint32_t phi) {
    switch (f) {
    case SCT_F_H1: return sct_pattern_key_f<SCT_F_H1>(plo, phi);
    case SCT_F_H2: return sct_pattern_key_f<SCT_F_H2>(plo, phi);
    case SCT_F_MID: return sct_pattern_key_f<SCT_F_MID>(plo, phi);
    case SCT_F_02: return sct_pattern_key_f<SCT_F_02>(plo, phi);
    case SCT_F_13: return sct_pattern_key_f<SCT_F_13>(plo, phi);
    case SCT_F_03: return sct_pattern_key_f<SCT_F_03>(plo, phi);
    case SCT_F_T01: return sct_pattern_key_f<SCT_F_T01>(plo, phi);
    case SCT_F_T12: return sct_pattern_key_f<SCT_F_T12>(plo, phi);
    default: return sct_pattern_key_f<SCT_F_T02>(plo, phi);
    }
}
SCT_HD uint32_t sct_family_nkeys(int f) {
    if (f >= SCT_NFAM_ALL) return 1u << 20;  // the wide families (SCT_WIDE_KEYS)
    switch (f) {
    case SCT_F_T01: return SctFamily<SCT_F_T01>::NKEYS;
    case SCT_F_T12: return SctFamily<SCT_F_T12>::NKEYS;
    case SCT_F_T02: return SctFamily<SCT_F_T02>::NKEYS;
    default: return SCT_KEYS;
    }
}

// ---------------------------------------------------------------------------------------------------
// The WIDE scheme (max_err = 2, barcode sets from a few hundred thousand up): 16-bit keys leave np / 32 768 barcodes in every
// bucket (110 at the 3.6 M-entry whitelist scale), and every one of them is run through the automaton.  Each quarter pair is
// therefore split by a THIRD untouched piece, one half (2 characters) of another quarter: 20-bit keys, 16 times shorter buckets.
// Which half can be relied on follows from what the pair is responsible for (the distributions of <= 2 edits over the quarters
// that no earlier pair covers, as above; an insertion counts for the piece of the character behind it, so untouched
// neighbours are contiguous in the text):
//   (0,1), everything with q0 = q1 = 0:   half 2L untouched: Q0 Q1 2L contiguous                                         W0
//                                         else 2R untouched and ONE edit in 2L: 2R at t + 10 + s, |s| <= 1                W1
//                                         else both edits in q2, q3 untouched: 3L at t + 12 + s, |s| <= 2                 W2
//   (2,3), everything with q2 = q3 = 0:   mirror images: 1R (W3), 1L |s| <= 1 (W4), 0R |s| <= 2 (W5)
//   (1,2), one edit in q0 and one in q3:  0R untouched: 0R Q1 Q2 contiguous (W6); else 3L untouched: Q1 Q2 3L (W7);
//                                         else the edits sit in 0R and 3L: 0L at t - 4 + s, |s| <= 1 (W8)
//   (0,2), one edit in q1 and one in q3:  1L untouched (behind Q0, W9) or 1R untouched (before Q2, W10); |d| <= 1 as before
//   (1,3), one edit in q0 and one in q2:  2L behind Q1 (W11) or 2R before Q3 (W12)
//   (0,3), one edit in q1 and one in q2:  1L behind Q0 (W13); else 2R before Q3 (W14), |d| <= 2 both; else the edits sit in 1L
//                                         and 2R: 1R at t + 6 + s, t2 = t + 12 + s + x, |s|, |x| <= 1 (W15)
// After W0 every barcode at distance 0 has been seen, after W0 W1 W3 W4 every one at distance <= 1 (a single edit in q2 spoils
// one half of it and leaves the other; one in q3 leaves 2L; likewise on the other side).  The candidate is verified exactly as
// in the quarter scheme, anchored on the two quarters (the half is part of the verified rest): same distances, same locators.
// tests/host_core_check.cpp checks completeness and the tiers against brute force, exhaustively around a pattern.
// ---------------------------------------------------------------------------------------------------
#define SCT_NWIDE 16
#define SCT_WIDE_KEYS (1u << 20)
#define SCT_F_WIDE0 SCT_NFAM_ALL               // table index of W0
#define SCT_NFAM_TABLES (SCT_NFAM_ALL + SCT_NWIDE)
template <int WF>
struct SctWide {
    static constexpr int BASE = WF < 3 ? SCT_F_H1 : WF < 6 ? SCT_F_H2 : WF < 9 ? SCT_F_MID : WF < 11 ? SCT_F_02 : WF < 13 ? SCT_F_13 : SCT_F_03;
    using Base = SctFamily<BASE>;
    // the half: where it starts in the barcode, which quarter's text position it is measured from, nominal distance from there
    static constexpr int PC = WF == 0 ? 8 : WF == 1 ? 10 : WF == 2 ? 12 : WF == 3 ? 6 : WF == 4 ? 4 : WF == 5 ? 2 : WF == 6 ? 2 : WF == 7 ? 12
                              : WF == 8 ? 0 : WF == 9 ? 4 : WF == 10 ? 6 : WF == 11 ? 8 : WF == 12 ? 10 : WF == 13 ? 4 : WF == 14 ? 10 : 6;
    static constexpr bool FROM_B = WF == 10 || WF == 12 || WF == 14;
    static constexpr int OFFC = PC - (FROM_B ? Base::PB : Base::PA);
    // t2 = t + GAP + d, u = (t or t2) + OFFC + s;  COUPLED: d = s + x with x in [DMIN, DMAX]
    static constexpr int SMAX = (WF == 1 || WF == 4 || WF == 8 || WF == 15) ? 1 : (WF == 2 || WF == 5) ? 2 : 0;
    static constexpr int DMAX = (WF >= 9 && WF <= 12) ? 1 : (WF == 13 || WF == 14) ? 2 : WF == 15 ? 1 : 0;
    static constexpr bool COUPLED = WF == 15;
    static constexpr int INDEX = SCT_F_WIDE0 + WF;
    static constexpr uint32_t NKEYS = SCT_WIDE_KEYS;
    // where the half sits among the 8 characters outside the quarter pair (the pair's `rest`, in barcode order)
    static constexpr int POS = PC - (Base::PA < PC ? 4 : 0) - (Base::PB < PC ? 4 : 0);
};
// the pair's 16-bit rest without / with the half's two characters (the half is part of the key: a fixed bucket keeps the other six)
SCT_HD uint32_t sct_wide_rest12(uint32_t r16, int pos) { return (r16 & sct_low_bits(2 * pos)) | ((r16 >> (2 * pos + 4)) << (2 * pos)); }
SCT_HD uint32_t sct_wide_rest16(uint32_t r12, uint32_t half4, int pos) {
    return (r12 & sct_low_bits(2 * pos)) | (half4 << (2 * pos)) | ((r12 >> (2 * pos)) << (2 * pos + 4));
}
SCT_HD int sct_wide_pos(int wf) {
    switch (wf) {
    case 0: return SctWide<0>::POS; case 1: return SctWide<1>::POS; case 2: return SctWide<2>::POS; case 3: return SctWide<3>::POS;
    case 4: return SctWide<4>::POS; case 5: return SctWide<5>::POS; case 6: return SctWide<6>::POS; case 7: return SctWide<7>::POS;
    case 8: return SctWide<8>::POS; case 9: return SctWide<9>::POS; case 10: return SctWide<10>::POS; case 11: return SctWide<11>::POS;
    case 12: return SctWide<12>::POS; case 13: return SctWide<13>::POS; case 14: return SctWide<14>::POS; default: return SctWide<15>::POS;
    }
}
SCT_HD int sct_wide_base(int wf) { return wf < 3 ? SCT_F_H1 : wf < 6 ? SCT_F_H2 : wf < 9 ? SCT_F_MID : wf < 11 ? SCT_F_02 : wf < 13 ? SCT_F_13 : SCT_F_03; }
#define SCT_WIDE_FB_CAP 16
// key: the pair's 16 bits, then the half's two lo-plane bits and its two hi-plane bits
template <int WF, typename W>
SCT_HD uint32_t sct_wide_key(W lo, W hi, int t, int t2, int u) {
    return sct_placement_key<SctWide<WF>::BASE, W>(lo, hi, t, t2) | (((uint32_t)(lo >> u) & 3u) << 16) | (((uint32_t)(hi >> u) & 3u) << 18);
}
template <int WF>
SCT_HD uint32_t sct_wide_pattern_key_f(uint32_t plo, uint32_t phi) {
    return sct_wide_key<WF, uint32_t>(plo, phi, SctWide<WF>::Base::PA, SctWide<WF>::Base::PB, SctWide<WF>::PC);
}
SCT_HD uint32_t sct_wide_pattern_key(int wf, uint32_t plo, uint32_t phi) {
    switch (wf) {
    case 0: return sct_wide_pattern_key_f<0>(plo, phi);
    case 1: return sct_wide_pattern_key_f<1>(plo, phi);
    case 2: return sct_wide_pattern_key_f<2>(plo, phi);
    case 3: return sct_wide_pattern_key_f<3>(plo, phi);
    case 4: return sct_wide_pattern_key_f<4>(plo, phi);
    case 5: return sct_wide_pattern_key_f<5>(plo, phi);
    case 6: return sct_wide_pattern_key_f<6>(plo, phi);
    case 7: return sct_wide_pattern_key_f<7>(plo, phi);
    case 8: return sct_wide_pattern_key_f<8>(plo, phi);
    case 9: return sct_wide_pattern_key_f<9>(plo, phi);
    case 10: return sct_wide_pattern_key_f<10>(plo, phi);
    case 11: return sct_wide_pattern_key_f<11>(plo, phi);
    case 12: return sct_wide_pattern_key_f<12>(plo, phi);
    case 13: return sct_wide_pattern_key_f<13>(plo, phi);
    case 14: return sct_wide_pattern_key_f<14>(plo, phi);
    default: return sct_wide_pattern_key_f<15>(plo, phi);
    }
}
// Placements (t, t2, u) of a wide family for a text of `len` characters: all three pieces inside the text.  Kernels build the
// same list for the longest text of the wave (fn is called combination by combination, t ascending) and test each lane's length.
template <int WF, typename Fn>
SCT_HD void sct_for_each_wide_placement(int len, Fn &&fn) {
    using F = SctWide<WF>;
    for (int s = -F::SMAX; s <= F::SMAX; s++)
        for (int x = -F::DMAX; x <= F::DMAX; x++) {
            const int d = F::COUPLED ? s + x : x;
            const int t2o = F::Base::GAP + d, uo = (F::FROM_B ? t2o : 0) + F::OFFC + s;
            const int t0 = uo < 0 ? -uo : 0;
            const int ext = t2o + F::Base::LB > uo + 2 ? t2o + F::Base::LB : uo + 2;
            for (int t = t0; t + ext <= len; t++) fn(t, t + t2o, t + uo);
        }
}
SCT_HD bool sct_wide_valid(int t2, int u, int len) { return t2 + 4 <= len && u + 2 <= len; }

// Fixed buckets of the thirds families (sct_tables.h, k_filter.hip.h run_family_fb): 16-bit entries per key, by key length
#define SCT_FB_CAP(keych) ((keych) == 10 ? 16 : 8)
#define SCT_FB_VALID 0x8000u     // entry: rest characters (at most 6 = 12 bits) | SCT_FB_VALID
#define SCT_FB_OVERFLOW 0x4000u  // first entry of a key with more entries than its bucket holds: use the offsets

// Seed placements: the pair of text positions (t, t2) at which the two pieces are looked for.  A placement is
// valid for a text of length L iff the second piece ends inside it: t2 + LB <= L.  Kernels walk the placements
// wave-uniformly up to the longest length in the wave and test validity per lane; the host-side completeness
// test (tests/host_core_check.cpp) goes through the same enumeration.
template <int F>
SCT_HD bool sct_placement_valid(int t2, int len) { return t2 + SctFamily<F>::LB <= len; }

template <int F, typename Fn>
SCT_HD void sct_for_each_placement(int len, Fn &&fn, int E = SCT_E_UNPRUNED) {
    sct_for_each_placement_of(SctFamily<F>{}, len, fn, E);
}

// Every (family, key) lookup for one text (planes lo/hi), family by family.
template <int F, typename W, typename Fn>
struct SctSeedExpand {
    W lo, hi;
    Fn &fn;
    SCT_HDM void operator()(int t, int t2) { fn(F, sct_placement_key<F, W>(lo, hi, t, t2)); }
};
// thirds scheme (max_err <= 1)
template <int E, typename W, typename Fn>
SCT_HD void sct_for_each_seed_thirds(W lo, W hi, int len, Fn &&fn) {
    { SctSeedExpand<SCT_F_T01, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_T01>(len, x, E); }
    if (E >= 1) {
        { SctSeedExpand<SCT_F_T12, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_T12>(len, x, E); }
        { SctSeedExpand<SCT_F_T02, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_T02>(len, x, E); }
    }
}
// quarters scheme (max_err <= 2)
template <int E, typename W, typename Fn>
SCT_HD void sct_for_each_seed(W lo, W hi, int len, Fn &&fn) {
    { SctSeedExpand<SCT_F_H1, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_H1>(len, x, E); }
    if (E >= 1) { SctSeedExpand<SCT_F_H2, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_H2>(len, x, E); }
    if (E >= 2) {
        { SctSeedExpand<SCT_F_MID, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_MID>(len, x, E); }
        { SctSeedExpand<SCT_F_02, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_02>(len, x, E); }
        { SctSeedExpand<SCT_F_13, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_13>(len, x, E); }
        { SctSeedExpand<SCT_F_03, W, Fn> x{lo, hi, fn}; sct_for_each_placement<SCT_F_03>(len, x, E); }
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

// The characters of a barcode OUTSIDE the two pieces of `fam`, in barcode order, 2 bits each from bit 0 (at most 8
// characters for the quarter families of any length 8..16): all the automaton below needs of a candidate, since the
// piece characters are the key it was found under.
template <typename Fam>
SCT_HD uint32_t sct_rest_of(const Fam &fam, uint32_t code) {
    const int n0 = fam.PA, s1 = fam.PA + fam.LA, n1 = fam.PB - s1, s2 = fam.PB + fam.LB, n2 = fam.BL - s2;
    uint32_t r = code & sct_low_bits(2 * n0);
    if (n1 > 0) r |= ((code >> (2 * s1)) & sct_low_bits(2 * n1)) << (2 * n0);
    if (n2 > 0) r |= ((code >> (2 * s2)) & sct_low_bits(2 * n2)) << (2 * (n0 + n1));
    return r;
}

template <int E, typename W, typename Fam, typename EmFetch>
SCT_HD int wm_anchored_rest(const Fam &fam, EmFetch &&em, W va, uint32_t rest, int t, int t2) {
    WmState<E, W> s;
    s.set(fam.PA == 0 ? ((W)1 << t) : va);
    for (int j = 0; j < fam.PA; j++, rest >>= 2) s.step(em(rest & 3u));
    if (fam.ADJ) {
        s.anchor(t, fam.LA + fam.LB);
    } else {
        s.anchor(t, fam.LA);
        for (int j = fam.PA + fam.LA; j < fam.PB; j++, rest >>= 2) s.step(em(rest & 3u));
        s.anchor(t2, fam.LB);
    }
    for (int j = fam.PB + fam.LB; j < fam.BL; j++, rest >>= 2) s.step(em(rest & 3u));
    return s.result(va);
}
template <int E, typename W, typename Fam, typename EmFetch>
SCT_HD int wm_anchored_of(const Fam &fam, EmFetch &&em, W va, uint32_t code, int t, int t2) {
    return wm_anchored_rest<E, W>(fam, em, va, sct_rest_of(fam, code), t, t2);
}
template <int E, int F, typename W, typename EmFetch>
SCT_HD int wm_anchored(EmFetch &&em, W va, uint32_t code, int t, int t2) {
    return wm_anchored_of<E, W>(SctFamily<F>{}, em, va, code, t, t2);
}

// ---------------------------------------------------------------------------------------------------
// Restricted starts.  With -t > 1 the reference only inserts the windows whose first k raw bytes are all in ACGT (k = the
// prefix length of its tasks, scTagger.py:681, :689-697), and a query walks a window from its FIRST character: an alignment
// must begin at an allowed window start s, and text characters skipped in front of the barcode are insertions that cost an edit
// each (Trie.dfs :587-588 at depth 0).  In the automaton that is a different INITIAL state and nothing else:
//   A0 = the allowed starts (bit s), A1 = A0 | A0 << 1, A2 = A1 | A1 << 1 (within the text)
// A_d = positions an alignment can stand at, no barcode character consumed yet, having spent at most d edits — instead of
// "anywhere, for free" (va).  The seed families stay complete: the leading insertions touch no quarter, so an alignment within E
// edits still leaves two quarters (one half, two thirds) untouched at exactly their text positions, the placement is enumerated
// as before, and the value computed under its hypothesis is the cost of a real alignment from an allowed start.  The same masks
// serve rows cut into overlapping windows of PLANE_MAX characters (k_filter's special launch): a window owns the starts in its
// first `step` characters.
// ---------------------------------------------------------------------------------------------------
// allowed starts of a text of `len` characters from its "byte is not one of ACGT" mask: s is allowed iff s + k <= len and none of
// the k bytes from s on is bad (k = 0: every position 0..len)
template <typename W>
SCT_HD W sct_allowed_starts(W bad, int len, int k) {
    W blocked = 0;
    for (int q = 0; q < k; q++) blocked |= bad >> q;
    const int ns = len - k + 1;
    const W in = ns <= 0 ? (W)0 : (ns >= (int)(8 * sizeof(W)) ? ~(W)0 : (((W)1 << ns) - 1));
    return ~blocked & in;
}
template <typename W>
struct SctStarts {
    W a[3];
    SCT_HDM void from_allowed(W a0, W va) {
        a[0] = a0 & va;
        a[1] = (a[0] | (a[0] << 1)) & va;
        a[2] = (a[1] | (a[1] << 1)) & va;
    }
};
template <int E, typename W, typename Fam, typename EmFetch>
SCT_HD int wm_anchored_rest_rs(const Fam &fam, EmFetch &&em, W va, const SctStarts<W> &st, uint32_t rest, int t, int t2) {
    WmState<E, W> s;
    s.R0 = st.a[0];   // PA == 0: the anchor keeps bit t of each — "an allowed start at most d characters before t"
    s.R1 = st.a[1];
    s.R2 = st.a[2];
    for (int j = 0; j < fam.PA; j++, rest >>= 2) s.step(em(rest & 3u));
    if (fam.ADJ) {
        s.anchor(t, fam.LA + fam.LB);
    } else {
        s.anchor(t, fam.LA);
        for (int j = fam.PA + fam.LA; j < fam.PB; j++, rest >>= 2) s.step(em(rest & 3u));
        s.anchor(t2, fam.LB);
    }
    for (int j = fam.PB + fam.LB; j < fam.BL; j++, rest >>= 2) s.step(em(rest & 3u));
    return s.result(va);
}

// ---------------------------------------------------------------------------------------------------
// The same anchored verification for TWO candidates at once, one in each 16-bit half of the state words (X in
// bits 0..15, Y in bits 16..31).  A run of the automaton only ever needs a window of the text: the left run must
// arrive exactly at t, so only the positions t - PA - E .. t matter; after an anchor the states sit at the first
// position behind the piece and move up by at most (run length + E) <= 10.  Each run therefore works on the 16
// positions from its own base (per candidate), the match masks are shifted down by that base, and the anchor
// re-bases the states.  Bitwise operations act on both halves at once; only the shifts are per half.
// Match masks of 32 bits (texts of at most 31 characters) or 64 bits; same result as wm_anchored_rest for each candidate.
// ---------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned short sct_us2 __attribute__((ext_vector_type(2)));
SCT_HD uint32_t sct_pk_shl1(uint32_t x) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(sct_us2, x) << (sct_us2)(1)); }
SCT_HD uint32_t sct_pk_shr(uint32_t x, uint32_t sh) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(sct_us2, x) >> __builtin_bit_cast(sct_us2, sh));
}
#else
SCT_HD uint32_t sct_pk_shl1(uint32_t x) { return (x << 1) & 0xFFFEFFFEu; }
SCT_HD uint32_t sct_pk_shr(uint32_t x, uint32_t sh) { return ((x & 0xFFFFu) >> (sh & 0xFFFFu)) | (((x >> 16) >> (sh >> 16)) << 16); }
#endif
SCT_HD uint32_t sct_pk(uint32_t lo, uint32_t hi) { return (lo & 0xFFFFu) | (hi << 16); }
// (a & b) | c and a | b | c.  On gfx950 the compiler selects v_and_or_b32 / v_or3_b32 for these, which issue at half rate
// (4.2 cycles per wave64 instruction per SIMD); v_bitop3_b32 computes any three-input boolean function at FULL rate (2.3 cycles;
// tools/valu_peak.hip, profiles/r03_valu_issue_rates.md).  Truth table: bit (a << 2 | b << 1 | c) of the immediate, i.e. the
// function applied to a = 0xF0, b = 0xCC, c = 0xAA.
#if defined(__HIP_DEVICE_COMPILE__)
SCT_HD uint32_t sct_and_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, (0xF0 & 0xCC) | 0xAA); }
SCT_HD uint32_t sct_or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF0 | 0xCC | 0xAA); }
#else
SCT_HD uint32_t sct_and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
SCT_HD uint32_t sct_or3(uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }
#endif

template <int E>
struct WmPair {
    uint32_t R0, R1, R2;
    SCT_HDM void step(uint32_t Em) {
        const uint32_t s0 = sct_pk_shl1(R0);
        const uint32_t n0 = s0 & Em;
        if (E >= 1) {
            const uint32_t s1 = sct_pk_shl1(R1);
            const uint32_t n1 = sct_or3(sct_and_or(s1, Em, s0), R0, sct_pk_shl1(n0));
            if (E >= 2) {
                const uint32_t s2 = sct_pk_shl1(R2);
                R2 = sct_or3(sct_and_or(s2, Em, s1), R1, sct_pk_shl1(n1));
            }
            R1 = n1;
        }
        R0 = n0;
    }
    // keep the states that sit exactly `at` positions above the window base (per half), and make that position plus
    // the skipped piece the new base; text characters inserted right behind the piece: closure under insertion
    SCT_HDM void anchor(uint32_t at) {
        const uint32_t one = 0x00010001u;
        R0 = sct_pk_shr(R0, at) & one;
        if (E >= 1) R1 = sct_and_or(sct_pk_shr(R1, at), one, R0 << 1);
        if (E >= 2) R2 = sct_and_or(sct_pk_shr(R2, at), one, R1 << 1);
    }
    SCT_HDM uint32_t result(uint32_t m) const {  // d of X | d of Y << 16
        const uint32_t a0 = R0 & m, a1 = E >= 1 ? (R1 & m) : 0u, a2 = E >= 2 ? (R2 & m) : 0u;
        const uint32_t dx = (a0 & 0xFFFFu) ? 0u : (a1 & 0xFFFFu) ? 1u : (a2 & 0xFFFFu) ? 2u : (uint32_t)(E + 1);
        const uint32_t dy = (a0 >> 16) ? 0u : (a1 >> 16) ? 1u : (a2 >> 16) ? 2u : (uint32_t)(E + 1);
        return dx | (dy << 16);
    }
};

// SX / SY: where the alignments may start.  SctFreeStart = anywhere (the usual case); otherwise a functor d -> A_d of the
// candidate's text (restricted starts, above).
struct SctFreeStart {};
template <int E, typename Fam, typename EmX, typename EmY, typename WV, typename SX = SctFreeStart, typename SY = SctFreeStart>
SCT_HD uint32_t wm_anchored_pair(const Fam &fam, EmX &&emx, EmY &&emy, WV vax, WV vay, uint32_t restx, uint32_t resty,
                                 int tx, int t2x, int ty, int t2y, SX &&sx = SctFreeStart{}, SY &&sy = SctFreeStart{}) {
    constexpr bool FREE = std::is_same<typename std::decay<SX>::type, SctFreeStart>::value;
    WmPair<E> s;
    int bx, by;
    if (fam.PA > 0) {  // left run: free start, must arrive exactly at t
        bx = tx - fam.PA - E > 0 ? tx - fam.PA - E : 0;
        by = ty - fam.PA - E > 0 ? ty - fam.PA - E : 0;
        if constexpr (FREE) {
            s.R0 = s.R1 = s.R2 = sct_pk((uint32_t)(vax >> bx), (uint32_t)(vay >> by));
        } else {
            s.R0 = sct_pk((uint32_t)(sx(0) >> bx), (uint32_t)(sy(0) >> by));
            s.R1 = E >= 1 ? sct_pk((uint32_t)(sx(1) >> bx), (uint32_t)(sy(1) >> by)) : 0u;
            s.R2 = E >= 2 ? sct_pk((uint32_t)(sx(2) >> bx), (uint32_t)(sy(2) >> by)) : 0u;
        }
        for (int j = 0; j < fam.PA; j++, restx >>= 2, resty >>= 2)
            s.step(sct_pk((uint32_t)(emx(restx & 3u) >> bx), (uint32_t)(emy(resty & 3u) >> by)));
        s.anchor(sct_pk((uint32_t)(tx - bx), (uint32_t)(ty - by)));
    } else if constexpr (FREE) {
        s.R0 = 0x00010001u;
        s.R1 = 0x00030003u;
        s.R2 = 0x00070007u;
    } else {
        // the barcode begins at t: an allowed start at most d characters before it (bit t of A_d), then the closure under insertion
        s.R0 = sct_pk((uint32_t)(sx(0) >> tx) & 1u, (uint32_t)(sy(0) >> ty) & 1u);
        s.R1 = E >= 1 ? (sct_pk((uint32_t)(sx(1) >> tx) & 1u, (uint32_t)(sy(1) >> ty) & 1u) | (s.R0 << 1)) : 0u;
        s.R2 = E >= 2 ? (sct_pk((uint32_t)(sx(2) >> tx) & 1u, (uint32_t)(sy(2) >> ty) & 1u) | (s.R1 << 1)) : 0u;
    }
    const int skip1 = fam.ADJ ? fam.LA + fam.LB : fam.LA;
    bx = tx + skip1;
    by = ty + skip1;
    if (!fam.ADJ) {  // middle run: from behind piece A exactly to t2
        for (int j = fam.PA + fam.LA; j < fam.PB; j++, restx >>= 2, resty >>= 2)
            s.step(sct_pk((uint32_t)(emx(restx & 3u) >> bx), (uint32_t)(emy(resty & 3u) >> by)));
        s.anchor(sct_pk((uint32_t)(t2x - bx), (uint32_t)(t2y - by)));
        bx = t2x + fam.LB;
        by = t2y + fam.LB;
    }
    for (int j = fam.PB + fam.LB; j < fam.BL; j++, restx >>= 2, resty >>= 2)  // right run: free end
        s.step(sct_pk((uint32_t)(emx(restx & 3u) >> bx), (uint32_t)(emy(resty & 3u) >> by)));
    return s.result(sct_pk((uint32_t)(vax >> bx), (uint32_t)(vay >> by)));
}

// ---------------------------------------------------------------------------------------------------
// Thirds families at max_err = 1 without the automaton (the partitioned join, k_join.hip.h).  A candidate found under family F at
// placement (t, t2) has its two pieces exactly in the text; what is left to compare is ONE short piece of the pattern — its
// "rest" (sct_rest_of: T01 the 6 characters behind the 10-mer, T12 the 5 in front of the 11-mer, T02 the 5 between the two
// pieces) — with the text beside the anchors, within one edit:
//   T01  rest vs the text behind the seed, free end          context = the next <= 7 characters
//   T12  rest vs the text in front of the seed, free start   context = the <= 6 characters before t, REVERSED (so it is a free END)
//   T02  rest vs the text between the pieces, both ends fixed context = the 5 + d characters between them, d = t2 - t - 10
// Characters are 2-bit codes at bits 2j, 2j+1 (tab_code / sct_rest_of order); the T12 context and rest are bit-reversed as a
// whole (v_bfrev: character order AND the two bits of a character swap — on both sides, so equality of characters is kept).
// Within one edit "distance" is: equal -> 0; one substitution, one deleted pattern character or one inserted text character -> 1;
// anything else -> 2 (= E + 1).  With X0 = mismatches rest[j] : ctx[j], X1 = rest[j+1] : ctx[j], X2 = rest[j] : ctx[j+1] (bit 2j
// each) and a = the first mismatch of X0: a deletion exists iff X1 is clear from min(a, n-1) on, an insertion iff X2 is clear from
// min(a, n) on.  Context characters that do not exist (text ends / starts) count as mismatches.
// The same value as wm_anchored<1, F> for every placement whose key matches (tests/host_core_check.cpp).
// ---------------------------------------------------------------------------------------------------
SCT_HD uint32_t sct_spread8(uint32_t v) {   // bit i -> bit 2i, 8 bits
    v = (v | (v << 4)) & 0x0F0Fu;
    v = (v | (v << 2)) & 0x3333u;
    return (v | (v << 1)) & 0x5555u;
}
SCT_HD uint32_t sct_rev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
// The context word of a tuple: bits 0..13 the context characters, bits 14..16 how many of them exist (T01, T12) or d + 1 (T02).
// lo / hi = the text's bit-planes (at most 31 characters), len its length.
SCT_HD uint32_t sct_join_ctx_t01(uint32_t lo, uint32_t hi, int len, int t) {
    const int at = t + 10, avail = len - at < 7 ? (len - at < 0 ? 0 : len - at) : 7;
    return sct_spread8((lo >> at) & 0x7Fu) | (sct_spread8((hi >> at) & 0x7Fu) << 1) | ((uint32_t)avail << 14);
}
SCT_HD uint32_t sct_join_ctx_t12(uint32_t lo, uint32_t hi, int t) {
    // character k of the context = text[t - 1 - k], its two bits swapped (hi plane at bit 2k)
    const int avail = t < 6 ? t : 6;
    const uint32_t rl = t > 0 ? sct_rev32(lo << (32 - t)) : 0u, rh = t > 0 ? sct_rev32(hi << (32 - t)) : 0u;
    return (sct_spread8(rl & 0x3Fu) << 1) | sct_spread8(rh & 0x3Fu) | ((uint32_t)avail << 14);
}
SCT_HD uint32_t sct_join_ctx_t02(uint32_t lo, uint32_t hi, int t, int t2) {
    const int at = t + 5, d = t2 - t - 10;
    return sct_spread8((lo >> at) & 0x3Fu) | (sct_spread8((hi >> at) & 0x3Fu) << 1) | ((uint32_t)(d + 1) << 14);
}
// rest of a T12 entry as the compare wants it: reversed like its context (5 characters, 10 bits)
SCT_HD uint32_t sct_join_rest_t12(uint32_t rest10) { return sct_rev32(rest10) >> 22; }

// distance (0, 1, or 2 = more) of an n-character rest against a context; m = context characters that exist;
// fixed_end: the alignment must consume exactly the m characters (T02), else it may end anywhere (T01, reversed T12)
SCT_HD uint32_t sct_join_distance(uint32_t rest, uint32_t ctx, int n, int m, bool fixed_end) {
    const uint32_t C = ctx & 0x3FFFu;
    const uint32_t exist = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);             // context characters 0 .. m-1
    const uint32_t maskn = (1u << (2 * n)) - 1u, maskn1 = (1u << (2 * (n - 1))) - 1u;
    const uint32_t ev = 0x55555555u;
    uint32_t x = rest ^ C;
    const uint32_t X0 = (((x | (x >> 1)) & ev) | (~exist & ev)) & maskn;                  // rest[j] : ctx[j]
    x = (rest >> 2) ^ C;
    const uint32_t X1 = (((x | (x >> 1)) & ev) | (~exist & ev)) & maskn1;                 // rest[j+1] : ctx[j], j < n - 1
    x = rest ^ (C >> 2);
    const uint32_t X2 = (((x | (x >> 1)) & ev) | (~(exist >> 2) & ev)) & maskn;          // rest[j] : ctx[j+1]
    // a = first mismatch of X0 (in bits: 2 a), n if none
    const uint32_t low = X0 & (0u - X0);                                                 // lowest set bit, 0 if none
    const uint32_t below = low ? low - 1u : 0xFFFFFFFFu;                                  // bits below the first mismatch
    const bool eq = X0 == 0u;
    const bool sub1 = !eq && (X0 & (X0 - 1u)) == 0u;
    // deletion of a pattern character: X1 clear from min(a, n - 1) on  <=>  no X1 bit at or above position min(a, n-1)
    const bool del_ok = (X1 & ~below) == 0u;
    // insertion of a text character: X2 clear from a on
    const bool ins_ok = (X2 & ~below) == 0u;
    if (fixed_end) {
        if (m == n) return eq ? 0u : (sub1 ? 1u : 2u);
        if (m == n + 1) return ins_ok ? 1u : 2u;
        return (m == n - 1 && del_ok) ? 1u : 2u;
    }
    if (m >= n && eq) return 0u;
    if (m >= n && sub1) return 1u;
    if (m >= n - 1 && del_ok) return 1u;
    if (m >= n + 1 && ins_ok) return 1u;
    return 2u;
}

// ---------------------------------------------------------------------------------------------------
// Bucket slots of the quarter families (k_filter.hip.h): one 8-byte word per seed key, read as two 32-bit words
// (a = bits 0..31, b = bits 32..63), that answers most lookups by itself.
//   bit 15 clear  inline bucket: bits 0..1 = n <= 3 entries, bits 2..14 = where its entries start inside their group of 64 keys
//                 (sct_slot_within), bits 16(j+1) .. 16(j+1)+15 = rest characters of entry j (sct_rest_of) — all a
//                 verification needs besides the key
//   bit 15 set    long bucket: bits 16..47 = index of its first entry, bits 0..14 | bits 48..63 << 15 = its length
// The kernel queues entry 0 as `mid`, entry 1 as the funnel shift {top, mid} >> 16 and entry 2 as `top`: the low 16
// bits of each are that entry's rest (the automaton ignores what lies above them).
// ---------------------------------------------------------------------------------------------------
// `within` (bits 2..14 of an inline slot): index of the bucket's first entry, counted from the first entry of the bucket's GROUP
// of 64 keys (SCT_GRP_*).  The pattern id of inline entry j is then ent[group base + within + j]: one small table of group
// bases (1 025 words per family) and the dense entry table replace a per-slot id table three times the size of the slots.
#define SCT_GRP_SHIFT 6
#define SCT_GRP_PER_FAM ((SCT_KEYS >> SCT_GRP_SHIFT) + 1)
#define SCT_SLOT_WITHIN_MAX 0x1FFFu
SCT_HD uint64_t sct_slot_inline(uint32_t n, const uint32_t *rest, uint32_t within) {
    uint64_t v = n | ((within & SCT_SLOT_WITHIN_MAX) << 2);
    for (uint32_t j = 0; j < n; j++) v |= (uint64_t)(rest[j] & 0xFFFFu) << (16 * (j + 1));
    return v;
}
SCT_HD uint32_t sct_slot_within(uint32_t a) { return (a >> 2) & SCT_SLOT_WITHIN_MAX; }
SCT_HD uint64_t sct_slot_long(uint32_t first, uint32_t n) {
    return 0x8000ull | (n & 0x7FFFu) | ((uint64_t)first << 16) | ((uint64_t)(n >> 15) << 48);
}
SCT_HD bool sct_slot_is_inline(uint32_t a) { return (a & 0x8000u) == 0u; }
SCT_HD uint32_t sct_slot_inline_count(uint32_t a) { return a & 3u; }
SCT_HD uint32_t sct_slot_mid(uint32_t a, uint32_t b) { return (uint32_t)((((uint64_t)b << 32) | a) >> 16); }  // bits 16..47
SCT_HD uint32_t sct_slot_top(uint32_t b) { return b >> 16; }                                                    // bits 48..63
SCT_HD uint32_t sct_slot_long_count(uint32_t a, uint32_t top) { return (a & 0x7FFFu) | (top << 15); }
SCT_HD uint32_t sct_slot_next(uint32_t mid, uint32_t top) { return (uint32_t)((((uint64_t)top << 32) | mid) >> 16); }

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
