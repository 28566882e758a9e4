// k_whitelist.hip.h — k_wl_count: extract_sr_bc_from_lr whitelist probes.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// k_wl_count: the matching loop of extract_sr_bc_from_lr (reference scTagger.py:818-826).  One lane per row.
// Every occurrence of every dictionary key in the segment counts as a hit (Automaton.iter); rows with exactly one
// hit add 1 to that barcode's counter (index |value|) and lower its first-seen row (the Counter's insertion order
// breaks ties in the reference's sort).  Genuinely memory-bound: ~7 random 24-byte table probes per row.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_wl_count(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, const uint8_t *__restrict__ skip, int64_t n,
           const SctWlSlot *__restrict__ table, uint32_t mask, uint32_t len_set, uint32_t *__restrict__ counts,
           uint32_t *__restrict__ first_seen) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (skip[r]) return;
    const int64_t b0 = offs[r];
    const int64_t len = offs[r + 1] - b0;
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(segs) + (b0 >> 2);
    const int sk = (int)(b0 & 3);
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // the last 16 bytes read, newest in the top byte of w3
    uint32_t hits = 0;
    int32_t value = 0;
    uint32_t word = 0;
    for (int64_t i = 0; i < len; i++) {
        const int64_t g = i + sk;
        if ((g & 3) == 0 || i == 0) word = wp[g >> 2];
        const uint32_t b = (word >> (8 * (g & 3))) & 0xFFu;
        w0 = (w0 >> 8) | (w1 << 24);
        w1 = (w1 >> 8) | (w2 << 24);
        w2 = (w2 >> 8) | (w3 << 24);
        w3 = (w3 >> 8) | (b << 24);
        for (uint32_t ls = len_set; ls; ls &= ls - 1) {
            const int L = __ffs(ls) - 1;  // key length, 1..16
            if (i + 1 < L) continue;
            // the last L bytes, zero padded: shift the 128-bit window right by 16-L bytes
            const int sb = 16 - L, dw = sb >> 2, bs = (sb & 3) * 8;
            uint32_t a0 = dw == 0 ? w0 : dw == 1 ? w1 : dw == 2 ? w2 : w3;
            uint32_t a1 = dw == 0 ? w1 : dw == 1 ? w2 : dw == 2 ? w3 : 0u;
            uint32_t a2 = dw == 0 ? w2 : dw == 1 ? w3 : 0u;
            uint32_t a3 = dw == 0 ? w3 : 0u;
            if (bs) {
                a0 = (a0 >> bs) | (a1 << (32 - bs));
                a1 = (a1 >> bs) | (a2 << (32 - bs));
                a2 = (a2 >> bs) | (a3 << (32 - bs));
                a3 = a3 >> bs;
            }
            uint32_t idx = sct_wl_hash(a0, a1, a2, a3, (uint32_t)L) & mask;
            for (;;) {
                const SctWlSlot s = table[idx];
                if (s.len == 0) break;
                if (s.len == (uint32_t)L && s.k[0] == a0 && s.k[1] == a1 && s.k[2] == a2 && s.k[3] == a3) {
                    hits++;
                    value = s.value;
                    break;
                }
                idx = (idx + 1) & mask;
            }
        }
    }
    if (hits == 1) {
        const uint32_t b = (uint32_t)(value < 0 ? -value : value);
        atomicAdd(&counts[b], 1u);
        atomicMin(&first_seen[b], (uint32_t)r);
    }
}

