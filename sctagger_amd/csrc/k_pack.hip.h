// k_pack.hip.h — k_pack: ASCII segments -> bit-planes + routing.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, pack, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// k_pack: one lane per segment. ASCII -> bit-planes of the first 63 characters, flags, and routing.
//   route 0: filter path; route 1: exhaustive Myers scan; route 2: restricted-start DP scan
// Segments shorter than bl - max_err cannot match anything (every window the reference inserts is at least
// that long, scTagger.py:681) and are finished here.
// ------------------------------------------------------------------------------------------------------------
__global__ void k_pack(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, int64_t n, int bl,
                       int max_err, int prefix_len, int filter_maxlen, uint64_t *__restrict__ lo_out,
                       uint64_t *__restrict__ hi_out, uint32_t *__restrict__ meta_out, uint8_t *__restrict__ min_e,
                       uint32_t *__restrict__ cnt, int32_t *__restrict__ list_scan, int32_t *__restrict__ list_dp,
                       uint32_t *__restrict__ counters) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b0 = offs[r];
    const int64_t len = offs[r + 1] - b0;
    // The segment's bytes are fetched as the aligned 32-bit words that cover them (neighbouring lanes read
    // neighbouring words, so the loads coalesce; the buffer is padded past its end).  Per byte: valid = one of
    // A C G T (bit test on b & 31 for b in 0x40..0x5F), code = ((b >> 1) ^ (b >> 2)) & 3 for those, else 0.
    uint32_t lo0 = 0, lo1 = 0, hi0 = 0, hi1 = 0;  // planes of characters 0..31 and 32..62
    bool nonacgt = false;
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(segs) + (b0 >> 2);
    const int skip = (int)(b0 & 3);
    const int nbytes = (int)(len < (1 << 20) ? len : (1 << 20)) + skip;  // bytes from the first word's start
    constexpr uint32_t ACGT_SET = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20);
    for (int k = 0; k * 4 < nbytes; k++) {
        const uint32_t w = wp[k];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = k * 4 + j - skip;  // character index
            if (i >= 0 && i < (int)len) {
                const uint32_t b = (w >> (8 * j)) & 0xFFu;
                const bool ok = (b >> 5) == 2u && ((ACGT_SET >> (b & 31u)) & 1u);
                const uint32_t c = ok ? ((b >> 1) ^ (b >> 2)) & 3u : 0u;
                nonacgt |= !ok;
                if (i < 32) {
                    lo0 |= (c & 1u) << i;
                    hi0 |= (c >> 1) << i;
                } else if (i < PLANE_MAX) {
                    lo1 |= (c & 1u) << (i - 32);
                    hi1 |= (c >> 1) << (i - 32);
                }
            }
        }
    }
    for (int64_t i = (1 << 20); i < len; i++) nonacgt |= !sct_is_acgt(segs[b0 + i]);  // absurdly long rows
    const uint64_t lo = lo0 | ((uint64_t)lo1 << 32), hi = hi0 | ((uint64_t)hi1 << 32);
    uint32_t meta = (uint32_t)(len < 255 ? len : 255);
    if (nonacgt) meta |= META_NONACGT;
    if (len > PLANE_MAX) meta |= META_LONG;
    lo_out[r] = lo;
    hi_out[r] = hi;
    meta_out[r] = meta;
    if (len < bl - max_err) {
        min_e[r] = SCT_NO_HIT;
        cnt[r] = 0;
        return;
    }
    if (nonacgt && prefix_len > 0) {
        const uint32_t k = atomicAdd(&counters[1], 1u);
        list_dp[k] = (int32_t)r;
        cnt[r] = CNT_DEFER;
    } else if (len > filter_maxlen) {  // filter_maxlen: 0 = no filter path, 31 = lane variant, 63 = wave kernels
        const uint32_t k = atomicAdd(&counters[0], 1u);
        list_scan[k] = (int32_t)r;
        cnt[r] = CNT_DEFER;
    } else {
        cnt[r] = 0;  // filter path fills it in
    }
}

