// k_decode.hip.h — ASCII segment bytes -> 2-bit bit-planes, inside the kernel that consumes them.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
//
// Round 1 had a separate pack kernel that wrote 24 bytes of planes per row for the filter kernel to read back
// (0.2 ms and 0.5 GB of HBM traffic per 10 M rows).  The filter wave now reads its 64 rows' ASCII bytes itself
// (non-temporal: the bytes are used once; through an LDS stage, see sct_decode_rows_staged, or word by word per lane),
// turns every aligned 32-bit word into three nibbles with the multiply-gather of sct_decode4 (sct_core.h) — low code
// bit, high code bit, "not one of ACGT" — and appends the nibbles to raw planes; one funnel shift by the row's byte
// misalignment and a length mask finish the job.
#pragma once

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// W = uint32_t: rows of at most 31 characters, W = uint64_t: at most 63.  `len` = characters to decode for this lane
// (0 = none; the lane still takes part in the wave-uniform loop), `skip` = 0..3 bytes of the first word that precede the
// row, fetch(k) = the k-th aligned 32-bit word covering the row.  Every lane of the wave must call it.
//   lo, hi   planes of map_char's code (scTagger.py:512-516): any byte that is not A, C, G or T decodes to 0 = 'A'
//   nonacgt  the row holds such a byte
template <typename W, typename Fetch>
__device__ __forceinline__ void sct_decode_words_mask(Fetch &&fetch, int skip, int len, W &lo, W &hi, W &bad) {
    constexpr int MAXC = sizeof(W) == 4 ? 31 : 63;
    constexpr int MAXW = (MAXC + 3 + 3) / 4;        // words covering MAXC characters at the worst misalignment: 9 / 17
    constexpr int NR = (MAXW + 7) / 8;              // 32-bit registers per raw plane (8 nibbles each): 2 / 3
    const int nwords = len > 0 ? (len + skip + 3) >> 2 : 0;
    const int wave_words = __builtin_amdgcn_readfirstlane(wave_max_i(nwords));
    uint32_t rl[NR], rh[NR], rb[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) rl[i] = rh[i] = rb[i] = 0u;
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        if (k < wave_words) {  // wave-uniform
            uint32_t w = 0x41414141u;
            if (k < nwords) w = fetch(k);
            uint32_t l4, h4, b4;
            sct_decode4(w, l4, h4, b4);
            rl[k >> 3] |= l4 << (4 * (k & 7));
            rh[k >> 3] |= h4 << (4 * (k & 7));
            rb[k >> 3] |= b4 << (4 * (k & 7));
        }
    }
    // character i sits at raw bit i + skip
    const W keep = len >= (int)(8 * sizeof(W)) ? ~(W)0 : (((W)1 << len) - 1);
    if constexpr (sizeof(W) == 4) {
        lo = (W)__builtin_amdgcn_alignbit(rl[1], rl[0], (uint32_t)skip) & keep;
        hi = (W)__builtin_amdgcn_alignbit(rh[1], rh[0], (uint32_t)skip) & keep;
        bad = (W)__builtin_amdgcn_alignbit(rb[1], rb[0], (uint32_t)skip) & keep;
    } else {
        lo = ((W)__builtin_amdgcn_alignbit(rl[1], rl[0], (uint32_t)skip) | ((W)__builtin_amdgcn_alignbit(rl[2], rl[1], (uint32_t)skip) << 32)) & keep;
        hi = ((W)__builtin_amdgcn_alignbit(rh[1], rh[0], (uint32_t)skip) | ((W)__builtin_amdgcn_alignbit(rh[2], rh[1], (uint32_t)skip) << 32)) & keep;
        bad = ((W)__builtin_amdgcn_alignbit(rb[1], rb[0], (uint32_t)skip) | ((W)__builtin_amdgcn_alignbit(rb[2], rb[1], (uint32_t)skip) << 32)) & keep;
    }
}
template <typename W, typename Fetch>
__device__ __forceinline__ void sct_decode_words(Fetch &&fetch, int skip, int len, W &lo, W &hi, bool &nonacgt) {
    W bad;
    sct_decode_words_mask<W>(fetch, skip, len, lo, hi, bad);
    nonacgt = bad != 0;
}

// The row's words straight from memory: each lane fetches the aligned 32-bit words covering its own row (neighbouring
// lanes read neighbouring words of one contiguous stretch; a word shared by two rows is fetched by both).  Used for
// listed rows (retry launch) and for waves whose rows do not fit the LDS stage.
template <typename W>
__device__ __forceinline__ void sct_decode_row(const uint8_t *__restrict__ segs, int64_t start, int len, W &lo, W &hi, bool &nonacgt) {
    // the word containing byte `start` of the buffer: aligned by ADDRESS (the buffer itself may start anywhere), so a word
    // that holds one byte of the buffer never leaves that byte's page
    const uintptr_t at = (uintptr_t)segs + (uintptr_t)start;
    // (an address computed as an integer loses its address space: say "global" again, or the loads are issued as flat ones)
    const __attribute__((address_space(1))) uint32_t *wp = (const __attribute__((address_space(1))) uint32_t *)(at & ~(uintptr_t)3);
    sct_decode_words<W>([&](int k) { return __builtin_nontemporal_load(wp + k); }, (int)(at & 3), len, lo, hi, nonacgt);
}

// ... and with the mask of the bytes that are not one of ACGT (bit i = character i): the special launch derives the allowed
// window starts from it (sct_allowed_starts, sct_core.h)
template <typename W>
__device__ __forceinline__ void sct_decode_row_mask(const uint8_t *__restrict__ segs, int64_t start, int len, W &lo, W &hi, W &bad) {
    const uintptr_t at = (uintptr_t)segs + (uintptr_t)start;
    const __attribute__((address_space(1))) uint32_t *wp = (const __attribute__((address_space(1))) uint32_t *)(at & ~(uintptr_t)3);
    sct_decode_words_mask<W>([&](int k) { return __builtin_nontemporal_load(wp + k); }, (int)(at & 3), len, lo, hi, bad);
}

// The wave's rows through LDS: in CSR order the rows of a wave are one contiguous stretch of bytes [span_lo, span_hi).
// The wave loads it ONCE with aligned 16-byte loads, one piece per lane and step (fully coalesced — every byte of the
// input crosses the memory system, or the PCIe link when the buffer is page-locked host memory, exactly once), parks it
// in `stage` (LDS the filter does not need before its first family runs) and every lane then picks its row's words
// from there.  Returns false, having done nothing, if the stretch does not fit `stage_bytes`.
typedef uint32_t sct_u4 __attribute__((ext_vector_type(4)));
template <typename W>
__device__ __forceinline__ bool sct_decode_rows_staged(const uint8_t *__restrict__ segs, int64_t start, int len, uint32_t *stage, int stage_bytes,
                                                       W &lo, W &hi, W &bad) {
    const int lane = threadIdx.x & 63;
    // stretch of the rows that are decoded (len > 0); lanes without one do not widen it
    const uint64_t a = (uint64_t)(uintptr_t)segs + (uint64_t)start;
    uint64_t smin = len > 0 ? a : ~0ull, smax = len > 0 ? a + (uint64_t)len : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t x = __shfl_xor(smin, o, 64), y = __shfl_xor(smax, o, 64);
        smin = x < smin ? x : smin;
        smax = y > smax ? y : smax;
    }
    if (smax <= smin) {  // nothing to decode in this wave
        lo = hi = bad = 0;
        return true;
    }
    const uint64_t base = smin & ~15ull;
    const uint64_t nbytes = smax - base;
    if (nbytes > (uint64_t)stage_bytes) return false;
    const int n16 = (int)((nbytes + 15) >> 4);
    const __attribute__((address_space(1))) sct_u4 *src = (const __attribute__((address_space(1))) sct_u4 *)(uintptr_t)base;
    for (int i = lane; i < n16; i += 64) {
        const sct_u4 v = __builtin_nontemporal_load(src + i);
        *reinterpret_cast<sct_u4 *>(stage + 4 * i) = v;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t so = len > 0 ? (uint32_t)(a - base) : 0u;
    const uint32_t *wp = stage + (so >> 2);
    sct_decode_words_mask<W>([&](int k) { return wp[k]; }, (int)(so & 3u), len, lo, hi, bad);
    __builtin_amdgcn_wave_barrier();  // the stage is the filter's queue and hit lists from here on
    return true;
}

// The BLOCK's rows through LDS (main launch, four waves per block): the 256 rows of a block are one contiguous stretch
// [lo, hi) of the byte buffer, and all 256 threads load it together, every 16-byte piece once.  Per-wave staging (above) makes
// two neighbouring waves both ask for the line their stretches share; for HBM that is nothing, but page-locked HOST memory is
// not cached on the device and every such line crosses the PCIe link twice: a plain streaming kernel reads the link at
// 56 GB/s, per-wave stretches of 1.3 - 1.4 KB at arbitrary offsets at 50 (tools/link_shape_probe.hip,
// profiles/r03_link_shape_probe.json), stretches of 5.6 KB at 56.  The caller synchronises the block before any lane reads
// `stage` and again before anything else is written to that LDS.
__device__ __forceinline__ void sct_block_stage_load(const uint8_t *__restrict__ segs, int64_t lo, int64_t hi, uint32_t *stage) {
    // whole 128-byte lines from the line that holds the first byte: a pass of the loop below (256 threads x 16 bytes = 32 lines)
    // then ends on a line boundary, and no line is asked for by two instructions.  (A line never leaves its page: reading the
    // bytes of the first and last line that lie outside [lo, hi) is safe, and they belong to the neighbouring blocks' rows anyway.)
    const uint64_t a = (uint64_t)(uintptr_t)segs + (uint64_t)lo;
    const uint64_t base = a & ~127ull;
    const int n16 = (int)(((((uint64_t)(uintptr_t)segs + (uint64_t)hi - base + 127) >> 7)) << 3);
    const __attribute__((address_space(1))) sct_u4 *src = (const __attribute__((address_space(1))) sct_u4 *)(uintptr_t)base;
    for (int i = (int)threadIdx.x; i < n16; i += (int)blockDim.x) {
        const sct_u4 v = __builtin_nontemporal_load(src + i);
        *reinterpret_cast<sct_u4 *>(stage + 4 * i) = v;
    }
}
// ... and a lane's row out of that stage (start = the row's byte offset in the buffer, lo = the block stretch's)
// -> the row's byte offset inside the stage (so that the caller can look at its raw bytes while they are there)
template <typename W>
__device__ __forceinline__ uint32_t sct_decode_row_from_block_stage(const uint8_t *__restrict__ segs, int64_t lo, int64_t start, int len, const uint32_t *stage,
                                                                    W &plo, W &phi, W &bad) {
    const uint64_t base = ((uint64_t)(uintptr_t)segs + (uint64_t)lo) & ~127ull;
    const uint32_t so = len > 0 ? (uint32_t)((uint64_t)(uintptr_t)segs + (uint64_t)start - base) : 0u;
    const uint32_t *wp = stage + (so >> 2);
    sct_decode_words_mask<W>([&](int k) { return wp[k]; }, (int)(so & 3u), len, plo, phi, bad);
    return so;
}

// Bytes >= 128: the reference raises when it reads one (sct_byte_is_read, sct_core.h).  Only bytes that are not one of ACGT can
// be such a byte: `bad` = their positions in the row.  row = the row's bytes (in memory, or still in the LDS stage), `at` = the row's
// offset in the caller's buffer (what the error message quotes).  Rare: called for rows with a bad byte only.
__device__ __forceinline__ void sct_report_high_bytes(const uint8_t *row, int len, uint64_t bad, int64_t at, int bl, int max_err, int k, uint32_t *ctr) {
    while (bad != 0) {
        const int i = __builtin_ctzll(bad);
        bad &= bad - 1;
        if (row[i] >= 128 && sct_byte_is_read(row, len, i, bl, max_err, k)) {
            atomicMin(reinterpret_cast<unsigned long long *>(ctr + CTR_BAD_BYTE), (unsigned long long)(at + i));
            break;
        }
    }
}
