// k_offsets.hip.h — prefix sums (hit counts -> CSR offsets, row lengths -> row offsets), the index-format kernels of the
// stream path, and k_fill.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// exclusive scan over n items: per-block sums, one-block scan of the sums, per-item offsets
// ------------------------------------------------------------------------------------------------------------
static constexpr int SCAN_ITEMS = 8;                       // per thread
static constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;        // elements per block

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *total, uint64_t *lds /* >= 5 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint64_t base = 0;
    for (int i = 0; i < w; i++) base += lds[i];
    if (threadIdx.x == blockDim.x - 1) lds[4] = base + inc;
    __syncthreads();
    *total = lds[4];
    const uint64_t r = base + inc - v;
    __syncthreads();
    return r;
}

// what an item contributes: a hit count (rows still marked CNT_DEFER wait for the slow path and count as 0) or a row length
template <typename T>
__device__ __forceinline__ uint64_t scan_value(T v) { return (uint64_t)v; }
template <>
__device__ __forceinline__ uint64_t scan_value<uint32_t>(uint32_t v) { return v == CNT_DEFER ? 0ull : (uint64_t)(v & CNT_MASK); }

// a thread's SCAN_ITEMS consecutive items; one- and two-byte items (row lengths, possibly read in place from page-locked
// host memory) come as ONE 8- or 16-byte load when the address allows it
template <typename T>
__device__ __forceinline__ void load_scan_items(const T *__restrict__ in, int64_t base, int64_t n, uint64_t (&v)[SCAN_ITEMS]) {
    static_assert(SCAN_ITEMS == 8, "the wide loads below fetch eight items");
    if constexpr (sizeof(T) == 1) {
        if (base + SCAN_ITEMS <= n && ((uintptr_t)(in + base) & 7) == 0) {
            const uint64_t w = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(in + base));
#pragma unroll
            for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (w >> (8 * i)) & 0xFFu;
            return;
        }
    } else if constexpr (sizeof(T) == 2) {
        if (base + SCAN_ITEMS <= n && ((uintptr_t)(in + base) & 15) == 0) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 w = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(in + base));
            v[0] = w.x & 0xFFFFu; v[1] = w.x >> 16; v[2] = w.y & 0xFFFFu; v[3] = w.y >> 16;
            v[4] = w.z & 0xFFFFu; v[5] = w.z >> 16; v[6] = w.w & 0xFFFFu; v[7] = w.w >> 16;
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) v[i] = base + i < n ? scan_value<T>(in[base + i]) : 0ull;
}

// `keep` != NULL: the items are also written there (a device copy of an index that was read in place from host memory, so
// that the second pass of the scan does not cross the link again)
template <typename T>
__global__ void __launch_bounds__(256) k_sum_blocks(const T *__restrict__ in, int64_t n, uint64_t *__restrict__ sums, T *__restrict__ keep) {
    __shared__ uint64_t lds[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    load_scan_items<T>(in, base, n, v);
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    if (keep) {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (base + i < n) keep[base + i] = (T)v[i];
    }
    uint64_t total;
    block_exclusive_scan(s, &total, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_sums(uint64_t *__restrict__ sums, int64_t nb, uint64_t *__restrict__ total_out) {
    __shared__ uint64_t lds[8];
    uint64_t carry = 0;
    for (int64_t c0 = 0; c0 < nb; c0 += 256) {
        const int64_t i = c0 + threadIdx.x;
        const uint64_t v = i < nb ? sums[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan(v, &total, lds);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// hit counts -> hit_offs[n+1] (int64, relative to the batch) and, on request, the counts in the caller's compact
// format: cnt8 (uint8, a row with more than 255 hits raises CTR_CNT_OVERFLOW) or cnt32
__global__ void __launch_bounds__(256) k_offsets(const uint32_t *__restrict__ cnt, int64_t n, const uint64_t *__restrict__ sums,
                                                  const uint64_t *__restrict__ total, int64_t *__restrict__ hit_offs,
                                                  uint8_t *__restrict__ cnt8, uint32_t *__restrict__ cnt32, uint32_t *__restrict__ ctr) {
    __shared__ uint64_t lds[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = base + i < n ? (uint32_t)scan_value<uint32_t>(cnt[base + i]) : 0u;
        s += v[i];
    }
    uint64_t tot;
    uint64_t run = sums[blockIdx.x] + block_exclusive_scan(s, &tot, lds);
    bool over = false;
    // the compact counts go out as one 8-byte (cnt8) or two 16-byte (cnt32) stores per thread where the address allows it:
    // they may be the caller's page-locked arrays, where every store instruction is a PCIe write of its own
    const bool whole = base + SCAN_ITEMS <= n;
    const bool wide8 = cnt8 && whole && ((uintptr_t)(cnt8 + base) & 7) == 0;
    const bool wide32 = cnt32 && whole && ((uintptr_t)(cnt32 + base) & 15) == 0;
    uint64_t pack8 = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) {
            hit_offs[base + i] = (int64_t)run;
            if (cnt8) {
                const uint32_t b = v[i] < 255u ? v[i] : 255u;
                if (wide8) pack8 |= (uint64_t)b << (8 * i);
                else cnt8[base + i] = (uint8_t)b;
                over |= v[i] > 255u;
            }
            if (cnt32 && !wide32) cnt32[base + i] = v[i];
        }
        run += v[i];
    }
    if (wide8) *reinterpret_cast<uint64_t *>(cnt8 + base) = pack8;
    if (wide32) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        static_assert(SCAN_ITEMS == 8, "two 16-byte stores");
        *reinterpret_cast<u4 *>(cnt32 + base) = u4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<u4 *>(cnt32 + base + 4) = u4{v[4], v[5], v[6], v[7]};
    }
    if (over) atomicOr(&ctr[CTR_CNT_OVERFLOW], 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) hit_offs[n] = (int64_t)*total;
}

// hit_offs += base (the hits of the earlier chunks of a stream call), in place, n + 1 entries
__global__ void __launch_bounds__(256) k_rebase(int64_t *__restrict__ hit_offs, int64_t n1, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) hit_offs[i] += base;
}

// filter path: slots -> CSR.  One lane per row, the 64 rows of a wave are the 64 rows of one wave of the main filter launch:
// their lists lie back to back from wave_base[r / 64], so a row's list starts behind the counts of the wave's earlier rows
// that the main launch stored itself (not deferred: neither retried nor left to the scan kernels).  Rows finished by the
// retry launch (CNT_RETRIED) have their own offset in slot_off.
__global__ void __launch_bounds__(256) k_fill(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ slots,
                                               const uint32_t *__restrict__ slots_retry, const uint64_t *__restrict__ wave_base,
                                               const uint64_t *__restrict__ slot_off, const int64_t *__restrict__ hit_offs,
                                               const uint32_t *__restrict__ route_cnt, int64_t n, uint32_t *__restrict__ hits,
                                               uint64_t hits_cap, const uint32_t *__restrict__ ctr, uint16_t *__restrict__ hits16) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = r < n;
    const uint32_t raw = in ? cnt[r] : 0u;
    const bool deferred = in && route_cnt[r] == CNT_DEFER;  // written by the scan path's fill pass
    const bool retried = !deferred && raw != CNT_DEFER && (raw & CNT_RETRIED) != 0u;
    const uint32_t c = (deferred || raw == CNT_DEFER) ? 0u : (raw & CNT_MASK);
    uint32_t before = retried ? 0u : c;   // inclusive scan of the counts the main launch stored
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(before, o, 64);
        if (lane >= o) before += t;
    }
    before -= retried ? 0u : c;
    // more hits than `hits` can take (only possible when it is the caller-visible page-locked buffer, sized by an estimate):
    // nothing is written; the host sees the total, enlarges the buffer and fills again
    if (*reinterpret_cast<const uint64_t *>(ctr + CTR_TOTAL) > hits_cap) return;
    if (c == 0) return;
    const int64_t o = hit_offs[r];
    const uint32_t *src = retried ? slots_retry + slot_off[r] : slots + (wave_base[r >> 6] + before);
    // hits16 != NULL: the caller takes 16-bit hits and no row needs the scan kernels' 32-bit array: straight into the result buffer
    if (hits16) {
        for (uint32_t i = 0; i < c; i++) hits16[o + i] = (uint16_t)src[i];
    } else {
        for (uint32_t i = 0; i < c; i++) hits[o + i] = src[i];
    }
}

// k_offsets and k_fill in one kernel (the filter path's normal case: no row waits for the scan kernels).  A block's 2 048 rows
// own one contiguous stretch of the hit array: the lists are gathered from the slot arrays into LDS and leave as aligned
// 16-byte stores (the hit array may be the caller's page-locked buffer: every store instruction is a PCIe write of its own,
// and a two-byte store per hit made k_fill the slowest step of the tail: 14 MB in 0.34 ms).  A thread owns eight consecutive
// rows, eight threads the 64 rows of one wave of the main launch (k_fill's rule for where a row's list starts).
// OutT: uint32_t (hits) or uint16_t (SCT_HITS_U16 straight into the result buffer).
static constexpr uint32_t FILL_STAGE = 8192;   // entries a block gathers in LDS; a block with more writes them one by one
template <typename OutT>
__global__ void __launch_bounds__(256)
k_offsets_fill(const uint32_t *__restrict__ cnt, int64_t n, const uint64_t *__restrict__ sums, const uint64_t *__restrict__ total,
               int64_t *__restrict__ hit_offs, uint8_t *__restrict__ cnt8, uint32_t *__restrict__ cnt32, uint32_t *__restrict__ ctr,
               const uint32_t *__restrict__ slots, const uint32_t *__restrict__ slots_retry, const uint64_t *__restrict__ wave_base,
               const uint64_t *__restrict__ slot_off, OutT *__restrict__ out, uint64_t out_cap) {
    static_assert(SCAN_ITEMS == 8, "eight rows per thread, eight threads per wave of the main launch");
    __shared__ uint64_t lds[8];
    __shared__ __attribute__((aligned(16))) OutT stage[FILL_STAGE + 16];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t raw[SCAN_ITEMS], v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        raw[i] = base + i < n ? cnt[base + i] : 0u;
        v[i] = (uint32_t)scan_value<uint32_t>(raw[i]);
        s += v[i];
    }
    uint64_t tot;
    const uint64_t first = sums[blockIdx.x];   // hits of the rows before this block
    const uint64_t run0 = first + block_exclusive_scan(s, &tot, lds);
    {   // ---- k_offsets' part
        uint64_t run = run0;
        bool over = false;
        const bool whole = base + SCAN_ITEMS <= n;
        const bool wide8 = cnt8 && whole && ((uintptr_t)(cnt8 + base) & 7) == 0;
        const bool wide32 = cnt32 && whole && ((uintptr_t)(cnt32 + base) & 15) == 0;
        uint64_t pack8 = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < n) {
                hit_offs[base + i] = (int64_t)run;
                if (cnt8) {
                    const uint32_t b = v[i] < 255u ? v[i] : 255u;
                    if (wide8) pack8 |= (uint64_t)b << (8 * i);
                    else cnt8[base + i] = (uint8_t)b;
                    over |= v[i] > 255u;
                }
                if (cnt32 && !wide32) cnt32[base + i] = v[i];
            }
            run += v[i];
        }
        if (wide8) *reinterpret_cast<uint64_t *>(cnt8 + base) = pack8;
        if (wide32) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4 *>(cnt32 + base) = u4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<u4 *>(cnt32 + base + 4) = u4{v[4], v[5], v[6], v[7]};
        }
        if (over) atomicOr(&ctr[CTR_CNT_OVERFLOW], 1u);
        if (blockIdx.x == 0 && threadIdx.x == 0) hit_offs[n] = (int64_t)*total;
    }
    // ---- k_fill's part (every condition up to the barrier is the same for the whole block)
    if (*total > out_cap) return;   // more hits than `out` can take: nothing is written, the host enlarges the buffer and fills again
    if (tot == 0) return;
    // lists the main launch stored lie back to back from wave_base[r / 64]: a row's list starts behind those of the wave's earlier rows
    uint32_t own = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) own += (raw[i] != CNT_DEFER && (raw[i] & CNT_RETRIED) == 0u) ? v[i] : 0u;
    uint32_t inc = own;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 8);
        if ((int)(threadIdx.x & 7) >= o) inc += t;
    }
    uint32_t before = inc - own;
    constexpr uint32_t PER = 16 / sizeof(OutT);                                   // entries of a 16-byte store
    const uint32_t sh = (uint32_t)(((uintptr_t)(out + first) & 15) / sizeof(OutT));   // the stretch starts `sh` entries into its first 16 bytes
    const bool fits = tot <= (uint64_t)FILL_STAGE;
    uint64_t run = run0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (v[i]) {
            const int64_t r = base + i;
            const bool retried = (raw[i] & CNT_RETRIED) != 0u;
            const uint32_t *src = retried ? slots_retry + slot_off[r] : slots + (wave_base[r >> 6] + before);
            if (fits) {
                OutT *dst = stage + sh + (uint32_t)(run - first);
                for (uint32_t k = 0; k < v[i]; k++) dst[k] = (OutT)src[k];
            } else {
                for (uint32_t k = 0; k < v[i]; k++) out[run + k] = (OutT)src[k];
            }
            if (!retried) before += v[i];
            run += v[i];
        }
    }
    if (!fits) return;
    __syncthreads();
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t t32 = (uint32_t)tot;
    OutT *o0 = out + first - sh;                                                  // 16-byte aligned; entry e of the block = o0[sh + e]
    const uint32_t v0 = (sh + PER - 1) / PER, v1 = (sh + t32) / PER;             // whole 16-byte groups [v0, v1)
    if (v1 > v0) {
        for (uint32_t j = v0 + threadIdx.x; j < v1; j += 256)
            *reinterpret_cast<u4 *>(o0 + (size_t)j * PER) = *reinterpret_cast<const u4 *>(stage + (size_t)j * PER);
        // the entries in front of the first whole group and behind the last
        const uint32_t head_end = v0 * PER, tail_beg = v1 * PER;
        if (threadIdx.x < PER) {
            const uint32_t e = sh + threadIdx.x;
            if (e < head_end) o0[e] = stage[e];
        } else if (threadIdx.x < 2 * PER) {
            const uint32_t e = tail_beg + (threadIdx.x - PER);
            if (e < sh + t32) o0[e] = stage[e];
        }
    } else {
        for (uint32_t e = sh + threadIdx.x; e < sh + t32; e += 256) o0[e] = stage[e];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Index formats of the stream path (include/sctagger_hip.h: SCT_IDX_*): the caller's row index of one chunk ->
// uint32 offsets relative to the chunk's first byte, checked on the way (CTR_BAD_ROW = smallest offending row of the
// chunk).  Offsets: n + 1 entries in[0..n] are uploaded, out[i] = in[i] - in[0]; they must not decrease and must end
// at `bytes`.  Lengths: n entries, out = their exclusive prefix sum (k_sum_blocks / k_scan_sums first).
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_index_offs(const T *__restrict__ in, int64_t n, int64_t bytes, uint32_t *__restrict__ out,
                                                     unsigned long long *__restrict__ bad_row) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const int64_t base = (int64_t)in[0];
    const int64_t a = (int64_t)in[i] - base;
    bool bad = a < 0 || a > bytes;
    if (i < n) bad |= (int64_t)in[i + 1] < (int64_t)in[i];
    else bad |= a != bytes;
    if (bad) atomicMin(bad_row, (unsigned long long)(i < n ? i : (n > 0 ? n - 1 : 0)));
    out[i] = (uint32_t)(a < 0 ? 0 : (a > bytes ? bytes : a));
}

template <typename T>
__global__ void __launch_bounds__(256) k_index_lens(const T *__restrict__ lens, int64_t n, const uint64_t *__restrict__ sums,
                                                     const uint64_t *__restrict__ total, int64_t bytes, uint32_t *__restrict__ out,
                                                     unsigned long long *__restrict__ bad_row) {
    __shared__ uint64_t lds[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    load_scan_items<T>(lens, base, n, v);
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint64_t tot;
    uint64_t run = sums[blockIdx.x] + block_exclusive_scan(s, &tot, lds);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = (uint32_t)(run < (uint64_t)bytes ? run : (uint64_t)bytes);
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[n] = (uint32_t)(*total < (uint64_t)bytes ? *total : (uint64_t)bytes);
        if (*total != (uint64_t)bytes) atomicMin(bad_row, (unsigned long long)(n > 0 ? n - 1 : 0));
    }
}

// uint32 chunk offsets -> int64 (the scan kernels of the slow path take int64 offsets)
__global__ void __launch_bounds__(256) k_widen_offs(const uint32_t *__restrict__ in, int64_t n1, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) out[i] = (int64_t)in[i];
}

// ------------------------------------------------------------------------------------------------------------
// k_route_all: routing without the filter kernel (barcode lengths or error budgets the seed filter does not cover, or
// option force_exhaustive): one lane per row.  Rows shorter than bl - max_err are finished here; the others go to the
// restricted-start DP (a byte outside ACGT while the reference's prefix length k > 0) or to the exhaustive scan.
// ------------------------------------------------------------------------------------------------------------
template <typename OffT>
__global__ void __launch_bounds__(256)
k_route_all(const uint8_t *__restrict__ segs, const OffT *__restrict__ offs, int64_t seg_bytes, int64_t n, int bl, int max_err,
            int prefix_len, uint8_t *__restrict__ min_e, uint32_t *__restrict__ cnt, int32_t *__restrict__ list_scan,
            int32_t *__restrict__ list_dp, uint32_t *__restrict__ ctr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int64_t b0, len;
    sct_row_span(offs, r, seg_bytes, b0, len);
    if (len < bl - max_err) {
        min_e[r] = SCT_NO_HIT;
        cnt[r] = 0;
        return;
    }
    // a row the reference inserts windows of: it raises on a byte >= 128 in any of them (sct_byte_is_read, sct_core.h)
    bool nonacgt = false;
    for (int64_t i = 0; i < len; i++) {
        const uint8_t b = segs[b0 + i];
        nonacgt |= !sct_is_acgt(b);
        if (b >= 128 && sct_byte_is_read(segs + b0, len, i, bl, max_err, prefix_len)) {
            atomicMin(reinterpret_cast<unsigned long long *>(ctr + CTR_BAD_BYTE), (unsigned long long)(b0 + i));
            break;
        }
    }
    if (nonacgt && prefix_len > 0) list_dp[atomicAdd(&ctr[CTR_DP], 1u)] = (int32_t)r;
    else list_scan[atomicAdd(&ctr[CTR_SCAN], 1u)] = (int32_t)r;
    cnt[r] = CNT_DEFER;
}

// hits (uint32 on the device) -> uint16 in the caller-visible buffer (SCT_HITS_U16: half the bytes over the link when the
// barcode set has at most 32 768 entries).  `total` is read on the device; nothing is written if it exceeds `cap`.
__global__ void __launch_bounds__(256) k_narrow_hits(const uint32_t *__restrict__ hits, const uint32_t *__restrict__ ctr, uint64_t cap,
                                                      uint16_t *__restrict__ out) {
    const uint64_t total = *reinterpret_cast<const uint64_t *>(ctr + CTR_TOTAL);
    if (total > cap) return;
    // eight entries (16 bytes out) per thread where the addresses allow it
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i0 >= total) return;
    if (i0 + 8 <= total && ((uintptr_t)(out + i0) & 15) == 0) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 a = *reinterpret_cast<const u4 *>(hits + i0), b = *reinterpret_cast<const u4 *>(hits + i0 + 4);
        *reinterpret_cast<u4 *>(out + i0) = u4{(a.x & 0xFFFFu) | (a.y << 16), (a.z & 0xFFFFu) | (a.w << 16), (b.x & 0xFFFFu) | (b.y << 16),
                                               (b.z & 0xFFFFu) | (b.w << 16)};
    } else {
        for (uint64_t i = i0; i < i0 + 8 && i < total; i++) out[i] = (uint16_t)hits[i];
    }
}
