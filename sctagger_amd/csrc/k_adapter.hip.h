// k_adapter.hip.h — k_adapter_count / k_adapter_fill: extract_lr_bc adapter alignment.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// extract_lr_bc (reference scTagger.py:176-196): the adapter and its reverse complement are aligned to every whole
// read in edlib's HW (infix) mode.  One lane per read, Myers bit-vector over the read's bytes (fetched as aligned
// 32-bit words), match masks from a 256-entry table in LDS (symbols are raw bytes, as in edlib).
//   k_adapter_count   both patterns in one pass: minimum distance of each and the number of end positions at
//                     which it is reached; decides the strand ('+' if the adapter is strictly better, '-' if its
//                     reverse complement is, none on a tie) and how many locations the read will report
//   k_adapter_fill    second pass with the winning pattern only: writes the locations — end positions for '+';
//                     for '-' the start of the longest optimal alignment ending there (edlib: reverse alignment in
//                     prefix mode, last position with the same score), stored as start - len - 1 like the reference
// ------------------------------------------------------------------------------------------------------------
// one text byte through both automata, tracking for each the minimum, how often it is reached and where first / last
struct AdapterTrack {
    MyersState<uint32_t> s1, s2;
    int best1, best2;
    uint32_t c1 = 0, c2 = 0;
    int64_t first1 = 0, last1 = 0, first2 = 0, last2 = 0;
    __device__ __forceinline__ void step(uint32_t e1, uint32_t e2, int m, int64_t i) {
        myers_step_eq<uint32_t, false>(s1, e1, m);
        myers_step_eq<uint32_t, false>(s2, e2, m);
        if (s1.score <= best1) {
            if (s1.score < best1) {
                c1 = 0;
                first1 = i;
            }
            c1++;
            last1 = i;
            best1 = s1.score;
        }
        if (s2.score <= best2) {
            if (s2.score < best2) {
                c2 = 0;
                first2 = i;
            }
            c2++;
            last2 = i;
            best2 = s2.score;
        }
    }
};

// The read is streamed in 64-byte aligned chunks held in registers (four 16-byte loads per lane and chunk: every
// cache line is fetched exactly once, although neighbouring lanes walk different reads); bytes of the first and
// last chunk that lie outside the read are skipped.  (Fetching word by word made every lane revisit its line 16
// times with 64 different lines per wave instruction in between: 8.5 ms instead of 2.5 ms per 2 GB.)
__global__ void __launch_bounds__(256)
k_adapter_count(const uint8_t *__restrict__ reads, const int64_t *__restrict__ offs, int64_t n, const uint32_t *__restrict__ peq,
                int m, int8_t *__restrict__ strand, int32_t *__restrict__ dist, uint32_t *__restrict__ cnt,
                int64_t *__restrict__ win_first, int64_t *__restrict__ win_last) {
    __shared__ uint32_t P1[256], P2[256];
    P1[threadIdx.x] = peq[threadIdx.x];
    P2[threadIdx.x] = peq[256 + threadIdx.x];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b0 = offs[r], b1 = offs[r + 1], len = b1 - b0;
    AdapterTrack T;
    myers_init(T.s1, m);
    myers_init(T.s2, m);
    T.best1 = T.best2 = m + 1;
    const uint4 *base = reinterpret_cast<const uint4 *>(reads);
    for (int64_t c = b0 >> 6; (c << 6) < b1; c++) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = base[c * 4 + q];
        const int64_t p0 = c << 6;
        const bool whole = p0 >= b0 && p0 + 64 <= b1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int64_t pos = p0 + q * 16 + k * 4 + j;
                    if (whole || (pos >= b0 && pos < b1)) {
                        const uint32_t b = (w[k] >> (8 * j)) & 0xFFu;
                        T.step(P1[b], P2[b], m, pos - b0);
                    }
                }
            }
        }
    }
    int8_t st = 0;
    int32_t d = -1;
    uint32_t cc = 0;
    int64_t wf = 0, wl = -1;
    if (len > 0 && T.best1 != T.best2) {
        if (T.best1 < T.best2) {
            st = 1;
            d = T.best1;
            cc = T.c1;
            wf = T.first1;
            wl = T.last1;
        } else {
            st = 2;
            d = T.best2;
            cc = T.c2;
            wf = T.first2;
            wl = T.last2;
        }
    }
    strand[r] = st;
    dist[r] = d;
    cnt[r] = cc;
    win_first[r] = wf;
    win_last[r] = wl;
}

// Second pass, winning pattern only, over the window [first end - 2m, last end] of the read: D[m][j] depends on at
// most the 2m characters before j (an alignment of an m-character pattern with <= m edits is at most 2m long and the
// start is free), so an automaton started 2m characters before the first reported end gives the same scores there.
__global__ void __launch_bounds__(256)
k_adapter_fill(const uint8_t *__restrict__ reads, const int64_t *__restrict__ offs, int64_t n, const uint32_t *__restrict__ peq,
               int m, const int8_t *__restrict__ strand, const int32_t *__restrict__ dist, const int64_t *__restrict__ win_first,
               const int64_t *__restrict__ win_last, const int64_t *__restrict__ loc_offs, int32_t *__restrict__ locs) {
    __shared__ uint32_t P1[256], P2[256], P2R[256];
    P1[threadIdx.x] = peq[threadIdx.x];
    P2[threadIdx.x] = peq[256 + threadIdx.x];
    P2R[threadIdx.x] = peq[512 + threadIdx.x];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int st = strand[r];
    if (st == 0) return;
    const int64_t b0 = offs[r], len = offs[r + 1] - b0;
    const int d = dist[r];
    const uint32_t *P = st == 1 ? P1 : P2;
    int64_t out = loc_offs[r];
    const int64_t first = win_first[r], last = win_last[r];
    const int64_t from = first - 2 * m > 0 ? first - 2 * m : 0;
    MyersState<uint32_t> s;
    myers_init(s, m);
    // bytes through a one-word cache: the forward walk and the short backward walks touch each word once
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(reads);
    int64_t wi = -1;
    uint32_t word = 0;
    auto byte_at = [&](int64_t i) {
        const int64_t g = b0 + i;
        if ((g >> 2) != wi) {
            wi = g >> 2;
            word = wp[wi];
        }
        return (word >> (8 * (g & 3))) & 0xFFu;
    };
    for (int64_t i = from; i <= last; i++) {
        myers_step_eq<uint32_t, false>(s, P[byte_at(i)], m);
        if (i >= first && s.score == d) {
            if (st == 1) {
                locs[out++] = (int32_t)i;
            } else {
                // reverse alignment in prefix mode from this end; a position further back than m + d characters
                // cannot reach score d (the characters before the match are charged)
                MyersState<uint32_t> q;
                myers_init(q, m);
                int lastpos = 0;
                const int lim = (int)((int64_t)(m + d) < i + 1 ? (int64_t)(m + d) : i + 1);
                for (int t = 0; t < lim; t++) {
                    myers_step_eq<uint32_t, true>(q, P2R[byte_at(i - t)], m);
                    if (q.score == d) lastpos = t;
                }
                locs[out++] = (int32_t)((i - lastpos) - len - 1);
            }
        }
    }
}

