// k_scan.hip.h — the exhaustive scan path: k_scan, k_scan_pk, k_scan_reduce.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// k_scan<DP, FILL>: one wave per listed segment, lanes = patterns (64 per step), text wave-uniform.
//   DP = false: Myers bit-vector (free start anywhere);  DP = true: plain DP, row 0 carries the allowed starts
//   FILL = false: writes min_e / cnt;  FILL = true: writes the pattern ids at the minimum, ascending
// The text is loaded 64 characters at a time, one per lane, and broadcast column by column with readlane.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static constexpr int DP_BIG = 100;

template <bool DP, bool FILL>
__global__ void __launch_bounds__(256)
k_scan(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, int64_t seg_bytes, const int32_t *__restrict__ list,
       int64_t n_list, int n_slices, const uint2 *__restrict__ pat, int np, int bl, int max_err, int prefix_len,
       const uint8_t *__restrict__ min_e, uint32_t *__restrict__ part_best, uint32_t *__restrict__ part_cnt,
       const int64_t *__restrict__ hit_offs, uint32_t *__restrict__ hits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int tiles = (np + 63) / 64;
    const int tiles_per_slice = (tiles + n_slices - 1) / n_slices;
    for (int64_t w = wave; w < n_list * n_slices; w += n_waves) {
        const int64_t li = w / n_slices;
        const int slice = (int)(w - li * n_slices);
        const int64_t r = (int64_t)list[li];
        int64_t b0, len;
        sct_row_span(offs, r, seg_bytes, b0, len);
        const int p_begin = slice * tiles_per_slice * 64;
        const int p_end = min(np, p_begin + tiles_per_slice * 64);
        int target = 0;
        int64_t out_base = 0;
        if (FILL) {
            target = min_e[r];
            if (target == SCT_NO_HIT) continue;
            // a slice whose own minimum is worse than the row's holds no hit: nothing to recompute
            if (part_best[w] != (uint32_t)target) continue;
            // hits of the earlier slices at the same minimum come first
            uint32_t mine = 0;
            for (int sl = lane; sl < slice; sl += 64)
                if (part_best[li * n_slices + sl] == (uint32_t)target) mine += part_cnt[li * n_slices + sl];
            out_base = hit_offs[r] + wave_sum(mine);
        }
        int lane_best = max_err + 1;
        uint32_t lane_cnt = 0;
        uint32_t written = 0;  // wave-uniform
        for (int p0 = p_begin; p0 < p_end; p0 += 64) {
            const int p = p0 + lane;
            const bool pvalid = p < p_end;
            const uint2 pp = pvalid ? pat[p] : make_uint2(0, 0);
            int d;
            if (!DP) {
                MyersState<uint32_t> s;
                myers_init(s, bl);
                int best = bl;
                for (int64_t c0 = 0; c0 < len; c0 += 64) {
                    const int64_t ci = c0 + lane;
                    const uint32_t my = ci < len ? sct_code(segs[b0 + ci]) : 0u;
                    const int ncol = (int)(len - c0 < 64 ? len - c0 : 64);
                    for (int j = 0; j < ncol; j++) {
                        const uint32_t c = __builtin_amdgcn_readlane(my, j);
                        myers_step<uint32_t>(s, pp.x, pp.y, c, bl);
                        best = min(best, s.score);
                    }
                }
                d = best;
            } else {
                // col[i] = D[i][j] for i = 0..bl, in registers.  Row 0 is the distance back to the nearest text
                // position at which the reference would start a window (scTagger.py:679-682): a window may start at
                // position s iff s + k <= len and its first k raw bytes are all in "ACGT" (k = prefix_len).  Letting a
                // window start at s lowers D[0][s] to 0, and through vertical moves D[i][s] to at most i.
                int col[SCT_MAX_BL + 1];
#pragma unroll
                for (int i = 0; i <= SCT_MAX_BL; i++) col[i] = DP_BIG;
                int best = DP_BIG;
                for (int64_t c0 = 0; c0 < len; c0 += 64) {
                    const int64_t ci = c0 + lane;
                    uint32_t my = 0;
                    bool al = false;
                    if (ci < len) {
                        my = sct_code(segs[b0 + ci]);
                        al = ci + prefix_len <= len;
                        for (int q = 0; q < prefix_len && al; q++) al = sct_is_acgt(segs[b0 + ci + q]);
                    }
                    const uint64_t almask = __ballot(al);
                    const int ncol = (int)(len - c0 < 64 ? len - c0 : 64);
                    for (int j = 0; j < ncol; j++) {
                        const uint32_t c = __builtin_amdgcn_readlane(my, j);
                        if ((almask >> j) & 1) {
#pragma unroll
                            for (int i = 0; i <= SCT_MAX_BL; i++) col[i] = min(col[i], i);
                        }
                        int diag = col[0];
                        col[0] = min(col[0] + 1, DP_BIG);
#pragma unroll
                        for (int i = 1; i <= SCT_MAX_BL; i++) {
                            if (i <= bl) {
                                const uint32_t pc = ((pp.x >> (i - 1)) & 1u) | (((pp.y >> (i - 1)) & 1u) << 1);
                                const int up = col[i];
                                const int v = min(min(diag + (pc != c ? 1 : 0), up + 1), col[i - 1] + 1);
                                diag = up;
                                col[i] = min(v, DP_BIG);
                                if (i == bl) best = min(best, col[i]);
                            }
                        }
                    }
                }
                d = best;
            }
            const bool ok = pvalid && d <= max_err;
            if (!FILL) {
                if (ok) {
                    if (d < lane_best) {
                        lane_best = d;
                        lane_cnt = 0;
                    }
                    if (d == lane_best) lane_cnt++;
                }
            } else {
                const bool hit = ok && d == target;
                const uint64_t m = __ballot(hit);
                if (hit) {
                    const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    hits[out_base + written + before] = (uint32_t)p;
                }
                written += (uint32_t)__popcll(m);
            }
        }
        if (!FILL) {
            const int wb = wave_min(lane_best);
            const uint32_t total = wave_sum(lane_best == wb ? lane_cnt : 0u);
            if (lane == 0) {
                part_best[w] = (uint32_t)wb;
                part_cnt[w] = total;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_scan_pk<FILL>: the Myers scan of k_scan for barcodes of at most 16 characters with TWO barcodes per lane, one in
// each 16-bit half of the 32-bit state words (128 barcodes per wave step).  Bitwise operations and the +-1 score
// updates act on both halves at once; only the addition of the carry chain needs the packed 16-bit add
// (v_pk_add_u16) and the running minimum the packed min; the shifted words are masked so that nothing crosses from
// the low half into the high one.  Same task decomposition, partial results and output order as k_scan.
// ------------------------------------------------------------------------------------------------------------
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2_t, a) + __builtin_bit_cast(us2_t, b));
}
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}

template <bool FILL>
__global__ void __launch_bounds__(256)
k_scan_pk(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, int64_t seg_bytes, const int32_t *__restrict__ list,
          int64_t n_list, int n_slices, const uint2 *__restrict__ pat, int np, int bl, int max_err,
          const uint8_t *__restrict__ min_e, uint32_t *__restrict__ part_best, uint32_t *__restrict__ part_cnt,
          const int64_t *__restrict__ hit_offs, uint32_t *__restrict__ hits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int tiles = (np + 63) / 64;
    const int tiles_per_slice = (tiles + n_slices - 1) / n_slices;
    const uint32_t one = 0x00010001u;
    for (int64_t w = wave; w < n_list * n_slices; w += n_waves) {
        const int64_t li = w / n_slices;
        const int slice = (int)(w - li * n_slices);
        const int64_t r = (int64_t)list[li];
        int64_t b0, len;
        sct_row_span(offs, r, seg_bytes, b0, len);
        const int p_begin = slice * tiles_per_slice * 64;
        const int p_end = min(np, p_begin + tiles_per_slice * 64);
        int target = 0;
        int64_t out_base = 0;
        if (FILL) {
            target = min_e[r];
            if (target == SCT_NO_HIT) continue;
            if (part_best[w] != (uint32_t)target) continue;  // a slice whose own minimum is worse holds no hit
            uint32_t mine = 0;
            for (int sl = lane; sl < slice; sl += 64)
                if (part_best[li * n_slices + sl] == (uint32_t)target) mine += part_cnt[li * n_slices + sl];
            out_base = hit_offs[r] + wave_sum(mine);
        }
        int lane_best = max_err + 1;
        uint32_t lane_cnt = 0;
        uint32_t written = 0;  // wave-uniform
        for (int p0 = p_begin; p0 < p_end; p0 += 128) {
            const int pa = p0 + lane, pb = p0 + 64 + lane;
            const bool va = pa < p_end, vb = pb < p_end;
            const uint2 qa = va ? pat[pa] : make_uint2(0, 0);
            const uint2 qb = vb ? pat[pb] : make_uint2(0, 0);
            const uint32_t LO = (qa.x & 0xFFFFu) | (qb.x << 16), HI = (qa.y & 0xFFFFu) | (qb.y << 16);
            uint32_t Pv = 0xFFFFFFFFu, Mv = 0u;
            uint32_t score = (uint32_t)bl * one, best = score;
            for (int64_t c0 = 0; c0 < len; c0 += 64) {
                const int64_t ci = c0 + lane;
                const uint32_t my = ci < len ? sct_code(segs[b0 + ci]) : 0u;
                const int ncol = (int)(len - c0 < 64 ? len - c0 : 64);
                for (int j = 0; j < ncol; j++) {
                    const uint32_t c = __builtin_amdgcn_readlane(my, j);
                    const uint32_t ml = 0u - (c & 1u), mh = 0u - (c >> 1);
                    const uint32_t Eq = ~(LO ^ ml) & ~(HI ^ mh);
                    const uint32_t Xv = Eq | Mv;
                    const uint32_t Xh = (pk_add16(Eq & Pv, Pv) ^ Pv) | Eq;
                    uint32_t Ph = Mv | ~(Xh | Pv);
                    uint32_t Mh = Pv & Xh;
                    score = score + ((Ph >> (bl - 1)) & one) - ((Mh >> (bl - 1)) & one);
                    Ph = (Ph << 1) & 0xFFFEFFFEu;
                    Mh = (Mh << 1) & 0xFFFEFFFEu;
                    Pv = Mh | ~(Xv | Ph);
                    Mv = Ph & Xv;
                    best = pk_min16(best, score);
                }
            }
            const int da = (int)(best & 0xFFFFu), db = (int)(best >> 16);
            const bool oka = va && da <= max_err, okb = vb && db <= max_err;
            if (!FILL) {
                if (oka) {
                    if (da < lane_best) {
                        lane_best = da;
                        lane_cnt = 0;
                    }
                    if (da == lane_best) lane_cnt++;
                }
                if (okb) {
                    if (db < lane_best) {
                        lane_best = db;
                        lane_cnt = 0;
                    }
                    if (db == lane_best) lane_cnt++;
                }
            } else {
                const bool hita = oka && da == target, hitb = okb && db == target;
                const uint64_t m1 = __ballot(hita), m2 = __ballot(hitb);
                const uint64_t below = (1ull << lane) - 1ull;
                if (hita) hits[out_base + written + (uint32_t)__popcll(m1 & below)] = (uint32_t)pa;
                written += (uint32_t)__popcll(m1);
                if (hitb) hits[out_base + written + (uint32_t)__popcll(m2 & below)] = (uint32_t)pb;
                written += (uint32_t)__popcll(m2);
            }
        }
        if (!FILL) {
            const int wb = wave_min(lane_best);
            const uint32_t total = wave_sum(lane_best == wb ? lane_cnt : 0u);
            if (lane == 0) {
                part_best[w] = (uint32_t)wb;
                part_cnt[w] = total;
            }
        }
    }
}

// fold the per-slice partial results of k_scan's count pass: one lane per listed segment
__global__ void __launch_bounds__(256)
k_scan_reduce(const int32_t *__restrict__ list, int64_t n_list, int n_slices, int max_err,
              const uint32_t *__restrict__ part_best, const uint32_t *__restrict__ part_cnt,
              uint8_t *__restrict__ min_e, uint32_t *__restrict__ cnt) {
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_list) return;
    uint32_t best = (uint32_t)max_err + 1u, total = 0;
    for (int sl = 0; sl < n_slices; sl++) {
        const uint32_t b = part_best[li * n_slices + sl], c = part_cnt[li * n_slices + sl];
        if (b < best) {
            best = b;
            total = 0;
        }
        if (b == best) total += c;
    }
    const int64_t r = list[li];
    if (best <= (uint32_t)max_err) {
        min_e[r] = (uint8_t)best;
        cnt[r] = total;
    } else {
        min_e[r] = SCT_NO_HIT;
        cnt[r] = 0;
    }
}


// ------------------------------------------------------------------------------------------------------------
// k_route_long: rows longer than PLANE_MAX characters, which the filter kernel only listed (list_long): one wave per
// row decides between the exhaustive Myers scan and, when the reference's prefix length k > 0 and the row holds a byte
// that is not one of ACGT, the restricted-start DP (scTagger.py:681).  Runs on the slow path only.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_route_long(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, int64_t seg_bytes, const int32_t *__restrict__ list_long,
             int64_t n_long, int bl, int max_err, int prefix_len, int32_t *__restrict__ list_scan, int32_t *__restrict__ list_dp, uint32_t *__restrict__ ctr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t li = wave; li < n_long; li += n_waves) {
        const int64_t r = list_long[li];
        bool bad = false;
        {
            int64_t b0, len;
            sct_row_span(offs, r, seg_bytes, b0, len);
            for (int64_t i = lane; i < len; i += 64) {
                const uint8_t b = segs[b0 + i];
                bad |= prefix_len > 0 && !sct_is_acgt(b);
                // (a row of more than PLANE_MAX characters: the reference raises on a byte >= 128 in any window it inserts)
                if (b >= 128 && sct_byte_is_read(segs + b0, len, i, bl, max_err, prefix_len)) atomicMin(reinterpret_cast<unsigned long long *>(ctr + CTR_BAD_BYTE), (unsigned long long)(b0 + i));
            }
        }
        const bool dp = __ballot(bad) != 0;
        if (lane == 0) {
            if (dp) list_dp[atomicAdd(&ctr[CTR_DP], 1u)] = (int32_t)r;
            else list_scan[atomicAdd(&ctr[CTR_SCAN], 1u)] = (int32_t)r;
        }
    }
}
