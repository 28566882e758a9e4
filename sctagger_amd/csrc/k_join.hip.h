// k_join.hip.h — the thirds scheme at max_err = 1 as a PARTITIONED JOIN (very large barcode sets: BASELINE configs[4], 3.6 M barcodes).
// Part of libsctagger_hip.so (translation unit join_e1.hip).  gfx950 only.
//
// Why.  With 7.2 M patterns the seed tables (fixed buckets, sct_tables.h) are 160 MB; the filter kernel (k_filter.hip.h) makes ~26
// lookups per row, each a random 16- or 32-byte read, and the memory system serves 5.5 x 10^10 random lines per second beyond the L2
// whatever the table's size (tools/gather_probe.hip, profiles/r03_gather_probe.jsonl): 5.0 ms per 10 M rows, 27 missed lines per row,
// nothing the kernel itself can improve.  From a table slice of 2 MB the same reads come at 2.6 x 10^11/s (L2).  So the lookups are
// turned into a join: every lookup becomes an 8-byte TUPLE (row, low key bits, the few text characters the verification needs),
// tuples are written partition by partition — a partition = the keys whose buckets lie in one 2 MB slice of a family's table —, and
// probed partition by partition, every workgroup on the chip inside the same slice.  A row's candidates meet again in a small
// per-row list (global atomics, one per verified hit) that the last kernel folds exactly as the filter kernel folds its LDS lists.
//
//   k_join_part0   rows -> bit-planes (kept for tier 1), routing as in the filter kernel, T01 tuples        (reads the rows once)
//   k_join_probe   T01 partitions: bucket (L2) -> compare (sct_join_distance, sct_core.h) -> per-row hit list
//   k_join_part1   rows without a distance-0 hit -> T12 and T02 tuples                                         (tiering, as the filter)
//   k_join_probe   T12 / T02 partitions
//   k_join_finish  per-row list -> pattern ids, minimum, distinct, sorted; stored exactly as the filter's main launch stores them
// Rows of 32 .. 63 characters (whole waves of them: the filter kernel's own rule), rows the retry launch must redo and everything
// else stay with k_filter: the join replaces its 32-bit main launch only.  Verification needs no automaton at one edit: the two
// seed pieces are exact, the remaining 5 or 6 pattern characters are compared with the text beside the anchors by mismatch masks
// (host-checked against the anchored automaton for every placement, tests/host_core_check.cpp).
//
// Measured (cfg5, 10 M rows, profiles/r03join_*): the lookups do hit the L2 — 11.3 GB of HBM traffic per batch against the filter
// kernel's 17.5, 12.4 missed lines per row against 27.1 — and the six kernels take 4.76 ms against the filter kernel's 4.85: what
// the memory system no longer costs, packing, partitioning, ranking and the per-row atomics do (165 VALU wave-instructions per row
// against 197).  So this path is an OPTION (sct_set_option "join" = 1), not the default; DESIGN.md 6.3 has the table.
#pragma once

static constexpr int JOIN_P0 = 16;           // T01: 2^20 keys x 32 B = 32 MB -> 16 slices of 2 MB (key >> 16)
static constexpr int JOIN_P1 = 64;           // T12 and T02: 2^22 keys x 16 B = 64 MB each -> 32 slices each (key >> 17)
static constexpr int JOIN_CAP0 = 512;        // tuples a producer block (256 rows) may leave in one partition: T01 (mean 122 at 22 characters: 8 placements a row)
static constexpr int JOIN_CAP1 = 384;        // ... T12 / T02 (mean 42 / 105; T02 tuples cluster — see join_part_t02 — worst seen 260 with every row active)
static constexpr int JOIN_GROUP = 16;        // producer blocks per probe block
static constexpr int64_t JOIN_SUB_ROWS = (int64_t)1 << 22;   // rows per pass (a multiple of 256): the tuple regions hold one pass
static constexpr int JOIN_HITS = 8;          // entries of a row's list (more: the row goes to the retry launch)
static constexpr int JOIN_STAGE_BYTES = 256 * 31 + 384;

template <typename OffT>
struct JoinArgs {
    FilterArgs<OffT> fa;          // rows, tables, result arrays, routing lists, counters: as the filter's main launch gets them
    int64_t row0, row1;           // the rows of this pass (the tuple buffers hold JOIN_SUB_ROWS rows' worth)
    uint2 *planes;                // per row: the bit-planes (0 for rows the join does not handle)
    uint8_t *jlen;                // per row: its length if the join handles it, else 0
    uint8_t *wave_kind;           // per 64 rows: 1 = the join handles this wave (k_join_finish stores it), 0 = k_filter's 64-bit launch does
    uint32_t *jcnt;               // per row: hits handed out (bits 0..15) | distance-0 hits (bits 16..30)
    uint32_t *jslots;             // per row: JOIN_HITS entries, distance << 30 | locator
    unsigned long long *tup0, *tup1;
    uint32_t *cnt0, *cnt1;        // tuples per (producer block, partition)
};

// T02's partition of a key.  The key's top five bits are the hi-plane bits of piece B's characters 1..5: the (up to three) placements
// of a row that share a position of piece B (t2 = t + 10 + d for d = -1, 0, 1) would all fall into one partition, and a producer
// block's partition fills would vary three times as much.  So the top bits are mixed with key bits 3..7 (piece A's side); bits 0..2
// stay out of it: the 8 buckets of a 128-byte line belong to one partition, whose footprint in the table stays 2 MB.
__device__ __forceinline__ uint32_t join_part_t02(uint32_t key) { return ((key >> 17) ^ (key >> 3)) & 31u; }
__device__ __forceinline__ uint32_t join_key_t02(uint32_t q, uint32_t keylow) { return (((q ^ (keylow >> 3)) & 31u) << 17) | keylow; }

// tuple: bits 0..29 row | bits 30..46 low key bits | bits 47..63 context (sct_join_ctx_*)
__device__ __forceinline__ unsigned long long join_tuple(int64_t row, uint32_t keylow, uint32_t ctx) {
    return (unsigned long long)(uint32_t)row | ((unsigned long long)keylow << 30) | ((unsigned long long)ctx << 47);
}

// ------------------------------------------------------------------------------------------------------------
// pass 0: decode + route (the prologue of filter_rows, E = 1, 32-bit masks) + T01 tuples
// ------------------------------------------------------------------------------------------------------------
template <typename OffT>
__global__ void __launch_bounds__(256) k_join_part0(const JoinArgs<OffT> ja) {
    const FilterArgs<OffT> &a = ja.fa;
    constexpr int E = 1, bl = 16;
    __shared__ uint32_t stage[JOIN_STAGE_BYTES / 4];
    __shared__ uint32_t pcnt[JOIN_P0];
    __shared__ uint32_t block_flags;
    const int lane = threadIdx.x & 63, widx = threadIdx.x >> 6;
    const int64_t n = ja.row1;
    const int64_t first = ja.row0 + ((int64_t)blockIdx.x * 4 + widx) * 64;
    const int64_t gid = first + lane;
    const bool inrange = gid < n;
    const int64_t r = gid;
    int64_t start = 0, len64 = 0;
    if (inrange) sct_row_span(a.offs, r, a.seg_bytes, start, len64);
    const int len_all = (int)(len64 < 0x7FFFFFFF ? len64 : 0x7FFFFFFF);
    const bool fits = inrange && len_all >= bl - E && len_all <= PLANE_MAX;
    int len = fits ? len_all : 0;
    const int maxlen = wave_max_i(len);
    const bool mykind = maxlen <= 31;
    const bool anyrow = __ballot(inrange) != 0;
    if (threadIdx.x < JOIN_P0) pcnt[threadIdx.x] = 0;
    if (lane == 0) reinterpret_cast<uint8_t *>(&block_flags)[widx] = !anyrow ? 2 : (mykind ? 1 : 0);
    __syncthreads();
    uint32_t lo = 0, hi = 0, badmask = 0;
    bool decoded = false;
    {
        const uint32_t f4 = block_flags;
        if ((f4 & 0xFFu) != 0u && (f4 & 0xFF00u) != 0u && (f4 & 0xFF0000u) != 0u && (f4 & 0xFF000000u) != 0u) {
            const int64_t bf = first - 64 * widx;
            const int64_t be = bf + 256 < n ? bf + 256 : n;
            int64_t A = (int64_t)a.offs[bf], B = (int64_t)a.offs[be];
            A = A < 0 ? 0 : (A > a.seg_bytes ? a.seg_bytes : A);
            B = B < A ? A : (B > a.seg_bytes ? a.seg_bytes : B);
            if (B > A && B - A + 256 <= (int64_t)JOIN_STAGE_BYTES) {   // block-uniform
                sct_block_stage_load(a.segs, A, B, stage);
                __syncthreads();
                const int dl = (start >= A && start + len <= B) ? len : 0;
                sct_decode_row_from_block_stage<uint32_t>(a.segs, A, start, dl, stage, lo, hi, badmask);
                decoded = true;
            }
        }
    }
    const bool mine = mykind && anyrow;      // this wave's rows are the join's
    if (mine && !decoded) sct_decode_row_mask<uint32_t>(a.segs, start, len, lo, hi, badmask);
    const bool nonacgt = badmask != 0;
    if (lane == 0 && first < n) ja.wave_kind[first >> 6] = mine ? 1 : 0;
    bool active = false;
    if (mine) {
        if (__ballot(nonacgt) != 0) {
            if (nonacgt) sct_report_high_bytes(a.segs + start, len, (uint64_t)badmask, start, bl, E, a.prefix_len, a.ctr);
        }
        active = fits;
        const bool line = first + 64 <= a.n && (((uintptr_t)(a.min_e + first)) & 15) == 0 && (a.cnt8 == nullptr || (((uintptr_t)(a.cnt8 + first)) & 15) == 0);
        if (inrange) {
            // routing: as the filter's main launch does it (k_join_finish's store covers the wave's min_e / cnt8 lines)
            if (!line && a.cnt8 && (!fits || (nonacgt && a.prefix_len > 0))) a.cnt8[r] = 0;
            if (!fits) {
                if (len_all > PLANE_MAX) {
                    const uint32_t k = atomicAdd(&a.ctr[CTR_SPECIAL], 1u);
                    a.list_special[k] = (int32_t)r;
                    if (a.rec_out != nullptr && k < a.rec_cap) a.rec_out[k] = SctSpecialRec{0, 0, 0, SCT_REC_NONE};
                    a.cnt[r] = CNT_DEFER;
                } else {
                    if (!line) a.min_e[r] = SCT_NO_HIT;
                    a.cnt[r] = 0;
                }
            } else if (nonacgt && a.prefix_len > 0) {
                const uint32_t k = atomicAdd(&a.ctr[CTR_SPECIAL], 1u);
                a.list_special[k] = (int32_t)r;
                if (a.rec_out != nullptr && k < a.rec_cap) a.rec_out[k] = SctSpecialRec{(uint64_t)lo, (uint64_t)hi, (uint64_t)badmask, (uint64_t)len};
                a.cnt[r] = CNT_DEFER;
                active = false;
            }
        }
    }
    if (inrange) {   // (also for the waves k_filter's 64-bit launch owns: pass 1 and the finish must find zeros there)
        ja.planes[r] = active ? make_uint2(lo, hi) : make_uint2(0u, 0u);
        ja.jlen[r] = active ? (uint8_t)len : (uint8_t)0;
        ja.jcnt[r] = 0u;
    }
    if (!active) len = 0;
    // T01 tuples: exact 10-mer at t, the 6 characters behind it within one edit (placements pruned as sct_placement_need prunes them:
    // the last five of those characters must exist: t + 15 <= len)
    const int pmax = wave_max_i(mine ? len : 0) - 15;
    const int64_t prod = (int64_t)blockIdx.x;      // producer block index within this pass
    for (int t = 0; t <= pmax; t++) {
        if (len > 0 && t + 15 <= len) {
            const SctFamily<SCT_F_T01> fam{};
            const uint32_t key = sct_key_of<uint32_t>(fam, lo, hi, t, t + fam.GAP);
            const uint32_t q = key >> 16;
            const uint32_t pos = atomicAdd(&pcnt[q], 1u);
            if (pos < (uint32_t)JOIN_CAP0)
                ja.tup0[((size_t)prod * JOIN_P0 + q) * JOIN_CAP0 + pos] = join_tuple(r, key & 0xFFFFu, sct_join_ctx_t01(lo, hi, len, t));
            else
                atomicOr(&a.ctr[CTR_JOIN_OVERFLOW], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < JOIN_P0) ja.cnt0[(size_t)prod * JOIN_P0 + threadIdx.x] = pcnt[threadIdx.x] < (uint32_t)JOIN_CAP0 ? pcnt[threadIdx.x] : (uint32_t)JOIN_CAP0;
}

// ------------------------------------------------------------------------------------------------------------
// pass 1: rows without a distance-0 hit -> T12 and T02 tuples (partitions 0..31: T12, 32..63: T02)
// ------------------------------------------------------------------------------------------------------------
template <typename OffT>
__global__ void __launch_bounds__(256) k_join_part1(const JoinArgs<OffT> ja) {
    const FilterArgs<OffT> &a = ja.fa;
    constexpr int E = 1;
    __shared__ uint32_t pcnt[JOIN_P1];
    const int64_t r = ja.row0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < JOIN_P1) pcnt[threadIdx.x] = 0;
    __syncthreads();
    int len = 0;
    uint32_t lo = 0, hi = 0;
    if (r < ja.row1) {
        len = (int)ja.jlen[r];
        if (len > 0 && (ja.jcnt[r] >> 16) != 0u) len = 0;     // a barcode at distance 0 is known: only the minimum is reported
        if (len > 0) {
            const uint2 p = ja.planes[r];
            lo = p.x;
            hi = p.y;
        }
    }
    const int64_t prod = (int64_t)blockIdx.x;
    const int maxlen = wave_max_i(len);
    auto put = [&](uint32_t q, uint32_t keylow, uint32_t ctx) {
        const uint32_t pos = atomicAdd(&pcnt[q], 1u);
        if (pos < (uint32_t)JOIN_CAP1) ja.tup1[((size_t)prod * JOIN_P1 + q) * JOIN_CAP1 + pos] = join_tuple(r, keylow, ctx);
        else atomicOr(&a.ctr[CTR_JOIN_OVERFLOW], q < 32u ? 2u : 4u);
    };
    {   // T12: exact 11-mer (pattern characters 5..15) at t, the 5 characters in front of it within one edit
        const SctFamily<SCT_F_T12> fam{};
        for (int t = 0; t + fam.LA + fam.LB <= maxlen; t++) {
            const int t2 = t + fam.GAP;
            if (len > 0 && sct_placement_need(fam, E, t, t2) <= len) {
                const uint32_t key = sct_key_of<uint32_t>(fam, lo, hi, t, t2);
                put(key >> 17, key & 0x1FFFFu, sct_join_ctx_t12(lo, hi, t));
            }
        }
    }
    {   // T02: exact 5-mer at t and exact 6-mer at t2 = t + 10 + d, the 5 + d characters between them within one edit
        const SctFamily<SCT_F_T02> fam{};
#pragma unroll
        for (int d = -1; d <= 1; d++) {
            for (int t = 0; t + fam.GAP + d + fam.LB <= maxlen; t++) {
                const int t2 = t + fam.GAP + d;
                if (len > 0 && sct_placement_need(fam, E, t, t2) <= len) {
                    const uint32_t key = sct_key_of<uint32_t>(fam, lo, hi, t, t2);
                    put(32u + join_part_t02(key), key & 0x1FFFFu, sct_join_ctx_t02(lo, hi, t, t2));
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < JOIN_P1) ja.cnt1[(size_t)prod * JOIN_P1 + threadIdx.x] = pcnt[threadIdx.x] < (uint32_t)JOIN_CAP1 ? pcnt[threadIdx.x] : (uint32_t)JOIN_CAP1;
}

// ------------------------------------------------------------------------------------------------------------
// probe: one workgroup = one partition x JOIN_GROUP producer blocks; a wave takes a producer block at a time, a lane a tuple
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void join_hit(uint32_t *__restrict__ jcnt, uint32_t *__restrict__ jslots, uint32_t row, uint32_t d, uint32_t loc) {
    const uint32_t v = atomicAdd(&jcnt[row], 1u + (d == 0u ? 0x10000u : 0u));
    const uint32_t k = v & 0xFFFFu;
    if (k < (uint32_t)JOIN_HITS) jslots[(size_t)row * JOIN_HITS + k] = (d << 30) | loc;
}

// FAM: SCT_F_T01 (partitions of tup0 / cnt0, P = JOIN_P0, q0 = 0), SCT_F_T12 (tup1, P = JOIN_P1, q0 = 0), SCT_F_T02 (tup1, q0 = 32)
template <int FAM, typename OffT>
__global__ void __launch_bounds__(256) k_join_probe(const JoinArgs<OffT> ja, int n_prod, int groups) {
    const FilterArgs<OffT> &a = ja.fa;
    constexpr bool T01 = FAM == SCT_F_T01;
    constexpr int P = T01 ? JOIN_P0 : JOIN_P1, CAP = T01 ? JOIN_CAP0 : JOIN_CAP1, Q0 = FAM == SCT_F_T02 ? 32 : 0;
    constexpr int KSHIFT = T01 ? 16 : 17, CAPE = T01 ? 16 : 8, NREST = T01 ? 6 : 5;
    const SctFamily<FAM> fam{};
    const unsigned long long *tup = T01 ? ja.tup0 : ja.tup1;
    const uint32_t *cnt = T01 ? ja.cnt0 : ja.cnt1;
    const int lane = threadIdx.x & 63, widx = threadIdx.x >> 6;
    const int ql = (int)(blockIdx.x / (unsigned)groups), g = (int)(blockIdx.x % (unsigned)groups);
    const int q = Q0 + ql;
    const uint32_t *off = a.tab_off + a.ft.off_base[FAM];
    const uint4 *fb = reinterpret_cast<const uint4 *>(a.tab_fb + a.ft.fb_base[FAM - SCT_F_T01]);
    __shared__ uint32_t pairs_all[4][CAPE * 64];      // per wave: entry (12 bits) | owning lane << 16 | place in the bucket << 22
    __shared__ uint32_t rows_all[4][64], keys_all[4][64];
    uint32_t *pairs = pairs_all[widx], *rows_l = rows_all[widx], *keys_l = keys_all[widx];
    // The wave's regions (producer blocks g * JOIN_GROUP + widx + 4 i, i = 0..3) are read as one run of tuples, 64 at a time across
    // the region ends (42 tuples a region for T12: one region at a time left a third of the lanes without a tuple), and the next 64
    // are requested before this round's buckets are waited for.
    static_assert(JOIN_GROUP == 16, "four regions per wave");
    uint32_t ends[4];
    {
        uint32_t at = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int b = g * JOIN_GROUP + widx + 4 * i;
            at += b < n_prod ? cnt[(size_t)b * P + q] : 0u;
            ends[i] = at;
        }
    }
    const uint32_t c = ends[3];
    auto tuple_at = [&](uint32_t idx) -> unsigned long long {
        if (idx >= c) return 0ull;
        const int i = (int)(idx >= ends[0]) + (int)(idx >= ends[1]) + (int)(idx >= ends[2]);
        const uint32_t within = idx - (i == 0 ? 0u : (i == 1 ? ends[0] : (i == 2 ? ends[1] : ends[2])));
        const int b = g * JOIN_GROUP + widx + 4 * i;
        return __builtin_nontemporal_load(tup + ((size_t)b * P + q) * CAP + within);
    };
    {
        unsigned long long tp_next = tuple_at((uint32_t)lane);
        for (uint32_t i0 = 0; i0 < c; i0 += 64) {   // wave-uniform
            const bool have = i0 + (uint32_t)lane < c;
            const unsigned long long tp = tp_next;
            const uint32_t row = (uint32_t)tp & 0x3FFFFFFFu;
            const uint32_t keylow = (uint32_t)(tp >> 30) & ((1u << KSHIFT) - 1u);
            const uint32_t key = FAM == SCT_F_T02 ? join_key_t02((uint32_t)ql, keylow) : (((uint32_t)ql << KSHIFT) | keylow);
            const uint32_t ctx = (uint32_t)(tp >> 47);
            uint32_t w[CAPE / 2];
#pragma unroll
            for (int k = 0; k < CAPE / 2; k++) w[k] = 0u;
            if (have) {   // 16 bytes per request: the second half of a T01 bucket only where the first is full (27 % of them at 3.6 M barcodes)
                const uint4 v = fb[(size_t)key * (CAPE / 8)];
                tp_next = tuple_at(i0 + 64u + (uint32_t)lane);   // (issued behind the bucket's load: loads return in order)
                w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
                if (CAPE == 16 && (v.w & (SCT_FB_VALID << 16)) != 0u) {
                    const uint4 v2 = fb[(size_t)key * (CAPE / 8) + 1];
                    w[CAPE / 2 - 4] = v2.x, w[CAPE / 2 - 3] = v2.y, w[CAPE / 2 - 2] = v2.z, w[CAPE / 2 - 1] = v2.w;
                }
            } else {
                tp_next = 0ull;   // (no tuple this round: none in the next)
            }
            const bool over = have && (w[0] & SCT_FB_OVERFLOW) != 0u;
            // A bucket is filled from its first place on: its entries are the valid bits.  The wave's (tuple, entry) pairs are laid
            // out one after the other in LDS and compared 64 at a time, whatever the spread of the bucket lengths (mean 6.9 of 16
            // for T01 and 1.7 of 8 for T12 / T02 at 3.6 M barcodes: a loop to the longest bucket of the wave left half / three
            // quarters of the lanes idle).
            uint32_t nv = 0;
            if (have && !over) {
#pragma unroll
                for (int k = 0; k < CAPE / 2; k++) nv += (uint32_t)__popc(w[k] & (SCT_FB_VALID | (SCT_FB_VALID << 16)));
            }
            uint32_t inc = nv;
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, sft);
                if (lane >= sft) inc += up;
            }
            const uint32_t pre = inc - nv;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            __builtin_amdgcn_wave_barrier();   // (the previous round's reads of this wave's LDS are issued: LDS runs in issue order)
            rows_l[lane] = row;
            keys_l[lane] = key;
#pragma unroll
            for (int j = 0; j < CAPE; j++) {
                if (__ballot((uint32_t)j < nv) == 0) break;
                if ((uint32_t)j < nv) pairs[pre + (uint32_t)j] = ((w[j >> 1] >> (16 * (j & 1))) & 0x0FFFu) | ((uint32_t)lane << 16) | ((uint32_t)j << 22);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k0 = 0; k0 < total; k0 += 64) {   // wave-uniform: every lane takes part in the shuffle
                const uint32_t pi = k0 + (uint32_t)lane;
                const bool live = pi < total;
                const uint32_t pr = pairs[live ? pi : 0u];
                const uint32_t owner = (pr >> 16) & 63u;
                const uint32_t c = (uint32_t)__shfl((int)ctx, (int)owner);
                // context -> (characters that exist, fixed end)
                const int m = FAM == SCT_F_T02 ? NREST + (int)(c >> 14) - 1 : (int)(c >> 14);
                const uint32_t rest = FAM == SCT_F_T12 ? sct_join_rest_t12(pr & 0x03FFu) : (pr & 0x0FFFu);
                const uint32_t d = sct_join_distance(rest, c, NREST, m, FAM == SCT_F_T02);
                if (live && d <= 1u)
                    join_hit(ja.jcnt, ja.jslots, rows_l[owner], d, LOC_INLINE | ((a.ft.off_base[FAM] + keys_l[owner]) << 5) | (pr >> 22));
            }
            const int m = FAM == SCT_F_T02 ? NREST + (int)(ctx >> 14) - 1 : (int)(ctx >> 14);
            if (__ballot(over) != 0 && over) {
                // more entries than a fixed bucket holds (0.03 % of the keys at 3.6 M barcodes): through the offsets, entry by entry
                const uint32_t s = off[key], en = off[key + 1];
                for (uint32_t gk = s; gk < en; gk++) {
                    const uint32_t r0 = sct_rest_of(fam, a.tab_code[gk]);
                    const uint32_t rest = FAM == SCT_F_T12 ? sct_join_rest_t12(r0 & 0x03FFu) : (r0 & 0x0FFFu);
                    const uint32_t d = sct_join_distance(rest, ctx, NREST, m, FAM == SCT_F_T02);
                    if (d <= 1u) join_hit(ja.jcnt, ja.jslots, row, d, gk);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// finish: a row's list -> minimum + distinct pattern ids, stored as the filter's main launch stores its rows
// ------------------------------------------------------------------------------------------------------------
template <typename OffT>
__global__ void __launch_bounds__(256) k_join_finish(const JoinArgs<OffT> ja) {
    const FilterArgs<OffT> &a = ja.fa;
    constexpr int E = 1;
    const int lane = threadIdx.x & 63;
    const int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (first >= a.n) return;
    if (ja.wave_kind[first >> 6] != 1) return;        // k_filter's 64-bit launch owns this wave
    const int64_t r = first + lane;
    const bool inrange = r < a.n;
    const bool active = inrange && ja.jlen[r] > 0;
    const uint32_t v = active ? ja.jcnt[r] : 0u;
    const uint32_t handed = v & 0xFFFFu;
    const uint32_t c = handed < (uint32_t)JOIN_HITS ? handed : (uint32_t)JOIN_HITS;
    uint32_t ent[JOIN_HITS];
    {
        const uint4 *p = reinterpret_cast<const uint4 *>(ja.jslots + (size_t)(inrange ? r : 0) * JOIN_HITS);
        const uint4 x = active ? p[0] : make_uint4(0, 0, 0, 0), y = active ? p[1] : make_uint4(0, 0, 0, 0);
        ent[0] = x.x; ent[1] = x.y; ent[2] = x.z; ent[3] = x.w;
        ent[4] = y.x; ent[5] = y.y; ent[6] = y.z; ent[7] = y.w;
    }
    // the minimum first: only entries at the minimum cost a table read
    uint32_t best = E + 1;
#pragma unroll
    for (int i = 0; i < JOIN_HITS; i++)
        if ((uint32_t)i < c) best = min(best, ent[i] >> 30);
    // locators -> pattern ids: all of a row's offset reads first, then all its id reads (two rounds of independent loads per lane
    // instead of a chain of up to sixteen)
    uint32_t idx[JOIN_HITS], pid[JOIN_HITS];
    bool use[JOIN_HITS];
#pragma unroll
    for (int i = 0; i < JOIN_HITS; i++) {
        use[i] = (uint32_t)i < c && (ent[i] >> 30) == best;
        const uint32_t loc = ent[i] & LOC_MASK;
        idx[i] = loc;
        if (use[i] && (loc & LOC_INLINE)) {
            const uint32_t l = loc & (LOC_INLINE - 1u);
            idx[i] = a.tab_off[l >> 5] + (l & 31u);
        }
    }
#pragma unroll
    for (int i = 0; i < JOIN_HITS; i++) pid[i] = use[i] ? (a.ft.pid16 ? (uint32_t)a.tab_ent16[idx[i]] : a.tab_ent[idx[i]]) : 0u;
    LaneResult<E, K_SLOTS> res;
    res.init();
#pragma unroll
    for (int i = 0; i < JOIN_HITS; i++)
        if (use[i]) res.add((int)best, pid[i]);
    const bool line = first + 64 <= a.n && (((uintptr_t)(a.min_e + first)) & 15) == 0 && (a.cnt8 == nullptr || (((uintptr_t)(a.cnt8 + first)) & 15) == 0);
    // more hits than the list holds: entries at the minimum may be missing -> the retry launch does the row from scratch
    unsigned long long arena_off;
    uint64_t arena_end;
    unsigned long long *cursor = sct_slot_arena(a, arena_off, arena_end);
    res.store(active, r, handed > (uint32_t)JOIN_HITS, line, a.min_e, a.cnt8, a.cnt, a.slots, arena_end, a.wave_base, first, a.list_defer, a.defer_count,
              cursor, arena_off);
}
