// k_filter.hip.h — the seed filter: LaneResult, k_filter (wave-cooperative, bucket slots; reads the ASCII rows itself).
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// Per-lane result of the filter kernels: running minimum distance and up to K_SLOTS distinct pattern ids at
// that minimum, in registers (static indexing only).  More than K_SLOTS ties defer the segment to the scan path.
// ------------------------------------------------------------------------------------------------------------
// The wave's 64 per-row bytes (min_e or the one-byte hit count of rows r0 .. r0 + 63, `mine` = this lane's byte) as four 16-byte
// stores instead of 64 one-byte ones: when the array is the caller's page-locked one every store instruction crosses the PCIe
// link as it is, and one-byte writes cost 0.8 ms per 10 M rows.  &min_e[r0] must be 16-byte aligned.
__device__ __forceinline__ void store_byte_line(uint8_t *__restrict__ min_e, int64_t r0, uint32_t mine) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = mine | (__shfl_down(mine, 1, 64) << 8) | (__shfl_down(mine, 2, 64) << 16) | (__shfl_down(mine, 3, 64) << 24);
    const uint32_t w1 = __shfl_down(w, 4, 64), w2 = __shfl_down(w, 8, 64), w3 = __shfl_down(w, 12, 64);
    if ((lane & 15) == 0) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u4{w, w1, w2, w3}, reinterpret_cast<u4 *>(min_e + r0 + lane));  // written once, read by nobody here: past the L2's tables
    }
}

template <int E, int KS = K_SLOTS>
struct LaneResult {
    static constexpr int K_SLOTS = KS;  // shadows the global default inside this struct
    int best = E + 1;
    int nh = 0;
    uint32_t h[KS];

    __device__ __forceinline__ void init() {
        best = E + 1;
        nh = 0;
#pragma unroll
        for (int i = 0; i < K_SLOTS; i++) h[i] = 0;
    }
    __device__ __forceinline__ void add(int d, uint32_t pid) {
        if (d > E) return;
        if (d < best) {
            best = d;
            nh = 0;
        }
        if (d != best) return;
        bool dup = false;
#pragma unroll
        for (int i = 0; i < K_SLOTS; i++) dup |= (i < nh) && (h[i] == pid);
        if (dup) return;
#pragma unroll
        for (int i = 0; i < K_SLOTS; i++)
            if (i == nh) h[i] = pid;
        nh++;  // may run past K_SLOTS: that is the defer signal
    }
    // Wave-cooperative store (every lane of the wave must call it; `row` says whether the lane owns a row).
    // The hit lists of the wave are written back to back into the dense `slots` array: one wave-wide prefix sum
    // of the list lengths and one 64-bit atomic per wave reserve the space, wave_base[first / 64] remembers where the
    // wave's lists start (see below).  Rows with too many ties are pushed to the retry list instead.
    // `slots_cap` = entries the slot array holds: a wave whose lists would run past it defers all its rows (the main
    // launch's array holds K_SLOTS entries for every row and cannot overflow; the retry launch's is sized for the usual case).
    // `line`: the wave's rows are 64 consecutive ones starting at r - lane and their min_e bytes are written as one line
    // (store_byte_line; lanes without a result of their own contribute SCT_NO_HIT)
    // cnt8 != NULL: the caller's one-byte counts are written here as well (a list never holds more than K_SLOTS entries; rows
    // without a result of their own get 0 until the launch or the slow path that owns them writes theirs)
    // The wave's lists lie back to back from `base` in row order, so ONE 8-byte word per wave (wave_base[first / 64]) tells
    // k_fill where they are: row r's list starts `sum of the counts of the wave's earlier rows` behind it (round 1 wrote an
    // 8-byte offset per row: 80 MB per 10 M rows out and in again).
    __device__ __forceinline__ void store(bool row, int64_t r, bool overflow, bool line, uint8_t *__restrict__ min_e, uint8_t *__restrict__ cnt8,
                                          uint32_t *__restrict__ cnt, uint32_t *__restrict__ slots, uint64_t slots_cap,
                                          uint64_t *__restrict__ wave_base, int64_t first, int32_t *__restrict__ list_defer,
                                          uint32_t *__restrict__ defer_count, unsigned long long *__restrict__ slot_cursor,
                                          unsigned long long base_off = 0) {
        const int lane = threadIdx.x & 63;
        bool defer = row && (overflow || nh > K_SLOTS);
        const uint32_t mine = (row && !defer) ? (uint32_t)nh : 0u;
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const uint32_t total = __shfl(inc, 63, 64);
        unsigned long long base = 0;
        // (base_off: the slot array may be cut into arenas with a cursor each — sct_slot_arena — so that 150 000 waves do not queue
        // at ONE atomic: the join's finish kernel spent 1.9 of its 2.0 ms there; slots_cap is then the END of the wave's arena)
        if (lane == 0 && total) base = base_off + atomicAdd(slot_cursor, (unsigned long long)total);
        base = __shfl(base, 0, 64);
        if (base + total > slots_cap) defer = row;  // wave-uniform: no room for this wave's lists
        if (lane == 0) __builtin_nontemporal_store((uint64_t)base, &wave_base[first >> 6]);
        if (line) {
            store_byte_line(min_e, r - lane, (row && !defer && mine) ? (uint32_t)best : (uint32_t)SCT_NO_HIT);
            if (cnt8) store_byte_line(cnt8, r - lane, (row && !defer) ? mine : 0u);
        } else if (cnt8 && row && !defer) {
            cnt8[r] = (uint8_t)mine;
        }
        if (!row) return;
        if (defer) {
            const uint32_t k = atomicAdd(defer_count, 1u);
            list_defer[k] = (int32_t)r;
            cnt[r] = CNT_DEFER;
            return;
        }
        __builtin_nontemporal_store(mine, &cnt[r]);
        if (mine == 0) {
            if (!line) min_e[r] = SCT_NO_HIT;
            return;
        }
        if (nh > 1) {  // sort ascending: odd-even transposition network on registers
#pragma unroll
            for (int i = 0; i < K_SLOTS; i++)
                if (i >= nh) h[i] = 0xFFFFFFFFu;
#pragma unroll
            for (int round = 0; round < K_SLOTS; round++) {
#pragma unroll
                for (int i = round & 1; i + 1 < K_SLOTS; i += 2) {
                    const uint32_t a = h[i], b = h[i + 1];
                    h[i] = a < b ? a : b;
                    h[i + 1] = a < b ? b : a;
                }
            }
        }
        if (!line) min_e[r] = (uint8_t)best;
        const uint64_t start = base + inc - mine;
#pragma unroll
        for (int i = 0; i < K_SLOTS; i++)
            if (i < nh) __builtin_nontemporal_store(h[i], &slots[start + i]);
    }
};

// ------------------------------------------------------------------------------------------------------------
// k_filter<E>: one WAVE per 64 segments, cooperating through LDS so that verification always runs on
// full waves.  The seed families are processed one after the other; within a family
//   lookup   every lane computes the seed key of its own segment for the current placement (t, t2) — the
//            placements are walked wave-uniformly up to the longest segment of the wave — and reads its bucket
//            range; the loads are issued one placement ahead of the queue work that consumes them;
//   push     round by round, every lane that still has a candidate appends (table entry, lane, t, t2) to the
//            wave's queue in LDS (position = tail + mbcnt(ballot));
//   drain    whenever 64 candidates are queued, lane i takes entry head+i: reads the barcode's 2-bit codes from
//            the table (L2), runs the ANCHORED Wu-Manber automaton of this family (8 of the 16 barcode
//            characters, sct_core.h) against the owning segment's match masks in LDS, and on d <= E appends
//            (d, pattern id) to that segment's list in LDS (LDS atomic counter);
//   finish   (after the last family) every lane folds its own list into (minimum, distinct pattern ids), sorts
//            and stores.
// A segment with more than HMAX verified matches AT ITS MINIMUM (duplicates included; insert_hit) or more than K_SLOTS ties is deferred.
// All cross-lane traffic is inside one wave: LDS operations of a wave execute in issue order, so no barrier is
// needed; __builtin_amdgcn_wave_barrier() only pins the compiler's ordering.
// ------------------------------------------------------------------------------------------------------------
// queue capacity (entries): >= 127 left over + what one pass appends — 64 lanes x 3 inline slot entries, or one entry per
// lane.  With slots and 32-bit masks the entries are one word each and the ring runs over both arrays of 256.
template <bool SL, int HM, typename W>
struct QueueCap {
    static constexpr int value = (SL && sizeof(W) == 8) ? 512 : 256;
};
// Verified matches kept per segment by the main launch before the row is handed to the retry launch.  With the tiers a
// settled segment stops early, so few rows collect more.  With bucket slots (small barcode sets) 6 keep a block's LDS at
// 24.5 KB = six blocks per CU for real (27 KB, the size with 8, is rounded up by the allocation granule and fits only five:
// 3.52 vs 3.39 ms at 25 k barcodes, 0.09 % of the rows retried); larger sets have more ties and keep 8 (400 k barcodes:
// 24 % instead of 16 % of the rows retried with 6).
static constexpr int HMAX = 8, HMAX_SL = 6;  // main launch
static constexpr int K_RETRY = 32, H_RETRY = 80;  // retry launch: 80 x 256 B of LDS hold 320 hit-list entries for each of a wave's 16 rows
// retry launch with slots: 4 rows per wave as well (cfg3's 8 854 retried rows: 0.105 ms with 16 rows per wave, 0.073 with 4, 0.116 with 2)
static constexpr int RETRY_RPW_SL = 4;
static constexpr int H_RETRY_SL = H_RETRY * RETRY_RPW_SL / 16;
// special launch with slots, first stage: 16 rows per wave — 4 window lanes and 80 hit-list entries per row in the same LDS.  Rows
// of more than 4 windows (201 characters at bl = 16, E = 2) or with longer lists go to the second stage, the retry launch's shape
// (4 rows per wave, 16 windows, 320 entries).  95 k one-window rows: 0.345 ms at 4 rows per wave (the waves are latency-bound and
// LDS limits how many are resident), see DESIGN.md for the two-stage figure.
static constexpr int SPECIAL_RPW_SL = 16;
// without slots (larger barcode sets): the same 320 entries per row for 4 rows per wave — a quarter of the LDS per (one-wave) block,
// four times the waves per CU; these launches wait for table reads and are short of waves, not of lanes.  Retry launch per 1 M
// rows with 16 / 8 / 4 rows per wave: 3.6 M barcodes wide scheme 5.4 / 4.0 / 4.1 ms, 1 M wide 5.0 / 5.0 / 4.7, 3.6 M quarters
// 12.6 / 8.4 / 7.9, 400 k quarters 2.3 / 1.8 / 1.6
static constexpr int H_RETRY_NS = 20, RETRY_RPW_NS = 4;
static constexpr uint32_t LOC_MASK = (1u << 30) - 1u;    // locator in a hit-list entry (2 bits of distance above it)
static constexpr uint32_t LOC_INLINE = 1u << 29;         // locator of an inline slot entry (entry indices stay below it)

// where each family's offsets start inside the concatenated offset table; the offsets themselves index the
// concatenated entry tables (family f's entries start at f * np)
struct FamTables {
    uint32_t off_base[SCT_NFAM_TABLES];
    uint32_t fb_base[3];   // thirds families: where each one's fixed buckets start in tab_fb (16-bit entries)
    uint32_t pid16;  // the pattern ids are also held in 16 bits (fewer than 65 536 patterns): half the footprint
};

// WIDE: the placement table holds (t, t2, u) triples of the wide families (sct_core.h), up to 9 combinations of shifts
// SP: the special launch (below, filter_rows) — one more array, the start masks of every lane's text
template <typename W, int HM, int QC, bool WIDE = false, bool SP = false>
struct FilterWaveLds {
    static constexpr int QCAP = QC;
    typedef typename std::conditional<WIDE || sizeof(W) == 8, uint32_t, uint16_t>::type PtabT;
    static constexpr int PTAB = WIDE ? (sizeof(W) == 4 ? 160 : 448) : (sizeof(W) == 4 ? 128 : 320);
    static constexpr int HMAX = HM;
    alignas(32) W em[64][4];  // em[lane][c]: bit i+1 set iff character i of the lane's segment has code c (bits 1..len)
    W va[sizeof(W) == 4 ? 64 : 1];  // bits 0..len (64-bit instantiations derive it from slen: their LDS is the scarcer)
    struct Planes {
        W lo, hi;
    } pl[64];            // the segments' bit-planes (lookup lanes read the planes of the segment they work for)
    uint8_t slen[64];    // and lengths
    uint8_t alist[64];   // lanes whose segment still generates candidates, compacted
    PtabT ptab[PTAB];    // this family's placements for the longest segment: t | t2 << PF | need << 2 PF, PF = 5 / 6 bits (32- / 64-bit
                         // masks; FilterWave::build_ptab); wide families: t | t2 << 8 | u << 16
    uint32_t best[64];   // smallest distance verified so far for the lane's segment
    uint32_t hcnt[64];
    alignas(16) uint32_t hlist[HM][64];  // hlist, qk and qm double as the stage of the ASCII decode (k_decode.hip.h)
    uint32_t qk[QCAP];   // queue: global table entry index f*np + k
    uint32_t qm[QCAP];   //        lane | t << 8 | t2 << 16
    W al[SP ? 64 : 0][4];  // SP: al[lane][d] = A_d of the lane's text (SctStarts, sct_core.h); no bytes otherwise
};

__device__ __forceinline__ uint32_t lane_rank(uint64_t m) {  // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

struct __attribute__((packed, aligned(4))) OffPair {
    uint32_t a, b;
};

template <typename W>
struct EmLds {
    uint32_t base;  // LDS byte address of em[segment lane][0]; aligned to 4 sizeof(W), so the mask's offset is OR-ed in
    __device__ __forceinline__ explicit EmLds(const W *p)
        : base((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) W *)p) {}
    __device__ __forceinline__ W operator()(uint32_t c) const {
        return *(const __attribute__((address_space(3))) W *)(uintptr_t)(base | (c * (uint32_t)sizeof(W)));
    }
};

template <typename W>
struct AlLds {   // d -> A_d of one lane's text (restricted starts), read from LDS
    uint32_t base;
    __device__ __forceinline__ explicit AlLds(const W *p)
        : base((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) W *)p) {}
    __device__ __forceinline__ W operator()(int d) const {
        return *(const __attribute__((address_space(3))) W *)(uintptr_t)(base + (uint32_t)d * (uint32_t)sizeof(W));
    }
};

// Bucket slots (SL = true, the quarter families): one 8-byte word per key that answers most lookups by itself.
//   bit 15 clear: bits 0..1 = n <= 3 entries in the bucket, bits 16j+16 .. 16j+31 = the REST characters of entry j
//                 (sct_rest_of: all the verification needs besides the key); no other table is read unless the
//                 entry verifies, and then its pattern id is tab_ent[off[key] + j]
//   bit 15 set:   a longer bucket: bits 16..47 = its first entry index, bits 0..14 | bits 48..63 << 15 = its length;
//                 the entries are queued by index and verified from tab_code as without slots
// A queued entry is (qk, qm): qm = lane | t << 8 | t2 << 16 | j << 24 | inline << 31; qk = rest characters (inline)
// or entry index.
static constexpr uint32_t QM_INLINE = 0x80000000u;
static constexpr size_t SLOT_TOTAL = (size_t)SCT_NFAM * (SCT_KEYS + 1);  // slots of the six quarter families

template <int E, typename W, int HM, bool SL, int RPW = 64, bool WIDE = false, bool THIRDS = false, bool SP = false>
struct FilterWave {
    // SP (special launch): the lanes are WINDOWS of the wave's RPW rows, lane = window * RPW + row; minimum, hit count and hit
    // list belong to the row
    __device__ static __forceinline__ uint32_t row_of(uint32_t src) { return SP ? (src & (uint32_t)(RPW - 1)) : src; }
    static constexpr int HMAX = HM;
    // hit list entries per row: the LDS array hlist[HM][64] is shared out among the wave's RPW rows, entry k of row r at
    // [k * RPW + r] — HM per row in the main launch (64 rows), four times as many in the retry launch (16 rows)
    static constexpr int CAP = HM * 64 / RPW;
    __device__ __forceinline__ uint32_t &hl(uint32_t k, uint32_t row) { return (&L.hlist[0][0])[k * (uint32_t)RPW + row]; }
    static constexpr int QCAP = QueueCap<SL, HM, W>::value;
    // an inline locator is LOC_INLINE | (index of the key among all offsets) << LOC_J_BITS | position in the bucket: 3 entries
    // per bucket slot, 16 per fixed bucket
    static constexpr int LOC_J_BITS = (THIRDS || WIDE) ? 5 : 2;
    // wide families: bits 6-7 and 14-15 of a queue entry's meta word carry the key's four half bits (lo plane, hi plane)
    __device__ static __forceinline__ uint32_t meta_kb4(uint32_t mx) { return ((mx >> 6) & 3u) | (((mx >> 14) & 3u) << 2); }
    template <typename Fam>
    __device__ __forceinline__ uint32_t inline_locator(const Fam &fam, uint32_t mx, uint32_t sx, int tx, int t2x) const {
        const uint32_t key = sct_key_of<W>(fam, L.pl[sx].lo, L.pl[sx].hi, tx, t2x), j = (mx >> 24) & ((1u << LOC_J_BITS) - 1u);
        if constexpr (WIDE) return LOC_INLINE | ((((wf_ << 20) | (meta_kb4(mx) << 16) | key) << LOC_J_BITS) | j);
        else return LOC_INLINE | ((ft.off_base[fam.INDEX] + key) << LOC_J_BITS) | j;
    }
    FilterWaveLds<W, HM, QCAP, WIDE, SP> &L;
    const uint32_t *__restrict__ tab_off;
    const uint32_t *__restrict__ tab_ent;   // pattern id of every table entry
    const uint32_t *__restrict__ tab_code;  // the same entries as 2-bit codes, character j at bits 2j..2j+1
    const uint2 *__restrict__ tab_slot;     // bucket slots, indexed like tab_off
    const uint32_t *__restrict__ tab_grp;   // entry index at the first key of every group of 64 keys (quarter families)
    const uint16_t *__restrict__ tab_ent16; // tab_ent in 16 bits (fewer than 65 536 patterns)
    const uint16_t *__restrict__ tab_fb;    // fixed buckets of the thirds families (NULL: not built / switched off)
    const uint16_t *__restrict__ tab_wfb;   // fixed buckets of the wide families (NULL: not built / switched off)
    FamTables ft;
    int lane;
    bool active;
    int len;
    W lo, hi;
    uint32_t head = 0, tail = 0;  // wave-uniform
    uint32_t wf_ = 0;             // the wide family whose candidates are in the queue (their inline locators carry it)

    // PK (bucket slots, texts of at most 31 characters): the queue holds ONE word per candidate and a lane verifies TWO
    // candidates at once, one in each 16-bit half of the automaton's words (wm_anchored_pair, sct_core.h), 128 per
    // batch.  Entry = rest characters | segment lane << 16 | t << 22 | (t2 - t - GAP + SLACK) << 27 | j << 30, j = the
    // entry's position in its inline bucket, j = 3: an entry of a long bucket (its rest was read from the code table
    // when it was queued).  No table is read while a batch is verified; the ring runs over qk and qm.
    static constexpr bool PK = SL && sizeof(W) == 4;
    static constexpr uint32_t RING = 2u * (uint32_t)QCAP;
    __device__ __forceinline__ uint32_t *ring() { return &L.qk[0]; }

    // A verified candidate (distance d <= E, locator loc) of the segment in lane `src`.  Only the candidates at the segment's
    // FINAL minimum are reported, and the minimum only falls: one that is farther away than the best seen so far is dropped,
    // and when the list is full a newcomer takes the place of an entry that is farther away than itself (compare-and-swap: two
    // lanes may come for the same place).  The row is deferred only if more entries AT ITS MINIMUM turn up than the list holds
    // — with a few million barcodes most rows see a dozen candidates two edits away before the one that is one edit away, and
    // round 1 deferred them all (92 % of the rows at 3.6 M barcodes, e <= 2).
    // hcnt = entries handed out so far | one overflow mark per distance (bit 29 + d): what counts when the row is finished is the
    // mark of its FINAL minimum — a list that ran over with candidates two edits away is fine again once one that is one edit
    // away has turned up
    static constexpr uint32_t HCNT_OVERFLOW = 0xE0000000u;
    static constexpr int HCNT_OVERFLOW_SHIFT = 29;
    __device__ static __forceinline__ bool overflowed(uint32_t hc, uint32_t best) {
        return best <= (uint32_t)E && ((hc >> (HCNT_OVERFLOW_SHIFT + best)) & 1u) != 0u;
    }
    __device__ __forceinline__ void insert_hit(uint32_t seg_lane, uint32_t d, uint32_t loc) {
        const uint32_t src = row_of(seg_lane);
        if constexpr (SL) {
            // bucket slots = small barcode sets (up to ~40 k): a row sees few candidates besides the right one, the plain
            // list is enough and 4 % faster at 25 k barcodes (3.16 vs 3.28 ms per 10 M rows)
            atomicMin(&L.best[src], d);
            const uint32_t k = atomicAdd(&L.hcnt[src], 1u);
            if (k < (uint32_t)CAP) hl(k, src) = (d << 30) | loc;   // (a count beyond CAP is the overflow mark of this variant)
            return;
        }
        const uint32_t before = atomicMin(&L.best[src], d);
        if (d > before) return;
        const uint32_t entry = (d << 30) | loc;
        const uint32_t k = atomicAdd(&L.hcnt[src], 1u) & ~HCNT_OVERFLOW;
        if (k < (uint32_t)CAP) hl(k, src) = entry;
        // every lane that was given a place has written it before any lane looks for one to take over (two ifs in a row, not an
        // if / else whose sides the compiler may order either way: a place still holding old bytes must not be "replaced")
        __builtin_amdgcn_wave_barrier();
        if (k >= (uint32_t)CAP) {
            bool placed = false;
            for (int i = 0; i < CAP && !placed; i++) {
                uint32_t cur = hl((uint32_t)i, src);
                while ((cur >> 30) > d) {
                    const uint32_t seen = atomicCAS(&hl((uint32_t)i, src), cur, entry);
                    if (seen == cur) {
                        placed = true;
                        break;
                    }
                    cur = seen;
                }
            }
            // no place: unless a closer one has turned up meanwhile the row has more candidates at its minimum than the list holds
            if (!placed && L.best[src] >= d) atomicOr(&L.hcnt[src], 1u << (HCNT_OVERFLOW_SHIFT + d));
        }
    }

    template <typename Fam>
    __device__ __forceinline__ void record(const Fam &fam, uint32_t e, uint32_t d) {
        const uint32_t src = (e >> 16) & 63u, j = e >> 30;
        const int t = (int)((e >> 22) & 31u), t2 = t + fam.GAP + (int)((e >> 27) & 7u) - fam.SLACK;
        const uint32_t oi = ft.off_base[fam.INDEX] + sct_key_of<W>(fam, L.pl[src].lo, L.pl[src].hi, t, t2);
        if (j < 3u) {
            insert_hit(src, d, LOC_INLINE | (oi << 2) | j);
        } else {
            // an entry of a long bucket: every entry of the bucket with these rest characters is this barcode
            // sequence (duplicates are distinct ids and all of them hits; each is also queued itself, the ids are
            // deduplicated when the segment is finished)
            const uint32_t s = tab_off[oi], en = tab_off[oi + 1];
            for (uint32_t g = s; g < en; g++) {
                if (((sct_rest_of(fam, tab_code[g]) ^ e) & 0xFFFFu) != 0u) continue;
                insert_hit(src, d, g);
            }
        }
    }

    template <typename Fam>
    __device__ __forceinline__ void verify2(const Fam &fam, uint32_t nq) {  // the nq <= 128 entries from head
        using Lds = FilterWaveLds<W, HM, QCAP, WIDE, SP>;
        static_assert(__builtin_offsetof(Lds, qm) == __builtin_offsetof(Lds, qk) + 4 * QCAP && __builtin_offsetof(Lds, qk) % 8 == 0,
                      "the ring runs over qk and qm");
        const uint2 pr = *reinterpret_cast<const uint2 *>(&ring()[(head + 2u * (uint32_t)lane) & (RING - 1)]);
        const bool vx = 2u * (uint32_t)lane < nq, vy = 2u * (uint32_t)lane + 1u < nq;
        const uint32_t ex = vx ? pr.x : 0u, ey = vy ? pr.y : 0u;
        const uint32_t sx = (ex >> 16) & 63u, sy = (ey >> 16) & 63u;
        const int tx = (int)((ex >> 22) & 31u), ty = (int)((ey >> 22) & 31u);
        const int t2x = tx + fam.GAP + (int)((ex >> 27) & 7u) - fam.SLACK, t2y = ty + fam.GAP + (int)((ey >> 27) & 7u) - fam.SLACK;
        uint32_t r;
        if constexpr (SP)
            r = wm_anchored_pair<E>(fam, EmLds<W>(&L.em[sx][0]), EmLds<W>(&L.em[sy][0]), (uint32_t)L.va[sx], (uint32_t)L.va[sy], ex, ey, tx, t2x,
                                    ty, t2y, AlLds<W>(&L.al[sx][0]), AlLds<W>(&L.al[sy][0]));
        else
            r = wm_anchored_pair<E>(fam, EmLds<W>(&L.em[sx][0]), EmLds<W>(&L.em[sy][0]), (uint32_t)L.va[sx], (uint32_t)L.va[sy], ex, ey, tx, t2x,
                                    ty, t2y);
        const uint32_t dx = r & 0xFFFFu, dy = r >> 16;
        if (vx && dx <= (uint32_t)E) record(fam, ex, dx);
        if (vy && dy <= (uint32_t)E) record(fam, ey, dy);
        __builtin_amdgcn_wave_barrier();
    }

    // Everywhere else — no slots, or 64-bit masks — the queue holds two-word entries (entry index or, for an inline
    // slot entry, its rest characters; lane | t << 8 | t2 << 16 | j << 24 | inline << 31) and pairs are verified the same
    // way, 128 entries per batch; the codes of entries queued by index are read from the table first.
    __device__ __forceinline__ W va_of(uint32_t src) const {
        if constexpr (sizeof(W) == 4) return L.va[src];
        else return sct_valid_mask<W>((int)L.slen[src]);
    }
    template <typename Fam>
    __device__ __forceinline__ void verify2w(const Fam &fam, uint32_t nq) {  // the nq <= 128 entries from head
        if (2u * (uint32_t)lane < nq) {
            const uint32_t i0 = (head + 2u * (uint32_t)lane) & (QCAP - 1);  // head is even: the pair does not wrap
            const bool vy = 2u * (uint32_t)lane + 1u < nq;
            uint32_t gx = L.qk[i0], gy = vy ? L.qk[i0 + 1] : gx;
            const uint32_t mx = L.qm[i0], my = vy ? L.qm[i0 + 1] : mx;
            const bool ix = (SL || THIRDS || WIDE) && (mx & QM_INLINE) != 0u, iy = (SL || THIRDS || WIDE) && (my & QM_INLINE) != 0u;
            uint32_t rx = gx, ry = gy;  // an inline entry carries its rest characters (bits above them are ignored)
            if (!ix) rx = sct_rest_of(fam, tab_code[gx]);
            if (!iy) ry = sct_rest_of(fam, tab_code[gy]);
            const uint32_t sx = mx & 63u, sy = my & 63u;
            const int tx = (int)((mx >> 8) & 63u), t2x = (int)((mx >> 16) & 63u);
            const int ty = (int)((my >> 8) & 63u), t2y = (int)((my >> 16) & 63u);
            uint32_t r;
            if constexpr (SP)
                r = wm_anchored_pair<E>(fam, EmLds<W>(&L.em[sx][0]), EmLds<W>(&L.em[sy][0]), va_of(sx), va_of(sy), rx, ry, tx, t2x, ty, t2y,
                                        AlLds<W>(&L.al[sx][0]), AlLds<W>(&L.al[sy][0]));
            else
                r = wm_anchored_pair<E>(fam, EmLds<W>(&L.em[sx][0]), EmLds<W>(&L.em[sy][0]), va_of(sx), va_of(sy), rx, ry, tx, t2x, ty, t2y);
            const uint32_t dx = r & 0xFFFFu, dy = r >> 16;
            // what is kept is the entry's LOCATOR; the pattern ids are read once, when the segment is finished
            if (dx <= (uint32_t)E) {
                if (ix) gx = inline_locator(fam, mx, sx, tx, t2x);
                insert_hit(sx, dx, gx);
            }
            if (vy && dy <= (uint32_t)E) {
                if (iy) gy = inline_locator(fam, my, sy, ty, t2y);
                insert_hit(sy, dy, gy);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // pattern id of a verified entry: locator = entry index, or LOC_INLINE | (slot index) << 2 | j.  An inline slot says where
    // its bucket starts inside its group of 64 keys; with the group's base (a 24 KB table) that is the entry index.  Everything
    // read here — the slot again, the group bases, the dense entry ids (16-bit below 65 536 patterns: 0.6 MB at 25 k barcodes) —
    // stays in the L2 beside the slots; round 1's per-slot id table (2.4 MB, three quarters of it empty) did not, and its
    // misses were 60 % of the kernel's memory-side traffic.
    __device__ __forceinline__ uint32_t resolve(uint32_t loc) const {
        uint32_t idx = loc;
        if (WIDE && (loc & LOC_INLINE)) {     // (wide family, key, j): the families' offsets follow one another, nkeys + 1 each
            const uint32_t l = loc & (LOC_INLINE - 1u), kk = l >> LOC_J_BITS;
            idx = tab_off[ft.off_base[SCT_F_WIDE0] + (kk >> 20) * (SCT_WIDE_KEYS + 1u) + (kk & (SCT_WIDE_KEYS - 1u))] + (l & ((1u << LOC_J_BITS) - 1u));
        }
        if (THIRDS && (loc & LOC_INLINE)) {   // entry j of the key's bucket: the offsets say where the bucket starts
            const uint32_t l = loc & (LOC_INLINE - 1u);
            idx = tab_off[l >> LOC_J_BITS] + (l & ((1u << LOC_J_BITS) - 1u));
        }
        if (SL && (loc & LOC_INLINE)) {
            const uint32_t l = loc & (LOC_INLINE - 1u), oi = l >> 2, j = l & 3u;
            const uint32_t f = oi / (uint32_t)(SCT_KEYS + 1), key = oi - f * (uint32_t)(SCT_KEYS + 1);
            idx = tab_grp[f * (uint32_t)SCT_GRP_PER_FAM + (key >> SCT_GRP_SHIFT)] + sct_slot_within(tab_slot[oi].x) + j;
        }
        if (ft.pid16) return reinterpret_cast<const uint16_t *>(tab_ent16)[idx];
        return tab_ent[idx];
    }

    // Tiering: the families are ordered so that after H1 every barcode at distance 0 has been verified, and after
    // H1 + H2 every barcode at distance <= 1 (one half is untouched).  A segment whose minimum is already settled
    // stops generating candidates for the later families — only the minimum tier is reported.
    __device__ __forceinline__ void settle(uint32_t tier) {
        if (active && L.best[row_of((uint32_t)lane)] <= tier) active = false;
    }

    // Once every candidate closer than E has been seen (after settle(E - 1)) a minimum of E is final, and so is the overflow mark of
    // distance E: such a row goes to the next launch whatever else turns up, and stops generating candidates here.  (At 3.6 M
    // barcodes the rows whose minimum is two edits tie 40 - 90 ways: 11.5 % of the rows, and they are the ones that run all families.)
    __device__ __forceinline__ void give_up() {
        if constexpr (!SL) {
            if (active && L.best[row_of((uint32_t)lane)] == (uint32_t)E && overflowed(L.hcnt[row_of((uint32_t)lane)], (uint32_t)E)) active = false;
        }
    }

    // Append the bucket of every lane to the queue; 64-entry batches are verified as they form.  (a, b) are the two
    // words the lookup loaded.
    //   SL:  a bucket slot.  All inline entries of the wave (<= 3 per lane) are placed in ONE pass: the two bits of
    //        the lane's count are balloted separately, so the lane's first queue position is
    //        tail + #(bit 0 below) + 2 #(bit 1 below).  Long buckets (rare while the barcode set is small) follow one
    //        at a time, spread over the lanes: lane i queues entry first + i.
    //   !SL: the bucket's first and end entry index; one entry per lane and pass.
    template <typename Fam>
    __device__ __forceinline__ void push_range(const Fam &fam, uint32_t a, uint32_t b, uint32_t meta) {
        if constexpr (PK) {
            // one-word entries (meta = segment lane << 16 | t << 22 | placement << 27); batches of 128
            const bool inl = sct_slot_is_inline(a);
            const uint32_t c = inl ? sct_slot_inline_count(a) : 0u;
            const uint64_t m0 = __ballot((c & 1u) != 0u), m1 = __ballot((c & 2u) != 0u);
            const uint32_t A = sct_slot_mid(a, b), B = sct_slot_top(b);
            if ((m0 | m1) != 0) {
                const uint32_t pos = tail + lane_rank(m0) + 2u * lane_rank(m1);
                if (c >= 1u) ring()[pos & (RING - 1)] = (A & 0xFFFFu) | meta;
                if (c >= 2u) ring()[(pos + 1u) & (RING - 1)] = (A >> 16) | meta | (1u << 30);
                if (c >= 3u) ring()[(pos + 2u) & (RING - 1)] = B | meta | (2u << 30);
                tail += (uint32_t)__popcll(m0) + 2u * (uint32_t)__popcll(m1);
                __builtin_amdgcn_wave_barrier();
                while (tail - head >= 128u) {
                    verify2(fam, 128u);
                    head += 128u;
                }
            }
            uint64_t om = __ballot(!inl);
            while (om != 0) {  // a long bucket, spread over the lanes: lane i reads the code of entry first + i
                const int src = __builtin_ctzll(om);
                om &= om - 1;
                const uint32_t first = __builtin_amdgcn_readlane(A, src);
                const uint32_t n = sct_slot_long_count(__builtin_amdgcn_readlane(a, src), __builtin_amdgcn_readlane(B, src));
                const uint32_t hdr = __builtin_amdgcn_readlane(meta, src) | (3u << 30);
                for (uint32_t k0 = 0; k0 < n; k0 += 64) {
                    const uint32_t take = n - k0 < 64u ? n - k0 : 64u;
                    if ((uint32_t)lane < take)
                        ring()[(tail + (uint32_t)lane) & (RING - 1)] =
                            (sct_rest_of(fam, tab_code[first + k0 + (uint32_t)lane]) & 0xFFFFu) | hdr;
                    tail += take;
                    __builtin_amdgcn_wave_barrier();
                    while (tail - head >= 128u) {
                        verify2(fam, 128u);
                        head += 128u;
                    }
                }
            }
        } else if constexpr (SL) {
            const bool inl = sct_slot_is_inline(a);
            const uint32_t c = inl ? sct_slot_inline_count(a) : 0u;
            const uint64_t m0 = __ballot((c & 1u) != 0u), m1 = __ballot((c & 2u) != 0u);
            const uint32_t A = sct_slot_mid(a, b);  // slot bits 16..47: two rests / first entry index
            const uint32_t B = sct_slot_top(b);     // slot bits 48..63: the third rest / length >> 15
            if ((m0 | m1) != 0) {
                const uint32_t pos = tail + lane_rank(m0) + 2u * lane_rank(m1);
                const uint32_t mi = meta | QM_INLINE;
                if (c >= 1u) {
                    const uint32_t idx = pos & (QCAP - 1);
                    L.qk[idx] = A;
                    L.qm[idx] = mi;
                }
                if (c >= 2u) {
                    const uint32_t idx = (pos + 1u) & (QCAP - 1);
                    L.qk[idx] = sct_slot_next(A, B);
                    L.qm[idx] = mi + (1u << 24);
                }
                if (c >= 3u) {
                    const uint32_t idx = (pos + 2u) & (QCAP - 1);
                    L.qk[idx] = B;
                    L.qm[idx] = mi + (2u << 24);
                }
                tail += (uint32_t)__popcll(m0) + 2u * (uint32_t)__popcll(m1);
                __builtin_amdgcn_wave_barrier();
                while (tail - head >= 128u) {
                    verify2w(fam, 128u);
                    head += 128u;
                }
            }
            uint64_t om = __ballot(!inl);
            while (om != 0) {
                const int src = __builtin_ctzll(om);
                om &= om - 1;
                const uint32_t first = __builtin_amdgcn_readlane(A, src);
                const uint32_t n = sct_slot_long_count(__builtin_amdgcn_readlane(a, src), __builtin_amdgcn_readlane(B, src));
                const uint32_t mt = __builtin_amdgcn_readlane(meta, src);
                for (uint32_t k0 = 0; k0 < n; k0 += 64) {
                    const uint32_t take = n - k0 < 64u ? n - k0 : 64u;
                    if ((uint32_t)lane < take) {
                        const uint32_t idx = (tail + (uint32_t)lane) & (QCAP - 1);
                        L.qk[idx] = first + k0 + (uint32_t)lane;
                        L.qm[idx] = mt;
                    }
                    tail += take;
                    __builtin_amdgcn_wave_barrier();
                    while (tail - head >= 128u) {
                        verify2w(fam, 128u);
                        head += 128u;
                    }
                }
            }
        } else {
            uint32_t c = b - a;
            for (;;) {
                const bool has = c > 0;
                const uint64_t m = __ballot(has);
                if (m == 0) break;
                if (has) {
                    const uint32_t idx = (tail + lane_rank(m)) & (QCAP - 1);
                    L.qk[idx] = a;
                    L.qm[idx] = meta;
                    a++;
                    c--;
                }
                tail += (uint32_t)__popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (tail - head >= 128u) {
                    verify2w(fam, 128u);
                    head += 128u;
                }
            }
        }
    }

    // The family's placements for a text of `maxlen` characters (the wave's longest), pruned (sct_placement_need, sct_core.h):
    // one lane per start t, shift after shift, so the table is ordered by (d, t) as the unpruned walk was.  An entry carries the
    // shortest text the placement is worth a lookup for; a lane working for a shorter segment skips it.
    static constexpr int PF = sizeof(W) == 4 ? 5 : 6;
    static constexpr uint32_t PFM = (1u << PF) - 1u;
    template <typename Fam>
    __device__ __forceinline__ int build_ptab(const Fam &fam, int maxlen) {
        int P = 0;
#pragma unroll
        for (int d = -fam.SLACK; d <= fam.SLACK; d++) {
            const int t2 = lane + fam.GAP + d;
            const int need = sct_placement_need(fam, E, lane, t2);
            const bool ok = need <= maxlen;
            const uint64_t m = __ballot(ok);
            if (ok) L.ptab[P + (int)lane_rank(m)] = (typename FilterWaveLds<W, HM, QCAP, WIDE, SP>::PtabT)((uint32_t)lane | ((uint32_t)t2 << PF) | ((uint32_t)need << (2 * PF)));
            P += (int)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        return P;
    }

    // One family.  The lookups are spread over the lanes as (active segment, placement) PAIRS: with A segments of
    // the wave still generating candidates and P placements, the A * P pairs are numbered placement-major and every
    // pass takes the next 64 of them, so the wave-uniform cost of a pass is shared by 64 lookups even when most
    // segments have settled (a lane works for a different segment in every pass and reads its planes from LDS).  The placements of the family for the longest segment of the wave are
    // tabulated in LDS once; a lane skips those its own segment is too short for.  The bucket range of a pass is
    // queued while the loads of the next pass are in flight.
    template <typename Fam>
    __device__ __forceinline__ void run_family(const Fam &fam, int maxlen) {
        const uint64_t am = __ballot(active);
        if (am == 0) return;
        const int A = (int)__popcll(am);
        if (active) L.alist[lane_rank(am)] = (uint8_t)lane;
        const int P = build_ptab(fam, maxlen);
        if (P == 0) return;
        // SL:  pair q = 64 * pass + lane is placement q / A of the (q mod A)-th active segment; advancing q by 64 moves
        //      the lane dp placements on and da segments further (with a carry): every pass takes 64 pairs
        // !SL: lane = g * A + a works for the a-th active segment throughout, on placement dp * pass + g (dp = 64 / A);
        //      lanes beyond dp * A idle.  (With seed tables far beyond the L2 the fuller passes of the first scheme
        //      cost more than they save: 10.5 vs 8.4 ms at 1 M barcodes.)
        const uint32_t inv = 65536u / (uint32_t)A + 1u;           // lane / A for lane < 64 (checked for every A <= 64)
        int p = (int)(((uint32_t)lane * inv) >> 16), a = lane - p * A;
        const int dp = 64 / A, da = 64 - dp * A;
        const bool worker = SL || p < dp;
        const int passes = SL ? (A * P + 63) / 64 : (P + dp - 1) / dp;
        uint32_t sl = L.alist[a];
        W slo = L.pl[sl].lo, shi = L.pl[sl].hi;
        int slen = (int)L.slen[sl];
        const uint32_t *off = tab_off + ft.off_base[fam.INDEX];
        const uint2 *slot = tab_slot + ft.off_base[fam.INDEX];
        uint32_t ps = 0, pe = 0, pmeta = 0;  // the pass whose loads were issued last
        bool pend = false;
        for (int pass = 0; pass < passes; pass++) {
            if constexpr (SL) {
                if (pass > 0) {
                    sl = L.alist[a];
                    slo = L.pl[sl].lo;
                    shi = L.pl[sl].hi;
                    slen = (int)L.slen[sl];
                }
            }
            uint32_t s = 0, e = 0, meta = 0;
            if constexpr (SL) {
                // branch-free: a lane without a lookup of its own reads the spare all-zero slot behind the family's keys
                const uint32_t tt = L.ptab[p < P ? p : P - 1];
                const int t = (int)(tt & PFM), t2 = (int)((tt >> PF) & PFM);
                const bool ok = worker && p < P && (int)(tt >> (2 * PF)) <= slen;
                const uint32_t key = sct_key_of<W>(fam, slo, shi, t, t2);
                const uint2 sv = slot[ok ? key : (uint32_t)fam.NKEYS];
                s = sv.x;
                e = sv.y;
                if constexpr (PK) meta = (sl << 16) | ((uint32_t)t << 22) | ((uint32_t)(t2 - t - fam.GAP + fam.SLACK) << 27);
                else meta = sl | ((uint32_t)t << 8) | ((uint32_t)t2 << 16);
            } else if (worker && p < P) {
                const uint32_t tt = L.ptab[p];
                const int t = (int)(tt & PFM), t2 = (int)((tt >> PF) & PFM);
                if ((int)(tt >> (2 * PF)) <= slen) {
                    // one 8-byte load: the bucket's slot, or its start and end (neighbouring words, 4-byte aligned;
                    // absolute entry indices).  Nothing depends on the load until the bucket is queued, one pass later.
                    const uint32_t key = sct_key_of<W>(fam, slo, shi, t, t2);
                    if constexpr (SL) {
                        const uint2 sv = slot[key];
                        s = sv.x;
                        e = sv.y;
                    } else {
                        const OffPair op = *reinterpret_cast<const OffPair *>(off + key);
                        s = op.a;
                        e = op.b;
                    }
                    meta = sl | ((uint32_t)t << 8) | ((uint32_t)t2 << 16);
                }
            }
            p += dp;
            if constexpr (SL) {
                a += da;
                if (a >= A) {
                    a -= A;
                    p++;
                }
            }
            if (pend) push_range(fam, ps, pe, pmeta);
            pend = true;
            ps = s;
            pe = e;
            pmeta = meta;
        }
        if (pend) push_range(fam, ps, pe, pmeta);
        if constexpr (PK) {
            if (tail != head) verify2(fam, tail - head);  // the family's last, partial batch
            head = tail = 0;                              // pairs stay 8-byte aligned in the ring
        } else {
            if (tail != head) verify2w(fam, tail - head);
            head = tail = 0;
        }
    }

    // One family of the thirds scheme through its FIXED BUCKETS (sct_tables.h): every key owns room for CAPE 16-bit entries
    // — rest characters | 0x8000 — in one aligned block, so a lookup reads ONE line and has everything a verification
    // needs; offsets and entry tables are read only for a verified match (its pattern id) and for the rare bucket that holds more
    // than CAPE entries (first entry = 0x4000).  With offsets + codes every lookup cost two lines nothing else had touched:
    // 81 missed lines per segment at 3.6 M barcodes, 3.4 TB/s of random reads (profiles/r02e_cfg5_counters.json).
    template <int CAPW>
    struct FbBucket {
        uint32_t w[CAPW];  // two entries per word
    };
    template <typename Fam, int CAPW>
    __device__ __forceinline__ void push_fb(const Fam &fam, const FbBucket<CAPW> &bk, uint32_t meta, const uint32_t *off) {
        const bool over = (bk.w[0] & SCT_FB_OVERFLOW) != 0u;
        FbBucket<CAPW> bb = bk;
        // (a rolled loop that shifts the bucket down by one entry per round: unrolled, every round carries its own copy of the
        // verification and the translation unit takes half an hour to compile)
#pragma unroll 1
        for (int j = 0; j < 2 * CAPW; j++) {
            const uint32_t e = bb.w[0] & 0xFFFFu;
            const bool has = (e & SCT_FB_VALID) != 0u;
            const uint64_t m = __ballot(has);
            if (m == 0) break;  // a bucket is filled from its first place on
            if (has) {
                const uint32_t idx = (tail + lane_rank(m)) & (QCAP - 1);
                L.qk[idx] = e & 0x0FFFu;
                L.qm[idx] = meta | ((uint32_t)j << 24) | QM_INLINE;
            }
            tail += (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (tail - head >= 128u) {
                verify2w(fam, 128u);
                head += 128u;
            }
#pragma unroll
            for (int q = 0; q + 1 < CAPW; q++) bb.w[q] = __builtin_amdgcn_alignbit(bb.w[q + 1], bb.w[q], 16);
            bb.w[CAPW - 1] >>= 16;
        }
        if (__ballot(over) != 0) {  // more entries than a fixed bucket holds: through the offsets, entry by entry
            uint32_t a = 0, b = 0;
            if (over) {
                const uint32_t sl = meta & 63u;
                const int t = (int)((meta >> 8) & 63u), t2 = (int)((meta >> 16) & 63u);
                const OffPair op = *reinterpret_cast<const OffPair *>(off + sct_key_of<W>(fam, L.pl[sl].lo, L.pl[sl].hi, t, t2));
                a = op.a;
                b = op.b;
            }
            push_range(fam, a, b, meta);
        }
    }
    template <typename Fam>
    __device__ __forceinline__ void run_family_fb(const Fam &fam, int maxlen) {
        static_assert(THIRDS && !SL, "fixed buckets belong to the thirds families");
        constexpr int CAPW = SCT_FB_CAP(Fam::KEYCH) / 2;  // T01 (20-bit keys): 16 entries, T12 / T02 (22-bit keys): 8
        const uint64_t am = __ballot(active);
        if (am == 0) return;
        const int A = (int)__popcll(am);
        if (active) L.alist[lane_rank(am)] = (uint8_t)lane;
        const int P = build_ptab(fam, maxlen);
        if (P == 0) return;
        const uint32_t inv = 65536u / (uint32_t)A + 1u;
        int p = (int)(((uint32_t)lane * inv) >> 16);
        const int a = lane - p * A;
        const int dp = 64 / A;
        const bool worker = p < dp;
        const int passes = (P + dp - 1) / dp;
        const uint32_t sl = L.alist[a];
        const W slo = L.pl[sl].lo, shi = L.pl[sl].hi;
        const int slen = (int)L.slen[sl];
        const uint32_t *off = tab_off + ft.off_base[fam.INDEX];
        const uint4 *fb = reinterpret_cast<const uint4 *>(tab_fb + ft.fb_base[Fam::INDEX - SCT_F_T01]);
        FbBucket<CAPW> prev{};
        uint32_t pmeta = 0;
        bool pend = false;
        for (int pass = 0; pass < passes; pass++) {
            FbBucket<CAPW> cur{};
            uint32_t meta = 0;
            if (worker && p < P) {
                const uint32_t tt = L.ptab[p];
                const int t = (int)(tt & PFM), t2 = (int)((tt >> PF) & PFM);
                if ((int)(tt >> (2 * PF)) <= slen) {
                    const uint32_t key = sct_key_of<W>(fam, slo, shi, t, t2);
#pragma unroll
                    for (int q = 0; q < CAPW / 4; q++) {
                        const uint4 v = fb[(size_t)key * (CAPW / 4) + q];
                        cur.w[4 * q + 0] = v.x;
                        cur.w[4 * q + 1] = v.y;
                        cur.w[4 * q + 2] = v.z;
                        cur.w[4 * q + 3] = v.w;
                    }
                    meta = sl | ((uint32_t)t << 8) | ((uint32_t)t2 << 16);
                }
            }
            p += dp;
            if (pend) push_fb(fam, prev, pmeta, off);
            pend = true;
            prev = cur;
            pmeta = meta;
        }
        if (pend) push_fb(fam, prev, pmeta, off);
        if (tail != head) verify2w(fam, tail - head);
        head = tail = 0;
    }

    // One family of the wide scheme (sct_core.h, SctWide<WF>): the lookups of run_family's table variant with a third piece in
    // the key — placements are (t, t2, u) —; what is queued and how it is verified is the quarter pair's business (Base).
    // Fixed buckets as for the thirds: 16 entries per key, each the SIX rest characters that are not in the key; the half's two
    // characters come back from the key when the entry is queued (sct_wide_rest16), so the automaton sees the pair's usual rest.
    template <int WF, typename Fam>
    __device__ __forceinline__ void push_wfb(const Fam &fam, const FbBucket<8> &bk, uint32_t meta, const uint32_t *off) {
        constexpr int POS = SctWide<WF>::POS;
        const bool over = (bk.w[0] & SCT_FB_OVERFLOW) != 0u;
        const uint32_t kb4 = meta_kb4(meta);
        // plane bits (character 0 and 1 of the half: lo bits 0-1, hi bits 2-3) -> two 2-bit codes
        const uint32_t half4 = (kb4 & 1u) | (((kb4 >> 2) & 1u) << 1) | (((kb4 >> 1) & 1u) << 2) | (((kb4 >> 3) & 1u) << 3);
        FbBucket<8> bb = bk;
#pragma unroll 1
        for (int j = 0; j < 16; j++) {  // rolled, as in push_fb
            const uint32_t e = bb.w[0] & 0xFFFFu;
            const bool has = (e & SCT_FB_VALID) != 0u;
            const uint64_t m = __ballot(has);
            if (m == 0) break;
            if (has) {
                const uint32_t idx = (tail + lane_rank(m)) & (QCAP - 1);
                L.qk[idx] = sct_wide_rest16(e & 0x0FFFu, half4, POS);
                L.qm[idx] = meta | ((uint32_t)j << 24) | QM_INLINE;
            }
            tail += (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (tail - head >= 128u) {
                verify2w(fam, 128u);
                head += 128u;
            }
#pragma unroll
            for (int q = 0; q + 1 < 8; q++) bb.w[q] = __builtin_amdgcn_alignbit(bb.w[q + 1], bb.w[q], 16);
            bb.w[7] >>= 16;
        }
        if (__ballot(over) != 0) {
            uint32_t a = 0, b = 0;
            if (over) {
                const uint32_t sl = meta & 63u;
                const int t = (int)((meta >> 8) & 63u), t2 = (int)((meta >> 16) & 63u);
                const OffPair op = *reinterpret_cast<const OffPair *>(off + (sct_key_of<W>(fam, L.pl[sl].lo, L.pl[sl].hi, t, t2) | (kb4 << 16)));
                a = op.a;
                b = op.b;
            }
            push_range(fam, a, b, meta);
        }
    }
    template <int WF>
    __device__ __forceinline__ void run_family_wide_fb(int maxlen) {
        static_assert(WIDE && !SL, "the wide families come with the plain tables");
        using F = SctWide<WF>;
        const typename F::Base fam{};
        wf_ = (uint32_t)WF;
        const uint64_t am = __ballot(active);
        if (am == 0) return;
        const int A = (int)__popcll(am);
        if (active) L.alist[lane_rank(am)] = (uint8_t)lane;
        int P = 0;
#pragma unroll
        for (int s = -F::SMAX; s <= F::SMAX; s++) {
#pragma unroll
            for (int x = -F::DMAX; x <= F::DMAX; x++) {
                const int d = F::COUPLED ? s + x : x;
                const int t2o = fam.GAP + d, uo = (F::FROM_B ? t2o : 0) + F::OFFC + s;
                const int t0 = uo < 0 ? -uo : 0;
                const int ext = t2o + fam.LB > uo + 2 ? t2o + fam.LB : uo + 2;
                const int c = maxlen - ext - t0 + 1;
                if (c > 0) {
                    for (int i = lane; i < c; i += 64)
                        L.ptab[P + i] = (uint32_t)(t0 + i) | ((uint32_t)(t0 + i + t2o) << 8) | ((uint32_t)(t0 + i + uo) << 16);
                    P += c;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (P == 0) return;
        const uint32_t inv = 65536u / (uint32_t)A + 1u;
        int p = (int)(((uint32_t)lane * inv) >> 16);
        const int a = lane - p * A;
        const int dp = 64 / A;
        const bool worker = p < dp;
        const int passes = (P + dp - 1) / dp;
        const uint32_t sl = L.alist[a];
        const W slo = L.pl[sl].lo, shi = L.pl[sl].hi;
        const int slen = (int)L.slen[sl];
        const uint32_t *off = tab_off + ft.off_base[F::INDEX];
        const uint4 *fb = reinterpret_cast<const uint4 *>(tab_wfb + ((size_t)WF * SCT_WIDE_KEYS) * SCT_WIDE_FB_CAP);
        FbBucket<8> prev{};
        uint32_t pmeta = 0;
        bool pend = false;
        for (int pass = 0; pass < passes; pass++) {
            FbBucket<8> cur{};
            uint32_t meta = 0;
            if (worker && p < P) {
                const uint32_t tt = L.ptab[p];
                const int t = (int)(tt & 0xFFu), t2 = (int)((tt >> 8) & 0xFFu), u = (int)(tt >> 16);
                if (sct_wide_valid(t2, u, slen)) {
                    const uint32_t key = sct_wide_key<WF, W>(slo, shi, t, t2, u);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const uint4 v = fb[(size_t)key * 2 + q];
                        cur.w[4 * q + 0] = v.x;
                        cur.w[4 * q + 1] = v.y;
                        cur.w[4 * q + 2] = v.z;
                        cur.w[4 * q + 3] = v.w;
                    }
                    const uint32_t kb4 = key >> 16;
                    meta = sl | ((uint32_t)t << 8) | ((uint32_t)t2 << 16) | ((kb4 & 3u) << 6) | ((kb4 >> 2) << 14);
                }
            }
            p += dp;
            if (pend) push_wfb<WF>(fam, prev, pmeta, off);
            pend = true;
            prev = cur;
            pmeta = meta;
        }
        if (pend) push_wfb<WF>(fam, prev, pmeta, off);
        if (tail != head) verify2w(fam, tail - head);
        head = tail = 0;
    }

    // The same family through offsets and entry tables (two lines per lookup, fewer instructions per queued entry): what the
    // RETRY launch uses — it is bound by its instruction count, not by the table reads (fixed buckets: 4.1 -> 4.5 ms per 1 M rows
    // at 3.6 M barcodes, while the main launch, bound by the reads, goes 3.4 -> 2.6 ms).
    template <int WF>
    __device__ __forceinline__ void run_family_wide(int maxlen) {
        static_assert(WIDE && !SL, "the wide families come with the plain tables");
        using F = SctWide<WF>;
        const typename F::Base fam{};
        const uint64_t am = __ballot(active);
        if (am == 0) return;
        const int A = (int)__popcll(am);
        if (active) L.alist[lane_rank(am)] = (uint8_t)lane;
        int P = 0;
#pragma unroll
        for (int s = -F::SMAX; s <= F::SMAX; s++) {
#pragma unroll
            for (int x = -F::DMAX; x <= F::DMAX; x++) {
                const int d = F::COUPLED ? s + x : x;
                const int t2o = fam.GAP + d, uo = (F::FROM_B ? t2o : 0) + F::OFFC + s;
                const int t0 = uo < 0 ? -uo : 0;
                const int ext = t2o + fam.LB > uo + 2 ? t2o + fam.LB : uo + 2;
                const int c = maxlen - ext - t0 + 1;
                if (c > 0) {
                    for (int i = lane; i < c; i += 64)
                        L.ptab[P + i] = (uint32_t)(t0 + i) | ((uint32_t)(t0 + i + t2o) << 8) | ((uint32_t)(t0 + i + uo) << 16);
                    P += c;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (P == 0) return;
        const uint32_t inv = 65536u / (uint32_t)A + 1u;
        int p = (int)(((uint32_t)lane * inv) >> 16);
        const int a = lane - p * A;
        const int dp = 64 / A;
        const bool worker = p < dp;
        const int passes = (P + dp - 1) / dp;
        const uint32_t sl = L.alist[a];
        const W slo = L.pl[sl].lo, shi = L.pl[sl].hi;
        const int slen = (int)L.slen[sl];
        const uint32_t *off = tab_off + ft.off_base[F::INDEX];
        uint32_t ps = 0, pe = 0, pmeta = 0;
        bool pend = false;
        for (int pass = 0; pass < passes; pass++) {
            uint32_t s = 0, e = 0, meta = 0;
            if (worker && p < P) {
                const uint32_t tt = L.ptab[p];
                const int t = (int)(tt & 0xFFu), t2 = (int)((tt >> 8) & 0xFFu), u = (int)(tt >> 16);
                if (sct_wide_valid(t2, u, slen)) {
                    const uint32_t key = sct_wide_key<WF, W>(slo, shi, t, t2, u);
                    const OffPair op = *reinterpret_cast<const OffPair *>(off + key);
                    s = op.a;
                    e = op.b;
                    meta = sl | ((uint32_t)t << 8) | ((uint32_t)t2 << 16);
                }
            }
            p += dp;
            if (pend) push_range(fam, ps, pe, pmeta);
            pend = true;
            ps = s;
            pe = e;
            pmeta = meta;
        }
        if (pend) push_range(fam, ps, pe, pmeta);
        if (tail != head) verify2w(fam, tail - head);
        head = tail = 0;
    }
};

// W = uint32_t handles the waves whose longest segment has at most 31 characters, W = uint64_t the others
// (up to 63); both instantiations are launched over all rows and a wave leaves at once if it is not its kind.
// THIRDS = true (max_err <= 1 only) walks the 20/22-bit families of the thirds scheme instead of the quarters.
// KS / HM: ties kept per segment / verified matches kept per segment before the row is deferred.  The main launch
// (KS 8, HM 16) covers all rows; rows it defers are retried by a launch with KS 32, HM 64 over the list `rows`
// (4 listed rows per wave), and only what that launch defers again goes to the exhaustive scan.
// RT = true: barcode length `bl` (8..16) other than 16, piece boundaries computed at run time (sct_family_rt).
// SL = true: lookups read bucket slots (FilterWave); chosen by the host while few buckets are longer than a slot.
// OffT: type of the CSR offsets (FilterArgs).
//
// Everything a launch needs besides its template parameters.
// What the main launch leaves for the special launch about a row of at most PLANE_MAX characters it lists (a byte outside ACGT
// under a prefix length k > 0): planes, the positions of the bytes that are not one of ACGT, length — the special launch need not
// read the row again (over the PCIe link, when the rows are the caller's page-locked buffer: 0.76 instead of 0.35 ms per 95 k rows).
struct __attribute__((aligned(32))) SctSpecialRec {
    uint64_t lo, hi, bad, len;
};
// len = SCT_REC_NONE: nothing is known about the row (read its bytes where the caller has them); SCT_REC_GATHERED: the row has more
// than PLANE_MAX characters and k_special_gather copied it to the device: its first byte is at gather[lo]
static constexpr uint64_t SCT_REC_NONE = ~0ull, SCT_REC_GATHERED = ~1ull;

template <typename OffT>
struct FilterArgs {
    const uint8_t *segs;          // the rows' ASCII bytes (padded by 64 bytes past the end)
    const OffT *offs;             // CSR offsets of the rows: int64 (resident upload) or uint32 relative to the chunk (stream)
    int64_t seg_bytes;
    int64_t n;                    // rows (main launch) / listed rows (retry launch) ...
    const uint32_t *n_dev;        // ... or, if not NULL, where that number lives on the device (the retry launch is queued
                                  // before the host knows how many rows the main launch handed over)
    const int32_t *rows;          // NULL = all rows; else the listed rows (retry launch)
    const uint32_t *tab_off, *tab_ent, *tab_code;
    const uint2 *tab_slot;
    const uint32_t *tab_grp;
    const uint16_t *tab_ent16;
    const uint16_t *tab_fb;       // fixed buckets of the thirds families, or NULL
    const uint16_t *tab_wfb;      // fixed buckets of the wide families, or NULL
    FamTables ft;
    uint8_t *min_e;
    uint8_t *cnt8;                // NULL, or where the rows' hit counts go as one byte each (the caller's compact format)
    uint32_t *cnt;
    uint32_t *slots;
    uint64_t slots_cap;
    uint64_t *slot_off;           // retry launch: where each of its rows' lists starts in `slots`
    uint64_t *wave_base;          // main launch: where each wave's lists start in `slots`
    int32_t *list_defer;          // main launch: list_retry; retry launch: list_scan
    uint32_t *defer_count;
    int32_t *list_special;        // main launch: rows for the special launch (restricted starts, more than PLANE_MAX characters)
    SctSpecialRec *rec_out;       // ... and what it knows about them, entry for entry (first rec_cap list positions); the special launch's
                                  // first stage: the same for the rows it hands to the second (beside list_defer)
    const SctSpecialRec *rec_in;  // special launch: the records of its listed rows
    uint32_t rec_cap;
    const uint8_t *gather;        // special launch: where the long rows of a zero-copy batch were copied (SCT_REC_GATHERED records), or NULL
    bool skip64_unless_needed;    // main launch, both instantiations over all rows: the 32-bit one marks ctr[CTR_NEED64] when it meets a
                                  //   wave of the other kind, the 64-bit one (launched behind it) leaves at once if nobody did
    uint8_t *gather_out;          // main launch of a zero-copy batch: copy the long rows there while the block's bytes are in LDS
    uint64_t gather_cap;          //   (bytes; cursor = ctr[CTR_GATHER]); what it cannot take is left to k_special_gather
    uint32_t *ctr;                // the batch's counter record (CTR_*)
    unsigned long long *slot_cursor;
    unsigned long long *arena_cursors;   // main launch of a large batch: SLOT_ARENAS cursors, 64 bytes apart (NULL: the one slot_cursor)
    uint64_t arena_cap;                  // entries of the slot array per arena
    int bl, prefix_len;
};

// Which cursor a main-launch block reserves its hit lists with, where its arena starts and where it ends.
static constexpr uint32_t SLOT_ARENAS = 64;
template <typename OffT>
__device__ __forceinline__ unsigned long long *sct_slot_arena(const FilterArgs<OffT> &a, unsigned long long &base_off, uint64_t &cap_end) {
    if (a.arena_cursors == nullptr) {
        base_off = 0;
        cap_end = a.slots_cap;
        return a.slot_cursor;
    }
    const uint32_t k = blockIdx.x & (SLOT_ARENAS - 1u);
    base_off = (unsigned long long)k * a.arena_cap;
    cap_end = base_off + a.arena_cap;
    return a.arena_cursors + 8u * k;
}

// k_special_gather: the listed rows of more than PLANE_MAX characters that the main launch did not copy itself (filter_rows, COOP:
// it copies them from its LDS stage; not the rows of blocks that staged per wave), copied from where the caller has them (page-locked
// host memory of a zero-copy batch) into the gather buffer on the device, one wave per row, whole 16-byte pieces.  The special
// launch's window lanes read 4-byte words of their own: over the link that took four times the filtering itself (1 % rows of 80
// characters: 2.1 of 7.1 ms per 10 M rows).  This kernel's row-sized reads are bound by the link's outstanding requests — 12 ns a
// row whatever its length —, which is why the main launch takes the rows from its own streaming pass where it can.  A row that
// finds no room keeps SCT_REC_NONE.
template <typename OffT>
__global__ void __launch_bounds__(256) k_special_gather(const uint8_t *__restrict__ segs, const OffT *__restrict__ offs, int64_t seg_bytes,
                                                        const int32_t *__restrict__ rows, const uint32_t *__restrict__ n_dev, SctSpecialRec *__restrict__ rec,
                                                        uint32_t rec_cap, uint8_t *__restrict__ gather, uint64_t gather_cap, uint32_t *__restrict__ ctr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int64_t n = (int64_t)*n_dev;
    if (n > (int64_t)rec_cap) n = rec_cap;
    for (int64_t k = wave; k < n; k += n_waves) {
        const int64_t r = rows[k];
        int64_t start, len;
        sct_row_span(offs, r, seg_bytes, start, len);
        if (len <= PLANE_MAX) continue;   // (a row of one window: the listing launch left its planes)
        if (rec[k].len == SCT_REC_GATHERED) continue;   // (the listing launch copied the row while it had it in LDS)
        const uint64_t a0 = ((uint64_t)(uintptr_t)segs + (uint64_t)start) & ~15ull, a1 = (uint64_t)(uintptr_t)segs + (uint64_t)(start + len);
        const uint64_t bytes = (a1 - a0 + 15) & ~15ull;
        unsigned long long off = 0;
        if (lane == 0) off = atomicAdd(reinterpret_cast<unsigned long long *>(ctr + CTR_GATHER), (unsigned long long)bytes);
        off = __shfl(off, 0, 64);
        if (off + bytes > gather_cap) continue;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(1))) u4 *src = (const __attribute__((address_space(1))) u4 *)(uintptr_t)a0;
        for (uint64_t i = (uint64_t)lane; i < bytes / 16; i += 64) *reinterpret_cast<u4 *>(gather + off + 16 * i) = __builtin_nontemporal_load(src + i);
        if (lane == 0) rec[k] = SctSpecialRec{off + (((uint64_t)(uintptr_t)segs + (uint64_t)start) - a0), 0, 0, SCT_REC_GATHERED};
    }
}

// One wave's RPW rows, starting at row (or list position) `first`.
// SCHEME: 0 = the six quarter families, 1 = thirds (max_err <= 1), 2 = wide (max_err = 2); sct_core.h
static constexpr int SCHEME_QUARTERS = 0, SCHEME_THIRDS = 1, SCHEME_WIDE = 2;
// COOP (main launch of the 32-bit instantiation, four waves per block): the block stages its 256 rows' bytes together
// (sct_block_stage_load) in `block_lds`, the LDS of all four waves, which nothing else uses yet; `block_flags` = four bytes of LDS.
// Every wave of the block must make the call, whether or not it has rows (the function synchronises the block).
//
// SP = the SPECIAL launch (a third launch over the rows the main launch listed in list_special; listed rows, RPW rows per wave as
// in the retry launch): the rows the planes of ONE lane cannot express as the reference sees them —
//   * rows with a byte outside ACGT while the reference's prefix length k > 0: its window starts are restricted
//     (scTagger.py:681; sct_core.h "Restricted starts"): every lane carries the start masks A_0..A_2 of its text;
//   * rows of more than PLANE_MAX characters: cut into WINDOWS of PLANE_MAX characters, `step` = PLANE_MAX - (bl + E - 1) apart (an
//     alignment within E edits spans at most bl + E characters, so it lies inside the window that owns its start: the window's
//     first `step` characters, all of them in the row's last window).  Lane = window * RPW + row: the 64 / RPW lanes of a row
//     look up and verify their windows like 64 / RPW segments and share the row's minimum and hit list (FilterWave::row_of).
// Both at once is the general case: a window's allowed starts are its owned starts that the prefix rule allows.  Rows of more than
// 64 / RPW windows (753 characters at bl = 16, E = 2) and rows whose lists overflow go on to list_long -> the scan kernels.
template <int E, typename W, int SCHEME, int KS, int HM, int RPW, bool RT, bool SL, typename OffT, bool COOP = false, bool SP = false>
__device__ __forceinline__ void filter_rows(const FilterArgs<OffT> &a, FilterWaveLds<W, HM, QueueCap<SL, HM, W>::value, SCHEME == SCHEME_WIDE, SP> &L, int64_t first,
                                            int64_t n, uint32_t *block_lds = nullptr, int block_lds_bytes = 0, uint8_t *block_flags = nullptr) {
    static_assert(!SP || (RPW < 64 && !COOP), "the special launch works on listed rows, several lanes per row");
    const int bl = RT ? a.bl : 16;
    const int lane = threadIdx.x & 63;
    const int win = SP ? lane / RPW : 0;                       // SP: which window of its row this lane is
    const int64_t gid = first + (SP ? (lane & (RPW - 1)) : lane);
    const bool inrange = (SP || lane < RPW) && gid < n;
    int64_t r = gid;
    if (inrange && a.rows) r = a.rows[gid];  // a retried row: listed explicitly
    int64_t start = 0, len64 = 0;
    if (inrange) sct_row_span(a.offs, r, a.seg_bytes, start, len64);
    const int len_all = (int)(len64 < 0x7FFFFFFF ? len64 : 0x7FFFFFFF);
    // rows the filter can take by their length: at least bl - E characters (shorter ones cannot match anything: every
    // window the reference inserts is at least that long, scTagger.py:681) and at most PLANE_MAX
    bool fits = inrange && len_all >= bl - E && len_all <= PLANE_MAX;
    int len = fits ? len_all : 0;
    // SP: the lane's window of the row
    const int sp_ov = bl + E - 1, sp_step = PLANE_MAX - sp_ov;
    const int nwin = len_all <= PLANE_MAX ? 1 : (len_all - sp_ov + sp_step - 1) / sp_step;
    const bool toolong = SP && inrange && nwin > 64 / RPW;
    [[maybe_unused]] const int64_t row_start = start;
    if constexpr (SP) {
        const int w0 = win * sp_step;
        const int wlen = (inrange && win < nwin && !toolong) ? (len_all - w0 < PLANE_MAX ? len_all - w0 : PLANE_MAX) : 0;
        fits = wlen >= bl - E;
        len = fits ? wlen : 0;
        start += w0;
    }
    const int maxlen = wave_max_i(len);
    // W = uint32_t handles the waves whose longest such row has at most 31 characters (waves without any included:
    // their rows still have to be routed), W = uint64_t the others; a wave leaves at once if it is not its kind
    const bool mykind = (maxlen <= 31) == (sizeof(W) == 4);
    const bool anyrow = __ballot(inrange) != 0;
    W lo, hi, badmask = 0;
    bool decoded = false, high_checked = false;
    [[maybe_unused]] uint64_t gathered_at = ~0ull;   // COOP: where this lane's long row went in the gather buffer
    if constexpr (COOP) {
        static_assert(RPW == 64, "the cooperative stage belongs to the main launch");
        const int widx = (int)(threadIdx.x >> 6);
        if (lane == 0) block_flags[widx] = !anyrow ? 2 : (mykind ? 1 : 0);
        __syncthreads();
        const uint32_t f4 = *reinterpret_cast<const uint32_t *>(block_flags);
        // every wave of the block is of this kind (or has no rows): one load of the block's stretch instead of four overlapping ones
        if ((f4 & 0xFFu) != 0u && (f4 & 0xFF00u) != 0u && (f4 & 0xFF0000u) != 0u && (f4 & 0xFF000000u) != 0u) {
            const int64_t bf = first - 64 * widx;
            const int64_t be = bf + 256 < n ? bf + 256 : n;
            int64_t A = (int64_t)a.offs[bf], B = (int64_t)a.offs[be];
            A = A < 0 ? 0 : (A > a.seg_bytes ? a.seg_bytes : A);
            B = B < A ? A : (B > a.seg_bytes ? a.seg_bytes : B);
            if (B > A && B - A + 256 <= (int64_t)block_lds_bytes) {   // block-uniform (whole lines: up to 127 bytes more at either end)
                sct_block_stage_load(a.segs, A, B, block_lds);
                __syncthreads();
                // (a row outside the block's stretch means an inconsistent index: the call fails on CTR_BAD_ROW; it must not read past the stage)
                const int dl = (start >= A && start + len <= B) ? len : 0;
                const uint32_t so = sct_decode_row_from_block_stage<W>(a.segs, A, start, dl, block_lds, lo, hi, badmask);
                // rare: a byte outside ACGT — is it >= 128?  Looked up while the row's bytes are still in LDS (from the caller's
                // page-locked buffer every byte load is a round trip over the link: k_filter 4.03 -> 5.23 ms with 10 % such rows)
                if (__ballot(badmask != 0) != 0) {
                    if (badmask != 0) sct_report_high_bytes(reinterpret_cast<const uint8_t *>(block_lds) + so, dl, (uint64_t)badmask, start, bl, E, a.prefix_len, a.ctr);
                }
                high_checked = true;
                // a zero-copy batch: rows of more than PLANE_MAX characters go to the special launch, whose window lanes would fetch
                // them over the link once more, row by row — 12 ns a row whatever its length (the link's outstanding reads), 1.2 ms
                // for 1 % of 10 M rows.  Their bytes are in LDS now: the wave copies them to the device's gather buffer, word by word.
                if (a.gather_out != nullptr) {
                    const bool lg = inrange && len_all > PLANE_MAX && start >= A && start + (int64_t)len_all <= B;
                    uint64_t todo = __ballot(lg);
                    const uint64_t base = ((uint64_t)(uintptr_t)a.segs + (uint64_t)A) & ~127ull;
                    const uint32_t so_mine = lg ? (uint32_t)((uint64_t)(uintptr_t)a.segs + (uint64_t)start - base) : 0u;
                    while (todo != 0) {
                        const int src = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const uint32_t so_r = (uint32_t)__shfl((int)so_mine, src, 64);
                        const int ln = __shfl(len_all, src, 64);
                        const int nw = (int)(((so_r & 3u) + (uint32_t)ln + 3u) >> 2);
                        const uint64_t pad = ((uint64_t)nw * 4 + 15) & ~15ull;
                        unsigned long long off = 0;
                        if (lane == 0) off = atomicAdd(reinterpret_cast<unsigned long long *>(a.ctr + CTR_GATHER), (unsigned long long)pad);
                        off = __shfl(off, 0, 64);
                        if (off + pad > a.gather_cap) continue;
                        uint32_t *dst = reinterpret_cast<uint32_t *>(a.gather_out + off);
                        const uint32_t *wp = block_lds + (so_r >> 2);
                        for (int i = lane; i < nw; i += 64) dst[i] = wp[i];
                        if (lane == src) gathered_at = off + (so_r & 3u);
                    }
                }
                __syncthreads();   // the stage is every wave's working LDS from here on
                decoded = true;
            }
        }
    }
    if constexpr (COOP) {
        if (anyrow && !mykind && lane == 0 && a.skip64_unless_needed) a.ctr[CTR_NEED64] = 1u;
        // ... and leaves the rows without hits for now: the 64-bit instantiation behind this launch writes their counts — or nobody
        // does, when the batch was queued without it on the guess that no row needs it (sct_ctx::need64); the batch is then done again,
        // but its sums and its fill kernel run first, and what they read must not be whatever the buffer held before (round 5: a
        // fresh context's counts were a previous context's, and k_offsets_fill ran off the end of the slot array)
        if (!mykind && inrange && a.skip64_unless_needed) a.cnt[r] = 0u;
    }
    if (!mykind) return;
    if (!anyrow) return;
    [[maybe_unused]] bool has_rec = false;
    [[maybe_unused]] SctSpecialRec rec{0, 0, 0, SCT_REC_NONE};
    [[maybe_unused]] const uint8_t *row_bytes = a.segs + row_start;
    if constexpr (SP) {
        if (toolong && win == 0) {   // more windows than the row has lanes: the next stage, or the scan kernels (via k_route_long).
            const uint32_t k = atomicAdd(a.defer_count, 1u);            // (Such a row adds nothing to the wave's longest text, so
            a.list_defer[k] = (int32_t)r;                               // exactly one of the two mask widths gets here with it.)
            if (a.rec_out != nullptr && k < a.rec_cap)                  // its record (where its bytes were gathered) goes along
                a.rec_out[k] = (a.rec_in != nullptr && (uint64_t)gid < (uint64_t)a.rec_cap) ? a.rec_in[gid] : SctSpecialRec{0, 0, 0, SCT_REC_NONE};
            a.cnt[r] = CNT_DEFER;
        }
        // a row of one window: the listing launch left its planes; longer rows are decoded from their bytes
        // what the listing launch left about the row: its planes (one window), or where its bytes were gathered (longer rows)
        if (a.rec_in != nullptr && inrange && (uint64_t)gid < (uint64_t)a.rec_cap) rec = a.rec_in[gid];
        const bool gathered = rec.len == SCT_REC_GATHERED && a.gather != nullptr && len_all > PLANE_MAX;
        has_rec = rec.len <= (uint64_t)PLANE_MAX && win == 0 && len_all <= PLANE_MAX;
        row_bytes = gathered ? a.gather + rec.lo : a.segs + row_start;
        sct_decode_row_mask<W>(row_bytes, (int64_t)win * sp_step, has_rec ? 0 : len, lo, hi, badmask);
        if (has_rec) {
            lo = len > 0 ? (W)rec.lo : (W)0;
            hi = len > 0 ? (W)rec.hi : (W)0;
            badmask = len > 0 ? (W)rec.bad : (W)0;
        }
        decoded = true;
    }
    [[maybe_unused]] const W lo0 = lo, hi0 = hi, bad0 = badmask;
    if (!decoded) {
        using Lds = FilterWaveLds<W, HM, QueueCap<SL, HM, W>::value, SCHEME == SCHEME_WIDE, SP>;
        static_assert(__builtin_offsetof(Lds, qk) == __builtin_offsetof(Lds, hlist) + 4 * 64 * HM &&
                          __builtin_offsetof(Lds, qm) == __builtin_offsetof(Lds, qk) + 4 * Lds::QCAP && __builtin_offsetof(Lds, hlist) % 16 == 0,
                      "hlist, qk and qm form the decode stage");
        constexpr int STAGE_BYTES = 4 * (64 * HM + 2 * Lds::QCAP);
        // all rows: one coalesced pass over the wave's stretch of bytes; listed rows lie anywhere: word by word
        if (a.rows != nullptr || !sct_decode_rows_staged<W>(a.segs, start, len, &L.hlist[0][0], STAGE_BYTES, lo, hi, badmask))
            sct_decode_row_mask<W>(a.segs, start, len, lo, hi, badmask);
    }
    const bool nonacgt = badmask != 0;
    if (!high_checked && __ballot(nonacgt) != 0) {
        // rare: is one of the offending bytes >= 128?  The reference raises when it reads one (sct_byte_is_read, sct_core.h)
        if constexpr (SP) {
            // rows of at most PLANE_MAX characters were checked by the main launch that listed them; of a longer row every window
            // checks the bytes it owns, against the ROW (a byte is read iff some window the reference inserts covers it)
            if (nonacgt && len_all > PLANE_MAX) {
                const int own = win == nwin - 1 ? len : (len < sp_step ? len : sp_step);
                uint64_t bad = (uint64_t)badmask & (own >= 64 ? ~0ull : ((1ull << own) - 1ull));
                while (bad != 0) {
                    const int i = __builtin_ctzll(bad);
                    bad &= bad - 1;
                    if (row_bytes[(int64_t)win * sp_step + i] >= 128 && sct_byte_is_read(row_bytes, len_all, (int64_t)win * sp_step + i, bl, E, a.prefix_len)) {
                        atomicMin(reinterpret_cast<unsigned long long *>(a.ctr + CTR_BAD_BYTE), (unsigned long long)(start + i));
                        break;
                    }
                }
            }
        } else if (nonacgt) {
            sct_report_high_bytes(a.segs + start, len, (uint64_t)badmask, start, bl, E, a.prefix_len, a.ctr);
        }
    }
    bool active = fits;
    // all 64 rows of the wave exist and lie side by side: their min_e bytes go out as one 64-byte line
    const bool line = a.rows == nullptr && first + 64 <= n && (((uintptr_t)(a.min_e + first)) & 15) == 0 &&
                      (a.cnt8 == nullptr || (((uintptr_t)(a.cnt8 + first)) & 15) == 0);
    if (a.rows == nullptr && inrange) {
        // routing (main launch): what k_pack did in round 1
        if (!line && a.cnt8 && (!fits || (nonacgt && a.prefix_len > 0))) a.cnt8[r] = 0;  // with `line` the wave's store covers them
        if (!fits) {
            if (len_all > PLANE_MAX) {  // more characters than one lane's planes hold: the special launch cuts the row into windows
                const uint32_t k = atomicAdd(&a.ctr[CTR_SPECIAL], 1u);
                a.list_special[k] = (int32_t)r;
                a.ctr[CTR_NEED64] = 1u;   // ... of PLANE_MAX characters: they belong to the special launch's 64-bit instantiation
                if (a.rec_out != nullptr && k < a.rec_cap)
                    a.rec_out[k] = gathered_at != ~0ull ? SctSpecialRec{gathered_at, 0, 0, SCT_REC_GATHERED} : SctSpecialRec{0, 0, 0, SCT_REC_NONE};
                a.cnt[r] = CNT_DEFER;
            } else {
                if (!line) a.min_e[r] = SCT_NO_HIT;
                a.cnt[r] = 0;
            }
        } else if (nonacgt && a.prefix_len > 0) {
            // the reference drops every window whose first k raw bytes are not all in ACGT (scTagger.py:681): the special launch
            // runs the same families with restricted starts
            const uint32_t k = atomicAdd(&a.ctr[CTR_SPECIAL], 1u);
            a.list_special[k] = (int32_t)r;
            if (a.rec_out != nullptr && k < a.rec_cap) a.rec_out[k] = SctSpecialRec{(uint64_t)lo, (uint64_t)hi, (uint64_t)badmask, (uint64_t)len};
            a.cnt[r] = CNT_DEFER;
            active = false;
        }
    }
    if (!active) {
        len = 0;
        lo = hi = 0;
    }
    if (__ballot(active) == 0) {  // the whole wave has nothing to filter
        if (line) {
            store_byte_line(a.min_e, first, SCT_NO_HIT);
            if (a.cnt8) store_byte_line(a.cnt8, first, 0u);
        }
        return;
    }
    const W va = sct_valid_mask<W>(len);
    const W ve = va & ~(W)1, l1 = lo << 1, h1 = hi << 1;
    L.em[lane][0] = ~l1 & ~h1 & ve;
    L.em[lane][1] = l1 & ~h1 & ve;
    L.em[lane][2] = ~l1 & h1 & ve;
    L.em[lane][3] = l1 & h1 & ve;
    if constexpr (sizeof(W) == 4) L.va[lane] = va;
    if constexpr (SP) {
        // allowed starts of this window: the prefix rule (k = 0, or no byte outside ACGT: every position) on the starts it owns.
        // (s + k <= the ROW's length: in a window that is not the last, an owned start has sp_ov >= k more characters in the window)
        W a0 = sct_allowed_starts<W>(badmask, len, a.prefix_len);
        if constexpr (sizeof(W) == 8) {   // (32-bit masks: rows of at most 31 characters, one window)
            if (win < nwin - 1) a0 &= ((W)1 << sp_step) - 1;
        }
        SctStarts<W> st;
        st.from_allowed(a0, va);
        L.al[lane][0] = st.a[0];
        L.al[lane][1] = st.a[1];
        L.al[lane][2] = st.a[2];
    }
    L.pl[lane].lo = lo;
    L.pl[lane].hi = hi;
    L.slen[lane] = (uint8_t)len;
    L.best[lane] = E + 1;
    L.hcnt[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    constexpr bool THIRDS = SCHEME == SCHEME_THIRDS, WIDE = SCHEME == SCHEME_WIDE;
    static_assert(!(SL && (SCHEME != SCHEME_QUARTERS || RT)), "bucket slots exist for the 16-character quarter families only");
    static_assert(!WIDE || (E == 2 && !RT), "the wide families are those of max_err = 2 and 16-character barcodes");
    FilterWave<E, W, HM, SL, RPW, WIDE, THIRDS, SP> W_{L, a.tab_off, a.tab_ent, a.tab_code, a.tab_slot, a.tab_grp, a.tab_ent16, a.tab_fb, a.tab_wfb, a.ft, lane, active, len, lo, hi};
#define RUN_FAMILY(FID)                                                         \
    do {                                                                        \
        if constexpr (RT) W_.run_family(sct_family_rt(FID, bl), maxlen);        \
        else W_.run_family(SctFamily<FID>{}, maxlen);                           \
    } while (0)
    if constexpr (WIDE) {
      if constexpr (RPW == 64) {   // main launch: fixed buckets (bound by the table reads); retry launch: offsets + entries
        W_.template run_family_wide_fb<0>(maxlen);
        W_.settle(0);
        W_.template run_family_wide_fb<1>(maxlen);
        W_.template run_family_wide_fb<3>(maxlen);
        W_.template run_family_wide_fb<4>(maxlen);
        W_.settle(1);
        W_.give_up();
        W_.template run_family_wide_fb<2>(maxlen);
        W_.give_up();
        W_.template run_family_wide_fb<5>(maxlen);
        W_.give_up();
        W_.template run_family_wide_fb<6>(maxlen);
        W_.template run_family_wide_fb<7>(maxlen);
        W_.template run_family_wide_fb<8>(maxlen);
        W_.give_up();
        W_.template run_family_wide_fb<9>(maxlen);
        W_.template run_family_wide_fb<10>(maxlen);
        W_.give_up();
        W_.template run_family_wide_fb<11>(maxlen);
        W_.template run_family_wide_fb<12>(maxlen);
        W_.give_up();
        W_.template run_family_wide_fb<13>(maxlen);
        W_.template run_family_wide_fb<14>(maxlen);
        W_.template run_family_wide_fb<15>(maxlen);
      } else {
        W_.template run_family_wide<0>(maxlen);
        W_.settle(0);
        W_.template run_family_wide<1>(maxlen);
        W_.template run_family_wide<3>(maxlen);
        W_.template run_family_wide<4>(maxlen);
        W_.settle(1);
        W_.give_up();
        W_.template run_family_wide<2>(maxlen);
        W_.give_up();
        W_.template run_family_wide<5>(maxlen);
        W_.give_up();
        W_.template run_family_wide<6>(maxlen);
        W_.template run_family_wide<7>(maxlen);
        W_.template run_family_wide<8>(maxlen);
        W_.give_up();
        W_.template run_family_wide<9>(maxlen);
        W_.template run_family_wide<10>(maxlen);
        W_.give_up();
        W_.template run_family_wide<11>(maxlen);
        W_.template run_family_wide<12>(maxlen);
        W_.give_up();
        W_.template run_family_wide<13>(maxlen);
        W_.template run_family_wide<14>(maxlen);
        W_.template run_family_wide<15>(maxlen);
      }
    } else if constexpr (THIRDS) {
        if (a.tab_fb != nullptr) {
            W_.run_family_fb(SctFamily<SCT_F_T01>{}, maxlen);
            if (E >= 1) {
                W_.settle(0);
                W_.run_family_fb(SctFamily<SCT_F_T12>{}, maxlen);
                W_.run_family_fb(SctFamily<SCT_F_T02>{}, maxlen);
            }
        } else {
            RUN_FAMILY(SCT_F_T01);
            if (E >= 1) {
                W_.settle(0);
                RUN_FAMILY(SCT_F_T12);
                RUN_FAMILY(SCT_F_T02);
            }
        }
    } else {
        RUN_FAMILY(SCT_F_H1);
        if (E >= 1) {
            W_.settle(0);
            RUN_FAMILY(SCT_F_H2);
        }
        if (E >= 2) {
            W_.settle(1);
            W_.give_up();
            RUN_FAMILY(SCT_F_MID);
            W_.give_up();
            RUN_FAMILY(SCT_F_02);
            W_.give_up();
            RUN_FAMILY(SCT_F_13);
            W_.give_up();
            RUN_FAMILY(SCT_F_03);
        }
    }
#undef RUN_FAMILY
    // Locators -> pattern ids, in place, by all lanes of the wave: with RPW rows per wave, 64 / RPW lanes share a row's
    // list, up to four table reads in flight per lane.
    using FW = FilterWave<E, W, HM, SL, RPW, WIDE, THIRDS, SP>;
    constexpr uint32_t CAP = (uint32_t)FW::CAP;
    {
        constexpr int PARTS = 64 / RPW;
        const int row = lane % RPW, part = lane / RPW;
        const uint32_t c_row = L.hcnt[row] & ~FW::HCNT_OVERFLOW;
        const uint32_t n_row = c_row < CAP ? c_row : CAP;  // 0 for rows without work
        for (uint32_t k0 = (uint32_t)part; k0 < n_row; k0 += 4 * PARTS) {
            uint32_t v[4], pid[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = k0 + i * PARTS < n_row ? W_.hl(k0 + i * PARTS, (uint32_t)row) : 0u;
#pragma unroll
            for (int i = 0; i < 4; i++) pid[i] = k0 + i * PARTS < n_row ? W_.resolve(v[i] & LOC_MASK) : 0u;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (k0 + i * PARTS < n_row) W_.hl(k0 + i * PARTS, (uint32_t)row) = (v[i] & ~LOC_MASK) | pid[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the lanes that own a row's result: every lane with a row (main, retry), the lane of the row's first window (special)
    const bool ractive = SP ? (lane < RPW && inrange && !toolong) : active;
    const uint32_t hc = ractive ? L.hcnt[lane] : 0u;
    const uint32_t nc = hc & ~FW::HCNT_OVERFLOW;
    if constexpr (RPW == 64) {
        // main launch: one lane per row folds its (short) list into minimum + distinct pattern ids in registers
        LaneResult<E, KS> res;
        res.init();
        const uint32_t lim = nc < CAP ? nc : CAP;
        for (uint32_t k = 0; k < lim; k++) {
            const uint32_t v = W_.hl(k, (uint32_t)lane);
            res.add((int)(v >> 30), v & LOC_MASK);
        }
        unsigned long long arena_off;
        uint64_t arena_end;
        unsigned long long *cursor = sct_slot_arena(a, arena_off, arena_end);
        res.store(active, r, SL ? nc > CAP : FW::overflowed(hc, L.best[lane]), line, a.min_e, a.cnt8, a.cnt, a.slots, arena_end, a.wave_base, first,
                  a.list_defer, a.defer_count, cursor, arena_off);
    } else {
        // Retry launch: lists of up to CAP entries per row (hundreds of ties at a few million barcodes and two edits) are
        // folded by the whole wave, row after row, in LDS: entries that are not at the row's minimum are struck, then every entry
        // that has an equal one before it, then every remaining entry counts the smaller ones — its place in the sorted,
        // duplicate-free list.  The lists go to the slot array at (row's share of) a block reserved with one atomic per wave.
        constexpr int PER = (int)((CAP + 63) / 64);
        constexpr uint32_t SENT = 0xFFFFFFFFu;
        const bool ovf = SL ? nc > CAP : FW::overflowed(hc, L.best[lane]);
        const uint32_t n_own = (ractive && !ovf) ? (nc < CAP ? nc : CAP) : 0u;
        // pass A, row after row: strike what is not at the row's minimum, strike duplicates, count what is left
        uint32_t my_cnt = 0;
        for (int row = 0; row < RPW; row++) {
            const uint32_t n = __builtin_amdgcn_readlane(n_own, row);
            if (n == 0) continue;
            const uint32_t best = L.best[row];
            uint32_t val[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)i;
                uint32_t x = SENT;
                if (idx < n) {
                    const uint32_t e = W_.hl(idx, (uint32_t)row);
                    x = (e >> 30) == best ? (e & LOC_MASK) : SENT;
                    W_.hl(idx, (uint32_t)row) = x;
                }
                val[i] = x;
            }
            __builtin_amdgcn_wave_barrier();
            bool dup[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) dup[i] = false;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t y = W_.hl(j, (uint32_t)row);  // the same address for every lane: one broadcast read
#pragma unroll
                for (int i = 0; i < PER; i++) dup[i] |= (y == val[i]) && (j < (uint32_t)lane + 64u * (uint32_t)i);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)i;
                if (val[i] != SENT) {
                    if (dup[i]) W_.hl(idx, (uint32_t)row) = SENT;
                    else mine++;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
            if (lane == row) my_cnt = mine;
            __builtin_amdgcn_wave_barrier();
        }
        // room in the slot array for exactly what the wave's rows hold: one atomic per wave
        uint32_t inc = my_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const uint32_t total = __shfl(inc, 63, 64);
        unsigned long long base = 0;
        if (lane == 0 && total) base = atomicAdd(a.slot_cursor, (unsigned long long)total);
        base = __shfl(base, 0, 64);
        const bool no_room = base + total > a.slots_cap;  // wave-uniform
        if (ractive && (ovf || no_room)) {
            const uint32_t k = atomicAdd(a.defer_count, 1u);
            a.list_defer[k] = (int32_t)r;
            if constexpr (SP) {   // first stage of the special launch: the planes go along with the row
                if (a.rec_out != nullptr && k < a.rec_cap)
                    a.rec_out[k] = has_rec ? SctSpecialRec{(uint64_t)lo0, (uint64_t)hi0, (uint64_t)bad0, (uint64_t)len_all}
                                           : (rec.len == SCT_REC_GATHERED ? rec : SctSpecialRec{0, 0, 0, SCT_REC_NONE});
            }
            a.cnt[r] = CNT_DEFER;
        }
        const uint32_t start_own = inc - my_cnt;
        // pass B: every entry that is left counts the smaller ones — its place in the row's sorted list
        for (int row = 0; row < RPW && !no_room; row++) {
            const uint32_t n = __builtin_amdgcn_readlane(n_own, row);
            if (n == 0 || __builtin_amdgcn_readlane(my_cnt, row) == 0) continue;
            const unsigned long long start = base + __builtin_amdgcn_readlane(start_own, row);
            uint32_t val[PER], rank[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)i;
                val[i] = idx < n ? W_.hl(idx, (uint32_t)row) : SENT;
                rank[i] = 0;
            }
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t y = W_.hl(j, (uint32_t)row);
#pragma unroll
                for (int i = 0; i < PER; i++) rank[i] += (y < val[i]) ? 1u : 0u;
            }
#pragma unroll
            for (int i = 0; i < PER; i++)
                if (val[i] != SENT) a.slots[start + rank[i]] = val[i];
        }
        if (ractive && !ovf && !no_room) {
            a.cnt[r] = my_cnt | CNT_RETRIED;
            a.min_e[r] = my_cnt ? (uint8_t)L.best[lane] : (uint8_t)SCT_NO_HIT;
            if (my_cnt) a.slot_off[r] = base + start_own;
            if (a.cnt8) {
                a.cnt8[r] = (uint8_t)(my_cnt < 255u ? my_cnt : 255u);
                if (my_cnt > 255u) atomicOr(&a.ctr[CTR_CNT_OVERFLOW], 1u);
            }
        }
    }
}

// RPW rows per wave: 64 in the main launch; the retry launch gives a wave only 4 rows, so that the pair compaction of
// run_family spreads each row's lookups over four lanes and the few retried rows finish sooner.  The waves
// of the retry launch — whose row count only exists on the device when it is queued — and of the 64-bit instantiation —
// which usually has nothing to do — come as a small fixed grid and stride over the rows.
// Waves per workgroup: four for the main launch's 32-bit instantiation; ONE for the striding launches.  Their blocks then
// need a quarter of the LDS (10 - 20 KB instead of 40 - 80 KB) and fit into whatever a retiring block of another chunk's main
// launch frees on a CU; as four-wave blocks they waited until that other grid had been dispatched to its last block
// (profiles/r02_stream_timeline_zero_copy_3lanes.txt: 0.9 ms for a launch that takes 10 us alone).
template <typename W, int RPW>
struct FilterWavesPerBlock {
    static constexpr int value = (RPW == 64 && sizeof(W) == 4) ? 4 : 1;
};

// The retry launch without slots waits for table reads at one-wave blocks: four waves per SIMD (at most 128 VGPRs) instead of the
// three its 160-odd registers would allow.
template <int E, typename W, int SCHEME, int KS, int HM, int RPW, bool RT, bool SL, typename OffT, bool SP = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((RPW < 64 && !SL && sizeof(W) == 4) ? 4 : 1)))
k_filter(const FilterArgs<OffT> a) {
    constexpr int WPB = FilterWavesPerBlock<W, RPW>::value;
    using Lds = FilterWaveLds<W, HM, QueueCap<SL, HM, W>::value, SCHEME == SCHEME_WIDE, SP>;
    __shared__ Lds lds[WPB];
    const int64_t n = a.n_dev ? (int64_t)*a.n_dev : a.n;
    const int64_t first = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * RPW;
    if constexpr (WPB == 4) {
        // the main launch's 32-bit instantiation — where the time goes — is launched over all rows and stays loop-free:
        // wrapped in the loop below the compiler hoists enough address arithmetic out of it to cost two waves per SIMD.
        // Its four waves stage the block's rows together (COOP): every wave makes the call, rows or not.
        __shared__ uint32_t block_flags;
        filter_rows<E, W, SCHEME, KS, HM, RPW, RT, SL, OffT, true>(a, lds[threadIdx.x >> 6], first, n, reinterpret_cast<uint32_t *>(&lds[0]),
                                                                   (int)sizeof(lds), reinterpret_cast<uint8_t *>(&block_flags));
    } else {
        if constexpr (RPW == 64 && sizeof(W) == 8 && !SP) {
            // the main launch's 64-bit instantiation behind its 32-bit one: 42 us for a pass over every row's length that finds
            // nothing to do when all rows have at most 31 characters (the BASELINE shapes)
            if (a.skip64_unless_needed && a.ctr[CTR_NEED64] == 0u) return;
        }
        const int64_t step = (int64_t)gridDim.x * WPB * RPW;
        for (int64_t f = first; f < n; f += step) {
            filter_rows<E, W, SCHEME, KS, HM, RPW, RT, SL, OffT, false, SP>(a, lds[0], f, n);
            __builtin_amdgcn_wave_barrier();
        }
    }
}
