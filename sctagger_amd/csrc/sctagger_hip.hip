// sctagger_hip.hip — gfx950 kernels and the C-ABI (include/sctagger_hip.h).
//
// match_trie (reference scTagger.py:643-716, observable semantics in SURVEY.md §8(a)-Q): for every long-read
// segment the minimum infix edit distance (<= max_err) to any barcode on either strand, and the sorted list of
// all (bid, strand) at that minimum.  The reference gets there with a trie of segment windows and a recursive
// DFS per barcode; none of that is imitated here.
//
// Device data (all in HBM, resident for the life of the context / upload):
//   patterns   np = 2*n_bc entries, index p = bid*2 + strand (strand 1 = forward, 0 = reverse complement, so
//              ascending p is the reference's sorted(bids) order), stored as bit-planes uint2{lo, hi}
//   seed tables (bl == 16): per family an offsets array (65537 entries; 2^20+1 / 2^22+1 for the thirds scheme of
//              large barcode sets) and np entries as pattern ids and as 2-bit codes, see sct_core.h
//   segments   the caller's CSR bytes + int64 offsets, plus per segment the packed bit-planes {lo, hi, meta}
//   results    min_e u8[n], cnt u32[n], slot_off u64[n] + dense slots u32[] (filter path), hit_offs i64[n+1],
//              hits u32[total]
//
// Kernels of the match_trie path:
//   k_pack          ASCII -> bit-planes, routes each segment to the filter / exhaustive / restricted-start path
//   k_filter<...>   one wave per 64 segments: seed lookups spread over the lanes, candidates queued in LDS and
//                   verified 64 at a time by the anchored transposed Wu-Manber automaton, tiers, retry launch
//   k_filter_lane   the first, lane-per-segment version of the filter (option filter_mode = 1, for A/B)
//   k_scan<...>     one wave per (segment, barcode slice), lanes = barcodes: Myers bit-vector over every barcode
//                   (any bl <= 32, any max_err) or, for segments with non-ACGT bytes when the reference's prefix
//                   length k > 0, a plain DP whose row 0 encodes the allowed window starts; count + fill pass
//   k_sum_* / k_offsets   exclusive scan of the per-segment hit counts into the CSR offsets
//   k_fill          copies the filter path's hit lists into the CSR hit array
// Kernels of the two neighbouring stages:
//   k_wl_count                        extract_sr_bc_from_lr: whitelist dictionary probes (scTagger.py:818-826)
//   k_adapter_count / k_adapter_fill  extract_lr_bc: adapter alignment over whole reads (scTagger.py:176-196)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sctagger_hip.h"
#include "sct_core.h"

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(SCT_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static constexpr int K_SLOTS = 8;          // hits kept per segment by the filter kernel before it defers
static constexpr uint32_t CNT_DEFER = 0xFFFFFFFFu;
static constexpr int THIRDS_MIN_PATTERNS = 1 << 19;  // from here on (8 barcodes per 16-bit bucket) the thirds scheme pays
static constexpr int META_NONACGT = 1 << 8;
static constexpr int META_LONG = 1 << 9;   // more than 63 characters: not representable in the filter's planes
static constexpr int PLANE_MAX = 63;       // characters kept in the packed bit-planes

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

// ------------------------------------------------------------------------------------------------------------
// Per-lane result of the filter kernels: running minimum distance and up to K_SLOTS distinct pattern ids at
// that minimum, in registers (static indexing only).  More than K_SLOTS ties defer the segment to the scan path.
// ------------------------------------------------------------------------------------------------------------
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
    // of the list lengths and one 64-bit atomic per wave reserve the space, slot_off[r] remembers where row r's
    // list starts.  Rows with too many ties are pushed to the scan list instead.
    __device__ __forceinline__ void store(bool row, int64_t r, bool overflow, uint8_t *__restrict__ min_e,
                                          uint32_t *__restrict__ cnt, uint32_t *__restrict__ slots,
                                          uint64_t *__restrict__ slot_off, int32_t *__restrict__ list_defer,
                                          uint32_t *__restrict__ defer_count, unsigned long long *__restrict__ slot_cursor,
                                          uint64_t slot_tag) {
        const int lane = threadIdx.x & 63;
        const bool defer = row && (overflow || nh > K_SLOTS);
        const uint32_t mine = (row && !defer) ? (uint32_t)nh : 0u;
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const uint32_t total = __shfl(inc, 63, 64);
        unsigned long long base = 0;
        if (lane == 0 && total) base = atomicAdd(slot_cursor, (unsigned long long)total);
        base = __shfl(base, 0, 64);
        if (!row) return;
        if (defer) {
            const uint32_t k = atomicAdd(defer_count, 1u);
            list_defer[k] = (int32_t)r;
            cnt[r] = CNT_DEFER;
            return;
        }
        cnt[r] = mine;
        if (mine == 0) {
            min_e[r] = SCT_NO_HIT;
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
        min_e[r] = (uint8_t)best;
        const uint64_t start = base + inc - mine;
        slot_off[r] = start | slot_tag;
#pragma unroll
        for (int i = 0; i < K_SLOTS; i++)
            if (i < nh) slots[start + i] = h[i];
    }
};

// ------------------------------------------------------------------------------------------------------------
// k_filter_lane<E> (first version, kept for A/B: option "filter_mode" = 1): one lane per segment, each lane walks
// its own buckets and verifies its own candidates; lanes idle while the longest bucket of the wave is walked.
// ------------------------------------------------------------------------------------------------------------
template <int E>
struct FilterLane {
    const uint32_t *__restrict__ tab_off;
    const uint32_t *__restrict__ tab_ent;
    const uint2 *__restrict__ pat;
    int np;
    uint32_t tls, ths, ve, va;
    LaneResult<E> res;

    __device__ __forceinline__ void operator()(int f, uint32_t key) {
        const uint32_t *off = tab_off + (size_t)f * (SCT_KEYS + 1) + key;
        const uint32_t s = off[0], e = off[1];
        for (uint32_t k = s; k < e; k++) {  // offsets index the concatenated entry table
            const uint32_t pid = tab_ent[k];
            const uint2 p = pat[pid];
            res.add(wm_verify<E>(tls, ths, ve, va, p.x, p.y, 16), pid);
        }
    }
};

template <int E>
__global__ void __launch_bounds__(256)
k_filter_lane(const uint64_t *__restrict__ lo_in, const uint64_t *__restrict__ hi_in, const uint32_t *__restrict__ meta_in,
              int64_t n, const uint32_t *__restrict__ tab_off, const uint32_t *__restrict__ tab_ent,
              const uint2 *__restrict__ pat, int np, uint8_t *__restrict__ min_e, uint32_t *__restrict__ cnt,
              uint32_t *__restrict__ slots, uint64_t *__restrict__ slot_off, int32_t *__restrict__ list_scan,
              uint32_t *__restrict__ counters) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool active = r < n;
    if (active) active = cnt[r] != CNT_DEFER;  // routed to a scan path by k_pack
    int len = 0;
    if (active) {
        len = (int)(meta_in[r] & 0xFF);
        active = len >= 16 - E;  // shorter rows were finished by k_pack
    }
    FilterLane<E> L;
    L.res.init();
    if (active) {
        const uint32_t lo = (uint32_t)lo_in[r], hi = (uint32_t)hi_in[r];
        L.tab_off = tab_off;
        L.tab_ent = tab_ent;
        L.pat = pat;
        L.np = np;
        L.tls = lo << 1;
        L.ths = hi << 1;
        L.va = len >= 31 ? 0xFFFFFFFFu : ((1u << (len + 1)) - 1u);
        L.ve = L.va & ~1u;
        sct_for_each_seed<E, uint32_t>(lo, hi, len, L);
    }
    L.res.store(active, r, false, min_e, cnt, slots, slot_off, list_scan, counters + 0,
                reinterpret_cast<unsigned long long *>(counters + 4), 0ull);
}

// ------------------------------------------------------------------------------------------------------------
// k_filter<E> (default): one WAVE per 64 segments, cooperating through LDS so that verification always runs on
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
// A segment with more than HMAX verified matches (duplicates included) or more than K_SLOTS ties is deferred.
// All cross-lane traffic is inside one wave: LDS operations of a wave execute in issue order, so no barrier is
// needed; __builtin_amdgcn_wave_barrier() only pins the compiler's ordering.
// ------------------------------------------------------------------------------------------------------------
static constexpr int QCAP = 256;  // >= 63 left over + 64 lanes x 3 inline entries pushed in one pass
#ifndef SCT_HMAX
#define SCT_HMAX 8  // with the tiers a settled segment stops early, so few rows collect more than 8 verified matches
#endif
static constexpr int HMAX = SCT_HMAX;  // main launch
static constexpr int K_RETRY = 32, H_RETRY = 64;  // retry launch
static constexpr uint64_t SLOT_TAG_RETRY = 1ull << 63;  // slot_off values that point into the retry launch's slot buffer
static constexpr uint32_t LOC_MASK = (1u << 30) - 1u;    // locator in a hit-list entry (2 bits of distance above it)
static constexpr uint32_t LOC_INLINE = 1u << 29;         // locator of an inline slot entry (entry indices stay below it)

// where each family's offsets start inside the concatenated offset table; the offsets themselves index the
// concatenated entry tables (family f's entries start at f * np)
struct FamTables {
    uint32_t off_base[SCT_NFAM_ALL];
};

template <typename W, int HM>
struct FilterWaveLds {
    static constexpr int HMAX = HM;
    alignas(32) W em[64][4];  // em[lane][c]: bit i+1 set iff character i of the lane's segment has code c (bits 1..len)
    W va[64];            // bits 0..len
    W lo[64], hi[64];    // the segments' bit-planes (lookup lanes read the planes of the segment they work for)
    uint8_t slen[64];    // and lengths
    uint8_t alist[64];   // lanes whose segment still generates candidates, compacted
    uint16_t ptab[sizeof(W) == 4 ? 128 : 320];  // this family's placements for the longest segment: t | t2 << 8
    uint32_t best[64];   // smallest distance verified so far for the lane's segment
    uint32_t hcnt[64];
    uint32_t hlist[HM][64];
    uint32_t qk[QCAP];   // queue: global table entry index f*np + k
    uint32_t qm[QCAP];   //        lane | t << 8 | t2 << 16
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

// Bucket slots (SL = true, the quarter families): one 8-byte word per key that answers most lookups by itself.
//   bit 15 clear: bits 0..1 = n <= 3 entries in the bucket, bits 16j+16 .. 16j+31 = the REST characters of entry j
//                 (sct_rest_of: all the verification needs besides the key); no other table is read unless the
//                 entry verifies, and then its pattern id is tab_ent[off[key] + j]
//   bit 15 set:   a longer bucket: bits 16..47 = its first entry index, bits 0..14 | bits 48..63 << 15 = its length;
//                 the entries are queued by index and verified from tab_code as without slots
// A queued entry is (qk, qm): qm = lane | t << 8 | t2 << 16 | j << 24 | inline << 31; qk = rest characters (inline)
// or entry index.
static constexpr uint32_t QM_INLINE = 0x80000000u;

template <int E, typename W, int HM, bool SL>
struct FilterWave {
    static constexpr int HMAX = HM;
    FilterWaveLds<W, HM> &L;
    const uint32_t *__restrict__ tab_off;
    const uint32_t *__restrict__ tab_ent;   // pattern id of every table entry
    const uint32_t *__restrict__ tab_code;  // the same entries as 2-bit codes, character j at bits 2j..2j+1
    const uint2 *__restrict__ tab_slot;     // bucket slots, indexed like tab_off
    FamTables ft;
    int lane;
    bool active;
    int len;
    W lo, hi;
    uint32_t head = 0, tail = 0;  // wave-uniform

    template <typename Fam>
    __device__ __forceinline__ void verify(const Fam &fam, bool take) {
        if (take) {
            const uint32_t idx = (head + lane) & (QCAP - 1);
            uint32_t gk = L.qk[idx];
            const uint32_t meta = L.qm[idx];
            const uint32_t src = meta & 63u;
            const int t = (int)((meta >> 8) & 63u), t2 = (int)((meta >> 16) & 63u);
            const bool inl = SL && (meta & QM_INLINE) != 0u;
            uint32_t rest = gk;  // an inline entry carries its rest characters (bits above them are ignored)
#ifdef SCT_ABLATE_MISS  // timing-only build: no table is read after the slot (results are wrong)
            if (!inl) rest = gk * 2654435761u;
#else
            if (!inl) rest = sct_rest_of(fam, tab_code[gk]);
#endif
#ifdef SCT_ABLATE_VERIFY  // timing-only build: candidates are fetched but never run through the automaton
            asm volatile("" ::"v"(rest), "v"(t), "v"(t2));
            const int d = E + 1;
#else
            const int d = wm_anchored_rest<E, W>(fam, EmLds<W>(&L.em[src][0]), L.va[src], rest, t, t2);
#endif
            if (d <= E) {
                // what is kept is the entry's LOCATOR; the pattern ids are read once, when the segment is finished
                // (resolve), so that no table read sits between two batches
                if (inl)
                    gk = LOC_INLINE | ((ft.off_base[fam.INDEX] + sct_key_of<W>(fam, L.lo[src], L.hi[src], t, t2)) << 2) |
                         ((meta >> 24) & 3u);
                atomicMin(&L.best[src], (uint32_t)d);
                const uint32_t k = atomicAdd(&L.hcnt[src], 1u);
                if (k < HMAX) L.hlist[k][src] = ((uint32_t)d << 30) | gk;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // pattern id of a verified entry: locator = entry index, or LOC_INLINE | (index into tab_off) << 2 | j
    __device__ __forceinline__ uint32_t resolve(uint32_t loc) const {
        if (SL && (loc & LOC_INLINE)) loc = tab_off[(loc & (LOC_INLINE - 1u)) >> 2] + (loc & 3u);
        return tab_ent[loc];
    }

    // Tiering: the families are ordered so that after H1 every barcode at distance 0 has been verified, and after
    // H1 + H2 every barcode at distance <= 1 (one half is untouched).  A segment whose minimum is already settled
    // stops generating candidates for the later families — only the minimum tier is reported.
    __device__ __forceinline__ void settle(uint32_t tier) {
        if (active && L.best[lane] <= tier) active = false;
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
        if constexpr (SL) {
            const bool inl = (a & 0x8000u) == 0u;
            const uint32_t c = inl ? (a & 3u) : 0u;
            const uint64_t m0 = __ballot((c & 1u) != 0u), m1 = __ballot((c & 2u) != 0u);
            const uint32_t A = __builtin_amdgcn_alignbit(b, a, 16);  // slot bits 16..47: two rests / first entry index
            const uint32_t B = b >> 16;                              // slot bits 48..63: the third rest / length >> 15
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
                    L.qk[idx] = __builtin_amdgcn_alignbit(B, A, 16);
                    L.qm[idx] = mi + (1u << 24);
                }
                if (c >= 3u) {
                    const uint32_t idx = (pos + 2u) & (QCAP - 1);
                    L.qk[idx] = B;
                    L.qm[idx] = mi + (2u << 24);
                }
                tail += (uint32_t)__popcll(m0) + 2u * (uint32_t)__popcll(m1);
                __builtin_amdgcn_wave_barrier();
                while (tail - head >= 64) {
                    verify(fam, true);
                    head += 64;
                }
            }
            uint64_t om = __ballot(!inl);
            while (om != 0) {
                const int src = __builtin_ctzll(om);
                om &= om - 1;
                const uint32_t first = __builtin_amdgcn_readlane(A, src);
                const uint32_t n = (__builtin_amdgcn_readlane(a, src) & 0x7FFFu) | (__builtin_amdgcn_readlane(B, src) << 15);
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
                    while (tail - head >= 64) {
                        verify(fam, true);
                        head += 64;
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
                if (tail - head >= 64) {
                    verify(fam, true);
                    head += 64;
                }
            }
        }
    }

    // One family.  The lookups are spread over the lanes as (active segment, placement) PAIRS: with A segments of
    // the wave still generating candidates, G = 64 / A placements are looked up per pass (lane = g * A + a works on
    // placement p0 + g for the a-th active segment), so the wave-uniform cost of a pass is shared by 64 lookups
    // even when most segments have settled.  The placements of the family for the longest segment of the wave are
    // tabulated in LDS once; a lane skips those its own segment is too short for.  The bucket range of a pass is
    // queued while the loads of the next pass are in flight.
    template <typename Fam>
    __device__ __forceinline__ void run_family(const Fam &fam, int maxlen) {
        const uint64_t am = __ballot(active);
        if (am == 0) return;
        const int A = (int)__popcll(am);
        if (active) L.alist[lane_rank(am)] = (uint8_t)lane;
        // placement table
        int P = 0;
#pragma unroll
        for (int d = -fam.SLACK; d <= fam.SLACK; d++) {
            const int c = maxlen - fam.GAP - d - fam.LB + 1;
            if (c > 0) {
                for (int t = lane; t < c; t += 64) L.ptab[P + t] = (uint16_t)(t | ((t + fam.GAP + d) << 8));
                P += c;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (P == 0) return;
        const int G = 64 / A;
        const uint32_t inv = 65536u / (uint32_t)A + 1u;           // lane / A for lane < 64 (checked for every A <= 64)
        const int g = (int)(((uint32_t)lane * inv) >> 16), a = lane - g * A;
        const bool worker = g < G;
        uint32_t sl = 0;
        W slo = 0, shi = 0;
        int slen = 0;
        if (worker) {
            sl = L.alist[a];
            slo = L.lo[sl];
            shi = L.hi[sl];
            slen = (int)L.slen[sl];
        }
        const uint32_t *off = tab_off + ft.off_base[fam.INDEX];
        const uint2 *slot = tab_slot + ft.off_base[fam.INDEX];
        uint32_t ps = 0, pe = 0, pmeta = 0;  // the pass whose loads were issued last
        bool pend = false;
        for (int p0 = 0; p0 < P; p0 += G) {
            const int p = p0 + g;
            uint32_t s = 0, e = 0, meta = 0;
            if (worker && p < P) {
                const uint32_t tt = L.ptab[p];
                const int t = (int)(tt & 0xFFu), t2 = (int)(tt >> 8);
                if (sct_valid_of(fam, t2, slen)) {
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
#ifdef SCT_ABLATE_PUSH  // timing-only build: the bucket range is loaded but nothing is queued
                    asm volatile("" ::"v"(s), "v"(e), "v"(meta));
                    s = e = 0;
#endif
                }
            }
            if (pend) push_range(fam, ps, pe, pmeta);
            pend = true;
            ps = s;
            pe = e;
            pmeta = meta;
        }
        if (pend) push_range(fam, ps, pe, pmeta);
        if (tail != head) {  // the family's last, partial batch
            verify(fam, (uint32_t)lane < tail - head);
            head = tail;
        }
    }
};

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// W = uint32_t handles the waves whose longest segment has at most 31 characters, W = uint64_t the others
// (up to 63); both instantiations are launched over all rows and a wave leaves at once if it is not its kind.
// THIRDS = true (max_err <= 1 only) walks the 20/22-bit families of the thirds scheme instead of the quarters.
// KS / HM: ties kept per segment / verified matches kept per segment before the row is deferred.  The main launch
// (KS 8, HM 16) covers all rows; rows it defers are retried by a launch with KS 32, HM 64 over the list `rows`
// (16 listed rows per wave), and only what that launch defers again goes to the exhaustive scan.
// RT = true: barcode length `bl` (8..16) other than 16, piece boundaries computed at run time (sct_family_rt).
template <int E, typename W, bool THIRDS, int KS, int HM, int RPW, bool RT>
__global__ void __launch_bounds__(256)
k_filter(const uint64_t *__restrict__ lo_in, const uint64_t *__restrict__ hi_in, const uint32_t *__restrict__ meta_in,
         int64_t n, const int32_t *__restrict__ rows, const uint32_t *__restrict__ tab_off,
         const uint32_t *__restrict__ tab_ent, const uint32_t *__restrict__ tab_code,
         const uint2 *__restrict__ tab_slot, FamTables ft, uint8_t *__restrict__ min_e, uint32_t *__restrict__ cnt, uint32_t *__restrict__ slots,
         uint64_t *__restrict__ slot_off, int32_t *__restrict__ list_defer, uint32_t *__restrict__ defer_count,
         unsigned long long *__restrict__ slot_cursor, uint64_t slot_tag, int bl_arg) {
    const int bl = RT ? bl_arg : 16;
    __shared__ FilterWaveLds<W, HM> lds[4];
    FilterWaveLds<W, HM> &L = lds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    // RPW rows per wave: 64 in the main launch; the retry launch gives a wave only 16 rows, so that the pair
    // compaction of run_family spreads each row's lookups over four lanes and the few retried rows finish sooner
    const int64_t gid = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * RPW + lane;
    bool active = lane < RPW && gid < n;
    int64_t r = gid;
    int len = 0;
    if (active) {
        // per-row data is read once: non-temporal, so that it does not push the seed tables out of the L2
        if (rows) r = rows[gid];                       // a retried row: listed explicitly
        else active = __builtin_nontemporal_load(&cnt[r]) != CNT_DEFER;  // routed to a scan path by k_pack
    }
    if (active) {
        len = (int)(__builtin_nontemporal_load(&meta_in[r]) & 0xFF);
        active = len >= bl - E;  // shorter rows were finished by k_pack
    }
    if (!active) len = 0;
    const int maxlen = wave_max_i(len);
    if (maxlen == 0) return;  // the whole wave has nothing to do
    if ((maxlen <= 31) != (sizeof(W) == 4)) return;  // the other instantiation's wave
    W lo = 0, hi = 0;
    if (active) {
        lo = (W)__builtin_nontemporal_load(&lo_in[r]);
        hi = (W)__builtin_nontemporal_load(&hi_in[r]);
    }
    const W va = sct_valid_mask<W>(len);
    const W ve = va & ~(W)1, l1 = lo << 1, h1 = hi << 1;
    L.em[lane][0] = ~l1 & ~h1 & ve;
    L.em[lane][1] = l1 & ~h1 & ve;
    L.em[lane][2] = ~l1 & h1 & ve;
    L.em[lane][3] = l1 & h1 & ve;
    L.va[lane] = va;
    L.lo[lane] = lo;
    L.hi[lane] = hi;
    L.slen[lane] = (uint8_t)len;
    L.best[lane] = E + 1;
    L.hcnt[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    FilterWave<E, W, HM, !THIRDS> W_{L, tab_off, tab_ent, tab_code, tab_slot, ft, lane, active, len, lo, hi};
#define RUN_FAMILY(FID)                                                         \
    do {                                                                        \
        if constexpr (RT) W_.run_family(sct_family_rt(FID, bl), maxlen);        \
        else W_.run_family(SctFamily<FID>{}, maxlen);                           \
    } while (0)
    if (THIRDS) {
        RUN_FAMILY(SCT_F_T01);
        if (E >= 1) {
            W_.settle(0);
            RUN_FAMILY(SCT_F_T12);
            RUN_FAMILY(SCT_F_T02);
        }
    } else {
        RUN_FAMILY(SCT_F_H1);
        if (E >= 1) {
            W_.settle(0);
            RUN_FAMILY(SCT_F_H2);
        }
        if (E >= 2) {
            W_.settle(1);
            RUN_FAMILY(SCT_F_MID);
            RUN_FAMILY(SCT_F_02);
            RUN_FAMILY(SCT_F_13);
            RUN_FAMILY(SCT_F_03);
        }
    }
#undef RUN_FAMILY
    const uint32_t nc = active ? L.hcnt[lane] : 0u;
    LaneResult<E, KS> res;
    res.init();
    const uint32_t lim = nc < (uint32_t)HM ? nc : (uint32_t)HM;
    for (uint32_t k = 0; k < lim; k++) {
        const uint32_t v = L.hlist[k][lane];
        res.add((int)(v >> 30), W_.resolve(v & LOC_MASK));
    }
    res.store(active, r, nc > (uint32_t)HM, min_e, cnt, slots, slot_off, list_defer, defer_count, slot_cursor, slot_tag);
}

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
k_scan(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, const int32_t *__restrict__ list,
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
        const int64_t b0 = offs[r];
        const int64_t len = offs[r + 1] - b0;
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
k_scan_pk(const uint8_t *__restrict__ segs, const int64_t *__restrict__ offs, const int32_t *__restrict__ list,
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
        const int64_t b0 = offs[r];
        const int64_t len = offs[r + 1] - b0;
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
// exclusive scan of cnt[n] (u32) into hit_offs[n+1] (i64): per-block sums, one-block scan of the sums, offsets
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

__global__ void __launch_bounds__(256) k_sum_blocks(const uint32_t *__restrict__ cnt, int64_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t lds[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += cnt[base + i];
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

__global__ void __launch_bounds__(256) k_offsets(const uint32_t *__restrict__ cnt, int64_t n, const uint64_t *__restrict__ sums,
                                                  const uint64_t *__restrict__ total, int64_t *__restrict__ hit_offs) {
    __shared__ uint64_t lds[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = base + i < n ? cnt[base + i] : 0u;
        s += v[i];
    }
    uint64_t tot;
    uint64_t run = sums[blockIdx.x] + block_exclusive_scan(s, &tot, lds);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) hit_offs[base + i] = (int64_t)run;
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) hit_offs[n] = (int64_t)*total;
}

// filter path: slots -> CSR
__global__ void __launch_bounds__(256) k_fill(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ slots,
                                               const uint32_t *__restrict__ slots_retry,
                                               const uint64_t *__restrict__ slot_off, const int64_t *__restrict__ hit_offs,
                                               const uint32_t *__restrict__ route_cnt, int64_t n, uint32_t *__restrict__ hits) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (route_cnt[r] == CNT_DEFER) return;  // written by the scan path's fill pass
    const uint32_t c = cnt[r];
    const int64_t o = hit_offs[r];
    if (c == 0) return;
    const uint64_t so = slot_off[r];
    const uint32_t *src = (so & SLOT_TAG_RETRY) ? slots_retry : slots;
    const uint64_t at = so & ~SLOT_TAG_RETRY;
    for (uint32_t i = 0; i < c; i++) hits[o + i] = src[at + i];
}

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

// ------------------------------------------------------------------------------------------------------------
// k_validate: input checks of sct_upload_segments on the device.  bad[0] = smallest row whose offsets are not
// non-decreasing (or n if offs[n] != bytes), bad[1] = smallest byte offset holding a value >= 128; ~0 = none.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_validate(const uint8_t *__restrict__ segs, int64_t bytes, const int64_t *__restrict__ offs, int64_t n,
           unsigned long long *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t a = offs[i], b = offs[i + 1];
        if (b < a || a < 0 || b > bytes) atomicMin(&bad[0], (unsigned long long)i);
    }
    const int64_t p = i * 16;
    if (p < bytes) {  // the buffer is padded by 64 bytes, so the 16-byte load may run past `bytes`
        const uint4 v = *reinterpret_cast<const uint4 *>(segs + p);
        if ((v.x | v.y | v.z | v.w) & 0x80808080u) {
            for (int k = 0; k < 16 && p + k < bytes; k++)
                if (segs[p + k] >= 128) {
                    atomicMin(&bad[1], (unsigned long long)(p + k));
                    break;
                }
        }
    }
}

static inline unsigned blocks_for_n(int64_t n, int per) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(SCT_E_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const { return (T *)p; }
};

struct TimedKernel {
    const char *name;
    hipEvent_t a, b;
};

struct sct_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // barcodes
    int n_bc = 0, bl = 0, np = 0;
    bool have_bc = false, have_tables = false, have_thirds = false;
    FamTables ft{};
    DevBuf pat, tab_off, tab_ent, tab_code, tab_slot;
    // segments
    int64_t n_seg = -1, seg_bytes = 0;
    DevBuf segs, offs;
    // work / results
    DevBuf lo, hi, meta, min_e, cnt, route, slots, slots_retry, list_retry, slot_off, list_scan, list_dp, counters, hit_offs, sums, total, hits, part_best, part_cnt;
    int64_t n_hits = 0;
    bool have_result = false;
    // whitelist dictionary (extract_sr_bc_from_lr)
    DevBuf wl_table, wl_counts, wl_first, wl_skip;
    DevBuf ad_peq, ad_strand, ad_dist, ad_locs, ad_first, ad_last;
    uint32_t wl_mask = 0, wl_len_set = 0;
    int32_t wl_n = -1;
    // options
    bool force_exhaustive = false, timing = true, no_thirds = false;
    int filter_mode = 0;  // 0 = wave-cooperative k_filter, 1 = k_filter_lane
    // timing
    std::vector<TimedKernel> timed;
    std::vector<std::pair<const char *, float>> last_ms;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    int64_t stats[5] = {0, 0, 0, 0, 0};

    hipEvent_t ev() {
        if (pool_used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[pool_used++];
    }
};

struct Timer {
    sct_ctx *c;
    TimedKernel t{};
    bool on;
    Timer(sct_ctx *c_, const char *name) : c(c_), on(c_->timing) {
        if (!on) return;
        t.name = name;
        t.a = c->ev();
        t.b = c->ev();
        if (!t.a || !t.b) { on = false; return; }
        (void)hipEventRecord(t.a, c->stream);
    }
    ~Timer() {
        if (!on) return;
        (void)hipEventRecord(t.b, c->stream);
        c->timed.push_back(t);
    }
};

extern "C" const char *sct_last_error(void) { return g_err.c_str(); }

extern "C" int sct_prefix_len(int threads) {
    int k = 0;
    long long n = 1;
    while (threads > n) {
        n *= 4;
        k++;
    }
    return k;
}

extern "C" int sct_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sct_create(int device, sct_ctx **out) {
    if (!out) return fail(SCT_E_ARG, "sct_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SCT_E_NODEV, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(SCT_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    sct_ctx *c = new sct_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(SCT_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return SCT_OK;
}

extern "C" void sct_destroy(sct_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->pat, &c->tab_off, &c->tab_ent, &c->tab_code, &c->tab_slot, &c->segs, &c->offs, &c->lo, &c->hi, &c->meta, &c->min_e,
                      &c->cnt, &c->route, &c->slots, &c->slots_retry, &c->list_retry, &c->slot_off, &c->list_scan, &c->list_dp, &c->counters, &c->hit_offs, &c->sums,
                      &c->total, &c->hits, &c->part_best, &c->part_cnt, &c->wl_table, &c->wl_counts, &c->wl_first, &c->wl_skip, &c->ad_peq, &c->ad_strand, &c->ad_dist, &c->ad_locs, &c->ad_first, &c->ad_last};
    for (DevBuf *b : bufs) b->release();
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// page-locked host memory: buffers allocated here move over PCIe by DMA at the full link rate (pageable
// buffers are staged through a bounce buffer by the runtime)
extern "C" void *sct_host_alloc(sct_ctx *c, size_t bytes) {
    if (!c) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(SCT_E_HIP, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void sct_host_free(sct_ctx *c, void *p) {
    if (!p) return;
    if (c) (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

extern "C" int sct_sync(sct_ctx *c) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SCT_OK;
}

extern "C" int sct_set_option(sct_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return fail(SCT_E_ARG, "sct_set_option: NULL argument");
    if (!strcmp(key, "force_exhaustive")) c->force_exhaustive = value != 0;
    else if (!strcmp(key, "timing")) c->timing = value != 0;
    else if (!strcmp(key, "filter_mode")) c->filter_mode = (int)value;
    else if (!strcmp(key, "no_thirds")) c->no_thirds = value != 0;
    else return fail(SCT_E_ARG, "unknown option '%s'", key);
    return SCT_OK;
}

// rev_compl of the reference (scTagger.py:148-156) on raw bytes
static inline uint8_t comp_byte(uint8_t b) {
    switch (b) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return b;
    }
}

extern "C" int sct_set_barcodes(sct_ctx *c, const char *bc, int32_t n, int32_t bl) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && !bc)) return fail(SCT_E_ARG, "sct_set_barcodes: bad barcode buffer");
    if (bl < 1) return fail(SCT_E_ARG, "barcode length must be positive (got %d)", bl);
    if (bl > SCT_MAX_BL) return fail(SCT_E_UNSUPPORTED, "barcode length %d exceeds the supported maximum %d", bl, SCT_MAX_BL);
    if ((int64_t)n * 2 > 0x7FFFFFFF) return fail(SCT_E_ARG, "too many barcodes");
    const uint8_t *b = (const uint8_t *)bc;
    for (int64_t i = 0; i < (int64_t)n * bl; i++)
        if (b[i] >= 128) return fail(SCT_E_ARG, "barcode %lld contains byte 0x%02x >= 128", (long long)(i / bl), b[i]);
    HIPCHK(hipSetDevice(c->device));
    const int np = 2 * n;
    std::vector<uint2> pat((size_t)np);
    for (int i = 0; i < n; i++) {
        const uint8_t *s = b + (size_t)i * bl;
        uint32_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
        for (int j = 0; j < bl; j++) {
            const uint32_t cf = sct_code(s[j]);
            const uint32_t cr = sct_code(comp_byte(s[bl - 1 - j]));
            flo |= (cf & 1u) << j;
            fhi |= (cf >> 1) << j;
            rlo |= (cr & 1u) << j;
            rhi |= (cr >> 1) << j;
        }
        pat[(size_t)2 * i + 0] = make_uint2(rlo, rhi);  // strand 0 = reverse complement (False sorts first)
        pat[(size_t)2 * i + 1] = make_uint2(flo, fhi);  // strand 1 = forward
    }
    if (c->pat.ensure(sizeof(uint2) * (size_t)(np ? np : 1))) return SCT_E_HIP;
    if (np) HIPCHK(hipMemcpyAsync(c->pat.p, pat.data(), sizeof(uint2) * (size_t)np, hipMemcpyHostToDevice, c->stream));
    c->have_tables = false;
    c->have_thirds = false;
    if (bl >= 8 && bl <= 16) {
        // seed tables: counting sort of the pattern ids by key, per family (ids ascending inside a bucket).
        // The thirds families (20/22-bit keys) are only worth their 38 MB of offsets for very large barcode sets.
        const bool thirds = bl == 16 && np >= THIRDS_MIN_PATTERNS;
        const int nfam = thirds ? SCT_NFAM_ALL : SCT_NFAM;
        FamTables ft{};
        size_t off_total = 0;
        for (int f = 0; f < nfam; f++) {
            if (off_total + sct_family_nkeys(f) + 1 > 0xFFFFFFFFull || (uint64_t)(f + 1) * (uint64_t)np > 0xFFFFFFFFull)
                return fail(SCT_E_UNSUPPORTED, "too many barcodes for the seed tables");
            ft.off_base[f] = (uint32_t)off_total;
            off_total += (size_t)sct_family_nkeys(f) + 1;
        }
        std::vector<uint32_t> off(off_total, 0u), ent((size_t)nfam * (np ? np : 1)), code((size_t)nfam * (np ? np : 1));
        std::vector<uint32_t> inter((size_t)(np ? np : 1));
        for (int p = 0; p < np; p++) {
            uint32_t v = 0;
            for (int j = 0; j < bl; j++) v |= (((pat[p].x >> j) & 1u) | (((pat[p].y >> j) & 1u) << 1)) << (2 * j);
            inter[p] = v;
        }
        for (int f = 0; f < nfam; f++) {
            const uint32_t nk = sct_family_nkeys(f);
            uint32_t *o = off.data() + ft.off_base[f];
            auto key_of = [&](int p) {  // quarter families: piece boundaries from the barcode length
                if (f >= SCT_NFAM) return sct_pattern_key(f, pat[p].x, pat[p].y);
                const SctFamilyRT fam = sct_family_rt(f, bl);
                return sct_key_of<uint32_t>(fam, pat[p].x, pat[p].y, fam.PA, fam.PB);
            };
            for (int p = 0; p < np; p++) o[key_of(p) + 1]++;
            o[0] = (uint32_t)((size_t)f * np);
            for (uint32_t k = 0; k < nk; k++) o[k + 1] += o[k];
            std::vector<uint32_t> cur(o, o + nk);
            for (int p = 0; p < np; p++) {
                const uint32_t at = cur[key_of(p)]++;
                ent[at] = (uint32_t)p;
                code[at] = inter[p];
            }
        }
        // bucket slots of the quarter families (layout: FilterWave)
        size_t slot_total = 0;
        for (int f = 0; f < SCT_NFAM; f++) slot_total += (size_t)sct_family_nkeys(f) + 1;
        std::vector<uint64_t> slot(slot_total, 0ull);
        for (int f = 0; f < SCT_NFAM; f++) {
            const SctFamilyRT fam = sct_family_rt(f, bl);
            const uint32_t *o = off.data() + ft.off_base[f];
            uint64_t *sl = slot.data() + ft.off_base[f];
            for (uint32_t k = 0; k < sct_family_nkeys(f); k++) {
                const uint32_t cnt = o[k + 1] - o[k];
                uint64_t v;
                if (cnt <= 3) {
                    v = cnt;
                    for (uint32_t j = 0; j < cnt; j++) v |= (uint64_t)(sct_rest_of(fam, code[o[k] + j]) & 0xFFFFu) << (16 * (j + 1));
                } else {
                    v = 0x8000ull | (cnt & 0x7FFFu) | ((uint64_t)o[k] << 16) | ((uint64_t)(cnt >> 15) << 48);
                }
                sl[k] = v;
            }
        }
        if (c->tab_slot.ensure(slot.size() * sizeof(uint64_t))) return SCT_E_HIP;
        HIPCHK(hipMemcpyAsync(c->tab_slot.p, slot.data(), slot.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        if (c->tab_off.ensure(off.size() * sizeof(uint32_t) + 16)) return SCT_E_HIP;
        if (c->tab_ent.ensure(ent.size() * sizeof(uint32_t))) return SCT_E_HIP;
        if (c->tab_code.ensure(code.size() * sizeof(uint32_t))) return SCT_E_HIP;
        HIPCHK(hipMemcpyAsync(c->tab_code.p, code.data(), code.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->tab_off.p, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->tab_ent.p, ent.data(), ent.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->ft = ft;
        c->have_tables = true;
        c->have_thirds = thirds;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_bc = n;
    c->bl = bl;
    c->np = np;
    c->have_bc = true;
    c->have_result = false;
    return SCT_OK;
}

extern "C" int sct_upload_segments(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (n < 0 || !offs) return fail(SCT_E_ARG, "sct_upload_segments: bad offsets");
    if (n >= 0x7FFFFFF0) return fail(SCT_E_ARG, "at most 2^31-16 segments per call (shard the input)");
    if (offs[0] != 0) return fail(SCT_E_ARG, "offs[0] must be 0");
    const int64_t bytes = offs[n];
    if (bytes < 0) return fail(SCT_E_ARG, "offs[n] is negative");
    if (bytes > 0 && !segs) return fail(SCT_E_ARG, "segs is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (c->segs.ensure((size_t)bytes + 64)) return SCT_E_HIP;
    if (c->offs.ensure(sizeof(int64_t) * (size_t)(n + 1))) return SCT_E_HIP;
    if (c->counters.ensure(64)) return SCT_E_HIP;
    hipStream_t st = c->stream;
    if (bytes) HIPCHK(hipMemcpyAsync(c->segs.p, segs, (size_t)bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->offs.p, offs, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, st));
    // validation on the device, before any kernel trusts the offsets: non-decreasing, inside the byte buffer, and
    // no byte >= 128 (the reference raises on those); the host only reads back two words
    HIPCHK(hipMemsetAsync(c->counters.p, 0xFF, 16, st));
    const int64_t items = (n + 1) > (bytes + 15) / 16 ? (n + 1) : (bytes + 15) / 16;
    k_validate<<<blocks_for_n(items, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), bytes, c->offs.as<int64_t>(), n,
                                                        c->counters.as<unsigned long long>());
    unsigned long long bad[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(bad, c->counters.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    c->n_seg = -1;
    if (bad[0] != ~0ull) return fail(SCT_E_ARG, "offs must be non-decreasing and end at the size of segs (row %llu)", bad[0]);
    if (bad[1] != ~0ull) return fail(SCT_E_ARG, "segment bytes contain a value >= 128 at offset %llu", bad[1]);
    c->n_seg = n;
    c->seg_bytes = bytes;
    c->have_result = false;
    return SCT_OK;
}

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int sct_run(sct_ctx *c, int32_t max_err, int32_t threads, int64_t *n_hits_out) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (!c->have_bc) return fail(SCT_E_STATE, "sct_run before sct_set_barcodes");
    if (c->n_seg < 0) return fail(SCT_E_STATE, "sct_run before sct_upload_segments");
    if (max_err < 0 || max_err >= c->bl) return fail(SCT_E_ARG, "need 0 <= max_err < barcode length (got %d, bl %d)", max_err, c->bl);
    if (threads < 1) return fail(SCT_E_ARG, "threads must be >= 1");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->n_seg;
    const int np = c->np, bl = c->bl;
    const int prefix_len = sct_prefix_len(threads);
    const bool filter_ok = c->have_tables && max_err <= 2 && !c->force_exhaustive && np > 0 && (uint64_t)SCT_NFAM_ALL * (uint64_t)np < LOC_INLINE;
    const bool use_thirds = filter_ok && c->have_thirds && max_err <= 1 && c->filter_mode != 1 && !c->no_thirds;
    const bool rt_families = bl != 16;  // quarter families with run-time piece boundaries
    const size_t n1 = (size_t)(n ? n : 1);
    if (c->lo.ensure(8 * n1) || c->hi.ensure(8 * n1) || c->meta.ensure(4 * n1) || c->min_e.ensure(n1) ||
        c->cnt.ensure(4 * n1) || c->route.ensure(4 * n1) || c->list_scan.ensure(4 * n1) || c->list_dp.ensure(4 * n1) ||
        c->counters.ensure(64) || c->hit_offs.ensure(8 * (n1 + 1)) || c->total.ensure(8))
        return SCT_E_HIP;
    if (filter_ok && (c->slots.ensure(4 * n1 * K_SLOTS) || c->slot_off.ensure(8 * n1) || c->list_retry.ensure(4 * n1))) return SCT_E_HIP;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (c->sums.ensure(8 * (size_t)(nb ? nb : 1))) return SCT_E_HIP;
    c->timed.clear();
    c->pool_used = 0;
    c->have_result = false;
    hipStream_t st = c->stream;
    HIPCHK(hipMemsetAsync(c->counters.p, 0, 64, st));
    uint32_t *counters = c->counters.as<uint32_t>();
    if (n > 0) {
        {
            Timer t(c, "k_pack");
            k_pack<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), n, bl, max_err, prefix_len,
                                                       !filter_ok ? 0 : ((c->filter_mode == 1 && bl == 16) ? 31 : PLANE_MAX), c->lo.as<uint64_t>(),
                                                       c->hi.as<uint64_t>(),
                                                       c->meta.as<uint32_t>(), c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>(),
                                                       c->list_scan.as<int32_t>(), c->list_dp.as<int32_t>(), counters);
        }
        if (filter_ok) {
            Timer t(c, "k_filter");
            const unsigned g = blocks_for(n, 256);
#define LANE_ARGS                                                                                                     \
    c->lo.as<uint64_t>(), c->hi.as<uint64_t>(), c->meta.as<uint32_t>(), n, c->tab_off.as<uint32_t>(),                  \
        c->tab_ent.as<uint32_t>(), c->pat.as<uint2>(), np, c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>(),              \
        c->slots.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->list_scan.as<int32_t>(), counters
#define WAVE_ARGS                                                                                                     \
    c->lo.as<uint64_t>(), c->hi.as<uint64_t>(), c->meta.as<uint32_t>(), n, (const int32_t *)nullptr,                   \
        c->tab_off.as<uint32_t>(), c->tab_ent.as<uint32_t>(), c->tab_code.as<uint32_t>(), c->tab_slot.as<uint2>(), c->ft, c->min_e.as<uint8_t>(), \
        c->cnt.as<uint32_t>(), c->slots.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->list_retry.as<int32_t>(),       \
        counters + 2, reinterpret_cast<unsigned long long *>(counters + 4), 0ull, bl
#define LAUNCH_WAVE(E, TH)                                                                                            \
    do {                                                                                                              \
        if (rt_families) {                                                                                            \
            k_filter<E, uint32_t, false, K_SLOTS, HMAX, 64, true><<<g, 256, 0, st>>>(WAVE_ARGS);                       \
            k_filter<E, uint64_t, false, K_SLOTS, HMAX, 64, true><<<g, 256, 0, st>>>(WAVE_ARGS);                       \
        } else {                                                                                                      \
            k_filter<E, uint32_t, TH, K_SLOTS, HMAX, 64, false><<<g, 256, 0, st>>>(WAVE_ARGS);                         \
            k_filter<E, uint64_t, TH, K_SLOTS, HMAX, 64, false><<<g, 256, 0, st>>>(WAVE_ARGS);                         \
        }                                                                                                             \
    } while (0)
#define LAUNCH_FILTER(E)                                                                                              \
    do {                                                                                                              \
        if (c->filter_mode == 1 && !rt_families) k_filter_lane<E><<<g, 256, 0, st>>>(LANE_ARGS);                                       \
        else LAUNCH_WAVE(E, false);                                                                                   \
    } while (0)
            if (use_thirds && max_err == 0) LAUNCH_WAVE(0, true);
            else if (use_thirds) LAUNCH_WAVE(1, true);
            else if (max_err == 0) LAUNCH_FILTER(0);
            else if (max_err == 1) LAUNCH_FILTER(1);
            else LAUNCH_FILTER(2);
#undef LAUNCH_FILTER
#undef LAUNCH_WAVE
#undef WAVE_ARGS
#undef LANE_ARGS
        }
    }
    // host sync #1: how many rows go through the scan paths (normally none or a handful) / are to be retried
    uint32_t hc[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(hc, c->counters.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t n_retried = hc[2];
    if (n > 0 && filter_ok && !(c->filter_mode == 1 && !rt_families) && hc[2] > 0) {
        // rows with more ties / verified matches than the main launch keeps: once more with larger lists
        const int64_t nr = hc[2];
        if (c->slots_retry.ensure(4 * (size_t)nr * K_RETRY)) return SCT_E_HIP;
        Timer t(c, "k_filter_retry");
        const unsigned g = blocks_for(nr, 64);  // 4 waves x 16 rows per block
#define RETRY_ARGS                                                                                                    \
    c->lo.as<uint64_t>(), c->hi.as<uint64_t>(), c->meta.as<uint32_t>(), nr, c->list_retry.as<int32_t>(),               \
        c->tab_off.as<uint32_t>(), c->tab_ent.as<uint32_t>(), c->tab_code.as<uint32_t>(), c->tab_slot.as<uint2>(), c->ft, c->min_e.as<uint8_t>(), \
        c->cnt.as<uint32_t>(), c->slots_retry.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->list_scan.as<int32_t>(),  \
        counters + 0, reinterpret_cast<unsigned long long *>(counters + 6), SLOT_TAG_RETRY, bl
#define LAUNCH_RETRY(E, TH)                                                                                           \
    do {                                                                                                              \
        if (rt_families) {                                                                                            \
            k_filter<E, uint32_t, false, K_RETRY, H_RETRY, 16, true><<<g, 256, 0, st>>>(RETRY_ARGS);                   \
            k_filter<E, uint64_t, false, K_RETRY, H_RETRY, 16, true><<<g, 256, 0, st>>>(RETRY_ARGS);                   \
        } else {                                                                                                      \
            k_filter<E, uint32_t, TH, K_RETRY, H_RETRY, 16, false><<<g, 256, 0, st>>>(RETRY_ARGS);                     \
            k_filter<E, uint64_t, TH, K_RETRY, H_RETRY, 16, false><<<g, 256, 0, st>>>(RETRY_ARGS);                     \
        }                                                                                                             \
    } while (0)
        if (use_thirds && max_err == 0) LAUNCH_RETRY(0, true);
        else if (use_thirds) LAUNCH_RETRY(1, true);
        else if (max_err == 0) LAUNCH_RETRY(0, false);
        else if (max_err == 1) LAUNCH_RETRY(1, false);
        else LAUNCH_RETRY(2, false);
#undef LAUNCH_RETRY
#undef RETRY_ARGS
    }
    if (n > 0) {
        // remember which rows are (still) deferred: cnt is about to be overwritten by the scan path's count pass
        HIPCHK(hipMemcpyAsync(c->route.p, c->cnt.p, 4 * (size_t)n, hipMemcpyDeviceToDevice, st));
        if (hc[2] > 0) {  // the retry may have deferred rows to the scan list
            HIPCHK(hipMemcpyAsync(hc, c->counters.p, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    // a short list is cut into pattern slices so that the whole chip works on it; a long one needs no slicing
    auto slices_for = [&](uint32_t rows) {
        if (rows == 0) return 1;
        const int tiles = (np + 63) / 64;
        int64_t sl = (16384 + rows - 1) / rows;   // aim at 16 k waves in flight
        if (sl < 16) sl = 16;                       // and at least 16 slices: the fill pass skips the slices without a hit
        if (sl > 4096) sl = 4096;
        if (sl > tiles) sl = tiles;
        return (int)(sl < 1 ? 1 : sl);
    };
    const int sl_scan = slices_for(hc[0]), sl_dp = slices_for(hc[1]);
    auto grid_for = [&](uint32_t rows, int sl) {
        const int64_t waves = (int64_t)rows * sl;
        const int64_t blocks = (waves + 3) / 4;
        return (unsigned)(blocks < 2048 ? (blocks < 1 ? 1 : blocks) : 2048);
    };
    if (c->part_best.ensure(4 * ((size_t)hc[0] * sl_scan + (size_t)hc[1] * sl_dp + 1)) ||
        c->part_cnt.ensure(4 * ((size_t)hc[0] * sl_scan + (size_t)hc[1] * sl_dp + 1)))
        return SCT_E_HIP;
    uint32_t *pb_scan = c->part_best.as<uint32_t>(), *pc_scan = c->part_cnt.as<uint32_t>();
    uint32_t *pb_dp = pb_scan + (size_t)hc[0] * sl_scan, *pc_dp = pc_scan + (size_t)hc[0] * sl_scan;
    if (n > 0) {
        if (hc[0]) {
            Timer t(c, "k_scan_count");
            if (bl <= 16)
                k_scan_pk<false><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                    np, bl, max_err, c->min_e.as<uint8_t>(), pb_scan, pc_scan, nullptr, nullptr);
            else
                k_scan<false, false><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                    np, bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_scan, pc_scan, nullptr, nullptr);
            k_scan_reduce<<<blocks_for(hc[0], 256), 256, 0, st>>>(c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, max_err, pb_scan,
                                                                  pc_scan, c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>());
        }
        if (hc[1]) {
            Timer t(c, "k_dp_count");
            k_scan<true, false><<<grid_for(hc[1], sl_dp), 256, 0, st>>>(
                c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_dp.as<int32_t>(), (int64_t)hc[1], sl_dp, c->pat.as<uint2>(), np,
                bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_dp, pc_dp, nullptr, nullptr);
            k_scan_reduce<<<blocks_for(hc[1], 256), 256, 0, st>>>(c->list_dp.as<int32_t>(), (int64_t)hc[1], sl_dp, max_err, pb_dp, pc_dp,
                                                                  c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>());
        }
        {
            Timer t(c, "k_offsets");
            k_sum_blocks<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>());
            k_scan_sums<<<1, 256, 0, st>>>(c->sums.as<uint64_t>(), nb, c->total.as<uint64_t>());
            k_offsets<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>(), c->total.as<uint64_t>(),
                                                    c->hit_offs.as<int64_t>());
        }
    } else {
        HIPCHK(hipMemsetAsync(c->hit_offs.p, 0, 8, st));
        HIPCHK(hipMemsetAsync(c->total.p, 0, 8, st));
    }
    // host sync #2: the total hit count sizes the CSR hit array
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->total.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (c->hits.ensure(4 * (size_t)(total ? total : 1))) return SCT_E_HIP;
    if (n > 0 && total > 0) {
        if (filter_ok) {
            Timer t(c, "k_fill");
            k_fill<<<blocks_for(n, 256), 256, 0, st>>>(c->cnt.as<uint32_t>(), c->slots.as<uint32_t>(), c->slots_retry.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->hit_offs.as<int64_t>(),
                                                       c->route.as<uint32_t>(), n, c->hits.as<uint32_t>());
        }
        if (hc[0]) {
            Timer t(c, "k_scan_fill");
            if (bl <= 16)
                k_scan_pk<true><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                    np, bl, max_err, c->min_e.as<uint8_t>(), pb_scan, pc_scan, c->hit_offs.as<int64_t>(), c->hits.as<uint32_t>());
            else
                k_scan<false, true><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                    np, bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_scan, pc_scan, c->hit_offs.as<int64_t>(), c->hits.as<uint32_t>());
        }
        if (hc[1]) {
            Timer t(c, "k_dp_fill");
            k_scan<true, true><<<grid_for(hc[1], sl_dp), 256, 0, st>>>(
                c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_dp.as<int32_t>(), (int64_t)hc[1], sl_dp, c->pat.as<uint2>(), np,
                bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_dp, pc_dp, c->hit_offs.as<int64_t>(), c->hits.as<uint32_t>());
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    c->last_ms.clear();
    for (const TimedKernel &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) c->last_ms.emplace_back(t.name, ms);
    }
    c->n_hits = (int64_t)total;
    c->stats[0] = n - hc[0] - hc[1];
    c->stats[1] = hc[0];
    c->stats[2] = hc[1];
    c->stats[3] = (int64_t)total;
    c->stats[4] = n_retried;
    c->have_result = true;
    if (n_hits_out) *n_hits_out = (int64_t)total;
    return SCT_OK;
}

extern "C" int sct_download(sct_ctx *c, uint8_t *min_e, int64_t *hit_offs, uint32_t *hits) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (!c->have_result) return fail(SCT_E_STATE, "sct_download without a completed sct_run");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->n_seg;
    if (min_e && n) HIPCHK(hipMemcpyAsync(min_e, c->min_e.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (hit_offs) HIPCHK(hipMemcpyAsync(hit_offs, c->hit_offs.p, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, c->stream));
    if (hits && c->n_hits) HIPCHK(hipMemcpyAsync(hits, c->hits.p, 4 * (size_t)c->n_hits, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SCT_OK;
}

extern "C" int sct_match(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n, int32_t max_err, int32_t threads,
                         uint8_t *min_e, int64_t *hit_offs, uint32_t **hits, int64_t *n_hits) {
    if (!hits || !n_hits || !min_e || !hit_offs) return fail(SCT_E_ARG, "sct_match: NULL output pointer");
    *hits = nullptr;
    *n_hits = 0;
    int rc = sct_upload_segments(c, segs, offs, n);
    if (rc) return rc;
    int64_t total = 0;
    rc = sct_run(c, max_err, threads, &total);
    if (rc) return rc;
    uint32_t *h = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(total ? total : 1));
    if (!h) return fail(SCT_E_ARG, "out of host memory for %lld hits", (long long)total);
    rc = sct_download(c, min_e, hit_offs, h);
    if (rc) {
        free(h);
        return rc;
    }
    *hits = h;
    *n_hits = total;
    return SCT_OK;
}

extern "C" void sct_free_hits(sct_ctx *, uint32_t *hits) { free(hits); }

extern "C" int sct_last_timing(sct_ctx *c, const char **names, float *ms, int cap) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    const int k = (int)c->last_ms.size();
    for (int i = 0; i < k && i < cap; i++) {
        if (names) names[i] = c->last_ms[i].first;
        if (ms) ms[i] = c->last_ms[i].second;
    }
    return k;
}

extern "C" int sct_last_stats(sct_ctx *c, int64_t *out, int cap) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    for (int i = 0; i < 5 && i < cap; i++) out[i] = c->stats[i];
    return 5;
}

// ------------------------------------------------------------------------------------------------------------
// extract_sr_bc_from_lr (reference scTagger.py:792-844): whitelist dictionary + counting
// ------------------------------------------------------------------------------------------------------------
static void wl_pack(const uint8_t *key, int len, bool rc, uint32_t k[4]) {
    uint8_t b[16] = {0};
    for (int j = 0; j < len; j++) b[j] = rc ? comp_byte(key[len - 1 - j]) : key[j];
    for (int w = 0; w < 4; w++) k[w] = (uint32_t)b[4 * w] | ((uint32_t)b[4 * w + 1] << 8) | ((uint32_t)b[4 * w + 2] << 16) | ((uint32_t)b[4 * w + 3] << 24);
}

extern "C" int sct_wl_set(sct_ctx *c, const char *keys, const int32_t *key_lens, int32_t n) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && (!keys || !key_lens))) return fail(SCT_E_ARG, "sct_wl_set: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    int64_t total = 0;
    for (int32_t i = 0; i < n; i++) {
        if (key_lens[i] < 0) return fail(SCT_E_ARG, "negative key length");
        if (key_lens[i] > 16) return fail(SCT_E_UNSUPPORTED, "whitelist entry %d has %d characters; at most 16 are supported", i, key_lens[i]);
        total += key_lens[i];
    }
    const uint8_t *kb = (const uint8_t *)keys;
    for (int64_t i = 0; i < total; i++)
        if (kb[i] >= 128) return fail(SCT_E_ARG, "whitelist contains byte 0x%02x >= 128", kb[i]);
    uint64_t cap = 16;
    while (cap < 4ull * (uint64_t)(n > 0 ? n : 1)) cap <<= 1;  // two words per barcode, load factor <= 0.5
    if (cap > (1ull << 31)) return fail(SCT_E_UNSUPPORTED, "whitelist too large");
    std::vector<SctWlSlot> tab((size_t)cap);
    memset(tab.data(), 0, tab.size() * sizeof(SctWlSlot));
    const uint32_t mask = (uint32_t)(cap - 1);
    uint32_t len_set = 0;
    // forward words in order, then reverse complements; adding an existing key replaces its value (add_word)
    for (int pass = 0; pass < 2; pass++) {
        int64_t at = 0;
        for (int32_t i = 0; i < n; i++) {
            const int L = key_lens[i];
            const uint8_t *key = kb + at;
            at += L;
            if (L == 0) continue;  // the empty word cannot be added to the automaton
            uint32_t k[4];
            wl_pack(key, L, pass == 1, k);
            uint32_t idx = sct_wl_hash(k[0], k[1], k[2], k[3], (uint32_t)L) & mask;
            for (;;) {
                SctWlSlot &s = tab[idx];
                if (s.len == 0) {
                    s.len = (uint32_t)L;
                    memcpy(s.k, k, sizeof(k));
                    s.value = pass == 0 ? i : -i;
                    break;
                }
                if (s.len == (uint32_t)L && !memcmp(s.k, k, sizeof(k))) {
                    s.value = pass == 0 ? i : -i;
                    break;
                }
                idx = (idx + 1) & mask;
            }
            len_set |= 1u << L;
        }
    }
    if (c->wl_table.ensure(tab.size() * sizeof(SctWlSlot))) return SCT_E_HIP;
    HIPCHK(hipMemcpyAsync(c->wl_table.p, tab.data(), tab.size() * sizeof(SctWlSlot), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->wl_mask = mask;
    c->wl_len_set = len_set;
    c->wl_n = n;
    return SCT_OK;
}

extern "C" int sct_wl_count(sct_ctx *c, const char *segs, const int64_t *offs, const uint8_t *skip, int64_t n, int64_t *counts,
                            int64_t *first_seen) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (c->wl_n < 0) return fail(SCT_E_STATE, "sct_wl_count before sct_wl_set");
    if (!counts || !first_seen || (n > 0 && !skip)) return fail(SCT_E_ARG, "sct_wl_count: NULL argument");
    int rc = sct_upload_segments(c, segs, offs, n);
    if (rc) return rc;
    const size_t nk = (size_t)(c->wl_n > 0 ? c->wl_n : 1);
    if (c->wl_counts.ensure(4 * nk) || c->wl_first.ensure(4 * nk) || c->wl_skip.ensure((size_t)(n ? n : 1))) return SCT_E_HIP;
    hipStream_t st = c->stream;
    c->timed.clear();
    c->pool_used = 0;
    HIPCHK(hipMemsetAsync(c->wl_counts.p, 0, 4 * nk, st));
    HIPCHK(hipMemsetAsync(c->wl_first.p, 0xFF, 4 * nk, st));
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(c->wl_skip.p, skip, (size_t)n, hipMemcpyHostToDevice, st));
        Timer t(c, "k_wl_count");
        k_wl_count<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->wl_skip.as<uint8_t>(), n,
                                                      c->wl_table.as<SctWlSlot>(), c->wl_mask, c->wl_len_set,
                                                      c->wl_counts.as<uint32_t>(), c->wl_first.as<uint32_t>());
    }
    std::vector<uint32_t> hc(nk), hf(nk);
    HIPCHK(hipMemcpyAsync(hc.data(), c->wl_counts.p, 4 * nk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hf.data(), c->wl_first.p, 4 * nk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    c->last_ms.clear();
    for (const TimedKernel &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) c->last_ms.emplace_back(t.name, ms);
    }
    for (int32_t i = 0; i < c->wl_n; i++) {
        counts[i] = hc[(size_t)i];
        first_seen[i] = hf[(size_t)i] == 0xFFFFFFFFu ? -1 : (int64_t)hf[(size_t)i];
    }
    return SCT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// extract_lr_bc (reference scTagger.py:176-196): adapter alignment over whole reads
// ------------------------------------------------------------------------------------------------------------
extern "C" int sct_adapter_align(sct_ctx *c, const char *adapter, int32_t m, const char *reads, const int64_t *offs, int64_t n,
                                 int8_t *strand, int32_t *dist, int64_t *loc_offs, int32_t **locs, int64_t *n_locs) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (!adapter || !strand || !dist || !loc_offs || !locs || !n_locs) return fail(SCT_E_ARG, "sct_adapter_align: NULL argument");
    *locs = nullptr;
    *n_locs = 0;
    if (m < 1) return fail(SCT_E_ARG, "adapter length must be positive");
    if (m > 32) return fail(SCT_E_UNSUPPORTED, "adapter length %d exceeds the supported maximum 32", m);
    for (int i = 0; i < m; i++)
        if ((uint8_t)adapter[i] >= 128) return fail(SCT_E_ARG, "adapter contains a byte >= 128");
    int rc = sct_upload_segments(c, reads, offs, n);
    if (rc) return rc;
    // match masks: adapter, its reverse complement (rev_compl, scTagger.py:148-156, :365), and that one reversed
    std::vector<uint32_t> peq(3 * 256, 0u);
    std::vector<uint8_t> a2((size_t)m);
    for (int i = 0; i < m; i++) a2[i] = comp_byte((uint8_t)adapter[m - 1 - i]);
    for (int i = 0; i < m; i++) {
        peq[(uint8_t)adapter[i]] |= 1u << i;
        peq[256 + a2[i]] |= 1u << i;
        peq[512 + a2[m - 1 - i]] |= 1u << i;
    }
    const size_t n1 = (size_t)(n ? n : 1);
    if (c->ad_peq.ensure(peq.size() * 4) || c->ad_strand.ensure(n1) || c->ad_dist.ensure(4 * n1) || c->cnt.ensure(4 * n1) ||
        c->ad_first.ensure(8 * n1) || c->ad_last.ensure(8 * n1) ||
        c->hit_offs.ensure(8 * (n1 + 1)) || c->total.ensure(8))
        return SCT_E_HIP;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (c->sums.ensure(8 * (size_t)(nb ? nb : 1))) return SCT_E_HIP;
    hipStream_t st = c->stream;
    c->timed.clear();
    c->pool_used = 0;
    c->have_result = false;
    HIPCHK(hipMemcpyAsync(c->ad_peq.p, peq.data(), peq.size() * 4, hipMemcpyHostToDevice, st));
    uint64_t total = 0;
    if (n > 0) {
        {
            Timer t(c, "k_adapter_count");
            k_adapter_count<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), n, c->ad_peq.as<uint32_t>(), m,
                                                               c->ad_strand.as<int8_t>(), c->ad_dist.as<int32_t>(), c->cnt.as<uint32_t>(),
                                                               c->ad_first.as<int64_t>(), c->ad_last.as<int64_t>());
        }
        {
            Timer t(c, "k_offsets");
            k_sum_blocks<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>());
            k_scan_sums<<<1, 256, 0, st>>>(c->sums.as<uint64_t>(), nb, c->total.as<uint64_t>());
            k_offsets<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>(), c->total.as<uint64_t>(),
                                                    c->hit_offs.as<int64_t>());
        }
        HIPCHK(hipMemcpyAsync(&total, c->total.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (c->ad_locs.ensure(4 * (size_t)(total ? total : 1))) return SCT_E_HIP;
        if (total) {
            Timer t(c, "k_adapter_fill");
            k_adapter_fill<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), n, c->ad_peq.as<uint32_t>(), m,
                                                              c->ad_strand.as<int8_t>(), c->ad_dist.as<int32_t>(), c->ad_first.as<int64_t>(),
                                                              c->ad_last.as<int64_t>(), c->hit_offs.as<int64_t>(), c->ad_locs.as<int32_t>());
        }
    } else {
        loc_offs[0] = 0;
    }
    int32_t *h = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
    if (!h) return fail(SCT_E_ARG, "out of host memory");
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(strand, c->ad_strand.p, (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(dist, c->ad_dist.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(loc_offs, c->hit_offs.p, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
        if (total) HIPCHK(hipMemcpyAsync(h, c->ad_locs.p, 4 * (size_t)total, hipMemcpyDeviceToHost, st));
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        free(h);
        return fail(SCT_E_HIP, "adapter alignment failed: %s", hipGetErrorString(e));
    }
    c->last_ms.clear();
    for (const TimedKernel &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) c->last_ms.emplace_back(t.name, ms);
    }
    *locs = h;
    *n_locs = (int64_t)total;
    return SCT_OK;
}

extern "C" void sct_free_locs(sct_ctx *, int32_t *locs) { free(locs); }
