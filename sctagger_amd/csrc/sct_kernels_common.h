// sct_kernels_common.h — shared constants of the match_trie kernels.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
static constexpr int K_SLOTS = 8;          // hits kept per segment by the filter kernel before it defers
static constexpr uint32_t CNT_DEFER = 0xFFFFFFFFu;
static constexpr uint32_t CNT_RETRIED = 0x80000000u;  // flag in cnt[r]: the row's list lies in the retry launch's slots (at slot_off[r])
static constexpr uint32_t CNT_MASK = 0x7FFFFFFFu;
// The bucket slots pay while nearly every lookup is answered inline; with many long buckets the plain offset / code
// tables are faster.  Measured per 2 M segments, slots off / on (tools/sweep_slots.sh): e <= 2: 25 k barcodes (0.8 % of the
// keys long) 0.98 / 0.73 ms, 35 k (2.4 %) 1.09 / 0.97, 50 k (7 %) 1.27 / 1.55, 100 k (35 %) 2.01 / 5.09;
// e <= 1: 50 k 0.54 / 0.51, 70 k (17 %) 0.58 / 0.76.
static constexpr double SLOTS_MAX_LONG_SHARE_E2 = 0.035, SLOTS_MAX_LONG_SHARE_E1 = 0.10;
// The slot kernels put every entry of a long bucket that shares a verified candidate's rest characters on the hit list by reading
// the bucket again (FilterWave::record): fine for the handful of entries a long bucket usually has, quadratic for a set in which
// one sequence is listed under thousands of ids.  Such sets use the plain tables (entries queued by index, no second pass).
static constexpr uint32_t SLOTS_MAX_BUCKET = 256;
static constexpr int THIRDS_MIN_PATTERNS = 1 << 19;  // from here on (8 barcodes per 16-bit bucket) the thirds scheme pays
// max_err = 2: from here on (524 288 barcodes) the sixteen 20-bit families are built and used.  Measured per 1 M segments, quarter
// scheme / wide scheme (tools/wide_sweep.sh, 10^6 segments/s): 100 k barcodes 778 / 283, 400 k 188 / 144, 500 k 141 / 134,
// 700 k 99 / 128, 1 M 80 / 126, 2 M 58 / 122, 3.6 M 45 / 112.
static constexpr int WIDE_MIN_PATTERNS = 1 << 20;
static constexpr int PLANE_MAX = 63;       // characters kept in the packed bit-planes

// Per-batch device counters (one 128-byte record per lane, read back by the host in ONE copy when the batch is done).
// 32-bit words:
enum {
    CTR_SCAN = 0,       // rows listed for the exhaustive scan (list_scan): rows longer than PLANE_MAX, rows the retry launch gave up on
    CTR_DP = 1,         // rows listed for the restricted-start DP (list_dp): what the special launch could not hold of the rows with
                        // non-ACGT bytes under a prefix length k > 0; all such rows when there are no seed tables
    CTR_RETRY = 2,      // rows the main filter launch hands to the retry launch (list_retry)
    CTR_LONG = 3,       // rows the special launch hands on (more windows than it has lanes, lists that overflow), not yet routed (list_long)
    CTR_CURSOR = 4,     // (64-bit, words 4-5)  entries used in the main launch's dense hit slots
    CTR_CURSOR_RETRY = 6,  // (64-bit, words 6-7)  the same for the retry launch
    CTR_BAD_ROW = 8,    // (64-bit, words 8-9)  smallest row whose index entry is invalid, ~0 = none
    CTR_BAD_BYTE = 10,  // (64-bit, words 10-11) smallest offset of a segment byte >= 128, ~0 = none
    CTR_CNT_OVERFLOW = 12,  // a row's hit count did not fit the requested count format
    CTR_JOIN_OVERFLOW = 13, // partitioned join (k_join.hip.h): a producer block had more tuples for one partition than its region holds
    CTR_TOTAL = 14,     // (64-bit, words 14-15) total number of hits of the batch
    CTR_SPECIAL = 16,   // rows the main launch lists for the special launch (list_special): restricted starts, more than PLANE_MAX characters
    CTR_SPECIAL2 = 17,  // rows the special launch's first stage (16 rows per wave, bucket slots) hands to its second (list_special2)
    CTR_GATHER = 18,    // (64-bit, words 18-19) bytes used in the gather buffer (k_special_gather: long listed rows copied to the device)
    CTR_NEED64 = 20,    // set by the main launch's 32-bit instantiation when a wave belongs to the 64-bit one (rows of 32 .. 63 characters)
    CTR_WORDS = 32
};

// Start and length of row r from its CSR offsets, forced into [0, bytes] so that no kernel reads outside the segment
// buffer even if the caller's index is inconsistent (the inconsistency itself is reported through CTR_BAD_ROW).
template <typename OffT>
__device__ __forceinline__ void sct_row_span(const OffT *__restrict__ offs, int64_t r, int64_t bytes, int64_t &start, int64_t &len) {
    int64_t a = (int64_t)offs[r], b = (int64_t)offs[r + 1];
    a = a < 0 ? 0 : (a > bytes ? bytes : a);
    b = b < a ? a : (b > bytes ? bytes : b);
    start = a;
    len = b - a;
}
