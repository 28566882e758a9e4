// sct_kernels_common.h — shared constants of the match_trie kernels.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, pack, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
static constexpr int K_SLOTS = 8;          // hits kept per segment by the filter kernel before it defers
static constexpr uint32_t CNT_DEFER = 0xFFFFFFFFu;
// The bucket slots pay while nearly every lookup is answered inline; with many long buckets the plain offset / code
// tables are faster.  Measured per 2 M segments, slots off / on (tools/sweep_slots.sh): e <= 2: 25 k barcodes (0.8 % of the
// keys long) 0.98 / 0.73 ms, 35 k (2.4 %) 1.09 / 0.97, 50 k (7 %) 1.27 / 1.55, 100 k (35 %) 2.01 / 5.09;
// e <= 1: 50 k 0.54 / 0.51, 70 k (17 %) 0.58 / 0.76.
static constexpr double SLOTS_MAX_LONG_SHARE_E2 = 0.035, SLOTS_MAX_LONG_SHARE_E1 = 0.10;
static constexpr int THIRDS_MIN_PATTERNS = 1 << 19;  // from here on (8 barcodes per 16-bit bucket) the thirds scheme pays
static constexpr int META_NONACGT = 1 << 8;
static constexpr int META_LONG = 1 << 9;   // more than 63 characters: not representable in the filter's planes
static constexpr int PLANE_MAX = 63;       // characters kept in the packed bit-planes

