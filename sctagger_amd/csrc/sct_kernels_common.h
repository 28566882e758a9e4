// sct_kernels_common.h — shared constants of the match_trie kernels.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, pack, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
static constexpr int K_SLOTS = 8;          // hits kept per segment by the filter kernel before it defers
static constexpr uint32_t CNT_DEFER = 0xFFFFFFFFu;
static constexpr int THIRDS_MIN_PATTERNS = 1 << 19;  // from here on (8 barcodes per 16-bit bucket) the thirds scheme pays
static constexpr int META_NONACGT = 1 << 8;
static constexpr int META_LONG = 1 << 9;   // more than 63 characters: not representable in the filter's planes
static constexpr int PLANE_MAX = 63;       // characters kept in the packed bit-planes

