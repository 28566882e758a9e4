// k_offsets.hip.h — hit-count scan into CSR offsets and k_fill.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, pack, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
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

