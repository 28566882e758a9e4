// gather_probe.hip — how fast can gfx950 serve RANDOM small reads from a table of T bytes?  (MI355X)
//
// Question: cfg5's seed lookups are one random 16/32-byte block per lookup out of 160 MB of fixed buckets and
// run at 5.2e10 lookups/s (profiles/r02f_cfg5_counters.json: 3.3 TB/s of 64-byte FETCH_SIZE lines).  Is that the ceiling of the
// memory system for this access shape — and how does it move with the table size (L2 4 MB per XCD, Infinity Cache 256 MB, HBM)
// and with the width of a read (16 / 32 / 64 / 128 bytes per lookup)?
//
// Method: every lane walks a private LCG over the table's blocks, K independent reads in flight per lane per round, XOR of
// what it reads goes to a sink.  Reported: lookups/s and the bytes/s that are USED.  Tables are filled once; a warm-up launch
// precedes the timed one.  `sorted` = 1 makes the 64 lanes of a wave read 64 consecutive blocks (the coalesced reference).
//
// Build and run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

template <int WORDS>  // WORDS x 16 bytes per lookup
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ tab, uint32_t block_mask, int rounds, int sorted, uint32_t *__restrict__ sink) {
    constexpr int K = 4;  // independent lookups in flight per lane
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[K][WORDS];
#pragma unroll
        for (int k = 0; k < K; k++) {
            x = x * 1664525u + 1013904223u;
            uint32_t b = (x >> 4) & block_mask;
            if (sorted) b = ((b & ~63u) | (threadIdx.x & 63u)) & block_mask;
#pragma unroll
            for (int w = 0; w < WORDS; w++) v[k][w] = tab[(size_t)b * WORDS + w];
        }
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int w = 0; w < WORDS; w++) acc ^= v[k][w].x ^ v[k][w].y ^ v[k][w].z ^ v[k][w].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int WORDS>
static void run(const uint4 *tab, size_t table_bytes, int sorted, uint32_t *sink) {
    const size_t blocks = table_bytes / (16 * WORDS);
    uint32_t mask = 1;
    while ((size_t)mask * 2 <= blocks) mask *= 2;
    mask -= 1;
    const int grid = 256 * 16, rounds = 256;   // 16 blocks of 4 waves per CU queued: the chip stays full
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    k_gather<WORDS><<<grid, 256>>>(tab, mask, 16, sorted, sink);
    CHK(hipEventRecord(a));
    k_gather<WORDS><<<grid, 256>>>(tab, mask, rounds, sorted, sink);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    const double lookups = (double)grid * 256 * rounds * 4;
    printf("{\"table_MB\": %.0f, \"bytes_per_lookup\": %d, \"sorted\": %d, \"ms\": %.3f, \"lookups_per_s\": %.3e, \"used_GBps\": %.1f, "
           "\"GBps_if_64B_lines\": %.1f, \"GBps_if_128B_lines\": %.1f}\n",
           ((double)(mask + 1) * 16 * WORDS) / 1e6, 16 * WORDS, sorted, ms, lookups / (ms * 1e-3), lookups * 16 * WORDS / (ms * 1e-3) / 1e9,
           lookups * (16 * WORDS > 64 ? 16 * WORDS : 64) / (ms * 1e-3) / 1e9, lookups * 128 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const size_t max_bytes = (size_t)2 << 30;
    uint4 *tab = nullptr;
    uint32_t *sink = nullptr;
    CHK(hipMalloc(&tab, max_bytes));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(tab, 0x5A, max_bytes));
    CHK(hipDeviceSynchronize());
    const size_t sizes[] = {(size_t)2 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)2 << 30};
    for (size_t s : sizes) {
        run<1>(tab, s, 0, sink);
        run<2>(tab, s, 0, sink);
        run<4>(tab, s, 0, sink);
        run<8>(tab, s, 0, sink);
    }
    run<1>(tab, (size_t)2 << 30, 1, sink);
    run<2>(tab, (size_t)2 << 30, 1, sink);
    return 0;
}
