// link_shape_probe.hip — why does k_filter read page-locked host memory at 50 GB/s when a plain streaming kernel reads it at 57?
//
// The filter wave's access shape, without the filter: wave w loads the stretch [w * S, (w + 1) * S) of a page-locked host buffer
// with 16-byte non-temporal loads (one per lane and step, the start rounded down to 16 bytes as sct_decode_rows_staged does),
// then "computes" for a given number of cycles (s_sleep loop: the wave keeps its slot, issues nothing), then optionally writes
// 2 result bytes per row back to host memory as 16-byte stores (min_e and the one-byte counts).  One block = 4 waves; blocks per
// CU limited through dynamic LDS like the real kernel (24.5 KB per block -> 6 blocks per CU).
//
// Build and run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/link_shape_probe.hip -o /tmp/link_shape_probe && /tmp/link_shape_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHK(x)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_shape(const uint8_t *__restrict__ src, size_t bytes, int stretch, int spin_cycles, uint8_t *__restrict__ out,
                                               uint32_t *__restrict__ sink) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t a = wave * (size_t)stretch;
    if (a >= bytes) return;
    size_t b = a + (size_t)stretch;
    if (b > bytes) b = bytes;
    const size_t base = a & ~(size_t)15;
    const int n16 = (int)((b - base + 15) >> 4);
    uint32_t acc = 0;
    uint32_t *stage = lds + (threadIdx.x >> 6) * 1024;
    for (int i = lane; i < n16; i += 64) {
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src + base) + i);
        if (i < 256) *reinterpret_cast<u4 *>(stage + 4 * i) = v;
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    __builtin_amdgcn_wave_barrier();
    acc ^= stage[lane];
    // "compute": keep the slot for spin_cycles
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)spin_cycles / 24) __builtin_amdgcn_s_sleep(8);   // s_memtime counts at 100 MHz, spin_cycles are 2.4 GHz cycles
    if (out) {  // 64 rows per wave: 64 bytes of min_e and 64 of counts, as 16-byte stores
        if ((lane & 15) == 0) {
            __builtin_nontemporal_store(u4{acc, acc, acc, acc}, reinterpret_cast<u4 *>(out + wave * 64 + lane));
            __builtin_nontemporal_store(u4{acc, acc, acc, acc}, reinterpret_cast<u4 *>(out + (bytes / 16) + wave * 64 + lane));
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main() {
    const size_t bytes = 209u << 20;
    uint8_t *h, *hout;
    CHK(hipHostMalloc((void **)&h, bytes + 4096, hipHostMallocDefault));
    CHK(hipHostMalloc((void **)&hout, bytes / 8 + 4096, hipHostMallocDefault));
    memset(h, 65, bytes + 4096);
    uint32_t *sink;
    CHK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_shape), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
    const int stretches[] = {1408, 1342, 2816, 5632, 16384};
    const int spins[] = {0, 100000, 200000};    // cycles at 2.4 GHz: 0, ~42 us, ~83 us
    printf("[\n");
    for (int with_out = 0; with_out < 2; with_out++)
        for (int lds_kb : {16, 25})
            for (int s : stretches)
                for (int spin : spins) {
                    const size_t waves = (bytes + s - 1) / s;
                    const unsigned blocks = (unsigned)((waves + 3) / 4);
                    float best = 1e9f;
                    for (int rep = 0; rep < 3; rep++) {
                        CHK(hipEventRecord(e0));
                        hipLaunchKernelGGL(k_shape, dim3(blocks), dim3(256), (size_t)lds_kb << 10, 0, h, bytes, s, spin, with_out ? hout : nullptr, sink);
                        CHK(hipEventRecord(e1));
                        CHK(hipEventSynchronize(e1));
                        float ms;
                        CHK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    printf(" {\"stretch\": %d, \"spin_cycles\": %d, \"lds_kb_per_block\": %d, \"writes_back\": %d, \"ms\": %.3f, \"GBps\": %.1f},\n", s, spin, lds_kb,
                           with_out, best, bytes / (best * 1e-3) / 1e9);
                    fflush(stdout);
                }
    printf(" {}\n]\n");
    return 0;
}
