// pcie_probe.hip — what the host link delivers to chunked, multi-stream copies (the shape sct_match_stream produces).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/pcie_probe.hip -o /tmp/pcie_probe && /tmp/pcie_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_spin(unsigned *p, int iters) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 12345u) p[0] = v;
}

// zero-copy: the kernel reads the page-locked host buffer itself
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k_read_host(const u4 *__restrict__ src, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u4 v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main() {
    const size_t total = 230u << 20;
    char *h, *h2;
    CHK(hipHostMalloc((void **)&h, total, hipHostMallocDefault));
    CHK(hipHostMalloc((void **)&h2, total, hipHostMallocDefault));
    memset(h, 65, total);
    memset(h2, 0, total);
    char *d, *d2;
    CHK(hipMalloc(&d, total));
    CHK(hipMalloc(&d2, total));
    unsigned *sink;
    CHK(hipMalloc(&sink, 64));
    hipStream_t st[4];
    for (int i = 0; i < 4; i++) CHK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    auto sync_all = [&]() { for (int i = 0; i < 4; i++) CHK(hipStreamSynchronize(st[i])); };
    printf("{\n");
    for (int rep = 0; rep < 2; rep++) {
        // 1. one copy
        sync_all();
        double t0 = now_ms();
        CHK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st[0]));
        sync_all();
        double t1 = now_ms();
        if (rep) printf("  \"h2d_single_230MB_GBps\": %.1f,\n", total / (t1 - t0) / 1e6);
        t0 = now_ms();
        CHK(hipMemcpyAsync(h2, d, total, hipMemcpyDeviceToHost, st[0]));
        sync_all();
        t1 = now_ms();
        if (rep) printf("  \"d2h_single_230MB_GBps\": %.1f,\n", total / (t1 - t0) / 1e6);
        // 2. chunked, one stream / three streams
        for (int ns = 1; ns <= 3; ns += 2) {
            for (size_t chunk : {(size_t)(11u << 20), (size_t)(23u << 20), (size_t)(46u << 20)}) {
                sync_all();
                t0 = now_ms();
                int k = 0;
                for (size_t o = 0; o < total; o += chunk, k++) {
                    const size_t m = total - o < chunk ? total - o : chunk;
                    CHK(hipMemcpyAsync(d + o, h + o, m, hipMemcpyHostToDevice, st[k % ns]));
                }
                sync_all();
                t1 = now_ms();
                if (rep) printf("  \"h2d_chunk%zuMB_streams%d_GBps\": %.1f,\n", chunk >> 20, ns, total / (t1 - t0) / 1e6);
            }
        }
        // 3. H2D on stream 0 with D2H on stream 1 at the same time
        sync_all();
        t0 = now_ms();
        CHK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st[0]));
        CHK(hipMemcpyAsync(h2, d2, total / 4, hipMemcpyDeviceToHost, st[1]));
        CHK(hipStreamSynchronize(st[0]));
        t1 = now_ms();
        sync_all();
        if (rep) printf("  \"h2d_230MB_beside_d2h_GBps\": %.1f,\n", total / (t1 - t0) / 1e6);
        // 4. H2D beside a compute kernel on another stream
        sync_all();
        hipLaunchKernelGGL(k_spin, dim3(256 * 8), dim3(256), 0, st[2], sink, 400000);
        t0 = now_ms();
        CHK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st[0]));
        CHK(hipStreamSynchronize(st[0]));
        t1 = now_ms();
        sync_all();
        double t2 = now_ms();
        if (rep) printf("  \"h2d_230MB_beside_kernel_GBps\": %.1f, \"kernel_beside_ms\": %.2f,\n", total / (t1 - t0) / 1e6, t2 - t0);
        // 5. chunked on 3 streams, each copy followed by a kernel on its stream (the pipeline's shape)
        sync_all();
        t0 = now_ms();
        int k = 0;
        const size_t chunk = 23u << 20;
        for (size_t o = 0; o < total; o += chunk, k++) {
            const size_t m = total - o < chunk ? total - o : chunk;
            CHK(hipMemcpyAsync(d + o, h + o, m, hipMemcpyHostToDevice, st[k % 3]));
            hipLaunchKernelGGL(k_spin, dim3(256 * 8), dim3(256), 0, st[k % 3], sink, 30000);
            CHK(hipMemcpyAsync(h2 + o / 4, d2 + o / 4, m / 4, hipMemcpyDeviceToHost, st[k % 3]));
        }
        sync_all();
        t1 = now_ms();
        if (rep) printf("  \"pipeline_shape_3streams_ms\": %.2f, \"pipeline_shape_GBps\": %.1f,\n", t1 - t0, total / (t1 - t0) / 1e6);
        // 6. zero-copy read of the host buffer by a kernel
        for (int blocks : {256, 1024, 4096}) {
            sync_all();
            t0 = now_ms();
            hipLaunchKernelGGL(k_read_host, dim3(blocks), dim3(256), 0, st[0], (const u4 *)h, total / 16, sink);
            sync_all();
            t1 = now_ms();
            if (rep) printf("  \"zero_copy_read_blocks%d_GBps\": %.1f,\n", blocks, total / (t1 - t0) / 1e6);
        }
    }
    printf("  \"bytes\": %zu\n}\n", total);
    return 0;
}
