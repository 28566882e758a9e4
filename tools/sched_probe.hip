// sched_probe.hip — how long does a small kernel wait behind another stream's large grid, by stream priority?
// (The stream path's follow-up kernels of chunk k were seen to wait for the whole dispatch of chunk k+1's filter kernel.)
// hipcc -O3 --offload-arch=gfx950 tools/sched_probe.hip -o /tmp/sched_probe && /tmp/sched_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CHK(x)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ~25 KB of LDS per block like the filter kernel (6 blocks per CU), `iters` of dependent integer work per thread
__global__ void __launch_bounds__(256) k_big(unsigned *p, int iters) {
    __shared__ unsigned lds[6272];
    unsigned v = threadIdx.x + blockIdx.x;
    lds[threadIdx.x] = v;
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u + lds[(v >> 8) & 255];
    if (v == 12345u) p[0] = v;
}
__global__ void k_small(unsigned *p) {
    if (p[1] == 77u) p[2] = 1;
}

int main() {
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("{\"priority_range_least_greatest\": [%d, %d],\n", lo, hi);
    unsigned *d;
    CHK(hipMalloc(&d, 64));
    CHK(hipMemset(d, 0, 64));
    hipStream_t big, small_n, small_h;
    CHK(hipStreamCreateWithFlags(&big, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&small_n, hipStreamNonBlocking));
    CHK(hipStreamCreateWithPriority(&small_h, hipStreamNonBlocking, hi));
    const int blocks = 1536 * 6;  // six rounds of the whole chip
    for (int rep = 0; rep < 3; rep++) {
        for (int which = 0; which < 2; which++) {
            hipStream_t ss = which ? small_h : small_n;
            CHK(hipDeviceSynchronize());
            const double t0 = now_ms();
            hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), 0, big, d, 20000);
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            const double t1 = now_ms();
            for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, ss, d);
            CHK(hipStreamSynchronize(ss));
            const double t2 = now_ms();
            CHK(hipStreamSynchronize(big));
            const double t3 = now_ms();
            if (rep) printf("  \"rep%d_%s\": {\"eight_small_kernels_done_after_ms\": %.3f, \"big_kernel_ms\": %.3f},\n", rep,
                            which ? "high_priority" : "same_priority", t2 - t1, t3 - t0);
        }
    }
    printf("  \"blocks\": %d}\n", blocks);
    return 0;
}
