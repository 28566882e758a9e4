// wait_value_probe.hip — can a second stream start work when HALF of a running kernel's blocks are done?  (MI355X)
//
// Figure (ii)'s tail (retry launch, offsets, hit lists: 0.5 ms per 10 M rows) runs with the up-link idle; the rows of the first
// half of a batch are finished long before the main launch ends.  Splitting the launch costs its ramp twice (measured), so the
// question is whether a stream can wait on a COUNTER the running kernel's blocks increment: hipStreamWaitValue32 on signal memory.
// Kernel A: `blocks` blocks, each spins `spin` clocks, then (thread 0) counts itself if its index is below `half` — the block that
// completes the half writes the signal word —; the last block stores the time.  Stream T: wait until counter >= half, then kernel B stores the time.  B's time must lie before A's end.
//
//   hipcc -O3 --offload-arch=gfx950 tools/wait_value_probe.hip -o /tmp/wvp && timeout -k 5 60 /tmp/wvp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                         \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void k_a(unsigned *counter, unsigned half, long long spin, unsigned long long *t_first, unsigned long long *t_end, unsigned *done_all) {
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_first = t0;
    while ((long long)(wall_clock64() - t0) < spin) {
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x < half) {
            // (20 000 system-scope atomics on the signal word itself take 20 ms: the blocks count in device memory and the one that
            // completes the half writes the signal word once)
            __threadfence();
            if (atomicAdd(done_all + 1, 1u) == half - 1u) {
                __threadfence_system();
                __hip_atomic_store(counter, half, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        const unsigned d = atomicAdd(done_all, 1u);
        if (d == gridDim.x - 1) *t_end = wall_clock64();
    }
}

__global__ void k_b(unsigned long long *t_b, const unsigned *counter, unsigned *seen) {
    *t_b = wall_clock64();
    *seen = *counter;
}

int main() {
    int can = 0;
    CHK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    if (!can) {
        printf("{\"can_use_stream_wait_value\": 0}\n");
        return 0;
    }
    unsigned *counter = nullptr;
    CHK(hipExtMallocWithFlags((void **)&counter, 8, hipMallocSignalMemory));
    CHK(hipMemset(counter, 0, 8));
    unsigned long long *t = nullptr;   // t[0] first block of A, t[1] end of A, t[2] kernel B
    unsigned *u = nullptr;             // u[0] blocks of A done, u[1] blocks of the first half done, u[2] counter as B saw it
    CHK(hipMalloc((void **)&t, 32));
    CHK(hipMalloc((void **)&u, 16));
    CHK(hipMemset(t, 0, 32));
    CHK(hipMemset(u, 0, 16));
    hipStream_t s, tt;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&tt, hipStreamNonBlocking));
    const unsigned blocks = 40000, half = 20000;
    const long long spin = 20000;   // wall-clock ticks (100 MHz): 0.2 ms per block; 40 000 blocks on ~2 000 slots: ~4 ms
    CHK(hipDeviceSynchronize());
    k_a<<<blocks, 256, 0, s>>>(counter, half, spin, t, t + 1, u);
    CHK(hipStreamWaitValue32(tt, counter, half, hipStreamWaitValueGte, 0xFFFFFFFFu));
    k_b<<<1, 1, 0, tt>>>(t + 2, counter, u + 2);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(s));
    CHK(hipStreamSynchronize(tt));
    unsigned long long ht[4];
    unsigned hu[4];
    CHK(hipMemcpy(ht, t, 32, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hu, u, 16, hipMemcpyDeviceToHost));
    const double tick_us = 0.01;   // wall_clock64: 100 MHz
    printf("{\"can_use_stream_wait_value\": 1, \"kernel_a_us\": %.1f, \"kernel_b_started_at_us\": %.1f, \"counter_seen_by_b\": %u, \"half\": %u, "
           "\"b_before_end_of_a\": %s}\n",
           (ht[1] - ht[0]) * tick_us, (ht[2] - ht[0]) * tick_us, hu[2], half, ht[2] < ht[1] ? "true" : "false");
    return 0;
}
