// pin_cost.hip — what page-locked host memory costs to get (hipHostMalloc / hipHostFree), by size; and what a CPU memcpy
// into it reaches on one thread.  Sizing input for the context's bounce ring.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    hipFree(0);
    const size_t sizes[] = {1u << 20, 4u << 20, 16u << 20, 64u << 20, 256u << 20};
    char *src = (char *)malloc(256u << 20);
    memset(src, 1, 256u << 20);
    for (int rep = 0; rep < 2; rep++)
        for (size_t s : sizes) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipHostMalloc(&p, s, hipHostMallocDefault);
            double t1 = now();
            if (e != hipSuccess) { printf("{\"bytes\": %zu, \"error\": \"%s\"}\n", s, hipGetErrorString(e)); continue; }
            memcpy(p, src, s);
            double t2 = now();
            memcpy(p, src, s);
            double t3 = now();
            hipHostFree(p);
            double t4 = now();
            printf("{\"rep\": %d, \"bytes\": %zu, \"alloc_ms\": %.3f, \"first_touch_copy_ms\": %.3f, \"copy_ms\": %.3f, \"copy_GBps\": %.1f, \"free_ms\": %.3f}\n", rep, s,
                   t1 - t0, t2 - t1, t3 - t2, s / (t3 - t2) / 1e6, t4 - t3);
        }
    return 0;
}
