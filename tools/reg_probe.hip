#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 230u << 20;
    char *p = (char *)malloc(n);
    memset(p, 65, n);
    hipFree(0);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now_ms();
        hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        double t1 = now_ms();
        void *d = nullptr;
        hipError_t e2 = hipHostGetDevicePointer(&d, p, 0);
        double t2 = now_ms();
        hipError_t e3 = hipHostUnregister(p);
        double t3 = now_ms();
        printf("register %.2f ms (%s), getdevptr %.3f ms (%s, same=%d), unregister %.2f ms (%s)\n", t1 - t0, hipGetErrorString(e), t2 - t1,
               hipGetErrorString(e2), d == (void *)p, t3 - t2, hipGetErrorString(e3));
    }
    return 0;
}
