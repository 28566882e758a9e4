// launch_chain.hip — what a chain of small dependent kernels costs on one stream, launched one by one and replayed as a captured
// hipGraph: the tail of a batch of the match_trie path is such a chain (retry / special launches that find no rows, sums, fill).
// Sizing input for DESIGN.md 13.8 ("small batches are launch-bound").   hipcc -O2 --offload-arch=gfx950 launch_chain.hip && ./a.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(const unsigned *n, unsigned *out) {
    if (*n == 0u) return;
    out[blockIdx.x] = threadIdx.x;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    unsigned *d;
    CK(hipMalloc(&d, 1 << 20));
    CK(hipMemset(d, 0, 1 << 20));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int grids[] = {1, 64, 1024, 8192};
    const int chains[] = {1, 4, 8, 16};
    for (int g : grids)
        for (int len : chains) {
            const int reps = 200;
            // one by one
            for (int i = 0; i < 20; i++) k_empty<<<g, 64, 0, st>>>(d, d + 64);
            CK(hipStreamSynchronize(st));
            double t0 = now();
            for (int r = 0; r < reps; r++) {
                for (int i = 0; i < len; i++) k_empty<<<g, 64, 0, st>>>(d, d + 64);
                CK(hipStreamSynchronize(st));
            }
            const double one = (now() - t0) / reps;
            // the same chain as a graph
            hipGraph_t graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < len; i++) k_empty<<<g, 64, 0, st>>>(d, d + 64);
            CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int i = 0; i < 5; i++) CK(hipGraphLaunch(exec, st));
            CK(hipStreamSynchronize(st));
            t0 = now();
            for (int r = 0; r < reps; r++) {
                CK(hipGraphLaunch(exec, st));
                CK(hipStreamSynchronize(st));
            }
            const double gr = (now() - t0) / reps;
            printf("{\"grid\": %d, \"chain\": %d, \"stream_us\": %.2f, \"graph_us\": %.2f, \"per_kernel_stream_us\": %.2f, \"per_kernel_graph_us\": %.2f}\n", g, len, one, gr,
                   one / len, gr / len);
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
        }
    return 0;
}
