// valu_peak.hip — integer-VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Question it answers: how many cycles does one SIMD need per wave64 instruction of the integer
// operations the seed-filter kernel is made of, as a function of the number of waves resident on the SIMD?  The
// result sets VALU_PEAK_WINST_S in bench.py.
//
// Method: every wave runs LOOPS iterations of a block of UNROLL x CHAINS instructions on CHAINS independent
// registers (so a wave alone is bounded by issue, not by the dependent-op latency), stamps s_memtime around the
// loop, and the host reports  cycles per wave-instruction per SIMD = (median wave cycles) * / (instructions issued
// by all waves of that SIMD) — i.e. elapsed cycles of the SIMD divided by the wave-instructions it issued.
// The grid is 256 CUs x W blocks of 256 threads, so each SIMD holds W waves (W = 1, 2, 4, 8).
//
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static constexpr int CHAINS = 8, UNROLL = 8;

// one instruction on each of the 8 chains, UNROLL times over, in ONE asm statement (the compiler pads the seam
// between two asm statements with an s_nop, which would take an issue slot); A/B are loop-invariant operands
#define ROUND3(name)                                                                \
    name " %0, %0, %8, %9\n\t" name " %1, %1, %8, %9\n\t" name " %2, %2, %8, %9\n\t" \
    name " %3, %3, %8, %9\n\t" name " %4, %4, %8, %9\n\t" name " %5, %5, %8, %9\n\t" \
    name " %6, %6, %8, %9\n\t" name " %7, %7, %8, %9\n\t"
#define ROUND2(name)                                                                \
    name " %0, %8, %0\n\t" name " %1, %8, %1\n\t" name " %2, %8, %2\n\t"             \
    name " %3, %8, %3\n\t" name " %4, %8, %4\n\t" name " %5, %8, %5\n\t"             \
    name " %6, %8, %6\n\t" name " %7, %8, %7\n\t"
#define X8(x) x x x x x x x x
#define OP3(name)                                                                    \
    asm volatile(X8(ROUND3(name))                                                     \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(a), "v"(b))
#define OP2(name)                                                                    \
    asm volatile(X8(ROUND2(name))                                                     \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(a))

enum Op { AND_OR, LSHLREV, PK_LSHLREV16, PK_ADD_U16, XOR, ADD_U32, BFE, LSHL_OR, OR3, MUL_LO, FMA_F32, PK_FMA_F32, AND, XAD, PERM,
          // round 3: the rest of what the filter kernel is made of (static histogram of its disassembly, tools/valu_mix.py)
          LSHRREV, OR, SUB_U32, MOV, CNDMASK, CMP_GT_U32, BITOP3, LSHL_ADD_U32, ADD3, MAD_I24, MIN_U32, MBCNT_LO, PK_LSHRREV16, LSHRREV_SDWA, N_OPS };
static const char *OP_NAMES[N_OPS] = {"v_and_or_b32", "v_lshlrev_b32", "v_pk_lshlrev_b16", "v_pk_add_u16", "v_xor_b32", "v_add_u32",
                                      "v_bfe_u32", "v_lshl_or_b32", "v_or3_b32", "v_mul_lo_u32", "v_fma_f32", "v_pk_fma_f32",
                                      "v_and_b32", "v_xad_u32", "v_perm_b32",
                                      "v_lshrrev_b32", "v_or_b32", "v_sub_u32", "v_mov_b32", "v_cndmask_b32", "v_cmp_gt_u32", "v_bitop3_b32",
                                      "v_lshl_add_u32", "v_add3_u32", "v_mad_i32_i24", "v_min_u32", "v_mbcnt_lo_u32_b32", "v_pk_lshrrev_b16",
                                      "v_lshrrev_b32_sdwa"};

template <int OP>
__global__ void __launch_bounds__(256) k_issue(int loops, uint32_t seed, unsigned long long *__restrict__ cycles, uint32_t *__restrict__ sink) {
    uint32_t r[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) r[i] = seed * (i + 1) + threadIdx.x;
    uint32_t a = (seed & 3u) + 1u, b = seed | 1u;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS];
    f2 pa = {1.0001f, 0.9999f}, pb = {1e-7f, -1e-7f};
#pragma unroll
    for (int i = 0; i < CHAINS; i++) p[i] = f2{(float)r[i], 1.0f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < loops; l++) {
        {
            if constexpr (OP == AND_OR) OP3("v_and_or_b32");
            else if constexpr (OP == LSHLREV) OP2("v_lshlrev_b32");
            else if constexpr (OP == PK_LSHLREV16) OP2("v_pk_lshlrev_b16");
            else if constexpr (OP == PK_ADD_U16) OP2("v_pk_add_u16");
            else if constexpr (OP == XOR) OP2("v_xor_b32");
            else if constexpr (OP == ADD_U32) OP2("v_add_u32");
            else if constexpr (OP == BFE) OP3("v_bfe_u32");
            else if constexpr (OP == LSHL_OR) OP3("v_lshl_or_b32");
            else if constexpr (OP == OR3) OP3("v_or3_b32");
            else if constexpr (OP == MUL_LO) OP2("v_mul_lo_u32");
            else if constexpr (OP == FMA_F32) OP3("v_fma_f32");
            else if constexpr (OP == AND) OP2("v_and_b32");
            else if constexpr (OP == XAD) OP3("v_xad_u32");
            else if constexpr (OP == PERM) OP3("v_perm_b32");
            else if constexpr (OP == LSHRREV) OP2("v_lshrrev_b32");
            else if constexpr (OP == OR) OP2("v_or_b32");
            else if constexpr (OP == SUB_U32) OP2("v_sub_u32");
            else if constexpr (OP == MIN_U32) OP2("v_min_u32");
            else if constexpr (OP == MBCNT_LO) OP2("v_mbcnt_lo_u32_b32");
            else if constexpr (OP == PK_LSHRREV16) OP2("v_pk_lshrrev_b16");
            else if constexpr (OP == LSHL_ADD_U32) OP3("v_lshl_add_u32");
            else if constexpr (OP == ADD3) OP3("v_add3_u32");
            else if constexpr (OP == MAD_I24) OP3("v_mad_i32_i24");
            else if constexpr (OP == MOV) {
#define ROUNDM "v_mov_b32 %0, %8\n\t v_mov_b32 %1, %8\n\t v_mov_b32 %2, %8\n\t v_mov_b32 %3, %8\n\t v_mov_b32 %4, %8\n\t v_mov_b32 %5, %8\n\t v_mov_b32 %6, %8\n\t v_mov_b32 %7, %8\n\t"
                asm volatile(X8(ROUNDM) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a));
            } else if constexpr (OP == CNDMASK) {
#define ROUNDC "v_cndmask_b32 %0, %0, %8, vcc\n\t v_cndmask_b32 %1, %1, %8, vcc\n\t v_cndmask_b32 %2, %2, %8, vcc\n\t v_cndmask_b32 %3, %3, %8, vcc\n\t v_cndmask_b32 %4, %4, %8, vcc\n\t v_cndmask_b32 %5, %5, %8, vcc\n\t v_cndmask_b32 %6, %6, %8, vcc\n\t v_cndmask_b32 %7, %7, %8, vcc\n\t"
                asm volatile(X8(ROUNDC) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a) : "vcc");
            } else if constexpr (OP == CMP_GT_U32) {
#define ROUNDP "v_cmp_gt_u32 vcc, %8, %0\n\t v_cmp_gt_u32 vcc, %8, %1\n\t v_cmp_gt_u32 vcc, %8, %2\n\t v_cmp_gt_u32 vcc, %8, %3\n\t v_cmp_gt_u32 vcc, %8, %4\n\t v_cmp_gt_u32 vcc, %8, %5\n\t v_cmp_gt_u32 vcc, %8, %6\n\t v_cmp_gt_u32 vcc, %8, %7\n\t"
                asm volatile(X8(ROUNDP) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a) : "vcc");
            } else if constexpr (OP == BITOP3) {
#define ROUNDB "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n\t v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n\t"
                asm volatile(X8(ROUNDB) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));
            } else if constexpr (OP == LSHRREV_SDWA) {
#define ROUNDS "v_lshrrev_b32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t v_lshrrev_b32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                asm volatile(X8(ROUNDS) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a));
            }
            else if constexpr (OP == PK_FMA_F32) {
                asm volatile(X8(ROUND3("v_pk_fma_f32"))
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                             : "v"(pa), "v"(pb));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc ^= r[i] ^ __builtin_bit_cast(uint32_t, p[i].x);
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the chains alive
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*KernelFn)(int, uint32_t, unsigned long long *, uint32_t *);
template <int OP>
static KernelFn kernel_of() { return k_issue<OP>; }
static KernelFn KERNELS[N_OPS] = {kernel_of<0>(), kernel_of<1>(), kernel_of<2>(), kernel_of<3>(), kernel_of<4>(), kernel_of<5>(), kernel_of<6>(),
                                  kernel_of<7>(), kernel_of<8>(), kernel_of<9>(), kernel_of<10>(), kernel_of<11>(), kernel_of<12>(), kernel_of<13>(), kernel_of<14>(),
                                  kernel_of<15>(), kernel_of<16>(), kernel_of<17>(), kernel_of<18>(), kernel_of<19>(), kernel_of<20>(), kernel_of<21>(),
                                  kernel_of<22>(), kernel_of<23>(), kernel_of<24>(), kernel_of<25>(), kernel_of<26>(), kernel_of<27>(), kernel_of<28>()};
static_assert(N_OPS == 29, "KERNELS lists every op");

int main(int argc, char **argv) {
    const int loops = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_cycles;
    uint32_t *d_sink;
    const int max_waves = cus * 4 * 8;
    CHK(hipMalloc(&d_cycles, sizeof(unsigned long long) * max_waves));
    CHK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const double insts_per_wave = (double)loops * UNROLL * CHAINS;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"loops\": %d, \"insts_per_wave\": %.0f, \"rows\": [\n", prop.gcnArchName, cus,
           prop.clockRate, loops, insts_per_wave);
    bool first = true;
    for (int op = 0; op < N_OPS; op++) {
        for (int w = 1; w <= 8; w *= 2) {
            const int blocks = cus * w;  // W blocks of 4 waves per CU -> W waves per SIMD
            const int waves = blocks * 4;
            for (int rep = 0; rep < 2; rep++) {  // first launch warms up
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(KERNELS[op], dim3(blocks), dim3(256), 0, 0, loops, 12345u + rep, d_cycles, d_sink);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
            }
            CHK(hipGetLastError());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> cyc(waves);
            CHK(hipMemcpy(cyc.data(), d_cycles, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
            std::sort(cyc.begin(), cyc.end());
            const double med = (double)cyc[waves / 2];
            // s_memtime ticks at the shader clock (MI355X_MICROARCH.md): a SIMD that holds w waves issued w * insts in `med` cycles
            const double cyc_per_inst_simd = med / (w * insts_per_wave);
            const double winst_per_s_chip = (double)waves * insts_per_wave / (ms * 1e-3);
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"median_wave_cycles\": %.0f, "
                   "\"cycles_per_wave_inst_per_simd\": %.3f, \"wave_insts_per_s_chip\": %.4g, \"clock_ghz_implied\": %.3f}",
                   first ? "" : ",\n", OP_NAMES[op], w, ms, med, cyc_per_inst_simd, winst_per_s_chip, med / (ms * 1e-3) / 1e9);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
