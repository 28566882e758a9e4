// sctagger_hip.hip — contexts, seed-table construction, kernel launches and the C-ABI (include/sctagger_hip.h);
// the gfx950 kernels themselves are in the k_*.hip.h headers included below.
//
// match_trie (reference scTagger.py:643-716, observable semantics in SURVEY.md §8(a)-Q): for every long-read
// segment the minimum infix edit distance (<= max_err) to any barcode on either strand, and the sorted list of
// all (bid, strand) at that minimum.  The reference gets there with a trie of segment windows and a recursive
// DFS per barcode; none of that is imitated here.
//
// Device data (all in HBM, resident for the life of the context / upload):
//   patterns   np = 2*n_bc entries, index p = bid*2 + strand (strand 1 = forward, 0 = reverse complement, so
//              ascending p is the reference's sorted(bids) order), stored as bit-planes uint2{lo, hi}
//   seed tables (bl 8..16): per family 8-byte bucket slots and an offsets array (65537 entries; 2^20+1 / 2^22+1
//              offsets, no slots, for the thirds scheme of large barcode sets) and np entries as pattern ids and as
//              2-bit codes, see sct_core.h and k_filter.hip.h
//   segments   the caller's CSR bytes + int64 offsets, plus per segment the packed bit-planes {lo, hi, meta}
//   results    min_e u8[n], cnt u32[n], slot_off u64[n] + dense slots u32[] (filter path), hit_offs i64[n+1],
//              hits u32[total]
//
// Kernels of the match_trie path:
//   k_pack          ASCII -> bit-planes, routes each segment to the filter / exhaustive / restricted-start path
//   k_filter<...>   one wave per 64 segments: seed lookups spread over the lanes, candidates queued in LDS and
//                   verified 64 at a time by the anchored transposed Wu-Manber automaton, tiers, retry launch
//   k_filter_lane   the first, lane-per-segment version of the filter (option filter_mode = 1, for A/B)
//   k_scan<...>     one wave per (segment, barcode slice), lanes = barcodes: Myers bit-vector over every barcode
//                   (any bl <= 32, any max_err) or, for segments with non-ACGT bytes when the reference's prefix
//                   length k > 0, a plain DP whose row 0 encodes the allowed window starts; count + fill pass
//   k_sum_* / k_offsets   exclusive scan of the per-segment hit counts into the CSR offsets
//   k_fill          copies the filter path's hit lists into the CSR hit array
// Kernels of the two neighbouring stages:
//   k_wl_count                        extract_sr_bc_from_lr: whitelist dictionary probes (scTagger.py:818-826)
//   k_adapter_count / k_adapter_fill  extract_lr_bc: adapter alignment over whole reads (scTagger.py:176-196)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sctagger_hip.h"
#include "sct_core.h"
#include "sct_tables.h"

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(SCT_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)


#include "sct_kernels_common.h"
#include "k_pack.hip.h"
#include "k_filter.hip.h"
#include "k_scan.hip.h"
#include "k_offsets.hip.h"
#include "k_whitelist.hip.h"
#include "k_adapter.hip.h"
#include "k_validate.hip.h"


static inline unsigned blocks_for_n(int64_t n, int per) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(SCT_E_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const { return (T *)p; }
};

struct TimedKernel {
    const char *name;
    hipEvent_t a, b;
};

struct sct_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // barcodes
    int n_bc = 0, bl = 0, np = 0;
    bool have_bc = false, have_tables = false, have_thirds = false;
    FamTables ft{};
    DevBuf pat, tab_off, tab_ent, tab_code, tab_slot;
    // segments
    int64_t n_seg = -1, seg_bytes = 0;
    DevBuf segs, offs;
    // work / results
    DevBuf lo, hi, meta, min_e, cnt, route, slots, slots_retry, list_retry, slot_off, list_scan, list_dp, counters, hit_offs, sums, total, hits, part_best, part_cnt;
    int64_t n_hits = 0;
    bool have_result = false;
    // whitelist dictionary (extract_sr_bc_from_lr)
    DevBuf wl_table, wl_counts, wl_first, wl_skip;
    DevBuf ad_peq, ad_strand, ad_dist, ad_locs, ad_first, ad_last;
    uint32_t wl_mask = 0, wl_len_set = 0;
    int32_t wl_n = -1;
    // options
    bool force_exhaustive = false, timing = true, no_thirds = false;
    int slots_mode = -1;       // -1 = by the share of long buckets, 0 = never, 1 = whenever the tables hold them
    double long_share = 1.0;   // share of the seed keys whose bucket does not fit a slot (sct_set_barcodes)
    int filter_mode = 0;  // 0 = wave-cooperative k_filter, 1 = k_filter_lane
    // timing
    std::vector<TimedKernel> timed;
    std::vector<std::pair<const char *, float>> last_ms;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    int64_t stats[5] = {0, 0, 0, 0, 0};

    hipEvent_t ev() {
        if (pool_used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[pool_used++];
    }
};

struct Timer {
    sct_ctx *c;
    TimedKernel t{};
    bool on;
    Timer(sct_ctx *c_, const char *name) : c(c_), on(c_->timing) {
        if (!on) return;
        t.name = name;
        t.a = c->ev();
        t.b = c->ev();
        if (!t.a || !t.b) { on = false; return; }
        (void)hipEventRecord(t.a, c->stream);
    }
    ~Timer() {
        if (!on) return;
        (void)hipEventRecord(t.b, c->stream);
        c->timed.push_back(t);
    }
};

extern "C" const char *sct_last_error(void) { return g_err.c_str(); }

extern "C" int sct_prefix_len(int threads) {
    int k = 0;
    long long n = 1;
    while (threads > n) {
        n *= 4;
        k++;
    }
    return k;
}

extern "C" int sct_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sct_create(int device, sct_ctx **out) {
    if (!out) return fail(SCT_E_ARG, "sct_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SCT_E_NODEV, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(SCT_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    sct_ctx *c = new sct_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(SCT_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return SCT_OK;
}

extern "C" void sct_destroy(sct_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->pat, &c->tab_off, &c->tab_ent, &c->tab_code, &c->tab_slot, &c->segs, &c->offs, &c->lo, &c->hi, &c->meta, &c->min_e,
                      &c->cnt, &c->route, &c->slots, &c->slots_retry, &c->list_retry, &c->slot_off, &c->list_scan, &c->list_dp, &c->counters, &c->hit_offs, &c->sums,
                      &c->total, &c->hits, &c->part_best, &c->part_cnt, &c->wl_table, &c->wl_counts, &c->wl_first, &c->wl_skip, &c->ad_peq, &c->ad_strand, &c->ad_dist, &c->ad_locs, &c->ad_first, &c->ad_last};
    for (DevBuf *b : bufs) b->release();
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// page-locked host memory: buffers allocated here move over PCIe by DMA at the full link rate (pageable
// buffers are staged through a bounce buffer by the runtime)
extern "C" void *sct_host_alloc(sct_ctx *c, size_t bytes) {
    if (!c) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(SCT_E_HIP, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void sct_host_free(sct_ctx *c, void *p) {
    if (!p) return;
    if (c) (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

extern "C" int sct_sync(sct_ctx *c) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SCT_OK;
}

extern "C" int sct_set_option(sct_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return fail(SCT_E_ARG, "sct_set_option: NULL argument");
    if (!strcmp(key, "force_exhaustive")) c->force_exhaustive = value != 0;
    else if (!strcmp(key, "timing")) c->timing = value != 0;
    else if (!strcmp(key, "filter_mode")) c->filter_mode = (int)value;
    else if (!strcmp(key, "no_thirds")) c->no_thirds = value != 0;
    else if (!strcmp(key, "slots")) c->slots_mode = value < 0 ? -1 : (value != 0);
    else return fail(SCT_E_ARG, "unknown option '%s'", key);
    return SCT_OK;
}

// rev_compl of the reference (scTagger.py:148-156) on raw bytes
static inline uint8_t comp_byte(uint8_t b) {
    switch (b) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return b;
    }
}

extern "C" int sct_set_barcodes(sct_ctx *c, const char *bc, int32_t n, int32_t bl) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && !bc)) return fail(SCT_E_ARG, "sct_set_barcodes: bad barcode buffer");
    if (bl < 1) return fail(SCT_E_ARG, "barcode length must be positive (got %d)", bl);
    if (bl > SCT_MAX_BL) return fail(SCT_E_UNSUPPORTED, "barcode length %d exceeds the supported maximum %d", bl, SCT_MAX_BL);
    if ((int64_t)n * 2 > 0x7FFFFFFF) return fail(SCT_E_ARG, "too many barcodes");
    const uint8_t *b = (const uint8_t *)bc;
    for (int64_t i = 0; i < (int64_t)n * bl; i++)
        if (b[i] >= 128) return fail(SCT_E_ARG, "barcode %lld contains byte 0x%02x >= 128", (long long)(i / bl), b[i]);
    HIPCHK(hipSetDevice(c->device));
    const int np = 2 * n;
    std::vector<uint2> pat((size_t)np);
    for (int i = 0; i < n; i++) {
        const uint8_t *s = b + (size_t)i * bl;
        uint32_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
        for (int j = 0; j < bl; j++) {
            const uint32_t cf = sct_code(s[j]);
            const uint32_t cr = sct_code(comp_byte(s[bl - 1 - j]));
            flo |= (cf & 1u) << j;
            fhi |= (cf >> 1) << j;
            rlo |= (cr & 1u) << j;
            rhi |= (cr >> 1) << j;
        }
        pat[(size_t)2 * i + 0] = make_uint2(rlo, rhi);  // strand 0 = reverse complement (False sorts first)
        pat[(size_t)2 * i + 1] = make_uint2(flo, fhi);  // strand 1 = forward
    }
    if (c->pat.ensure(sizeof(uint2) * (size_t)(np ? np : 1))) return SCT_E_HIP;
    if (np) HIPCHK(hipMemcpyAsync(c->pat.p, pat.data(), sizeof(uint2) * (size_t)np, hipMemcpyHostToDevice, c->stream));
    c->have_tables = false;
    c->have_thirds = false;
    if (bl >= 8 && bl <= 16) {
        // seed tables (sct_tables.h): counting sort of the pattern ids by key, per family, plus the bucket slots.
        // The thirds families (20/22-bit keys) are only worth their 38 MB of offsets for very large barcode sets.
        const bool thirds = bl == 16 && np >= THIRDS_MIN_PATTERNS;
        SctSeedTables tb;
        if (!sct_build_tables(pat.data(), np, bl, thirds, tb)) return fail(SCT_E_UNSUPPORTED, "too many barcodes for the seed tables");
        FamTables ft{};
        for (int f = 0; f < tb.nfam; f++) ft.off_base[f] = tb.off_base[f];
        const std::vector<uint32_t> &off = tb.off, &ent = tb.ent, &code = tb.code;
        const std::vector<uint64_t> &slot = tb.slot;
        c->long_share = tb.long_share;
        // one buffer: the slots, then the pattern ids of their inline entries (k_filter.hip.h: SLOT_PID_AT)
        if (slot.size() != SLOT_TOTAL) return fail(SCT_E_STATE, "seed tables: unexpected slot count");
        if (c->tab_slot.ensure(slot.size() * sizeof(uint64_t) + tb.slot_pid.size() * sizeof(uint32_t))) return SCT_E_HIP;
        HIPCHK(hipMemcpyAsync(c->tab_slot.p, slot.data(), slot.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        ft.pid16 = np <= 65536 ? 1u : 0u;
        std::vector<uint16_t> pid16;
        if (ft.pid16) {
            pid16.assign(tb.slot_pid.begin(), tb.slot_pid.end());
            HIPCHK(hipMemcpyAsync((char *)c->tab_slot.p + slot.size() * sizeof(uint64_t), pid16.data(), pid16.size() * sizeof(uint16_t),
                                  hipMemcpyHostToDevice, c->stream));
        } else {
            HIPCHK(hipMemcpyAsync((char *)c->tab_slot.p + slot.size() * sizeof(uint64_t), tb.slot_pid.data(),
                                  tb.slot_pid.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        }
        if (c->tab_off.ensure(off.size() * sizeof(uint32_t) + 16)) return SCT_E_HIP;
        if (c->tab_ent.ensure(ent.size() * sizeof(uint32_t))) return SCT_E_HIP;
        if (c->tab_code.ensure(code.size() * sizeof(uint32_t))) return SCT_E_HIP;
        HIPCHK(hipMemcpyAsync(c->tab_code.p, code.data(), code.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->tab_off.p, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->tab_ent.p, ent.data(), ent.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->ft = ft;
        c->have_tables = true;
        c->have_thirds = thirds;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_bc = n;
    c->bl = bl;
    c->np = np;
    c->have_bc = true;
    c->have_result = false;
    return SCT_OK;
}

extern "C" int sct_upload_segments(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (n < 0 || !offs) return fail(SCT_E_ARG, "sct_upload_segments: bad offsets");
    if (n >= 0x7FFFFFF0) return fail(SCT_E_ARG, "at most 2^31-16 segments per call (shard the input)");
    if (offs[0] != 0) return fail(SCT_E_ARG, "offs[0] must be 0");
    const int64_t bytes = offs[n];
    if (bytes < 0) return fail(SCT_E_ARG, "offs[n] is negative");
    if (bytes > 0 && !segs) return fail(SCT_E_ARG, "segs is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (c->segs.ensure((size_t)bytes + 64)) return SCT_E_HIP;
    if (c->offs.ensure(sizeof(int64_t) * (size_t)(n + 1))) return SCT_E_HIP;
    if (c->counters.ensure(64)) return SCT_E_HIP;
    hipStream_t st = c->stream;
    if (bytes) HIPCHK(hipMemcpyAsync(c->segs.p, segs, (size_t)bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->offs.p, offs, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, st));
    // validation on the device, before any kernel trusts the offsets: non-decreasing, inside the byte buffer, and
    // no byte >= 128 (the reference raises on those); the host only reads back two words
    HIPCHK(hipMemsetAsync(c->counters.p, 0xFF, 16, st));
    const int64_t items = (n + 1) > (bytes + 15) / 16 ? (n + 1) : (bytes + 15) / 16;
    k_validate<<<blocks_for_n(items, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), bytes, c->offs.as<int64_t>(), n,
                                                        c->counters.as<unsigned long long>());
    unsigned long long bad[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(bad, c->counters.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    c->n_seg = -1;
    if (bad[0] != ~0ull) return fail(SCT_E_ARG, "offs must be non-decreasing and end at the size of segs (row %llu)", bad[0]);
    if (bad[1] != ~0ull) return fail(SCT_E_ARG, "segment bytes contain a value >= 128 at offset %llu", bad[1]);
    c->n_seg = n;
    c->seg_bytes = bytes;
    c->have_result = false;
    return SCT_OK;
}

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int sct_run(sct_ctx *c, int32_t max_err, int32_t threads, int64_t *n_hits_out) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (!c->have_bc) return fail(SCT_E_STATE, "sct_run before sct_set_barcodes");
    if (c->n_seg < 0) return fail(SCT_E_STATE, "sct_run before sct_upload_segments");
    if (max_err < 0 || max_err >= c->bl) return fail(SCT_E_ARG, "need 0 <= max_err < barcode length (got %d, bl %d)", max_err, c->bl);
    if (threads < 1) return fail(SCT_E_ARG, "threads must be >= 1");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->n_seg;
    const int np = c->np, bl = c->bl;
    const int prefix_len = sct_prefix_len(threads);
    const bool filter_ok = c->have_tables && max_err <= 2 && !c->force_exhaustive && np > 0 && (uint64_t)SCT_NFAM_ALL * (uint64_t)np < LOC_INLINE;
    const bool use_slots = bl == 16 && (c->slots_mode < 0 ? c->long_share <= (max_err <= 1 ? SLOTS_MAX_LONG_SHARE_E1 : SLOTS_MAX_LONG_SHARE_E2)
                                                          : c->slots_mode == 1);
    const bool use_thirds = filter_ok && c->have_thirds && max_err <= 1 && c->filter_mode != 1 && !c->no_thirds;
    const bool rt_families = bl != 16;  // quarter families with run-time piece boundaries
    const size_t n1 = (size_t)(n ? n : 1);
    if (c->lo.ensure(8 * n1) || c->hi.ensure(8 * n1) || c->meta.ensure(4 * n1) || c->min_e.ensure(n1) ||
        c->cnt.ensure(4 * n1) || c->route.ensure(4 * n1) || c->list_scan.ensure(4 * n1) || c->list_dp.ensure(4 * n1) ||
        c->counters.ensure(64) || c->hit_offs.ensure(8 * (n1 + 1)) || c->total.ensure(8))
        return SCT_E_HIP;
    if (filter_ok && (c->slots.ensure(4 * n1 * K_SLOTS) || c->slot_off.ensure(8 * n1) || c->list_retry.ensure(4 * n1))) return SCT_E_HIP;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (c->sums.ensure(8 * (size_t)(nb ? nb : 1))) return SCT_E_HIP;
    c->timed.clear();
    c->pool_used = 0;
    c->have_result = false;
    hipStream_t st = c->stream;
    HIPCHK(hipMemsetAsync(c->counters.p, 0, 64, st));
    uint32_t *counters = c->counters.as<uint32_t>();
    if (n > 0) {
        {
            Timer t(c, "k_pack");
            k_pack<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), n, bl, max_err, prefix_len,
                                                       !filter_ok ? 0 : ((c->filter_mode == 1 && bl == 16) ? 31 : PLANE_MAX), c->lo.as<uint64_t>(),
                                                       c->hi.as<uint64_t>(),
                                                       c->meta.as<uint32_t>(), c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>(),
                                                       c->list_scan.as<int32_t>(), c->list_dp.as<int32_t>(), counters);
        }
        if (filter_ok) {
            Timer t(c, "k_filter");
            const unsigned g = blocks_for(n, 256);
#define LANE_ARGS                                                                                                     \
    c->lo.as<uint64_t>(), c->hi.as<uint64_t>(), c->meta.as<uint32_t>(), n, c->tab_off.as<uint32_t>(),                  \
        c->tab_ent.as<uint32_t>(), c->pat.as<uint2>(), np, c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>(),              \
        c->slots.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->list_scan.as<int32_t>(), counters
#define WAVE_ARGS                                                                                                     \
    c->lo.as<uint64_t>(), c->hi.as<uint64_t>(), c->meta.as<uint32_t>(), n, (const int32_t *)nullptr,                   \
        c->tab_off.as<uint32_t>(), c->tab_ent.as<uint32_t>(), c->tab_code.as<uint32_t>(), c->tab_slot.as<uint2>(), c->ft, c->min_e.as<uint8_t>(), \
        c->cnt.as<uint32_t>(), c->slots.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->list_retry.as<int32_t>(),       \
        counters + 2, reinterpret_cast<unsigned long long *>(counters + 4), 0ull, bl
#define LAUNCH_WAVE(E, TH)                                                                                            \
    do {                                                                                                              \
        if (rt_families) {                                                                                            \
            k_filter<E, uint32_t, false, K_SLOTS, HMAX, 64, true, false><<<g, 256, 0, st>>>(WAVE_ARGS);                \
            k_filter<E, uint64_t, false, K_SLOTS, HMAX, 64, true, false><<<g, 256, 0, st>>>(WAVE_ARGS);                \
        } else if (use_slots && !TH) {                                                                                \
            k_filter<E, uint32_t, false, HMAX_SL, HMAX_SL, 64, false, true><<<g, 256, 0, st>>>(WAVE_ARGS);             \
            k_filter<E, uint64_t, false, HMAX_SL, HMAX_SL, 64, false, true><<<g, 256, 0, st>>>(WAVE_ARGS);             \
        } else {                                                                                                      \
            k_filter<E, uint32_t, TH, K_SLOTS, HMAX, 64, false, false><<<g, 256, 0, st>>>(WAVE_ARGS);                  \
            k_filter<E, uint64_t, TH, K_SLOTS, HMAX, 64, false, false><<<g, 256, 0, st>>>(WAVE_ARGS);                  \
        }                                                                                                             \
    } while (0)
#define LAUNCH_FILTER(E)                                                                                              \
    do {                                                                                                              \
        if (c->filter_mode == 1 && !rt_families) k_filter_lane<E><<<g, 256, 0, st>>>(LANE_ARGS);                                       \
        else LAUNCH_WAVE(E, false);                                                                                   \
    } while (0)
            if (use_thirds && max_err == 0) LAUNCH_WAVE(0, true);
            else if (use_thirds) LAUNCH_WAVE(1, true);
            else if (max_err == 0) LAUNCH_FILTER(0);
            else if (max_err == 1) LAUNCH_FILTER(1);
            else LAUNCH_FILTER(2);
#undef LAUNCH_FILTER
#undef LAUNCH_WAVE
#undef WAVE_ARGS
#undef LANE_ARGS
        }
    }
    // host sync #1: how many rows go through the scan paths (normally none or a handful) / are to be retried
    uint32_t hc[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(hc, c->counters.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t n_retried = hc[2];
    if (n > 0 && filter_ok && !(c->filter_mode == 1 && !rt_families) && hc[2] > 0) {
        // rows with more ties / verified matches than the main launch keeps: once more with larger lists
        const int64_t nr = hc[2];
        if (c->slots_retry.ensure(4 * (size_t)nr * K_RETRY)) return SCT_E_HIP;
        Timer t(c, "k_filter_retry");
        const unsigned g = blocks_for(nr, 64);  // 4 waves x 16 rows per block
#define RETRY_ARGS                                                                                                    \
    c->lo.as<uint64_t>(), c->hi.as<uint64_t>(), c->meta.as<uint32_t>(), nr, c->list_retry.as<int32_t>(),               \
        c->tab_off.as<uint32_t>(), c->tab_ent.as<uint32_t>(), c->tab_code.as<uint32_t>(), c->tab_slot.as<uint2>(), c->ft, c->min_e.as<uint8_t>(), \
        c->cnt.as<uint32_t>(), c->slots_retry.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->list_scan.as<int32_t>(),  \
        counters + 0, reinterpret_cast<unsigned long long *>(counters + 6), SLOT_TAG_RETRY, bl
#define LAUNCH_RETRY(E, TH)                                                                                           \
    do {                                                                                                              \
        if (rt_families) {                                                                                            \
            k_filter<E, uint32_t, false, K_RETRY, H_RETRY, 16, true, false><<<g, 256, 0, st>>>(RETRY_ARGS);            \
            k_filter<E, uint64_t, false, K_RETRY, H_RETRY, 16, true, false><<<g, 256, 0, st>>>(RETRY_ARGS);            \
        } else if (use_slots && !TH) {                                                                                \
            k_filter<E, uint32_t, false, K_RETRY, H_RETRY_SL, 16, false, true><<<g, 256, 0, st>>>(RETRY_ARGS);            \
            k_filter<E, uint64_t, false, K_RETRY, H_RETRY_SL, 16, false, true><<<g, 256, 0, st>>>(RETRY_ARGS);            \
        } else {                                                                                                      \
            k_filter<E, uint32_t, TH, K_RETRY, H_RETRY, 16, false, false><<<g, 256, 0, st>>>(RETRY_ARGS);              \
            k_filter<E, uint64_t, TH, K_RETRY, H_RETRY, 16, false, false><<<g, 256, 0, st>>>(RETRY_ARGS);              \
        }                                                                                                             \
    } while (0)
        if (use_thirds && max_err == 0) LAUNCH_RETRY(0, true);
        else if (use_thirds) LAUNCH_RETRY(1, true);
        else if (max_err == 0) LAUNCH_RETRY(0, false);
        else if (max_err == 1) LAUNCH_RETRY(1, false);
        else LAUNCH_RETRY(2, false);
#undef LAUNCH_RETRY
#undef RETRY_ARGS
    }
    // The rows of the scan lists are those k_pack routed there (known since sync #1) and, rarely, rows the retry launch
    // could not hold either.  The tail below — scan-path count passes, offsets, sync #2 — first runs with the lists as
    // known; sync #2 also brings the counters back, and only if the retry launch did add rows is the tail run again
    // with them (the count passes rewrite the same values for the rows they had already seen).
    uint32_t *pb_scan = nullptr, *pc_scan = nullptr, *pb_dp = nullptr, *pc_dp = nullptr;
    int sl_scan = 1, sl_dp = 1;
    const uint32_t *route = c->cnt.as<uint32_t>();  // which rows are deferred (CNT_DEFER), as k_fill sees it
    uint64_t total = 0;
    auto slices_for = [&](uint32_t rows) {  // a short list is cut into pattern slices so that the whole chip works on it
        if (rows == 0) return 1;
        const int tiles = (np + 63) / 64;
        int64_t sl = (16384 + rows - 1) / rows;   // aim at 16 k waves in flight
        if (sl < 16) sl = 16;                       // and at least 16 slices: the fill pass skips the slices without a hit
        if (sl > 4096) sl = 4096;
        if (sl > tiles) sl = tiles;
        return (int)(sl < 1 ? 1 : sl);
    };
    auto grid_for = [&](uint32_t rows, int sl) {
        const int64_t waves = (int64_t)rows * sl;
        const int64_t blocks = (waves + 3) / 4;
        return (unsigned)(blocks < 2048 ? (blocks < 1 ? 1 : blocks) : 2048);
    };
    auto tail = [&]() -> int {
        if (n > 0 && (hc[0] || hc[1]) && route == c->cnt.as<uint32_t>()) {
            // remember which rows are deferred: their cnt is about to be overwritten by the count passes (taken once:
            // the marks of the retry launch's rows are already in place when the tail first runs)
            HIPCHK(hipMemcpyAsync(c->route.p, c->cnt.p, 4 * (size_t)n, hipMemcpyDeviceToDevice, st));
            route = c->route.as<uint32_t>();
        }
        sl_scan = slices_for(hc[0]);
        sl_dp = slices_for(hc[1]);
        if (c->part_best.ensure(4 * ((size_t)hc[0] * sl_scan + (size_t)hc[1] * sl_dp + 1)) ||
            c->part_cnt.ensure(4 * ((size_t)hc[0] * sl_scan + (size_t)hc[1] * sl_dp + 1)))
            return SCT_E_HIP;
        pb_scan = c->part_best.as<uint32_t>();
        pc_scan = c->part_cnt.as<uint32_t>();
        pb_dp = pb_scan + (size_t)hc[0] * sl_scan;
        pc_dp = pc_scan + (size_t)hc[0] * sl_scan;
        if (n > 0) {
            if (hc[0]) {
                Timer t(c, "k_scan_count");
                if (bl <= 16)
                    k_scan_pk<false><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                        c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                        np, bl, max_err, c->min_e.as<uint8_t>(), pb_scan, pc_scan, nullptr, nullptr);
                else
                    k_scan<false, false><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                        c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                        np, bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_scan, pc_scan, nullptr, nullptr);
                k_scan_reduce<<<blocks_for(hc[0], 256), 256, 0, st>>>(c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, max_err, pb_scan,
                                                                      pc_scan, c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>());
            }
            if (hc[1]) {
                Timer t(c, "k_dp_count");
                k_scan<true, false><<<grid_for(hc[1], sl_dp), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_dp.as<int32_t>(), (int64_t)hc[1], sl_dp, c->pat.as<uint2>(), np,
                    bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_dp, pc_dp, nullptr, nullptr);
                k_scan_reduce<<<blocks_for(hc[1], 256), 256, 0, st>>>(c->list_dp.as<int32_t>(), (int64_t)hc[1], sl_dp, max_err, pb_dp, pc_dp,
                                                                      c->min_e.as<uint8_t>(), c->cnt.as<uint32_t>());
            }
            {
                Timer t(c, "k_offsets");
                k_sum_blocks<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>());
                k_scan_sums<<<1, 256, 0, st>>>(c->sums.as<uint64_t>(), nb, c->total.as<uint64_t>());
                k_offsets<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>(), c->total.as<uint64_t>(),
                                                        c->hit_offs.as<int64_t>());
            }
        } else {
            HIPCHK(hipMemsetAsync(c->hit_offs.p, 0, 8, st));
            HIPCHK(hipMemsetAsync(c->total.p, 0, 8, st));
        }
        return SCT_OK;
    };
    const uint32_t listed0 = hc[0];
    if (int rc = tail()) return rc;
    // host sync #2: the total hit count sizes the CSR hit array; the counters say whether the retry launch deferred rows
    uint32_t hc2[4] = {hc[0], hc[1], hc[2], 0};
    HIPCHK(hipMemcpyAsync(&total, c->total.p, 8, hipMemcpyDeviceToHost, st));
    if (n > 0) HIPCHK(hipMemcpyAsync(hc2, c->counters.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (hc2[0] != listed0) {
        hc[0] = hc2[0];
        if (int rc = tail()) return rc;
        HIPCHK(hipMemcpyAsync(&total, c->total.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (c->hits.ensure(4 * (size_t)(total ? total : 1))) return SCT_E_HIP;
    if (n > 0 && total > 0) {
        if (filter_ok) {
            Timer t(c, "k_fill");
            k_fill<<<blocks_for(n, 256), 256, 0, st>>>(c->cnt.as<uint32_t>(), c->slots.as<uint32_t>(), c->slots_retry.as<uint32_t>(), c->slot_off.as<uint64_t>(), c->hit_offs.as<int64_t>(),
                                                       route, n, c->hits.as<uint32_t>());
        }
        if (hc[0]) {
            Timer t(c, "k_scan_fill");
            if (bl <= 16)
                k_scan_pk<true><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                    np, bl, max_err, c->min_e.as<uint8_t>(), pb_scan, pc_scan, c->hit_offs.as<int64_t>(), c->hits.as<uint32_t>());
            else
                k_scan<false, true><<<grid_for(hc[0], sl_scan), 256, 0, st>>>(
                    c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_scan.as<int32_t>(), (int64_t)hc[0], sl_scan, c->pat.as<uint2>(),
                    np, bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_scan, pc_scan, c->hit_offs.as<int64_t>(), c->hits.as<uint32_t>());
        }
        if (hc[1]) {
            Timer t(c, "k_dp_fill");
            k_scan<true, true><<<grid_for(hc[1], sl_dp), 256, 0, st>>>(
                c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->list_dp.as<int32_t>(), (int64_t)hc[1], sl_dp, c->pat.as<uint2>(), np,
                bl, max_err, prefix_len, c->min_e.as<uint8_t>(), pb_dp, pc_dp, c->hit_offs.as<int64_t>(), c->hits.as<uint32_t>());
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    c->last_ms.clear();
    for (const TimedKernel &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) c->last_ms.emplace_back(t.name, ms);
    }
    c->n_hits = (int64_t)total;
    c->stats[0] = n - hc[0] - hc[1];
    c->stats[1] = hc[0];
    c->stats[2] = hc[1];
    c->stats[3] = (int64_t)total;
    c->stats[4] = n_retried;
    c->have_result = true;
    if (n_hits_out) *n_hits_out = (int64_t)total;
    return SCT_OK;
}

extern "C" int sct_download(sct_ctx *c, uint8_t *min_e, int64_t *hit_offs, uint32_t *hits) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (!c->have_result) return fail(SCT_E_STATE, "sct_download without a completed sct_run");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->n_seg;
    if (min_e && n) HIPCHK(hipMemcpyAsync(min_e, c->min_e.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (hit_offs) HIPCHK(hipMemcpyAsync(hit_offs, c->hit_offs.p, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, c->stream));
    if (hits && c->n_hits) HIPCHK(hipMemcpyAsync(hits, c->hits.p, 4 * (size_t)c->n_hits, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SCT_OK;
}

extern "C" int sct_match(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n, int32_t max_err, int32_t threads,
                         uint8_t *min_e, int64_t *hit_offs, uint32_t **hits, int64_t *n_hits) {
    if (!hits || !n_hits || !min_e || !hit_offs) return fail(SCT_E_ARG, "sct_match: NULL output pointer");
    *hits = nullptr;
    *n_hits = 0;
    int rc = sct_upload_segments(c, segs, offs, n);
    if (rc) return rc;
    int64_t total = 0;
    rc = sct_run(c, max_err, threads, &total);
    if (rc) return rc;
    uint32_t *h = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(total ? total : 1));
    if (!h) return fail(SCT_E_ARG, "out of host memory for %lld hits", (long long)total);
    rc = sct_download(c, min_e, hit_offs, h);
    if (rc) {
        free(h);
        return rc;
    }
    *hits = h;
    *n_hits = total;
    return SCT_OK;
}

extern "C" void sct_free_hits(sct_ctx *, uint32_t *hits) { free(hits); }

extern "C" int sct_last_timing(sct_ctx *c, const char **names, float *ms, int cap) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    const int k = (int)c->last_ms.size();
    for (int i = 0; i < k && i < cap; i++) {
        if (names) names[i] = c->last_ms[i].first;
        if (ms) ms[i] = c->last_ms[i].second;
    }
    return k;
}

extern "C" int sct_last_stats(sct_ctx *c, int64_t *out, int cap) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    for (int i = 0; i < 5 && i < cap; i++) out[i] = c->stats[i];
    return 5;
}

// ------------------------------------------------------------------------------------------------------------
// extract_sr_bc_from_lr (reference scTagger.py:792-844): whitelist dictionary + counting
// ------------------------------------------------------------------------------------------------------------
static void wl_pack(const uint8_t *key, int len, bool rc, uint32_t k[4]) {
    uint8_t b[16] = {0};
    for (int j = 0; j < len; j++) b[j] = rc ? comp_byte(key[len - 1 - j]) : key[j];
    for (int w = 0; w < 4; w++) k[w] = (uint32_t)b[4 * w] | ((uint32_t)b[4 * w + 1] << 8) | ((uint32_t)b[4 * w + 2] << 16) | ((uint32_t)b[4 * w + 3] << 24);
}

extern "C" int sct_wl_set(sct_ctx *c, const char *keys, const int32_t *key_lens, int32_t n) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && (!keys || !key_lens))) return fail(SCT_E_ARG, "sct_wl_set: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    int64_t total = 0;
    for (int32_t i = 0; i < n; i++) {
        if (key_lens[i] < 0) return fail(SCT_E_ARG, "negative key length");
        if (key_lens[i] > 16) return fail(SCT_E_UNSUPPORTED, "whitelist entry %d has %d characters; at most 16 are supported", i, key_lens[i]);
        total += key_lens[i];
    }
    const uint8_t *kb = (const uint8_t *)keys;
    for (int64_t i = 0; i < total; i++)
        if (kb[i] >= 128) return fail(SCT_E_ARG, "whitelist contains byte 0x%02x >= 128", kb[i]);
    uint64_t cap = 16;
    while (cap < 4ull * (uint64_t)(n > 0 ? n : 1)) cap <<= 1;  // two words per barcode, load factor <= 0.5
    if (cap > (1ull << 31)) return fail(SCT_E_UNSUPPORTED, "whitelist too large");
    std::vector<SctWlSlot> tab((size_t)cap);
    memset(tab.data(), 0, tab.size() * sizeof(SctWlSlot));
    const uint32_t mask = (uint32_t)(cap - 1);
    uint32_t len_set = 0;
    // forward words in order, then reverse complements; adding an existing key replaces its value (add_word)
    for (int pass = 0; pass < 2; pass++) {
        int64_t at = 0;
        for (int32_t i = 0; i < n; i++) {
            const int L = key_lens[i];
            const uint8_t *key = kb + at;
            at += L;
            if (L == 0) continue;  // the empty word cannot be added to the automaton
            uint32_t k[4];
            wl_pack(key, L, pass == 1, k);
            uint32_t idx = sct_wl_hash(k[0], k[1], k[2], k[3], (uint32_t)L) & mask;
            for (;;) {
                SctWlSlot &s = tab[idx];
                if (s.len == 0) {
                    s.len = (uint32_t)L;
                    memcpy(s.k, k, sizeof(k));
                    s.value = pass == 0 ? i : -i;
                    break;
                }
                if (s.len == (uint32_t)L && !memcmp(s.k, k, sizeof(k))) {
                    s.value = pass == 0 ? i : -i;
                    break;
                }
                idx = (idx + 1) & mask;
            }
            len_set |= 1u << L;
        }
    }
    if (c->wl_table.ensure(tab.size() * sizeof(SctWlSlot))) return SCT_E_HIP;
    HIPCHK(hipMemcpyAsync(c->wl_table.p, tab.data(), tab.size() * sizeof(SctWlSlot), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->wl_mask = mask;
    c->wl_len_set = len_set;
    c->wl_n = n;
    return SCT_OK;
}

extern "C" int sct_wl_count(sct_ctx *c, const char *segs, const int64_t *offs, const uint8_t *skip, int64_t n, int64_t *counts,
                            int64_t *first_seen) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (c->wl_n < 0) return fail(SCT_E_STATE, "sct_wl_count before sct_wl_set");
    if (!counts || !first_seen || (n > 0 && !skip)) return fail(SCT_E_ARG, "sct_wl_count: NULL argument");
    int rc = sct_upload_segments(c, segs, offs, n);
    if (rc) return rc;
    const size_t nk = (size_t)(c->wl_n > 0 ? c->wl_n : 1);
    if (c->wl_counts.ensure(4 * nk) || c->wl_first.ensure(4 * nk) || c->wl_skip.ensure((size_t)(n ? n : 1))) return SCT_E_HIP;
    hipStream_t st = c->stream;
    c->timed.clear();
    c->pool_used = 0;
    HIPCHK(hipMemsetAsync(c->wl_counts.p, 0, 4 * nk, st));
    HIPCHK(hipMemsetAsync(c->wl_first.p, 0xFF, 4 * nk, st));
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(c->wl_skip.p, skip, (size_t)n, hipMemcpyHostToDevice, st));
        Timer t(c, "k_wl_count");
        k_wl_count<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), c->wl_skip.as<uint8_t>(), n,
                                                      c->wl_table.as<SctWlSlot>(), c->wl_mask, c->wl_len_set,
                                                      c->wl_counts.as<uint32_t>(), c->wl_first.as<uint32_t>());
    }
    std::vector<uint32_t> hc(nk), hf(nk);
    HIPCHK(hipMemcpyAsync(hc.data(), c->wl_counts.p, 4 * nk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hf.data(), c->wl_first.p, 4 * nk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    c->last_ms.clear();
    for (const TimedKernel &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) c->last_ms.emplace_back(t.name, ms);
    }
    for (int32_t i = 0; i < c->wl_n; i++) {
        counts[i] = hc[(size_t)i];
        first_seen[i] = hf[(size_t)i] == 0xFFFFFFFFu ? -1 : (int64_t)hf[(size_t)i];
    }
    return SCT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// extract_lr_bc (reference scTagger.py:176-196): adapter alignment over whole reads
// ------------------------------------------------------------------------------------------------------------
extern "C" int sct_adapter_align(sct_ctx *c, const char *adapter, int32_t m, const char *reads, const int64_t *offs, int64_t n,
                                 int8_t *strand, int32_t *dist, int64_t *loc_offs, int32_t **locs, int64_t *n_locs) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    if (!adapter || !strand || !dist || !loc_offs || !locs || !n_locs) return fail(SCT_E_ARG, "sct_adapter_align: NULL argument");
    *locs = nullptr;
    *n_locs = 0;
    if (m < 1) return fail(SCT_E_ARG, "adapter length must be positive");
    if (m > 32) return fail(SCT_E_UNSUPPORTED, "adapter length %d exceeds the supported maximum 32", m);
    for (int i = 0; i < m; i++)
        if ((uint8_t)adapter[i] >= 128) return fail(SCT_E_ARG, "adapter contains a byte >= 128");
    int rc = sct_upload_segments(c, reads, offs, n);
    if (rc) return rc;
    // match masks: adapter, its reverse complement (rev_compl, scTagger.py:148-156, :365), and that one reversed
    std::vector<uint32_t> peq(3 * 256, 0u);
    std::vector<uint8_t> a2((size_t)m);
    for (int i = 0; i < m; i++) a2[i] = comp_byte((uint8_t)adapter[m - 1 - i]);
    for (int i = 0; i < m; i++) {
        peq[(uint8_t)adapter[i]] |= 1u << i;
        peq[256 + a2[i]] |= 1u << i;
        peq[512 + a2[m - 1 - i]] |= 1u << i;
    }
    const size_t n1 = (size_t)(n ? n : 1);
    if (c->ad_peq.ensure(peq.size() * 4) || c->ad_strand.ensure(n1) || c->ad_dist.ensure(4 * n1) || c->cnt.ensure(4 * n1) ||
        c->ad_first.ensure(8 * n1) || c->ad_last.ensure(8 * n1) ||
        c->hit_offs.ensure(8 * (n1 + 1)) || c->total.ensure(8))
        return SCT_E_HIP;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (c->sums.ensure(8 * (size_t)(nb ? nb : 1))) return SCT_E_HIP;
    hipStream_t st = c->stream;
    c->timed.clear();
    c->pool_used = 0;
    c->have_result = false;
    HIPCHK(hipMemcpyAsync(c->ad_peq.p, peq.data(), peq.size() * 4, hipMemcpyHostToDevice, st));
    uint64_t total = 0;
    if (n > 0) {
        {
            Timer t(c, "k_adapter_count");
            k_adapter_count<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), n, c->ad_peq.as<uint32_t>(), m,
                                                               c->ad_strand.as<int8_t>(), c->ad_dist.as<int32_t>(), c->cnt.as<uint32_t>(),
                                                               c->ad_first.as<int64_t>(), c->ad_last.as<int64_t>());
        }
        {
            Timer t(c, "k_offsets");
            k_sum_blocks<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>());
            k_scan_sums<<<1, 256, 0, st>>>(c->sums.as<uint64_t>(), nb, c->total.as<uint64_t>());
            k_offsets<<<(unsigned)nb, 256, 0, st>>>(c->cnt.as<uint32_t>(), n, c->sums.as<uint64_t>(), c->total.as<uint64_t>(),
                                                    c->hit_offs.as<int64_t>());
        }
        HIPCHK(hipMemcpyAsync(&total, c->total.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (c->ad_locs.ensure(4 * (size_t)(total ? total : 1))) return SCT_E_HIP;
        if (total) {
            Timer t(c, "k_adapter_fill");
            k_adapter_fill<<<blocks_for(n, 256), 256, 0, st>>>(c->segs.as<uint8_t>(), c->offs.as<int64_t>(), n, c->ad_peq.as<uint32_t>(), m,
                                                              c->ad_strand.as<int8_t>(), c->ad_dist.as<int32_t>(), c->ad_first.as<int64_t>(),
                                                              c->ad_last.as<int64_t>(), c->hit_offs.as<int64_t>(), c->ad_locs.as<int32_t>());
        }
    } else {
        loc_offs[0] = 0;
    }
    int32_t *h = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
    if (!h) return fail(SCT_E_ARG, "out of host memory");
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(strand, c->ad_strand.p, (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(dist, c->ad_dist.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(loc_offs, c->hit_offs.p, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
        if (total) HIPCHK(hipMemcpyAsync(h, c->ad_locs.p, 4 * (size_t)total, hipMemcpyDeviceToHost, st));
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        free(h);
        return fail(SCT_E_HIP, "adapter alignment failed: %s", hipGetErrorString(e));
    }
    c->last_ms.clear();
    for (const TimedKernel &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) c->last_ms.emplace_back(t.name, ms);
    }
    *locs = h;
    *n_locs = (int64_t)total;
    return SCT_OK;
}

extern "C" void sct_free_locs(sct_ctx *, int32_t *locs) { free(locs); }
