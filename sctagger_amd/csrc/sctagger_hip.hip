// sctagger_hip.hip — contexts, lanes, seed-table construction, kernel launches and the C-ABI (include/sctagger_hip.h);
// the gfx950 kernels themselves are in the k_*.hip.h headers included below and, for the seed filter, in the three
// translation units filter_e0/1/2.hip.
//
// match_trie (reference scTagger.py:643-716, observable semantics in SURVEY.md §8(a)-Q): for every long-read
// segment the minimum infix edit distance (<= max_err) to any barcode on either strand, and the sorted list of
// all (bid, strand) at that minimum.  The reference gets there with a trie of segment windows and a recursive
// DFS per barcode; none of that is imitated here.
//
// Device data:
//   patterns   np = 2*n_bc entries, index p = bid*2 + strand (strand 1 = forward, 0 = reverse complement, so
//              ascending p is the reference's sorted(bids) order), stored as bit-planes uint2{lo, hi}
//   seed tables (bl 8..16): per family 8-byte bucket slots and an offsets array (65537 entries; 2^20+1 / 2^22+1
//              offsets, no slots, for the thirds scheme of large barcode sets) and np entries as pattern ids and as
//              2-bit codes, see sct_core.h and k_filter.hip.h
//   per LANE (a stream with its own buffers; lane 0 serves the resident three-step API and the one zero-copy chunk a stream
//   call on page-locked buffers normally is; lanes 0..L-1, option "lanes", the chunks of a stream call that has to upload):
//     segments   the rows' ASCII bytes + CSR offsets (int64 as uploaded, or uint32 relative to the chunk)
//     results    min_e u8[n], cnt u32[n], wave_base u64[n/64] (+ slot_off u64[n] for retried rows) + dense slots u32[] (filter path), hit_offs i64[n+1],
//                hits u32[total], and a 64-byte counter record (CTR_*) the host reads back once per batch
//
// A batch (all resident rows, or one chunk) is queued WITHOUT any host synchronisation: main filter launch, retry launch
// (its row count is read on the device), offsets scan, fill, read-back of the counter record.  Only when that record
// says that rows went to the lists the filter cannot serve — longer than 63 characters, bytes outside ACGT under the
// reference's prefix rule, more ties than the retry launch keeps — does the host run the scan kernels for them
// (slow_path) and redo offsets and fill.
//
// Kernels of the match_trie path:
//   k_filter<...>   one wave per 64 rows: ASCII -> bit-planes (k_decode.hip.h), routing, seed lookups spread over the
//                   lanes, candidates queued in LDS and verified 128 at a time by the anchored transposed Wu-Manber
//                   automaton, tiers, retry launch
//   k_scan<...>     one wave per (row, barcode slice), lanes = barcodes: Myers bit-vector over every barcode
//                   (any bl <= 32, any max_err) or, for rows with non-ACGT bytes when the reference's prefix
//                   length k > 0, a plain DP whose row 0 encodes the allowed window starts; count + fill pass
//   k_sum_blocks / k_scan_sums / k_offsets   exclusive scan of the per-row hit counts into the CSR offsets
//   k_fill          copies the filter path's hit lists into the CSR hit array
//   k_index_*       stream path: the caller's row index (int64 / int32 offsets, uint8 / uint16 lengths) -> uint32 offsets
// Kernels of the two neighbouring stages:
//   k_wl_count                        extract_sr_bc_from_lr: whitelist dictionary probes (scTagger.py:818-826)
//   k_adapter_count / k_adapter_fill  extract_lr_bc: adapter alignment over whole reads (scTagger.py:176-196)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sctagger_hip.h"
#include "sct_copy_pool.h"
#include "sct_core.h"
#include "sct_tables.h"

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local int g_code = SCT_OK;  // the code of the last fail(): what a failed DevBuf::ensure makes its caller return
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    g_code = code;
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(SCT_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#include "sct_kernels_common.h"
#include "k_decode.hip.h"
#include "k_filter.hip.h"
#include "filter_launch.h"
#include "k_join.hip.h"
#include "k_scan.hip.h"
#include "k_offsets.hip.h"
#include "k_whitelist.hip.h"
#include "k_adapter.hip.h"
#include "k_validate.hip.h"

static inline unsigned blocks_for(int64_t n, int per) {
    const int64_t b = (n + per - 1) / per;
    return (unsigned)(b > 0 ? b : 1);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
// Fault injection for the tests of the error paths (option "fail_alloc_at" = k: the k-th allocation attempt the context makes from
// now on — hipMalloc in DevBuf::ensure, hipHostMalloc, hipHostRegister, a lane's stream — fails once, as if memory had run out;
// option "alloc_calls" reads how many attempts there were).  The counters live in the context the calling thread's API call
// belongs to (FaultScope); outside an API call nothing is counted.
struct FaultCounters {
    int64_t fail_at = 0;   // > 0: attempts left until the injected failure
    int64_t calls = 0;     // attempts since "fail_alloc_at" was last set
};
static thread_local FaultCounters *t_fault = nullptr;
struct FaultScope {
    FaultCounters *prev;
    explicit FaultScope(FaultCounters *f) : prev(t_fault) { t_fault = f; }
    ~FaultScope() { t_fault = prev; }
};
static bool inject_alloc_failure() {
    if (!t_fault) return false;
    t_fault->calls++;
    return t_fault->fail_at > 0 && --t_fault->fail_at == 0;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        }
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

struct TimedSpan {
    const char *name;
    hipEvent_t a, b;
};

// What one batch is run with (set when it is queued, used again by the slow path).
struct BatchParams {
    int max_err = 0, threads = 1, prefix_len = 0;
    bool filter_ok = false, use_slots = false, use_thirds = false, use_wide = false, rt_families = false;
    bool use_join = false;   // thirds scheme at max_err = 1: the 32-bit main launch is the partitioned join (k_join.hip.h)
    int count_format = SCT_CNT_OFFS64;  // SCT_CNT_*: which compact form of the counts the offsets kernel also writes
};

// A lane: one stream with every buffer a batch of rows needs.
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t filters_done = nullptr;   // option "chunk_order": recorded behind a chunk's filter launches ...
    hipEvent_t wait_for = nullptr;       // ... and what the next chunk's filter launches wait for (set per batch by the stream call)
    bool record_filters_done = false;
    uint32_t *h_ctr = nullptr;  // page-locked copy of the counter record
    // rows
    int64_t n_seg = -1, seg_bytes = 0;
    bool wide = true;           // offsets are int64 in offs64 (resident upload); else uint32 in offs32 (stream chunk)
    bool have_offs64 = false;   // offs64 holds this batch's offsets (uploaded, or widened by the slow path)
    const uint8_t *segs_ptr = nullptr;  // the rows' bytes as the kernels see them: `segs` below, or the caller's page-locked buffer itself
    const char *host_segs = nullptr;    // zero-copy batches: the same bytes on the host (the slow path copies them to `segs`)
    DevBuf segs, offs64, offs32, index_in;
    // work / results
    DevBuf min_e, cnt, route, slots, slots_retry, list_retry, list_special, list_special2, rec_special, rec_special2, slot_off, wave_base, list_long, list_scan, list_dp, counters, hit_offs, sums, hits,
        part_best, part_cnt, cnt_out;
    DevBuf gather;      // zero-copy batches: the listed rows of more than PLANE_MAX characters, copied to the device (k_special_gather)
    DevBuf arena_cur;   // SLOT_ARENAS cursors of the main launch's slot array, 64 bytes apart (sct_slot_arena)
    DevBuf j_planes, j_len, j_wave, j_cnt, j_slots, j_tup0, j_cnt0, j_tup1, j_cnt1;   // the partitioned join (k_join.hip.h)
    uint64_t slots_cap = 0, retry_cap = 0;
    uint32_t rec_cap = 0;       // entries of rec_special / rec_special2 (SctSpecialRec): the first rec_cap rows of list_special / list_special2
    // where the kernels write the results the caller sees: the lane's own buffers (copied out afterwards) or, for a stream
    // call whose output arrays are page-locked, those arrays themselves through their device mapping
    uint8_t *min_e_ptr = nullptr;
    uint8_t *cnt8_ptr = nullptr;
    uint32_t *cnt32_ptr = nullptr;
    uint32_t *hits_ptr = nullptr;
    uint64_t hits_cap = 0;      // entries hits_ptr can take
    uint16_t *hits16_ptr = nullptr;  // 16-bit hits of a one-chunk stream call: the fill kernel writes them itself (hits_cap = their room)
    bool direct_min_e = false, direct_cnt = false, direct_hits = false;
    bool no64 = false;          // this batch was queued without the 64-bit instantiations of the filter launches (sct_ctx::need64)
    bool no_special = false;    // ... without the special launch (sct_ctx::need_special)
    bool cnt8_soft = false;     // a row with more than 255 hits is not an error of this batch: the one-byte counts go into a slot of the ring, whose
                                // way out takes the chunk's 64-bit offsets instead (sct_stream.hip.h)
    BatchParams bp;
    int64_t n_hits = 0;
    int64_t stats[6] = {0, 0, 0, 0, 0, 0};
    bool have_result = false;
    // timing
    std::vector<TimedSpan> timed;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    hipEvent_t last_end = nullptr;   // the event that closed the last span (Timer): a span that follows it directly starts there
    bool index_just_timed = false;   // the span "k_index" was the last thing queued (sct_match_stream): "k_filter" follows it directly

    hipEvent_t ev() {
        if (pool_used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[pool_used++];
    }
    uint32_t *ctr() const { return counters.as<uint32_t>(); }
    void release() {
        DevBuf *bufs[] = {&segs, &offs64, &offs32, &index_in, &min_e, &cnt, &route, &slots, &slots_retry, &list_retry, &list_special, &list_special2, &rec_special, &rec_special2, &slot_off, &wave_base, &list_long,
                          &list_scan, &list_dp, &counters, &hit_offs, &sums, &hits, &part_best, &part_cnt, &cnt_out,
                          &gather, &arena_cur, &j_planes, &j_len, &j_wave, &j_cnt, &j_slots, &j_tup0, &j_cnt0, &j_tup1, &j_cnt1};
        for (DevBuf *b : bufs) b->release();
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        pool.clear();
        if (done) (void)hipEventDestroy(done);
        if (filters_done) (void)hipEventDestroy(filters_done);
        filters_done = nullptr;
        if (h_ctr) (void)hipHostFree(h_ctr);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        done = nullptr;
        h_ctr = nullptr;
    }
};

static constexpr int STREAM_LANES_MAX = 8;

// A slot of the context's page-locked ring (sct_stream.hip.h): ONE hipHostMalloc holding a chunk's segment bytes, its index as the
// copy threads narrowed it, and what the kernels write back per row (min_e, one-byte counts) plus the device's block sums.
struct RingSlot {
    char *base = nullptr;     // host address
    char *dev = nullptr;      // the same memory as the device addresses it
    size_t bytes = 0;
    size_t seg_cap = 0;       // segment bytes the slot takes (at offset 0)
    int64_t row_cap = 0;      // rows
    size_t o_idx = 0, o_isums = 0, o_min = 0, o_cnt = 0, o_sums = 0;   // offsets of the index, its block prefix sums, min_e, counts and the hit counts' block sums
};

struct sct_ctx {
    int device = 0;
    // barcodes
    int n_bc = 0, bl = 0, np = 0;
    bool have_bc = false, have_tables = false, have_thirds = false, have_wide = false, have_fb = false, have_wfb = false;
    uint32_t max_bucket = 0;  // longest bucket of the quarter families (SLOTS_MAX_BUCKET)
    int last_scheme = SCT_SCHEME_NONE;  // seed scheme of the last match call (sct_last_stats [5])
    int64_t last_join = 0;   // sct_last_stats [6]: batches of the last call the partitioned join did | batches it handed back (overflow) << 32
    int tables_nfam = 0;   // families the seed tables hold (entry indices run up to tables_nfam * np)
    FamTables ft{};
    DevBuf pat, tab_off, tab_ent, tab_code, tab_slot, tab_grp, tab_ent16, tab_fb, tab_wfb;
    std::vector<Lane *> lanes;  // lanes[0] always exists
    // stream API: page-locked result buffer, owned by the context, valid until the next stream call
    uint32_t *h_hits = nullptr;
    size_t h_hits_cap = 0;  // entries
    // stream API on pageable caller arrays: the page-locked ring, the threads that copy into and out of it, a page-locked scratch
    std::vector<RingSlot> ring;
    CopyPool *pool = nullptr;
    void *h_scratch = nullptr;
    size_t h_scratch_cap = 0;
    // two page-locked pieces through which every other copy between the device and host memory that is not page-locked goes (copy_up / copy_down)
    void *bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    // whitelist dictionary (extract_sr_bc_from_lr) and adapter alignment (extract_lr_bc): lane 0's rows
    DevBuf wl_table, wl_counts, wl_first, wl_skip;
    DevBuf ad_peq, ad_strand, ad_dist, ad_locs, ad_first, ad_last;
    uint32_t wl_mask = 0, wl_len_set = 0;
    int32_t wl_n = -1;
    // options
    bool force_exhaustive = false, timing = true, no_thirds = false, no_gather = false;
    int thirds_mode = -1;  // -1 = thirds tables by the size of the set, 1 = built whatever the size (next sct_set_barcodes)
    int fb_mode = -1;      // thirds families: -1 / 1 = lookups read the fixed buckets, 0 = offsets + codes
    int wide_mode = -1;    // -1 = by the size of the barcode set, 0 = never, 1 = build and use the wide families whatever the size
    int join_mode = 0;     // thirds scheme at max_err = 1: 1 = the partitioned join (k_join.hip.h) does the 32-bit rows, 0 (default) = k_filter does:
                           // the join halves the HBM traffic and takes the same time, so it is an option, not the default
    int slots_mode = -1;       // -1 = by the share of long buckets, 0 = never, 1 = whenever the tables hold them
    double long_share = 1.0;   // share of the seed keys whose bucket does not fit a slot (sct_set_barcodes)
    bool zero_copy = true;     // stream API: kernels read page-locked caller buffers in place instead of uploading them
    bool use_ring = true;      // stream API: pageable caller arrays go through the context's page-locked ring (sct_stream.hip.h)
    int ring_slots = 0;        // slots of the ring; 0 = lanes in use + 2
    int ring_lanes = 4;        // chunks of the ring route in flight on the device (profiles/r05_pageable_lanes_sweep.jsonl: 5.7 ms per 10 M rows; 3: 5.95, 5: 6.1)
    // Rows of 32 .. 63 characters, and the 63-character windows of longer rows, take the 64-bit instantiations of the filter launches:
    // four launches of a batch's seventeen that find nothing to do on 22-character segments — and that, being 8 192 one-wave blocks
    // each, wait 100 - 240 us for the chip when another chunk's main launch holds it (profiles/r05_ring_kernel_timeline.txt).  Until a
    // batch of this context has needed them they are not launched: the 32-bit main launch marks CTR_NEED64 when it meets such a row,
    // and finish_batch then does the batch again with both widths and stops speculating.  Option "need64": -1 = this, 1 = always both.
    bool need64 = false;
    int need64_mode = -1;
    // The same for the special launch (gather + its two stages: three dispatches that find no rows on clean input): not launched until a
    // batch of the context has listed a row for it (a byte outside ACGT under the reference's prefix rule, more than 63 characters);
    // that batch is then done again.  Option "need_special": -1 = this, 1 = always launched.
    bool need_special = false;
    int need_special_mode = -1;
    int64_t redone64 = 0;      // batches done again because the speculation failed (sct_get_option "redone64")
    bool chunk_order = false;  // zero-copy / staged chunks: a chunk's filter launches wait for the previous chunk's (experiment)
    int64_t ring_rows = 0;     // rows per chunk of the ring route; 0 = 2^20 (a third of the call's rows if that is less, at least 2^17)
    int ring_ramp = 0;         // r > 0: the first chunk holds 1 / 2^r of that, the next ones double
    int copy_threads = -1;     // threads of the copy pool; -1 = half of the CPUs the process may use, at most 8
    int last_route = 0;        // route of the last stream call: 0 staged copies, 1 zero copy, 2 ring
    int stream_lanes = 5;      // chunks of a stream call in flight (each on a lane of its own)
    int64_t chunk_rows = 0;    // rows per chunk of the stream API; 0 = chosen by where the caller's buffers live
    // what the last run / stream call measured and counted
    struct Span {
        double ms = 0;
        int launches = 0;
    };
    FaultCounters fault;       // test-only fault injection (options "fail_alloc_at", "alloc_calls")
    std::vector<std::pair<const char *, Span>> last_ms;
    int64_t stats[6] = {0, 0, 0, 0, 0, 0};   // rows by path: filter, exhaustive scan, restricted-start DP; hits; rows retried; rows of the special launch

    Lane &lane0() { return *lanes[0]; }
};

// follows_previous: nothing was queued on the lane's stream since the previous span ended — this one starts at that span's end event.
// (An event record between two kernels is a barrier packet: the kernel trace shows 10.4 us between two spans that each record their
// own, 12 records and 62 us per batch; with shared boundaries 7 records.)
struct Timer {
    Lane *l;
    TimedSpan t{};
    bool on;
    Timer(sct_ctx *c, Lane &lane, const char *name, bool follows_previous = false) : l(&lane), on(c->timing) {
        if (!on) return;
        t.name = name;
        const bool shared = follows_previous && l->last_end != nullptr;
        t.a = shared ? l->last_end : l->ev();
        t.b = l->ev();
        if (!t.a || !t.b) { on = false; return; }
        if (!shared) (void)hipEventRecord(t.a, l->stream);
    }
    ~Timer() {
        if (!on) return;
        (void)hipEventRecord(t.b, l->stream);
        l->last_end = t.b;
        l->timed.push_back(t);
    }
};

// fold a lane's recorded spans into the context's table (the lane's stream must be idle)
static void collect_timing(sct_ctx *c, Lane &l) {
    for (const TimedSpan &t : l.timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) continue;
        bool found = false;
        for (auto &kv : c->last_ms)
            if (!strcmp(kv.first, t.name)) {
                kv.second.ms += ms;
                kv.second.launches++;
                found = true;
                break;
            }
        if (!found) c->last_ms.emplace_back(t.name, sct_ctx::Span{(double)ms, 1});
    }
    l.timed.clear();
    l.pool_used = 0;
    l.last_end = nullptr;
    l.index_just_timed = false;
}

static int lane_create(Lane **out) {
    Lane *l = new Lane();
    hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l->filters_done, hipEventDisableTiming);
    if (e == hipSuccess) e = inject_alloc_failure() ? hipErrorOutOfMemory : hipHostMalloc((void **)&l->h_ctr, sizeof(uint32_t) * CTR_WORDS, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        l->release();
        delete l;
        return fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "creating a lane: %s", hipGetErrorString(e));
    }
    *out = l;
    return SCT_OK;
}

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

extern "C" int sct_host_cpus(void) { return sct_effective_cpus(); }

extern "C" int sct_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sct_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 16) return fail(SCT_E_ARG, "sct_device_pci_bus_id: buffer of at least 16 bytes needed");
    buf[0] = 0;
    if (hipDeviceGetPCIBusId(buf, len, device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SCT_E_ARG, "no PCI bus id for device %d", device);
    }
    return SCT_OK;
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
    Lane *l = nullptr;
    if (int rc = lane_create(&l)) {
        delete c;
        return rc;
    }
    c->lanes.push_back(l);
    *out = c;
    return SCT_OK;
}

extern "C" void sct_destroy(sct_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (Lane *l : c->lanes) {
        if (l->stream) (void)hipStreamSynchronize(l->stream);
        l->release();
        delete l;
    }
    DevBuf *bufs[] = {&c->pat, &c->tab_off, &c->tab_ent, &c->tab_code, &c->tab_slot, &c->tab_grp, &c->tab_ent16, &c->tab_fb, &c->tab_wfb, &c->wl_table, &c->wl_counts, &c->wl_first, &c->wl_skip,
                      &c->ad_peq, &c->ad_strand, &c->ad_dist, &c->ad_locs, &c->ad_first, &c->ad_last};
    for (DevBuf *b : bufs) b->release();
    if (c->h_hits) (void)hipHostFree(c->h_hits);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    for (int i = 0; i < 2; i++) {
        if (c->bounce[i]) (void)hipHostFree(c->bounce[i]);
        if (c->bounce_ev[i]) (void)hipEventDestroy(c->bounce_ev[i]);
    }
    delete c->pool;
    for (RingSlot &s : c->ring)
        if (s.base) (void)hipHostFree(s.base);
    delete c;
}

// page-locked host memory: buffers allocated here move over PCIe by DMA at the full link rate and asynchronously
// (copies from or to pageable buffers are staged through a bounce buffer by the runtime and block the calling thread)
extern "C" void *sct_host_alloc(sct_ctx *c, size_t bytes) {
    if (!c) return nullptr;
    FaultScope fs(&c->fault);
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    const hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));   // sct_last_error tells which
        return nullptr;
    }
    return p;
}

extern "C" void sct_host_free(sct_ctx *c, void *p) {
    if (!p) return;
    if (c) (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

extern "C" int sct_get_option(sct_ctx *c, const char *key, int64_t *value) {
    if (!c || !key || !value) return fail(SCT_E_ARG, "sct_get_option: NULL argument");
    if (!strcmp(key, "fail_alloc_at")) *value = c->fault.fail_at;
    else if (!strcmp(key, "alloc_calls")) *value = c->fault.calls;
    else if (!strcmp(key, "join")) *value = c->join_mode;
    else if (!strcmp(key, "lanes")) *value = c->stream_lanes;
    else if (!strcmp(key, "chunk_rows")) *value = c->chunk_rows;
    else if (!strcmp(key, "zero_copy")) *value = c->zero_copy;
    else if (!strcmp(key, "force_exhaustive")) *value = c->force_exhaustive;
    else if (!strcmp(key, "ring")) *value = c->use_ring;
    else if (!strcmp(key, "ring_rows")) *value = c->ring_rows;
    else if (!strcmp(key, "ring_slots")) *value = c->ring_slots;
    else if (!strcmp(key, "copy_threads")) *value = c->pool ? c->pool->threads() : c->copy_threads;
    else if (!strcmp(key, "last_route")) *value = c->last_route;
    else if (!strcmp(key, "need64")) *value = c->need64 || c->need64_mode > 0;
    else if (!strcmp(key, "need_special")) *value = c->need_special || c->need_special_mode > 0;
    else if (!strcmp(key, "redone64")) *value = c->redone64;
    else if (!strcmp(key, "ring_bytes")) {
        size_t b = 0;
        for (const RingSlot &sl : c->ring) b += sl.bytes;
        *value = (int64_t)b;
    }
    else if (!strcmp(key, "device_free_bytes") || !strcmp(key, "device_total_bytes")) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemGetInfo(&fr, &tot));
        *value = (int64_t)(key[7] == 'f' ? fr : tot);
    } else return fail(SCT_E_ARG, "unknown option '%s'", key);
    return SCT_OK;
}

extern "C" int sct_sync(sct_ctx *c) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    for (Lane *l : c->lanes) HIPCHK(hipStreamSynchronize(l->stream));
    return SCT_OK;
}

extern "C" int sct_set_option(sct_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return fail(SCT_E_ARG, "sct_set_option: NULL argument");
    if (!strcmp(key, "force_exhaustive")) c->force_exhaustive = value != 0;
    else if (!strcmp(key, "timing")) c->timing = value != 0;
    else if (!strcmp(key, "no_thirds")) c->no_thirds = value != 0;
    else if (!strcmp(key, "thirds")) c->thirds_mode = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "fixed_buckets")) c->fb_mode = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "wide")) c->wide_mode = value < 0 ? -1 : (value != 0);  // takes effect at the next sct_set_barcodes
    else if (!strcmp(key, "join")) c->join_mode = value > 0 ? 1 : 0;
    else if (!strcmp(key, "slots")) c->slots_mode = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "chunk_rows")) {
        if (value < 0 || value > 0x7FFFFFF0) return fail(SCT_E_ARG, "chunk_rows must be in 0 .. 2^31-16");
        c->chunk_rows = value;
    } else if (!strcmp(key, "zero_copy")) c->zero_copy = value != 0;
    else if (!strcmp(key, "gather")) c->no_gather = value == 0;
    else if (!strcmp(key, "ring")) c->use_ring = value != 0;
    else if (!strcmp(key, "ring_ramp")) {
        if (value < 0 || value > 6) return fail(SCT_E_ARG, "ring_ramp must be in 0 .. 6");
        c->ring_ramp = (int)value;
    }
    else if (!strcmp(key, "chunk_order")) c->chunk_order = value != 0;
    else if (!strcmp(key, "need64")) {
        c->need64_mode = value > 0 ? 1 : -1;
        if (value <= 0) c->need64 = false;   // (start speculating again)
    } else if (!strcmp(key, "need_special")) {
        c->need_special_mode = value > 0 ? 1 : -1;
        if (value <= 0) c->need_special = false;
    }
    else if (!strcmp(key, "ring_lanes")) {
        if (value < 1 || value > STREAM_LANES_MAX) return fail(SCT_E_ARG, "ring_lanes must be in 1 .. %d", STREAM_LANES_MAX);
        c->ring_lanes = (int)value;
    }
    else if (!strcmp(key, "ring_rows")) {
        if (value < 0 || value > ((int64_t)1 << 24)) return fail(SCT_E_ARG, "ring_rows must be in 0 .. 2^24");
        c->ring_rows = value;
    } else if (!strcmp(key, "ring_slots")) {
        if (value < 0 || value > 32) return fail(SCT_E_ARG, "ring_slots must be in 0 .. 32");
        c->ring_slots = (int)value;
    } else if (!strcmp(key, "copy_threads")) {
        if (value < -1 || value > 64) return fail(SCT_E_ARG, "copy_threads must be in -1 .. 64");
        c->copy_threads = (int)value;
        delete c->pool;   // (idle between calls: a context is used by one host thread at a time)
        c->pool = nullptr;
    }
    else if (!strcmp(key, "fail_alloc_at")) {   // tests of the error paths: see FaultCounters
        if (value < 0) return fail(SCT_E_ARG, "fail_alloc_at must be >= 0");
        c->fault.fail_at = value;
        c->fault.calls = 0;
    } else if (!strcmp(key, "lanes")) {
        if (value < 1 || value > STREAM_LANES_MAX) return fail(SCT_E_ARG, "lanes must be in 1 .. %d", STREAM_LANES_MAX);
        c->stream_lanes = (int)value;
    } else return fail(SCT_E_ARG, "unknown option '%s'", key);
    return SCT_OK;
}


// ------------------------------------------------------------------------------------------------------------
// Copies between the device and host memory that is NOT page-locked.  Handed to hipMemcpy as it is, such memory is page-locked in
// place by the runtime for the duration of a large copy (the same mechanism as hipHostRegister, whose use on caller arrays ended
// soak runs in GPU memory faults in round 4: DESIGN.md 5c) or staged on the calling thread.  So nothing but page-locked memory ever
// reaches the runtime from this library: other memory goes through two page-locked pieces of the context, filled and emptied by
// memcpy on the copy pool.  (The stream seam has its own ring: sct_stream.hip.h.)
// ------------------------------------------------------------------------------------------------------------
static constexpr size_t BOUNCE_BYTES = (size_t)4 << 20;
static constexpr size_t BOUNCE_DIRECT_MAX = (size_t)64 << 10;   // up to this size the runtime copies through its own staging buffer

static void par_copy(sct_ctx *c, void *dst, const void *src, size_t bytes);

static bool is_page_locked(const void *p) {
    hipPointerAttribute_t at;
    const bool yes = p && hipPointerGetAttributes(&at, p) == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice);
    (void)hipGetLastError();
    return yes;
}

static int bounce_ensure(sct_ctx *c) {
    for (int i = 0; i < 2; i++) {
        if (!c->bounce[i]) {
            hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipHostMalloc(&c->bounce[i], BOUNCE_BYTES, hipHostMallocDefault);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                c->bounce[i] = nullptr;
                return fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "hipHostMalloc(%zu): %s", BOUNCE_BYTES, hipGetErrorString(e));
            }
        }
        if (!c->bounce_ev[i]) HIPCHK(hipEventCreateWithFlags(&c->bounce_ev[i], hipEventDisableTiming));
    }
    return SCT_OK;
}

// host -> device, queued on st.  Page-locked `src`: queued only (src must stay until st is synchronised).  Other memory: on return
// src has been read (it may be freed or changed).
static int copy_up(sct_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (!bytes) return SCT_OK;
    if (is_page_locked(src)) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return SCT_OK;
    }
    if (bytes <= BOUNCE_DIRECT_MAX) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return SCT_OK;
    }
    if (int rc = bounce_ensure(c)) return rc;
    bool used[2] = {false, false};
    for (size_t at = 0, i = 0; at < bytes; at += BOUNCE_BYTES, i ^= 1) {
        const size_t m = bytes - at < BOUNCE_BYTES ? bytes - at : BOUNCE_BYTES;
        if (used[i]) HIPCHK(hipEventSynchronize(c->bounce_ev[i]));
        par_copy(c, c->bounce[i], (const char *)src + at, m);
        HIPCHK(hipMemcpyAsync((char *)dst + at, c->bounce[i], m, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(c->bounce_ev[i], st));
        used[i] = true;
    }
    for (int i = 0; i < 2; i++)
        if (used[i]) HIPCHK(hipEventSynchronize(c->bounce_ev[i]));   // (the pieces are the context's: the next copy_up may fill them at once)
    return SCT_OK;
}

// device -> host, behind everything queued on st.  Page-locked `dst`: queued only.  Other memory: on return dst holds the bytes.
static int copy_down(sct_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (!bytes) return SCT_OK;
    if (is_page_locked(dst)) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        return SCT_OK;
    }
    if (bytes <= BOUNCE_DIRECT_MAX) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return SCT_OK;
    }
    if (int rc = bounce_ensure(c)) return rc;
    size_t queued_at[2] = {0, 0}, queued_m[2] = {0, 0};
    bool used[2] = {false, false};
    auto drain = [&](int i) -> int {
        HIPCHK(hipEventSynchronize(c->bounce_ev[i]));
        par_copy(c, (char *)dst + queued_at[i], c->bounce[i], queued_m[i]);
        used[i] = false;
        return SCT_OK;
    };
    for (size_t at = 0, i = 0; at < bytes; at += BOUNCE_BYTES, i ^= 1) {
        const size_t m = bytes - at < BOUNCE_BYTES ? bytes - at : BOUNCE_BYTES;
        if (used[i])
            if (int rc = drain((int)i)) return rc;
        HIPCHK(hipMemcpyAsync(c->bounce[i], (const char *)src + at, m, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(c->bounce_ev[i], st));
        queued_at[i] = at;
        queued_m[i] = m;
        used[i] = true;
    }
    // (the older piece first: they were queued in that order)
    const int first = (used[0] && used[1] && queued_at[1] < queued_at[0]) ? 1 : 0;
    for (int k = 0; k < 2; k++) {
        const int i = k == 0 ? first : 1 - first;
        if (used[i])
            if (int rc = drain(i)) return rc;
    }
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
    FaultScope fault_scope(&c->fault);
    if (n < 0 || (n > 0 && !bc)) return fail(SCT_E_ARG, "sct_set_barcodes: bad barcode buffer");
    if (bl < 1) return fail(SCT_E_ARG, "barcode length must be positive (got %d)", bl);
    if (bl > SCT_MAX_BL) return fail(SCT_E_UNSUPPORTED, "barcode length %d exceeds the supported maximum %d", bl, SCT_MAX_BL);
    if ((int64_t)n * 2 > 0x7FFFFFFF) return fail(SCT_E_ARG, "too many barcodes");
    const uint8_t *b = (const uint8_t *)bc;
    for (int64_t i = 0; i < (int64_t)n * bl; i++)
        if (b[i] >= 128) return fail(SCT_E_ARG, "barcode %lld contains byte 0x%02x >= 128", (long long)(i / bl), b[i]);
    HIPCHK(hipSetDevice(c->device));
    // the tables are shared by every lane: nothing may still be reading the old ones
    for (Lane *l : c->lanes) HIPCHK(hipStreamSynchronize(l->stream));
    hipStream_t st = c->lane0().stream;
    c->have_bc = false;
    c->have_tables = false;
    c->have_thirds = false;
    c->have_wide = false;
    c->have_fb = false;
    c->have_wfb = false;
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
    if (c->pat.ensure(sizeof(uint2) * (size_t)(np ? np : 1))) return g_code;
    // every copy below reads a host vector of this scope: each path out of the function synchronises the stream first
    if (np)
        if (int rc = copy_up(c, c->pat.p, pat.data(), sizeof(uint2) * (size_t)np, st)) return rc;
    if (bl >= 8 && bl <= 16) {
        // seed tables (sct_tables.h): counting sort of the pattern ids by key, per family, plus the bucket slots.
        // The thirds families (20/22-bit keys) are only worth their 38 MB of offsets for very large barcode sets.
        // Likewise the wide families of max_err = 2 (sixteen tables of np entries and 4 MB of offsets each).
        const bool thirds = bl == 16 && np > 0 && (c->thirds_mode < 0 ? np >= THIRDS_MIN_PATTERNS : c->thirds_mode == 1);
        const bool wide = bl == 16 && np > 0 && (c->wide_mode < 0 ? np >= WIDE_MIN_PATTERNS : c->wide_mode == 1);
        SctSeedTables tb;
        std::vector<uint16_t> pid16;
        int rc = SCT_OK;
        do {
            if (!sct_build_tables(pat.data(), np, bl, thirds, wide, tb)) {
                rc = fail(SCT_E_UNSUPPORTED, "too many barcodes for the seed tables");
                break;
            }
            FamTables ft{};
            for (int f = 0; f < SCT_NFAM_TABLES; f++) ft.off_base[f] = tb.off_base[f];
            for (int f = 0; f < 3; f++) ft.fb_base[f] = tb.fb_base[f];
            c->long_share = tb.long_share;
            c->max_bucket = tb.max_bucket;
            if (tb.slot.size() != SLOT_TOTAL) {
                rc = fail(SCT_E_STATE, "seed tables: unexpected slot count");
                break;
            }
            ft.pid16 = np <= 65536 ? 1u : 0u;
            if (ft.pid16) pid16.assign(tb.ent.begin(), tb.ent.end());
            if (c->tab_slot.ensure(tb.slot.size() * sizeof(uint64_t)) || c->tab_grp.ensure(tb.grp.size() * sizeof(uint32_t)) ||
                c->tab_ent16.ensure(pid16.size() * sizeof(uint16_t) + 16) || c->tab_off.ensure(tb.off.size() * sizeof(uint32_t) + 16) ||
                c->tab_ent.ensure(tb.ent.size() * sizeof(uint32_t)) || c->tab_code.ensure(tb.code.size() * sizeof(uint32_t)) ||
                c->tab_fb.ensure(tb.fb.size() * sizeof(uint16_t) + 16) || c->tab_wfb.ensure(tb.wfb.size() * sizeof(uint16_t) + 16)) {
                rc = g_code;
                break;
            }
            struct Up {
                void *dst;
                const void *src;
                size_t bytes;
            } ups[] = {{c->tab_slot.p, tb.slot.data(), tb.slot.size() * sizeof(uint64_t)},
                       {c->tab_grp.p, tb.grp.data(), tb.grp.size() * sizeof(uint32_t)},
                       {c->tab_ent16.p, pid16.data(), ft.pid16 ? pid16.size() * sizeof(uint16_t) : 0},
                       {c->tab_code.p, tb.code.data(), tb.code.size() * sizeof(uint32_t)},
                       {c->tab_off.p, tb.off.data(), tb.off.size() * sizeof(uint32_t)},
                       {c->tab_ent.p, tb.ent.data(), tb.ent.size() * sizeof(uint32_t)},
                       {c->tab_fb.p, tb.fb.data(), tb.fb.size() * sizeof(uint16_t)},
                       {c->tab_wfb.p, tb.wfb.data(), tb.wfb.size() * sizeof(uint16_t)}};
            for (const Up &u : ups)
                if (rc == SCT_OK && u.bytes) rc = copy_up(c, u.dst, u.src, u.bytes, st);
            const hipError_t es = hipStreamSynchronize(st);
            if (rc == SCT_OK && es != hipSuccess) rc = fail(SCT_E_HIP, "uploading the seed tables: %s", hipGetErrorString(es));
            if (rc) break;
            c->ft = ft;
            c->have_tables = true;
            c->have_thirds = thirds;
            c->have_wide = wide;
            c->have_fb = thirds && !tb.fb.empty();
            c->have_wfb = wide && !tb.wfb.empty();
            c->tables_nfam = tb.nfam;
        } while (0);
        if (rc) {
            (void)hipStreamSynchronize(st);
            return rc;
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    c->n_bc = n;
    c->bl = bl;
    c->np = np;
    c->have_bc = true;
    for (Lane *l : c->lanes) l->have_result = false;
    return SCT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// one batch of rows on one lane
// ------------------------------------------------------------------------------------------------------------
static int batch_params(sct_ctx *c, int32_t max_err, int32_t threads, int count_format, BatchParams &bp) {
    if (!c->have_bc) return fail(SCT_E_STATE, "matching before sct_set_barcodes");
    if (max_err < 0 || max_err >= c->bl) return fail(SCT_E_ARG, "need 0 <= max_err < barcode length (got %d, bl %d)", max_err, c->bl);
    if (threads < 1) return fail(SCT_E_ARG, "threads must be >= 1");
    bp.max_err = max_err;
    bp.threads = threads;
    bp.prefix_len = sct_prefix_len(threads);
    // With a prefix as long as the shortest window (k >= bl - max_err) the reference's own sentinel insert
    // (`trie.insert(prefix, -1)`, scTagger.py:675, :563) reaches the result dict as rid -1; that needs -t > 4^(bl - max_err - 1)
    //  and is not reproduced.
    if (bp.prefix_len >= c->bl - max_err)
        return fail(SCT_E_UNSUPPORTED, "threads = %d gives the reference a window prefix of %d characters, not below bl - max_err = %d: "
                    "its sentinel entry would enter the results (scTagger.py:675); not supported", threads, bp.prefix_len, c->bl - max_err);
    bp.filter_ok = c->have_tables && max_err <= 2 && !c->force_exhaustive && c->np > 0 && (uint64_t)c->tables_nfam * (uint64_t)c->np < LOC_INLINE;
    bp.rt_families = c->bl != 16;  // quarter families with run-time piece boundaries
    bp.use_thirds = bp.filter_ok && c->have_thirds && max_err <= 1 && !c->no_thirds;
    bp.use_wide = bp.filter_ok && c->have_wide && c->have_wfb && max_err == 2 && c->wide_mode != 0;
    bp.use_slots = c->bl == 16 && !bp.use_thirds && !bp.use_wide &&
                   (c->slots_mode < 0 ? c->long_share <= (max_err <= 1 ? SLOTS_MAX_LONG_SHARE_E1 : SLOTS_MAX_LONG_SHARE_E2) && c->max_bucket <= SLOTS_MAX_BUCKET
                                       : c->slots_mode == 1);
    // the join reads the fixed buckets and packs a row number into 30 bits of a tuple
    bp.use_join = bp.use_thirds && max_err == 1 && c->join_mode == 1 && c->have_fb && c->fb_mode != 0;
    bp.count_format = count_format;
    c->last_join = 0;
    c->last_scheme = !bp.filter_ok ? SCT_SCHEME_NONE : bp.use_thirds ? SCT_SCHEME_THIRDS : bp.use_wide ? SCT_SCHEME_WIDE
                     : (bp.use_slots && !bp.rt_families) ? SCT_SCHEME_QUARTER_SLOTS : SCT_SCHEME_QUARTER_TABLES;
    return SCT_OK;
}

// buffers of a batch of n rows
static int lane_ensure(Lane &l, int64_t n, const BatchParams &bp) {
    const size_t n1 = (size_t)(n ? n : 1);
    if (l.min_e.ensure(n1) || l.cnt.ensure(4 * n1) || l.list_long.ensure(4 * n1) || l.list_scan.ensure(4 * n1) || l.list_dp.ensure(4 * n1) ||
        l.counters.ensure(sizeof(uint32_t) * CTR_WORDS) || l.hit_offs.ensure(8 * (n1 + 1)))
        return g_code;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (l.sums.ensure(8 * (size_t)(nb ? nb : 1))) return g_code;
    if (bp.count_format == SCT_CNT_U8 && l.cnt_out.ensure(n1)) return g_code;
    if (bp.count_format == SCT_CNT_U32 && l.cnt_out.ensure(4 * n1)) return g_code;
    l.slots_cap = l.retry_cap = 0;
    if (bp.filter_ok) {
        // the main launch keeps at most K_SLOTS hits per row, the retry launch shares a pool of the same size; the CSR hit
        // array can therefore hold whatever the two launches produce (rows of the scan lists are sized by the slow path)
        // (rows rounded up to whole rounds of SLOT_ARENAS blocks of 256: every arena then has room for K_SLOTS entries of each of
        // the rows of the blocks that use it, whatever the number of blocks — no wave is deferred for lack of room in its arena)
        const uint64_t arena_round = (uint64_t)SLOT_ARENAS * 256;
        l.slots_cap = ((uint64_t)n1 + arena_round - 1) / arena_round * arena_round * K_SLOTS;
        l.retry_cap = (uint64_t)n1 * K_SLOTS < 65536 ? 65536 : (uint64_t)n1 * K_SLOTS;
        if (l.slots.ensure(4 * l.slots_cap) || l.slots_retry.ensure(4 * l.retry_cap) || l.slot_off.ensure(8 * n1) || l.wave_base.ensure(8 * (n1 / 64 + 2)) ||
            l.list_retry.ensure(4 * n1) || l.list_special.ensure(4 * n1) || l.list_special2.ensure(4 * n1))
            return g_code;
        // planes of the rows listed for the special launch: room for an eighth of the rows (the others are decoded from their bytes)
        const uint64_t rc = n1 / 8 < 65536 ? 65536 : n1 / 8;
        l.rec_cap = (uint32_t)(rc < 0x7FFFFFFFull ? rc : 0x7FFFFFFFull);
        if (l.rec_special.ensure(sizeof(SctSpecialRec) * (size_t)l.rec_cap) || l.rec_special2.ensure(sizeof(SctSpecialRec) * (size_t)l.rec_cap)) return g_code;
        // the long rows of a zero-copy batch, gathered before the special launch reads them: 16 bytes per row of the batch (a tenth
        // of the rows at 150 characters), between 4 MB and 256 MB; rows that find no room are read where the caller has them
        const size_t gc = 16 * n1 < ((size_t)4 << 20) ? ((size_t)4 << 20) : (16 * n1 < ((size_t)256 << 20) ? 16 * n1 : ((size_t)256 << 20));
        if (l.gather.ensure(gc)) return g_code;
    }
    if (l.hits.ensure(4 * (size_t)(l.slots_cap + l.retry_cap + 1))) return g_code;
    if (bp.use_join && n < ((int64_t)1 << 30)) {
        // per row: planes 8 + length 1 + hit count 4 + JOIN_HITS entries 32 bytes; per pass of JOIN_SUB_ROWS rows: the tuple regions
        const int64_t sub = n < JOIN_SUB_ROWS ? n : JOIN_SUB_ROWS;
        const size_t prod = (size_t)((sub + 255) / 256) + 1;
        if (l.j_planes.ensure(8 * n1) || l.j_len.ensure(n1) || l.j_wave.ensure(n1 / 64 + 2) || l.j_cnt.ensure(4 * n1) ||
            l.j_slots.ensure(4 * (size_t)JOIN_HITS * n1 + 64) || l.j_tup0.ensure(8 * prod * JOIN_P0 * JOIN_CAP0) || l.j_cnt0.ensure(4 * prod * JOIN_P0) ||
            l.j_tup1.ensure(8 * prod * JOIN_P1 * JOIN_CAP1) || l.j_cnt1.ensure(4 * prod * JOIN_P1))
            return g_code;
    }
    l.min_e_ptr = l.min_e.as<uint8_t>();
    l.cnt8_ptr = bp.count_format == SCT_CNT_U8 ? l.cnt_out.as<uint8_t>() : nullptr;
    l.cnt32_ptr = bp.count_format == SCT_CNT_U32 ? l.cnt_out.as<uint32_t>() : nullptr;
    l.hits_ptr = l.hits.as<uint32_t>();
    l.hits_cap = l.hits.cap / 4;
    l.hits16_ptr = nullptr;
    l.direct_min_e = l.direct_cnt = l.direct_hits = false;
    l.cnt8_soft = false;
    l.record_filters_done = false;
    l.wait_for = nullptr;
    return SCT_OK;
}

template <typename OffT>
static void launch_filter(int max_err, SctFilterVariant v, const FilterArgs<OffT> &fa, unsigned g, unsigned g64, hipStream_t st) {
    if (v.special) {
        if (max_err == 0) sct_launch_filter_sp_e0(v, fa, g, g64, st);
        else if (max_err == 1) sct_launch_filter_sp_e1(v, fa, g, g64, st);
        else sct_launch_filter_sp_e2(v, fa, g, g64, st);
        return;
    }
    if (max_err == 0) sct_launch_filter_e0(v, fa, g, g64, st);
    else if (max_err == 1) sct_launch_filter_e1(v, fa, g, g64, st);
    else sct_launch_filter_e2(v, fa, g, g64, st);
}

// grids of the striding launches (k_filter): enough blocks to fill the chip several times over, few enough to cost nothing
// when their waves find no work
static constexpr unsigned GRID_STRIDE_MAX = 8192;  // one-wave blocks
static inline unsigned capped(int64_t g) { return (unsigned)(g < 1 ? 1 : (g < GRID_STRIDE_MAX ? g : GRID_STRIDE_MAX)); }

template <typename OffT>
static void queue_filter_launches(sct_ctx *c, Lane &l, const OffT *offs) {
    const BatchParams &bp = l.bp;
    const int64_t n = l.n_seg;
    hipStream_t st = l.stream;
    uint32_t *ctr = l.ctr();
    FilterArgs<OffT> fa{};
    fa.segs = l.segs_ptr;
    fa.offs = offs;
    fa.seg_bytes = l.seg_bytes;
    fa.tab_off = c->tab_off.as<uint32_t>();
    fa.tab_ent = c->tab_ent.as<uint32_t>();
    fa.tab_code = c->tab_code.as<uint32_t>();
    fa.tab_slot = c->tab_slot.as<uint2>();
    fa.tab_grp = c->tab_grp.as<uint32_t>();
    fa.tab_ent16 = c->tab_ent16.as<uint16_t>();
    fa.tab_fb = (c->have_fb && c->fb_mode != 0) ? c->tab_fb.as<uint16_t>() : nullptr;
    fa.tab_wfb = c->have_wfb ? c->tab_wfb.as<uint16_t>() : nullptr;
    fa.ft = c->ft;
    fa.min_e = l.min_e_ptr;
    fa.cnt8 = l.cnt8_ptr;  // the filter launches write the one-byte counts themselves (queue_batch leaves them out of k_offsets)
    fa.cnt = l.cnt.as<uint32_t>();
    fa.slot_off = l.slot_off.as<uint64_t>();
    fa.wave_base = l.wave_base.as<uint64_t>();
    fa.list_special = l.list_special.as<int32_t>();
    fa.rec_out = l.rec_special.as<SctSpecialRec>();
    fa.rec_in = nullptr;
    fa.rec_cap = l.rec_cap;
    const bool gather = l.host_segs != nullptr && l.gather.p != nullptr && !c->no_gather;   // a zero-copy batch (see k_special_gather)
    fa.gather = nullptr;
    fa.gather_out = gather ? l.gather.as<uint8_t>() : nullptr;
    fa.gather_cap = (uint64_t)l.gather.cap;
    fa.ctr = ctr;
    fa.bl = c->bl;
    fa.prefix_len = bp.prefix_len;
    SctFilterVariant v{false, bp.rt_families, bp.use_slots && !bp.rt_families, bp.use_thirds, bp.use_wide, false, false, false};
    // (see sct_ctx::need64; the partitioned join launches the 64-bit instantiation alone and marks nothing)
    const bool join_route = bp.use_join && n < ((int64_t)1 << 30);
    l.no64 = !c->need64 && c->need64_mode < 0 && !join_route;
    l.no_special = !c->need_special && c->need_special_mode < 0 && !join_route;
    const auto g64_of = [&](int64_t g) { return l.no64 ? 0u : capped(g); };
    {
        Timer t(c, l, "k_filter", l.index_just_timed);
        l.index_just_timed = false;
        fa.n = n;
        fa.n_dev = nullptr;
        fa.rows = nullptr;
        fa.slots = l.slots.as<uint32_t>();
        fa.slots_cap = l.slots_cap;
        fa.list_defer = l.list_retry.as<int32_t>();
        fa.defer_count = ctr + CTR_RETRY;
        fa.slot_cursor = reinterpret_cast<unsigned long long *>(ctr + CTR_CURSOR);
        // large batches: the slot array in SLOT_ARENAS arenas with a cursor each (a block's arena = its index mod SLOT_ARENAS, so an
        // arena sees 1 / SLOT_ARENAS of the rows and holds K_SLOTS entries for each of them: it cannot overflow either)
        fa.arena_cursors = (n >= 65536 && l.arena_cur.p) ? l.arena_cur.as<unsigned long long>() : nullptr;
        fa.arena_cap = l.slots_cap / SLOT_ARENAS;
        const bool join_ready = l.j_planes.p && l.j_len.p && l.j_wave.p && l.j_cnt.p && l.j_slots.p && l.j_tup0.p && l.j_cnt0.p && l.j_tup1.p && l.j_cnt1.p &&
                                l.j_planes.cap >= 8 * (size_t)n && fa.tab_fb != nullptr;
        if (bp.use_join && n < ((int64_t)1 << 30) && join_ready) {
            // the 32-bit rows through the partitioned join, pass after pass of JOIN_SUB_ROWS rows (the tuple regions are sized for
            // one pass), then its finish over all rows; k_filter's 64-bit instantiation takes the waves of 32 .. 63-character rows
            JoinArgs<OffT> ja{};
            ja.fa = fa;
            ja.planes = l.j_planes.as<uint2>();
            ja.jlen = l.j_len.as<uint8_t>();
            ja.wave_kind = l.j_wave.as<uint8_t>();
            ja.jcnt = l.j_cnt.as<uint32_t>();
            ja.jslots = l.j_slots.as<uint32_t>();
            ja.tup0 = l.j_tup0.as<unsigned long long>();
            ja.cnt0 = l.j_cnt0.as<uint32_t>();
            ja.tup1 = l.j_tup1.as<unsigned long long>();
            ja.cnt1 = l.j_cnt1.as<uint32_t>();
            for (int64_t r0 = 0; r0 < n; r0 += JOIN_SUB_ROWS) {
                ja.row0 = r0;
                ja.row1 = r0 + JOIN_SUB_ROWS < n ? r0 + JOIN_SUB_ROWS : n;
                sct_launch_join_pass(ja, st);
            }
            sct_launch_join_finish(ja, st);
            SctFilterVariant v64 = v;
            v64.only64 = true;
            launch_filter(bp.max_err, v64, fa, 1, capped((n + 63) / 64), st);
        } else {
            fa.skip64_unless_needed = true;     // (both instantiations over all rows, the 32-bit one first)
            launch_filter(bp.max_err, v, fa, blocks_for(n, 256), g64_of((n + 63) / 64), st);
            fa.skip64_unless_needed = false;
        }
    }
    {
        // rows with more ties / verified matches than the main launch keeps: once more with larger lists.  How many
        // there are is only known on the device: a fixed grid strides over them.  Rows this launch cannot hold either go
        // to the exhaustive scan (list_scan).
        Timer t(c, l, "k_filter_retry", true);
        v.retry = true;
        fa.rec_out = nullptr;
        fa.gather_out = nullptr;
        fa.n = 0;
        fa.n_dev = ctr + CTR_RETRY;
        fa.rows = l.list_retry.as<int32_t>();
        fa.slots = l.slots_retry.as<uint32_t>();
        fa.slots_cap = l.retry_cap;
        fa.list_defer = l.list_scan.as<int32_t>();
        fa.defer_count = ctr + CTR_SCAN;
        fa.slot_cursor = reinterpret_cast<unsigned long long *>(ctr + CTR_CURSOR_RETRY);
        fa.arena_cursors = nullptr;
        launch_filter(bp.max_err, v, fa, capped((n + 3) / 4), g64_of((n + 3) / 4), st);  // one wave of 4 rows per block, striding
    }
    if (!l.no_special) {
        // rows the main launch listed because one lane's planes cannot express them as the reference sees them (restricted window
        // starts; more than PLANE_MAX characters): the retry launch's shape with start masks and windows (filter_rows, SP).  Its lists
        // go to the retry launch's slot array; what it cannot hold goes on to list_long -> the scan kernels of the slow path.
        // With bucket slots (the BASELINE shapes) in two stages: 16 rows per wave first — up to 4 windows and 80 list entries per row —,
        // then 4 rows per wave for what that hands on.
        Timer t(c, l, "k_filter_special", true);
        v.retry = false;
        v.special = true;
        fa.n = 0;
        fa.slots = l.slots_retry.as<uint32_t>();
        fa.slots_cap = l.retry_cap;
        fa.slot_cursor = reinterpret_cast<unsigned long long *>(ctr + CTR_CURSOR_RETRY);
        fa.arena_cursors = nullptr;
        fa.n_dev = ctr + CTR_SPECIAL;
        fa.rows = l.list_special.as<int32_t>();
        fa.rec_in = l.rec_special.as<SctSpecialRec>();
        fa.rec_out = nullptr;
        if (gather) {
            // a zero-copy batch: every window lane of a long row would read its words over the link.  The main launch copied the rows
            // it had in LDS; one wave per other listed row copies that row to the device first (striding, as the launch itself: how
            // many rows there are is only known on the device).
            k_special_gather<OffT><<<capped((n + 63) / 64), 256, 0, st>>>(fa.segs, offs, l.seg_bytes, fa.rows, fa.n_dev, l.rec_special.as<SctSpecialRec>(), l.rec_cap,
                                                                      l.gather.as<uint8_t>(), (uint64_t)l.gather.cap, ctr);
            fa.gather = l.gather.as<uint8_t>();
        }
        if (v.slots) {
            v.special16 = true;
            fa.list_defer = l.list_special2.as<int32_t>();
            fa.defer_count = ctr + CTR_SPECIAL2;
            fa.rec_out = l.rec_special2.as<SctSpecialRec>();
            launch_filter(bp.max_err, v, fa, capped((n + 15) / 16), g64_of((n + 15) / 16), st);
            v.special16 = false;
            fa.n_dev = ctr + CTR_SPECIAL2;
            fa.rows = l.list_special2.as<int32_t>();
            fa.rec_in = l.rec_special2.as<SctSpecialRec>();
            fa.rec_out = nullptr;
        }
        fa.list_defer = l.list_long.as<int32_t>();
        fa.defer_count = ctr + CTR_LONG;
        launch_filter(bp.max_err, v, fa, capped((n + 3) / 4), g64_of((n + 3) / 4), st);
    }
}

// cnt -> hit_offs (+ the compact counts) -> total, on the lane's stream
static void queue_offsets(sct_ctx *c, Lane &l, bool counts_done = false) {
    const int64_t n = l.n_seg;
    hipStream_t st = l.stream;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    uint64_t *total = reinterpret_cast<uint64_t *>(l.ctr() + CTR_TOTAL);
    if (n > 0) {
        Timer t(c, l, "k_offsets");
        k_sum_blocks<uint32_t><<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), nullptr);
        k_scan_sums<<<1, 256, 0, st>>>(l.sums.as<uint64_t>(), nb, total);
        k_offsets<<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), total, l.hit_offs.as<int64_t>(),
                                                counts_done ? nullptr : l.cnt8_ptr, l.cnt32_ptr, l.ctr());
    } else {
        (void)hipMemsetAsync(l.hit_offs.p, 0, 8, st);
    }
}

// Queue everything a batch needs on the lane's stream — no host synchronisation.  The rows (segs + offsets, l.n_seg,
// l.seg_bytes, l.wide) are already on the device or their copies are queued on the same stream.  The counter record must
// have been reset (reset_counters) before whatever checks the input wrote to it.
// counters: all zero, but "smallest bad row / byte" = ~0 (CTR_BAD_ROW, CTR_BAD_BYTE: words 8..11); slot-arena cursors: zero
__global__ void __launch_bounds__(256) k_reset_counters(uint32_t *__restrict__ ctr, uint32_t *__restrict__ arena, int arena_words) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < CTR_WORDS) ctr[i] = (i >= CTR_BAD_ROW && i < CTR_BAD_ROW + 4) ? 0xFFFFFFFFu : 0u;
    if (i < arena_words) arena[i] = 0u;
}

static int reset_counters(Lane &l) {
    if (l.counters.ensure(sizeof(uint32_t) * CTR_WORDS) || l.arena_cur.ensure(64 * SLOT_ARENAS)) return g_code;
    // one launch at the head of the batch instead of three memsets (each a launch of its own in front of the first kernel)
    k_reset_counters<<<(64 * SLOT_ARENAS / 4 + 255) / 256, 256, 0, l.stream>>>(l.ctr(), l.arena_cur.as<uint32_t>(), (int)(64 * SLOT_ARENAS / 4));
    HIPCHK(hipGetLastError());
    return SCT_OK;
}

static int queue_batch(sct_ctx *c, Lane &l) {
    const BatchParams &bp = l.bp;
    const int64_t n = l.n_seg;
    hipStream_t st = l.stream;
    l.have_result = false;
    if (l.wait_for) {   // option "chunk_order": this chunk's filter launches behind the previous chunk's (its fill then runs beside them)
        HIPCHK(hipStreamWaitEvent(st, l.wait_for, 0));
        l.wait_for = nullptr;
    }
    if (n > 0) {
        if (bp.filter_ok) {
            if (l.wide) queue_filter_launches<int64_t>(c, l, l.offs64.as<int64_t>());
            else queue_filter_launches<uint32_t>(c, l, l.offs32.as<uint32_t>());
            if (l.record_filters_done) HIPCHK(hipEventRecord(l.filters_done, st));
        } else {
            l.no64 = l.no_special = false;
            Timer t(c, l, "k_route");
            if (l.wide)
                k_route_all<int64_t><<<blocks_for(n, 256), 256, 0, st>>>(l.segs_ptr, l.offs64.as<int64_t>(), l.seg_bytes, n, c->bl,
                                                                        bp.max_err, bp.prefix_len, l.min_e_ptr, l.cnt.as<uint32_t>(),
                                                                        l.list_scan.as<int32_t>(), l.list_dp.as<int32_t>(), l.ctr());
            else
                k_route_all<uint32_t><<<blocks_for(n, 256), 256, 0, st>>>(l.segs_ptr, l.offs32.as<uint32_t>(), l.seg_bytes, n, c->bl,
                                                                         bp.max_err, bp.prefix_len, l.min_e_ptr, l.cnt.as<uint32_t>(),
                                                                         l.list_scan.as<int32_t>(), l.list_dp.as<int32_t>(), l.ctr());
        }
    }
    if (n > 0 && bp.filter_ok) {
        // hit counts -> offsets -> hit array: block sums and their scan (span "k_offsets"), then offsets and lists in ONE kernel
        // (span "k_fill"; k_offsets_fill).  The filter launches wrote the one-byte counts themselves.
        const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
        uint64_t *total = reinterpret_cast<uint64_t *>(l.ctr() + CTR_TOTAL);
        {
            Timer t(c, l, "k_offsets", true);    // (right behind the filter launches)
            k_sum_blocks<uint32_t><<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), nullptr);
            k_scan_sums<<<1, 256, 0, st>>>(l.sums.as<uint64_t>(), nb, total);
        }
        Timer t(c, l, "k_fill", true);
        if (l.hits16_ptr)
            k_offsets_fill<uint16_t><<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), total, l.hit_offs.as<int64_t>(), nullptr,
                                                                  l.cnt32_ptr, l.ctr(), l.slots.as<uint32_t>(), l.slots_retry.as<uint32_t>(),
                                                                  l.wave_base.as<uint64_t>(), l.slot_off.as<uint64_t>(), l.hits16_ptr, l.hits_cap);
        else
            k_offsets_fill<uint32_t><<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), total, l.hit_offs.as<int64_t>(), nullptr,
                                                                  l.cnt32_ptr, l.ctr(), l.slots.as<uint32_t>(), l.slots_retry.as<uint32_t>(),
                                                                  l.wave_base.as<uint64_t>(), l.slot_off.as<uint64_t>(), l.hits_ptr, l.hits_cap);
    } else {
        queue_offsets(c, l);
    }
    HIPCHK(hipMemcpyAsync(l.h_ctr, l.counters.p, sizeof(uint32_t) * CTR_WORDS, hipMemcpyDeviceToHost, st));
    HIPCHK(hipGetLastError());
    return SCT_OK;
}

static int ensure_host_hits(sct_ctx *c, size_t entries);
static uint32_t *host_hits_device_view(sct_ctx *c);
static inline uint64_t ctr64(const uint32_t *h, int at) { return (uint64_t)h[at] | ((uint64_t)h[at + 1] << 32); }

// The rows the filter kernels left on the lists: scan kernels, then offsets and fill once more.  Synchronises the lane.
static int slow_path(sct_ctx *c, Lane &l) {
    const BatchParams &bp = l.bp;
    const int64_t n = l.n_seg;
    const int np = c->np, bl = c->bl;
    hipStream_t st = l.stream;
    uint32_t *ctr = l.ctr();
    uint32_t *h = l.h_ctr;
    if (l.hits16_ptr) {  // the scan kernels write 32-bit hits: back to the lane's own array (narrowed afterwards)
        l.hits16_ptr = nullptr;
        l.hits_ptr = l.hits.as<uint32_t>();
        l.hits_cap = l.hits.cap / 4;
    }
    if (l.host_segs) {
        // a zero-copy batch: the scan kernels read every listed row once per barcode slice — bring the bytes to the device first
        if (l.segs.ensure((size_t)l.seg_bytes + 64)) return g_code;
        if (l.seg_bytes) HIPCHK(hipMemcpyAsync(l.segs.p, l.host_segs, (size_t)l.seg_bytes, hipMemcpyHostToDevice, st));
        l.segs_ptr = l.segs.as<uint8_t>();
        l.host_segs = nullptr;
    }
    // the scan kernels take int64 offsets
    if (!l.have_offs64) {
        if (l.offs64.ensure(8 * (size_t)(n + 1))) return g_code;
        k_widen_offs<<<blocks_for(n + 1, 256), 256, 0, st>>>(l.offs32.as<uint32_t>(), n + 1, l.offs64.as<int64_t>());
        l.have_offs64 = true;
    }
    const int64_t *offs = l.offs64.as<int64_t>();
    if (h[CTR_LONG]) {  // rows the filter only listed: exhaustive scan or DP?
        k_route_long<<<blocks_for((int64_t)h[CTR_LONG], 4), 256, 0, st>>>(l.segs_ptr, offs, l.seg_bytes, l.list_long.as<int32_t>(),
                                                                         (int64_t)h[CTR_LONG], bl, bp.max_err, bp.prefix_len, l.list_scan.as<int32_t>(),
                                                                         l.list_dp.as<int32_t>(), ctr);
        HIPCHK(hipMemcpyAsync(h, l.counters.p, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    const uint32_t n_scan = h[CTR_SCAN], n_dp = h[CTR_DP];
    // remember which rows are deferred: their cnt is about to be overwritten by the count passes (k_fill skips them)
    if (l.route.ensure(4 * (size_t)(n ? n : 1))) return g_code;
    HIPCHK(hipMemcpyAsync(l.route.p, l.cnt.p, 4 * (size_t)n, hipMemcpyDeviceToDevice, st));
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
    const int sl_scan = slices_for(n_scan), sl_dp = slices_for(n_dp);
    if (l.part_best.ensure(4 * ((size_t)n_scan * sl_scan + (size_t)n_dp * sl_dp + 1)) ||
        l.part_cnt.ensure(4 * ((size_t)n_scan * sl_scan + (size_t)n_dp * sl_dp + 1)))
        return g_code;
    uint32_t *pb_scan = l.part_best.as<uint32_t>(), *pc_scan = l.part_cnt.as<uint32_t>();
    uint32_t *pb_dp = pb_scan + (size_t)n_scan * sl_scan, *pc_dp = pc_scan + (size_t)n_scan * sl_scan;
    const uint8_t *segs = l.segs_ptr;
    if (n_scan) {
        Timer t(c, l, "k_scan_count");
        if (bl <= 16)
            k_scan_pk<false><<<grid_for(n_scan, sl_scan), 256, 0, st>>>(segs, offs, l.seg_bytes, l.list_scan.as<int32_t>(), (int64_t)n_scan, sl_scan,
                                                                       c->pat.as<uint2>(), np, bl, bp.max_err, l.min_e_ptr, pb_scan,
                                                                       pc_scan, nullptr, nullptr);
        else
            k_scan<false, false><<<grid_for(n_scan, sl_scan), 256, 0, st>>>(segs, offs, l.seg_bytes, l.list_scan.as<int32_t>(), (int64_t)n_scan,
                                                                           sl_scan, c->pat.as<uint2>(), np, bl, bp.max_err, bp.prefix_len,
                                                                           l.min_e_ptr, pb_scan, pc_scan, nullptr, nullptr);
        k_scan_reduce<<<blocks_for(n_scan, 256), 256, 0, st>>>(l.list_scan.as<int32_t>(), (int64_t)n_scan, sl_scan, bp.max_err, pb_scan, pc_scan,
                                                              l.min_e_ptr, l.cnt.as<uint32_t>());
    }
    if (n_dp) {
        Timer t(c, l, "k_dp_count");
        k_scan<true, false><<<grid_for(n_dp, sl_dp), 256, 0, st>>>(segs, offs, l.seg_bytes, l.list_dp.as<int32_t>(), (int64_t)n_dp, sl_dp,
                                                                  c->pat.as<uint2>(), np, bl, bp.max_err, bp.prefix_len, l.min_e_ptr,
                                                                  pb_dp, pc_dp, nullptr, nullptr);
        k_scan_reduce<<<blocks_for(n_dp, 256), 256, 0, st>>>(l.list_dp.as<int32_t>(), (int64_t)n_dp, sl_dp, bp.max_err, pb_dp, pc_dp,
                                                            l.min_e_ptr, l.cnt.as<uint32_t>());
    }
    HIPCHK(hipMemsetAsync(ctr + CTR_CNT_OVERFLOW, 0, 4, st));
    queue_offsets(c, l);
    HIPCHK(hipMemcpyAsync(h, l.counters.p, sizeof(uint32_t) * CTR_WORDS, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t total = ctr64(h, CTR_TOTAL);
    if (l.hits_cap < total) {
        if (l.direct_hits) {
            if (int rc = ensure_host_hits(c, (size_t)total + 1)) return rc;
            l.hits_ptr = host_hits_device_view(c);
            if (!l.hits_ptr) return fail(SCT_E_STATE, "the page-locked result buffer is not mapped into the device");
            l.hits_cap = c->h_hits_cap;
        } else {
            if (l.hits.ensure(4 * (size_t)(total + 1))) return g_code;
            l.hits_ptr = l.hits.as<uint32_t>();
            l.hits_cap = l.hits.cap / 4;
        }
    }
    if (n > 0 && total > 0) {
        if (bp.filter_ok) {
            Timer t(c, l, "k_fill");
            k_fill<<<blocks_for(n, 256), 256, 0, st>>>(l.cnt.as<uint32_t>(), l.slots.as<uint32_t>(), l.slots_retry.as<uint32_t>(),
                                                       l.wave_base.as<uint64_t>(), l.slot_off.as<uint64_t>(), l.hit_offs.as<int64_t>(), l.route.as<uint32_t>(), n,
                                                       l.hits_ptr, l.hits_cap, l.ctr(), nullptr);
        }
        if (n_scan) {
            Timer t(c, l, "k_scan_fill");
            if (bl <= 16)
                k_scan_pk<true><<<grid_for(n_scan, sl_scan), 256, 0, st>>>(segs, offs, l.seg_bytes, l.list_scan.as<int32_t>(), (int64_t)n_scan,
                                                                          sl_scan, c->pat.as<uint2>(), np, bl, bp.max_err, l.min_e_ptr,
                                                                          pb_scan, pc_scan, l.hit_offs.as<int64_t>(), l.hits_ptr);
            else
                k_scan<false, true><<<grid_for(n_scan, sl_scan), 256, 0, st>>>(segs, offs, l.seg_bytes, l.list_scan.as<int32_t>(), (int64_t)n_scan,
                                                                              sl_scan, c->pat.as<uint2>(), np, bl, bp.max_err, bp.prefix_len,
                                                                              l.min_e_ptr, pb_scan, pc_scan, l.hit_offs.as<int64_t>(),
                                                                              l.hits_ptr);
        }
        if (n_dp) {
            Timer t(c, l, "k_dp_fill");
            k_scan<true, true><<<grid_for(n_dp, sl_dp), 256, 0, st>>>(segs, offs, l.seg_bytes, l.list_dp.as<int32_t>(), (int64_t)n_dp, sl_dp,
                                                                     c->pat.as<uint2>(), np, bl, bp.max_err, bp.prefix_len, l.min_e_ptr,
                                                                     pb_dp, pc_dp, l.hit_offs.as<int64_t>(), l.hits_ptr);
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    return SCT_OK;
}

// After the lane's queued work is done (the caller waited for it): read the counter record, report bad input, run the
// slow path if rows were listed.  On return the batch's results are complete on the device; *slow says whether the slow
// path ran (copies of min_e / counts queued before it are then stale, and the hits are 32-bit entries in the lane's own array),
// *redone whether the batch was queued a second time (stale copies likewise; the hits are where the first pass would have put them).
static int finish_batch(sct_ctx *c, Lane &l, int64_t row_base, bool *slow, bool *redone = nullptr) {
    const uint32_t *h = l.h_ctr;
    if (slow) *slow = false;
    const uint64_t bad_row = ctr64(h, CTR_BAD_ROW), bad_byte = ctr64(h, CTR_BAD_BYTE);
    if (bad_row != ~0ull)
        return fail(SCT_E_ARG, "the row index must not decrease and must end at the size of segs (row %llu)",
                    (unsigned long long)(bad_row + (uint64_t)row_base));
    if (bad_byte != ~0ull) return fail(SCT_E_ARG, "segment bytes contain a value >= 128 (row block starting at row %lld, byte offset %llu in it)",
                                       (long long)row_base, (unsigned long long)bad_byte);
    if (h[CTR_JOIN_OVERFLOW] && l.bp.use_join) {
        // a producer block of the partitioned join had more tuples for one partition than its region holds (rows far longer than
        // the usual 22 characters, or keys that crowd into one partition): the whole batch once more through the filter kernel
        l.bp.use_join = false;
        c->last_join += ((int64_t)1 << 32) | ((int64_t)h[CTR_JOIN_OVERFLOW] << 40);
        if (int rc = reset_counters(l)) return rc;
        if (int rc = queue_batch(c, l)) return rc;
        HIPCHK(hipStreamSynchronize(l.stream));
        HIPCHK(hipGetLastError());
        {
            // the batch was queued again: whatever the caller copied (or had copied) out of the first pass — min_e, counts, the block
            // sums a ring slot holds — is stale
            const int rc = finish_batch(c, l, row_base, slow, redone);
            if (redone) *redone = true;
            return rc;
        }
    }
    if (l.bp.filter_ok && l.n_seg > 0 && ((l.no64 && h[CTR_NEED64]) || (l.no_special && h[CTR_SPECIAL]))) {
        // the batch has rows for launches that were left out on the guess that no row needs them (sct_ctx::need64, need_special):
        // once more, with them
        if (l.no64 && h[CTR_NEED64]) c->need64 = true;
        if (l.no_special && h[CTR_SPECIAL]) c->need_special = true;
        c->redone64++;
        if (int rc = reset_counters(l)) return rc;
        if (int rc = queue_batch(c, l)) return rc;
        HIPCHK(hipStreamSynchronize(l.stream));
        HIPCHK(hipGetLastError());
        {
            // the batch was queued again: whatever the caller copied (or had copied) out of the first pass — min_e, counts, the block
            // sums a ring slot holds — is stale
            const int rc = finish_batch(c, l, row_base, slow, redone);
            if (redone) *redone = true;
            return rc;
        }
    }
    const int64_t n_retried = h[CTR_RETRY];
    if (h[CTR_LONG] || h[CTR_SCAN] || h[CTR_DP]) {
        if (int rc = slow_path(c, l)) return rc;
        if (slow) *slow = true;
        if (ctr64(h, CTR_BAD_BYTE) != ~0ull)
            return fail(SCT_E_ARG, "segment bytes contain a value >= 128 (row block starting at row %lld, byte offset %llu in it)",
                        (long long)row_base, (unsigned long long)ctr64(h, CTR_BAD_BYTE));
    }
    if (l.direct_hits && ctr64(h, CTR_TOTAL) > l.hits_cap) {
        // the fill kernel found the caller-visible hit buffer too small and wrote nothing: enlarge it, fill again
        if (int rc = ensure_host_hits(c, (size_t)ctr64(h, CTR_TOTAL) + 1)) return rc;
        l.hits_ptr = host_hits_device_view(c);
        if (!l.hits_ptr) return fail(SCT_E_STATE, "the page-locked result buffer is not mapped into the device");
        l.hits_cap = c->h_hits_cap;
        if (l.bp.filter_ok && l.n_seg > 0)
            k_fill<<<blocks_for(l.n_seg, 256), 256, 0, l.stream>>>(l.cnt.as<uint32_t>(), l.slots.as<uint32_t>(), l.slots_retry.as<uint32_t>(),
                                                                   l.wave_base.as<uint64_t>(), l.slot_off.as<uint64_t>(), l.hit_offs.as<int64_t>(), l.cnt.as<uint32_t>(),
                                                                   l.n_seg, l.hits_ptr, l.hits_cap, l.ctr(), nullptr);
        HIPCHK(hipStreamSynchronize(l.stream));
        HIPCHK(hipGetLastError());
    }
    if (l.bp.count_format == SCT_CNT_U8 && h[CTR_CNT_OVERFLOW] && !l.cnt8_soft)
        return fail(SCT_E_RANGE, "a row has more than 255 hits: ask for SCT_CNT_U32 or SCT_CNT_OFFS64 counts");
    if (l.bp.use_join) c->last_join += 1;
    l.n_hits = (int64_t)ctr64(h, CTR_TOTAL);
    l.stats[0] = l.n_seg - h[CTR_SCAN] - h[CTR_DP];
    l.stats[1] = h[CTR_SCAN];
    l.stats[2] = h[CTR_DP];
    l.stats[3] = l.n_hits;
    l.stats[4] = n_retried;
    l.stats[5] = h[CTR_SPECIAL];
    l.have_result = true;
    return SCT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// the resident three-step API (lane 0)
// ------------------------------------------------------------------------------------------------------------
// check_bytes: refuse any byte >= 128 (the two neighbouring stages); the match_trie path checks the rows it reads instead
static int upload_segments(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n, bool check_bytes) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    FaultScope fault_scope(&c->fault);
    if (n < 0 || !offs) return fail(SCT_E_ARG, "sct_upload_segments: bad offsets");
    if (n >= 0x7FFFFFF0) return fail(SCT_E_ARG, "at most 2^31-16 segments per call (shard the input)");
    if (offs[0] != 0) return fail(SCT_E_ARG, "offs[0] must be 0");
    const int64_t bytes = offs[n];
    if (bytes < 0) return fail(SCT_E_ARG, "offs[n] is negative");
    if (bytes > 0 && !segs) return fail(SCT_E_ARG, "segs is NULL");
    HIPCHK(hipSetDevice(c->device));
    Lane &l = c->lane0();
    if (l.segs.ensure((size_t)bytes + 64)) return g_code;
    if (l.offs64.ensure(sizeof(int64_t) * (size_t)(n + 1))) return g_code;
    hipStream_t st = l.stream;
    l.n_seg = -1;
    l.have_result = false;
    if (int rc = reset_counters(l)) return rc;
    if (int rc = copy_up(c, l.segs.p, segs, (size_t)bytes, st)) return rc;
    if (int rc = copy_up(c, l.offs64.p, offs, sizeof(int64_t) * (size_t)(n + 1), st)) return rc;
    // validation on the device, before any kernel trusts the offsets: non-decreasing, inside the byte buffer; the host only
    // reads back two words.  Bytes >= 128: see k_validate.
    const int64_t items = (!check_bytes || (n + 1) > (bytes + 15) / 16) ? (n + 1) : (bytes + 15) / 16;
    k_validate<<<blocks_for(items, 256), 256, 0, st>>>(l.segs.as<uint8_t>(), bytes, l.offs64.as<int64_t>(), n,
                                                      reinterpret_cast<unsigned long long *>(l.ctr() + CTR_BAD_ROW), check_bytes);
    unsigned long long bad[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(bad, l.ctr() + CTR_BAD_ROW, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (bad[0] != ~0ull) return fail(SCT_E_ARG, "offs must be non-decreasing and end at the size of segs (row %llu)", bad[0]);
    if (bad[1] != ~0ull) return fail(SCT_E_ARG, "segment bytes contain a value >= 128 at offset %llu", bad[1]);
    l.n_seg = n;
    l.seg_bytes = bytes;
    l.wide = true;
    l.have_offs64 = true;
    l.segs_ptr = l.segs.as<uint8_t>();
    l.host_segs = nullptr;
    return SCT_OK;
}

extern "C" int sct_upload_segments(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n) { return upload_segments(c, segs, offs, n, false); }

extern "C" int sct_run(sct_ctx *c, int32_t max_err, int32_t threads, int64_t *n_hits_out) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    FaultScope fault_scope(&c->fault);
    if (!c->have_bc) return fail(SCT_E_STATE, "sct_run before sct_set_barcodes");
    Lane &l = c->lane0();
    if (l.n_seg < 0 || !l.wide) return fail(SCT_E_STATE, "sct_run before sct_upload_segments");
    HIPCHK(hipSetDevice(c->device));
    if (int rc = batch_params(c, max_err, threads, SCT_CNT_OFFS64, l.bp)) return rc;
    if (int rc = lane_ensure(l, l.n_seg, l.bp)) return rc;
    c->last_ms.clear();
    l.timed.clear();
    l.pool_used = 0;
    l.last_end = nullptr;
    l.index_just_timed = false;
    if (int rc = reset_counters(l)) return rc;
    if (int rc = queue_batch(c, l)) return rc;
    HIPCHK(hipStreamSynchronize(l.stream));  // the one host synchronisation of a run without listed rows
    HIPCHK(hipGetLastError());
    if (int rc = finish_batch(c, l, 0, nullptr)) return rc;
    collect_timing(c, l);
    memcpy(c->stats, l.stats, sizeof(c->stats));
    if (n_hits_out) *n_hits_out = l.n_hits;
    return SCT_OK;
}

extern "C" int sct_download(sct_ctx *c, uint8_t *min_e, int64_t *hit_offs, uint32_t *hits) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    FaultScope fault_scope(&c->fault);
    Lane &l = c->lane0();
    if (!l.have_result) return fail(SCT_E_STATE, "sct_download without a completed sct_run");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = l.n_seg;
    if (min_e && n)
        if (int rc = copy_down(c, min_e, l.min_e.p, (size_t)n, l.stream)) return rc;
    if (hit_offs)
        if (int rc = copy_down(c, hit_offs, l.hit_offs.p, 8 * (size_t)(n + 1), l.stream)) return rc;
    if (hits && l.n_hits)
        if (int rc = copy_down(c, hits, l.hits.p, 4 * (size_t)l.n_hits, l.stream)) return rc;
    HIPCHK(hipStreamSynchronize(l.stream));
    return SCT_OK;
}

// sct_match's hit arrays: malloc'ed, handed to the caller, given back through sct_free_hits.  The last few that came back are kept
// for the next call (any context, any thread): 33 MB per 10 M rows at 25 k barcodes are 8 000 pages to map, fault in and unmap
// again every call otherwise — a millisecond of a call that takes six.
struct HitArrays {
    static constexpr int KEEP = 2;
    static constexpr size_t KEEP_BYTES_MAX = (size_t)256 << 20;   // (larger arrays go back to the allocator at once)
    std::mutex mu;
    struct Known {
        uint32_t *p;
        size_t bytes;
    };
    std::vector<Known> out;    // arrays the callers hold: their sizes
    std::vector<Known> kept;   // arrays that came back
    uint32_t *get(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < kept.size(); i++)
                if (kept[i].bytes >= bytes && kept[i].bytes <= 2 * bytes + ((size_t)1 << 20)) {
                    Known k = kept[i];
                    kept.erase(kept.begin() + (ptrdiff_t)i);
                    out.push_back(k);
                    return k.p;
                }
        }
        uint32_t *p = (uint32_t *)malloc(bytes ? bytes : 4);
        if (p) {
            std::lock_guard<std::mutex> lk(mu);
            out.push_back(Known{p, bytes});
        }
        return p;
    }
    void put(uint32_t *p) {
        if (!p) return;
        uint32_t *drop = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t bytes = 0;
            bool known = false;
            for (size_t i = 0; i < out.size(); i++)
                if (out[i].p == p) {
                    bytes = out[i].bytes;
                    out.erase(out.begin() + (ptrdiff_t)i);
                    known = true;
                    break;
                }
            if (!known || bytes < ((size_t)1 << 20) || bytes > KEEP_BYTES_MAX) drop = p;   // small ones: the allocator's business
            else {
                kept.push_back(Known{p, bytes});
                if ((int)kept.size() > KEEP) {
                    drop = kept.front().p;
                    kept.erase(kept.begin());
                }
            }
        }
        free(drop);
    }
};
static HitArrays g_hit_arrays;

#include "sct_stream.hip.h"

extern "C" int sct_match(sct_ctx *c, const char *segs, const int64_t *offs, int64_t n, int32_t max_err, int32_t threads,
                         uint8_t *min_e, int64_t *hit_offs, uint32_t **hits, int64_t *n_hits) {
    if (!hits || !n_hits || !min_e || !hit_offs) return fail(SCT_E_ARG, "sct_match: NULL output pointer");
    FaultScope fault_scope(c ? &c->fault : nullptr);
    *hits = nullptr;
    *n_hits = 0;
    if (n < 0 || !offs) return fail(SCT_E_ARG, "sct_match: bad offsets");
    const void *h = nullptr;
    int64_t total = 0;
    uint32_t *sink = nullptr;   // the caller's hit array, filled chunk by chunk while the call runs (NULL: it could not be, see below)
    int rc = match_stream(c, segs, offs[n], offs, SCT_IDX_OFFS64, n, max_err, threads, min_e, hit_offs, SCT_CNT_OFFS64, SCT_HITS_U32, &h, &total, &sink);
    if (rc) return rc;
    if (!sink) {
        // (a call of no rows; an array that could not be had; a result buffer that grew while the call ran)
        sink = inject_alloc_failure() ? nullptr : g_hit_arrays.get(sizeof(uint32_t) * (size_t)(total ? total : 1));
        if (!sink) return fail(SCT_E_NOMEM, "out of host memory for %lld hits", (long long)total);
        par_copy(c, sink, h, sizeof(uint32_t) * (size_t)total);   // (33 MB per 10 M rows at 25 k barcodes: on the copy pool)
    }
    *hits = sink;
    *n_hits = total;
    return SCT_OK;
}

extern "C" void sct_free_hits(sct_ctx *, uint32_t *hits) { g_hit_arrays.put(hits); }

extern "C" int sct_last_timing_ex(sct_ctx *c, const char **names, float *ms, int32_t *launches, int cap) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    const int k = (int)c->last_ms.size();
    for (int i = 0; i < k && i < cap; i++) {
        if (names) names[i] = c->last_ms[i].first;
        if (ms) ms[i] = (float)c->last_ms[i].second.ms;
        if (launches) launches[i] = c->last_ms[i].second.launches;
    }
    return k;
}

extern "C" int sct_last_timing(sct_ctx *c, const char **names, float *ms, int cap) { return sct_last_timing_ex(c, names, ms, nullptr, cap); }

extern "C" int sct_last_stats(sct_ctx *c, int64_t *out, int cap) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    for (int i = 0; i < 5 && i < cap; i++) out[i] = c->stats[i];
    if (cap > 5) out[5] = c->last_scheme;
    if (cap > 6) out[6] = c->last_join;
    if (cap > 7) out[7] = c->stats[5];
    return 8;
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
    FaultScope fault_scope(&c->fault);
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
    hipStream_t st = c->lane0().stream;
    if (c->wl_table.ensure(tab.size() * sizeof(SctWlSlot))) return g_code;
    if (int rc = copy_up(c, c->wl_table.p, tab.data(), tab.size() * sizeof(SctWlSlot), st)) return rc;
    HIPCHK(hipStreamSynchronize(st));
    c->wl_mask = mask;
    c->wl_len_set = len_set;
    c->wl_n = n;
    return SCT_OK;
}

extern "C" int sct_wl_count(sct_ctx *c, const char *segs, const int64_t *offs, const uint8_t *skip, int64_t n, int64_t *counts,
                            int64_t *first_seen) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    FaultScope fault_scope(&c->fault);
    if (c->wl_n < 0) return fail(SCT_E_STATE, "sct_wl_count before sct_wl_set");
    if (!counts || !first_seen || (n > 0 && !skip)) return fail(SCT_E_ARG, "sct_wl_count: NULL argument");
    int rc = upload_segments(c, segs, offs, n, true);
    if (rc) return rc;
    Lane &l = c->lane0();
    const size_t nk = (size_t)(c->wl_n > 0 ? c->wl_n : 1);
    if (c->wl_counts.ensure(4 * nk) || c->wl_first.ensure(4 * nk) || c->wl_skip.ensure((size_t)(n ? n : 1))) return g_code;
    hipStream_t st = l.stream;
    c->last_ms.clear();
    l.timed.clear();
    l.pool_used = 0;
    l.last_end = nullptr;
    l.index_just_timed = false;
    HIPCHK(hipMemsetAsync(c->wl_counts.p, 0, 4 * nk, st));
    HIPCHK(hipMemsetAsync(c->wl_first.p, 0xFF, 4 * nk, st));
    if (n > 0) {
        if (int rc = copy_up(c, c->wl_skip.p, skip, (size_t)n, st)) return rc;
        Timer t(c, l, "k_wl_count");
        k_wl_count<<<blocks_for(n, 256), 256, 0, st>>>(l.segs.as<uint8_t>(), l.offs64.as<int64_t>(), c->wl_skip.as<uint8_t>(), n,
                                                      c->wl_table.as<SctWlSlot>(), c->wl_mask, c->wl_len_set,
                                                      c->wl_counts.as<uint32_t>(), c->wl_first.as<uint32_t>());
    }
    std::vector<uint32_t> hc(nk), hf(nk);
    if (int rc = copy_down(c, hc.data(), c->wl_counts.p, 4 * nk, st)) return rc;
    if (int rc = copy_down(c, hf.data(), c->wl_first.p, 4 * nk, st)) return rc;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    collect_timing(c, l);
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
    FaultScope fault_scope(&c->fault);
    if (!adapter || !strand || !dist || !loc_offs || !locs || !n_locs) return fail(SCT_E_ARG, "sct_adapter_align: NULL argument");
    *locs = nullptr;
    *n_locs = 0;
    if (m < 1) return fail(SCT_E_ARG, "adapter length must be positive");
    if (m > 32) return fail(SCT_E_UNSUPPORTED, "adapter length %d exceeds the supported maximum 32", m);
    for (int i = 0; i < m; i++)
        if ((uint8_t)adapter[i] >= 128) return fail(SCT_E_ARG, "adapter contains a byte >= 128");
    int rc = upload_segments(c, reads, offs, n, true);
    if (rc) return rc;
    Lane &l = c->lane0();
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
    if (c->ad_peq.ensure(peq.size() * 4) || c->ad_strand.ensure(n1) || c->ad_dist.ensure(4 * n1) || l.cnt.ensure(4 * n1) ||
        c->ad_first.ensure(8 * n1) || c->ad_last.ensure(8 * n1) || l.hit_offs.ensure(8 * (n1 + 1)) ||
        l.counters.ensure(sizeof(uint32_t) * CTR_WORDS))
        return g_code;
    const int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (l.sums.ensure(8 * (size_t)(nb ? nb : 1))) return g_code;
    hipStream_t st = l.stream;
    c->last_ms.clear();
    l.timed.clear();
    l.pool_used = 0;
    l.last_end = nullptr;
    l.index_just_timed = false;
    l.have_result = false;
    uint64_t *d_total = reinterpret_cast<uint64_t *>(l.ctr() + CTR_TOTAL);
    uint64_t total = 0;
    int32_t *h = nullptr;
    // every exit below goes through `done`: the stream is synchronised before peq (a local) goes away and `h` is released on failure
    hipError_t e = hipMemcpyAsync(c->ad_peq.p, peq.data(), peq.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n > 0) {
        {
            Timer t(c, l, "k_adapter_count");
            k_adapter_count<<<blocks_for(n, 256), 256, 0, st>>>(l.segs.as<uint8_t>(), l.offs64.as<int64_t>(), n, c->ad_peq.as<uint32_t>(), m,
                                                               c->ad_strand.as<int8_t>(), c->ad_dist.as<int32_t>(), l.cnt.as<uint32_t>(),
                                                               c->ad_first.as<int64_t>(), c->ad_last.as<int64_t>());
        }
        {
            Timer t(c, l, "k_offsets");
            k_sum_blocks<uint32_t><<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), nullptr);
            k_scan_sums<<<1, 256, 0, st>>>(l.sums.as<uint64_t>(), nb, d_total);
            k_offsets<<<(unsigned)nb, 256, 0, st>>>(l.cnt.as<uint32_t>(), n, l.sums.as<uint64_t>(), d_total, l.hit_offs.as<int64_t>(), nullptr,
                                                    nullptr, l.ctr());
        }
        e = hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st);
    }
    {
        const hipError_t es = hipStreamSynchronize(st);
        if (e == hipSuccess) e = es;
    }
    if (e == hipSuccess && n > 0 && total) {
        if (c->ad_locs.ensure(4 * (size_t)total)) return g_code;
        Timer t(c, l, "k_adapter_fill");
        k_adapter_fill<<<blocks_for(n, 256), 256, 0, st>>>(l.segs.as<uint8_t>(), l.offs64.as<int64_t>(), n, c->ad_peq.as<uint32_t>(), m,
                                                          c->ad_strand.as<int8_t>(), c->ad_dist.as<int32_t>(), c->ad_first.as<int64_t>(),
                                                          c->ad_last.as<int64_t>(), l.hit_offs.as<int64_t>(), c->ad_locs.as<int32_t>());
    }
    if (e == hipSuccess) {
        h = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
        if (!h) {
            (void)hipStreamSynchronize(st);
            return fail(SCT_E_NOMEM, "out of host memory");
        }
    }
    if (n == 0) loc_offs[0] = 0;
    if (e == hipSuccess && n > 0) {
        int rc2 = copy_down(c, strand, c->ad_strand.p, (size_t)n, st);
        if (!rc2) rc2 = copy_down(c, dist, c->ad_dist.p, 4 * (size_t)n, st);
        if (!rc2) rc2 = copy_down(c, loc_offs, l.hit_offs.p, 8 * (size_t)(n + 1), st);
        if (!rc2 && total) rc2 = copy_down(c, h, c->ad_locs.p, 4 * (size_t)total, st);
        if (rc2) {
            (void)hipStreamSynchronize(st);
            free(h);
            return rc2;
        }
    }
    {
        const hipError_t es = hipStreamSynchronize(st);
        if (e == hipSuccess) e = es;
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e != hipSuccess) {
        free(h);
        return fail(SCT_E_HIP, "adapter alignment failed: %s", hipGetErrorString(e));
    }
    collect_timing(c, l);
    *locs = h;
    *n_locs = (int64_t)total;
    return SCT_OK;
}

extern "C" void sct_free_locs(sct_ctx *, int32_t *locs) { free(locs); }
