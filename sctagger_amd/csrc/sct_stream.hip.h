// sct_stream.hip.h — the stream API: the seam (reference scTagger.py:688-716) in one call.  Part of sctagger_hip.hip (included
// there, behind the batch functions it drives: queue_batch, finish_batch, slow_path).
//
// Three routes, chosen by where the caller's arrays live:
//
//   ZERO COPY   page-locked caller buffers (sct_host_alloc, hipHostMalloc).  The kernels read segs / index and write min_e /
//               counts / hits in the caller's memory through its device mapping; the call is ONE chunk of up to 2^24 rows on
//               lane 0 (more rows, or more than 3 GiB of segment bytes: further chunks, one after the other on lanes 0..L-1):
//                 index -> uint32 offsets (k_index_*) -> filter launches (rows cross the link inside k_filter) -> offsets -> fill
//
//   RING        ordinary (pageable) caller arrays — what the reference's seam hands over (scTagger.py:616-627: RawArray) and
//               what sct_match gets from numpy.  The context owns a ring of page-locked slots and a pool of copy threads
//               (sct_copy_pool.h).  The rows are cut into chunks of a third of the call, at most 2^20 rows; for chunk k
//                 pool:    memcpy segs[k] into a slot, the 64- / 32-bit offsets narrowed to one-byte lengths on the way
//                          (sct_hostcopy.c), the lengths' block prefix sums read off the offsets
//                 device:  the zero-copy kernels on the slot (min_e and one-byte counts written back into it), hits by DMA
//                          into the context's page-locked result buffer
//                 pool:    slot -> caller: min_e copied, the counts widened to the caller's format (int64 CSR offsets:
//                          prefix sums seeded by the device's block sums); sct_match's hit array filled from the result buffer
//               with chunks k+1 .. on their way in while up to "ring_lanes" chunks are on the device and chunk k-1 is on its way
//               out; the last part of a chunk's way out hands its slot to the next user (no thread waits for a slot).  Nothing of
//               the caller's is page-locked and no caller pointer is ever given to the runtime: the link carries 220 MB up and
//               55 MB down per 10 M rows instead of 289 / 123, and the calling thread no longer sits in staged copies.
//               (Round 4 locked the caller's pages for the call instead — hipHostRegister, option "register": one soak run in a
//               few ended in a GPU memory fault on a host address; removed in round 5)
//
//   STAGED      page-locked buffers with the option "zero_copy" = 0, or pageable ones with "ring" = 0: the rows are uploaded in
//               chunks (option "chunk_rows") that move through L lanes (option "lanes", default 5)
//                 lane k mod L:   H2D(chunk k) -> index / byte checks -> filter launches -> offsets -> fill -> D2H(results)
//               (measured: DESIGN.md 5b — 6.6 - 8.2 ms per 10 M rows page-locked where zero copy needs 4.8).  Memory that is not
//               page-locked goes through the context's two bounce pieces here as everywhere else (copy_up / copy_down), one piece
//               after the other on the calling thread: 11 - 13 ms per 10 M rows — the ring route without its pipeline.
#pragma once
#include <deque>
#include <memory>

#include "sct_copy_pool.h"

static int ensure_host_hits(sct_ctx *c, size_t entries) {
    if (entries <= c->h_hits_cap) return SCT_OK;
    // copies into the old buffer may still be in flight
    for (Lane *l : c->lanes) HIPCHK(hipStreamSynchronize(l->stream));
    size_t want = c->h_hits_cap ? c->h_hits_cap : (size_t)1 << 20;
    while (want < entries) want *= 2;
    uint32_t *p = nullptr;
    hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipHostMalloc((void **)&p, want * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "hipHostMalloc(%zu) for the hit buffer: %s", want * sizeof(uint32_t), hipGetErrorString(e));
    }
    if (c->h_hits) {
        memcpy(p, c->h_hits, c->h_hits_cap * sizeof(uint32_t));
        (void)hipHostFree(c->h_hits);
    }
    c->h_hits = p;
    c->h_hits_cap = want;
    return SCT_OK;
}

// the page-locked result buffer as the kernels must address it (hipHostMalloc memory is mapped into the device's address space;
// on ROCm the two addresses coincide, but that is the runtime's business): NULL if it is not mapped
static uint32_t *host_hits_device_view(sct_ctx *c) {
    if (!c->h_hits) return nullptr;
    hipPointerAttribute_t at;
    uint32_t *d = nullptr;
    if (hipPointerGetAttributes(&at, c->h_hits) == hipSuccess && at.type == hipMemoryTypeHost) d = (uint32_t *)at.devicePointer;
    (void)hipGetLastError();
    return d;
}

// internal index format of a ring slot: uint32 offsets relative to the chunk (a chunk with a row of more than 65 535 characters)
enum { SCT_IDX_OFFSU32_INTERNAL = 4 };

static inline size_t index_entry_size(int fmt) {
    return fmt == SCT_IDX_OFFS64 ? 8 : (fmt == SCT_IDX_OFFS32 || fmt == SCT_IDX_OFFSU32_INTERNAL) ? 4 : fmt == SCT_IDX_LEN16 ? 2 : 1;
}

// Mapped (page-locked) host memory as the device addresses it, or NULL — for the WHOLE array: its first and its last byte must
// both lie in mapped host memory, the same distance apart on the device.
static const void *mapped_view(const void *p, size_t bytes) {
    if (!p) return nullptr;
    hipPointerAttribute_t at;
    const char *d0 = nullptr, *d1 = nullptr;
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) d0 = (const char *)at.devicePointer;
    (void)hipGetLastError();
    if (!d0) return nullptr;
    if (bytes <= 1) return d0;
    if (hipPointerGetAttributes(&at, (const char *)p + (bytes - 1)) == hipSuccess && at.type == hipMemoryTypeHost) d1 = (const char *)at.devicePointer;
    (void)hipGetLastError();
    return (d1 && d1 - d0 == (ptrdiff_t)(bytes - 1)) ? d0 : nullptr;
}

// ------------------------------------------------------------------------------------------------------------
// the context's page-locked ring and its copy pool
// ------------------------------------------------------------------------------------------------------------
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static void ring_slot_release(RingSlot &s) {
    if (s.base) (void)hipHostFree(s.base);
    s = RingSlot();
}

// room for seg_need segment bytes and row_need rows (sizes grow in powers of two, so that a run of calls of growing size
// does not allocate every time)
static int ring_slot_ensure(RingSlot &s, size_t seg_need, int64_t row_need) {
    if (s.base && s.seg_cap >= seg_need && s.row_cap >= row_need) return SCT_OK;
    size_t seg_cap = (size_t)1 << 16;
    while (seg_cap < seg_need) seg_cap <<= 1;
    int64_t row_cap = 4096;
    while (row_cap < row_need) row_cap <<= 1;
    if (s.base) {
        if (seg_cap < s.seg_cap) seg_cap = s.seg_cap;
        if (row_cap < s.row_cap) row_cap = s.row_cap;
    }
    ring_slot_release(s);
    const size_t o_idx = align256(seg_cap + 64);
    const size_t o_isums = o_idx + align256(4 * (size_t)(row_cap + 1) + 16);
    const size_t o_min = o_isums + align256(8 * (size_t)(row_cap / SCAN_BLOCK + 3));
    const size_t o_cnt = o_min + align256((size_t)row_cap);
    const size_t o_sums = o_cnt + align256((size_t)row_cap);
    const size_t total = o_sums + align256(8 * (size_t)(row_cap / SCAN_BLOCK + 2));
    void *p = nullptr;
    hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipHostMalloc(&p, total, hipHostMallocDefault);
    void *d = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, p, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipHostFree(p);
        return fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "hipHostMalloc(%zu) for a slot of the page-locked ring: %s", total, hipGetErrorString(e));
    }
    s.base = (char *)p;
    s.dev = (char *)d;
    s.bytes = total;
    s.seg_cap = seg_cap;
    s.row_cap = row_cap;
    s.o_idx = o_idx;
    s.o_isums = o_isums;
    s.o_min = o_min;
    s.o_cnt = o_cnt;
    s.o_sums = o_sums;
    return SCT_OK;
}

static CopyPool *ensure_pool(sct_ctx *c) {
    if (c->pool) return c->pool;
    int t = c->copy_threads;
    if (t < 0) {   // by the CPUs the process may use: half of them, at most 8 (the calling thread copies too while it waits)
        t = sct_effective_cpus() / 2;
        t = t < 1 ? 1 : (t > 8 ? 8 : t);
    }
    try {
        c->pool = new CopyPool(t);
    } catch (...) {   // no threads to be had: the calling thread does every copy itself (CopyPool::wait)
        c->pool = new CopyPool(0);
    }
    return c->pool;
}

// page-locked scratch of the context (a chunk whose one-byte counts overflowed: its 64-bit offsets come down here)
static int ensure_scratch(sct_ctx *c, size_t bytes) {
    if (bytes <= c->h_scratch_cap) return SCT_OK;
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    c->h_scratch = nullptr;
    c->h_scratch_cap = 0;
    void *p = nullptr;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? SCT_E_NOMEM : SCT_E_HIP, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
    }
    c->h_scratch = p;
    c->h_scratch_cap = want;
    return SCT_OK;
}

// n bytes from src to dst on the pool (the calling thread takes its share): what sct_match does with the hit array
struct ParCopy {
    char *dst;
    const char *src;
    size_t bytes;
};
static void par_copy_part(void *arg, int part, int parts) {
    const ParCopy *pc = (const ParCopy *)arg;
    const size_t a = (pc->bytes * (size_t)part / (size_t)parts) & ~(size_t)63;
    const size_t b = part + 1 == parts ? pc->bytes : ((pc->bytes * (size_t)(part + 1) / (size_t)parts) & ~(size_t)63);
    if (b > a) memcpy(pc->dst + a, pc->src + a, b - a);
}
static void par_copy(sct_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes < ((size_t)1 << 20)) {
        if (bytes) memcpy(dst, src, bytes);
        return;
    }
    CopyPool *pool = ensure_pool(c);
    ParCopy pc{(char *)dst, (const char *)src, bytes};
    CopyGroup g;
    size_t parts = bytes >> 19;   // pieces of at least half a megabyte
    if (parts > (size_t)pool->threads() + 1) parts = (size_t)pool->threads() + 1;
    pool->submit(&g, par_copy_part, &pc, (int)(parts < 1 ? 1 : parts));
    pool->wait(&g);
}

// ------------------------------------------------------------------------------------------------------------
// one call
// ------------------------------------------------------------------------------------------------------------
struct ChunkPlan {
    int64_t r0 = 0, r1 = 0;  // rows
    int64_t b0 = 0, b1 = 0;  // bytes of segs
    int64_t hit_base = 0;
    uint64_t narrow_cap = 0;  // 16-bit hits: what the narrowing kernel queued with the chunk could write
};

// where one chunk's rows come from and where its per-row results go
struct ChunkIO {
    const uint8_t *segs_dev = nullptr;  // mapped host memory the kernels read in place; NULL: the bytes are uploaded from segs_src
    const char *segs_src = nullptr;     // the chunk's bytes on the host (of a zero-copy batch: the same memory as segs_dev)
    const void *idx_dev = nullptr;      // the chunk's index entries in mapped host memory; NULL: uploaded from idx_src
    const void *idx_src = nullptr;
    int idx_format = SCT_IDX_OFFS64;
    const uint64_t *idx_sums_dev = nullptr;   // row lengths whose block prefix sums the host already has (the ring route narrowed them from offsets):
                                              // bytes in front of every SCAN_BLOCK rows, then the total — the device need not scan
    uint8_t *min_e_dev = nullptr;       // mapped host memory the kernels write min_e to; NULL: the lane's buffer, copied to min_e_dst
    uint8_t *min_e_dst = nullptr;
    void *cnt_dev = nullptr;            // the same for the compact counts (one or four bytes per row)
    void *cnt_dst = nullptr;
};

struct StreamCall;

// a chunk of the ring route
struct RingChunk {
    StreamCall *call = nullptr;
    int64_t k = 0;
    RingSlot *slot = nullptr;
    CopyGroup in, out;
    RingChunk *next_user = nullptr;      // the chunk that takes this chunk's slot next: its way in is queued by the last part of this chunk's way out
    std::atomic<uint64_t> len_bits{0};   // OR of the row lengths that do not fit one byte (top bit: offsets that decrease)
    int idx_format = SCT_IDX_LEN8;       // what the slot's index holds
    int64_t hit_base = 0, n_hits = 0;
    const int64_t *wide_offs = nullptr;  // a row with more than 255 hits: the chunk's 64-bit offsets (relative), instead of the slot's one-byte counts
    bool submitted_out = false;
    bool have_isums = false;             // the slot holds the block prefix sums of the chunk's row lengths (ring_wait_in)
    // SCT_TRACE_RING=1: when (ms since the call began) the chunk's copies ran and the calling thread passed its stations
    std::atomic<int64_t> in_first{INT64_MAX}, in_last{0}, out_first{INT64_MAX}, out_last{0};
    double t_in_ready = 0, t_queued = 0, t_wait = 0, t_device_done = 0, t_out_submitted = 0;
};

struct StreamCall {
    sct_ctx *c;
    // the call
    const char *segs;
    int64_t seg_bytes;
    const void *index;
    int index_format;
    int64_t n;
    uint8_t *min_e;
    void *counts;
    int count_format;
    bool hits16;
    BatchParams bp;
    bool by_offs = false;
    size_t isz = 0;
    // the caller's arrays as the device addresses them (NULL: not page-locked, or "zero_copy" off)
    const uint8_t *zc_segs = nullptr;
    const char *zc_index = nullptr;
    uint8_t *zc_min_e = nullptr;
    void *zc_counts = nullptr;
    uint32_t *zc_hits = nullptr;
    uint16_t *zc_hits16 = nullptr;
    bool pl_min_e = false, pl_counts = false;   // the caller's min_e / counts arrays are page-locked (a copy to them may be left queued)
    // the plan
    bool ring = false;
    int ramp = 0;            // ring route: the first chunk has chunk >> ramp rows, the next ones double up to `chunk` rows
    int64_t chunk = 0, n_chunks = 0;
    int L = 1, S = 0;
    std::vector<ChunkPlan> plan;
    std::unique_ptr<RingChunk[]> rc;
    int64_t hit_base = 0;
    std::atomic<int64_t> ns_in{0}, ns_out{0};   // the pool's time on this call's copies, all threads added up
    int64_t first_in_submitted = 0;   // ways in this thread queued itself (the first S chunks)
    // sct_match: its malloc'ed hit array is filled while the call runs — a chunk's hits are copied out of the page-locked result
    // buffer (on the pool) once the lane's next event says their DMA has landed — instead of in one piece at the end
    bool want_sink = false;
    uint32_t *sink = nullptr;
    size_t sink_entries = 0;
    struct HitSpan {
        int64_t base = 0, n = 0;
    };
    std::vector<HitSpan> lane_hits;      // per lane: the hits whose DMA was queued last and has not been copied out yet
    std::deque<ParCopy> hit_copies;      // (stable addresses: the pool's jobs point at them)
    CopyGroup hit_group;
    void sink_flush_lane(int lane);
    void sink_drop();
    bool trace = false;
    int64_t t_begin_ns = 0;
    double since_begin() const;
    void note(std::atomic<int64_t> &first, std::atomic<int64_t> &last, int64_t t0, int64_t t1) const;

    int64_t offs_at(int64_t i) const { return index_format == SCT_IDX_OFFS64 ? ((const int64_t *)index)[i] : (int64_t)((const int32_t *)index)[i]; }
    Lane &lane_of(int64_t k) const { return *c->lanes[(size_t)(k % L)]; }

    int choose_route();
    int make_plan();
    int phase_a(int64_t k, const ChunkIO &io);
    int phase_b(int64_t k);
    ChunkIO caller_io(int64_t k) const;
    // ring route
    int ring_prepare();
    void ring_submit_in(int64_t k);
    int ring_wait_in(int64_t k);
    ChunkIO ring_io(int64_t k) const;
    int ring_after_b(int64_t k, bool stale);
    void ring_quiesce();
    int run();
};

static inline double clk_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline int64_t clk_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double StreamCall::since_begin() const { return (double)(clk_ns() - t_begin_ns) * 1e-6; }
void StreamCall::note(std::atomic<int64_t> &first, std::atomic<int64_t> &last, int64_t t0, int64_t t1) const {
    if (!trace) return;
    int64_t cur = first.load(std::memory_order_relaxed);
    while (t0 < cur && !first.compare_exchange_weak(cur, t0, std::memory_order_relaxed)) {}
    cur = last.load(std::memory_order_relaxed);
    while (t1 > cur && !last.compare_exchange_weak(cur, t1, std::memory_order_relaxed)) {}
}

int StreamCall::choose_route() {
    by_offs = index_format == SCT_IDX_OFFS64 || index_format == SCT_IDX_OFFS32;
    isz = index_entry_size(index_format);
    const size_t counts_bytes = count_format == SCT_CNT_OFFS64 ? 8 * (size_t)(n + 1) : count_format == SCT_CNT_U32 ? 4 * (size_t)n : (size_t)n;
    const size_t index_bytes = isz * (size_t)(by_offs ? n + 1 : n);
    // Page-locked host memory (sct_host_alloc, hipHostMalloc) is mapped into the device's address space: the kernels read the rows
    // and their index from it directly ("zero copy"), at the rate of the PCIe link (57 GB/s measured, profiles/r02_pcie_probe.json),
    // every byte once — there is no upload step to wait for.
    if (c->zero_copy) {
        zc_segs = (const uint8_t *)mapped_view(segs, (size_t)seg_bytes);
        zc_index = (const char *)mapped_view(index, index_bytes);
        // outputs the kernels can write in place: min_e and the compact counts; the 64-bit offsets need the hits of the earlier
        // chunks added and stay on the copy path
        zc_min_e = (uint8_t *)mapped_view(min_e, (size_t)n);
        zc_counts = count_format == SCT_CNT_OFFS64 ? nullptr : (void *)mapped_view(counts, counts_bytes);
    }
    pl_min_e = is_page_locked(min_e) && is_page_locked(min_e + (n ? n - 1 : 0));
    pl_counts = is_page_locked(counts) && is_page_locked((const char *)counts + (counts_bytes ? counts_bytes - 1 : 0));
    hipPointerAttribute_t at;
    const bool pinned = zc_segs != nullptr || (segs && hipPointerGetAttributes(&at, segs) == hipSuccess && at.type == hipMemoryTypeHost);
    (void)hipGetLastError();
    ring = c->zero_copy && c->use_ring && !pinned;
    if (getenv("SCT_TRACE_STREAM"))   // one line per call: the route, and which of the caller's arrays the kernels touch in place
        fprintf(stderr, "sctagger_hip: stream n=%lld segs=%p+%lld index=%p+%zu min_e=%p counts=%p+%zu route=%s; in place: segs %p index %p min_e %p counts %p\n",
                (long long)n, (const void *)segs, (long long)seg_bytes, index, index_bytes, (void *)min_e, counts, counts_bytes,
                ring ? "ring" : zc_segs ? "zero-copy" : "staged", (const void *)zc_segs, (const void *)zc_index, (void *)zc_min_e, zc_counts);
    chunk = c->chunk_rows;
    if (ring) {
        // A chunk costs about a tenth of a millisecond beyond its rows (its seventeen dispatches, its ramp: every resident wave asking
        // for its rows at once) — 3 M rows take 2.4 ms in three chunks and 3.2 ms in eight (profiles/r05_pageable_chunk_sweep.jsonl) —
        // and the first chunk's way in and the last one's way out are not overlapped with anything: a third of the call's rows per
        // chunk, at most 2^20 (a slot of the ring is sized by it: 38 MB), at least 2^17.
        if (chunk <= 0) {
            chunk = c->ring_rows > 0 ? c->ring_rows : (int64_t)1 << 20;
            const int64_t third = (n + 2) / 3;
            if (third < chunk) chunk = third < ((int64_t)1 << 17) ? (int64_t)1 << 17 : third;
            // option "ring_ramp" = r: the first chunk holds 1 / 2^r of that, the next ones double (the device starts earlier; measured
            // without effect at 10 M rows and a loss on small calls: off by default)
            ramp = c->ring_ramp;
        }
    } else if (chunk <= 0) {
        // Zero copy: ONE launch over as many rows as possible — a launch begins with every resident wave asking for its rows at
        // once and ends with compute only, about a quarter of a millisecond in which link and chip do not overlap
        // (1 M-row launches: 0.58 ms each against 0.43 ms per 1 M rows of a 10 M-row launch).  Uploaded page-locked segments:
        // 2^20 rows per chunk keep the lanes busy.  Pageable segments with the ring switched off: every copy goes through the bounce
        // pieces on this thread, so nothing overlaps anyway: one chunk.
        chunk = zc_segs ? (int64_t)1 << 24 : pinned ? (int64_t)1 << 20 : (int64_t)1 << 24;
        // the offsets on the device are 32-bit and relative to the chunk
        const int64_t by_bytes = seg_bytes / ((int64_t)3 << 30) + 1;
        if ((n + chunk - 1) / chunk < by_bytes) chunk = (n + by_bytes - 1) / by_bytes;
    }
    return SCT_OK;
}

// Row lengths p[r0 .. r1): how many of them fit `budget` bytes (at least one), and their sum — block sums first, row by row only in
// the block that crosses the budget (10 M one-byte lengths: 1 ms instead of 15)
template <typename T>
static void lens_until(const T *p, int64_t r0, int64_t r1, int64_t budget, int64_t &end, int64_t &sum) {
    int64_t i = r0, s = 0;
    while (i < r1) {
        const int64_t e = i + 4096 < r1 ? i + 4096 : r1;
        int64_t t = 0;
        for (int64_t j = i; j < e; j++) t += p[j];
        if (s + t > budget) {
            for (; i < e; i++) {
                if (i > r0 && s + (int64_t)p[i] > budget) break;
                s += p[i];
            }
            break;
        }
        s += t;
        i = e;
    }
    end = i;
    sum = s;
}

// rows and bytes of every chunk.  Staged / zero copy: `chunk` rows each.  Ring: at most `chunk` rows and at most the bytes a slot
// of the ring is meant to hold (32 per row of a full chunk; a single longer row gets a slot of its own size).
int StreamCall::make_plan() {
    const int64_t seg_budget = ring ? (chunk * 32 < ((int64_t)1 << 20) ? (int64_t)1 << 20 : chunk * 32) : (int64_t)0xFFFFFFF0ll - 1;
    int64_t r0 = 0, b0 = 0;
    plan.clear();
    if (!by_offs && !ring && (n + chunk - 1) / chunk == 1) {
        // the only chunk of a call on row lengths: the device checks that they add up to seg_bytes (k_index_lens); the host need not read them
        ChunkPlan cp;
        cp.r0 = 0;
        cp.r1 = n;
        cp.b0 = 0;
        cp.b1 = seg_bytes;
        plan.push_back(cp);
    } else {
        int64_t rows = ramp > 0 ? ((chunk >> ramp) < 32768 ? (chunk < 32768 ? chunk : 32768) : (chunk >> ramp)) : chunk;
        while (r0 < n) {
            ChunkPlan cp;
            cp.r0 = r0;
            cp.b0 = by_offs ? offs_at(r0) : b0;
            int64_t r1 = r0 + rows < n ? r0 + rows : n;
            rows = rows * 2 < chunk ? rows * 2 : chunk;
            if (by_offs) {
                if (ring && offs_at(r1) - cp.b0 > seg_budget) {   // the largest r1 whose rows fit the budget (at least one row)
                    int64_t lo = r0 + 1, hi = r1;
                    while (lo < hi) {
                        const int64_t mid = lo + (hi - lo + 1) / 2;
                        if (offs_at(mid) - cp.b0 <= seg_budget) lo = mid;
                        else hi = mid - 1;
                    }
                    r1 = lo;
                }
                cp.b1 = offs_at(r1);
            } else {
                int64_t s = 0, i = r0;
                if (index_format == SCT_IDX_LEN8) lens_until((const uint8_t *)index, r0, r1, ring ? seg_budget : INT64_MAX, i, s);
                else lens_until((const uint16_t *)index, r0, r1, ring ? seg_budget : INT64_MAX, i, s);
                r1 = i;
                cp.b1 = cp.b0 + s;
            }
            cp.r1 = r1;
            if (cp.b0 < 0 || cp.b1 < cp.b0 || cp.b1 > seg_bytes)
                return fail(SCT_E_ARG, "the row index runs outside segs (rows %lld..%lld: bytes %lld..%lld of %lld)", (long long)cp.r0, (long long)cp.r1,
                            (long long)cp.b0, (long long)cp.b1, (long long)seg_bytes);
            if (cp.b1 - cp.b0 >= 0xFFFFFFF0ll)
                return fail(SCT_E_ARG, "a chunk of %lld rows holds %lld bytes: lower the chunk_rows option", (long long)(cp.r1 - cp.r0), (long long)(cp.b1 - cp.b0));
            plan.push_back(cp);
            r0 = r1;
            b0 = cp.b1;
        }
        if (!by_offs && b0 != seg_bytes)
            return fail(SCT_E_ARG, "the row lengths add up to %lld but segs holds %lld bytes", (long long)b0, (long long)seg_bytes);
    }
    n_chunks = (int64_t)plan.size();
    return SCT_OK;
}

ChunkIO StreamCall::caller_io(int64_t k) const {
    const ChunkPlan &cp = plan[(size_t)k];
    ChunkIO io;
    io.segs_src = segs + cp.b0;
    io.segs_dev = zc_segs ? zc_segs + cp.b0 : nullptr;
    io.idx_src = (const char *)index + isz * (size_t)cp.r0;
    io.idx_dev = zc_index ? (const void *)(zc_index + isz * (size_t)cp.r0) : nullptr;
    io.idx_format = index_format;
    io.min_e_dst = min_e + cp.r0;
    io.min_e_dev = zc_min_e ? zc_min_e + cp.r0 : nullptr;
    const size_t csz = count_format == SCT_CNT_U8 ? 1 : 4;
    if (count_format != SCT_CNT_OFFS64) {
        io.cnt_dst = (char *)counts + csz * (size_t)cp.r0;
        io.cnt_dev = zc_counts ? (void *)((char *)zc_counts + csz * (size_t)cp.r0) : nullptr;
    }
    return io;
}

// phase A of chunk k: everything that can be queued without knowing any result
int StreamCall::phase_a(int64_t k, const ChunkIO &io) {
    Lane &l = lane_of(k);
    ChunkPlan &cp = plan[(size_t)k];
    const int64_t m = cp.r1 - cp.r0;
    const int64_t bytes = cp.b1 - cp.b0;
    l.bp = bp;
    // the ring route's slots take one-byte counts whatever the caller asked for (its way out widens them)
    const bool slot_counts = ring && count_format != SCT_CNT_U8 && !zc_counts;
    if (slot_counts) l.bp.count_format = SCT_CNT_U8;
    l.n_seg = m;
    l.seg_bytes = bytes;
    l.wide = false;
    l.have_offs64 = false;
    if (int rc_ = lane_ensure(l, m, l.bp)) return rc_;
    l.cnt8_soft = slot_counts;
    if (io.min_e_dev) {
        l.min_e_ptr = io.min_e_dev;
        l.direct_min_e = true;
    }
    if (io.cnt_dev) {
        if (l.bp.count_format == SCT_CNT_U8) l.cnt8_ptr = (uint8_t *)io.cnt_dev;
        else l.cnt32_ptr = (uint32_t *)io.cnt_dev;
        l.direct_cnt = true;
    }
    if (zc_hits) {
        l.hits_ptr = zc_hits;
        l.hits_cap = c->h_hits_cap;
        l.direct_hits = true;
    }
    if (hits16 && n_chunks == 1) {   // the only chunk: its hits start at 0, the fill kernel can write them narrowed
        l.hits16_ptr = zc_hits16;
        l.hits_cap = 2 * (uint64_t)c->h_hits_cap;
        cp.narrow_cap = l.hits_cap;
    }
    const int ifmt = io.idx_format;
    const size_t esz = index_entry_size(ifmt);
    const bool offs_index = ifmt == SCT_IDX_OFFS64 || ifmt == SCT_IDX_OFFS32 || ifmt == SCT_IDX_OFFSU32_INTERNAL;
    if (l.offs32.ensure(4 * (size_t)(m + 1))) return g_code;
    if (!io.segs_dev && l.segs.ensure((size_t)bytes + 64)) return g_code;
    if ((!io.idx_dev || !offs_index) && l.index_in.ensure(esz * (size_t)(m + 1) + 16)) return g_code;
    hipStream_t st = l.stream;
    if (int rc_ = reset_counters(l)) return rc_;
    const void *idx_dev = io.idx_dev ? io.idx_dev : l.index_in.p;
    if (io.segs_dev) {
        l.segs_ptr = io.segs_dev;
        l.host_segs = io.segs_src;
    } else {
        l.segs_ptr = l.segs.as<uint8_t>();
        l.host_segs = nullptr;
    }
    if (!io.segs_dev || !io.idx_dev) {
        // (page-locked memory: queued; anything else goes through the context's bounce pieces — copy_up — and has been read on return)
        Timer t(c, l, "copy_h2d");
        if (!io.segs_dev && bytes)
            if (int rc_ = copy_up(c, l.segs.p, io.segs_src, (size_t)bytes, st)) return rc_;
        if (!io.idx_dev)
            if (int rc_ = copy_up(c, l.index_in.p, io.idx_src, esz * (size_t)(offs_index ? m + 1 : m), st)) return rc_;
    }
    {
        Timer t(c, l, "k_index");
        unsigned long long *bad = reinterpret_cast<unsigned long long *>(l.ctr() + CTR_BAD_ROW);
        if (ifmt == SCT_IDX_OFFS64)
            k_index_offs<int64_t><<<blocks_for(m + 1, 256), 256, 0, st>>>((const int64_t *)idx_dev, m, bytes, l.offs32.as<uint32_t>(), bad);
        else if (ifmt == SCT_IDX_OFFS32)
            k_index_offs<int32_t><<<blocks_for(m + 1, 256), 256, 0, st>>>((const int32_t *)idx_dev, m, bytes, l.offs32.as<uint32_t>(), bad);
        else if (ifmt == SCT_IDX_OFFSU32_INTERNAL)
            k_index_offs<uint32_t><<<blocks_for(m + 1, 256), 256, 0, st>>>((const uint32_t *)idx_dev, m, bytes, l.offs32.as<uint32_t>(), bad);
        else {
            const int64_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
            uint64_t *total = reinterpret_cast<uint64_t *>(l.ctr() + CTR_TOTAL);  // free until the hit counts are summed
            // a zero-copy index crosses the link once: the first pass leaves a copy on the device for the second.
            // (Tried in round 3: the index in two parts, the head's filter launch beside the rest's lengths on a second stream.
            // Nothing gained, 4.89 -> 5.04 ms per 10 M rows: the lengths are link traffic like the segments — 10 of the
            // 219 MB a 10 M-row call reads — and the link is busy from the call's first microsecond to the start of the tail.)
            if (io.idx_sums_dev) {
                // (one pass: the lengths cross the link once, inside k_index_lens)
                if (ifmt == SCT_IDX_LEN8)
                    k_index_lens<uint8_t><<<(unsigned)nb, 256, 0, st>>>((const uint8_t *)idx_dev, m, io.idx_sums_dev, io.idx_sums_dev + nb, bytes, l.offs32.as<uint32_t>(), bad);
                else
                    k_index_lens<uint16_t><<<(unsigned)nb, 256, 0, st>>>((const uint16_t *)idx_dev, m, io.idx_sums_dev, io.idx_sums_dev + nb, bytes, l.offs32.as<uint32_t>(), bad);
            } else if (ifmt == SCT_IDX_LEN8) {
                uint8_t *keep = io.idx_dev ? l.index_in.as<uint8_t>() : nullptr;
                k_sum_blocks<uint8_t><<<(unsigned)nb, 256, 0, st>>>((const uint8_t *)idx_dev, m, l.sums.as<uint64_t>(), keep);
                k_scan_sums<<<1, 256, 0, st>>>(l.sums.as<uint64_t>(), nb, total);
                k_index_lens<uint8_t><<<(unsigned)nb, 256, 0, st>>>(keep ? keep : (const uint8_t *)idx_dev, m, l.sums.as<uint64_t>(), total, bytes,
                                                                   l.offs32.as<uint32_t>(), bad);
            } else {
                uint16_t *keep = io.idx_dev ? l.index_in.as<uint16_t>() : nullptr;
                k_sum_blocks<uint16_t><<<(unsigned)nb, 256, 0, st>>>((const uint16_t *)idx_dev, m, l.sums.as<uint64_t>(), keep);
                k_scan_sums<<<1, 256, 0, st>>>(l.sums.as<uint64_t>(), nb, total);
                k_index_lens<uint16_t><<<(unsigned)nb, 256, 0, st>>>(keep ? keep : (const uint16_t *)idx_dev, m, l.sums.as<uint64_t>(), total, bytes,
                                                                    l.offs32.as<uint32_t>(), bad);
            }
        }
    }
    l.index_just_timed = c->timing;   // the span "k_index" was the last thing queued: "k_filter" starts where it ended
    l.record_filters_done = c->chunk_order && !ring && n_chunks > 1;
    l.wait_for = (l.record_filters_done && k > 0) ? lane_of(k - 1).filters_done : nullptr;
    if (l.wait_for) l.index_just_timed = false;   // (a wait sits between the two spans)
    if (int rc_ = queue_batch(c, l)) return rc_;
    if (ring) {
        // the way out seeds its prefix sums with the device's block sums (hits in front of every SCAN_BLOCK rows of the chunk)
        RingSlot &s = *rc[(size_t)k].slot;
        const int64_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
        if (slot_counts && count_format == SCT_CNT_OFFS64 && nb > 0)
            HIPCHK(hipMemcpyAsync(s.base + s.o_sums, l.sums.p, 8 * (size_t)nb, hipMemcpyDeviceToHost, st));
    } else if ((!l.direct_min_e && pl_min_e) || (!l.direct_cnt && count_format != SCT_CNT_OFFS64 && pl_counts)) {
        // results into page-locked arrays of the caller's: queued behind the kernels (other arrays: phase B, through the bounce pieces)
        Timer t(c, l, "copy_d2h");
        if (!l.direct_min_e && pl_min_e) HIPCHK(hipMemcpyAsync(io.min_e_dst, l.min_e.p, (size_t)m, hipMemcpyDeviceToHost, st));
        if (!l.direct_cnt && count_format != SCT_CNT_OFFS64 && pl_counts)
            HIPCHK(hipMemcpyAsync(io.cnt_dst, l.cnt_out.p, (count_format == SCT_CNT_U8 ? 1 : 4) * (size_t)m, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipEventRecord(l.done, st));
    return SCT_OK;
}

void StreamCall::sink_flush_lane(int lane) {
    HitSpan &hs = lane_hits[(size_t)lane];
    if (sink && hs.n > 0) {
        hit_copies.push_back(ParCopy{(char *)(sink + hs.base), (const char *)(c->h_hits + hs.base), 4 * (size_t)hs.n});
        const size_t bytes = 4 * (size_t)hs.n;
        size_t parts = bytes >> 19;
        if (parts > (size_t)c->pool->threads() + 1) parts = (size_t)c->pool->threads() + 1;
        c->pool->submit(&hit_group, par_copy_part, &hit_copies.back(), (int)(parts < 1 ? 1 : parts));
    }
    hs.n = 0;
}

// the result buffer is about to move (it grew): what was copied out so far came from the old one — finish, and leave the rest to
// the caller's one copy at the end
void StreamCall::sink_drop() {
    if (!sink) return;
    c->pool->wait(&hit_group);
    g_hit_arrays.put(sink);
    sink = nullptr;
}

// phase B of chunk k: wait for phase A, look at its counters, bring the hits home
int StreamCall::phase_b(int64_t k) {
    Lane &l = lane_of(k);
    ChunkPlan &cp = plan[(size_t)k];
    const int64_t m = cp.r1 - cp.r0;
    hipStream_t st = l.stream;
    if (trace && ring) rc[(size_t)k].t_wait = since_begin();
    HIPCHK(hipEventSynchronize(l.done));
    if (trace && ring) rc[(size_t)k].t_device_done = since_begin();
    if (sink) sink_flush_lane((int)(k % L));   // (the event behind this lane's previous chunk has passed: its hits have landed)
    bool slow = false, redone = false;
    if (int rc_ = finish_batch(c, l, cp.r0, &slow, &redone)) return rc_;
    cp.hit_base = hit_base;
    if (sink && (size_t)(hit_base + l.n_hits + 1) > c->h_hits_cap) sink_drop();
    if (int rc_ = ensure_host_hits(c, (size_t)(hit_base + l.n_hits + 1))) return rc_;
    {
        Timer t(c, l, "copy_d2h");
        if (!ring) {
            const ChunkIO io = caller_io(k);
            // what phase A copied (into page-locked arrays) was not final if the slow path ran; arrays that are not page-locked get
            // theirs here, through the context's bounce pieces (copy_down: back when the bytes have arrived)
            if (!l.direct_min_e && (slow || redone || !pl_min_e))
                if (int rc_ = copy_down(c, io.min_e_dst, l.min_e.p, (size_t)m, st)) return rc_;
            if (!l.direct_cnt && count_format != SCT_CNT_OFFS64 && (slow || redone || !pl_counts))
                if (int rc_ = copy_down(c, io.cnt_dst, l.cnt_out.p, (count_format == SCT_CNT_U8 ? 1 : 4) * (size_t)m, st)) return rc_;
            if (count_format == SCT_CNT_OFFS64) {
                // the chunk's offsets start at 0: add the hits of the chunks before it, then copy (the last entry of a chunk
                // and the first of the next are the same element of the caller's array, with the same value)
                if (hit_base) k_rebase<<<blocks_for(m + 1, 256), 256, 0, st>>>(l.hit_offs.as<int64_t>(), m + 1, hit_base);
                if (int rc_ = copy_down(c, (int64_t *)counts + cp.r0, l.hit_offs.p, 8 * (size_t)(m + 1), st)) return rc_;
            }
        }
        if (hits16) {
            // (the buffer was sized for uint32 entries: twice what the narrowed lists need)
            if (l.n_hits && (n_chunks > 1 || slow || (uint64_t)l.n_hits > cp.narrow_cap)) {
                if (l.hits16_ptr && !slow) {
                    // the fill kernel found the result buffer too small and wrote nothing: fill the 32-bit array on the device first
                    if (l.hits.ensure(4 * (size_t)(l.n_hits + 1))) return g_code;
                    k_fill<<<blocks_for(m, 256), 256, 0, st>>>(l.cnt.as<uint32_t>(), l.slots.as<uint32_t>(), l.slots_retry.as<uint32_t>(),
                                                               l.wave_base.as<uint64_t>(), l.slot_off.as<uint64_t>(), l.hit_offs.as<int64_t>(), l.cnt.as<uint32_t>(), m,
                                                               l.hits.as<uint32_t>(), l.hits.cap / 4, l.ctr(), nullptr);
                    l.hits_ptr = l.hits.as<uint32_t>();
                }
                // (ensure_host_hits above may have moved the buffer: ask for its device view again)
                uint16_t *out16 = (uint16_t *)host_hits_device_view(c);
                if (!out16) return fail(SCT_E_STATE, "the page-locked result buffer is not mapped into the device");
                k_narrow_hits<<<blocks_for(l.n_hits, 2048), 256, 0, st>>>(l.hits_ptr, l.ctr(), ~0ull, out16 + hit_base);
            }
        } else if (l.n_hits && !l.direct_hits) {
            HIPCHK(hipMemcpyAsync(c->h_hits + hit_base, l.hits_ptr, 4 * (size_t)l.n_hits, hipMemcpyDeviceToHost, st));
        }
    }
    if (sink) {
        if ((size_t)(hit_base + l.n_hits) > sink_entries) sink_drop();
        else lane_hits[(size_t)(k % L)] = HitSpan{hit_base, l.n_hits};
    }
    if (ring)
        if (int rc_ = ring_after_b(k, slow || redone)) return rc_;
    hit_base += l.n_hits;
    for (int i = 0; i < 6; i++) c->stats[i] += l.stats[i];
    return SCT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// the ring route's host side
// ------------------------------------------------------------------------------------------------------------
// the conversions on the way in and on the way out (sct_hostcopy.c: plain C, a clone per instruction set)
extern "C" {
uint64_t sct_narrow_offs64(const int64_t *offs, int64_t cnt, uint8_t *out);   // -> one-byte lengths; returns the OR of all differences
uint64_t sct_narrow_offs32(const int32_t *offs, int64_t cnt, uint8_t *out);
void sct_counts8_to_offs64(const uint8_t *c8, int64_t n, int64_t base, int64_t *dst);
void sct_counts8_to_u32(const uint8_t *c8, int64_t n, uint32_t *dst);
void sct_rebase_offs64(const int64_t *rel, int64_t n, int64_t base, int64_t *dst);
void sct_offs64_to_u32(const int64_t *rel, int64_t n, uint32_t *dst);
}


// part `part` of chunk k's way in: its share of the segment bytes and of the index
static void ring_copy_in(void *arg, int part, int parts) {
    RingChunk *r = (RingChunk *)arg;
    StreamCall *sc = r->call;
    const int64_t t0 = clk_ns();
    const ChunkPlan &cp = sc->plan[(size_t)r->k];
    RingSlot &s = *r->slot;
    const int64_t bytes = cp.b1 - cp.b0, m = cp.r1 - cp.r0;
    const int64_t a = (bytes * part / parts) & ~(int64_t)63, b = part + 1 == parts ? bytes : ((bytes * (part + 1) / parts) & ~(int64_t)63);
    if (b > a) memcpy(s.base + a, sc->segs + cp.b0 + a, (size_t)(b - a));
    const int64_t ra = m * part / parts, rb = m * (part + 1) / parts;
    if (rb > ra) {
        uint8_t *idx = (uint8_t *)(s.base + s.o_idx);
        uint64_t seen = 0;
        switch (sc->index_format) {
        case SCT_IDX_OFFS64: seen = sct_narrow_offs64((const int64_t *)sc->index + cp.r0 + ra, rb - ra, idx + ra); break;
        case SCT_IDX_OFFS32: seen = sct_narrow_offs32((const int32_t *)sc->index + cp.r0 + ra, rb - ra, idx + ra); break;
        case SCT_IDX_LEN8: memcpy(idx + ra, (const uint8_t *)sc->index + cp.r0 + ra, (size_t)(rb - ra)); break;
        default: memcpy(idx + 2 * ra, (const uint16_t *)sc->index + cp.r0 + ra, 2 * (size_t)(rb - ra)); break;
        }
        if (seen >> 8) r->len_bits.fetch_or(seen, std::memory_order_relaxed);
    }
    const int64_t t1 = clk_ns();
    sc->ns_in.fetch_add(t1 - t0, std::memory_order_relaxed);
    sc->note(r->in_first, r->in_last, t0, t1);
}

// part `part` of chunk k's way out: the slot's min_e and counts into the caller's arrays.  Parts are cut at the device's scan
// blocks: the hits in front of a block are known (block sums), so every part starts its prefix sum by itself.
static void ring_copy_out(void *arg, int part, int parts) {
    RingChunk *r = (RingChunk *)arg;
    StreamCall *sc = r->call;
    const int64_t t0 = clk_ns();
    const ChunkPlan &cp = sc->plan[(size_t)r->k];
    const RingSlot &s = *r->slot;
    const int64_t m = cp.r1 - cp.r0;
    const int64_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const int64_t ba = nb * part / parts, bb = nb * (part + 1) / parts;
    const int64_t ra = ba * SCAN_BLOCK, rb = bb * SCAN_BLOCK < m ? bb * SCAN_BLOCK : m;
    if (rb > ra) {
        if (!sc->zc_min_e) memcpy(sc->min_e + cp.r0 + ra, s.base + s.o_min + ra, (size_t)(rb - ra));
        if (!sc->zc_counts && sc->count_format != SCT_CNT_U8) {
            const uint8_t *c8 = (const uint8_t *)(s.base + s.o_cnt);
            const int64_t *wo = r->wide_offs;
            if (sc->count_format == SCT_CNT_OFFS64) {
                int64_t *dst = (int64_t *)sc->counts + cp.r0;
                if (wo) sct_rebase_offs64(wo + ra, rb - ra, r->hit_base, dst + ra);
                else sct_counts8_to_offs64(c8 + ra, rb - ra, r->hit_base + (int64_t)((const uint64_t *)(s.base + s.o_sums))[ba], dst + ra);
                if (rb == m) dst[m] = r->hit_base + r->n_hits;
            } else {   // SCT_CNT_U32
                uint32_t *dst = (uint32_t *)sc->counts + cp.r0;
                if (wo) sct_offs64_to_u32(wo + ra, rb - ra, dst + ra);
                else sct_counts8_to_u32(c8 + ra, rb - ra, dst + ra);
            }
        } else if (!sc->zc_counts) {
            memcpy((uint8_t *)sc->counts + cp.r0 + ra, s.base + s.o_cnt + ra, (size_t)(rb - ra));
        }
    }
    const int64_t t1 = clk_ns();
    sc->ns_out.fetch_add(t1 - t0, std::memory_order_relaxed);
    sc->note(r->out_first, r->out_last, t0, t1);
}

int StreamCall::ring_prepare() {
    int want = c->ring_slots > 0 ? c->ring_slots : L + 2;
    if (want < L + 1) want = L + 1;
    S = (int)(n_chunks < want ? n_chunks : want);
    if ((int)c->ring.size() < S) c->ring.resize((size_t)S);
    int64_t rows_max = 0, bytes_max = 0;
    for (const ChunkPlan &cp : plan) {
        rows_max = cp.r1 - cp.r0 > rows_max ? cp.r1 - cp.r0 : rows_max;
        bytes_max = cp.b1 - cp.b0 > bytes_max ? cp.b1 - cp.b0 : bytes_max;
    }
    for (int i = 0; i < S; i++)
        if (int rc_ = ring_slot_ensure(c->ring[(size_t)i], (size_t)bytes_max, rows_max)) return rc_;
    ensure_pool(c);
    rc.reset(new RingChunk[(size_t)n_chunks]);
    for (int64_t k = 0; k < n_chunks; k++) {
        RingChunk &r = rc[(size_t)k];
        r.call = this;
        r.k = k;
        r.slot = &c->ring[(size_t)(k % S)];
        if (k >= S) {
            rc[(size_t)(k - S)].next_user = &r;
            c->pool->hold(&r.in);      // (queued by chunk k - S's way out: ring_after_b)
        }
        r.idx_format = index_format == SCT_IDX_LEN16 ? SCT_IDX_LEN16 : SCT_IDX_LEN8;
    }
    return SCT_OK;
}

static inline int copy_parts(sct_ctx *c, int64_t bytes) {
    int64_t p = bytes >> 18;   // pieces of a quarter of a megabyte and more
    const int64_t most = (int64_t)c->pool->threads() + 1;
    p = p > most ? most : p;
    return (int)(p < 1 ? 1 : p);
}

void StreamCall::ring_submit_in(int64_t k) {
    RingChunk &r = rc[(size_t)k];
    const ChunkPlan &cp = plan[(size_t)k];
    c->pool->submit(&r.in, ring_copy_in, &r, copy_parts(c, (cp.b1 - cp.b0) + (int64_t)isz * (cp.r1 - cp.r0)));
}

// run by the thread that finishes the last part of a chunk's way out: the slot is free — its next user's way in may start
static void ring_slot_handover(void *arg) {
    RingChunk *next = (RingChunk *)arg;
    StreamCall *sc = next->call;
    sc->ring_submit_in(next->k);
    sc->c->pool->release(&next->in);   // (held open since ring_prepare; nothing of the call is touched after this line)
}

// the chunk is in its slot; rows whose length does not fit one byte: the index once more, wider (by this thread: rare)
int StreamCall::ring_wait_in(int64_t k) {
    RingChunk &r = rc[(size_t)k];
    c->pool->wait(&r.in);
    if (!by_offs) return SCT_OK;
    const ChunkPlan &cp = plan[(size_t)k];
    const int64_t m = cp.r1 - cp.r0;
    {   // bytes in front of every SCAN_BLOCK rows of the chunk, and the total: the device's scan of the lengths, read off the offsets
        uint64_t *isums = (uint64_t *)(r.slot->base + r.slot->o_isums);
        const int64_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
        for (int64_t b = 0; b < nb; b++) isums[b] = (uint64_t)(offs_at(cp.r0 + b * SCAN_BLOCK) - cp.b0);
        isums[nb] = (uint64_t)(cp.b1 - cp.b0);
        r.have_isums = true;
    }
    const uint64_t bits = r.len_bits.load();
    if (!(bits >> 8)) return SCT_OK;
    int64_t mx = 0;
    for (int64_t i = cp.r0; i < cp.r1; i++) {
        const int64_t d = offs_at(i + 1) - offs_at(i);
        if (d < 0) return fail(SCT_E_ARG, "the row index must not decrease and must end at the size of segs (row %lld)", (long long)i);
        mx = d > mx ? d : mx;
    }
    RingSlot &s = *r.slot;
    if (mx > 65535) {
        uint32_t *o = (uint32_t *)(s.base + s.o_idx);
        for (int64_t i = 0; i <= m; i++) o[i] = (uint32_t)(offs_at(cp.r0 + i) - cp.b0);
        r.idx_format = SCT_IDX_OFFSU32_INTERNAL;
    } else if (mx > 255) {
        uint16_t *o = (uint16_t *)(s.base + s.o_idx);
        for (int64_t i = 0; i < m; i++) o[i] = (uint16_t)(offs_at(cp.r0 + i + 1) - offs_at(cp.r0 + i));
        r.idx_format = SCT_IDX_LEN16;
    }
    return SCT_OK;
}

ChunkIO StreamCall::ring_io(int64_t k) const {
    const ChunkPlan &cp = plan[(size_t)k];
    const RingChunk &r = rc[(size_t)k];
    const RingSlot &s = *r.slot;
    ChunkIO io;
    io.segs_src = s.base;
    io.segs_dev = (const uint8_t *)s.dev;
    io.idx_src = s.base + s.o_idx;
    io.idx_dev = s.dev + s.o_idx;
    io.idx_format = r.idx_format;
    io.idx_sums_dev = r.have_isums ? (const uint64_t *)(s.dev + s.o_isums) : nullptr;
    io.min_e_dev = zc_min_e ? zc_min_e + cp.r0 : (uint8_t *)(s.dev + s.o_min);
    if (zc_counts) io.cnt_dev = (char *)zc_counts + (count_format == SCT_CNT_U8 ? 1 : 4) * (size_t)cp.r0;
    else io.cnt_dev = s.dev + s.o_cnt;
    return io;
}

// chunk k's results are complete in its slot (phase B waited for them): send them on their way out, and the slot's next user
// on its way in behind them
int StreamCall::ring_after_b(int64_t k, bool stale) {
    Lane &l = lane_of(k);
    RingChunk &r = rc[(size_t)k];
    const ChunkPlan &cp = plan[(size_t)k];
    const int64_t m = cp.r1 - cp.r0;
    RingSlot &s = *r.slot;
    r.hit_base = hit_base;
    r.n_hits = l.n_hits;
    const bool slot_counts = count_format != SCT_CNT_U8 && !zc_counts;
    bool sync_out = false;
    if (slot_counts && l.h_ctr[CTR_CNT_OVERFLOW]) {
        // a row with more than 255 hits: the slot's one-byte counts are no use; the chunk's 64-bit offsets come down instead
        if (int rc_ = ensure_scratch(c, 8 * (size_t)(m + 1))) return rc_;
        HIPCHK(hipMemcpyAsync(c->h_scratch, l.hit_offs.p, 8 * (size_t)(m + 1), hipMemcpyDeviceToHost, l.stream));
        HIPCHK(hipStreamSynchronize(l.stream));
        r.wide_offs = (const int64_t *)c->h_scratch;
        sync_out = true;   // (one scratch buffer: this chunk is out before the next one may need it)
    } else if (slot_counts && stale && count_format == SCT_CNT_OFFS64) {
        // the slow path (or a second pass of the batch) summed the counts again: the block sums phase A copied are stale
        const int64_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
        HIPCHK(hipMemcpyAsync(s.base + s.o_sums, l.sums.p, 8 * (size_t)nb, hipMemcpyDeviceToHost, l.stream));
        HIPCHK(hipStreamSynchronize(l.stream));
    }
    r.submitted_out = true;
    if (r.next_user) {
        r.out.then = ring_slot_handover;
        r.out.then_arg = r.next_user;
    }
    const int64_t per_row = (zc_min_e ? 0 : 1) + (zc_counts ? 0 : count_format == SCT_CNT_OFFS64 ? 8 : count_format == SCT_CNT_U32 ? 4 : 1);
    int parts = copy_parts(c, per_row * m);
    const int64_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (parts > nb) parts = (int)(nb < 1 ? 1 : nb);
    c->pool->submit(&r.out, ring_copy_out, &r, parts);
    if (trace) r.t_out_submitted = since_begin();
    if (sync_out) c->pool->wait(&r.out);
    return SCT_OK;
}

// every copy of this call that was handed to the pool has finished (they read and write the caller's arrays and this object)
void StreamCall::ring_quiesce() {
    if (!rc || !c->pool) return;
    for (int64_t k = 0; k < n_chunks; k++) {
        RingChunk &r = rc[(size_t)k];
        // a chunk's way in was queued by this thread (the first S chunks) or will be by its slot's previous user's way out, if that
        // was queued; what was never queued — an error cut the call short — is not waited for
        if (k < S && first_in_submitted > k) c->pool->wait(&r.in);
        if (r.submitted_out) {
            c->pool->wait(&r.out);
            if (r.next_user) c->pool->wait(&r.next_user->in);
        }
    }
}

int StreamCall::run() {
    if (int rc_ = choose_route()) return rc_;
    if (int rc_ = make_plan()) return rc_;
    // chunks in flight on the device: staged copies want several (upload, kernels and download of different chunks overlap); the
    // ring's chunks read their rows over the link themselves, and three in flight keep it as busy as eight (tools/pageable_probe.py)
    const int lanes_wanted = ring ? c->ring_lanes : c->stream_lanes;
    L = (int)(n_chunks < lanes_wanted ? n_chunks : lanes_wanted);
    if (L < 1) L = 1;
    while ((int)c->lanes.size() < L) {
        Lane *l = nullptr;
        if (int rc_ = lane_create(&l)) return rc_;
        c->lanes.push_back(l);
    }
    c->last_ms.clear();
    memset(c->stats, 0, sizeof(c->stats));
    for (int i = 0; i < L; i++) {
        c->lanes[(size_t)i]->timed.clear();
        c->lanes[(size_t)i]->pool_used = 0;
        c->lanes[(size_t)i]->last_end = nullptr;
        c->lanes[(size_t)i]->index_just_timed = false;
    }
    if (int rc_ = ensure_host_hits(c, (size_t)(n < (1 << 18) ? (1 << 18) : n + n / 4))) return rc_;
    // a call of one chunk lets the fill kernel write the hits into the page-locked result buffer itself
    zc_hits = (n_chunks == 1 && !hits16 && c->zero_copy) ? host_hits_device_view(c) : nullptr;
    // 16-bit hits: the lists are assembled on the device as always and a last kernel writes them, narrowed, into the result buffer
    zc_hits16 = hits16 ? (uint16_t *)host_hits_device_view(c) : nullptr;
    if (hits16 && !zc_hits16) return fail(SCT_E_STATE, "the page-locked result buffer is not mapped into the device");
    trace = ring && getenv("SCT_TRACE_RING") != nullptr;
    t_begin_ns = clk_ns();
    if (ring)
        if (int rc_ = ring_prepare()) return rc_;
    const double t_prepared = trace ? since_begin() : 0;
    if (want_sink && !hits16) {
        ensure_pool(c);
        lane_hits.assign((size_t)L, HitSpan{});
        sink_entries = c->h_hits_cap;
        sink = inject_alloc_failure() ? nullptr : g_hit_arrays.get(4 * sink_entries);   // (none to be had: the caller makes do with one copy at the end)
    }

    // host time spent queueing (phase A; on the ring route: waiting for a chunk's way in, too) and waiting + collecting (phase B),
    // reported beside the device spans
    double host_a = 0, host_b = 0;
    int rc_all = SCT_OK;
    if (ring)
        for (int64_t k = 0; k < S && k < n_chunks; k++) {
            ring_submit_in(k);
            first_in_submitted = k + 1;
        }
    for (int64_t k = 0; k < n_chunks + (L - 1) && rc_all == SCT_OK; k++) {
        const double t0 = clk_ms();
        if (k < n_chunks) {
            if (ring) {
                rc_all = ring_wait_in(k);
                if (trace) rc[(size_t)k].t_in_ready = since_begin();
                if (rc_all == SCT_OK) rc_all = phase_a(k, ring_io(k));
                if (trace) rc[(size_t)k].t_queued = since_begin();
            } else {
                rc_all = phase_a(k, caller_io(k));
            }
        }
        const double t1 = clk_ms();
        const int64_t j = k - (L - 1);
        if (rc_all == SCT_OK && j >= 0) rc_all = phase_b(j);
        host_a += t1 - t0;
        host_b += clk_ms() - t1;
    }
    if (rc_all != SCT_OK) {
        const std::string keep = g_err;
        sink_drop();
        ring_quiesce();
        for (Lane *l : c->lanes) (void)hipStreamSynchronize(l->stream);
        (void)hipGetLastError();
        g_err = keep;
        return rc_all;
    }
    const double t2 = clk_ms();
    ring_quiesce();
    for (int i = 0; i < L; i++) HIPCHK(hipStreamSynchronize(c->lanes[(size_t)i]->stream));
    HIPCHK(hipGetLastError());
    if (sink) {   // every lane's last hits have landed now
        for (int i = 0; i < L; i++) sink_flush_lane(i);
        c->pool->wait(&hit_group);
    }
    host_b += clk_ms() - t2;
    for (int i = 0; i < L; i++) collect_timing(c, *c->lanes[(size_t)i]);
    c->last_ms.emplace_back("host_queue", sct_ctx::Span{host_a, (int)n_chunks});
    c->last_ms.emplace_back("host_wait", sct_ctx::Span{host_b, (int)n_chunks});
    if (ring) {
        c->last_ms.emplace_back("ring_copy_in", sct_ctx::Span{(double)ns_in.load() * 1e-6, (int)n_chunks});
        c->last_ms.emplace_back("ring_copy_out", sct_ctx::Span{(double)ns_out.load() * 1e-6, (int)n_chunks});
    }
    if (trace) {
        fprintf(stderr, "sctagger_hip: ring call of %lld rows in %lld chunks, %d lanes, %d slots, %d copy threads; ring ready at %.3f ms, call done at %.3f ms\n",
                (long long)n, (long long)n_chunks, L, S, c->pool->threads(), t_prepared, since_begin());
        for (int64_t k = 0; k < n_chunks; k++) {
            const RingChunk &r = rc[(size_t)k];
            fprintf(stderr, "  chunk %3lld rows %8lld: in %.3f..%.3f ready %.3f queued %.3f | wait from %.3f device done %.3f | out submitted %.3f ran %.3f..%.3f\n", (long long)k,
                    (long long)(plan[(size_t)k].r1 - plan[(size_t)k].r0), (double)(r.in_first.load() - t_begin_ns) * 1e-6, (double)(r.in_last.load() - t_begin_ns) * 1e-6,
                    r.t_in_ready, r.t_queued, r.t_wait, r.t_device_done, r.t_out_submitted, (double)(r.out_first.load() - t_begin_ns) * 1e-6,
                    (double)(r.out_last.load() - t_begin_ns) * 1e-6);
        }
    }
    c->last_route = ring ? 2 : zc_segs ? 1 : 0;
    c->lane0().have_result = false;  // lane 0 no longer holds a resident upload's result
    c->lane0().n_seg = -1;
    return SCT_OK;
}

// sink_out != NULL (sct_match): *sink_out receives a malloc'ed array (g_hit_arrays) that holds the hits already — or NULL, and the
// caller copies them out of *hits_out itself
static int match_stream(sct_ctx *c, const char *segs, int64_t seg_bytes, const void *index, int32_t index_format, int64_t n, int32_t max_err,
                        int32_t threads, uint8_t *min_e, void *counts, int32_t count_format, int32_t hits_format, const void **hits_out,
                        int64_t *n_hits_out, uint32_t **sink_out) {
    if (!c) return fail(SCT_E_ARG, "ctx is NULL");
    FaultScope fault_scope(&c->fault);
    if (!hits_out || !n_hits_out) return fail(SCT_E_ARG, "sct_match_stream: NULL output pointer");
    *hits_out = nullptr;
    *n_hits_out = 0;
    if (n < 0 || seg_bytes < 0) return fail(SCT_E_ARG, "sct_match_stream: negative size");
    if (n > 0 && (!index || !min_e)) return fail(SCT_E_ARG, "sct_match_stream: NULL index or min_e");
    if (!counts) return fail(SCT_E_ARG, "sct_match_stream: counts is NULL");
    if (seg_bytes > 0 && !segs) return fail(SCT_E_ARG, "segs is NULL");
    if (index_format < SCT_IDX_OFFS64 || index_format > SCT_IDX_LEN16) return fail(SCT_E_ARG, "unknown index format %d", index_format);
    if (count_format < SCT_CNT_OFFS64 || count_format > SCT_CNT_U8) return fail(SCT_E_ARG, "unknown count format %d", count_format);
    if (hits_format != SCT_HITS_U32 && hits_format != SCT_HITS_U16) return fail(SCT_E_ARG, "unknown hits format %d", hits_format);
    const bool hits16 = hits_format == SCT_HITS_U16;
    if (hits16 && c->have_bc && c->np > 65536)
        return fail(SCT_E_RANGE, "%d barcodes give pattern ids beyond 16 bits: ask for SCT_HITS_U32", c->n_bc);
    HIPCHK(hipSetDevice(c->device));
    StreamCall sc{};
    sc.c = c;
    sc.segs = segs;
    sc.seg_bytes = seg_bytes;
    sc.index = index;
    sc.index_format = index_format;
    sc.n = n;
    sc.min_e = min_e;
    sc.counts = counts;
    sc.count_format = count_format;
    sc.hits16 = hits16;
    if (int rc = batch_params(c, max_err, threads, count_format, sc.bp)) return rc;
    if (index_format == SCT_IDX_OFFS64 || index_format == SCT_IDX_OFFS32) {
        if (index && sc.offs_at(0) != 0) return fail(SCT_E_ARG, "offs[0] must be 0");
        if (index && sc.offs_at(n) != seg_bytes)
            return fail(SCT_E_ARG, "offs[n] = %lld but segs holds %lld bytes", (long long)sc.offs_at(n), (long long)seg_bytes);
    }
    if (n == 0) {
        if (seg_bytes != 0) return fail(SCT_E_ARG, "no rows but %lld segment bytes", (long long)seg_bytes);
        if (count_format == SCT_CNT_OFFS64) ((int64_t *)counts)[0] = 0;
        c->last_ms.clear();
        memset(c->stats, 0, sizeof(c->stats));
        *hits_out = c->h_hits;
        return SCT_OK;
    }
    sc.want_sink = sink_out != nullptr;
    if (int rc = sc.run()) return rc;
    *hits_out = c->h_hits;
    *n_hits_out = sc.hit_base;
    if (sink_out) *sink_out = sc.sink;
    return SCT_OK;
}

extern "C" int sct_match_stream(sct_ctx *c, const char *segs, int64_t seg_bytes, const void *index, int32_t index_format, int64_t n,
                                int32_t max_err, int32_t threads, uint8_t *min_e, void *counts, int32_t count_format, int32_t hits_format,
                                const void **hits_out, int64_t *n_hits_out) {
    return match_stream(c, segs, seg_bytes, index, index_format, n, max_err, threads, min_e, counts, count_format, hits_format, hits_out, n_hits_out, nullptr);
}
