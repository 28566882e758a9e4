/*
 * sctagger_hip.h — C-ABI of libsctagger_hip.so, the MI355X (gfx950) implementation of scTagger's
 * match_trie barcode-matching stage.
 *
 * The reference has no FFI: the seam this library replaces is the in-process call
 *     run_get_matches_prefix(max_error, threads) -> dict[rid -> (e, set[(bid, strand)])]
 * (reference scTagger.py:688-716) together with the worker it fans out to
 * (get_matches_prefix :671-685, Trie :526-599, trie_search :643-668), reading the module globals
 * barcodes / barcodes_rc (:631-640) and lr_segs / lr_segs_idxs (:616-627).
 *
 * Conventions: plain pointers and sizes only; every buffer named below is caller-owned unless the
 * comment says otherwise; the library never frees caller memory; every call returns 0 or a negative
 * SCT_E_* code and the message is available from sct_last_error() (thread-local); no exception
 * crosses the boundary.  One context per device; a context is used by one host thread at a time
 * (multi-GPU = one context, and one process or thread, per device).  There is no CPU fallback:
 * without a usable HIP device sct_create fails.
 */
#ifndef SCTAGGER_HIP_H
#define SCTAGGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sct_ctx sct_ctx;

enum {
    SCT_OK = 0,
    SCT_E_ARG = -1,         /* bad argument (NULL, negative size, bl/max_err out of range, byte >= 128 where the reference maps it) */
    SCT_E_HIP = -2,         /* a HIP runtime call failed */
    SCT_E_NODEV = -3,       /* no usable device */
    SCT_E_STATE = -4,       /* call order (e.g. match before set_barcodes) */
    SCT_E_UNSUPPORTED = -5, /* barcode length above SCT_MAX_BL; a -t whose window prefix is not below bl - max_err */
    SCT_E_RANGE = -6,       /* a value does not fit the compact output format that was asked for */
    SCT_E_NOMEM = -7        /* out of device memory (hipMalloc), page-locked host memory or host heap */
};

#define SCT_MAX_BL 32       /* barcode length limit of the bit-parallel kernels */
#define SCT_NO_HIT 0xFF     /* min_e value for a segment absent from the reference's result dict */

/* Number of visible HIP devices (0 if none / runtime unusable). */
int sct_device_count(void);

/* PCI bus id of `device` ("0000:c1:00.0") into buf (len >= 16), for callers that want to place their threads and their
 * page-locked buffers on the NUMA node the device hangs off (/sys/bus/pci/devices/<id>/numa_node). */
int sct_device_pci_bus_id(int device, char *buf, int len);

/* CPUs the library sizes its host-side pool by (the copy threads of the stream seam: half of them, at most 8): the affinity mask
 * cut down to the cgroup's CPU quota (environment SCT_CPUS overrides the count), divided by LOCAL_WORLD_SIZE when several ranks
 * share the node (one process per GPU: the reference's Pool(threads), scTagger.py:703-704, sizes itself by -t alone). */
int sct_host_cpus(void);

/* Create / destroy a context bound to `device`. */
int sct_create(int device, sct_ctx **out);
void sct_destroy(sct_ctx *ctx);

/* Last error message of the calling thread ("" if none). */
const char *sct_last_error(void);

/* Prefix length k the reference derives from -t: smallest k with 4^k >= threads (scTagger.py:689-697). */
int sct_prefix_len(int threads);

/*
 * Replace the barcode set — the reference's read_sr_barcodes globals (scTagger.py:631-640).
 *   bc  n*bl ASCII bytes, forward barcodes only, row i = barcodes[i]; the reverse-complement strings
 *       are derived inside exactly as rev_compl does (:148-156: A<->T, C<->G, other bytes unchanged),
 *       and every byte is then mapped as map_char does (:512-516: ACGT -> 0..3, anything else -> 0).
 *   bl  barcode length (the reference asserts all barcodes have length -bl, :760-761), 1..SCT_MAX_BL.
 *       Lengths 8..16 with max_err <= 2 use the seed-filter kernels, everything else the exhaustive scan.
 * A byte >= 128 is rejected with SCT_E_ARG (the reference raises IndexError there).
 */
int sct_set_barcodes(sct_ctx *ctx, const char *bc, int32_t n, int32_t bl);

/*
 * The whole seam in one call: run_get_matches_prefix(max_error, threads) (scTagger.py:688-716).
 *   segs, offs   the reference's lr_segs / lr_segs_idxs (:616-627) in CSR form: segment rid is
 *                segs[offs[rid] .. offs[rid+1]); offs has n+1 non-decreasing entries, offs[0] == 0.
 *                (int64 offsets; the reference's int32 RawArray overflows at 2^31 bytes.)
 *   max_err      -mr; 0 <= max_err < bl (:143)
 *   threads      -t; only its prefix length matters: for k = sct_prefix_len(threads) > 0 a window
 *                whose first k bytes are not all in "ACGT" is not searched (raw-byte startswith, :681)
 * Outputs:
 *   min_e[n]       minimum edit distance over barcodes and strands, SCT_NO_HIT if the segment has
 *                  no barcode within max_err (such rids are absent from the reference's dict)
 *   hit_offs[n+1]  CSR offsets into *hits
 *   *hits          library-allocated (release with sct_free_hits): for each segment the sorted list of
 *                  (bid << 1 | strand), strand 1 = forward (True), 0 = reverse complement (False) —
 *                  ascending order is exactly sorted(bids) of scTagger.py:789
 *   *n_hits        total number of hits (== hit_offs[n])
 */
int sct_match(sct_ctx *ctx, const char *segs, const int64_t *offs, int64_t n, int32_t max_err, int32_t threads,
              uint8_t *min_e, int64_t *hit_offs, uint32_t **hits, int64_t *n_hits);
void sct_free_hits(sct_ctx *ctx, uint32_t *hits);

/*
 * The same seam for callers that care about the PCIe link: compact transfer formats and an engine that keeps the link busy;
 * sct_match above is this call with 64-bit offsets both ways.
 *   Page-locked buffers (sct_host_alloc, or any hipHostMalloc memory) for segs / index / min_e / counts are
 *   read and written IN PLACE by the kernels through their device mapping ("zero copy", option "zero_copy"): nothing is
 *   uploaded, the rows cross the link inside the filter kernel while it computes, and the call runs as ONE chunk of up to 2^24
 *   rows (more rows: several such chunks one after the other; a chunk also ends before 3 GiB of segment bytes, the offsets on
 *   the device being 32-bit and relative to the chunk).
 *   Ordinary (pageable) arrays — what the reference's seam hands over, scTagger.py:616-627, and what sct_match gets from numpy —
 *   go through a ring of page-locked slots OWNED BY THE CONTEXT (option "ring", default on): a pool of copy threads (option
 *   "copy_threads") moves chunk k+1 into a slot — the offsets narrowed to one byte per row on the way — while the zero-copy kernels
 *   read chunk k from its slot and chunk k-1's min_e / counts are widened into the caller's arrays.  Nothing of the caller's is
 *   page-locked, and no caller pointer is given to the HIP runtime.  6 - 7 ms per 10 M rows of 22 characters against 4.8 ms on
 *   page-locked arrays, 11 - 13 ms with the ring off (option "ring" = 0: the same memory through two 4 MB bounce pieces on the
 *   calling thread) and 20 ms through the runtime's own staged copies, which round 4 used.  Page-locked buffers with "zero_copy" = 0
 *   are uploaded in chunks of "chunk_rows" rows on "lanes" streams.
 *   Limits: row counts, seg_bytes, hit offsets and the hit total are 64-bit throughout (the reference's own c_int index,
 *   scTagger.py:618, ends at 2^31 segment bytes); sct_upload_segments takes at most 2^31 - 16 rows per call.  Tested: 20 M rows
 *   in one call at the default chunking, 4.5 GB of segment bytes in one call, and 4.43 x 10^9 hits (> 2^32; a 17.7 GB
 *   page-locked hit array) out of one call (tests/test_gpu_limits.py).
 *   Bytes >= 128: the reference raises (UnicodeDecodeError / IndexError, scTagger.py:682, :555, :512-516) when it inserts a
 *   window that contains one, and never looks at a byte no inserted window covers (:677-682: windows shorter than
 *   bl - max_err, and with -t > 1 windows whose first k raw bytes are not all in ACGT, are not inserted).  This library fails
 *   with SCT_E_ARG in exactly those cases and ignores such a byte in exactly those cases — in particular anywhere in a row of
 *   fewer than bl - max_err characters.  The same rule holds for sct_match and for sct_upload_segments + sct_run: the check
 *   is made by the kernel that reads the row.
 *   segs, seg_bytes   the reference's lr_segs (scTagger.py:616-627) and its size
 *   index             where each row lies in segs, in one of four forms (index_format):
 *       SCT_IDX_OFFS64  int64 offs[n+1]   CSR offsets, offs[0] == 0, offs[n] == seg_bytes
 *       SCT_IDX_OFFS32  int32 offs[n+1]   the same in the reference's own type (its lr_segs_idxs is a c_int array)
 *       SCT_IDX_LEN8    uint8 len[n]      row lengths (rows of at most 255 characters), sum == seg_bytes
 *       SCT_IDX_LEN16   uint16 len[n]     row lengths (at most 65 535 characters)
 *   max_err, threads  as for sct_match
 * Outputs:
 *   min_e[n]          as for sct_match
 *   counts            the number of hits of every row, in one of three forms (count_format):
 *       SCT_CNT_OFFS64  int64 hit_offs[n+1]  CSR offsets into *hits, exactly sct_match's hit_offs
 *       SCT_CNT_U32     uint32 cnt[n]        hits per row (offsets = their prefix sum, left to the caller)
 *       SCT_CNT_U8      uint8 cnt[n]         the same in one byte; fails with SCT_E_RANGE if a row has more than 255 hits
 *   *hits             all hit lists back to back in row order, (bid << 1 | strand) ascending inside a row, as
 *                     uint32 (hits_format SCT_HITS_U32) or uint16 (SCT_HITS_U16; fails with SCT_E_RANGE when the barcode
 *                     set has more than 32 768 entries).  The array is page-locked memory OWNED BY THE CONTEXT: valid
 *                     until the next sct_match_stream / sct_match / sct_destroy on it, not to be freed by the caller.
 *   *n_hits           total number of hits
 */
enum { SCT_IDX_OFFS64 = 0, SCT_IDX_OFFS32 = 1, SCT_IDX_LEN8 = 2, SCT_IDX_LEN16 = 3 };
enum { SCT_CNT_OFFS64 = 0, SCT_CNT_U32 = 1, SCT_CNT_U8 = 2 };
enum { SCT_HITS_U32 = 0, SCT_HITS_U16 = 1 };
int sct_match_stream(sct_ctx *ctx, const char *segs, int64_t seg_bytes, const void *index, int32_t index_format, int64_t n,
                     int32_t max_err, int32_t threads, uint8_t *min_e, void *counts, int32_t count_format, int32_t hits_format,
                     const void **hits, int64_t *n_hits);

/*
 * The same path in three steps, for pipelining and for measuring with the inputs resident in HBM:
 *   sct_upload_segments  host -> device copy of segs/offs (kept until the next upload); the offsets are validated here
 *                        (SCT_E_ARG), bytes >= 128 by sct_run under the rule stated for sct_match_stream
 *   sct_run              every kernel of the path on the resident segments; results stay on the device;
 *                        *n_hits receives the total hit count (one host synchronisation and one 64-byte read per run)
 *   sct_download         device -> host copy of the results of the last sct_run; hits must hold
 *                        n_hits entries
 */
int sct_upload_segments(sct_ctx *ctx, const char *segs, const int64_t *offs, int64_t n);
int sct_run(sct_ctx *ctx, int32_t max_err, int32_t threads, int64_t *n_hits);
int sct_download(sct_ctx *ctx, uint8_t *min_e, int64_t *hit_offs, uint32_t *hits);

/*
 * Page-locked host memory.  Optional: every entry point accepts ordinary (pageable) buffers — those are moved through
 * page-locked memory of the context's own by memcpy (the stream seam: a ring and copy threads; everything else: two 4 MB pieces),
 * never handed to the HIP runtime.  Buffers obtained here are read and written in place by the kernels (the stream seam) or copied
 * by DMA at the full PCIe rate.
 * sct_host_alloc returns NULL on failure (message in sct_last_error).
 */
void *sct_host_alloc(sct_ctx *ctx, size_t bytes);
void sct_host_free(sct_ctx *ctx, void *p);

/* Block until everything queued on the context's stream has finished. */
int sct_sync(sct_ctx *ctx);

/*
 * Options (key, value):
 *   "force_exhaustive" 0|1   skip the seed-filter kernel and scan every barcode for every segment
 *   "timing"           0|1   record HIP events around each kernel of sct_run (default 1)
 *   "no_thirds"        0|1   1 = never use the 20/22-bit seed families chosen for >= 2^19 patterns at max_err <= 1
 *   "thirds"          -1|0|1 -1 (default) = the thirds families are built from 2^19 patterns on, 1 = whatever the size, 0 = not built
 *                            (takes effect at the next sct_set_barcodes; "no_thirds" switches their USE off and on at any time)
 *   "fixed_buckets"   -1|0|1 thirds families: 1 / -1 (default) = a lookup reads the key's fixed bucket (one line: the rest characters of
 *                            up to 16 / 8 entries), 0 = offsets and codes (two lines).  (The main launch of the wide families always uses theirs.)
 *   "join"             0|1   thirds families at max_err = 1 with fixed buckets: 1 = the rows of up to 31 characters go through the partitioned
 *                            join (lookups as 8-byte tuples, partitioned by key range, probed table slice by table slice while the slice
 *                            sits in L2; csrc/k_join.hip.h) instead of the filter kernel's 32-bit launch; 0 (default) = the filter kernel.
 *                            Same results; about half the HBM traffic and the same time at 3.6 M barcodes.  May be
 *                            switched between calls.  A batch whose tuples do not fit their regions is redone by the filter kernel
 *                            (sct_last_stats [6]).
 *   "wide"            -1|0|1 the sixteen 20-bit seed families of max_err = 2 (sct_core.h): -1 = built and used from 2^20 patterns
 *                            (524 288 barcodes) on (default), 0 = not used, 1 = built whatever the size (takes effect at the
 *                            next sct_set_barcodes) and used
 *   "slots"           -1|0|1 8-byte bucket slots of the 16-character seed tables: -1 = used while few keys hold more
 *                            than three barcodes (3.5 % at max_err 2, 10 % below; default), 0 = never, 1 = always
 *   "chunk_rows"       n     rows per chunk of sct_match_stream / sct_match; 0 (default) = 2^24 when segs is read in place
 *                            (page-locked, "zero_copy" on), 2^20 when it is page-locked but uploaded ("zero_copy" off), "ring_rows" on the
 *                            ring route, 2^24 when it is pageable and the ring is off (staged copies do not overlap with anything)
 *   "lanes"            1..8  chunks of a stream call in flight at once, each on a stream of its own (default 5)
 *   "zero_copy"        0|1   1 (default): sct_match_stream's kernels read page-locked segs / index buffers in place over the
 *                            host link; 0: they are uploaded first
 *   "gather"           0|1   1 (default): in a zero-copy batch the rows of more than 63 characters are copied to the device
 *                            (one coalesced pass, k_special_gather) before the special launch cuts them into windows; 0: its
 *                            window lanes read them over the link.  Same results.
 *   "ring"             0|1   1 (default): a stream call whose segs are NOT page-locked goes through the context's page-locked ring
 *                            (above); 0: chunk by chunk through the context's two bounce pieces, on the calling thread.  Either way
 *                            the HIP runtime is given page-locked memory only: handed pageable memory, it page-locks the caller's
 *                            pages in place for the duration of each large copy.  (Round 4 had an option "register" that page-locked the caller's arrays
 *                            for the whole call, hipHostRegister, so that the kernels read them in place: one soak run in a few ended
 *                            in "Memory access fault by GPU" on a host address, cause never established.  Removed in round 5: the
 *                            ring is faster than it was and locks nothing of the caller's.)
 *   "ring_rows"        n     rows per chunk of the ring route; 0 (default) = 2^20, a third of the call's rows if that is less (at
 *                            least 2^17); "chunk_rows" > 0 overrides.  A slot holds 32 bytes of segments per row of a full chunk; a
 *                            chunk ends before that many bytes (a single longer row gets a slot of its own size).
 *   "ring_ramp"        0..6  r > 0: the first chunk holds 1 / 2^r of "ring_rows", the next ones double, so that the device starts early
 *                            (default 0: measured without effect at 10 M rows, a loss on small calls)
 *   "ring_lanes"       1..8  chunks of the ring route in flight on the device (default 4)
 *   "ring_slots"       n     slots of the ring; 0 (default) = "ring_lanes" + 2.  Slots are allocated when first needed, sized by the
 *                            call (powers of two), and kept: sct_get_option "ring_bytes" = page-locked bytes the ring holds
 *                            (240 MB after a 10 M-row call at the defaults; 0.2 ms per MB to allocate, once)
 *   "copy_threads"    -1|n   threads of the context's copy pool (created at the first call that needs it): -1 (default) = half of
 *                            the CPUs the process may use (affinity mask, cgroup CPU quota; environment SCT_CPUS overrides), at
 *                            most 8; 0 = the calling thread does every copy itself
 *                            Environment SCT_TRACE_STREAM=1: one line per call on stderr — the route, and which of the caller's arrays
 *                            the kernels touch in place; SCT_TRACE_RING=1: per chunk of the ring route, when its copies ran and when
 *                            the calling thread passed its stations.
 *   "need64"          -1|1   -1 (default): the 64-bit instantiations of the filter launches (rows of 32 .. 63 characters, the 63-character
 *                            windows of longer rows) are not launched until a batch of the context has needed them — the 32-bit main
 *                            launch marks it, that batch is done again with both widths, and the context launches both from then on;
 *                            1: always both.  Setting -1 starts guessing again.  sct_get_option "redone64" counts the batches done again.
 *   "need_special"    -1|1   the same for the special launch (a byte outside ACGT under the reference's prefix rule, more than 63 characters)
 *   "chunk_order"      0|1   experiment: a chunk's filter launches wait for the previous chunk's; no gain, off
 *   "fail_alloc_at"    k     TEST HOOK for the error paths (error codes instead of the reference's asserts,
 *                            scTagger.py:140-143, :760-761): the k-th allocation attempt the context makes from now on — device
 *                            memory, page-locked host memory (result buffer, ring slots, bounce pieces), a lane's stream — fails once
 *                            as if memory had run out (SCT_E_NOMEM); 0 = off.  sct_get_option "alloc_calls" counts the attempts since it was set.
 */
int sct_set_option(sct_ctx *ctx, const char *key, int64_t value);
/* Reads an option back ("join", "lanes", "chunk_rows", "zero_copy", "force_exhaustive", "fail_alloc_at", "ring", "ring_rows",
 * "ring_slots", "copy_threads") or one of the read-only values "alloc_calls" (allocation attempts since "fail_alloc_at" was last set),
 * "device_free_bytes", "device_total_bytes" (hipMemGetInfo of the context's device), "ring_bytes" (page-locked bytes the ring holds),
 * "last_route" (the last stream call: 0 = staged copies, 1 = zero copy, 2 = the ring). */
int sct_get_option(sct_ctx *ctx, const char *key, int64_t *value);

/*
 * Device time of the last sct_run / sct_match_stream / sct_match by kernel name, measured with HIP events on the stream
 * the kernels were launched on: ms[] = sum over the launches of that name, launches[] = how many there were (one per
 * chunk for a stream call).  The stream calls also report their copies as "copy_h2d" / "copy_d2h" and the HOST time
 * of the calling thread as "host_queue" (queueing the chunks) and "host_wait" (waiting for them, collecting the hits).
 * Fills up to cap entries of names[] (static strings), ms[] and launches[]; returns the number of names, or <0.
 */
int sct_last_timing(sct_ctx *ctx, const char **names, float *ms, int cap);
int sct_last_timing_ex(sct_ctx *ctx, const char **names, float *ms, int32_t *launches, int cap);

/* Counters of the last sct_run / sct_match_stream / sct_match: [0] segments on the seed-filter path, [1] segments on the exhaustive
 * path, [2] segments on the restricted-start (non-ACGT, k>0) path, [3] total hits, [4] segments the filter path
 * retried with larger lists (more than 6 or 8 verified matches, by the seed tables in use), [5] the seed scheme the filter path
 * used (SCT_SCHEME_*), [6] bits 0..31: batches whose 32-bit rows the partitioned join did (thirds scheme at max_err = 1), bits
 * 32..39: batches the join handed back to the filter kernel (a tuple region overflowed), [7] segments (counted in [0]) that the
 * seed filter took in its SPECIAL launch: rows with a byte outside ACGT while the reference's prefix length k > 0 (restricted window
 * starts, scTagger.py:681) and rows of more than 63 characters (cut into overlapping 63-character windows, up to 16 of them:
 * 753 characters at bl = 16, max_err = 2); only what that launch cannot hold reaches [1] / [2]. Fills up to cap entries; returns
 * the number of counters there are (8), or <0. */
enum {
    SCT_SCHEME_NONE = 0,            /* no seed filter: every row went to the exhaustive scan / restricted-start DP */
    SCT_SCHEME_QUARTER_SLOTS = 1,   /* six 16-bit quarter-pair families answered from 8-byte bucket slots (small barcode sets) */
    SCT_SCHEME_QUARTER_TABLES = 2,  /* the same families through offset / entry tables (larger sets, barcode lengths 8..15) */
    SCT_SCHEME_THIRDS = 3,          /* 20/22-bit families over barcode thirds (max_err <= 1, >= 2^19 patterns) */
    SCT_SCHEME_WIDE = 4             /* sixteen 20-bit families, quarter pair + half a quarter (max_err = 2, >= 2^20 patterns) */
};
int sct_last_stats(sct_ctx *ctx, int64_t *out, int cap);

/*
 * --- extract_sr_bc_from_lr (reference scTagger.py:792-844) ------------------
 * The reference builds an Aho-Corasick automaton (third-party pyahocorasick) over the whitelist and its reverse
 * complements and counts, per barcode, the segments that contain exactly one occurrence of any word.
 *
 * sct_wl_set      the dictionary of :801-809.  keys = the n whitelist entries back to back (entry i has
 *                 key_lens[i] raw bytes, 0..16; exactly what `l[:-1]` yields, :798).  Words are added in the
 *                 reference's order — all forward words with value i, then all reverse complements (rev_compl,
 *                 :148-156) with value -i — and adding an existing word replaces its value (add_word).
 * sct_wl_count    the loop of :818-826 over CSR segments: rows with skip[r] != 0 (the reference's p == 'NA') are
 *                 ignored; every occurrence of every word counts as a hit; rows with exactly one hit increment
 *                 counts[|value|].  first_seen[b] = first row that incremented b, -1 if none (the reference's
 *                 Counter keeps insertion order, which breaks ties in its sort, :828).  counts / first_seen hold
 *                 n (whitelist size) entries.
 */
int sct_wl_set(sct_ctx *ctx, const char *keys, const int32_t *key_lens, int32_t n);
int sct_wl_count(sct_ctx *ctx, const char *segs, const int64_t *offs, const uint8_t *skip, int64_t n, int64_t *counts,
                 int64_t *first_seen);

/*
 * --- extract_lr_bc (reference scTagger.py:176-196) ----------------------------
 * The reference aligns the short-read adapter and its reverse complement to every long read with the third-party
 * edlib (`edlib.align(adapter, read, 'HW', 'locations')`) and keeps the strictly better strand.
 *
 * sct_adapter_align   adapter: m raw bytes (1..32); reads/offs: CSR of the n reads (bytes < 128).  Per read:
 *     strand[r]  0 = both strands equally good (the reference's 'NA'), 1 = '+', 2 = '-'
 *     dist[r]    the edit distance of the winning strand, -1 when strand is 0
 *     locations  CSR loc_offs[n+1] into *locs (library-allocated, release with sct_free_locs): for '+' every
 *                end position (0-based, inclusive) at which the adapter reaches that distance; for '-' for every
 *                such end position of the reverse complement the start of the longest optimal alignment ending
 *                there, stored as start - len(read) - 1 exactly like the reference (:189); ascending by end.
 */
int sct_adapter_align(sct_ctx *ctx, const char *adapter, int32_t m, const char *reads, const int64_t *offs, int64_t n,
                      int8_t *strand, int32_t *dist, int64_t *loc_offs, int32_t **locs, int64_t *n_locs);
void sct_free_locs(sct_ctx *ctx, int32_t *locs);

#ifdef __cplusplus
}
#endif
#endif /* SCTAGGER_HIP_H */
