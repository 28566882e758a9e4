/*
 * sct_hostio.c — host-side ingest / egress helpers of the match_trie CLI (plain C, no GPU code).
 *
 * The reference parses the -lr TSV with a per-character Python loop (scTagger.py:611-626) and writes the
 * result with per-row f-strings (:777-789).  These two functions do the same byte shuffling at memory
 * speed so that the command line is not dominated by Python at 10^7 rows:
 *
 *   sctio_parse_lr   TSV bytes -> read-id spans (column 0) + CSR segments (column 3)
 *   sctio_render     result arrays -> exactly the bytes of the reference's output block
 *
 * Both are two-call APIs (size, then fill) over caller-owned buffers; bound from sctagger_amd/fastio.py.
 */
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

/* Number of lines (a last line without '\n' counts). */
int64_t sctio_count_lines(const uint8_t *buf, int64_t len) {
    int64_t n = 0;
    const uint8_t *p = buf, *end = buf + len;
    while (p < end) {
        const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
        n++;
        if (!q) break;
        p = q + 1;
    }
    return n;
}

/*
 * Column 0 span and column 3 bytes of every line (reference read_long_reads, scTagger.py:611-614: only
 * l[0] and l[3] are used; a 5th column and beyond are ignored).
 * Returns -1 on success, or the 0-based number of the first line with fewer than 4 tab-separated fields
 * (the reference raises IndexError there).  segs must hold `len` bytes; offs n+1 entries.
 */
int64_t sctio_parse_lr(const uint8_t *buf, int64_t len, int64_t n, int64_t *name_start, int64_t *name_end,
                       uint8_t *segs, int64_t *offs) {
    int64_t out = 0, p = 0;
    for (int64_t i = 0; i < n; i++) {
        /* one pass over the line: positions of the first four tabs and of the line end */
        int64_t tab[4] = {-1, -1, -1, -1};
        int nt = 0;
        int64_t q = p;
        for (; q < len; q++) {
            const uint8_t c = buf[q];
            if (c == '\n') break;
            if (c == '\t' && nt < 4) tab[nt++] = q;
        }
        if (nt < 3) return i;
        const int64_t s = tab[2] + 1, e = nt >= 4 ? tab[3] : q;
        name_start[i] = p;
        name_end[i] = tab[0];
        offs[i] = out;
        for (int64_t k = s; k < e; k++) segs[out++] = buf[k];
        p = q + 1;
    }
    offs[n] = out;
    return -1;
}

/*
 * The row parser of extract_sr_bc_from_lr (scTagger.py:819): `_,_,p,seg = l.rstrip('\n').split('\t')` needs
 * exactly four fields.  na[i] = 1 iff the third field is "NA".  Returns -1, or the first offending line number
 * (the reference raises ValueError there).
 */
int64_t sctio_parse_lr4(const uint8_t *buf, int64_t len, int64_t n, uint8_t *na, uint8_t *segs, int64_t *offs) {
    int64_t out = 0, p = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t tab[3] = {-1, -1, -1};
        int nt = 0;
        int64_t q = p;
        for (; q < len; q++) {
            const uint8_t c = buf[q];
            if (c == '\n') break;
            if (c == '\t') {
                if (nt < 3) tab[nt] = q;
                nt++;
            }
        }
        if (nt != 3) return i;
        na[i] = (tab[2] - tab[1] == 3 && buf[tab[1] + 1] == 'N' && buf[tab[1] + 2] == 'A') ? 1 : 0;
        offs[i] = out;
        for (int64_t k = tab[2] + 1; k < q; k++) segs[out++] = buf[k];
        p = q + 1;
    }
    offs[n] = out;
    return -1;
}

static int ndigits(uint64_t v) {
    int d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}
static uint8_t *put_num(uint8_t *o, uint64_t v) {
    int d = ndigits(v);
    for (int k = d - 1; k >= 0; k--) { o[k] = (uint8_t)('0' + v % 10); v /= 10; }
    return o + d;
}

/*
 * The reference's output block (scTagger.py:777-789) for rows [r0, r1): matched rows only, in rid order,
 *   name \t min_e \t n_hits \t segment \t comma-joined barcode strings \n
 * pattern_text: 2*n_bc rows of bl bytes, row p = string printed for hit p = bid*2+strand.
 * With out == NULL only the size is returned.
 */
int64_t sctio_render(const uint8_t *namebuf, const int64_t *name_start, const int64_t *name_end, const uint8_t *segs,
                     const int64_t *offs, const uint8_t *min_e, const int64_t *hit_offs, const uint32_t *hits,
                     const uint8_t *pattern_text, int32_t bl, int64_t r0, int64_t r1, uint8_t *out) {
    int64_t size = 0;
    uint8_t *o = out;
    for (int64_t r = r0; r < r1; r++) {
        if (min_e[r] == 0xFF) continue;
        const int64_t ln = name_end[r] - name_start[r], ls = offs[r + 1] - offs[r];
        const int64_t c = hit_offs[r + 1] - hit_offs[r];
        if (!out) {
            size += ln + ndigits(min_e[r]) + ndigits((uint64_t)c) + ls + c * bl + (c > 0 ? c - 1 : 0) + 5;
            continue;
        }
        memcpy(o, namebuf + name_start[r], (size_t)ln); o += ln;
        *o++ = '\t';
        o = put_num(o, min_e[r]);
        *o++ = '\t';
        o = put_num(o, (uint64_t)c);
        *o++ = '\t';
        memcpy(o, segs + offs[r], (size_t)ls); o += ls;
        *o++ = '\t';
        for (int64_t k = 0; k < c; k++) {
            if (k) *o++ = ',';
            memcpy(o, pattern_text + (int64_t)hits[hit_offs[r] + k] * bl, (size_t)bl); o += bl;
        }
        *o++ = '\n';
    }
    return out ? (int64_t)(o - out) : size;
}

/* ---- extract_lr_bc host helpers (reference scTagger.py:159-173 reader, :317-320 writer) ---- */
static int is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }  /* str.split() / str.rstrip() whitespace (ASCII) */

/*
 * read_fastqs for one file: line idx % 4 == 0 -> rnames.append(l.split()[0][1:]), idx % 4 == 1 -> seqs.append(l.rstrip()).
 * names / reads receive the bytes back to back, name_offs / read_offs the CSR offsets (capacity: number of lines / 4 + 2).
 * counts[0] = names written, counts[1] = reads written.  Returns -1, or the number of the first header line without a token
 * (the reference raises IndexError there).
 */
int64_t sctio_parse_fastq(const uint8_t *buf, int64_t len, uint8_t *names, int64_t *name_offs, uint8_t *reads, int64_t *read_offs,
                          int64_t *counts) {
    int64_t p = 0, idx = 0, nn = 0, nr = 0, no = 0, ro = 0;
    name_offs[0] = 0;
    read_offs[0] = 0;
    while (p < len) {
        int64_t q = p;
        while (q < len && buf[q] != '\n') q++;
        if ((idx & 3) == 0) {
            int64_t a = p;
            while (a < q && is_space(buf[a])) a++;
            if (a >= q) return idx;
            int64_t b = a;
            while (b < q && !is_space(buf[b])) b++;
            for (int64_t k = a + 1; k < b; k++) names[no++] = buf[k];
            name_offs[++nn] = no;
        } else if ((idx & 3) == 1) {
            int64_t b = q;
            while (b > p && is_space(buf[b - 1])) b--;
            for (int64_t k = p; k < b; k++) reads[ro++] = buf[k];
            read_offs[++nr] = ro;
        }
        idx++;
        p = q + 1;
    }
    counts[0] = nn;
    counts[1] = nr;
    return -1;
}

/*
 * output_matches (scTagger.py:317-320): name \t dist \t loc \t seq[start:stop] \n for rows [r0, r1).
 * loc_na[r] != 0 prints "NA" for loc; start/stop are already resolved to 0 <= start <= stop <= len(read).
 */
int64_t sctio_render_lr(const uint8_t *names, const int64_t *name_offs, const uint8_t *reads, const int64_t *read_offs,
                        const int32_t *dist, const int64_t *loc, const uint8_t *loc_na, const int64_t *start, const int64_t *stop,
                        int64_t r0, int64_t r1, uint8_t *out) {
    int64_t size = 0;
    uint8_t *o = out;
    for (int64_t r = r0; r < r1; r++) {
        const int64_t ln = name_offs[r + 1] - name_offs[r], ls = stop[r] - start[r];
        const int neg_d = dist[r] < 0, neg_l = !loc_na[r] && loc[r] < 0;
        const uint64_t ad = (uint64_t)(neg_d ? -(int64_t)dist[r] : dist[r]), al = (uint64_t)(neg_l ? -loc[r] : loc[r]);
        if (!out) {
            size += ln + neg_d + ndigits(ad) + (loc_na[r] ? 2 : neg_l + ndigits(al)) + ls + 4;
            continue;
        }
        memcpy(o, names + name_offs[r], (size_t)ln); o += ln;
        *o++ = '\t';
        if (neg_d) *o++ = '-';
        o = put_num(o, ad);
        *o++ = '\t';
        if (loc_na[r]) { *o++ = 'N'; *o++ = 'A'; }
        else { if (neg_l) *o++ = '-'; o = put_num(o, al); }
        *o++ = '\t';
        memcpy(o, reads + read_offs[r] + start[r], (size_t)ls); o += ls;
        *o++ = '\n';
    }
    return out ? (int64_t)(o - out) : size;
}

/* ---- block-wise helpers of the streaming match_trie command (sctagger_amd/fastio.py: stream_match_trie) ---- */

/*
 * The -lr reader (scTagger.py:611-614) on one BLOCK of the file: every complete line of buf[0..len) — and, if `final`, a last
 * line without '\n' — up to max_rows lines and seg_cap segment bytes.  Column 0 -> name spans (relative to buf), column 3 ->
 * segs / offs (int32, the reference's own index type, n + 1 entries) / lens8 (min(length, 255)).
 * info[0] = lines parsed, info[1] = bytes of buf consumed (a multiple of whole lines), info[2] = longest segment,
 * info[3] = 0-based number (within the block) of the first line with fewer than 4 fields, or -1,
 * info[4] = 1 if the block holds a byte >= 128 or a '\r' (the caller then falls back to the reference's text-mode reader).
 */
void sctio_parse_block(const uint8_t *buf, int64_t len, int32_t final, int64_t max_rows, int64_t seg_cap, int64_t *name_start,
                       int64_t *name_end, uint8_t *segs, int32_t *offs, uint8_t *lens8, int64_t *info) {
    int64_t out = 0, p = 0, i = 0, longest = 0;
    info[3] = -1;
    info[4] = 0;
    for (; i < max_rows && p < len; i++) {
        int64_t tab[4] = {-1, -1, -1, -1};
        int nt = 0;
        uint8_t hi = 0;
        int64_t q = -1, k = p;
#if defined(__x86_64__)
        /* sixteen bytes at a time: where the tabs and the line end are, and whether a byte needs the text-mode reader (>= 128, '\r');
         * a row of the usual shape is three loads instead of forty turns of the byte loop below (2.9 of the command's 27 CPU-seconds
         * per 10^8 rows were spent there) */
        {
            const __m128i vt = _mm_set1_epi8('\t'), vn = _mm_set1_epi8('\n'), vr = _mm_set1_epi8('\r');
            while (k + 16 <= len) {
                const __m128i c = _mm_loadu_si128((const __m128i *)(buf + k));
                uint32_t mt = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, vt));
                const uint32_t mn = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, vn));
                uint32_t mb = (uint32_t)_mm_movemask_epi8(_mm_or_si128(c, _mm_cmpeq_epi8(c, vr)));   /* (bit 7 of a byte = its movemask bit) */
                const uint32_t below = mn ? ((mn & (0u - mn)) - 1u) : 0xFFFFu;   /* the bytes in front of the first line end */
                mt &= below;
                mb &= below;
                if (mb) hi = 0x80;
                while (mt && nt < 4) {
                    tab[nt++] = k + __builtin_ctz(mt);
                    mt &= mt - 1;
                }
                if (mn) {
                    q = k + __builtin_ctz(mn);
                    break;
                }
                k += 16;
            }
        }
#endif
        if (q < 0) {   /* the last bytes of the buffer (or no SSE2): byte by byte */
            for (; k < len && buf[k] != '\n'; k++) {
                const uint8_t c = buf[k];
                hi |= (uint8_t)(c & 0x80) | (uint8_t)(c == '\r' ? 0x80 : 0);
                if (c == '\t' && nt < 4) tab[nt++] = k;
            }
            if (k >= len && !final) break;   /* an unfinished line: the caller brings it again with the next block */
            q = k;
        }
        if (hi) info[4] = 1;
        if (nt < 3) {
            if (info[3] < 0) info[3] = i;
            break;
        }
        const int64_t s = tab[2] + 1, e = nt >= 4 ? tab[3] : q, ls = e - s;
        if (out + ls > seg_cap) break;
        name_start[i] = p;
        name_end[i] = tab[0];
        offs[i] = (int32_t)out;
        lens8[i] = (uint8_t)(ls < 255 ? ls : 255);
        if (ls > longest) longest = ls;
        memcpy(segs + out, buf + s, (size_t)ls);
        out += ls;
        p = q + 1;
    }
    offs[i] = (int32_t)out;
    info[0] = i;
    info[1] = p < len ? p : len;
    info[2] = longest;
}

/*
 * sctio_render for a block whose hits come as per-row COUNTS (uint8 if cnt_bytes == 1, else uint32) and int32 segment offsets:
 * the hit list of row r starts where the counts of the rows before it end; the hits are uint16 or uint32 entries.  With
 * out == NULL only the size is returned.  out_cap = the bytes `out` can take: the exact size will do; with 64 bytes more every row
 * can be written with copies of fixed size (a name or a barcode as 16 bytes, a segment as 32, whatever their length: the next field
 * overwrites the surplus) — 10^8 rows are 5 x 10^8 calls of memcpy otherwise.
 */
int64_t sctio_render_block(const uint8_t *namebuf, const int64_t *name_start, const int64_t *name_end, const uint8_t *segs,
                           const int32_t *offs, const uint8_t *min_e, const void *cnt, int32_t cnt_bytes, const void *hits,
                           int32_t hit_bytes /* 2 = uint16 entries, else uint32 */, const uint8_t *pattern_text, int32_t bl, int64_t n,
                           uint8_t *out, int64_t out_cap) {
    int64_t size = 0, pos = 0;
    uint8_t *o = out;
    const int64_t seg_total = n > 0 ? (int64_t)offs[n] : 0;
    for (int64_t r = 0; r < n; r++) {
        const int64_t c = cnt_bytes == 1 ? (int64_t)((const uint8_t *)cnt)[r] : (int64_t)((const uint32_t *)cnt)[r];
        if (min_e[r] == 0xFF) {
            pos += c;
            continue;
        }
        const int64_t ln = name_end[r] - name_start[r], ls = (int64_t)offs[r + 1] - offs[r];
        if (!out) {
            size += ln + ndigits(min_e[r]) + ndigits((uint64_t)c) + ls + c * bl + (c > 0 ? c - 1 : 0) + 5;
            pos += c;
            continue;
        }
        /* the fast form: the usual shape of a row, and room for its surplus.  (A name is followed in its line by three more fields
         * and the segment: ln + 3 + ls bytes can be read from its start; a segment by the other segments up to seg_total.) */
        if (bl == 16 && ln <= 16 && ls <= 32 && ln + 3 + ls >= 16 && (int64_t)offs[r] + 32 <= seg_total &&
            (o - out) + 16 + 8 + 24 + 32 + 1 + c * 17 + 16 <= out_cap) {
            memcpy(o, namebuf + name_start[r], 16);
            o += ln;
            *o++ = '\t';
            o = put_num(o, min_e[r]);
            *o++ = '\t';
            o = put_num(o, (uint64_t)c);
            *o++ = '\t';
            memcpy(o, segs + offs[r], 32);
            o += ls;
            *o++ = '\t';
            if (hit_bytes == 2) {
                const uint16_t *h = (const uint16_t *)hits + pos;
                for (int64_t k = 0; k < c; k++) {
                    memcpy(o, pattern_text + (int64_t)h[k] * 16, 16);
                    o[16] = ',';
                    o += 17;
                }
            } else {
                const uint32_t *h = (const uint32_t *)hits + pos;
                for (int64_t k = 0; k < c; k++) {
                    memcpy(o, pattern_text + (int64_t)h[k] * 16, 16);
                    o[16] = ',';
                    o += 17;
                }
            }
            if (c > 0) o--;   /* the last comma becomes the line end */
            *o++ = '\n';
            pos += c;
            continue;
        }
        memcpy(o, namebuf + name_start[r], (size_t)ln); o += ln;
        *o++ = '\t';
        o = put_num(o, min_e[r]);
        *o++ = '\t';
        o = put_num(o, (uint64_t)c);
        *o++ = '\t';
        memcpy(o, segs + offs[r], (size_t)ls); o += ls;
        *o++ = '\t';
        for (int64_t k = 0; k < c; k++) {
            if (k) *o++ = ',';
            const int64_t p = hit_bytes == 2 ? (int64_t)((const uint16_t *)hits)[pos + k] : (int64_t)((const uint32_t *)hits)[pos + k];
            memcpy(o, pattern_text + p * bl, (size_t)bl); o += bl;
        }
        pos += c;
        *o++ = '\n';
    }
    return out ? (int64_t)(o - out) : size;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * gzip members, compressed side by side (the output side of the streaming command).
 *
 * The reference writes its TSV through gzip.open(..., "wt") — one serial level-9 stream (/root/reference/scTagger.py:770-772).
 * Here the rendered rows of a block are cut into pieces of `piece` bytes and every piece becomes a gzip member of its own
 * (any gzip reader concatenates members; the decompressed bytes are the reference's), compressed by `threads` threads inside
 * this one call: no Python thread, no interpreter lock in the way (round 3's pool of Python threads around zlib reached about
 * six-fold parallelism on 128 threads: 50 s for the 10^8-row output at level 9).  zlib does the compressing.
 * ------------------------------------------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <stdlib.h>
#include <zlib.h>

int64_t sctio_gzip_bound(int64_t n, int64_t piece) {
    if (piece < 4096) piece = 4096;
    const int64_t pieces = n / piece + 1, longest = n < piece ? n : piece;
    return pieces * (longest + longest / 1000 + 64);      /* every piece has a slot of its own (deflateBound + the gzip wrapper) */
}

typedef struct gz_job {
    const uint8_t *in;
    int64_t n, piece;
    int level, n_pieces;
    uint8_t *out;                 /* piece k is compressed to out + k * slot */
    int64_t slot;
    int64_t *sizes;
    int next;                     /* next piece nobody has taken (pool mutex) */
    int done;                     /* pieces finished (pool mutex) */
    int failed;
    struct gz_job *link;          /* jobs with pieces left, oldest first */
} gz_job;

/* One pool of compressing threads for the process, started at the first call and kept: creating threads per call costs an mmap /
 * munmap of a stack each — 58 000 of them over the 10^8-row command, every one a write lock on the address space that holds up the
 * page faults of every other thread (parse and render ran five times slower beside it). */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t work, finished;
    gz_job *head, *tail;
    int n_threads;
} gz_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, NULL, 0};

int64_t sctfd_gzip_member(const uint8_t *in, int64_t n, uint8_t *out, void *scratch_mem);   /* sct_fdeflate.c */
int64_t sctfd_scratch_bytes(void);
static __thread void *gz_fast_scratch = NULL;   /* per thread, kept while the thread lives */
/* ... and released when it ends: gz_piece also runs on the threads that call sctio_gzip_members (they take pieces while they wait),
 * which come and go with the command's workers */
static pthread_key_t gz_scratch_key;
static pthread_once_t gz_scratch_once = PTHREAD_ONCE_INIT;
static void gz_scratch_make_key(void) { (void)pthread_key_create(&gz_scratch_key, free); }
static void *gz_scratch(void) {
    if (!gz_fast_scratch) {
        pthread_once(&gz_scratch_once, gz_scratch_make_key);
        gz_fast_scratch = malloc((size_t)sctfd_scratch_bytes());
        if (gz_fast_scratch) (void)pthread_setspecific(gz_scratch_key, gz_fast_scratch);   /* the key's destructor frees it at thread exit */
    }
    return gz_fast_scratch;
}

static void gz_piece(gz_job *j, int k) {
    const int64_t a = (int64_t)k * j->piece, len = j->n - a < j->piece ? j->n - a : j->piece;
    if (j->level < 0) {   /* the command's own fast compressor (sct_fdeflate.c) */
        if (!gz_scratch()) {
            __atomic_store_n(&j->failed, 1, __ATOMIC_RELAXED);
            return;
        }
        j->sizes[k] = sctfd_gzip_member(j->in + a, len, j->out + (int64_t)k * j->slot, gz_fast_scratch);
        return;
    }
    z_stream z;
    memset(&z, 0, sizeof(z));
    int ok = deflateInit2(&z, j->level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY) == Z_OK;
    if (ok) {
        z.next_in = (Bytef *)(j->in + a);
        z.avail_in = (uInt)len;
        z.next_out = j->out + (int64_t)k * j->slot;
        z.avail_out = (uInt)j->slot;
        ok = deflate(&z, Z_FINISH) == Z_STREAM_END;
        j->sizes[k] = (int64_t)z.total_out;
        deflateEnd(&z);
    }
    if (!ok) __atomic_store_n(&j->failed, 1, __ATOMIC_RELAXED);
}

/* take the next piece of the oldest job that has one (mutex held); NULL if there is none */
static gz_job *gz_take(int *k) {
    gz_job *j = gz_pool.head;
    if (!j) return NULL;
    *k = j->next++;
    if (j->next >= j->n_pieces) {       /* every piece of this job is handed out: off the list */
        gz_pool.head = j->link;
        if (!gz_pool.head) gz_pool.tail = NULL;
    }
    return j;
}

static void *gz_thread(void *arg) {
    (void)arg;
    pthread_mutex_lock(&gz_pool.mu);
    for (;;) {
        int k = 0;
        gz_job *j = gz_take(&k);
        if (!j) {
            pthread_cond_wait(&gz_pool.work, &gz_pool.mu);
            continue;
        }
        pthread_mutex_unlock(&gz_pool.mu);
        gz_piece(j, k);
        pthread_mutex_lock(&gz_pool.mu);
        if (++j->done == j->n_pieces) pthread_cond_broadcast(&gz_pool.finished);
    }
    return NULL;
}

/* threads the pool should have (it only grows; at most 128).  Returns how many it has. */
int32_t sctio_gzip_pool(int32_t threads) {
    if (threads > 128) threads = 128;
    pthread_mutex_lock(&gz_pool.mu);
    while (gz_pool.n_threads < threads) {
        pthread_t t;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        pthread_attr_setstacksize(&at, 1 << 20);
        const int rc = pthread_create(&t, &at, gz_thread, NULL);
        pthread_attr_destroy(&at);
        if (rc != 0) break;
        gz_pool.n_threads++;
    }
    const int32_t n = gz_pool.n_threads;
    pthread_mutex_unlock(&gz_pool.mu);
    return n;
}

/*
 * in[0 .. n) -> consecutive gzip members of `piece` input bytes each (the last one shorter; n = 0: one empty member), written to
 * out[0 .. return value); out must hold sctio_gzip_bound(n, piece) bytes.  level 1..9: zlib; level < 0: the fast compressor of
 * sct_fdeflate.c.  The pieces are compressed by the process-wide pool
 * (sctio_gzip_pool; `threads` = how many threads it should have at least) and by the calling thread.  -1 on failure.
 */
int64_t sctio_gzip_members(const uint8_t *in, int64_t n, int32_t level, int64_t piece, int32_t threads, uint8_t *out, int64_t out_cap) {
    if (piece < 4096) piece = 4096;
    if (piece > (1 << 30)) piece = 1 << 30;
    gz_job j;
    memset(&j, 0, sizeof(j));
    j.in = in;
    j.n = n;
    j.piece = piece;
    j.level = level;
    j.n_pieces = (int)(n / piece + (n % piece != 0 || n == 0 ? 1 : 0));
    j.slot = (n < piece ? n : piece) + (n < piece ? n : piece) / 1000 + 64;
    if ((int64_t)j.n_pieces * j.slot > out_cap) return -1;
    j.out = out;
    j.sizes = (int64_t *)calloc((size_t)j.n_pieces, sizeof(int64_t));
    if (!j.sizes) return -1;
    if (threads > 1 && j.n_pieces > 1) sctio_gzip_pool(threads);
    pthread_mutex_lock(&gz_pool.mu);
    if (j.n_pieces > 1 && gz_pool.n_threads > 0) {
        if (gz_pool.tail) gz_pool.tail->link = &j;
        else gz_pool.head = &j;
        gz_pool.tail = &j;
        pthread_cond_broadcast(&gz_pool.work);
    }
    /* the calling thread compresses pieces of its own job too (all of them if there is no pool) */
    for (;;) {
        int k;
        if (j.next >= j.n_pieces) break;
        k = j.next++;
        if (j.next >= j.n_pieces && gz_pool.head) {          /* the last piece is taken: the job leaves the list */
            gz_job **pp = &gz_pool.head, *prev = NULL;
            while (*pp && *pp != &j) {
                prev = *pp;
                pp = &(*pp)->link;
            }
            if (*pp == &j) {
                *pp = j.link;
                if (gz_pool.tail == &j) gz_pool.tail = prev;
            }
        }
        pthread_mutex_unlock(&gz_pool.mu);
        gz_piece(&j, k);
        pthread_mutex_lock(&gz_pool.mu);
        j.done++;
    }
    while (j.done < j.n_pieces) pthread_cond_wait(&gz_pool.finished, &gz_pool.mu);
    pthread_mutex_unlock(&gz_pool.mu);
    int64_t total = -1;
    if (!j.failed) {
        total = 0;
        for (int k = 0; k < j.n_pieces; k++) {      /* close the gaps between the members, in order */
            if (total != (int64_t)k * j.slot) memmove(out + total, out + (int64_t)k * j.slot, (size_t)j.sizes[k]);
            total += j.sizes[k];
        }
    }
    free(j.sizes);
    return total;
}
