/*
 * oracle/trie_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's match_trie algorithm: the array-of-children
 * trie of segment windows, the error-bounded recursive DFS per barcode, the per-prefix
 * task split and the min/union merge.  It exists to check the HIP path and to be timed as
 * the "port" CPU baseline; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The shipped library never links or calls it.
 *
 * Parity: PINNED — tests/test_oracle.py checks this file against every golden fixture under
 * tests/golden/ (captured from the reference itself by tests/golden/make_golden.py) and
 * against an independent brute-force infix-Levenshtein checker (oracle/bruteforce.py).
 *
 * Each function cites the reference lines (/root/reference/scTagger.py) it follows.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ALPHABET 4

/* map_char: A,C,G,T -> 0..3, every other byte < 128 -> 0 (scTagger.py:512-516).
 * Bytes >= 128 raise IndexError in the reference; here they are reported as an error. */
static int map_char(unsigned char c) {
    switch (c) {
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 0;
    }
}

/* rev_compl: reverse, A<->T, C<->G, every other byte maps to itself (scTagger.py:148-156). */
static void rev_compl(const char *s, int n, char *out) {
    for (int i = 0; i < n; i++) {
        char c = s[n - 1 - i];
        switch (c) {
        case 'A': c = 'T'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        case 'T': c = 'A'; break;
        default: break;
        }
        out[i] = c;
    }
}

/* ---- int vector ---- */
typedef struct { int32_t *v; int64_t n, cap; } ivec;
static void ivec_push(ivec *a, int32_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 16;
        a->v = (int32_t *)realloc(a->v, (size_t)a->cap * sizeof(int32_t));
    }
    a->v[a->n++] = x;
}

/* ---- Trie (scTagger.py:526-599) ----
 * nid_to_child: uint32[4][capacity]; nid_to_rids: one list of read ids per node.
 * The per-node lists are kept as singly linked chains in one pool, appended at the tail so
 * that iteration order equals insertion order, as with the reference's Python lists. */
typedef struct {
    uint32_t *child;            /* [capacity][4] */
    int64_t *rid_head, *rid_tail; /* per node: first/last entry in the pool, -1 if none */
    int32_t *pool_rid;
    int64_t *pool_next;
    int64_t pool_n, pool_cap;
    int64_t size, capacity;
    int max_error, barcode_length;
} trie_t;

static trie_t *trie_new(int max_error, int barcode_length, int64_t capacity) { /* :527-535 */
    trie_t *t = (trie_t *)calloc(1, sizeof(trie_t));
    if (capacity < 16) capacity = 16;
    t->capacity = capacity;
    t->child = (uint32_t *)calloc((size_t)capacity * ALPHABET, sizeof(uint32_t));
    t->rid_head = (int64_t *)malloc((size_t)capacity * sizeof(int64_t));
    t->rid_tail = (int64_t *)malloc((size_t)capacity * sizeof(int64_t));
    t->rid_head[0] = t->rid_tail[0] = -1;
    t->size = 1;
    t->max_error = max_error;
    t->barcode_length = barcode_length;
    return t;
}

static void trie_free(trie_t *t) {
    if (!t) return;
    free(t->child); free(t->rid_head); free(t->rid_tail); free(t->pool_rid); free(t->pool_next); free(t);
}

static void trie_grow(trie_t *t) { /* __increase_capacity__ :537-548 (doubling) */
    int64_t nc = t->capacity * 2;
    t->child = (uint32_t *)realloc(t->child, (size_t)nc * ALPHABET * sizeof(uint32_t));
    memset(t->child + t->capacity * ALPHABET, 0, (size_t)(nc - t->capacity) * ALPHABET * sizeof(uint32_t));
    t->rid_head = (int64_t *)realloc(t->rid_head, (size_t)nc * sizeof(int64_t));
    t->rid_tail = (int64_t *)realloc(t->rid_tail, (size_t)nc * sizeof(int64_t));
    t->capacity = nc;
}

static void trie_append_rid(trie_t *t, int64_t nid, int32_t rid) {
    if (t->pool_n == t->pool_cap) {
        t->pool_cap = t->pool_cap ? t->pool_cap * 2 : 1024;
        t->pool_rid = (int32_t *)realloc(t->pool_rid, (size_t)t->pool_cap * sizeof(int32_t));
        t->pool_next = (int64_t *)realloc(t->pool_next, (size_t)t->pool_cap * sizeof(int64_t));
    }
    int64_t e = t->pool_n++;
    t->pool_rid[e] = rid;
    t->pool_next[e] = -1;
    if (t->rid_tail[nid] < 0) t->rid_head[nid] = e; else t->pool_next[t->rid_tail[nid]] = e;
    t->rid_tail[nid] = e;
}

/* Trie.insert (scTagger.py:550-564): walk/create the path of `word`; append read_id to every
 * node at 0-based index >= barcode_length - max_error - 1 (duplicates allowed).
 * Returns -1 if the word holds a byte >= 128: the reference raises before it inserts anything of such a word
 * (`sub_segment.decode()`, :682 — UnicodeDecodeError — or, for a valid multi-byte sequence, `map_char[ord(char)]`, :555). */
static int trie_insert(trie_t *t, const char *word, int len, int32_t read_id) {
    int64_t nid = 0;
    for (int index = 0; index < len; index++)
        if ((unsigned char)word[index] >= 128) return -1;
    for (int index = 0; index < len; index++) {
        int cid = map_char((unsigned char)word[index]);
        if (t->child[nid * ALPHABET + cid] == 0) {
            int64_t new_nid = t->size;
            t->size += 1;
            if (t->size >= t->capacity) trie_grow(t);
            t->child[nid * ALPHABET + cid] = (uint32_t)new_nid;
            t->rid_head[new_nid] = t->rid_tail[new_nid] = -1;
        }
        nid = t->child[nid * ALPHABET + cid];
        if (index >= t->barcode_length - t->max_error - 1) trie_append_rid(t, nid, read_id);
    }
    return 0;
}

/* Trie.dfs (scTagger.py:566-588): delete, then per child: match / mutation / insert. */
static void trie_dfs(const trie_t *t, int64_t nid, const char *barcode, int blen, int error, int index, ivec *result) {
    if (index == blen) {
        int edit_distance = t->max_error - error;
        for (int64_t e = t->rid_head[nid]; e >= 0; e = t->pool_next[e]) ivec_push(&result[edit_distance], t->pool_rid[e]);
        return;
    }
    if (error > 0) trie_dfs(t, nid, barcode, blen, error - 1, index + 1, result); /* delete */
    int want = map_char((unsigned char)barcode[index]);
    for (int cid = 0; cid < ALPHABET; cid++) {
        int64_t child_id = t->child[nid * ALPHABET + cid];
        if (child_id == 0) continue;
        if (cid == want) trie_dfs(t, child_id, barcode, blen, error, index + 1, result);              /* dont change */
        if (cid != want && error > 0) trie_dfs(t, child_id, barcode, blen, error - 1, index + 1, result); /* mutation */
        if (error > 0) trie_dfs(t, child_id, barcode, blen, error - 1, index, result);                /* insert */
    }
}

/* Trie.query (scTagger.py:590-599): result[d] = read ids reached with d edits, d = 0..max_error. */
static void trie_query(const trie_t *t, const char *barcode, int blen, ivec *result) {
    for (int d = 0; d <= t->max_error; d++) result[d].n = 0;
    trie_dfs(t, 0, barcode, blen, t->max_error, 0, result);
}

/* ---- per-read fold: result[rid] = (min e, {(bid, strand)}) (scTagger.py:643-668) ---- */
typedef struct {
    uint8_t *best;   /* n_seg; sentinel = bl + 1 clipped to 255 */
    ivec *set;       /* n_seg vectors of (bid << 1 | strand), strand 1 = forward (True) */
    int64_t n_seg;
    int sentinel;
} fold_t;

static fold_t *fold_new(int64_t n_seg, int bl) {
    fold_t *f = (fold_t *)calloc(1, sizeof(fold_t));
    f->n_seg = n_seg;
    f->sentinel = bl + 1 > 255 ? 255 : bl + 1;
    f->best = (uint8_t *)malloc((size_t)(n_seg ? n_seg : 1));
    memset(f->best, f->sentinel, (size_t)n_seg);
    f->set = (ivec *)calloc((size_t)(n_seg ? n_seg : 1), sizeof(ivec));
    return f;
}
static void fold_free(fold_t *f) {
    if (!f) return;
    for (int64_t i = 0; i < f->n_seg; i++) free(f->set[i].v);
    free(f->set); free(f->best); free(f);
}
static void fold_add(fold_t *f, int32_t rid, int e, int32_t code) {
    if (rid < 0) return; /* the prefix sentinel rid -1 (scTagger.py:675) — unreachable unless bl-e-1 < prefix length */
    if (e < f->best[rid]) { f->best[rid] = (uint8_t)e; f->set[rid].n = 0; }
    if (e == f->best[rid]) {
        ivec *s = &f->set[rid];
        for (int64_t k = s->n - 1; k >= 0; k--) if (s->v[k] == code) return; /* set semantics */
        ivec_push(s, code);
    }
}

/* trie_search (scTagger.py:643-668): barcode order, forward (True) then reverse complement (False). */
static void trie_search(const trie_t *t, const char *bc, const char *bc_rc, int n_bc, int bl, fold_t *f) {
    ivec *result = (ivec *)calloc((size_t)t->max_error + 1, sizeof(ivec));
    for (int bid = 0; bid < n_bc; bid++) {
        trie_query(t, bc + (size_t)bid * bl, bl, result);
        for (int e = 0; e <= t->max_error; e++)
            for (int64_t k = 0; k < result[e].n; k++) fold_add(f, result[e].v[k], e, (bid << 1) | 1);
        trie_query(t, bc_rc + (size_t)bid * bl, bl, result);
        for (int e = 0; e <= t->max_error; e++)
            for (int64_t k = 0; k < result[e].n; k++) fold_add(f, result[e].v[k], e, (bid << 1) | 0);
    }
    for (int e = 0; e <= t->max_error; e++) free(result[e].v);
    free(result);
}

/* get_matches_prefix (scTagger.py:671-685): windows segment[i:i+bl+e] for every i, kept iff they
 * start with the task's prefix (raw bytes) and are at least bl-e long. */
static fold_t *get_matches_prefix(const char *bc, const char *bc_rc, int n_bc, int bl, const char *segs,
                                  const int64_t *offs, int64_t n_seg, int max_error, const char *prefix, int plen) {
    trie_t *t = trie_new(max_error, bl, 1 << 16);
    trie_insert(t, prefix, plen, -1);
    for (int64_t rid = 0; rid < n_seg; rid++) {
        const char *segment = segs + offs[rid];
        int64_t len = offs[rid + 1] - offs[rid];
        for (int64_t i = 0; i < len; i++) {
            int64_t sl = len - i < bl + max_error ? len - i : bl + max_error;
            if (sl >= plen && memcmp(segment + i, prefix, (size_t)plen) == 0 && sl >= bl - max_error)
                if (trie_insert(t, segment + i, (int)sl, (int32_t)rid) != 0) { /* the worker raises; the pool re-raises it in the parent (:704) */
                    trie_free(t);
                    return NULL;
                }
        }
    }
    fold_t *f = fold_new(n_seg, bl);
    trie_search(t, bc, bc_rc, n_bc, bl, f);
    trie_free(t);
    return f;
}

/* ---- run_get_matches_prefix (scTagger.py:688-716) ---- */
typedef struct {
    const char *bc, *bc_rc, *segs; const int64_t *offs;
    int n_bc, bl, max_error, plen, n_tasks;
    int64_t n_seg;
    const char *prefixes; /* n_tasks * plen */
    fold_t **out;
    int next; pthread_mutex_t mu;
} job_t;

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int k = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (k >= j->n_tasks) break;
        j->out[k] = get_matches_prefix(j->bc, j->bc_rc, j->n_bc, j->bl, j->segs, j->offs, j->n_seg, j->max_error,
                                       j->prefixes + (size_t)k * j->plen, j->plen);
    }
    return NULL;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* Number of prefix characters for `threads` (scTagger.py:689-697): smallest k with 4^k >= threads. */
int orc_prefix_len(int threads) {
    int k = 0; long long n = 1;
    while (threads > n) { n *= 4; k++; }
    return k;
}

/*
 * The whole seam: run_get_matches_prefix(max_error, threads) over globals (barcodes, lr_segs, lr_segs_idxs).
 *   bc       n_bc*bl forward barcodes, ASCII (the rc strings are derived here with rev_compl)
 *   segs/offs  CSR segments (lr_segs / lr_segs_idxs, scTagger.py:616-627)
 *   threads  the reference's -t: decides the prefix length (and so the non-ACGT window rule)
 *   workers  OS threads actually used to run the 4^k tasks (the reference uses Pool(threads))
 * Outputs (caller frees *hits with orc_free): min_e[rid] (0xFF = rid absent from the dict),
 * hit_offs CSR, hits = sorted (bid<<1|strand) == sorted(bids) of scTagger.py:789.
 * Returns 0, or -1 if a byte >= 128 is met (the reference raises IndexError/UnicodeDecodeError).
 */
int orc_match(const char *bc, int n_bc, int bl, const char *segs, const int64_t *offs, int64_t n_seg, int max_error,
              int threads, int workers, uint8_t *min_e, int64_t *hit_offs, uint32_t **hits, int64_t *n_hits) {
    for (int64_t i = 0; i < (int64_t)n_bc * bl; i++) if ((unsigned char)bc[i] >= 128) return -1;
    /* a segment byte >= 128 is an error only where the reference reads it: in a window it inserts (get_matches_prefix) */
    char *bc_rc = (char *)malloc((size_t)n_bc * bl + 1);
    for (int b = 0; b < n_bc; b++) rev_compl(bc + (size_t)b * bl, bl, bc_rc + (size_t)b * bl);

    int plen = orc_prefix_len(threads);
    if (plen > 10) { free(bc_rc); return -2; } /* more than 4^10 prefix tasks: not a configuration the reference is run with */
    int n_tasks = 1;
    for (int i = 0; i < plen; i++) n_tasks *= 4;
    char *prefixes = (char *)malloc((size_t)n_tasks * (plen ? plen : 1));
    for (int k = 0; k < n_tasks; k++) { /* 'ACGT' product order, most significant first (:691-697) */
        int x = k;
        for (int p = plen - 1; p >= 0; p--) { prefixes[(size_t)k * plen + p] = "ACGT"[x & 3]; x >>= 2; }
    }
    job_t j;
    memset(&j, 0, sizeof(j));
    j.bc = bc; j.bc_rc = bc_rc; j.segs = segs; j.offs = offs; j.n_bc = n_bc; j.bl = bl; j.max_error = max_error;
    j.plen = plen; j.n_tasks = n_tasks; j.n_seg = n_seg; j.prefixes = prefixes;
    j.out = (fold_t **)calloc((size_t)n_tasks, sizeof(fold_t *));
    pthread_mutex_init(&j.mu, NULL);
    if (workers < 1) workers = 1;
    if (workers > n_tasks) workers = n_tasks;
    pthread_t *th = (pthread_t *)malloc((size_t)workers * sizeof(pthread_t));
    for (int w = 1; w < workers; w++) pthread_create(&th[w], NULL, worker, &j);
    worker(&j);
    for (int w = 1; w < workers; w++) pthread_join(th[w], NULL);
    free(th);

    int raised = 0;
    for (int k = 0; k < n_tasks; k++) if (!j.out[k]) raised = 1;
    if (raised) {
        for (int k = 0; k < n_tasks; k++) if (j.out[k]) fold_free(j.out[k]);
        free(j.out); free(prefixes); free(bc_rc);
        return -1;
    }
    /* merge (scTagger.py:705-715): min e per rid, union of the sets at that minimum */
    fold_t *m = fold_new(n_seg, bl);
    for (int k = 0; k < n_tasks; k++) {
        fold_t *f = j.out[k];
        for (int64_t rid = 0; rid < n_seg; rid++) {
            for (int64_t q = 0; q < f->set[rid].n; q++) fold_add(m, (int32_t)rid, f->best[rid], f->set[rid].v[q]);
        }
        fold_free(f);
    }
    free(j.out); free(prefixes); free(bc_rc);

    int64_t total = 0;
    for (int64_t rid = 0; rid < n_seg; rid++) {
        hit_offs[rid] = total;
        if (m->set[rid].n == 0) { min_e[rid] = 0xFF; continue; }
        min_e[rid] = m->best[rid];
        total += m->set[rid].n;
    }
    hit_offs[n_seg] = total;
    uint32_t *h = (uint32_t *)malloc((size_t)(total ? total : 1) * sizeof(uint32_t));
    for (int64_t rid = 0; rid < n_seg; rid++) {
        ivec *s = &m->set[rid];
        if (s->n == 0) continue;
        qsort(s->v, (size_t)s->n, sizeof(int32_t), cmp_i32); /* sorted(bids): bid asc, False(0) before True(1) */
        memcpy(h + hit_offs[rid], s->v, (size_t)s->n * sizeof(int32_t));
    }
    fold_free(m);
    *hits = h;
    *n_hits = total;
    return 0;
}

void orc_free(void *p) { free(p); }

/* ---- unit-level entry points for the Trie golden vectors (tests/golden/g4.trie_unit.json) ---- */
void *orc_trie_new(int max_error, int barcode_length) { return trie_new(max_error, barcode_length, 64); }
void orc_trie_free(void *t) { trie_free((trie_t *)t); }
void orc_trie_insert(void *t, const char *word, int len, int32_t rid) { (void)trie_insert((trie_t *)t, word, len, rid); }
int64_t orc_trie_size(void *t) { return ((trie_t *)t)->size; }
/* query -> flat list: counts[d] entries for d = 0..max_error, concatenated into out (capacity cap). Returns total or -1. */
int64_t orc_trie_query(void *tv, const char *barcode, int blen, int64_t *counts, int32_t *out, int64_t cap) {
    trie_t *t = (trie_t *)tv;
    ivec *result = (ivec *)calloc((size_t)t->max_error + 1, sizeof(ivec));
    trie_query(t, barcode, blen, result);
    int64_t total = 0;
    for (int d = 0; d <= t->max_error; d++) {
        counts[d] = result[d].n;
        for (int64_t k = 0; k < result[d].n; k++) { if (total < cap) out[total] = result[d].v[k]; total++; }
        free(result[d].v);
    }
    free(result);
    return total <= cap ? total : -1;
}
