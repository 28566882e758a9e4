// Host-side check of the arithmetic in sctagger_amd/csrc/sct_core.h (the header the gfx950 kernels are
// built from) against a plain dynamic-programming infix Levenshtein.  Test infrastructure only: built and
// run by tests/test_core_host.py with g++; nothing here is shipped or used by the product path.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <functional>
#include <set>

#include "../sctagger_amd/csrc/sct_core.h"
#include "../sctagger_amd/csrc/sct_tables.h"

static int brute_infix(const std::vector<int> &p, const std::vector<int> &t) {
    const int m = (int)p.size(), n = (int)t.size();
    std::vector<int> prev(m + 1), cur(m + 1);
    for (int i = 0; i <= m; i++) prev[i] = i;
    int best = prev[m];
    for (int j = 1; j <= n; j++) {
        cur[0] = 0;
        for (int i = 1; i <= m; i++) cur[i] = std::min({prev[i - 1] + (p[i - 1] != t[j - 1]), prev[i] + 1, cur[i - 1] + 1});
        best = std::min(best, cur[m]);
        std::swap(prev, cur);
    }
    return best;
}

static void planes(const std::vector<int> &s, uint64_t &lo, uint64_t &hi) {
    lo = hi = 0;
    for (size_t i = 0; i < s.size(); i++) {
        lo |= (uint64_t)(s[i] & 1) << i;
        hi |= (uint64_t)(s[i] >> 1) << i;
    }
}

template <typename W>
static int myers_infix(const std::vector<int> &p, const std::vector<int> &t) {
    uint64_t lo, hi;
    planes(p, lo, hi);
    MyersState<W> s;
    const int m = (int)p.size();
    myers_init(s, m);
    int best = m;
    for (int c : t) {
        myers_step<W>(s, (W)lo, (W)hi, (uint32_t)c, m);
        best = std::min(best, s.score);
    }
    return best;
}

template <int E>
static int wm(const std::vector<int> &p, const std::vector<int> &t) {
    uint64_t plo, phi, tlo, thi;
    planes(p, plo, phi);
    planes(t, tlo, thi);
    const int L = (int)t.size();
    const uint32_t va = L >= 31 ? 0xFFFFFFFFu : ((1u << (L + 1)) - 1u);
    const uint32_t ve = va & ~1u;
    return wm_verify<E>((uint32_t)tlo << 1, (uint32_t)thi << 1, ve, va, (uint32_t)plo, (uint32_t)phi, (int)p.size());
}

// minimum anchored distance over every enumerated placement whose key equals the pattern's key of that family
template <typename W>
struct EmArray {
    W m[4];
    W operator()(uint32_t c) const { return m[c]; }
};
template <int E, int F, typename W>
struct AnchoredMin {
    W tlo, thi, va;
    uint32_t pkey, code;
    EmArray<W> em;
    int best;
    void operator()(int t, int t2) {
        if (sct_placement_key<F, W>(tlo, thi, t, t2) != pkey) return;
        const int d = wm_anchored<E, F, W>(em, va, code, t, t2);
        if (d < best) best = d;
    }
};
template <int E, int F, typename W>
static int anchored_family(const std::vector<int> &p, const std::vector<int> &t) {
    uint64_t plo, phi, tlo, thi;
    planes(p, plo, phi);
    planes(t, tlo, thi);
    const int L = (int)t.size();
    AnchoredMin<E, F, W> am;
    am.tlo = (W)tlo; am.thi = (W)thi;
    am.pkey = sct_pattern_key(F, (uint32_t)plo, (uint32_t)phi);
    am.va = sct_valid_mask<W>(L);
    const W ve = am.va & ~(W)1, l1 = (W)tlo << 1, h1 = (W)thi << 1;
    am.em.m[0] = ~l1 & ~h1 & ve; am.em.m[1] = l1 & ~h1 & ve; am.em.m[2] = ~l1 & h1 & ve; am.em.m[3] = l1 & h1 & ve;
    am.code = 0;
    for (int j = 0; j < 16; j++) am.code |= (uint32_t)p[j] << (2 * j);
    am.best = E + 1;
    sct_for_each_placement<F>(L, am, E);   // pruned as the kernels prune: completeness is checked WITH the pruning
    return am.best;
}
template <int E, typename W = uint32_t>
static int anchored(const std::vector<int> &p, const std::vector<int> &t) {
    int b = anchored_family<E, SCT_F_H1, W>(p, t);
    if (E >= 1) b = std::min(b, anchored_family<E, SCT_F_H2, W>(p, t));
    if (E >= 2) {
        b = std::min(b, anchored_family<E, SCT_F_MID, W>(p, t));
        b = std::min(b, anchored_family<E, SCT_F_02, W>(p, t));
        b = std::min(b, anchored_family<E, SCT_F_13, W>(p, t));
        b = std::min(b, anchored_family<E, SCT_F_03, W>(p, t));
    }
    return b;
}

// the thirds scheme (families T01, T12, T02), complete for <= 1 edit
template <int E, typename W = uint32_t>
static int anchored_thirds(const std::vector<int> &p, const std::vector<int> &t) {
    int b = anchored_family<E, SCT_F_T01, W>(p, t);
    if (E >= 1) {
        b = std::min(b, anchored_family<E, SCT_F_T12, W>(p, t));
        b = std::min(b, anchored_family<E, SCT_F_T02, W>(p, t));
    }
    return b;
}


// The partitioned join's compare (sct_join_distance on a tuple's context, sct_core.h) against the anchored automaton, for every
// placement of family F whose key matches: the two must give the same distance (0, 1, or 2 = "more than one edit").
template <int F>
static long join_vs_automaton(const std::vector<int> &p, const std::vector<int> &t, bool *ok) {
    uint64_t plo, phi, tlo, thi;
    planes(p, plo, phi);
    planes(t, tlo, thi);
    const int L = (int)t.size();
    const uint32_t pkey = sct_pattern_key(F, (uint32_t)plo, (uint32_t)phi);
    const uint32_t va = sct_valid_mask<uint32_t>(L), ve = va & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
    EmArray<uint32_t> em;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    uint32_t code = 0;
    for (int j = 0; j < 16; j++) code |= (uint32_t)p[j] << (2 * j);
    const uint32_t rest = sct_rest_of(SctFamily<F>{}, code);
    long n_checked = 0;
    sct_for_each_placement<F>(L, [&](int a, int b) {
        if (sct_placement_key<F, uint32_t>((uint32_t)tlo, (uint32_t)thi, a, b) != pkey) return;
        const int want = std::min(wm_anchored<1, F, uint32_t>(em, va, code, a, b), 2);
        uint32_t got;
        if (F == SCT_F_T01) {
            const uint32_t c = sct_join_ctx_t01((uint32_t)tlo, (uint32_t)thi, L, a);
            got = sct_join_distance(rest, c, 6, (int)(c >> 14), false);
        } else if (F == SCT_F_T12) {
            const uint32_t c = sct_join_ctx_t12((uint32_t)tlo, (uint32_t)thi, a);
            got = sct_join_distance(sct_join_rest_t12(rest), c, 5, (int)(c >> 14), false);
        } else {
            const uint32_t c = sct_join_ctx_t02((uint32_t)tlo, (uint32_t)thi, a, b);
            got = sct_join_distance(rest, c, 5, 5 + (int)(c >> 14) - 1, true);
        }
        if ((int)got != want) {
            printf("FAIL join compare family %d placement (%d, %d) L=%d: automaton %d, join %u\n", F, a, b, L, want, got);
            *ok = false;
        }
        n_checked++;
    });
    return n_checked;
}

// the wide scheme (sixteen 20-bit families, max_err = 2): minimum anchored distance over the placements of family WF whose key
// equals the pattern's; `lookups` counts the placements
template <int WF, typename W>
static int anchored_wide_family(const std::vector<int> &p, const std::vector<int> &t, long *lookups = nullptr) {
    uint64_t plo, phi, tlo, thi;
    planes(p, plo, phi);
    planes(t, tlo, thi);
    const int L = (int)t.size();
    EmArray<W> em;
    const W va = sct_valid_mask<W>(L), ve = va & ~(W)1, l1 = (W)tlo << 1, h1 = (W)thi << 1;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    uint32_t code = 0;
    for (int j = 0; j < 16; j++) code |= (uint32_t)p[j] << (2 * j);
    const uint32_t pkey = sct_wide_pattern_key(WF, (uint32_t)plo, (uint32_t)phi);
    int best = 3;
    sct_for_each_wide_placement<WF>(L, [&](int a, int b, int u) {
        if (lookups) ++*lookups;
        if (a < 0 || u < 0 || !sct_wide_valid(b, u, L)) { printf("FAIL wide placement outside the text\n"); exit(1); }
        if (sct_wide_key<WF, W>((W)tlo, (W)thi, a, b, u) != pkey) return;
        best = std::min(best, wm_anchored<2, SctWide<WF>::BASE, W>(em, va, code, a, b));
    });
    return best;
}
// tier 0 = W0; tier 1 = W0 W1 W3 W4; tier 2 = all
template <typename W = uint32_t>
static int anchored_wide(const std::vector<int> &p, const std::vector<int> &t, int tier, long *lookups = nullptr) {
    int b = anchored_wide_family<0, W>(p, t, lookups);
    if (tier >= 1) {
        b = std::min(b, anchored_wide_family<1, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<3, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<4, W>(p, t, lookups));
    }
    if (tier >= 2) {
        b = std::min(b, anchored_wide_family<2, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<5, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<6, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<7, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<8, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<9, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<10, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<11, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<12, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<13, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<14, W>(p, t, lookups));
        b = std::min(b, anchored_wide_family<15, W>(p, t, lookups));
    }
    return b;
}
// distance as the wide scheme reports it: the minimum over all families; with the tiers: a text whose minimum is 0 has it after
// tier 0, one whose minimum is 1 after tier 1
template <typename W = uint32_t>
static bool check_wide(const std::vector<int> &p, const std::vector<int> &t, int want) {
    if (anchored_wide<W>(p, t, 2) != std::min(want, 3)) { printf("FAIL wide want=%d got=%d n=%zu\n", want, anchored_wide<W>(p, t, 2), t.size()); return false; }
    if (want == 0 && anchored_wide<W>(p, t, 0) != 0) { printf("FAIL wide tier 0\n"); return false; }
    if (want <= 1 && anchored_wide<W>(p, t, 1) != want) { printf("FAIL wide tier 1 want=%d got=%d\n", want, anchored_wide<W>(p, t, 1)); return false; }
    return true;
}

// run-time families (barcode lengths 8..16): minimum anchored distance over all six quarter families
template <int E>
static int anchored_rt(const std::vector<int> &p, const std::vector<int> &t) {
    const int bl = (int)p.size(), L = (int)t.size();
    uint64_t plo, phi, tlo, thi;
    planes(p, plo, phi);
    planes(t, tlo, thi);
    EmArray<uint32_t> em;
    const uint32_t va = sct_valid_mask<uint32_t>(L), ve = va & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    uint32_t code = 0;
    for (int j = 0; j < bl; j++) code |= (uint32_t)p[j] << (2 * j);
    int best = E + 1;
    const int nf = E == 0 ? 1 : E == 1 ? 2 : 6;
    for (int f = 0; f < nf; f++) {
        const SctFamilyRT fam = sct_family_rt(f, bl);
        const uint32_t pkey = sct_key_of<uint32_t>(fam, (uint32_t)plo, (uint32_t)phi, fam.PA, fam.PB);
        sct_for_each_placement_of(fam, L, [&](int a, int b) {
            if (sct_key_of<uint32_t>(fam, (uint32_t)tlo, (uint32_t)thi, a, b) != pkey) return;
            best = std::min(best, wm_anchored_of<E, uint32_t>(fam, em, va, code, a, b));
        }, E);
    }
    return best;
}

struct SeedHit {
    const uint32_t *keys;  // the pattern's family keys
    bool hit = false;
    void operator()(int f, uint32_t k) {
        if (keys[f] == k) hit = true;
    }
};

template <int E>
static bool seeds_cover(const std::vector<int> &p, const std::vector<int> &t) {
    uint64_t plo, phi, tlo, thi;
    planes(p, plo, phi);
    planes(t, tlo, thi);
    uint32_t keys[SCT_NFAM_ALL];
    for (int f = 0; f < SCT_NFAM_ALL; f++) keys[f] = sct_pattern_key(f, (uint32_t)plo, (uint32_t)phi);
    SeedHit sh;
    sh.keys = keys;
    sct_for_each_seed<E, uint32_t>((uint32_t)tlo, (uint32_t)thi, (int)t.size(), sh);
    return sh.hit;
}

// ---- adapter alignment (edlib HW + reverse SHW for the start), as k_adapter_count / k_adapter_fill compute it ----
static std::vector<int> dp_last_row(const std::vector<int> &q, const std::vector<int> &t, bool global_start) {
    const int m = (int)q.size(), n = (int)t.size();
    std::vector<int> col(m + 1), nw(m + 1), out(n);
    for (int i = 0; i <= m; i++) col[i] = i;
    for (int j = 0; j < n; j++) {
        nw[0] = global_start ? j + 1 : 0;
        for (int i = 1; i <= m; i++) nw[i] = std::min({col[i - 1] + (q[i - 1] != t[j]), col[i] + 1, nw[i - 1] + 1});
        col = nw;
        out[j] = col[m];
    }
    return out;
}
static bool check_adapter(const std::vector<int> &q, const std::vector<int> &t) {
    const int m = (int)q.size(), n = (int)t.size();
    uint32_t peq[4] = {0, 0, 0, 0}, peqr[4] = {0, 0, 0, 0};
    for (int i = 0; i < m; i++) { peq[q[i]] |= 1u << i; peqr[q[m - 1 - i]] |= 1u << i; }
    // forward: best and ends
    std::vector<int> row = dp_last_row(q, t, false);
    MyersState<uint32_t> s;
    myers_init(s, m);
    int best = m + 1, cnt = 0;
    std::vector<int> sc(n);
    for (int j = 0; j < n; j++) {
        myers_step_eq<uint32_t, false>(s, peq[t[j]], m);
        sc[j] = s.score;
        if (s.score != row[j]) return false;
        if (s.score <= best) { cnt = s.score < best ? 1 : cnt + 1; best = s.score; }
    }
    if (n == 0) return true;
    // a fresh automaton started at `from` reproduces the scores from position from + 2m on (windowed second pass)
    for (int from = 1; from < n; from += 7) {
        MyersState<uint32_t> w;
        myers_init(w, m);
        for (int j = from; j < n; j++) {
            myers_step_eq<uint32_t, false>(w, peq[t[j]], m);
            if (j >= from + 2 * m && w.score != row[j]) return false;
        }
    }
    int want_best = *std::min_element(row.begin(), row.end());
    if (best != want_best || cnt != (int)std::count(row.begin(), row.end(), want_best)) return false;
    for (int j = 0; j < n; j++) {
        if (sc[j] != best) continue;
        // kernel's reverse SHW
        MyersState<uint32_t> qy;
        myers_init(qy, m);
        int last = 0;
        const int lim = std::min(m + best, j + 1);
        for (int x = 0; x < lim; x++) {
            myers_step_eq<uint32_t, true>(qy, peqr[t[j - x]], m);
            if (qy.score == best) last = x;
        }
        // definition: reversed query vs reversed prefix, SHW, last position with the best score
        std::vector<int> rq(q.rbegin(), q.rend()), rt(t.begin(), t.begin() + j + 1);
        std::reverse(rt.begin(), rt.end());
        std::vector<int> rr = dp_last_row(rq, rt, true);
        if (*std::min_element(rr.begin(), rr.end()) != best) return false;
        int wl = 0;
        for (int x = 0; x <= j; x++) if (rr[x] == best) wl = x;
        if (wl != last) return false;
    }
    return true;
}

// ---- the filter path over the seed tables, emulated on the host: every (family, placement) lookup of a text goes
// through the bucket slots (and, second run, through the plain offsets), every candidate through the anchored
// automaton on its rest characters, pattern ids are resolved as the kernel does (inline entry j -> ent[group base + within + j]).
// Result = (minimum distance, set of pattern ids at it); must equal the plain DP over all patterns.
struct PatPlanes {
    uint32_t x, y;
};
template <int E>
static bool emulate_filter(const SctSeedTables &T, const std::vector<std::vector<int>> &pats, int bl, const std::vector<int> &t,
                           bool use_slots) {
    const int L = (int)t.size();
    uint64_t tlo, thi;
    planes(t, tlo, thi);
    EmArray<uint32_t> em;
    const uint32_t va = sct_valid_mask<uint32_t>(L), ve = va & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    int best = E + 1;
    std::set<uint32_t> hits;
    auto add = [&](int d, uint32_t pid) {
        if (d > E) return;
        if (d < best) { best = d; hits.clear(); }
        if (d == best) hits.insert(pid);
    };
    const int nf = E == 0 ? 1 : E == 1 ? 2 : 6;
    for (int f = 0; f < nf; f++) {
        const SctFamilyRT fam = sct_family_rt(f, bl);
        const uint32_t *off = T.off.data() + T.off_base[f];
        sct_for_each_placement_of(fam, L, [&](int a, int b) {
            const uint32_t key = sct_key_of<uint32_t>(fam, (uint32_t)tlo, (uint32_t)thi, a, b);
            if (use_slots) {
                const uint64_t v = T.slot[T.off_base[f] + key];
                const uint32_t sa = (uint32_t)v, sb = (uint32_t)(v >> 32), mid = sct_slot_mid(sa, sb), top = sct_slot_top(sb);
                if (sct_slot_is_inline(sa)) {
                    const uint32_t q[3] = {mid, sct_slot_next(mid, top), top};
                    for (uint32_t j = 0; j < sct_slot_inline_count(sa); j++)
                        add(wm_anchored_rest<E, uint32_t>(fam, em, va, q[j], a, b),
                            T.ent[T.grp[(size_t)f * SCT_GRP_PER_FAM + (key >> SCT_GRP_SHIFT)] + sct_slot_within(sa) + j]);
                } else {
                    for (uint32_t k = 0; k < sct_slot_long_count(sa, top); k++)
                        add(wm_anchored_rest<E, uint32_t>(fam, em, va, sct_rest_of(fam, T.code[mid + k]), a, b), T.ent[mid + k]);
                }
            } else {
                for (uint32_t gk = off[key]; gk < off[key + 1]; gk++)
                    add(wm_anchored_of<E, uint32_t>(fam, em, va, T.code[gk], a, b), T.ent[gk]);
            }
        }, E);
    }
    int want = E + 1;
    std::set<uint32_t> want_hits;
    for (size_t p = 0; p < pats.size(); p++) {
        const int d = brute_infix(pats[p], t);
        if (d > E) continue;
        if (d < want) { want = d; want_hits.clear(); }
        if (d == want) want_hits.insert((uint32_t)p);
    }
    (void)bl;
    return want == best && want_hits == hits;
}

// the same through the wide families' tables (max_err = 2, 16-character barcodes)
template <int WF>
static void emulate_wide_family(const SctSeedTables &T, uint64_t tlo, uint64_t thi, int L, const EmArray<uint32_t> &em, uint32_t va,
                                const std::function<void(int, uint32_t)> &add) {
    const uint32_t *off = T.off.data() + T.off_base[SctWide<WF>::INDEX];
    sct_for_each_wide_placement<WF>(L, [&](int a, int b, int u) {
        const uint32_t key = sct_wide_key<WF, uint32_t>((uint32_t)tlo, (uint32_t)thi, a, b, u);
        for (uint32_t gk = off[key]; gk < off[key + 1]; gk++)
            add(wm_anchored<2, SctWide<WF>::BASE, uint32_t>(em, va, T.code[gk], a, b), T.ent[gk]);
    });
}
static bool emulate_filter_wide(const SctSeedTables &T, const std::vector<std::vector<int>> &pats, const std::vector<int> &t) {
    const int L = (int)t.size();
    uint64_t tlo, thi;
    planes(t, tlo, thi);
    EmArray<uint32_t> em;
    const uint32_t va = sct_valid_mask<uint32_t>(L), ve = va & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    int best = 3;
    std::set<uint32_t> hits;
    const std::function<void(int, uint32_t)> add = [&](int d, uint32_t pid) {
        if (d > 2) return;
        if (d < best) { best = d; hits.clear(); }
        if (d == best) hits.insert(pid);
    };
    emulate_wide_family<0>(T, tlo, thi, L, em, va, add);  emulate_wide_family<1>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<2>(T, tlo, thi, L, em, va, add);  emulate_wide_family<3>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<4>(T, tlo, thi, L, em, va, add);  emulate_wide_family<5>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<6>(T, tlo, thi, L, em, va, add);  emulate_wide_family<7>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<8>(T, tlo, thi, L, em, va, add);  emulate_wide_family<9>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<10>(T, tlo, thi, L, em, va, add); emulate_wide_family<11>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<12>(T, tlo, thi, L, em, va, add); emulate_wide_family<13>(T, tlo, thi, L, em, va, add);
    emulate_wide_family<14>(T, tlo, thi, L, em, va, add); emulate_wide_family<15>(T, tlo, thi, L, em, va, add);
    int want = 3;
    std::set<uint32_t> want_hits;
    for (size_t p = 0; p < pats.size(); p++) {
        const int d = brute_infix(pats[p], t);
        if (d > 2) continue;
        if (d < want) { want = d; want_hits.clear(); }
        if (d == want) want_hits.insert((uint32_t)p);
    }
    return want == best && want_hits == hits;
}


// ---- restricted starts (sct_core.h: sct_allowed_starts, SctStarts, wm_anchored_rest_rs, wm_anchored_pair with start functors) ----
// the reference's rule as a plain DP: an alignment begins at an allowed start; characters skipped in front of the pattern cost one
// edit each (row 0 is the distance back to the nearest allowed start)
static int brute_restricted(const std::vector<int> &p, const std::vector<int> &t, uint64_t allowed) {
    const int m = (int)p.size(), n = (int)t.size(), BIG = 1000;
    std::vector<int> prev(m + 1), cur(m + 1);
    prev[0] = (allowed & 1) ? 0 : BIG;
    for (int i = 1; i <= m; i++) prev[i] = std::min(BIG, prev[i - 1] + 1);
    int best = prev[m];
    for (int j = 1; j <= n; j++) {
        cur[0] = std::min(((allowed >> j) & 1) ? 0 : BIG, prev[0] + 1);
        for (int i = 1; i <= m; i++) cur[i] = std::min({prev[i - 1] + (p[i - 1] != t[j - 1]), prev[i] + 1, cur[i - 1] + 1, BIG});
        best = std::min(best, cur[m]);
        std::swap(prev, cur);
    }
    return best;
}
template <typename W>
struct StartFn {
    const SctStarts<W> *s;
    W operator()(int d) const { return s->a[d]; }
};
// minimum over the placements of `fam` (pruned for E) whose key equals the pattern's; the packed pair form must agree with the plain one
template <int E, typename W, typename Fam>
static int rs_family(const Fam &fam, const std::vector<int> &p, uint64_t tlo, uint64_t thi, int L, const SctStarts<W> &st, bool *pair_ok) {
    uint64_t plo, phi;
    planes(p, plo, phi);
    const uint32_t pkey = sct_key_of<uint32_t>(fam, (uint32_t)plo, (uint32_t)phi, fam.PA, fam.PB);
    EmArray<W> em;
    const W va = sct_valid_mask<W>(L), ve = va & ~(W)1, l1 = (W)tlo << 1, h1 = (W)thi << 1;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    uint32_t code = 0;
    for (size_t j = 0; j < p.size(); j++) code |= (uint32_t)p[j] << (2 * j);
    const uint32_t rest = sct_rest_of(fam, code);
    int best = E + 1, pt = -1, pt2 = -1, pd = 0;
    sct_for_each_placement_of(fam, L, [&](int a, int b) {
        if (sct_key_of<W>(fam, (W)tlo, (W)thi, a, b) != pkey) return;
        const int d = wm_anchored_rest_rs<E, W>(fam, em, va, st, rest, a, b);
        best = std::min(best, d);
        if (pt >= 0) {   // this placement and the one before it as the two halves of a pair
            const uint32_t r = wm_anchored_pair<E>(fam, em, em, va, va, rest, rest, pt, pt2, a, b, StartFn<W>{&st}, StartFn<W>{&st});
            if ((int)(r & 0xFFFFu) != pd || (int)(r >> 16) != d) *pair_ok = false;
        }
        pt = a; pt2 = b; pd = d;
    }, E);
    return best;
}
template <int E, typename W>
static int rs_quarters(const std::vector<int> &p, uint64_t tlo, uint64_t thi, int L, const SctStarts<W> &st, int tier, bool *pair_ok) {
    const int bl = (int)p.size();
    int b = rs_family<E, W>(sct_family_rt(SCT_F_H1, bl), p, tlo, thi, L, st, pair_ok);
    if (E >= 1 && tier >= 1) b = std::min(b, rs_family<E, W>(sct_family_rt(SCT_F_H2, bl), p, tlo, thi, L, st, pair_ok));
    if (E >= 2 && tier >= 2)
        for (int f = SCT_F_MID; f <= SCT_F_03; f++) b = std::min(b, rs_family<E, W>(sct_family_rt(f, bl), p, tlo, thi, L, st, pair_ok));
    return b;
}
template <int E, typename W>
static int rs_thirds(const std::vector<int> &p, uint64_t tlo, uint64_t thi, int L, const SctStarts<W> &st, int tier, bool *pair_ok) {
    int b = rs_family<E, W>(SctFamily<SCT_F_T01>{}, p, tlo, thi, L, st, pair_ok);
    if (E >= 1 && tier >= 1) {
        b = std::min(b, rs_family<E, W>(SctFamily<SCT_F_T12>{}, p, tlo, thi, L, st, pair_ok));
        b = std::min(b, rs_family<E, W>(SctFamily<SCT_F_T02>{}, p, tlo, thi, L, st, pair_ok));
    }
    return b;
}
template <int WF, typename W>
static int rs_wide_family(const std::vector<int> &p, uint64_t tlo, uint64_t thi, int L, const SctStarts<W> &st) {
    uint64_t plo, phi;
    planes(p, plo, phi);
    EmArray<W> em;
    const W va = sct_valid_mask<W>(L), ve = va & ~(W)1, l1 = (W)tlo << 1, h1 = (W)thi << 1;
    em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
    uint32_t code = 0;
    for (int j = 0; j < 16; j++) code |= (uint32_t)p[j] << (2 * j);
    const uint32_t pkey = sct_wide_pattern_key(WF, (uint32_t)plo, (uint32_t)phi);
    const typename SctWide<WF>::Base fam{};
    int best = 3;
    sct_for_each_wide_placement<WF>(L, [&](int a, int b, int u) {
        if (sct_wide_key<WF, W>((W)tlo, (W)thi, a, b, u) != pkey) return;
        best = std::min(best, wm_anchored_rest_rs<2, W>(fam, em, va, st, sct_rest_of(fam, code), a, b));
    });
    return best;
}
template <typename W>
static int rs_wide(const std::vector<int> &p, uint64_t tlo, uint64_t thi, int L, const SctStarts<W> &st, int tier) {
    int b = rs_wide_family<0, W>(p, tlo, thi, L, st);
    if (tier >= 1) {
        b = std::min(b, rs_wide_family<1, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<3, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<4, W>(p, tlo, thi, L, st));
    }
    if (tier >= 2) {
        b = std::min(b, rs_wide_family<2, W>(p, tlo, thi, L, st));   b = std::min(b, rs_wide_family<5, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<6, W>(p, tlo, thi, L, st));   b = std::min(b, rs_wide_family<7, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<8, W>(p, tlo, thi, L, st));   b = std::min(b, rs_wide_family<9, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<10, W>(p, tlo, thi, L, st));  b = std::min(b, rs_wide_family<11, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<12, W>(p, tlo, thi, L, st));  b = std::min(b, rs_wide_family<13, W>(p, tlo, thi, L, st));
        b = std::min(b, rs_wide_family<14, W>(p, tlo, thi, L, st));  b = std::min(b, rs_wide_family<15, W>(p, tlo, thi, L, st));
    }
    return b;
}
// one text (codes t, "not ACGT" flags bad), prefix length k, every scheme: the anchored minimum under restricted starts must be the
// restricted DP's value where that is <= E (tier by tier: a minimum of 0 after the first tier, of 1 after the second), more otherwise
template <int E, typename W>
static bool check_restricted(const std::vector<int> &p, const std::vector<int> &t, uint64_t bad, int k) {
    const int L = (int)t.size();
    uint64_t tlo, thi;
    planes(t, tlo, thi);
    const W va = sct_valid_mask<W>(L);
    const W a0 = sct_allowed_starts<W>((W)bad, L, k);
    // the definition, byte by byte
    uint64_t want_a0 = 0;
    for (int s = 0; s + k <= L; s++) {
        bool ok = true;
        for (int q = 0; q < k; q++) ok = ok && !((bad >> (s + q)) & 1);
        if (ok && s < (int)(8 * sizeof(W))) want_a0 |= 1ull << s;
    }
    if ((uint64_t)a0 != (want_a0 & (uint64_t)~(W)0)) { printf("FAIL allowed starts L=%d k=%d\n", L, k); return false; }
    SctStarts<W> st;
    st.from_allowed(a0, va);
    const int want = brute_restricted(p, t, want_a0);
    bool pair_ok = true;
    const int bl = (int)p.size();
    for (int tier = 0; tier <= E; tier++) {
        const int got = rs_quarters<E, W>(p, tlo, thi, L, st, tier == E ? 2 : tier, &pair_ok);
        if (want <= tier ? got != want : (tier == E && got <= E)) { printf("FAIL restricted quarters E=%d bl=%d L=%d k=%d tier=%d want=%d got=%d\n", E, bl, L, k, tier, want, got); return false; }
        if (want > tier && got < std::min(want, E + 1)) { printf("FAIL restricted quarters: %d below the DP's %d\n", got, want); return false; }
    }
    if (bl == 16 && E <= 1) {
        for (int tier = 0; tier <= E; tier++) {
            const int got = rs_thirds<E, W>(p, tlo, thi, L, st, tier, &pair_ok);
            if (want <= tier ? got != want : (tier == E && got <= E)) { printf("FAIL restricted thirds E=%d L=%d k=%d tier=%d want=%d got=%d\n", E, L, k, tier, want, got); return false; }
        }
    }
    if (bl == 16 && E == 2) {
        for (int tier = 0; tier <= 2; tier++) {
            const int got = rs_wide<W>(p, tlo, thi, L, st, tier);
            if (want <= tier ? got != want : (tier == 2 && got <= 2)) { printf("FAIL restricted wide L=%d k=%d tier=%d want=%d got=%d\n", L, k, tier, want, got); return false; }
        }
    }
    if (!pair_ok) { printf("FAIL restricted: packed pair differs from the plain automaton (E=%d bl=%d L=%d k=%d)\n", E, bl, L, k); return false; }
    return true;
}
// a long text cut into windows of 63 characters, `step` = 63 - (bl + E - 1) apart, each owning the starts in its first `step`
// characters (the last one: all of its own): the minimum over the windows is the restricted DP's value over the whole text
template <int E>
static bool check_windows(const std::vector<int> &p, const std::vector<int> &t, const std::vector<int> &badv, int k) {
    const int bl = (int)p.size(), L = (int)t.size(), ov = bl + E - 1, step = 63 - ov;
    const int nwin = L <= 63 ? 1 : (L - ov + step - 1) / step;
    // the DP over the whole text (allowed starts as a vector: the text is longer than a word)
    const int BIG = 1000;
    std::vector<int> prev(bl + 1), cur(bl + 1);
    auto allowed = [&](int s) {
        if (s + k > L) return false;
        for (int q = 0; q < k; q++) if (badv[s + q]) return false;
        return true;
    };
    prev[0] = allowed(0) ? 0 : BIG;
    for (int i = 1; i <= bl; i++) prev[i] = std::min(BIG, prev[i - 1] + 1);
    int want = prev[bl];
    for (int j = 1; j <= L; j++) {
        cur[0] = std::min(allowed(j) ? 0 : BIG, prev[0] + 1);
        for (int i = 1; i <= bl; i++) cur[i] = std::min({prev[i - 1] + (p[i - 1] != t[j - 1]), prev[i] + 1, cur[i - 1] + 1, BIG});
        want = std::min(want, cur[bl]);
        std::swap(prev, cur);
    }
    int got = E + 1;
    bool pair_ok = true;
    for (int w = 0; w < nwin; w++) {
        const int w0 = w * step, wl = std::min(63, L - w0);
        if (w == nwin - 1 && w0 + wl != L) { printf("FAIL windows do not cover the text L=%d\n", L); return false; }
        std::vector<int> tw(t.begin() + w0, t.begin() + w0 + wl);
        uint64_t bad = 0;
        for (int i = 0; i < wl; i++) bad |= (uint64_t)badv[w0 + i] << i;
        uint64_t tlo, thi;
        planes(tw, tlo, thi);
        const uint64_t va = sct_valid_mask<uint64_t>(wl);
        // s + k <= L for the ROW; inside a window that is not the last every owned start has k more characters in the window
        uint64_t a0 = sct_allowed_starts<uint64_t>(bad, wl, k);
        if (w != nwin - 1) a0 &= (1ull << step) - 1;
        SctStarts<uint64_t> st;
        st.from_allowed(a0, va);
        got = std::min(got, rs_quarters<E, uint64_t>(p, tlo, thi, wl, st, 2, &pair_ok));
    }
    if (std::min(want, E + 1) != got) { printf("FAIL windows E=%d bl=%d L=%d k=%d want=%d got=%d\n", E, bl, L, k, want, got); return false; }
    if (!pair_ok) { printf("FAIL windows: packed pair differs\n"); return false; }
    return true;
}

int main(int argc, char **argv) {
    // argv[1]: divisor of the random-case counts of sections 2 - 2c (the sanitizer build runs a quarter of them)
    const int scale = argc > 1 && atoi(argv[1]) > 0 ? atoi(argv[1]) : 1;
    std::mt19937_64 rng(12345);
    auto rnd = [&](int n) { return (int)(rng() % (uint64_t)n); };
    long checked = 0;
    // 0. four ASCII bytes at once (sct_decode4, what the filter kernel decodes its rows with) against the byte-wise map_char:
    //    every byte value in every lane of the word, beside random and beside valid neighbours
    for (int rep = 0; rep < 300; rep++)
        for (int lane = 0; lane < 4; lane++)
            for (int b = 0; b < 256; b++) {
                uint32_t w = 0;
                for (int j = 0; j < 4; j++) w |= (uint32_t)(rep % 3 == 0 ? (uint8_t)"ACGTNacgt-"[rnd(10)] : (uint8_t)rnd(256)) << (8 * j);
                w = (w & ~(0xFFu << (8 * lane))) | ((uint32_t)b << (8 * lane));
                uint32_t lo, hi, bad;
                sct_decode4(w, lo, hi, bad);
                if ((lo | hi | bad) >> 4) { printf("FAIL decode4 stray bits w=%08x\n", w); return 1; }
                for (int j = 0; j < 4; j++) {
                    const uint8_t c = (uint8_t)(w >> (8 * j));
                    const uint32_t code = sct_code(c);
                    if (((lo >> j) & 1u) != (code & 1u) || ((hi >> j) & 1u) != (code >> 1) || ((bad >> j) & 1u) != (sct_is_acgt(c) ? 0u : 1u)) {
                        printf("FAIL decode4 w=%08x byte %d\n", w, j);
                        return 1;
                    }
                }
                checked++;
            }
    // 1. Myers, 32- and 64-bit words, every pattern length, texts 0..70, small alphabets to force matches
    for (int it = 0; it < 60000; it++) {
        const int m = 1 + rnd(it % 2 ? 32 : 64), n = rnd(71), alpha = 1 + rnd(4);
        std::vector<int> p(m), t(n);
        for (auto &x : p) x = rnd(alpha);
        for (auto &x : t) x = rnd(alpha);
        if (n >= m && rnd(2)) {  // plant an edited copy
            int at = rnd(n - m + 1);
            for (int i = 0; i < m; i++) t[at + i] = p[i];
            for (int k = rnd(4); k > 0; k--) t[rnd(n)] = rnd(4);
        }
        const int want = brute_infix(p, t);
        if (m <= 32 && myers_infix<uint32_t>(p, t) != want) { printf("FAIL myers32 m=%d n=%d\n", m, n); return 1; }
        if (myers_infix<uint64_t>(p, t) != want) { printf("FAIL myers64 m=%d n=%d\n", m, n); return 1; }
        checked++;
    }
    // 1b. adapter alignment arithmetic (HW ends + reverse SHW starts), query 1..32, text 0..120
    for (int it = 0; it < 40000; it++) {
        const int m = 1 + rnd(32), n = rnd(121), alpha = 1 + rnd(4);
        std::vector<int> q(m), t(n);
        for (auto &x : q) x = rnd(alpha);
        for (auto &x : t) x = rnd(alpha);
        if (n >= m && rnd(2)) {
            int at = rnd(n - m + 1);
            for (int i = 0; i < m; i++) t[at + i] = q[i];
            for (int k = rnd(5); k > 0; k--) t[rnd(n)] = rnd(4);
        }
        if (!check_adapter(q, t)) { printf("FAIL adapter m=%d n=%d\n", m, n); return 1; }
        checked++;
    }
    // 1c. bucket slots: what the kernel decodes (and queues) is what the table builder packed
    for (int it = 0; it < 200000; it++) {
        const uint32_t n = (uint32_t)rnd(4);
        uint32_t rest[3] = {(uint32_t)rng(), (uint32_t)rng(), (uint32_t)rng()};
        const uint32_t within = (uint32_t)rnd(SCT_SLOT_WITHIN_MAX + 1);
        const uint64_t v = sct_slot_inline(n, rest, within);
        const uint32_t a = (uint32_t)v, b = (uint32_t)(v >> 32);
        const uint32_t mid = sct_slot_mid(a, b), top = sct_slot_top(b);
        const uint32_t q[3] = {mid, sct_slot_next(mid, top), top};  // entries 0, 1, 2 as queued
        bool ok = sct_slot_is_inline(a) && sct_slot_inline_count(a) == n && sct_slot_within(a) == within;
        for (uint32_t j = 0; j < n; j++) ok = ok && (q[j] & 0xFFFFu) == (rest[j] & 0xFFFFu);
        const uint32_t first = (uint32_t)rng();
        const uint32_t len = it % 3 == 0 ? 4u + (uint32_t)rnd(70000) : 4u + (uint32_t)(rng() % 0x7FFFFFFCull);
        const uint64_t w = sct_slot_long(first, len);
        const uint32_t wa = (uint32_t)w, wb = (uint32_t)(w >> 32);
        ok = ok && !sct_slot_is_inline(wa) && sct_slot_mid(wa, wb) == first && sct_slot_long_count(wa, sct_slot_top(wb)) == len;
        if (!ok) { printf("FAIL slot n=%u\n", n); return 1; }
        checked++;
    }
    // 1e. two candidates per word in 16-bit windows == the full-width anchored automaton, every family and length
    for (int it = 0; it < 60000; it++) {
        const int bl = it % 2 ? 16 : 8 + rnd(9);
        const int f = rnd(6);
        const SctFamilyRT fam = sct_family_rt(f, bl);
        EmArray<uint32_t> em[2];
        uint32_t va[2], rest[2];
        int tt[2], tt2[2];
        bool have = true;
        for (int h = 0; h < 2; h++) {
            const int L = rnd(32), alpha = 1 + rnd(4);
            std::vector<int> t(L);
            for (auto &x : t) x = rnd(alpha);
            uint64_t tlo, thi;
            planes(t, tlo, thi);
            va[h] = sct_valid_mask<uint32_t>(L);
            const uint32_t ve = va[h] & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
            em[h].m[0] = ~l1 & ~h1 & ve; em[h].m[1] = l1 & ~h1 & ve; em[h].m[2] = ~l1 & h1 & ve; em[h].m[3] = l1 & h1 & ve;
            rest[h] = alpha == 1 ? 0u : (uint32_t)rng() & (rnd(2) ? 0xFFFFu : 0x5555u * (uint32_t)rnd(4));
            std::vector<std::pair<int, int>> pl;
            sct_for_each_placement_of(fam, L, [&](int a, int b) { pl.push_back({a, b}); });
            if (pl.empty()) { have = false; break; }
            const auto ab = pl[rnd((int)pl.size())];
            tt[h] = ab.first;
            tt2[h] = ab.second;
        }
        if (!have) continue;
        bool ok = true;
        {
            const uint32_t r = wm_anchored_pair<0>(fam, em[0], em[1], va[0], va[1], rest[0], rest[1], tt[0], tt2[0], tt[1], tt2[1]);
            ok = ok && (int)(r & 0xFFFFu) == wm_anchored_rest<0, uint32_t>(fam, em[0], va[0], rest[0], tt[0], tt2[0]) &&
                 (int)(r >> 16) == wm_anchored_rest<0, uint32_t>(fam, em[1], va[1], rest[1], tt[1], tt2[1]);
        }
        {
            const uint32_t r = wm_anchored_pair<1>(fam, em[0], em[1], va[0], va[1], rest[0], rest[1], tt[0], tt2[0], tt[1], tt2[1]);
            ok = ok && (int)(r & 0xFFFFu) == wm_anchored_rest<1, uint32_t>(fam, em[0], va[0], rest[0], tt[0], tt2[0]) &&
                 (int)(r >> 16) == wm_anchored_rest<1, uint32_t>(fam, em[1], va[1], rest[1], tt[1], tt2[1]);
        }
        {
            const uint32_t r = wm_anchored_pair<2>(fam, em[0], em[1], va[0], va[1], rest[0], rest[1], tt[0], tt2[0], tt[1], tt2[1]);
            ok = ok && (int)(r & 0xFFFFu) == wm_anchored_rest<2, uint32_t>(fam, em[0], va[0], rest[0], tt[0], tt2[0]) &&
                 (int)(r >> 16) == wm_anchored_rest<2, uint32_t>(fam, em[1], va[1], rest[1], tt[1], tt2[1]);
        }
        if (!ok) { printf("FAIL pair bl=%d f=%d t=(%d,%d) (%d,%d)\n", bl, f, tt[0], tt2[0], tt[1], tt2[1]); return 1; }
        checked++;
    }
    //     ... 64-bit match masks (texts of up to 63 characters)
    for (int it = 0; it < 20000; it++) {
        const int bl = it % 2 ? 16 : 8 + rnd(9);
        std::vector<int> p(bl);
        for (auto &x : p) x = rnd(4);
        std::vector<int> t = p;
        for (int e = rnd(4); e > 0; e--) {
            const int op = rnd(3), pos = t.empty() ? 0 : rnd((int)t.size());
            if (op == 0 && !t.empty()) t[pos] = rnd(4);
            else if (op == 1) t.insert(t.begin() + pos, rnd(4));
            else if (!t.empty()) t.erase(t.begin() + pos);
        }
        for (int l = rnd(40); l > 0; l--) t.insert(t.begin(), rnd(4));
        for (int r = rnd(40); r > 0; r--) t.push_back(rnd(4));
        if ((int)t.size() > 63) t.resize(63);
        const int L = (int)t.size();
        uint64_t tlo, thi;
        planes(t, tlo, thi);
        EmArray<uint64_t> em;
        const uint64_t va = sct_valid_mask<uint64_t>(L), ve = va & ~(uint64_t)1, l1 = tlo << 1, h1 = thi << 1;
        em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
        uint32_t code = 0;
        for (int j = 0; j < bl; j++) code |= (uint32_t)p[j] << (2 * j);
        for (int f = 0; f < 6; f++) {
            const SctFamilyRT fam = sct_family_rt(f, bl);
            const uint32_t rest = sct_rest_of(fam, code);
            std::vector<std::pair<int, int>> pl;
            sct_for_each_placement_of(fam, L, [&](int a, int b) { pl.push_back({a, b}); });
            for (size_t i = 0; i + 1 < pl.size(); i += 2) {
                const uint32_t r = wm_anchored_pair<2>(fam, em, em, va, va, rest, rest, pl[i].first, pl[i].second, pl[i + 1].first, pl[i + 1].second);
                const int d0 = wm_anchored_rest<2, uint64_t>(fam, em, va, rest, pl[i].first, pl[i].second);
                const int d1 = wm_anchored_rest<2, uint64_t>(fam, em, va, rest, pl[i + 1].first, pl[i + 1].second);
                if ((int)(r & 0xFFFFu) != d0 || (int)(r >> 16) != d1) { printf("FAIL pair (64-bit) bl=%d f=%d L=%d\n", bl, f, L); return 1; }
                checked++;
            }
        }
    }
    //     ... the thirds families (static descriptions, <= 1 edit)
    for (int it = 0; it < 30000; it++) {
        std::vector<int> p(16);
        for (auto &x : p) x = rnd(4);
        std::vector<int> t = p;
        for (int e = rnd(3); e > 0; e--) {
            const int op = rnd(3), pos = t.empty() ? 0 : rnd((int)t.size());
            if (op == 0 && !t.empty()) t[pos] = rnd(4);
            else if (op == 1) t.insert(t.begin() + pos, rnd(4));
            else if (!t.empty()) t.erase(t.begin() + pos);
        }
        for (int l = rnd(7); l > 0; l--) t.insert(t.begin(), rnd(4));
        for (int r = rnd(7); r > 0; r--) t.push_back(rnd(4));
        if ((int)t.size() > 31) t.resize(31);
        const int L = (int)t.size();
        uint64_t tlo, thi;
        planes(t, tlo, thi);
        EmArray<uint32_t> em;
        const uint32_t va = sct_valid_mask<uint32_t>(L), ve = va & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
        em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
        uint32_t code = 0;
        for (int j = 0; j < 16; j++) code |= (uint32_t)p[j] << (2 * j);
        auto check = [&](auto fam) {
            const uint32_t rest = sct_rest_of(fam, code);
            std::vector<std::pair<int, int>> pl;
            sct_for_each_placement_of(fam, L, [&](int a, int b) { pl.push_back({a, b}); });
            for (size_t i = 0; i + 1 < pl.size(); i += 2) {
                const uint32_t r = wm_anchored_pair<1>(fam, em, em, va, va, rest, rest, pl[i].first, pl[i].second, pl[i + 1].first, pl[i + 1].second);
                if ((int)(r & 0xFFFFu) != wm_anchored_rest<1, uint32_t>(fam, em, va, rest, pl[i].first, pl[i].second) ||
                    (int)(r >> 16) != wm_anchored_rest<1, uint32_t>(fam, em, va, rest, pl[i + 1].first, pl[i + 1].second))
                    return false;
                checked++;
            }
            return true;
        };
        if (!check(SctFamily<SCT_F_T01>{}) || !check(SctFamily<SCT_F_T12>{}) || !check(SctFamily<SCT_F_T02>{})) {
            printf("FAIL pair (thirds)\n");
            return 1;
        }
    }
    //     ... and on planted copies (small distances are the common case there): every placement of the text
    for (int it = 0; it < 20000; it++) {
        const int bl = it % 2 ? 16 : 8 + rnd(9);
        std::vector<int> p(bl);
        for (auto &x : p) x = rnd(4);
        std::vector<int> t = p;
        for (int e = rnd(4); e > 0; e--) {
            const int op = rnd(3), pos = t.empty() ? 0 : rnd((int)t.size());
            if (op == 0 && !t.empty()) t[pos] = rnd(4);
            else if (op == 1) t.insert(t.begin() + pos, rnd(4));
            else if (!t.empty()) t.erase(t.begin() + pos);
        }
        for (int l = rnd(7); l > 0; l--) t.insert(t.begin(), rnd(4));
        for (int r = rnd(7); r > 0; r--) t.push_back(rnd(4));
        if ((int)t.size() > 31) t.resize(31);
        const int L = (int)t.size();
        uint64_t tlo, thi;
        planes(t, tlo, thi);
        EmArray<uint32_t> em;
        const uint32_t va = sct_valid_mask<uint32_t>(L), ve = va & ~1u, l1 = (uint32_t)tlo << 1, h1 = (uint32_t)thi << 1;
        em.m[0] = ~l1 & ~h1 & ve; em.m[1] = l1 & ~h1 & ve; em.m[2] = ~l1 & h1 & ve; em.m[3] = l1 & h1 & ve;
        uint32_t code = 0;
        for (int j = 0; j < bl; j++) code |= (uint32_t)p[j] << (2 * j);
        for (int f = 0; f < 6; f++) {
            const SctFamilyRT fam = sct_family_rt(f, bl);
            const uint32_t rest = sct_rest_of(fam, code);
            std::vector<std::pair<int, int>> pl;
            sct_for_each_placement_of(fam, L, [&](int a, int b) { pl.push_back({a, b}); });
            for (size_t i = 0; i + 1 < pl.size(); i += 2) {
                const uint32_t r = wm_anchored_pair<2>(fam, em, em, va, va, rest, rest, pl[i].first, pl[i].second, pl[i + 1].first, pl[i + 1].second);
                const int d0 = wm_anchored_rest<2, uint32_t>(fam, em, va, rest, pl[i].first, pl[i].second);
                const int d1 = wm_anchored_rest<2, uint32_t>(fam, em, va, rest, pl[i + 1].first, pl[i + 1].second);
                if ((int)(r & 0xFFFFu) != d0 || (int)(r >> 16) != d1) { printf("FAIL pair (planted) bl=%d f=%d\n", bl, f); return 1; }
                checked++;
            }
        }
    }
    // 1d. the whole filter path on the host over tables built by sct_build_tables: pattern sets with shared seed
    //     pieces (buckets of 1..8 entries, i.e. inline and long ones), exact duplicates, every barcode length 8..16
    for (int it = 0; it < 260; it++) {
        const int bl = it % 3 == 0 ? 8 + rnd(9) : 16;
        const int groups = 1 + rnd(24);
        std::vector<std::vector<int>> pats;
        for (int g = 0; g < groups; g++) {
            std::vector<int> base(bl);
            for (auto &x : base) x = rnd(4);
            const int members = 1 + rnd(8);
            for (int m = 0; m < members; m++) {
                std::vector<int> p = base;
                const int keep = rnd(3);  // which part stays shared: the first half, the second half, or everything but one character
                for (int j = 0; j < bl; j++)
                    if ((keep == 0 && j >= bl / 2) || (keep == 1 && j < bl / 2)) p[j] = rnd(4);
                if (keep == 2 && m > 0) p[rnd(bl)] = rnd(4);
                pats.push_back(p);
            }
        }
        if (pats.size() > 2) pats.push_back(pats[rnd((int)pats.size())]);  // an exact duplicate: two ids, one sequence
        std::vector<PatPlanes> pp(pats.size());
        for (size_t p = 0; p < pats.size(); p++) {
            uint64_t lo, hi;
            planes(pats[p], lo, hi);
            pp[p] = {(uint32_t)lo, (uint32_t)hi};
        }
        SctSeedTables T;
        const bool wide = bl == 16 && it % 20 == 1;  // now and then (64 MB of offsets and 512 MB of fixed buckets to clear each time)
        if (!sct_build_tables(pp.data(), (int)pp.size(), bl, false, wide, T)) { printf("FAIL build\n"); return 1; }
        for (int k = 0; k < 30; k++) {
            std::vector<int> t;
            const int n = rnd(32);
            if (rnd(5)) {
                const std::vector<int> &p = pats[rnd((int)pats.size())];
                t = p;
                for (int e = rnd(4); e > 0; e--) {
                    const int op = rnd(3), pos = t.empty() ? 0 : rnd((int)t.size());
                    if (op == 0 && !t.empty()) t[pos] = rnd(4);
                    else if (op == 1) t.insert(t.begin() + pos, rnd(4));
                    else if (!t.empty()) t.erase(t.begin() + pos);
                }
                for (int l = rnd(6); l > 0; l--) t.insert(t.begin(), rnd(4));
                for (int r = rnd(6); r > 0; r--) t.push_back(rnd(4));
                if ((int)t.size() > 31) t.resize(31);
            } else {
                t.resize(n);
                for (auto &x : t) x = rnd(4);
            }
            for (int sl = 0; sl < 2; sl++) {
                const bool ok = emulate_filter<0>(T, pats, bl, t, sl) && emulate_filter<1>(T, pats, bl, t, sl) &&
                                emulate_filter<2>(T, pats, bl, t, sl);
                if (!ok) { printf("FAIL filter emulation bl=%d patterns=%zu slots=%d\n", bl, pats.size(), sl); return 1; }
            }
            if (wide && !emulate_filter_wide(T, pats, t)) { printf("FAIL wide filter emulation patterns=%zu\n", pats.size()); return 1; }
            checked++;
        }
    }
    // 1e. a set large enough for the threaded build (sct_tables.h): all 25 families; every pattern sits once in the bucket of its
    //     key, ids ascending inside a bucket, the families' entries follow one another
    {
        const int np = 120000;
        std::vector<PatPlanes> pp((size_t)np);
        for (auto &q : pp) q = {(uint32_t)(rng() & 0xFFFFu), (uint32_t)(rng() & 0xFFFFu)};
        // keys with more entries than a fixed bucket holds: 40 patterns under one T01 key (characters 0..9), 20 under one T12 key
        // (5..15), 20 under one T02 key (0..4 and 10..15)
        for (int i = 1; i < 40; i++) pp[i] = {(pp[0].x & 0x03FFu) | (pp[i].x & 0xFC00u), (pp[0].y & 0x03FFu) | (pp[i].y & 0xFC00u)};
        for (int i = 41; i < 60; i++) pp[i] = {(pp[40].x & 0xFFE0u) | (pp[i].x & 0x001Fu), (pp[40].y & 0xFFE0u) | (pp[i].y & 0x001Fu)};
        for (int i = 61; i < 80; i++) pp[i] = {(pp[60].x & 0xFC1Fu) | (pp[i].x & 0x03E0u), (pp[60].y & 0xFC1Fu) | (pp[i].y & 0x03E0u)};
        SctSeedTables T;
        if (!sct_build_tables(pp.data(), np, 16, true, true, T) || T.nfam != SCT_NFAM_TABLES) { printf("FAIL threaded build\n"); return 1; }
        for (int f = 0; f < SCT_NFAM_TABLES; f++) {
            const uint32_t nk = sct_family_nkeys(f);
            const uint32_t *o = T.off.data() + T.off_base[f];
            if (o[0] != (uint32_t)((size_t)f * np) || o[nk] != (uint32_t)((size_t)(f + 1) * np)) { printf("FAIL threaded build: family %d range\n", f); return 1; }
            for (uint32_t k = 0; k < nk; k++) {
                if (o[k] > o[k + 1]) { printf("FAIL threaded build: offsets of family %d\n", f); return 1; }
                for (uint32_t g = o[k]; g < o[k + 1]; g++) {
                    const uint32_t p = T.ent[g];
                    const uint32_t key = f >= SCT_F_WIDE0 ? sct_wide_pattern_key(f - SCT_F_WIDE0, pp[p].x, pp[p].y) : sct_pattern_key(f, pp[p].x, pp[p].y);
                    if (key != k || (g > o[k] && T.ent[g - 1] >= p)) { printf("FAIL threaded build: entry of family %d\n", f); return 1; }
                    uint32_t code = 0;
                    for (int j = 0; j < 16; j++) code |= (((pp[p].x >> j) & 1u) | (((pp[p].y >> j) & 1u) << 1)) << (2 * j);
                    if (T.code[g] != code) { printf("FAIL threaded build: code of family %d\n", f); return 1; }
                }
            }
            checked += np;
        }
        // the fixed buckets of the thirds families repeat the rest characters of the entry tables, entry j of a key at place j;
        // a key with more entries than its bucket holds says so in its first place
        int overflowing = 0;
        for (int f = SCT_F_T01; f <= SCT_F_T02; f++) {
            const uint32_t nk = sct_family_nkeys(f), cap = SCT_FB_CAP(f == SCT_F_T01 ? 10 : 11);
            const uint32_t *o = T.off.data() + T.off_base[f];
            const uint16_t *fb = T.fb.data() + T.fb_base[f - SCT_F_T01];
            for (uint32_t k = 0; k < nk; k++) {
                const uint32_t n = o[k + 1] - o[k];
                for (uint32_t j = 0; j < cap; j++) {
                    uint32_t want = 0;
                    if (n > cap) { want = j == 0 ? SCT_FB_OVERFLOW : 0; overflowing += j == 0; }
                    else if (j < n) {
                        const uint32_t code = T.code[o[k] + j];
                        const uint32_t rest = f == SCT_F_T01 ? sct_rest_of(SctFamily<SCT_F_T01>{}, code)
                                              : f == SCT_F_T12 ? sct_rest_of(SctFamily<SCT_F_T12>{}, code) : sct_rest_of(SctFamily<SCT_F_T02>{}, code);
                        if (rest > 0x0FFFu) { printf("FAIL fixed buckets: rest of family %d has more than 12 bits\n", f); return 1; }
                        want = rest | SCT_FB_VALID;
                    }
                    if (fb[(size_t)k * cap + j] != want) { printf("FAIL fixed buckets: family %d key %u place %u\n", f, k, j); return 1; }
                }
            }
            checked += nk;
        }
        if (overflowing < 3) { printf("FAIL fixed buckets: the overflowing keys of the test set are not marked\n"); return 1; }
        // wide families: a bucket entry holds the six rest characters outside the key; with the half's two characters taken from the
        // key (plane bits -> codes, as push_wfb does) it is the quarter pair's rest again
        int wide_over = 0;
        for (int wf = 0; wf < SCT_NWIDE; wf++) {
            const SctFamilyRT base = sct_family_rt(sct_wide_base(wf), 16);
            const int pos = sct_wide_pos(wf);
            const uint32_t *o = T.off.data() + T.off_base[SCT_F_WIDE0 + wf];
            const uint16_t *fb = T.wfb.data() + ((size_t)wf * SCT_WIDE_KEYS) * SCT_WIDE_FB_CAP;
            for (uint32_t k = 0; k < SCT_WIDE_KEYS; k++) {
                const uint32_t n = o[k + 1] - o[k], kb4 = k >> 16;
                const uint32_t half4 = (kb4 & 1u) | (((kb4 >> 2) & 1u) << 1) | (((kb4 >> 1) & 1u) << 2) | (((kb4 >> 3) & 1u) << 3);
                for (uint32_t j = 0; j < SCT_WIDE_FB_CAP; j++) {
                    const uint32_t e = fb[(size_t)k * SCT_WIDE_FB_CAP + j];
                    if (n > SCT_WIDE_FB_CAP) {
                        wide_over += j == 0;
                        if (e != (j == 0 ? SCT_FB_OVERFLOW : 0u)) { printf("FAIL wide buckets: overflow mark, family %d\n", wf); return 1; }
                    } else if (j >= n) {
                        if (e != 0) { printf("FAIL wide buckets: stray entry, family %d\n", wf); return 1; }
                    } else if (!(e & SCT_FB_VALID) || sct_wide_rest16(e & 0x0FFFu, half4, pos) != (sct_rest_of(base, T.code[o[k] + j]) & 0xFFFFu)) {
                        printf("FAIL wide buckets: family %d key %u place %u\n", wf, k, j);
                        return 1;
                    }
                }
            }
            checked += SCT_WIDE_KEYS;
        }
        if (wide_over < 1) { printf("FAIL wide buckets: no overflowing key in the test set\n"); return 1; }
    }
    // 2. transposed Wu-Manber + seed completeness, bl = 16, texts 0..31, planted copies with 0..3 edits
    for (int it = 0; it < 300000 / scale; it++) {
        const int m = 16, alpha = (it % 7 == 0) ? 2 : 4;
        int n = rnd(32);
        std::vector<int> p(m), t;
        for (auto &x : p) x = rnd(alpha);
        if (rnd(4)) {
            std::vector<int> c(p);
            for (int k = rnd(4); k > 0; k--) {
                int kind = rnd(3), pos = rnd((int)c.size() + (kind == 1));
                if (kind == 0 && !c.empty()) c[std::min(pos, (int)c.size() - 1)] = rnd(4);
                else if (kind == 1) c.insert(c.begin() + pos, rnd(4));
                else if (!c.empty()) c.erase(c.begin() + std::min(pos, (int)c.size() - 1));
            }
            int left = rnd(8);
            for (int i = 0; i < left; i++) t.push_back(rnd(alpha));
            t.insert(t.end(), c.begin(), c.end());
            while ((int)t.size() < n) t.push_back(rnd(alpha));
            if ((int)t.size() > 31) t.resize(31);
        } else {
            t.resize(n);
            for (auto &x : t) x = rnd(alpha);
        }
        const int want = brute_infix(p, t);
        if (wm<0>(p, t) != std::min(want, 1)) { printf("FAIL wm0\n"); return 1; }
        if (wm<1>(p, t) != std::min(want, 2)) { printf("FAIL wm1\n"); return 1; }
        if (wm<2>(p, t) != std::min(want, 3)) { printf("FAIL wm2 want=%d got=%d n=%zu\n", want, wm<2>(p, t), t.size()); return 1; }
        if (anchored_thirds<0>(p, t) != std::min(want, 1)) { printf("FAIL thirds0 want=%d got=%d\n", want, anchored_thirds<0>(p, t)); return 1; }
        if (anchored_thirds<1>(p, t) != std::min(want, 2)) { printf("FAIL thirds1 want=%d got=%d n=%zu\n", want, anchored_thirds<1>(p, t), t.size()); return 1; }
        if (anchored<0>(p, t) != std::min(want, 1)) { printf("FAIL anchored0 want=%d got=%d\n", want, anchored<0>(p, t)); return 1; }
        if (anchored<1>(p, t) != std::min(want, 2)) { printf("FAIL anchored1 want=%d got=%d\n", want, anchored<1>(p, t)); return 1; }
        if (anchored<2>(p, t) != std::min(want, 3)) { printf("FAIL anchored2 want=%d got=%d n=%zu\n", want, anchored<2>(p, t), t.size()); return 1; }
        if (!check_wide(p, t, want)) return 1;
        if (want <= 0 && !seeds_cover<0>(p, t)) { printf("FAIL seeds0\n"); return 1; }
        if (want <= 1 && !seeds_cover<1>(p, t)) { printf("FAIL seeds1\n"); return 1; }
        if (want <= 2 && !seeds_cover<2>(p, t)) { printf("FAIL seeds2 want=%d\n", want); return 1; }
        checked++;
    }
    // 2b. the 64-bit variant of the anchored automaton: texts of 0..63 characters
    for (int it = 0; it < 120000 / scale; it++) {
        const int m = 16, alpha = (it % 7 == 0) ? 2 : 4;
        int n = rnd(64);
        std::vector<int> p(m), t;
        for (auto &x : p) x = rnd(alpha);
        std::vector<int> c(p);
        for (int k = rnd(4); k > 0; k--) {
            int kind = rnd(3), pos = rnd((int)c.size() + (kind == 1));
            if (kind == 0 && !c.empty()) c[std::min(pos, (int)c.size() - 1)] = rnd(4);
            else if (kind == 1) c.insert(c.begin() + pos, rnd(4));
            else if (!c.empty()) c.erase(c.begin() + std::min(pos, (int)c.size() - 1));
        }
        int left = rnd(45);
        for (int i = 0; i < left; i++) t.push_back(rnd(alpha));
        if (rnd(5)) t.insert(t.end(), c.begin(), c.end());
        while ((int)t.size() < n) t.push_back(rnd(alpha));
        if ((int)t.size() > 63) t.resize(63);
        const int want = brute_infix(p, t);
        if (anchored_thirds<1, uint64_t>(p, t) != std::min(want, 2)) { printf("FAIL thirds64/1\n"); return 1; }
        if (anchored<0, uint64_t>(p, t) != std::min(want, 1)) { printf("FAIL anchored64/0\n"); return 1; }
        if (anchored<1, uint64_t>(p, t) != std::min(want, 2)) { printf("FAIL anchored64/1\n"); return 1; }
        if (anchored<2, uint64_t>(p, t) != std::min(want, 3)) { printf("FAIL anchored64/2 want=%d got=%d n=%zu\n", want, anchored<2, uint64_t>(p, t), t.size()); return 1; }
        if (t.size() <= 31 && anchored<2, uint32_t>(p, t) != std::min(want, 3)) { printf("FAIL anchored32 vs 64\n"); return 1; }
        if (!check_wide<uint64_t>(p, t, want)) return 1;
        checked++;
    }
    // 2c. run-time piece boundaries: barcode lengths 8..16, texts 0..31
    for (int it = 0; it < 200000 / scale; it++) {
        const int m = 8 + rnd(9), alpha = (it % 7 == 0) ? 2 : 4;
        int n = rnd(32);
        std::vector<int> p(m), t;
        for (auto &x : p) x = rnd(alpha);
        std::vector<int> c(p);
        for (int k = rnd(4); k > 0; k--) {
            int kind = rnd(3), pos = rnd((int)c.size() + (kind == 1));
            if (kind == 0 && !c.empty()) c[std::min(pos, (int)c.size() - 1)] = rnd(4);
            else if (kind == 1) c.insert(c.begin() + pos, rnd(4));
            else if (!c.empty()) c.erase(c.begin() + std::min(pos, (int)c.size() - 1));
        }
        int left = rnd(8);
        for (int i = 0; i < left; i++) t.push_back(rnd(alpha));
        if (rnd(5)) t.insert(t.end(), c.begin(), c.end());
        while ((int)t.size() < n) t.push_back(rnd(alpha));
        if ((int)t.size() > 31) t.resize(31);
        const int want = brute_infix(p, t);
        if (anchored_rt<0>(p, t) != std::min(want, 1)) { printf("FAIL rt0 m=%d\n", m); return 1; }
        if (anchored_rt<1>(p, t) != std::min(want, 2)) { printf("FAIL rt1 m=%d want=%d got=%d\n", m, want, anchored_rt<1>(p, t)); return 1; }
        if (anchored_rt<2>(p, t) != std::min(want, 3)) { printf("FAIL rt2 m=%d want=%d got=%d n=%zu\n", m, want, anchored_rt<2>(p, t), t.size()); return 1; }
        if (m == 16 && anchored_rt<2>(p, t) != anchored<2>(p, t)) { printf("FAIL rt vs static\n"); return 1; }
        checked++;
    }
    // 3. exhaustive seed completeness around one pattern: every text obtained by <= 2 edits of it
    {
        std::vector<int> p(16);
        for (auto &x : p) x = rnd(4);
        std::vector<std::vector<int>> level{p}, next;
        long n_texts = 0;
        for (int depth = 0; depth < 2; depth++) {
            next.clear();
            for (auto &s : level) {
                for (size_t pos = 0; pos <= s.size(); pos++) {
                    for (int b = 0; b < 4; b++) {
                        std::vector<int> a(s);
                        a.insert(a.begin() + pos, b);
                        next.push_back(a);
                        if (pos < s.size()) { a = s; a[pos] = b; next.push_back(a); }
                    }
                    if (pos < s.size()) { std::vector<int> a(s); a.erase(a.begin() + pos); next.push_back(a); }
                }
            }
            std::sort(next.begin(), next.end());
            next.erase(std::unique(next.begin(), next.end()), next.end());
            for (auto &s : next) {
                std::vector<int> t;
                for (int i = 0; i < 3; i++) t.push_back(rnd(4));
                t.insert(t.end(), s.begin(), s.end());
                for (int i = 0; i < 3; i++) t.push_back(rnd(4));
                const int want = brute_infix(p, t);
                const int w2 = brute_infix(p, s);  // the bare edited string with nothing around it
                if (want <= 2 && !seeds_cover<2>(p, t)) { printf("FAIL seeds2 exhaustive\n"); return 1; }
                if (want <= 1 && !seeds_cover<1>(p, t)) { printf("FAIL seeds1 exhaustive\n"); return 1; }
                if (wm<2>(p, t) != std::min(want, 3)) { printf("FAIL wm2 exhaustive\n"); return 1; }
                if (anchored<2>(p, t) != std::min(want, 3)) { printf("FAIL anchored2 exhaustive\n"); return 1; }
                if (anchored<1>(p, t) != std::min(want, 2)) { printf("FAIL anchored1 exhaustive\n"); return 1; }
                if (anchored_thirds<1>(p, t) != std::min(want, 2)) { printf("FAIL thirds1 exhaustive\n"); return 1; }
                if (!check_wide(p, t, want)) return 1;
                if (s.size() <= 31 && !check_wide(p, s, w2)) return 1;
                if (s.size() <= 31 && anchored<2>(p, s) != std::min(w2 <= 2 ? w2 : 3, 3)) { printf("FAIL anchored2 bare\n"); return 1; }
                if (s.size() <= 31 && w2 <= 2 && !seeds_cover<2>(p, s)) { printf("FAIL seeds2 bare\n"); return 1; }
                n_texts++;
            }
            level = next;
        }
        checked += n_texts;
    }
    // 9. the partitioned join's compare == the anchored automaton at max_err = 1, thirds families: planted patterns with 0..3 edits
    //    and texts over small alphabets (so that keys match by chance, at placements the planted copy does not explain)
    {
        long n_join = 0;
        bool ok = true;
        for (int it = 0; it < 400000 / scale && ok; it++) {
            const int alpha = 1 + rnd(4);
            std::vector<int> p(16), t;
            for (auto &x : p) x = rnd(alpha);
            const int pre = rnd(9), post = rnd(9);
            for (int i = 0; i < pre; i++) t.push_back(rnd(alpha));
            std::vector<int> core = p;
            for (int e = rnd(4); e > 0 && !core.empty(); e--) {
                const int k = rnd(3), at = rnd((int)core.size() + (k == 1 ? 1 : 0));
                if (k == 0 && at < (int)core.size()) core[at] = rnd(alpha);
                else if (k == 1) core.insert(core.begin() + at, rnd(alpha));
                else if (at < (int)core.size()) core.erase(core.begin() + at);
            }
            t.insert(t.end(), core.begin(), core.end());
            for (int i = 0; i < post; i++) t.push_back(rnd(alpha));
            if (t.size() > 31) t.resize(31);
            n_join += join_vs_automaton<SCT_F_T01>(p, t, &ok);
            n_join += join_vs_automaton<SCT_F_T12>(p, t, &ok);
            n_join += join_vs_automaton<SCT_F_T02>(p, t, &ok);
        }
        if (!ok) return 1;
        if (n_join < 100000) { printf("FAIL join compare: only %ld placements with a matching key\n", n_join); return 1; }
        checked += n_join;
    }
    // 10. restricted starts (the reference's raw-byte prefix rule, scTagger.py:681) and rows cut into windows: planted copies
    //     with 0..3 edits, bad bytes sprinkled over the text, k = 0..3; 32- and 64-bit masks, barcode lengths 8..16
    {
        auto plant = [&](int bl, int lmin, int lmax, std::vector<int> &p, std::vector<int> &t, std::vector<int> &badv) {
            const int alpha = rnd(5) == 0 ? 2 : 4;
            p.resize(bl);
            for (auto &x : p) x = rnd(alpha);
            const int L = lmin + rnd(lmax - lmin + 1);
            t.resize(L);
            for (auto &x : t) x = rnd(alpha);
            std::vector<int> core = p;
            for (int e = rnd(4); e > 0 && !core.empty(); e--) {
                const int kind = rnd(3), at = rnd((int)core.size() + (kind == 1 ? 1 : 0));
                if (kind == 0 && at < (int)core.size()) core[at] = rnd(alpha);
                else if (kind == 1) core.insert(core.begin() + at, rnd(alpha));
                else if (at < (int)core.size()) core.erase(core.begin() + at);
            }
            if ((int)core.size() <= L) {
                const int at = rnd(L - (int)core.size() + 1);
                std::copy(core.begin(), core.end(), t.begin() + at);
            }
            badv.assign(L, 0);
            const int nbad = rnd(4) == 0 ? 0 : 1 + rnd(3);
            for (int i = 0; i < nbad && L > 0; i++) {
                const int at = rnd(L);
                badv[at] = 1;
                t[at] = 0;   // map_char: a byte outside ACGT matches as A
            }
        };
        for (int it = 0; it < 120000 / scale; it++) {
            const int bl = it % 3 == 0 ? 8 + rnd(9) : 16, k = rnd(4);
            std::vector<int> p, t, badv;
            plant(bl, bl - 2, it % 2 ? 31 : 63, p, t, badv);
            uint64_t bad = 0;
            for (size_t i = 0; i < t.size(); i++) bad |= (uint64_t)badv[i] << i;
            bool ok;
            if (t.size() <= 31) ok = check_restricted<0, uint32_t>(p, t, bad, k) && check_restricted<1, uint32_t>(p, t, bad, k) && check_restricted<2, uint32_t>(p, t, bad, k);
            else ok = check_restricted<0, uint64_t>(p, t, bad, k) && check_restricted<1, uint64_t>(p, t, bad, k) && check_restricted<2, uint64_t>(p, t, bad, k);
            if (!ok) return 1;
            checked++;
        }
        for (int it = 0; it < 30000 / scale; it++) {
            const int bl = it % 3 == 0 ? 8 + rnd(9) : 16, k = rnd(4);
            std::vector<int> p, t, badv;
            plant(bl, 40, 400, p, t, badv);
            if (!check_windows<1>(p, t, badv, k) || !check_windows<2>(p, t, badv, k)) return 1;
            checked++;
        }
    }
    printf("OK %ld cases\n", checked);
    return 0;
}
