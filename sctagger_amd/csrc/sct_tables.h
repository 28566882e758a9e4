// sct_tables.h — host-side construction of the seed tables of the filter kernel (plain C++, no HIP): used by
// sct_set_barcodes and, in tests/host_core_check.cpp, by a CPU emulation of the filter path over the same tables.
//
// Per family f that is built (sct_core.h; the six quarter families always, the thirds and the wide families on request): an
// offsets array of nkeys(f) + 1 entries holding ABSOLUTE indices into the entry tables (the families' entries follow one
// another, np each), the entries as pattern ids (ascending inside a bucket) and as 2-bit codes, and
// — for the thirds families — fixed buckets (the rest characters of up to SCT_FB_CAP entries per key in one aligned block), and
// — for the six quarter families — one 8-byte bucket slot per key (sct_slot_*), indexed like the offsets, and the entry
// index at the start of every group of 64 keys (an inline slot says where its entries start inside its group).
#pragma once
#include <stdint.h>

#include <atomic>
#include <thread>
#include <vector>

#include "sct_core.h"

struct SctSeedTables {
    int nfam = 0;
    uint32_t off_base[SCT_NFAM_TABLES] = {};  // where family f's offsets (and slots) start (families that are not built: 0)
    std::vector<uint32_t> off, ent, code;
    std::vector<uint64_t> slot;
    std::vector<uint32_t> grp;             // per quarter family SCT_GRP_PER_FAM words: entry index at the first key of every group of 64 keys
    double long_share = 1.0;               // share of the quarter families' keys whose bucket does not fit a slot
    uint32_t max_bucket = 0;               // entries in the longest bucket of the quarter families
    std::vector<uint16_t> fb;              // thirds families: fixed buckets (SCT_FB_CAP entries per key), family after family
    uint32_t fb_base[3] = {0, 0, 0};       // where each thirds family's buckets start in fb (entries)
    std::vector<uint16_t> wfb;             // wide families: fixed buckets, SCT_WIDE_FB_CAP entries per key, family wf at wf << 24
};

// pat[p].x / .y: bit-planes of pattern p (np patterns of bl characters, 8 <= bl <= 16); thirds / wide: also build the
// 20/22-bit families of max_err <= 1 / the sixteen 20-bit families of max_err = 2 (bl == 16 only).  Returns false if the
// tables would not be addressable with 32-bit indices.
template <typename Pat>
static inline bool sct_build_tables(const Pat *pat, int np, int bl, bool thirds, bool wide, SctSeedTables &T) {
    auto built = [&](int f) { return f < SCT_NFAM || (f < SCT_NFAM_ALL ? thirds : wide); };
    T.nfam = 0;
    size_t off_total = 0;
    for (int f = 0; f < SCT_NFAM_TABLES; f++) {
        T.off_base[f] = 0;
        if (!built(f)) continue;
        if (off_total + sct_family_nkeys(f) + 1 > 0xFFFFFFFFull || (uint64_t)(T.nfam + 1) * (uint64_t)np > 0x3FFFFFFFull) return false;
        T.off_base[f] = (uint32_t)off_total;
        off_total += (size_t)sct_family_nkeys(f) + 1;
        T.nfam++;
    }
    const size_t npn = np ? (size_t)np : 1;
    T.off.assign(off_total, 0u);
    T.ent.assign((size_t)T.nfam * npn, 0u);
    T.code.assign((size_t)T.nfam * npn, 0u);
    std::vector<uint32_t> inter(npn);
    for (int p = 0; p < np; p++) {
        uint32_t v = 0;
        for (int j = 0; j < bl; j++) v |= (((pat[p].x >> j) & 1u) | (((pat[p].y >> j) & 1u) << 1)) << (2 * j);
        inter[p] = v;
    }
    // the families are independent (disjoint parts of off / ent / code): a few host threads share them out
    std::vector<int> fams;
    for (int f = 0; f < SCT_NFAM_TABLES; f++)
        if (built(f)) fams.push_back(f);
    auto build_family = [&](int rank) {
        const int f = fams[(size_t)rank];
        std::vector<uint32_t> keys(npn), cur;
        const uint32_t nk = sct_family_nkeys(f);
        uint32_t *o = T.off.data() + T.off_base[f];
        if (f >= SCT_F_WIDE0) {
            for (int p = 0; p < np; p++) keys[p] = sct_wide_pattern_key(f - SCT_F_WIDE0, pat[p].x, pat[p].y);
        } else if (f >= SCT_NFAM) {
            for (int p = 0; p < np; p++) keys[p] = sct_pattern_key(f, pat[p].x, pat[p].y);
        } else {  // quarter families: piece boundaries from the barcode length
            const SctFamilyRT fam = sct_family_rt(f, bl);
            for (int p = 0; p < np; p++) keys[p] = sct_key_of<uint32_t>(fam, pat[p].x, pat[p].y, fam.PA, fam.PB);
        }
        for (int p = 0; p < np; p++) o[keys[p] + 1]++;
        o[0] = (uint32_t)((size_t)rank * (size_t)np);
        for (uint32_t k = 0; k < nk; k++) o[k + 1] += o[k];
        cur.assign(o, o + nk);
        for (int p = 0; p < np; p++) {
            const uint32_t at = cur[keys[p]]++;
            T.ent[at] = (uint32_t)p;
            T.code[at] = inter[p];
        }
    };
    {
        unsigned nt = np >= 100000 ? std::thread::hardware_concurrency() : 1u;
        nt = nt < 1u ? 1u : (nt > 8u ? 8u : nt);
        if (nt > fams.size()) nt = (unsigned)fams.size();
        std::atomic<int> next{0};
        auto worker = [&]() {
            for (int r = next.fetch_add(1); r < (int)fams.size(); r = next.fetch_add(1)) build_family(r);
        };
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; i++) th.emplace_back(worker);
        worker();
        for (auto &t : th) t.join();
    }
    // fixed buckets of the thirds families: the rest characters of up to SCT_FB_CAP entries per key in one aligned block, in the
    // order of the entry tables (entry j of the block is entry off[key] + j)
    T.fb.clear();
    if (thirds) {
        size_t total = 0;
        for (int f = SCT_NFAM; f < SCT_NFAM_ALL; f++) {
            const int keych = f == SCT_F_T01 ? 10 : 11;
            if (total > 0xFFFFFFFFull) return false;
            T.fb_base[f - SCT_NFAM] = (uint32_t)total;
            total += (size_t)sct_family_nkeys(f) * SCT_FB_CAP(keych);
        }
        T.fb.assign(total, (uint16_t)0);
        auto fill = [&](int f) {
            const int keych = f == SCT_F_T01 ? 10 : 11;
            const uint32_t cap = SCT_FB_CAP(keych), nk = sct_family_nkeys(f);
            const uint32_t *o = T.off.data() + T.off_base[f];
            uint16_t *out = T.fb.data() + T.fb_base[f - SCT_NFAM];
            for (uint32_t k = 0; k < nk; k++) {
                const uint32_t b = o[k], n = o[k + 1] - b;
                uint16_t *blk = out + (size_t)k * cap;
                if (n > cap) {
                    blk[0] = (uint16_t)SCT_FB_OVERFLOW;
                    continue;
                }
                for (uint32_t j = 0; j < n; j++) {
                    uint32_t rest;
                    if (f == SCT_F_T01) rest = sct_rest_of(SctFamily<SCT_F_T01>{}, T.code[b + j]);
                    else if (f == SCT_F_T12) rest = sct_rest_of(SctFamily<SCT_F_T12>{}, T.code[b + j]);
                    else rest = sct_rest_of(SctFamily<SCT_F_T02>{}, T.code[b + j]);
                    blk[j] = (uint16_t)((rest & 0x0FFFu) | SCT_FB_VALID);
                }
            }
        };
        std::vector<std::thread> th;
        if (np >= 100000) {
            for (int f = SCT_NFAM + 1; f < SCT_NFAM_ALL; f++) th.emplace_back(fill, f);
            fill(SCT_NFAM);
            for (auto &t : th) t.join();
        } else {
            for (int f = SCT_NFAM; f < SCT_NFAM_ALL; f++) fill(f);
        }
    }
    // fixed buckets of the wide families: the six rest characters that are not in the key (sct_wide_rest12)
    T.wfb.clear();
    if (wide) {
        T.wfb.assign((size_t)SCT_NWIDE * SCT_WIDE_KEYS * SCT_WIDE_FB_CAP, (uint16_t)0);
        auto fill = [&](int wf) {
            const int f = SCT_F_WIDE0 + wf, pos = sct_wide_pos(wf);
            const SctFamilyRT base = sct_family_rt(sct_wide_base(wf), 16);
            const uint32_t *o = T.off.data() + T.off_base[f];
            uint16_t *out = T.wfb.data() + ((size_t)wf * SCT_WIDE_KEYS) * SCT_WIDE_FB_CAP;
            for (uint32_t k = 0; k < SCT_WIDE_KEYS; k++) {
                const uint32_t b = o[k], n = o[k + 1] - b;
                uint16_t *blk = out + (size_t)k * SCT_WIDE_FB_CAP;
                if (n > SCT_WIDE_FB_CAP) {
                    blk[0] = (uint16_t)SCT_FB_OVERFLOW;
                    continue;
                }
                for (uint32_t j = 0; j < n; j++)
                    blk[j] = (uint16_t)(sct_wide_rest12(sct_rest_of(base, T.code[b + j]) & 0xFFFFu, pos) | SCT_FB_VALID);
            }
        };
        if (np >= 100000) {
            unsigned nt = std::thread::hardware_concurrency();
            nt = nt < 1u ? 1u : (nt > 8u ? 8u : nt);
            std::atomic<int> next{0};
            auto worker = [&]() {
                for (int wf = next.fetch_add(1); wf < SCT_NWIDE; wf = next.fetch_add(1)) fill(wf);
            };
            std::vector<std::thread> th;
            for (unsigned i = 1; i < nt; i++) th.emplace_back(worker);
            worker();
            for (auto &t : th) t.join();
        } else {
            for (int wf = 0; wf < SCT_NWIDE; wf++) fill(wf);
        }
    }
    size_t slot_total = 0;
    for (int f = 0; f < SCT_NFAM; f++) slot_total += (size_t)sct_family_nkeys(f) + 1;
    T.slot.assign(slot_total, 0ull);
    T.grp.assign((size_t)SCT_NFAM * SCT_GRP_PER_FAM, 0u);
    T.max_bucket = 0;
    size_t long_keys = 0;
    for (int f = 0; f < SCT_NFAM; f++) {
        const SctFamilyRT fam = sct_family_rt(f, bl);
        const uint32_t *o = T.off.data() + T.off_base[f];
        uint64_t *sl = T.slot.data() + T.off_base[f];
        uint32_t *grp = T.grp.data() + (size_t)f * SCT_GRP_PER_FAM;
        for (uint32_t g = 0; g < SCT_GRP_PER_FAM; g++) grp[g] = o[(size_t)g << SCT_GRP_SHIFT < SCT_KEYS ? (size_t)g << SCT_GRP_SHIFT : SCT_KEYS];
        for (uint32_t k = 0; k < sct_family_nkeys(f); k++) {
            const uint32_t cnt = o[k + 1] - o[k];
            if (cnt > T.max_bucket) T.max_bucket = cnt;
            const uint32_t within = o[k] - grp[k >> SCT_GRP_SHIFT];
            // (a short bucket behind thousands of entries of its own group cannot say where it starts: it is listed as a long one)
            if (cnt <= 3 && within <= SCT_SLOT_WITHIN_MAX) {
                uint32_t rest[3] = {0, 0, 0};
                for (uint32_t j = 0; j < cnt; j++) rest[j] = sct_rest_of(fam, T.code[o[k] + j]);
                sl[k] = sct_slot_inline(cnt, rest, within);
            } else {
                long_keys++;
                sl[k] = sct_slot_long(o[k], cnt);
            }
        }
    }
    T.long_share = (double)long_keys / (double)slot_total;
    return true;
}
