// filter_tu.hip.h — body of the translation units filter_e0/1/2.hip (main and retry launch) and filter_sp_e0/1/2.hip (special
// launch, SCT_FILTER_SPECIAL = 1): the instantiations of k_filter for ONE error budget E (SCT_FILTER_E) and the host function that picks among them.  The seed-filter kernel
// is by far the largest part of the library's device code; cutting it by E lets `make -j` build the parts side by side.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "../../include/sctagger_hip.h"
#include "sct_core.h"
#include "sct_kernels_common.h"
#include "k_decode.hip.h"
#include "k_filter.hip.h"
#include "filter_launch.h"

#ifndef SCT_FILTER_E
#error "define SCT_FILTER_E (0, 1 or 2) before including filter_tu.hip.h"
#endif
#ifndef SCT_FILTER_SPECIAL
#define SCT_FILTER_SPECIAL 0     // 1: the instantiations of the special launch (translation units filter_sp_e0/1/2.hip)
#endif

namespace {
template <typename OffT>
void launch(const SctFilterVariant v, const FilterArgs<OffT> &fa, unsigned g, unsigned g64, hipStream_t st) {
    constexpr int E = SCT_FILTER_E;
    // both mask widths are launched over all rows; a wave leaves a group of rows at once if it is not its kind (k_filter).
    // g64 == 0: the 64-bit instantiation is not launched at all — the caller speculates that no row of the batch has more than 31
    // characters; the 32-bit main launch marks CTR_NEED64 if one does, and the batch is then done again with both.
    // g = grid of the 32-bit instantiation, g64 = the (smaller, striding) grid of the 64-bit one; the striding launches
    // (every one but the main launch's 32-bit instantiation) run one wave per block (FilterWavesPerBlock)
#if !SCT_FILTER_SPECIAL
    if (!v.retry && !v.special) {
        if (v.rt) {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_SLOTS, HMAX, 64, true, false, OffT><<<g, 256, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_SLOTS, HMAX, 64, true, false, OffT><<<g64, 64, 0, st>>>(fa);
        } else if (v.thirds) {
            if constexpr (E <= 1) {
                if (!v.only64) k_filter<E, uint32_t, SCHEME_THIRDS, K_SLOTS, HMAX, 64, false, false, OffT><<<g, 256, 0, st>>>(fa);
                if (g64) k_filter<E, uint64_t, SCHEME_THIRDS, K_SLOTS, HMAX, 64, false, false, OffT><<<g64, 64, 0, st>>>(fa);
            }
        } else if (v.wide) {
            if constexpr (E == 2) {
                k_filter<E, uint32_t, SCHEME_WIDE, K_SLOTS, HMAX, 64, false, false, OffT><<<g, 256, 0, st>>>(fa);
                if (g64) k_filter<E, uint64_t, SCHEME_WIDE, K_SLOTS, HMAX, 64, false, false, OffT><<<g64, 64, 0, st>>>(fa);
            }
        } else if (v.slots) {
            k_filter<E, uint32_t, SCHEME_QUARTERS, HMAX_SL, HMAX_SL, 64, false, true, OffT><<<g, 256, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, HMAX_SL, HMAX_SL, 64, false, true, OffT><<<g64, 64, 0, st>>>(fa);
        } else {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_SLOTS, HMAX, 64, false, false, OffT><<<g, 256, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_SLOTS, HMAX, 64, false, false, OffT><<<g64, 64, 0, st>>>(fa);
        }
    } else {
        if (v.rt) {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, true, false, OffT><<<g, 64, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, true, false, OffT><<<g64, 64, 0, st>>>(fa);
        } else if (v.thirds) {
            if constexpr (E <= 1) {
                k_filter<E, uint32_t, SCHEME_THIRDS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT><<<g, 64, 0, st>>>(fa);
                if (g64) k_filter<E, uint64_t, SCHEME_THIRDS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT><<<g64, 64, 0, st>>>(fa);
            }
        } else if (v.wide) {
            if constexpr (E == 2) {
                k_filter<E, uint32_t, SCHEME_WIDE, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT><<<g, 64, 0, st>>>(fa);
                if (g64) k_filter<E, uint64_t, SCHEME_WIDE, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT><<<g64, 64, 0, st>>>(fa);
            }
        } else if (v.slots) {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_SL, RETRY_RPW_SL, false, true, OffT><<<g, 64, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_SL, RETRY_RPW_SL, false, true, OffT><<<g64, 64, 0, st>>>(fa);
        } else {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT><<<g, 64, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT><<<g64, 64, 0, st>>>(fa);
        }
    }
#else
    if (v.special16) {
        k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_SL, SPECIAL_RPW_SL, false, true, OffT, true><<<g, 64, 0, st>>>(fa);
        if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_SL, SPECIAL_RPW_SL, false, true, OffT, true><<<g64, 64, 0, st>>>(fa);
    } else if (v.special) {
        if (v.rt) {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, true, false, OffT, true><<<g, 64, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, true, false, OffT, true><<<g64, 64, 0, st>>>(fa);
        } else if (v.thirds) {
            if constexpr (E <= 1) {
                k_filter<E, uint32_t, SCHEME_THIRDS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT, true><<<g, 64, 0, st>>>(fa);
                if (g64) k_filter<E, uint64_t, SCHEME_THIRDS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT, true><<<g64, 64, 0, st>>>(fa);
            }
        } else if (v.wide) {
            if constexpr (E == 2) {
                k_filter<E, uint32_t, SCHEME_WIDE, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT, true><<<g, 64, 0, st>>>(fa);
                if (g64) k_filter<E, uint64_t, SCHEME_WIDE, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT, true><<<g64, 64, 0, st>>>(fa);
            }
        } else if (v.slots) {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_SL, RETRY_RPW_SL, false, true, OffT, true><<<g, 64, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_SL, RETRY_RPW_SL, false, true, OffT, true><<<g64, 64, 0, st>>>(fa);
        } else {
            k_filter<E, uint32_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT, true><<<g, 64, 0, st>>>(fa);
            if (g64) k_filter<E, uint64_t, SCHEME_QUARTERS, K_RETRY, H_RETRY_NS, RETRY_RPW_NS, false, false, OffT, true><<<g64, 64, 0, st>>>(fa);
        }
    }
#endif
}
}  // namespace

#define SCT_CAT2(a, b) a##b
#define SCT_CAT(a, b) SCT_CAT2(a, b)
#if SCT_FILTER_SPECIAL
#define SCT_LAUNCH_NAME SCT_CAT(sct_launch_filter_sp_e, SCT_FILTER_E)
#else
#define SCT_LAUNCH_NAME SCT_CAT(sct_launch_filter_e, SCT_FILTER_E)
#endif
void SCT_LAUNCH_NAME(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned g, unsigned g64, hipStream_t st) { launch(v, fa, g, g64, st); }
void SCT_LAUNCH_NAME(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned g, unsigned g64, hipStream_t st) { launch(v, fa, g, g64, st); }
