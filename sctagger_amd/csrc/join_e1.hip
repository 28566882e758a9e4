// join_e1.hip — the partitioned join of the thirds scheme at max_err = 1 (k_join.hip.h) and the host functions that launch it.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "../../include/sctagger_hip.h"
#include "sct_core.h"
#include "sct_kernels_common.h"
#include "k_decode.hip.h"
#include "k_filter.hip.h"
#include "filter_launch.h"
#include "k_join.hip.h"

namespace {
template <typename OffT>
void pass(const JoinArgs<OffT> &ja, hipStream_t st) {
    const int64_t rows = ja.row1 - ja.row0;
    if (rows <= 0) return;
    const int n_prod = (int)((rows + 255) / 256);
    const int groups = (n_prod + JOIN_GROUP - 1) / JOIN_GROUP;
    k_join_part0<OffT><<<(unsigned)n_prod, 256, 0, st>>>(ja);
    k_join_probe<SCT_F_T01, OffT><<<(unsigned)(JOIN_P0 * groups), 256, 0, st>>>(ja, n_prod, groups);
    k_join_part1<OffT><<<(unsigned)n_prod, 256, 0, st>>>(ja);
    k_join_probe<SCT_F_T12, OffT><<<(unsigned)(32 * groups), 256, 0, st>>>(ja, n_prod, groups);
    k_join_probe<SCT_F_T02, OffT><<<(unsigned)(32 * groups), 256, 0, st>>>(ja, n_prod, groups);
}
template <typename OffT>
void finish(const JoinArgs<OffT> &ja, hipStream_t st) {
    if (ja.fa.n <= 0) return;
    k_join_finish<OffT><<<(unsigned)((ja.fa.n + 255) / 256), 256, 0, st>>>(ja);
}
}  // namespace

void sct_launch_join_pass(const JoinArgs<int64_t> &ja, hipStream_t st) { pass(ja, st); }
void sct_launch_join_pass(const JoinArgs<uint32_t> &ja, hipStream_t st) { pass(ja, st); }
void sct_launch_join_finish(const JoinArgs<int64_t> &ja, hipStream_t st) { finish(ja, st); }
void sct_launch_join_finish(const JoinArgs<uint32_t> &ja, hipStream_t st) { finish(ja, st); }
