// filter_launch.h — host entry points of the seed-filter translation units (filter_e0/1/2.hip), called by sctagger_hip.hip.
#pragma once
struct SctFilterVariant {
    bool retry;   // the retry launch (larger lists, 4 listed rows per wave) instead of the main one
    bool rt;      // barcode length other than 16: piece boundaries computed at run time
    bool slots;   // lookups read the 8-byte bucket slots
    bool thirds;  // the 20/22-bit families of very large barcode sets (max_err <= 1)
    bool wide;    // the sixteen 20-bit families of large barcode sets (max_err = 2)
    bool only64;  // main launch: the 64-bit instantiation only (the partitioned join, k_join.hip.h, has done the 32-bit rows)
    bool special; // the special launch: the retry launch's shape over list_special, restricted starts and rows cut into windows
    bool special16; // ... its first stage with bucket slots: 16 rows per wave (the rows it hands on take the 4-row shape)
};
template <typename OffT>
struct FilterArgs;
void sct_launch_filter_e0(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e0(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e1(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e1(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e2(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e2(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
// the special launch's instantiations (v.special), translation units filter_sp_e0/1/2.hip
void sct_launch_filter_sp_e0(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_sp_e0(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_sp_e1(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_sp_e1(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_sp_e2(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_sp_e2(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);

// the partitioned join of the thirds scheme at max_err = 1 (k_join.hip.h, translation unit join_e1.hip): the passes for rows
// [row0, row1) — tuples, probes —, and the finish over all rows
template <typename OffT>
struct JoinArgs;
void sct_launch_join_pass(const JoinArgs<int64_t> &ja, hipStream_t st);
void sct_launch_join_pass(const JoinArgs<uint32_t> &ja, hipStream_t st);
void sct_launch_join_finish(const JoinArgs<int64_t> &ja, hipStream_t st);
void sct_launch_join_finish(const JoinArgs<uint32_t> &ja, hipStream_t st);
