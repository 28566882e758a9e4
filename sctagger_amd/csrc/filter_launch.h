// filter_launch.h — host entry points of the seed-filter translation units (filter_e0/1/2.hip), called by sctagger_hip.hip.
#pragma once
struct SctFilterVariant {
    bool retry;   // the retry launch (larger lists, 4 listed rows per wave) instead of the main one
    bool rt;      // barcode length other than 16: piece boundaries computed at run time
    bool slots;   // lookups read the 8-byte bucket slots
    bool thirds;  // the 20/22-bit families of very large barcode sets (max_err <= 1)
    bool wide;    // the sixteen 20-bit families of large barcode sets (max_err = 2)
};
template <typename OffT>
struct FilterArgs;
void sct_launch_filter_e0(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e0(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e1(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e1(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e2(SctFilterVariant v, const FilterArgs<int64_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
void sct_launch_filter_e2(SctFilterVariant v, const FilterArgs<uint32_t> &fa, unsigned grid, unsigned grid64, hipStream_t st);
