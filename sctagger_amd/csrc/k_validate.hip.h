// k_validate.hip.h — k_validate: device-side input checks.
// Part of libsctagger_hip.so; included by sctagger_hip.hip (in this order: common, decode, filter, scan,
// offsets, whitelist, adapter, validate).  gfx950 only.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// k_validate: input checks of sct_upload_segments on the device.  bad[0] = smallest row whose offsets are not
// non-decreasing (or n if offs[n] != bytes), bad[1] = smallest byte offset holding a value >= 128; ~0 = none.
// check_bytes = false: offsets only.  The match_trie path looks for bytes >= 128 where the reference would map them — in
// the rows of at least bl - max_err characters, inside the kernels that read those rows (k_filter, k_route_all,
// k_route_long); the two neighbouring stages (whitelist counting, adapter alignment) refuse any such byte here.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_validate(const uint8_t *__restrict__ segs, int64_t bytes, const int64_t *__restrict__ offs, int64_t n,
           unsigned long long *__restrict__ bad, bool check_bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (offs && i < n) {  // offs == NULL: the bytes only (the stream path checks its index in k_index_*)
        const int64_t a = offs[i], b = offs[i + 1];
        if (b < a || a < 0 || b > bytes) atomicMin(&bad[0], (unsigned long long)i);
    }
    const int64_t p = i * 16;
    if (check_bytes && p < bytes) {  // the buffer is padded by 64 bytes, so the 16-byte load may run past `bytes`
        const uint4 v = *reinterpret_cast<const uint4 *>(segs + p);
        if ((v.x | v.y | v.z | v.w) & 0x80808080u) {
            for (int k = 0; k < 16 && p + k < bytes; k++)
                if (segs[p + k] >= 128) {
                    atomicMin(&bad[1], (unsigned long long)(p + k));
                    break;
                }
        }
    }
}
