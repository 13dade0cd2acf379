// rocPRIM wrappers for the large-`limit` search path (limit above what the LDS top-k kernels hold): a stable
// segmented radix sort of (float64 distance bits, candidate reference) pairs -- one segment per query -- and an
// exclusive scan.  Kept in their own translation unit: the rocPRIM templates are slow to compile.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <cstdint>

#include "common.h"

// Two-call convention like rocPRIM's: temp == nullptr -> only *temp_bytes is set.
int cis_seg_sort_u64(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint64_t* vals_in,
                     uint64_t* vals_out, int64_t n, int nseg, const int64_t* seg_begin, const int64_t* seg_end, hipStream_t st) {
    CIS_REQUIRE(n >= 0 && n < ((int64_t)1 << 32) && nseg >= 0, "segmented sort: size out of range");
    size_t bytes = *temp_bytes;
    CIS_CHECK_HIP(rocprim::segmented_radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, (unsigned int)n,
                                                      (unsigned int)nseg, seg_begin, seg_end, 0, 64, st));
    *temp_bytes = bytes;
    return CIS_OK;
}

int cis_exclusive_scan_i64(void* temp, size_t* temp_bytes, const int64_t* in, int64_t* out, int64_t n, hipStream_t st) {
    size_t bytes = *temp_bytes;
    CIS_CHECK_HIP(rocprim::exclusive_scan(temp, bytes, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), st));
    *temp_bytes = bytes;
    return CIS_OK;
}
