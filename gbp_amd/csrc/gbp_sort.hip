// gbp_sort.hip -- the one library primitive of the graph build: a stable LSD radix sort of (key, value) int pairs on the
// device (rocPRIM, AMD's own primitives library).  Kept in its own translation unit so that the kernels' file does not pay for
// the rocPRIM headers.  Used twice per gbp_ba_create (gbp_build.hpp): observations by camera (the reference's factor order,
// gbp_ba.py:128-130) and reference factors by landmark (VariableNode.adj_factors order, gbp_ba.py:139).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace gbp {

// bytes of temporary storage for sort_pairs (n pairs, keys below 2^bits)
size_t sort_pairs_tmp_bytes(size_t n, int bits)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<const int *>(nullptr), static_cast<int *>(nullptr),
                                    static_cast<const int *>(nullptr), static_cast<int *>(nullptr), n, 0u, (unsigned)bits, nullptr);
    return bytes;
}

// stable sort of (keys, vals) by the low `bits` bits of the non-negative keys; returns a hipError_t value
int sort_pairs(void *tmp, size_t tmp_bytes, const int *keys_in, int *keys_out, const int *vals_in, int *vals_out, size_t n, int bits,
               hipStream_t stream)
{
    return (int)rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)bits, stream);
}

}  // namespace gbp
