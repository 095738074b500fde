// vx_sort.hip -- device sort of the octree's 64-bit Morton items (replaces std::sort(par_unseq), octTree.hpp:363).
// Keys only, so stability is immaterial: duplicates are identical values.  rocPRIM's radix sort is the library path;
// it is restricted to the `bits` low bits actually used (3 * bitsPerAxis).
#include "vx_internal.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace vx {

size_t sort_tmp_bytes(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0);
    return bytes ? bytes : 16;
}

void launch_sort_u64(uint64_t* keys_in, uint64_t* keys_out, uint64_t n, int bits, void* tmp, size_t tmp_bytes, hipStream_t s)
{
    if (!n) return;
    if (bits < 1) bits = 1;
    if (bits > 64) bits = 64;
    (void)rocprim::radix_sort_keys(tmp, tmp_bytes, (const uint64_t*)keys_in, keys_out, (size_t)n, 0u, (unsigned)bits, s);
}

}  // namespace vx
