// prims.hip -- device-wide sort / scan from rocPRIM (AMD's own primitives
// library, tuned per-arch incl. gfx950). Kept in one translation unit so the
// heavy headers are compiled once. Everything domain-specific (packing,
// bucket pair search, components, dissection) is hand-written elsewhere.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fqd_internal.h"

namespace fqd {

size_t sort_pairs_u32_u32_temp(uint64_t n, int begin_bit, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, n, begin_bit, end_bit);
    return bytes;
}

hipError_t sort_pairs_u32_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, uint64_t n, int begin_bit, int end_bit,
                              hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
}

size_t sort_keys_u64_temp(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, n);
    return bytes;
}

hipError_t sort_keys_u64(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, uint64_t n,
                         int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_keys(tmp, tmp_bytes, kin, kout, n, 0, end_bit, st);
}

size_t sort_pairs_u64_u32_temp(uint64_t n, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, end_bit);
    return bytes;
}

hipError_t sort_pairs_u64_u32(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                              uint32_t *vout, uint64_t n, int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, st);
}

size_t scan_max_u32_temp(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, n,
                                  rocprim::maximum<uint32_t>());
    return bytes;
}

hipError_t inclusive_scan_max_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n,
                                  hipStream_t st)
{
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::maximum<uint32_t>(), st);
}

size_t scan_u32_temp(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, n,
                                  rocprim::plus<uint32_t>());
    return bytes;
}

hipError_t inclusive_scan_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n,
                              hipStream_t st)
{
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::plus<uint32_t>(), st);
}

}  // namespace fqd
