// exchange.hip -- multi-GPU stage 2a: group this rank's packed reads by owner rank
// (owner = key_hash mod G) so that one all-to-all(v) can move them. No reference
// counterpart (the reference is single-process).
//
// owner_kernel -> one radix pass over ceil(log2 G) bits (stable: every destination keeps
// its reads in id order, which the receiving collapse relies on for "first holder")
// -> gather_by_owner_kernel: Q = stride/4 lanes per record, one uint4 each, so the write
// side is a perfect stream and the read side fetches whole records.
#include "fqd_internal.h"

namespace {

__global__ void owner_kernel(const uint32_t *__restrict__ hashes, uint64_t n, uint32_t parts,
                             uint32_t *__restrict__ owner)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        owner[i] = hashes[i] % parts;
}

__global__ __launch_bounds__(256) void gather_by_owner_kernel(const uint32_t *__restrict__ order, uint64_t n, KeyShape sh,
                                                              const uint32_t *__restrict__ recs,
                                                              const uint32_t *__restrict__ lens,
                                                              const uint32_t *__restrict__ weights, uint64_t id0,
                                                              uint32_t *__restrict__ recs_out,
                                                              uint32_t *__restrict__ lens_out,
                                                              uint64_t *__restrict__ ids_out,
                                                              uint32_t *__restrict__ ids32_out,
                                                              uint32_t *__restrict__ weights_out)
{
    const uint32_t Q = sh.stride / 4;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / Q;
    const uint32_t q = (uint32_t)(t - i * Q);
    if (i >= n)
        return;
    const uint32_t src = order[i];
    reinterpret_cast<uint4 *>(recs_out + i * sh.stride)[q] =
        reinterpret_cast<const uint4 *>(recs + (uint64_t)src * sh.stride)[q];
    if (q == 0) {
        if (ids_out)
            ids_out[i] = id0 + src;
        if (ids32_out)
            ids32_out[i] = (uint32_t)id0 + src;
        if (lens_out)
            lens_out[i] = sh.ragged ? lens[src] : sh.max_len;
        if (weights_out)
            weights_out[i] = weights ? weights[src] : 1u;
    }
}

// counts[p] = number of sorted owners equal to p (binary search, one thread per part)
__global__ void owner_counts_kernel(const uint32_t *__restrict__ owner_sorted, uint64_t n, uint32_t parts,
                                    uint64_t *__restrict__ counts)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= parts)
        return;
    auto lower = [&](uint32_t key) {
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (owner_sorted[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    counts[p] = lower(p + 1) - lower(p);
}

}  // namespace

namespace fqd {

hipError_t launch_owner(const uint32_t *hashes, uint64_t n, uint32_t parts, uint32_t *owner, hipStream_t st)
{
    if (n)
        owner_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(hashes, n, parts, owner);
    return hipGetLastError();
}

hipError_t launch_gather_by_owner(const uint32_t *order, uint64_t n, KeyShape sh, const uint32_t *recs,
                                  const uint32_t *lens, const uint32_t *weights, uint64_t id0, uint32_t *recs_out,
                                  uint32_t *lens_out, uint64_t *ids_out, uint32_t *ids32_out, uint32_t *weights_out,
                                  hipStream_t st)
{
    if (n) {
        const uint64_t threads = n * (sh.stride / 4);
        gather_by_owner_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(order, n, sh, recs, lens, weights, id0,
                                                                                recs_out, lens_out, ids_out,
                                                                                ids32_out, weights_out);
    }
    return hipGetLastError();
}

hipError_t launch_owner_counts(const uint32_t *owner_sorted, uint64_t n, uint32_t parts, uint64_t *counts,
                               hipStream_t st)
{
    owner_counts_kernel<<<(parts + 63) / 64, 64, 0, st>>>(owner_sorted, n, parts, counts);
    return hipGetLastError();
}

}  // namespace fqd
