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
                                                              uint32_t *__restrict__ weights_out, uint32_t stamp_word)
{
    const uint32_t Q = sh.stride / 4;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / Q;
    const uint32_t q = (uint32_t)(t - i * Q);
    if (i >= n)
        return;
    const uint32_t src = order[i];
    uint4 v = reinterpret_cast<const uint4 *>(recs + (uint64_t)src * sh.stride)[q];
    if (stamp_word && q == stamp_word / 4) {
        // the read's index on THIS rank travels in the record's first padding word (no id array
        // on the wire); the receiver adds the rank's id base (IdSource)
        const uint32_t e = stamp_word & 3u;
        if (e == 0) v.x = src; else if (e == 1) v.y = src; else if (e == 2) v.z = src; else v.w = src;
    }
    reinterpret_cast<uint4 *>(recs_out + i * sh.stride)[q] = v;
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

// ---- stable multi-split (parts <= SPLIT_MAX_PARTS): two light passes instead of a radix sort ----
// A tile is SPLIT_TILE consecutive rows; matrix[p * n_tiles + t] = rows of tile t owned by p. An
// inclusive scan of the flat matrix (part-major) turns it into destination offsets.
constexpr uint32_t SPLIT_TILE = 2048, SPLIT_THREADS = 256, SPLIT_MAX_PARTS = 256;

__global__ __launch_bounds__(SPLIT_THREADS) void split_count_kernel(const uint32_t *__restrict__ owner, uint64_t n,
                                                                    uint32_t parts, uint32_t n_tiles,
                                                                    uint32_t *__restrict__ matrix)
{
    __shared__ uint32_t cnt[SPLIT_MAX_PARTS];
    for (uint32_t p = threadIdx.x; p < parts; p += SPLIT_THREADS)
        cnt[p] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * SPLIT_TILE;
    for (uint32_t k = 0; k < SPLIT_TILE; k += SPLIT_THREADS) {
        const uint64_t i = base + k + threadIdx.x;
        const bool live = i < n;
        const uint32_t o = live ? owner[i] : 0xFFFFFFFFu;
        // one LDS add per (wave, distinct owner): with a handful of parts, per-row adds would all
        // hit the same few LDS words
        unsigned long long todo = __ballot(live);
        while (todo) {
            const uint32_t lead = __shfl(o, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(live && o == lead);
            if (o == lead && (same & fqd_lanemask_lt()) == 0)
                atomicAdd(&cnt[lead], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < parts; p += SPLIT_THREADS)
        matrix[(size_t)p * n_tiles + blockIdx.x] = cnt[p];
}

// order[dst] = source row, dst = rows of smaller parts + rows of this part in earlier tiles +
// rows of this part earlier in the tile (stable: every part keeps its rows in input order).
__global__ __launch_bounds__(SPLIT_THREADS) void split_order_kernel(const uint32_t *__restrict__ owner, uint64_t n,
                                                                    uint32_t parts, uint32_t n_tiles,
                                                                    const uint32_t *__restrict__ matrix,
                                                                    const uint32_t *__restrict__ matrix_incl,
                                                                    uint32_t *__restrict__ order)
{
    constexpr uint32_t WAVES = SPLIT_THREADS / 64;
    __shared__ uint32_t running[SPLIT_MAX_PARTS];          // next free slot of part p for this tile
    __shared__ uint32_t wave_cnt[WAVES][SPLIT_MAX_PARTS];  // rows of part p in wave w, this round
    for (uint32_t p = threadIdx.x; p < parts; p += SPLIT_THREADS) {
        const size_t m = (size_t)p * n_tiles + blockIdx.x;
        running[p] = matrix_incl[m] - matrix[m];
        for (uint32_t w = 0; w < WAVES; w++)
            wave_cnt[w][p] = 0;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SPLIT_TILE;
    for (uint32_t k = 0; k < SPLIT_TILE; k += SPLIT_THREADS) {
        const uint64_t i = base + k + threadIdx.x;
        const bool live = i < n;
        const uint32_t o = live ? owner[i] : 0xFFFFFFFFu;
        // rank among the wave's earlier lanes with the same owner: one ballot per distinct owner
        uint32_t rank = 0;
        unsigned long long todo = __ballot(live);
        while (todo) {
            const uint32_t lead = __shfl(o, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(live && o == lead);
            if (o == lead) {
                rank = (uint32_t)__popcll(same & fqd_lanemask_lt());
                if (rank == 0)
                    wave_cnt[wave][lead] = (uint32_t)__popcll(same);
            }
            todo &= ~same;
        }
        __syncthreads();
        if (live) {
            uint32_t before = running[o];
            for (uint32_t w = 0; w < wave; w++)
                before += wave_cnt[w][o];
            order[before + rank] = (uint32_t)i;
        }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < parts; p += SPLIT_THREADS) {
            uint32_t add = 0;
            for (uint32_t w = 0; w < WAVES; w++) {
                add += wave_cnt[w][p];
                wave_cnt[w][p] = 0;
            }
            running[p] += add;
        }
        __syncthreads();
    }
}

// counts[p] = rows of part p, from the scanned matrix
__global__ void split_totals_kernel(const uint32_t *__restrict__ matrix_incl, uint32_t parts, uint32_t n_tiles,
                                    uint64_t *__restrict__ counts)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= parts)
        return;
    const uint32_t hi = matrix_incl[(size_t)(p + 1) * n_tiles - 1];
    const uint32_t lo = p ? matrix_incl[(size_t)p * n_tiles - 1] : 0u;
    counts[p] = hi - lo;
}

__global__ void extract_ids_kernel(IdSource src, uint32_t *__restrict__ recs, uint64_t n, uint64_t *__restrict__ ids64)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    ids64[i] = src.at((uint32_t)i);
    recs[i * src.stride + src.spare_word] = 0;
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
                                  hipStream_t st, uint32_t stamp_word)
{
    if (n) {
        const uint64_t threads = n * (sh.stride / 4);
        gather_by_owner_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(order, n, sh, recs, lens, weights, id0,
                                                                                recs_out, lens_out, ids_out,
                                                                                ids32_out, weights_out, stamp_word);
    }
    return hipGetLastError();
}

uint32_t split_tiles(uint64_t n) { return (uint32_t)((n + SPLIT_TILE - 1) / SPLIT_TILE); }
uint32_t split_max_parts() { return SPLIT_MAX_PARTS; }

hipError_t launch_split_count(const uint32_t *owner, uint64_t n, uint32_t parts, uint32_t *matrix, hipStream_t st)
{
    const uint32_t tiles = split_tiles(n);
    if (tiles)
        split_count_kernel<<<tiles, SPLIT_THREADS, 0, st>>>(owner, n, parts, tiles, matrix);
    return hipGetLastError();
}

hipError_t launch_split_order(const uint32_t *owner, uint64_t n, uint32_t parts, const uint32_t *matrix,
                              const uint32_t *matrix_incl, uint32_t *order, uint64_t *counts, hipStream_t st)
{
    const uint32_t tiles = split_tiles(n);
    if (tiles) {
        split_order_kernel<<<tiles, SPLIT_THREADS, 0, st>>>(owner, n, parts, tiles, matrix, matrix_incl, order);
        split_totals_kernel<<<(parts + 63) / 64, 64, 0, st>>>(matrix_incl, parts, tiles, counts);
    }
    return hipGetLastError();
}

hipError_t launch_extract_ids(IdSource src, uint32_t *recs, uint64_t n, uint64_t *ids64, hipStream_t st)
{
    if (n)
        extract_ids_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, recs, n, ids64);
    return hipGetLastError();
}

hipError_t launch_owner_counts(const uint32_t *owner_sorted, uint64_t n, uint32_t parts, uint64_t *counts,
                               hipStream_t st)
{
    owner_counts_kernel<<<(parts + 63) / 64, 64, 0, st>>>(owner_sorted, n, parts, counts);
    return hipGetLastError();
}

}  // namespace fqd
