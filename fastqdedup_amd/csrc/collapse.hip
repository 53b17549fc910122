// collapse.hip -- stage 2: exact duplicates -> unique keys + counts + first holders.
//
// Replaces the duplicate-count half of TrieNode_AddSequence (reference
// _triemodule.c:235-239, :261-264: "identical key => count += n").
//
// Reads are radix-sorted by (a prefix of) their 32-bit key hash (prims.hip);
// the kernels here turn the sorted order into runs of IDENTICAL KEYS. A hash
// only proposes: a read opens a new run unless its full record equals its
// predecessor's, and runs of equal hash that hold more than one distinct key
// (hash collisions) are re-ordered by (key, id) so that equal keys
// become adjacent. The sort is stable and ids ascend, so the head of a run is
// the FIRST holder of the key in input order (pass-2 rule, __init__.py:201-206).
#include "fqd_internal.h"

namespace {

__device__ __forceinline__ bool records_equal(const uint32_t *__restrict__ recs, const uint32_t *__restrict__ lens,
                                              const KeyShape &sh, uint32_t a, uint32_t b)
{
    if (sh.ragged && lens[a] != lens[b])
        return false;
    const uint4 *pa = reinterpret_cast<const uint4 *>(recs + (uint64_t)a * sh.stride);
    const uint4 *pb = reinterpret_cast<const uint4 *>(recs + (uint64_t)b * sh.stride);
    for (uint32_t j = 0; j < sh.stride / 4; j++) {
        const uint4 x = pa[j], y = pb[j];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w)
            return false;
    }
    return true;
}

// Any total order on (record, id); only used to regroup a collision run.
__device__ __forceinline__ int record_order(const uint32_t *__restrict__ recs, const uint32_t *__restrict__ lens,
                                            const KeyShape &sh, uint32_t a, uint32_t b)
{
    if (sh.ragged && lens[a] != lens[b])
        return lens[a] < lens[b] ? -1 : 1;
    const uint32_t *pa = recs + (uint64_t)a * sh.stride, *pb = recs + (uint64_t)b * sh.stride;
    for (uint32_t j = 0; j < sh.words * sh.planes; j++)
        if (pa[j] != pb[j])
            return pa[j] < pb[j] ? -1 : 1;
    return 0;
}

__global__ void iota_kernel(uint32_t *out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (uint32_t)i;
}

// A read opens a new run unless its hash AND its full record equal its predecessor's.
// Q = stride/4 lanes serve one read (one uint4 of its record each, so the gather of a record is
// one coalesced request); a wave holds 64/Q reads. The predecessor's chunk comes from the lane
// Q below (__shfl_up); only the first read of a wave fetches its predecessor's record itself.
// Reads whose hash is unique among their neighbours gather nothing.
__global__ __launch_bounds__(256) void head_flags_kernel(const uint32_t *__restrict__ hs, const uint32_t *__restrict__ ids,
                                                         const uint32_t *__restrict__ recs,
                                                         const uint32_t *__restrict__ lens, uint64_t n, KeyShape sh,
                                                         uint32_t hash_mask, uint32_t *__restrict__ flags,
                                                         uint32_t *__restrict__ n_collision_runs,
                                                         uint32_t *__restrict__ collision_runs, uint32_t cap)
{
    const uint32_t Q = sh.stride / 4, rpw = 64u / Q, lane = fqd_lane();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t rl = lane / Q, q = lane - rl * Q;
    const uint64_t i = wave * rpw + rl;
    const bool valid = rl < rpw && i < n;
    const uint32_t h = valid ? (hs[i] & hash_mask) : 0u;
    const uint32_t id = valid ? ids[i] : 0u;
    const bool same_prev = valid && i > 0 && (hs[i - 1] & hash_mask) == h;
    const bool same_next = valid && i + 1 < n && (hs[i + 1] & hash_mask) == h;
    const bool need = same_prev || same_next;
    const bool fetch_prev = same_prev && rl == 0;       // the read below lives in another wave
    const uint32_t prev_id = fetch_prev ? ids[i - 1] : 0u;
    bool neq = false;
    if (sh.ragged) {
        const uint32_t mine = need ? lens[id] : 0u;
        uint32_t prev = __shfl_up(mine, Q);
        if (fetch_prev)
            prev = lens[prev_id];
        neq = mine != prev;
    }
    uint4 m = make_uint4(0, 0, 0, 0);
    if (need)
        m = reinterpret_cast<const uint4 *>(recs + (uint64_t)id * sh.stride)[q];
    uint4 p;
    p.x = __shfl_up(m.x, Q);
    p.y = __shfl_up(m.y, Q);
    p.z = __shfl_up(m.z, Q);
    p.w = __shfl_up(m.w, Q);
    if (fetch_prev)
        p = reinterpret_cast<const uint4 *>(recs + (uint64_t)prev_id * sh.stride)[q];
    neq = neq || m.x != p.x || m.y != p.y || m.z != p.z || m.w != p.w;
    // a read differs from its predecessor when any of its Q lanes saw a difference
    const unsigned long long diff = __ballot(same_prev && neq);
    const unsigned long long group = Q >= 64 ? ~0ull : (((1ull << Q) - 1ull) << (rl * Q));
    const bool eq = (diff & group) == 0;
    if (!valid || q != 0)
        return;
    uint32_t head = 1;
    if (same_prev) {
        if (eq) {
            head = 0;
        } else {
            // equal hash, different key: report the run once (by its first such position)
            uint64_t a = i - 1;
            bool first = true;
            while (a > 0 && (hs[a - 1] & hash_mask) == h) {
                if (!records_equal(recs, lens, sh, ids[a], ids[a - 1])) {
                    first = false;
                    break;
                }
                a--;
            }
            if (first) {
                const uint32_t slot = atomicAdd(n_collision_runs, 1u);
                if (slot < cap)
                    collision_runs[slot] = (uint32_t)a;
            }
        }
    }
    flags[i] = head;
}

// One thread per reported run [a, b): insertion sort of ids by (record, id), then
// rewrite the head flags of the run. Runs are a handful of reads unless the hash
// has been narrowed on purpose (FQD_HASH_BITS, tests).
__global__ void fix_collision_runs_kernel(const uint32_t *__restrict__ hs, uint32_t *__restrict__ ids,
                                          const uint32_t *__restrict__ recs, const uint32_t *__restrict__ lens,
                                          uint64_t n, KeyShape sh, uint32_t hash_mask, uint32_t *__restrict__ flags,
                                          const uint32_t *__restrict__ collision_runs, uint32_t n_runs)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs)
        return;
    const uint64_t a = collision_runs[r];
    const uint32_t h = hs[a] & hash_mask;
    uint64_t b = a + 1;
    while (b < n && (hs[b] & hash_mask) == h)
        b++;
    for (uint64_t i = a + 1; i < b; i++) {
        const uint32_t x = ids[i];
        uint64_t j = i;
        while (j > a) {
            const uint32_t y = ids[j - 1];
            const int c = record_order(recs, lens, sh, y, x);
            if (c < 0 || (c == 0 && y < x))
                break;
            ids[j] = y;
            j--;
        }
        ids[j] = x;
    }
    flags[a] = 1;
    for (uint64_t i = a + 1; i < b; i++)
        flags[i] = records_equal(recs, lens, sh, ids[i], ids[i - 1]) ? 0u : 1u;
}

__global__ void run_starts_kernel(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ run_idx,
                                  uint64_t n, uint32_t *__restrict__ run_start)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (flags[i])
        run_start[run_idx[i] - 1] = (uint32_t)i;
    if (i == n - 1)
        run_start[run_idx[i]] = (uint32_t)n;
}

__global__ void run_weights_kernel(const uint32_t *__restrict__ run_start, uint32_t n_runs,
                                   const uint32_t *__restrict__ ids, const uint32_t *__restrict__ weights,
                                   uint32_t *__restrict__ run_weight, uint32_t *__restrict__ live_flag)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs)
        return;
    const uint32_t a = run_start[r], b = run_start[r + 1];
    uint32_t w = 0;
    if (weights) {
        for (uint32_t i = a; i < b; i++)
            w += weights[ids[i]];
    } else {
        w = b - a;
    }
    run_weight[r] = w;
    live_flag[r] = w ? 1u : 0u;
}

// Q = stride/4 lanes copy one record, one uint4 each: coalesced for any record size.
__global__ __launch_bounds__(256) void write_unique_kernel(const uint32_t *__restrict__ run_start, const uint32_t *__restrict__ run_weight,
                                    const uint32_t *__restrict__ live_flag, const uint32_t *__restrict__ live_idx,
                                    uint32_t n_runs, const uint32_t *__restrict__ ids,
                                    const uint32_t *__restrict__ recs, const uint32_t *__restrict__ lens,
                                    const uint64_t *__restrict__ read_ids, KeyShape sh,
                                    uint32_t *__restrict__ urecs, uint32_t *__restrict__ ulens,
                                    uint32_t *__restrict__ ucounts, uint64_t *__restrict__ ufirst)
{
    const uint32_t Q = sh.stride / 4;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = t / Q;
    const uint32_t q = (uint32_t)(t - r * Q);
    if (r >= n_runs || !live_flag[r])
        return;
    const uint32_t u = live_idx[r] - 1;  // inclusive scan of live flags
    const uint32_t id = ids[run_start[r]];
    reinterpret_cast<uint4 *>(urecs + (uint64_t)u * sh.stride)[q] =
        reinterpret_cast<const uint4 *>(recs + (uint64_t)id * sh.stride)[q];
    if (q == 0) {
        if (sh.ragged)
            ulens[u] = lens[id];
        ucounts[u] = run_weight[r];
        ufirst[u] = read_ids ? read_ids[id] : (uint64_t)id;
    }
}

__global__ void sum_u32_kernel(const uint32_t *__restrict__ in, uint64_t n, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        s += in[i];
    for (int o = 32; o; o >>= 1)
        s += __shfl_xor(s, o);
    if (fqd_lane() == 0 && s)
        atomicAdd(out, s);
}

__global__ void max_u64_kernel(const uint64_t *__restrict__ in, uint64_t n, unsigned long long *__restrict__ out)
{
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        m = in[i] > m ? in[i] : m;
    for (int o = 32; o; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o);
        m = other > m ? other : m;
    }
    if (fqd_lane() == 0)
        atomicMax(out, m);
}

inline unsigned grid_for(uint64_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }

}  // namespace

namespace fqd {

hipError_t launch_iota_u32(uint32_t *out, uint64_t n, hipStream_t st)
{
    if (n)
        iota_kernel<<<grid_for(n), 256, 0, st>>>(out, n);
    return hipGetLastError();
}

hipError_t launch_head_flags(const uint32_t *hs, const uint32_t *ids, const uint32_t *recs, const uint32_t *lens,
                             uint64_t n, KeyShape sh, uint32_t hash_mask, uint32_t *flags,
                             uint32_t *n_collision_runs, uint32_t *collision_runs, uint32_t cap, hipStream_t st)
{
    if (n) {
        if (sh.stride / 4 > 64)
            return hipErrorInvalidValue;
        const uint64_t rpw = 64 / (sh.stride / 4), waves = (n + rpw - 1) / rpw;
        head_flags_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(hs, ids, recs, lens, n, sh, hash_mask, flags,
                                                                     n_collision_runs, collision_runs, cap);
    }
    return hipGetLastError();
}

hipError_t launch_fix_collision_runs(const uint32_t *hs, uint32_t *ids, const uint32_t *recs, const uint32_t *lens,
                                     uint64_t n, KeyShape sh, uint32_t hash_mask, uint32_t *flags,
                                     const uint32_t *collision_runs, uint32_t n_runs, hipStream_t st)
{
    if (n_runs)
        fix_collision_runs_kernel<<<grid_for(n_runs, 64), 64, 0, st>>>(hs, ids, recs, lens, n, sh, hash_mask,
                                                                       flags, collision_runs, n_runs);
    return hipGetLastError();
}

hipError_t launch_run_starts(const uint32_t *flags, const uint32_t *run_idx, uint64_t n, uint32_t *run_start,
                             hipStream_t st)
{
    if (n)
        run_starts_kernel<<<grid_for(n), 256, 0, st>>>(flags, run_idx, n, run_start);
    return hipGetLastError();
}

hipError_t launch_run_weights(const uint32_t *run_start, uint32_t n_runs, uint64_t, const uint32_t *ids,
                              const uint32_t *weights, uint32_t *run_weight, uint32_t *live_flag, hipStream_t st)
{
    if (n_runs)
        run_weights_kernel<<<grid_for(n_runs), 256, 0, st>>>(run_start, n_runs, ids, weights, run_weight,
                                                             live_flag);
    return hipGetLastError();
}

hipError_t launch_write_unique(const uint32_t *run_start, const uint32_t *run_weight, const uint32_t *live_flag,
                               const uint32_t *live_idx, uint32_t n_runs, const uint32_t *ids,
                               const uint32_t *recs, const uint32_t *lens, const uint64_t *read_ids, KeyShape sh,
                               uint32_t *urecs, uint32_t *ulens, uint32_t *ucounts, uint64_t *ufirst,
                               hipStream_t st)
{
    if (n_runs)
        write_unique_kernel<<<grid_for((uint64_t)n_runs * (sh.stride / 4)), 256, 0, st>>>(run_start, run_weight, live_flag, live_idx, n_runs,
                                                              ids, recs, lens, read_ids, sh, urecs, ulens,
                                                              ucounts, ufirst);
    return hipGetLastError();
}

hipError_t launch_max_u64(const uint64_t *in, uint64_t n, unsigned long long *out, hipStream_t st)
{
    if (n) {
        unsigned g = grid_for(n);
        if (g > 1024)
            g = 1024;
        max_u64_kernel<<<g, 256, 0, st>>>(in, n, out);
    }
    return hipGetLastError();
}

hipError_t launch_sum_u32(const uint32_t *in, uint64_t n, unsigned long long *out, hipStream_t st)
{
    if (n) {
        unsigned g = grid_for(n);
        if (g > 1024)
            g = 1024;
        sum_u32_kernel<<<g, 256, 0, st>>>(in, n, out);
    }
    return hipGetLastError();
}

}  // namespace fqd
