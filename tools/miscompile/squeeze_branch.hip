// Reproducer attempt for the "miscompiled branch in the level-2 squeeze" (commit acf9ee8, DESIGN.md section 6 hygiene):
// the OLD form of CompactPolicy::load -- the item's words assigned on the two sides of a nested branch, with an early
// return on the rare side -- inlined 16 times into an unrolled, guarded loop, against the NEW form (selects).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/squeeze_branch tools/miscompile/squeeze_branch.hip && /tmp/squeeze_branch
// Output: for both forms, how many of the n items (none of which has an N) came out marked "rare" (id == ~0).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Rec12 { uint32_t a, b, id; };
struct Side { uint4 *recs; uint32_t *cursor; uint32_t n_slabs, cap; uint32_t *overflow; };
struct Source { const uint4 *in; uint32_t squeeze; Side side; };

__device__ __forceinline__ uint32_t hash12(uint32_t a, uint32_t b)
{
    uint32_t h = (a ^ 0x9E3779B9u) * 0x9E3779B1u;
    h = (h ^ (h >> 15) ^ b) * 0x85EBCA77u;
    return h ^ (h >> 13);
}

__device__ __forceinline__ uint32_t load_old(const Source &s, uint32_t i, Rec12 &v)
{
    const uint4 r = s.in[i];
    if (s.squeeze == 1) {
        if (r.x & r.y) {
            const uint32_t slab = blockIdx.x & (s.side.n_slabs - 1);
            const uint32_t pos = atomicAdd(&s.side.cursor[slab], 1u);
            if (pos < (slab + 1) * s.side.cap)
                s.side.recs[pos] = r;
            else
                atomicOr(s.side.overflow, 16u);
            v = Rec12{0u, 0u, 0xFFFFFFFFu};
            return 0u;
        }
        v = Rec12{r.x | r.z, r.y | r.z, r.w};
    } else {
        v = Rec12{r.x, r.y, r.w};
    }
    return hash12(v.a, v.b);
}

__device__ __forceinline__ uint32_t load_new(const Source &s, uint32_t i, Rec12 &v)
{
    const uint4 r = s.in[i];
    const bool squeeze = s.squeeze == 1;
    const bool rare = squeeze && (r.x & r.y) != 0u;
    if (rare) {
        const uint32_t slab = blockIdx.x & (s.side.n_slabs - 1);
        const uint32_t pos = atomicAdd(&s.side.cursor[slab], 1u);
        if (pos < (slab + 1) * s.side.cap)
            s.side.recs[pos] = r;
        else
            atomicOr(s.side.overflow, 16u);
    }
    v.a = squeeze ? r.x | r.z : r.x;
    v.b = squeeze ? r.y | r.z : r.y;
    v.id = rare ? 0xFFFFFFFFu : r.w;
    return rare ? 0u : hash12(v.a, v.b);
}

constexpr uint32_t EPT = 16, THREADS = 256;

// the shape of partition.cuh's scatter_body at that commit: guarded loads into register arrays, LDS ranks by bin
template <bool OLD>
__global__ __launch_bounds__(THREADS) void scatter_like(Source src, uint32_t n, uint32_t shift, uint32_t n_bins,
                                                        Rec12 *out, uint32_t *n_rare, uint32_t *hist)
{
    __shared__ uint32_t s_hist[1024];
    for (uint32_t b = threadIdx.x; b < n_bins; b += THREADS)
        s_hist[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * THREADS * EPT, hi = min(lo + THREADS * EPT, n);
    Rec12 v[EPT];
    uint32_t h[EPT], rank[EPT];
#pragma unroll
    for (uint32_t e = 0; e < EPT; e++) {
        const uint32_t i = lo + e * THREADS + threadIdx.x;
        if (i < hi)
            h[e] = OLD ? load_old(src, i, v[e]) : load_new(src, i, v[e]);
    }
#pragma unroll
    for (uint32_t e = 0; e < EPT; e++) {
        const uint32_t i = lo + e * THREADS + threadIdx.x;
        rank[e] = 0xFFFFFFFFu;
        if (i < hi && v[e].id != 0xFFFFFFFFu)
            rank[e] = atomicAdd(&s_hist[(h[e] >> shift) & (n_bins - 1)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < EPT; e++) {
        const uint32_t i = lo + e * THREADS + threadIdx.x;
        if (i < hi) {
            if (rank[e] == 0xFFFFFFFFu)
                atomicAdd(n_rare, 1u);
            else
                out[i] = v[e];
        }
    }
    for (uint32_t b = threadIdx.x; b < n_bins; b += THREADS)
        if (s_hist[b])
            atomicAdd(&hist[b], s_hist[b]);
}

int main()
{
    const uint32_t n = 1u << 22, n_slabs = 256, cap = 1024;
    std::vector<uint4> in(n);
    uint64_t x = 88172645463325252ull;
    for (uint32_t i = 0; i < n; i++) {            // N-free "ACGNT" planes: at most one plane bit per base
        uint32_t p[3] = {0, 0, 0};
        for (int b = 0; b < 32; b++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const uint32_t code = (uint32_t)(x % 4);         // A, C, G, T (no N)
            if (code == 1) p[0] |= 1u << b;
            if (code == 2) p[1] |= 1u << b;
            if (code == 3) p[2] |= 1u << b;
        }
        in[i] = make_uint4(p[0], p[1], p[2], i);
    }
    uint4 *d_in, *d_side;
    Rec12 *d_out;
    uint32_t *d_cursor, *d_flags, *d_hist;
    hipMalloc(&d_in, n * 16); hipMalloc(&d_side, (size_t)n_slabs * cap * 16); hipMalloc(&d_out, (size_t)n * 12);
    hipMalloc(&d_cursor, n_slabs * 4); hipMalloc(&d_flags, 16); hipMalloc(&d_hist, 1024 * 4);
    hipMemcpy(d_in, in.data(), n * 16, hipMemcpyHostToDevice);
    std::vector<uint32_t> cur(n_slabs);
    for (uint32_t s = 0; s < n_slabs; s++) cur[s] = s * cap;
    int rc = 0;
    for (int old = 1; old >= 0; old--) {
        hipMemcpy(d_cursor, cur.data(), n_slabs * 4, hipMemcpyHostToDevice);
        hipMemset(d_flags, 0, 16); hipMemset(d_hist, 0, 1024 * 4);
        Source src{d_in, 1u, Side{d_side, d_cursor, n_slabs, cap, d_flags + 1}};
        const uint32_t grid = (n + THREADS * EPT - 1) / (THREADS * EPT);
        if (old) scatter_like<true><<<grid, THREADS>>>(src, n, 22, 1024, d_out, d_flags, d_hist);
        else scatter_like<false><<<grid, THREADS>>>(src, n, 22, 1024, d_out, d_flags, d_hist);
        uint32_t flags[4];
        hipMemcpy(flags, d_flags, 16, hipMemcpyDeviceToHost);
        printf("%s form: %u of %u N-free items came out marked rare\n", old ? "OLD (branches)" : "NEW (selects)", flags[0], n);
        rc |= flags[0] != 0;
    }
    return rc;
}
