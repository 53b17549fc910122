// graph.hip -- stage 4 (components) and stage 5 (dissection).
//
// Stage 4 replaces the BFS of Trie.pop_cluster (reference _triemodule.c:865-895):
// a popped cluster is exactly one connected component of the "within distance"
// graph on unique keys, so a lock-free union-find over the
// edge list gives the same partition. Hooks always point a larger root at a
// smaller one, so the final label of a component is its smallest uid.
//
// Stage 5 replaces cluster_dissection_* (reference __init__.py:60-122). All
// three start from sorted(cluster), i.e. the order "(count, key)" with Python
// str comparison -- rank_greater() below. They are restated as order-free fixed
// points over the edge list (derivations in DESIGN.md, checked against the
// reference's own functions in tests/):
//   highest_count  keep the rank-maximum of each component.
//   directional    with arcs t->c for every edge where 2*count_c-1 <= count_t
//                  (:84), key v is kept iff no key of higher rank reaches v along
//                  arcs. best[v] = max rank over {u : u ~> v}; kept iff best[v]==v.
//                  Closed form (no rounds): a key with count >= 2
//                  is reached only from keys of strictly larger count, so it is kept iff
//                  it has NO in-arc; keys of count 1 take an arc from every neighbour and
//                  reach each other, so a connected set S of count-1 keys is dropped
//                  entirely when it touches a key of count >= 2, and otherwise (S is a
//                  whole component) keeps exactly its largest key.
//   adjacency      greedy "take the max, drop its neighbours, repeat" (:112-122) is
//                  the lexicographically-first maximal independent set in rank
//                  order: v is kept iff none of its higher-rank neighbours is kept.
#include <algorithm>
#include "fqd_internal.h"

#ifndef FQD_KB_KPT
#define FQD_KB_KPT 16   // keys per thread of kept_bin_kernel (8: 0.337 instead of 0.238 ms -- twice the cursor atomics)
#endif
#ifndef FQD_KB_SUBS
#define FQD_KB_SUBS 1   // lists per id bin of the kept-id list (4: no change, 0.238 ms either way)
#endif
namespace {

__device__ __forceinline__ bool rank_greater(uint32_t a, uint32_t b, const uint32_t *__restrict__ ucounts,
                                             const uint32_t *__restrict__ urecs,
                                             const uint32_t *__restrict__ ulens, const KeyShape &sh)
{
    const uint32_t ca = ucounts[a], cb = ucounts[b];
    if (ca != cb)
        return ca > cb;
    return fqd_key_cmp(urecs + (uint64_t)a * sh.stride, fqd_key_len(sh, ulens, a),
                       urecs + (uint64_t)b * sh.stride, fqd_key_len(sh, ulens, b), sh.planes, sh.words) > 0;
}

__device__ __forceinline__ uint32_t load_relaxed(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- union-find ------------------------------------------------------------------
__global__ void uf_init_kernel(uint32_t *parent, uint64_t U)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < U)
        parent[i] = (uint32_t)i;
}

// FRESH = false: the walk reads through the CU's vector cache (plain loads). A stale parent is still an ancestor --
// every value a node's parent ever had is one -- so a cached walk ends at a node that WAS a root; whether it still is
// one is the hooking compare-and-swap's business (it fails, and the caller walks again with FRESH loads, which read
// the L2 and guarantee progress). With device-scope loads all the way, every find of a giant component (65 536 keys,
// 786 K edges in the skewed workload) read the one line holding the component's root from its L2 channel: 3 ms.
template <bool FRESH = true>
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
    uint32_t p = FRESH ? load_relaxed(&parent[x]) : parent[x];
    while (p != x) {
        const uint32_t g = FRESH ? load_relaxed(&parent[p]) : parent[p];
        // path halving by a plain (relaxed) store: g is an ancestor of x whatever other threads have written to
        // parent[x] meanwhile (a root is never un-rooted except by hooking it under a smaller node, and every value
        // ever stored is an ancestor), so the worst a lost race does is keep a longer path. An atomicMin here put
        // one read-modify-write per step on the few lines at the top of a giant component's tree.
#ifndef FQD_UF_NO_HALVING
        if (g != p)
            __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        x = p;
        p = g;
    }
    return x;
}

// n_hooks (may be NULL; FQD_HOOK_SLOTS x 8 words, summed by the host) += successful hooks: every hook merges two components, so
// components = nodes - hooks without a sweep over the nodes.
__global__ __launch_bounds__(256) void uf_union_kernel(uint32_t *parent, const uint32_t *__restrict__ edges, uint64_t E,
                                                       unsigned long long *n_hooks, uint32_t phase)
{
    // phase 0: every edge. A context that has met a giant component (many lanes had to walk a second time: word 1 of
    // the hook slots counts them) runs the unions in two launches: phase 1 takes every 16th edge -- a sixteenth of
    // the threads pulling at the same roots --, phase 2 the rest, most of which then find both ends under one root
    // on their first, cached walk and send no compare-and-swap at all.
    // The lanes of a wave hook TOGETHER: lanes that are about to hang the same root under the same node send ONE
    // compare-and-swap (the lowest such lane does; the others read its answer from LDS). In a component of tens of
    // thousands of keys the last few roots are every lane's target at once, and compare-and-swaps on ONE word are
    // served at ~88 per microsecond: the 786 K edges of the skewed workload's 65 536-key component took 1.6-3.4 ms.
    // For ordinary data (components of two or three keys) every lane is its own leader: one round, a few LDS
    // instructions more than before.
    __shared__ uint32_t s_slot[4][128], s_res[4][64];
    const uint32_t lane = fqd_lane(), wave = threadIdx.x >> 6;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hooked = false;
    uint32_t a = 0, b = 0;
    if (e < E && (phase == 0 || ((e & 15u) == 0) == (phase == 1))) {
        const uint2 uv = reinterpret_cast<const uint2 *>(edges)[e];
        a = uv.x;
        b = uv.y;
    }
    // (second and later walks of the wave's lanes, counted per walk: a lane of a giant component retries many times,
    // a lane of a small cluster whose neighbour hooked first retries once)
    bool active = a != b;
    uint32_t walks_again = 0;
    for (bool fresh = false; __ballot(active); fresh = true) {
        if (fresh)
            walks_again += (uint32_t)__popcll(__ballot(active));
        if (active) {
            // the first walk through the CU's cache (see uf_find): a common ancestor found there IS one
            a = fresh ? uf_find<true>(parent, a) : uf_find<false>(parent, a);
            b = fresh ? uf_find<true>(parent, b) : uf_find<false>(parent, b);
            if (a == b)
                active = false;
            if (a > b) {
                const uint32_t t = a;
                a = b;
                b = t;
            }
        }
        // leaders: per (a, b), the lowest lane that wants to hang root b under a
        s_slot[wave][lane] = 0xFFFFFFFFu;
        s_slot[wave][lane + 64] = 0xFFFFFFFFu;
        const uint32_t h = ((b * 0x9E3779B1u) ^ (a * 0x85EBCA6Bu)) >> 25;
        if (active)
            atomicMin(&s_slot[wave][h], lane);
        const uint32_t owner = active ? s_slot[wave][h] : lane;
        const uint32_t ob = __shfl(b, owner & 63u), oa = __shfl(a, owner & 63u);
        const bool follower = active && owner != lane && ob == b && oa == a;
        uint32_t res = 0xFFFFFFFFu;
        if (active && !follower)
            res = atomicCAS(&parent[b], b, a);          // hook the larger root under the smaller one
        s_res[wave][lane] = res;
        if (follower)
            res = s_res[wave][owner];
        if (active && res == b) {
            active = false;                               // hung (by this lane or by its leader)
            hooked = !follower;
        }
    }
    if (n_hooks) {
        // FQD_HOOK_SLOTS counters one cache line apart: tens of thousands of waves adding to ONE
        // word serialise in the L2 (measured: 350 us against 98 us for the unions themselves)
        const unsigned long long m = __ballot(hooked);
        if (m && lane == (uint32_t)(__ffsll((long long)m) - 1))
            atomicAdd(n_hooks + (size_t)(blockIdx.x % FQD_HOOK_SLOTS) * 8, (unsigned long long)__popcll(m));
        if (walks_again && lane == 0)
            atomicAdd(n_hooks + (size_t)(blockIdx.x % FQD_HOOK_SLOTS) * 8 + 1, (unsigned long long)walks_again);
    }
}

// An edge with an end taken out of the store (fqd_store_remove: a popped cluster's key) becomes a
// self-loop, which every consumer skips: a popped key must not bridge two stored keys when the
// caller pops again with another distance (the reference's trie no longer holds it, _triemodule.c:830).
__global__ void mask_dead_edges_kernel(uint32_t *edges, uint64_t E, const uint8_t *__restrict__ alive)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E)
        return;
    const uint2 uv = reinterpret_cast<const uint2 *>(edges)[e];
    if (uv.x != uv.y && !(alive[uv.x] && alive[uv.y]))
        reinterpret_cast<uint2 *>(edges)[e] = make_uint2(uv.x, uv.x);
}

// *n_components = n_nodes - sum of the hook slots (one wave)
__global__ void hook_total_kernel(const unsigned long long *__restrict__ slots, uint64_t n_nodes,
                                  unsigned long long *n_components, unsigned long long *n_second_walks /* may be NULL */)
{
    unsigned long long sum = 0, again = 0;
    for (uint32_t i = threadIdx.x; i < FQD_HOOK_SLOTS; i += 64) {
        sum += slots[(size_t)i * 8];
        again += slots[(size_t)i * 8 + 1];
    }
    for (int o = 32; o; o >>= 1) {
        sum += __shfl_xor(sum, o);
        again += __shfl_xor(again, o);
    }
    if (threadIdx.x == 0) {
        *n_components = n_nodes - sum;
        if (n_second_walks)
            *n_second_walks = again;
    }
}

// roots[e] = component label (smallest node) of edge e's first end, after all unions are done
__global__ void edge_roots_kernel(uint32_t *parent, const uint32_t *__restrict__ edges, uint64_t E,
                                  uint32_t *__restrict__ roots)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E)
        roots[e] = uf_find(parent, edges[2 * e]);
}

// *bad = 1 when some idx[i] >= limit (checked before a gather trusts a caller's indices)
// ---- a rank's share of the clusters as a subgraph of its own (multi-GPU, sharded.py step 4) ----
// Edges whose cluster this rank dissects (root % n_parts == part) are appended to `sub` (any
// order) and their ends flagged; a scan of the flags numbers the flagged nodes in ascending order.
__global__ __launch_bounds__(1024) void subgraph_mark_kernel(const uint32_t *__restrict__ uv, const uint32_t *__restrict__ roots,
                                                             uint64_t E, uint32_t n_parts, uint32_t part,
                                                             uint32_t *__restrict__ flags, uint32_t *__restrict__ sub,
                                                             unsigned long long *__restrict__ n_sub)
{
    // one reservation per WORKGROUP of 1024 edges on the job-wide counter (one per wave: 26 K atomics on
    // one address for 1.7 M edges, ~11 ns each, were 0.3 ms)
    __shared__ uint32_t s_wave[16];
    __shared__ unsigned long long s_base;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6;
    const bool mine = e < E && roots[e] % n_parts == part;
    uint32_t u = 0, v = 0;
    if (mine) {
        u = uv[2 * e];
        v = uv[2 * e + 1];
        flags[u] = 1u;
        flags[v] = 1u;
    }
    const unsigned long long ballot = __ballot(mine);
    if (fqd_lane() == 0)
        s_wave[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < 16; w++) {
            const uint32_t c = s_wave[w];
            s_wave[w] = total;
            total += c;
        }
        s_base = total ? atomicAdd(n_sub, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (mine) {
        const uint64_t at = s_base + s_wave[wave] + __popcll(ballot & fqd_lanemask_lt());
        sub[2 * at] = u;
        sub[2 * at + 1] = v;
    }
}

// ---- ... with HOME clusters apart (sharded.py step 4, the in-place dissection): a cluster all of whose keys live in
// ONE rank's range of the unique table is dissected by that rank on the table it holds -- no key travels, no verdict
// comes back. span[root] = 1 for a cluster with an edge between two ranks' ranges (a connected cluster with keys on
// two ranks has one); this rank then takes (a) the edges of its home clusters, ends renumbered to rows of its own
// table, and (b) its share (root % n_parts == part) of the spanning clusters, as above.
__device__ __forceinline__ uint32_t uid_rank(const fqd::UidBounds &b, uint32_t u)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t k = 1; k < FQD_MAX_HOME_RANKS; k++)
        r += (k < b.n && u >= b.lo[k]) ? 1u : 0u;
    return r;
}

__global__ void span_mark_kernel(const uint32_t *__restrict__ uv, const uint32_t *__restrict__ roots, uint64_t E,
                                 fqd::UidBounds bounds, uint8_t *__restrict__ span)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E)
        return;
    const uint2 ends = reinterpret_cast<const uint2 *>(uv)[e];
    if (uid_rank(bounds, ends.x) != uid_rank(bounds, ends.y))
        span[roots[e]] = 1;
}

__global__ __launch_bounds__(1024) void subgraph_mark_home_kernel(
    const uint32_t *__restrict__ uv, const uint32_t *__restrict__ roots, uint64_t E, uint32_t n_parts, uint32_t part,
    fqd::UidBounds bounds, const uint8_t *__restrict__ span, uint32_t *__restrict__ flags, uint32_t *__restrict__ sub,
    unsigned long long *__restrict__ n_sub, uint32_t *__restrict__ home, unsigned long long *__restrict__ n_home,
    unsigned long long *__restrict__ n_span /* edges of spanning clusters, whoever dissects them: the same on every rank */)
{
    __shared__ uint32_t s_wave[2][16];
    __shared__ unsigned long long s_base[2];
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t u = 0, v = 0;
    bool mine = false, at_home = false, spanning = false;
    if (e < E) {
        const uint2 ends = reinterpret_cast<const uint2 *>(uv)[e];
        u = ends.x;
        v = ends.y;
        const uint32_t root = roots[e];
        spanning = span[root] != 0;
        if (spanning)
            mine = root % n_parts == part;
        else
            at_home = uid_rank(bounds, u) == part;
    }
    {
        const unsigned long long bs = __ballot(spanning);
        if (bs && fqd_lane() == 0)
            atomicAdd(n_span, (unsigned long long)__popcll(bs));
    }
    if (mine) {
        flags[u] = 1u;
        flags[v] = 1u;
    }
    const unsigned long long b0 = __ballot(mine), b1 = __ballot(at_home);
    if (fqd_lane() == 0) {
        s_wave[0][wave] = (uint32_t)__popcll(b0);
        s_wave[1][wave] = (uint32_t)__popcll(b1);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < 16; w++) {
            const uint32_t c = s_wave[threadIdx.x][w];
            s_wave[threadIdx.x][w] = total;
            total += c;
        }
        s_base[threadIdx.x] = total ? atomicAdd(threadIdx.x ? n_home : n_sub, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (mine) {
        const uint64_t at = s_base[0] + s_wave[0][wave] + __popcll(b0 & fqd_lanemask_lt());
        sub[2 * at] = u;
        sub[2 * at + 1] = v;
    }
    if (at_home) {
        const uint64_t at = s_base[1] + s_wave[1][wave] + __popcll(b1 & fqd_lanemask_lt());
        home[2 * at] = u - bounds.lo[part];
        home[2 * at + 1] = v - bounds.lo[part];
    }
}

// the verdicts of clusters dissected elsewhere, over a dissection of this table in which those keys stood alone:
// "dropped" in every method's reading of best / state (kept_verdict)
__global__ void mark_dropped_after_kernel(uint8_t *state, uint32_t *best, uint64_t U, const uint32_t *__restrict__ dropped,
                                          uint64_t n, uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t v = dropped[i];
    if (v < U) {
        state[v] = 2;
        best[v] = 0xFFFFFFFFu;
    } else {
        *bad = 1;
    }
}

__global__ void subgraph_nodes_kernel(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ flags_incl,
                                      uint64_t n_nodes, uint32_t *__restrict__ touched)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes && flags[i])
        touched[flags_incl[i] - 1] = (uint32_t)i;
}

__global__ void subgraph_renumber_kernel(uint32_t *__restrict__ sub, const unsigned long long *__restrict__ n_sub,
                                         const uint32_t *__restrict__ flags_incl)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * *n_sub)
        sub[i] = flags_incl[sub[i]] - 1;
}

__global__ void check_indices_kernel(const uint32_t *__restrict__ idx, uint64_t n, uint64_t limit, uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= limit)
        *bad = 1;
}

// state[v] = 2 for every listed v (the dissection's verdict for v, computed elsewhere: "dropped")
__global__ void mark_dropped_kernel(uint8_t *state, uint64_t U, const uint32_t *__restrict__ dropped, uint64_t n,
                                    uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t v = dropped[i];
    if (v < U)
        state[v] = 2;
    else
        *bad = 1;
}

__global__ __launch_bounds__(256) void uf_flatten_kernel(uint32_t *parent, uint64_t U, unsigned long long *n_roots)
{
    // grid-stride so that the single root counter sees one atomic per wave of a capped grid
    unsigned long long roots = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i, p = load_relaxed(&parent[x]);
        while (p != x) {
            x = p;
            p = load_relaxed(&parent[x]);
        }
        // writing the root early is harmless: it is a valid ancestor for every reader
        __hip_atomic_store(&parent[i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        roots += x == (uint32_t)i ? 1ull : 0ull;
    }
    for (int o = 32; o; o >>= 1)
        roots += __shfl_xor(roots, o);
    if (fqd_lane() == 0 && roots)
        atomicAdd(n_roots, roots);
}

// ---- dissection ------------------------------------------------------------------
// The state byte of the closed-form directional dissection carries the key's COUNT in its high nibble (15: "15 or
// more, look it up"), its verdict bits in the low one (0 kept so far, 2, 3, bit 8: see directional_edges_kernel). One
// random byte per end of an edge then answers both "how many copies" and "what do we know": the pass over the edges
// fetched a count sector AND a state sector per end before (3.9 GB for the 9.3 M edges of config 4).
__device__ __forceinline__ uint32_t dstate_init(uint32_t count) { return (count < 15u ? count : 15u) << 4; }
// verdict bits into the low nibble of state[v], whatever the other bytes of its word are doing
__device__ __forceinline__ void dstate_or(uint8_t *state, uint32_t v, uint32_t bits)
{
    atomicOr(reinterpret_cast<uint32_t *>(state) + (v >> 2), bits << (8u * (v & 3u)));
}

__global__ void dstate_init_kernel(uint8_t *__restrict__ state, const uint32_t *__restrict__ ucounts, uint64_t U)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < U)
        state[i] = (uint8_t)dstate_init(ucounts[i]);
}

// Everything stages 4 and 5 set up over the unique table, in ONE launch (fqd_api_graph_preinit: five
// launches and two fills of ~5 us each, plus the gaps between them, were 70 us of a 2.9 ms job):
// parent[i] = i (components), best[i] = i, state[i] = 0 (dissection), and for the closed-form
// directional dissection (root_taint != NULL) root_taint[i] = 0 and state[i] = the key's count nibble; block 0 also
// clears the hook counters.
__global__ void graph_preinit_kernel(uint32_t *__restrict__ parent, uint32_t *__restrict__ best,
                                     uint8_t *__restrict__ state, uint8_t *__restrict__ root_taint /* may be NULL */, uint64_t U,
                                     unsigned long long *__restrict__ hook_slots, uint32_t hook_words,
                                     uint32_t *__restrict__ zero32 /* may be NULL */, uint32_t zero32_words,
                                     unsigned long long *__restrict__ zero64_a, unsigned long long *__restrict__ zero64_b,
                                     unsigned long long *__restrict__ zero64_c,
                                     const uint32_t *__restrict__ ucounts /* with root_taint: state[i] = count nibble (dstate_init) */)
{
    // four keys per thread: 16-byte stores for the word arrays, 4-byte stores for the byte arrays
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = q * 4;
    if (blockIdx.x == 0) {
        for (uint32_t w = threadIdx.x; w < hook_words; w += blockDim.x)
            hook_slots[w] = 0ull;
        // ... and what the tail of the job would clear with launches of its own: the cursors of the kept-id bins, the
        // counters of the kept keys and of the dissection's edge list
        if (zero32)
            for (uint32_t w = threadIdx.x; w < zero32_words; w += blockDim.x)
                zero32[w] = 0u;
        if (threadIdx.x == 0 && zero64_a)
            *zero64_a = 0ull;
        if (threadIdx.x == 0 && zero64_b)
            *zero64_b = 0ull;
        if (threadIdx.x == 0 && zero64_c)
            *zero64_c = 0ull;
    }
    if (i0 + 4 <= U) {
        const uint4 v = make_uint4((uint32_t)i0, (uint32_t)i0 + 1, (uint32_t)i0 + 2, (uint32_t)i0 + 3);
        reinterpret_cast<uint4 *>(parent)[q] = v;
        reinterpret_cast<uint4 *>(best)[q] = v;
        uint32_t st4 = 0u;
        if (root_taint && ucounts) {
            const uint4 c4 = reinterpret_cast<const uint4 *>(ucounts)[q];
            st4 = dstate_init(c4.x) | dstate_init(c4.y) << 8 | dstate_init(c4.z) << 16 | dstate_init(c4.w) << 24;
        }
        reinterpret_cast<uint32_t *>(state)[q] = st4;
        if (root_taint)
            reinterpret_cast<uint32_t *>(root_taint)[q] = 0u;
    } else {
        for (uint64_t i = i0; i < U; i++) {
            parent[i] = (uint32_t)i;
            best[i] = (uint32_t)i;
            state[i] = root_taint && ucounts ? (uint8_t)dstate_init(ucounts[i]) : (uint8_t)0;
            if (root_taint)
                root_taint[i] = 0;
        }
    }
}

__global__ void dissect_init_kernel(uint32_t *best, uint8_t *state, uint64_t U)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < U) {
        best[i] = (uint32_t)i;
        state[i] = 0;
    }
}

// best[dst] = max_rank(best[dst], cand)
__device__ __forceinline__ bool raise_best(uint32_t *best, uint32_t dst, uint32_t cand,
                                           const uint32_t *__restrict__ ucounts, const uint32_t *__restrict__ urecs,
                                           const uint32_t *__restrict__ ulens, const KeyShape &sh)
{
    for (;;) {
        const uint32_t old = load_relaxed(&best[dst]);
        if (old == cand || !rank_greater(cand, old, ucounts, urecs, ulens, sh))
            return false;
        if (atomicCAS(&best[dst], old, cand) == old)
            return true;
    }
}

__global__ void highest_count_kernel(const uint32_t *__restrict__ labels, const uint32_t *__restrict__ ucounts,
                                     const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens,
                                     KeyShape sh, uint64_t U, uint32_t *best)
{
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= U)
        return;
    const uint32_t r = labels[v];
    if (r != (uint32_t)v)
        raise_best(best, r, (uint32_t)v, ucounts, urecs, ulens, sh);
}

// One relaxation sweep over the edges. stamp[v] = last round in which best[v] rose; from the
// second round on, an edge is revisited only if one of its ends rose in the previous round
// (anything that rises during this round is stamped with it and revisited in the next).
__global__ void directional_round_kernel(const uint32_t *__restrict__ edges, uint64_t E,
                                         const uint32_t *__restrict__ ucounts, const uint32_t *__restrict__ urecs,
                                         const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t *best,
                                         uint32_t *stamp, uint32_t round, uint32_t *changed)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool moved = false;
    if (e < E) {
        const uint32_t u = edges[2 * e], v = edges[2 * e + 1];
        if (round == 1 || load_relaxed(&stamp[u]) + 1 == round || load_relaxed(&stamp[v]) + 1 == round) {
            const long long cu = ucounts[u], cv = ucounts[v];
            if (2 * cv - 1 <= cu && raise_best(best, v, load_relaxed(&best[u]), ucounts, urecs, ulens, sh)) {  // arc u -> v
                stamp[v] = round;
                moved = true;
            }
            if (2 * cu - 1 <= cv && raise_best(best, u, load_relaxed(&best[v]), ucounts, urecs, ulens, sh)) {  // arc v -> u
                stamp[u] = round;
                moved = true;
            }
        }
    }
    // one store per wave, not one per lane, on the round's single flag word
    if (__ballot(moved) && fqd_lane() == 0)
        *changed = 1;
}

// ---- directional, closed form -------------------------------------------------------
// A count-1 key is kept iff its COMPONENT holds count-1 keys only and it is the component's largest key: a maximal
// connected set S of count-1 keys that touches no bigger key has no edge leaving it at all (a neighbour outside S would
// be a count-1 key, hence in S, or a bigger key), so S is a whole component -- and a set that does touch one is dropped
// entirely. The sets therefore need no union-find of their own: the components' parents (uf_union_kernel, which runs
// beside pass 1 and must have finished before pass 2) are their parents. (Rounds 1-3 kept a second union-find over the
// edges between count-1 keys: 3.4 M unions of config 4's 9.35 M edges, 0.32 ms, and 52 MB of parents to set up.)
// Pass 1 over the edges: in-arcs of keys with count >= 2 (state = 2: dropped), count-1 keys that
// touch a bigger key (state = 3: tainted); the edges between count-1 keys (few) are listed for pass 2. Four edges per
// thread, one list reservation per workgroup.
constexpr uint32_t DE_EPT = 4;

__global__ __launch_bounds__(256) void directional_edges_kernel(const uint32_t *__restrict__ edges, uint64_t E,
                                                                const uint32_t *__restrict__ ucounts, uint8_t *state,
                                                                uint32_t *__restrict__ list11,
                                                                unsigned long long *__restrict__ list11_count)
{
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    uint32_t uu[DE_EPT], vv[DE_EPT], cu[DE_EPT], cv[DE_EPT];
    bool live[DE_EPT];
#pragma unroll
    for (uint32_t t = 0; t < DE_EPT; t++) {       // the edge and count gathers of all four edges in flight
        const uint64_t e = ((uint64_t)blockIdx.x * DE_EPT + t) * blockDim.x + threadIdx.x;
        live[t] = e < E;
        const uint2 uv = reinterpret_cast<const uint2 *>(edges)[min(e, E - 1)];    // (clamped, unconditional: in flight together)
        uu[t] = uv.x;
        vv[t] = uv.y;
    }
    uint8_t su[DE_EPT], sv[DE_EPT];
#pragma unroll
    for (uint32_t t = 0; t < DE_EPT; t++) {
        live[t] = live[t] && uu[t] != vv[t];
        su[t] = state[uu[t]];                       // count nibble + what is known of the key so far: ONE byte per end
        sv[t] = state[vv[t]];
    }
#pragma unroll
    for (uint32_t t = 0; t < DE_EPT; t++) {
        cu[t] = su[t] >> 4;
        cv[t] = sv[t] >> 4;
        if (cu[t] == 15u)                           // (15 or more copies: the count itself -- few keys)
            cu[t] = ucounts[uu[t]];
        if (cv[t] == 15u)
            cv[t] = ucounts[vv[t]];
    }
    uint32_t rank[DE_EPT];
#pragma unroll
    for (uint32_t t = 0; t < DE_EPT; t++) {
        rank[t] = 0xFFFFFFFFu;
        if (!live[t])
            continue;
        const uint32_t u = uu[t], v = vv[t];
        if (cu[t] == 1 && cv[t] == 1) {
            rank[t] = atomicAdd(&s_n, 1u);      // (an edge between count-1 keys: listed)
            continue;
        }
        // (an OR only where the byte -- as fetched above -- does not say so yet: the 786 K edges of a 65 536-key
        // component are 1.5 M updates of 64 KB otherwise, queueing on the same lines)
        const long long lu = cu[t], lv = cv[t];
        const uint32_t ku = su[t] & 15u, kv = sv[t] & 15u;
        if (lv >= 2 && 2 * lv - 1 <= lu && kv != 2)
            dstate_or(state, v, 2u);          // arc u -> v from a key of larger count
        if (lu >= 2 && 2 * lu - 1 <= lv && ku != 2)
            dstate_or(state, u, 2u);
        if (lu == 1 && ku != 3)
            dstate_or(state, u, 3u);          // here cv >= 2: v reaches u and outranks all of u's count-1 set
        if (lv == 1 && kv != 3)
            dstate_or(state, v, 3u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_n)
        s_base = (uint32_t)atomicAdd(list11_count, (unsigned long long)s_n);
    __syncthreads();
#pragma unroll
    for (uint32_t t = 0; t < DE_EPT; t++)
        if (rank[t] != 0xFFFFFFFFu)
            list11[s_base + rank[t]] =
                (uint32_t)(((uint64_t)blockIdx.x * DE_EPT + t) * blockDim.x + threadIdx.x);
}

// ---- components AND pass 1 of the closed-form directional dissection in ONE sweep over the edges, on NODE RECORDS ----
// node[x] = (parent of x, state byte of x): what uf_union_kernel and directional_edges_kernel fetch per end of an edge
// -- a parent word here, a state byte there, two random sectors per end -- lies in ONE 8-byte record, one sector per
// end. Both kernels sit on the memory system's random-access rate (57.7 G sectors/s, microbench row 3: four sectors per
// edge are 0.65 ms for config 4's 9.35 M edges, and the two kernels side by side took 0.8); the sectors halve. The
// records are taken apart again (unzip_nodes_kernel: parents to the labels array, states to the state array) for
// everything downstream, which is unchanged. fqd_cluster[_keys] with the closed-form directional dissection takes this
// way (FQD_NO_NODE_RECORDS=1: the two kernels on their arrays, as the stage-by-stage calls do).
template <bool FRESH>
__device__ __forceinline__ uint32_t uf_find2(uint32_t *node, uint32_t x)
{
    uint32_t p = FRESH ? load_relaxed(&node[2 * (size_t)x]) : node[2 * (size_t)x];
    while (p != x) {
        const uint32_t g = FRESH ? load_relaxed(&node[2 * (size_t)p]) : node[2 * (size_t)p];
        if (g != p)
            __hip_atomic_store(&node[2 * (size_t)x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (path halving, see uf_find)
        x = p;
        p = g;
    }
    return x;
}

constexpr uint32_t UD_EPT = 4;

__global__ __launch_bounds__(256) void union_directional_kernel(uint32_t *node /* [U][2] */, const uint32_t *__restrict__ edges,
                                                                uint64_t E, const uint32_t *__restrict__ ucounts,
                                                                uint32_t *__restrict__ list11,
                                                                unsigned long long *__restrict__ list11_count,
                                                                unsigned long long *n_hooks, uint32_t phase)
{
    __shared__ uint32_t s_slot[4][128], s_res[4][64];
    __shared__ uint32_t s_n, s_base;
    const uint32_t lane = fqd_lane(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    uint32_t uu[UD_EPT], vv[UD_EPT];
    bool live[UD_EPT];
#pragma unroll
    for (uint32_t t = 0; t < UD_EPT; t++) {           // the edges, then both ends' records of all four, in flight together
        const uint64_t e = ((uint64_t)blockIdx.x * UD_EPT + t) * blockDim.x + threadIdx.x;
        // (phases as in uf_union_kernel: 0 every edge; 1 every 16th, 2 the rest -- a context that has met a giant component)
        live[t] = e < E && (phase == 0 || ((e & 15u) == 0) == (phase == 1));
        const uint2 uv = reinterpret_cast<const uint2 *>(edges)[min(e, E - 1)];
        uu[t] = uv.x;
        vv[t] = uv.y;
    }
    uint2 nu[UD_EPT], nv[UD_EPT];
#pragma unroll
    for (uint32_t t = 0; t < UD_EPT; t++) {
        live[t] = live[t] && uu[t] != vv[t];
        nu[t] = reinterpret_cast<const uint2 *>(node)[uu[t]];
        nv[t] = reinterpret_cast<const uint2 *>(node)[vv[t]];
    }
    // ---- pass 1 of the dissection (directional_edges_kernel on the records' state bytes)
    uint32_t rank[UD_EPT];
#pragma unroll
    for (uint32_t t = 0; t < UD_EPT; t++) {
        rank[t] = 0xFFFFFFFFu;
        if (!live[t])
            continue;
        const uint32_t su = nu[t].y & 0xFFu, sv = nv[t].y & 0xFFu;
        uint32_t cu = su >> 4, cv = sv >> 4;
        if (cu == 15u)                              // (15 or more copies: the count itself -- few keys)
            cu = ucounts[uu[t]];
        if (cv == 15u)
            cv = ucounts[vv[t]];
        if (cu == 1 && cv == 1) {
            rank[t] = atomicAdd(&s_n, 1u);          // (an edge between count-1 keys: listed)
            continue;
        }
        const long long lu = cu, lv = cv;
        const uint32_t ku = su & 15u, kv = sv & 15u;
        if (lv >= 2 && 2 * lv - 1 <= lu && kv != 2)
            atomicOr(&node[2 * (size_t)vv[t] + 1], 2u);      // arc u -> v from a key of larger count
        if (lu >= 2 && 2 * lu - 1 <= lv && ku != 2)
            atomicOr(&node[2 * (size_t)uu[t] + 1], 2u);
        if (lu == 1 && ku != 3)
            atomicOr(&node[2 * (size_t)uu[t] + 1], 3u);      // here cv >= 2: v reaches u and outranks all of u's count-1 set
        if (lv == 1 && kv != 3)
            atomicOr(&node[2 * (size_t)vv[t] + 1], 3u);
    }
    // ---- the unions (uf_union_kernel: lanes that hang the same root under the same node send ONE compare-and-swap)
    uint32_t hooks = 0, walks_again = 0;
#pragma unroll
    for (uint32_t t = 0; t < UD_EPT; t++) {
        // the first step of either walk is in hand: the record's parent word
        uint32_t a = nu[t].x, b = nv[t].x;
        bool active = live[t];
        bool hooked = false;
        for (bool fresh = false; __ballot(active); fresh = true) {
            if (fresh)
                walks_again += (uint32_t)__popcll(__ballot(active));
            if (active) {
                a = fresh ? uf_find2<true>(node, a) : uf_find2<false>(node, a);
                b = fresh ? uf_find2<true>(node, b) : uf_find2<false>(node, b);
                if (a == b)
                    active = false;
                if (a > b) {
                    const uint32_t x = a;
                    a = b;
                    b = x;
                }
            }
            s_slot[wave][lane] = 0xFFFFFFFFu;
            s_slot[wave][lane + 64] = 0xFFFFFFFFu;
            const uint32_t h = ((b * 0x9E3779B1u) ^ (a * 0x85EBCA6Bu)) >> 25;
            if (active)
                atomicMin(&s_slot[wave][h], lane);
            const uint32_t owner = active ? s_slot[wave][h] : lane;
            const uint32_t ob = __shfl(b, owner & 63u), oa = __shfl(a, owner & 63u);
            const bool follower = active && owner != lane && ob == b && oa == a;
            uint32_t res = 0xFFFFFFFFu;
            if (active && !follower)
                res = atomicCAS(&node[2 * (size_t)b], b, a);        // hook the larger root under the smaller one
            s_res[wave][lane] = res;
            if (follower)
                res = s_res[wave][owner];
            if (active && res == b) {
                active = false;
                hooked = !follower;
            }
        }
        hooks += (uint32_t)__popcll(__ballot(hooked));
    }
    if (n_hooks && lane == 0) {
        if (hooks)
            atomicAdd(n_hooks + (size_t)(blockIdx.x % FQD_HOOK_SLOTS) * 8, (unsigned long long)hooks);
        if (walks_again)
            atomicAdd(n_hooks + (size_t)(blockIdx.x % FQD_HOOK_SLOTS) * 8 + 1, (unsigned long long)walks_again);
    }
    // ---- the listed edges: one reservation per workgroup
    __syncthreads();
    if (threadIdx.x == 0 && s_n)
        s_base = (uint32_t)atomicAdd(list11_count, (unsigned long long)s_n);
    __syncthreads();
#pragma unroll
    for (uint32_t t = 0; t < UD_EPT; t++)
        if (rank[t] != 0xFFFFFFFFu)
            list11[s_base + rank[t]] = (uint32_t)(((uint64_t)blockIdx.x * UD_EPT + t) * blockDim.x + threadIdx.x);
}

// node records -> the parents (labels array) and the state bytes: four keys per thread
__global__ void unzip_nodes_kernel(const uint32_t *__restrict__ node, uint32_t *__restrict__ parent, uint8_t *__restrict__ state,
                                   uint64_t U)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = q * 4;
    if (i0 + 4 <= U) {
        const uint4 lo = reinterpret_cast<const uint4 *>(node)[2 * q], hi = reinterpret_cast<const uint4 *>(node)[2 * q + 1];
        reinterpret_cast<uint4 *>(parent)[q] = make_uint4(lo.x, lo.z, hi.x, hi.z);
        reinterpret_cast<uint32_t *>(state)[q] = (lo.y & 0xFFu) | (lo.w & 0xFFu) << 8 | (hi.y & 0xFFu) << 16 | (hi.w & 0xFFu) << 24;
    } else {
        for (uint64_t i = i0; i < U; i++) {
            parent[i] = node[2 * i];
            state[i] = (uint8_t)node[2 * i + 1];
        }
    }
}

// node[i] = (i, the count nibble of key i); with best[i] = i, root_taint[i] = 0 and the counters graph_preinit_kernel clears
__global__ void graph_preinit_nodes_kernel(uint32_t *__restrict__ node, uint32_t *__restrict__ best, uint8_t *__restrict__ root_taint,
                                           const uint32_t *__restrict__ ucounts, uint64_t U,
                                           unsigned long long *__restrict__ hook_slots, uint32_t hook_words,
                                           uint32_t *__restrict__ zero32, uint32_t zero32_words,
                                           unsigned long long *__restrict__ zero64_a, unsigned long long *__restrict__ zero64_b,
                                           unsigned long long *__restrict__ zero64_c)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = q * 4;
    if (blockIdx.x == 0) {
        for (uint32_t w = threadIdx.x; w < hook_words; w += blockDim.x)
            hook_slots[w] = 0ull;
        if (zero32)
            for (uint32_t w = threadIdx.x; w < zero32_words; w += blockDim.x)
                zero32[w] = 0u;
        if (threadIdx.x == 0 && zero64_a)
            *zero64_a = 0ull;
        if (threadIdx.x == 0 && zero64_b)
            *zero64_b = 0ull;
        if (threadIdx.x == 0 && zero64_c)
            *zero64_c = 0ull;
    }
    if (i0 + 4 <= U) {
        const uint4 c4 = reinterpret_cast<const uint4 *>(ucounts)[q];
        const uint32_t i = (uint32_t)i0;
        reinterpret_cast<uint4 *>(node)[2 * q] = make_uint4(i, dstate_init(c4.x), i + 1, dstate_init(c4.y));
        reinterpret_cast<uint4 *>(node)[2 * q + 1] = make_uint4(i + 2, dstate_init(c4.z), i + 3, dstate_init(c4.w));
        reinterpret_cast<uint4 *>(best)[q] = make_uint4(i, i + 1, i + 2, i + 3);
        reinterpret_cast<uint32_t *>(root_taint)[q] = 0u;
    } else {
        for (uint64_t i = i0; i < U; i++) {
            node[2 * i] = (uint32_t)i;
            node[2 * i + 1] = dstate_init(ucounts[i]);
            best[i] = (uint32_t)i;
            root_taint[i] = 0;
        }
    }
}

// Pass 1b, over the listed edges (both ends count 1), AFTER pass 1 has marked every count-1 key that touches a bigger
// one (state 3, "tainted": dropped whatever its set looks like): the edge stays on the list of pass 2 only if one of its
// ends is NOT tainted. An edge between two tainted keys has nothing more to say -- both are dropped -- and the taint of
// a set reaches its root through the listed edges: a set with a tainted and an untainted member has an edge between
// such a pair on the path that joins them. With d = 2 most listed edges join two error variants of one molecule, both
// next to the molecule's key (config 4: 3.4 M edges listed, a tenth of them left for pass 2).
__global__ __launch_bounds__(256) void directional_filter_kernel(const uint32_t *__restrict__ edges,
                                                                 const uint32_t *__restrict__ list11,
                                                                 const unsigned long long *__restrict__ list11_count,
                                                                 const uint8_t *__restrict__ state,
                                                                 uint32_t *__restrict__ list_out,
                                                                 unsigned long long *__restrict__ list_out_count)
{
    __shared__ uint32_t s_n, s_base;
    const uint64_t n = *list11_count;
    constexpr uint32_t UE = 4;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x * UE; base < n; base += (uint64_t)gridDim.x * blockDim.x * UE) {
        if (threadIdx.x == 0)
            s_n = 0;
        __syncthreads();
        uint32_t e[UE];
        uint2 uv[UE];
        uint8_t su[UE], sv[UE];
        bool live[UE];
#pragma unroll
        for (uint32_t t = 0; t < UE; t++) {           // (clamped, unconditional: in flight together)
            const uint64_t i = base + (uint64_t)t * blockDim.x + threadIdx.x;
            live[t] = i < n;
            e[t] = list11[min(i, n - 1)];
        }
#pragma unroll
        for (uint32_t t = 0; t < UE; t++)
            uv[t] = reinterpret_cast<const uint2 *>(edges)[e[t]];
#pragma unroll
        for (uint32_t t = 0; t < UE; t++) {
            su[t] = state[uv[t].x];
            sv[t] = state[uv[t].y];
        }
        uint32_t rank[UE];
#pragma unroll
        for (uint32_t t = 0; t < UE; t++) {
            rank[t] = 0xFFFFFFFFu;
            if (live[t] && ((su[t] & 15u) != 3u || (sv[t] & 15u) != 3u))
                rank[t] = atomicAdd(&s_n, 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_n)
            s_base = (uint32_t)atomicAdd(list_out_count, (unsigned long long)s_n);
        __syncthreads();
#pragma unroll
        for (uint32_t t = 0; t < UE; t++)
            if (rank[t] != 0xFFFFFFFFu)
                list_out[s_base + rank[t]] = e[t];
        __syncthreads();
    }
}

// Pass 2 (after every union of pass 1), over the edges between count-1 keys only, in two launches.
// 2a: each end finds the root of its set (remembered for 2b), passes its taint on to it and is marked
// (state bit 8) as a member of a set of several keys. 2b: the ends of UNTAINTED sets report their
// key to the root (CAS-max by key: a comparison of two whole keys per try) -- a tainted set is
// dropped whatever its largest key is, and most sets with edges are tainted. A count-1 key without
// such an edge is a set of its own, and the state byte alone is then the verdict of EVERY key: 0
// kept, 2 dropped (count >= 2 with an in-arc), 3 dropped (count 1 next to a bigger key), bit 8: ask
// the root.
__global__ void directional_roots_kernel(const uint32_t *__restrict__ edges,
                                         const uint32_t *__restrict__ list11,
                                         const unsigned long long *__restrict__ list11_count,
                                         const uint32_t *__restrict__ parent1, uint8_t *state,
                                         uint8_t *root_taint, uint32_t *__restrict__ roots /* 2 per listed edge */)
{
    const uint64_t n = *list11_count;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = list11[i];
        const uint32_t ends[2] = {edges[2 * e], edges[2 * e + 1]};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t x = ends[k];
            // every union happened in pass 1 (an earlier launch): a plain read-only walk to the root,
            // no path halving (those are device-scope atomics)
            uint32_t r = x, pr = parent1[r];
            while (pr != r) {
                r = pr;
                pr = parent1[r];
            }
            roots[2 * i + k] = r;
            const uint8_t sx = state[x];              // (high nibble: the count, see dstate_init)
            if ((sx & 7) == 3)
                root_taint[r] = 1;
            if (!(sx & 8))
                state[x] = sx | 8;     // member of a set of several count-1 keys: its verdict is its root's
                                       // (every writer of this byte in this launch writes this same value)
        }
    }
}

// a > b for two DIFFERENT keys of one count, every word of both records requested before the first is looked at
// (fqd_key_cmp walks the words one dependent round trip after the other -- ten for two 300-nt keys that differ near
// their ends, and neighbours differ anywhere). Records of up to 8 uint4; longer ones: the walk.
__device__ __forceinline__ bool key_greater_inflight(uint32_t a, uint32_t b, const uint32_t *__restrict__ urecs,
                                                     const uint32_t *__restrict__ ulens, const KeyShape &sh)
{
    const uint32_t q4 = sh.stride / 4u;
    const uint32_t *ra = urecs + (uint64_t)a * sh.stride, *rb = urecs + (uint64_t)b * sh.stride;
    if (q4 > 8u || (sh.stride & 3u))
        return fqd_key_cmp(ra, fqd_key_len(sh, ulens, a), rb, fqd_key_len(sh, ulens, b), sh.planes, sh.words) > 0;
    uint4 va[8], vb[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {              // (clamped, unconditional: in flight together)
        va[q] = reinterpret_cast<const uint4 *>(ra)[min(q, q4 - 1u)];
        vb[q] = reinterpret_cast<const uint4 *>(rb)[min(q, q4 - 1u)];
    }
    // the first record word (of stride) in which they differ; word j belongs to 32-base word j / K
    uint32_t first = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 7; q >= 0; q--) {
        if ((uint32_t)q < q4) {
            if (va[q].w != vb[q].w) first = 4u * q + 3u;
            if (va[q].z != vb[q].z) first = 4u * q + 2u;
            if (va[q].y != vb[q].y) first = 4u * q + 1u;
            if (va[q].x != vb[q].x) first = 4u * q + 0u;
        }
    }
    // (ragged & 2: the rows' last word holds the key's length -- va[7] is the row's last uint4, the clamp above -- no look-up)
    const uint32_t la = (sh.ragged & 2u) ? va[7].w : fqd_key_len(sh, ulens, a);
    const uint32_t lb = (sh.ragged & 2u) ? vb[7].w : fqd_key_len(sh, ulens, b);
    if (first == 0xFFFFFFFFu || first >= sh.planes * sh.words)
        return la > lb;                              // (equal key words: one key is a prefix of the other)
    // (the words of that 32-base word again, from the cache: no register array indexed by a run-time value)
    const uint32_t w = first / sh.planes;
    const uint32_t d = fqd_diff_word_dyn(ra, rb, w, sh.planes);
    const uint32_t bit = (uint32_t)__ffs((int)d) - 1u, pos = w * 32u + bit;
    if (pos >= la || pos >= lb)
        return la > lb;
    return fqd_code_at(ra, w, bit, sh.planes) > fqd_code_at(rb, w, bit, sh.planes);
}

// best[root] = the largest key among the ends of the listed edges of an untainted set (all of count 1)
__global__ void directional_best_kernel(const uint32_t *__restrict__ edges, const uint32_t *__restrict__ list11,
                                        const unsigned long long *__restrict__ list11_count,
                                        const uint32_t *__restrict__ ucounts, const uint32_t *__restrict__ urecs,
                                        const uint32_t *__restrict__ ulens, KeyShape sh,
                                        const uint32_t *__restrict__ roots, const uint8_t *__restrict__ root_taint,
                                        uint32_t *best)
{
    const uint64_t n = *list11_count;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = list11[i];
        const uint2 rr = reinterpret_cast<const uint2 *>(roots)[i];
        const uint2 xx = reinterpret_cast<const uint2 *>(edges)[e];
        const uint8_t t0 = root_taint[rr.x], t1 = root_taint[rr.y];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t r = k ? rr.y : rr.x, x = k ? xx.y : xx.x;
            if ((k ? t1 : t0) || r == x)
                continue;
            for (;;) {                               // (raise_best for two keys of count 1: the keys alone decide)
                const uint32_t old = load_relaxed(&best[r]);
                if (old == x || !key_greater_inflight(x, old, urecs, ulens, sh))
                    break;
                if (atomicCAS(&best[r], old, x) == old)
                    break;
            }
        }
    }
}

// Put the endpoint of higher rank first, once: the adjacency rounds then never compare keys.
__global__ void orient_edges_kernel(uint32_t *__restrict__ edges, uint64_t E, const uint32_t *__restrict__ ucounts,
                                    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens,
                                    KeyShape sh)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E)
        return;
    const uint2 ab = reinterpret_cast<const uint2 *>(edges)[e];
    const uint32_t a = ab.x, b = ab.y;
    if (a == b)
        return;
    // counts first; keys of one count are compared with all their words in flight (word by word: one dependent round
    // trip per 32 bases -- 0.68 ms for config 5's 11 M edges)
    const uint32_t ca = ucounts[a], cb = ucounts[b];
    const bool a_first = ca != cb ? ca > cb : key_greater_inflight(a, b, urecs, ulens, sh);
    if (!a_first)
        reinterpret_cast<uint2 *>(edges)[e] = make_uint2(b, a);
}

// state: 0 undecided, 1 kept, 2 dropped. blocked[v] == round: v still has an
// undecided neighbour of higher rank. Edges are (higher rank, lower rank).
__global__ void adjacency_edges_kernel(const uint32_t *__restrict__ edges, uint64_t E, uint8_t *state,
                                       uint32_t *blocked, uint32_t round, uint32_t *changed)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool moved = false;
    if (e < E) {
        const uint32_t hi = edges[2 * e], lo = edges[2 * e + 1];
        if (hi != lo && state[lo] == 0) {
            const uint8_t s_hi = state[hi];
            if (s_hi == 1) {
                state[lo] = 2;
                moved = true;
            } else if (s_hi == 0) {
                blocked[lo] = round;
            }
        }
    }
    if (__ballot(moved) && fqd_lane() == 0)
        *changed = 1;
}

// The edges that can still matter after some sweeps: the lower end undecided, the higher end not
// dropped. (Every sweep over ALL edges costs two random state reads per edge; after the first two
// sweeps a few percent of the edges are left.) One list reservation per workgroup.
__global__ __launch_bounds__(256) void adjacency_live_edges_kernel(const uint32_t *__restrict__ edges, uint64_t E,
                                                                   const uint8_t *__restrict__ state,
                                                                   uint32_t *__restrict__ live,
                                                                   unsigned long long *__restrict__ live_count)
{
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    uint2 uv[4];
    uint32_t rank[4];
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        const uint64_t e = ((uint64_t)blockIdx.x * 4 + t) * blockDim.x + threadIdx.x;
        uv[t] = e < E ? reinterpret_cast<const uint2 *>(edges)[e] : make_uint2(0, 0);
    }
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        rank[t] = 0xFFFFFFFFu;
        if (uv[t].x != uv[t].y && state[uv[t].y] == 0 && state[uv[t].x] != 2)
            rank[t] = atomicAdd(&s_n, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_n)
        s_base = (uint32_t)atomicAdd(live_count, (unsigned long long)s_n);
    __syncthreads();
#pragma unroll
    for (uint32_t t = 0; t < 4; t++)
        if (rank[t] != 0xFFFFFFFFu)
            reinterpret_cast<uint2 *>(live)[s_base + rank[t]] = uv[t];
}

__global__ void adjacency_nodes_kernel(uint64_t U, uint8_t *state, const uint32_t *__restrict__ blocked,
                                       uint32_t round, uint32_t *changed)
{
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= U)
        return;
    if (state[v] == 0 && blocked[v] != round) {
        state[v] = 1;
        *changed = 1;
    }
}

// The dissection's verdict on unique key v (method 3: directional, closed form).
__device__ __forceinline__ bool kept_verdict(int method, uint32_t v, const uint32_t *__restrict__ labels,
                                             const uint32_t *__restrict__ best, const uint8_t *__restrict__ state,
                                             const uint32_t *__restrict__ ucounts,
                                             const uint32_t *__restrict__ parent1,
                                             const uint8_t *__restrict__ root_taint)
{
    if (method == 0)
        return best[labels[v]] == v;
    if (method == 2)
        return best[v] == v;
    if (method == 3) {
        const uint8_t st = state[v] & 15u;        // (the high nibble is the key's count, see dstate_init)
        if (!(st & 8))
            return st == 0;        // 2: in-arc from a bigger key, 3: count 1 next to a bigger key
        if ((st & 7) == 3)
            return false;
        uint32_t r = v, p = parent1[r];
        while (p != r) {
            r = p;
            p = parent1[r];
        }
        return !root_taint[r] && best[r] == v;
    }
    return state[v] == 1;
}

// N verdicts at once (their loads in flight together: key by key the kernel waits for a chain of
// dependent round trips per key, 0.2 ms for 14 M keys).
template <uint32_t N>
__device__ __forceinline__ void kept_verdicts(int method, const uint32_t (&v)[N], const bool (&valid)[N],
                                              const uint32_t *__restrict__ labels,
                                              const uint32_t *__restrict__ best, const uint8_t *__restrict__ state,
                                              const uint32_t *__restrict__ ucounts,
                                              const uint32_t *__restrict__ parent1,
                                              const uint8_t *__restrict__ root_taint, bool (&k)[N])
{
    if (method == 3) {
        // the state byte decides (pass 2 marked the few keys that have to ask their set's root)
        // (v[t] of an invalid entry is a clamped, readable index: the loads are unconditional, so that all N are in
        // flight together -- "valid ? state[v] : 2" compiles to N branches with a wait in each)
        uint8_t st[N];
#pragma unroll
        for (uint32_t t = 0; t < N; t++)
            st[t] = state[v[t]];
#pragma unroll
        for (uint32_t t = 0; t < N; t++) {
            st[t] &= 15u;                     // (the high nibble is the key's count)
            if (!valid[t])
                st[t] = 2;
            k[t] = st[t] == 0;
            if (st[t] & 8)
                k[t] = kept_verdict(3, v[t], labels, best, state, ucounts, parent1, root_taint);
        }
    } else {
#pragma unroll
        for (uint32_t t = 0; t < N; t++)     // (unconditional, as above)
            k[t] = kept_verdict(method, v[t], labels, best, state, ucounts, parent1, root_taint);
#pragma unroll
        for (uint32_t t = 0; t < N; t++)
            k[t] = k[t] && valid[t];
    }
}

// kept[v]: the dissection's verdict. A key is LISTED when it is kept AND its first holder lies in
// the id window [id_lo, id_hi) -- the ids this context lists (a rank lists its own reads):
// kept_u32[v] = 1 (scan + gather + sort path) or window_flags[id - id_lo] = 1 (compaction path).
__global__ __launch_bounds__(256) void kept_flags_kernel(int method, const uint32_t *__restrict__ labels,
                                                         const uint32_t *__restrict__ best,
                                                         const uint8_t *__restrict__ state,
                                                         const uint64_t *__restrict__ ufirst, uint64_t id_lo,
                                                         uint64_t id_hi, uint64_t U, uint8_t *kept, uint32_t *kept_u32,
                                                         uint8_t *window_flags, uint64_t window_size,
                                                         const uint32_t *__restrict__ ucounts,
                                                         const uint32_t *__restrict__ parent1,
                                                         const uint8_t *__restrict__ root_taint,
                                                         unsigned long long *n_kept_total)
{
    unsigned long long total = 0;
    // Four keys per thread and step: their loads are issued together, the verdicts computed, then
    // the stores go out (one key per step left every step waiting on its own chain of loads).
    constexpr uint32_t KF = 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t v0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < U; v0 += KF * stride) {
        bool k[KF];
        uint64_t id[KF];
#pragma unroll
        for (uint32_t t = 0; t < KF; t++) {
            const uint64_t v = v0 + t * stride;
            k[t] = false;
            id[t] = ufirst[min(v, U - 1)];
        }
        {
            uint32_t vv[KF];
            bool valid[KF];
#pragma unroll
            for (uint32_t t = 0; t < KF; t++) {
                vv[t] = (uint32_t)min(v0 + t * stride, U - 1);
                valid[t] = v0 + t * stride < U;
            }
            kept_verdicts<KF>(method, vv, valid, labels, best, state, ucounts, parent1, root_taint, k);
        }
#pragma unroll
        for (uint32_t t = 0; t < KF; t++) {
            const uint64_t v = v0 + t * stride;
            if (v >= U)
                continue;
            kept[v] = k[t] ? 1 : 0;
            const bool listed = k[t] && id[t] >= id_lo && id[t] < id_hi;
            if (window_flags) {
                // ids are distinct: one byte per id of the window, set for the listed ones; the
                // ascending id list is then a stream compaction of the window (no sort)
                if (listed && id[t] - id_lo < window_size)
                    window_flags[id[t] - id_lo] = 1;
            } else {
                kept_u32[v] = listed ? 1u : 0u;
            }
            total += k[t] ? 1ull : 0ull;
        }
    }
    for (int o = 32; o; o >>= 1)
        total += __shfl_xor(total, o);
    __shared__ unsigned long long s_total[4];
    if (fqd_lane() == 0)
        s_total[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) {             // one atomic per workgroup on the job-wide counter
        const unsigned long long sum = s_total[0] + s_total[1] + s_total[2] + s_total[3];
        if (sum)
            atomicAdd(n_kept_total, sum);
    }
}

// ---- ascending id list through id bins (no byte map of the window) ---------------------------
// Setting one byte per kept id in a map of the whole id window is 12 M scattered stores, each a
// 32-byte sector on the wire (0.27 of the 0.36 ms of kept_flags_kernel). Instead:
//   kept_bin_kernel   verdicts as in kept_flags_kernel; the listed ids of a tile of 4096 keys are
//                     partitioned in LDS by id bin (2^bin_shift consecutive ids; <= 512 bins) and leave as one
//                     run of 4-byte offsets per (tile, bin) behind an atomic cursor. A bin has KB_SUBS
//                     lists (KB_SUBS = 1: four lists per bin, tile t appending to list t % 4, changed nothing --
//                     0.240 against 0.247 ms -- so the cursors are not what bounds this kernel; the idea was that
//                     tile t appends to list t % KB_SUBS: 3400 tiles queueing on ONE cursor
//                     per bin cost 0.12 ms, a same-address atomic takes ~36 ns), each with room
//                     for every id of the bin (ids are distinct): no overflow.
//   kept_emit_kernel  four workgroups per bin: the bin's offsets set bits in LDS bit maps of the
//                     bin's quarters; the maps are scanned and the ids are written in ascending
//                     order behind the ids of everything before them.
constexpr uint32_t KB_KPT = FQD_KB_KPT, KB_MAX_BINS = 512;

// lists per id bin (tile t appends to list t % subs): FQD_KB_SUBS in the environment, else the compile-time default
static uint32_t kb_subs()
{
    static const uint32_t v = [] {
        const char *e = getenv("FQD_KB_SUBS");
        const int x = e ? atoi(e) : FQD_KB_SUBS;
        return (uint32_t)(x < 1 ? 1 : x > 8 ? 8 : x);
    }();
    return v;
}

// METHOD and the workgroup size are template parameters (with the method in a register the kernel carried the loads
// and registers of all four verdict rules). A thread takes KB_KPT / 4 groups of FOUR CONSECUTIVE keys: their first-
// holder ids are two 16-byte loads, their state bytes one dword, their kept flags one dword store (key by key: 16
// 8-byte loads, 16 byte loads and 16 byte stores per thread, each store instruction a 64-byte piece of a line);
// every load of the tile is issued before the first verdict (clamped addresses, no branch around a load). The
// staged offset names its own bin (offset >> bin_shift): no second staging array, so a 1024-thread workgroup stages
// 16 Ki offsets in 64 KB and a (tile, bin) run is 4 x as long as with 256 threads.
template <int METHOD, uint32_t THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 1024 ? 8 : 1) void kept_bin_kernel(
    uint32_t KB_SUBS, const uint32_t *__restrict__ labels, const uint32_t *__restrict__ best,
    const uint8_t *__restrict__ state, const uint64_t *__restrict__ ufirst, uint64_t id_lo, uint64_t window,
    uint64_t U, uint8_t *__restrict__ kept, const uint32_t *__restrict__ ucounts,
    const uint32_t *__restrict__ parent1, const uint8_t *__restrict__ root_taint, uint32_t bin_shift,
    uint32_t n_bins, uint32_t *__restrict__ cursor /* [n_bins][KB_SUBS], starts at (b * KB_SUBS + s) << bin_shift */,
    uint32_t *__restrict__ lists, unsigned long long *__restrict__ n_kept_total)
{
    constexpr uint32_t TILE = THREADS * KB_KPT, WAVES = THREADS / 64, GROUPS = KB_KPT / 4;
    constexpr uint32_t BPT = KB_MAX_BINS > THREADS ? KB_MAX_BINS / THREADS : 1u;       // bins per thread of the scan
    static_assert(KB_KPT % 4 == 0, "groups of four keys");
    __shared__ uint32_t s_hist[KB_MAX_BINS], s_off[KB_MAX_BINS], s_base[KB_MAX_BINS], s_wave[WAVES];
    __shared__ uint32_t s_stage[TILE];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t v0 = (uint64_t)blockIdx.x * TILE;
    const uint64_t last4 = (U - 1) & ~3ull;      // the last group of four (its tail may lie behind the table: the
                                                 // buffers have the slack, the verdicts are masked)
    for (uint32_t b = tid; b < n_bins; b += THREADS)
        s_hist[b] = 0;
    __syncthreads();
    uint32_t off[KB_KPT], rank[KB_KPT];
    uint32_t total = 0, listed_mask = 0, ask = 0;
    // two halves of GROUPS / 2 groups: the loads of a half are all in flight before its first verdict (all GROUPS at
    // once needed 66 VGPRs -- two registers too many for two 1024-thread workgroups per CU)
    constexpr uint32_t HALF = GROUPS >= 2 ? GROUPS / 2 : 1;
#pragma unroll
  for (uint32_t g0 = 0; g0 < GROUPS; g0 += HALF) {
    ulonglong2 idv[HALF][2];
    uint32_t st4[HALF];
#pragma unroll
    for (uint32_t h = 0; h < HALF; h++) {
        const uint32_t g = g0 + h;
        const uint64_t vb = min(v0 + ((uint64_t)g * THREADS + tid) * 4, last4);
        const ulonglong2 *f2 = reinterpret_cast<const ulonglong2 *>(ufirst + vb);
        idv[h][0] = f2[0];
        idv[h][1] = f2[1];
        st4[h] = METHOD == 3 || METHOD == 1 ? *reinterpret_cast<const uint32_t *>(state + vb) : 0u;
    }
#pragma unroll
    for (uint32_t h = 0; h < HALF; h++) {
        const uint32_t g = g0 + h;
        const uint64_t vg = v0 + ((uint64_t)g * THREADS + tid) * 4, vb = min(vg, last4);
        const uint64_t id[4] = {idv[h][0].x, idv[h][0].y, idv[h][1].x, idv[h][1].y};
        bool k[4];
        if (METHOD == 3 || METHOD == 1) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t st = (st4[h] >> (8 * j)) & (METHOD == 3 ? 0x0Fu : 0xFFu);   // (3: the high nibble is the count)
                k[j] = METHOD == 1 ? st == 1 : st == 0;
                if (METHOD == 3 && (st & 8) && vg == vb && vb + j < U)
                    ask |= 1u << (g * 4 + j);      // (few) a member of a set of count-1 keys asks its root: below
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)       // (unconditional loads at clamped keys)
                k[j] = kept_verdict(METHOD, (uint32_t)min(vb + j, U - 1), labels, best, state, ucounts, parent1, root_taint);
        }
        uint32_t flags = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const bool valid = vg == vb && vb + j < U;     // (a clamped group repeats keys another thread owns)
            k[j] = k[j] && valid;
            flags |= k[j] ? 1u << (8 * j) : 0u;
            total += k[j] ? 1u : 0u;
            if (k[j] && id[j] >= id_lo && id[j] - id_lo < window) {
                off[g * 4 + j] = (uint32_t)(id[j] - id_lo);
                rank[g * 4 + j] = atomicAdd(&s_hist[off[g * 4 + j] >> bin_shift], 1u);
                listed_mask |= 1u << (g * 4 + j);
            }
        }
        if (vg == vb) {
            if (vb + 4 <= U) {
                *reinterpret_cast<uint32_t *>(kept + vb) = flags;
            } else {
                for (uint32_t j = 0; vb + j < U; j++)
                    kept[vb + j] = (uint8_t)(flags >> (8 * j));
            }
        }
    }
  }
    // the keys that have to ask their set's root (directional, closed form: state bit 8), one by one -- their walk's
    // registers do not ride along in the loop above
    while (ask) {
        const uint32_t e = (uint32_t)__ffs((int)ask) - 1u;
        ask &= ask - 1u;
        const uint64_t v = v0 + ((uint64_t)(e >> 2) * THREADS + tid) * 4 + (e & 3u);
        if (kept_verdict(3, (uint32_t)v, labels, best, state, ucounts, parent1, root_taint)) {
            kept[v] = 1;
            total++;
            const uint64_t id = ufirst[v];
            if (id >= id_lo && id - id_lo < window) {
                const uint32_t o = (uint32_t)(id - id_lo), r = atomicAdd(&s_hist[o >> bin_shift], 1u);
#pragma unroll
                for (uint32_t q = 0; q < KB_KPT; q++)      // (compile-time indices: the arrays stay in registers)
                    if (q == e) {
                        off[q] = o;
                        rank[q] = r;
                    }
                listed_mask |= 1u << e;
            }
        }
    }
    __syncthreads();
    // exclusive scan of the bin counts (n_bins <= KB_MAX_BINS: BPT consecutive bins per thread) + one reservation per bin
    uint32_t cnt[BPT], mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < BPT; q++) {
        const uint32_t b = tid * BPT + q;
        cnt[q] = b < n_bins ? s_hist[b] : 0u;
        mine += cnt[q];
    }
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine, listed = 0;
    for (uint32_t wv = 0; wv < WAVES; wv++) {
        run += wv < wave ? s_wave[wv] : 0u;
        listed += s_wave[wv];
    }
    const uint32_t sub = blockIdx.x % KB_SUBS;
    // the cursor reservations of the thread are issued here and consumed after the tile has been staged (which
    // needs the tile-local offsets only): their round trip runs under the staging, not in front of it
    uint32_t gb[BPT], at[BPT];
#pragma unroll
    for (uint32_t q = 0; q < BPT; q++) {
        const uint32_t b = tid * BPT + q;
        at[q] = run;
        gb[q] = 0;
        if (b < n_bins) {
            s_off[b] = run;
            gb[q] = cnt[q] ? atomicAdd(&cursor[b * KB_SUBS + sub], cnt[q]) : 0u;     // (counts from 0: the list's start is added below)
        }
        run += cnt[q];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < KB_KPT; e++)
        if (listed_mask >> e & 1u)
            s_stage[s_off[off[e] >> bin_shift] + rank[e]] = off[e];
#pragma unroll
    for (uint32_t q = 0; q < BPT; q++)
        if (tid * BPT + q < n_bins)
            s_base[tid * BPT + q] = (((tid * BPT + q) * KB_SUBS + sub) << bin_shift) + gb[q] - at[q];
    __syncthreads();
    for (uint32_t p = tid; p < listed; p += THREADS) {
        const uint32_t o = s_stage[p];
        lists[s_base[o >> bin_shift] + p] = o;
    }
    // one atomic per workgroup on the job-wide counter (one per wave: 14 K atomics on one address,
    // ~11 ns each, were most of this kernel's 0.22 ms)
    for (int o = 32; o; o >>= 1)
        total += __shfl_xor(total, o);
    __syncthreads();
    if (lane == 0)
        s_wave[wave] = total;
    __syncthreads();
    if (tid == 0) {
        uint32_t sum = 0;
        for (uint32_t wv = 0; wv < WAVES; wv++)
            sum += s_wave[wv];
        if (sum)
            atomicAdd(n_kept_total, (unsigned long long)sum);
    }
}

// KE_Q workgroups per bin, each owning 1/KE_Q of the bin's id range: each reads the bin's offsets,
// counts those below its range and sets the bits of those inside in its LDS bit map; the map then
// leaves in chunks through an LDS staging area. Measured on the way (50 M-id window, 12 M ids):
// reading the offset lists is the slow part (~1.1 TB/s), so ONE pass over them (KE_Q = 1) beats
// four workgroups per bin (0.20 ms); a 1024-thread workgroup per bin whose threads wrote their own
// ids straight to `out`, load -> LDS-atomic rounds back to back: 0.13 ms; finding output slot j's
// bit by binary search over per-thread prefix counts so that stores coalesce: 0.21 ms.
constexpr uint32_t KE_Q = 1, KE_THREADS = 512;

__global__ __launch_bounds__(KE_THREADS) void kept_emit_kernel(uint32_t KB_SUBS, const uint32_t *__restrict__ cursor,
                                                               const uint32_t *__restrict__ lists,
                                                               uint32_t bin_shift, uint32_t n_bins, uint64_t id_base,
                                                               uint64_t *__restrict__ out,
                                                               uint32_t *__restrict__ n_listed)
{
    extern __shared__ uint32_t s_bits[];          // (1 << bin_shift) / 32 / KE_Q words
    __shared__ uint32_t s_part[3 * KE_THREADS / 64], s_scan[KE_THREADS / 64];
    __shared__ uint16_t s_ids[KE_THREADS * 32];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b = blockIdx.x / KE_Q, q = blockIdx.x % KE_Q;
    const uint32_t words = (1u << (bin_shift - 5)) / KE_Q;       // a multiple of KE_THREADS: bin_shift >= 15
    for (uint32_t w = tid; w < words; w += KE_THREADS)
        s_bits[w] = 0;
    // ids of the bins before this one and of all bins (n_bins <= 512: two per thread)
    uint32_t before = 0, all = 0;
    for (uint32_t x = tid; x < n_bins; x += KE_THREADS) {
        uint32_t cnt = 0;
        for (uint32_t k = 0; k < KB_SUBS; k++)
            cnt += cursor[x * KB_SUBS + k];
        all += cnt;
        before += x < b ? cnt : 0u;
    }
    __syncthreads();                                             // the bit map is clear
    const uint32_t first = b << bin_shift;                       // id offset of the bin's first id
    const uint32_t q_lo = q * words * 32u, q_hi = q_lo + words * 32u;
    uint32_t below = 0;
    for (uint32_t k = 0; k < KB_SUBS; k++) {
        const uint32_t slab = (b * KB_SUBS + k) << bin_shift, n = cursor[b * KB_SUBS + k];
        auto mark = [&](uint32_t off) {
            const uint32_t o = off - first;
            below += o < q_lo ? 1u : 0u;
            if (o >= q_lo && o < q_hi)
                atomicOr(&s_bits[(o - q_lo) >> 5], 1u << (o & 31u));
        };
        // 4 x 16 bytes per thread in flight (a bin's list is read by ONE workgroup: the loads in
        // flight are what bounds it; 8 x 4 bytes per thread took 0.10 ms for 49 MB of lists)
        const uint4 *list4 = reinterpret_cast<const uint4 *>(lists + slab);     // slab is 2^bin_shift entries: aligned
        const uint32_t n4 = n / 4;
        for (uint32_t i0 = 0; i0 < n4; i0 += 4 * KE_THREADS) {
            uint4 o[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * KE_THREADS + tid;
                o[u] = i < n4 ? list4[i] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                if (i0 + u * KE_THREADS + tid < n4) {
                    mark(o[u].x);
                    mark(o[u].y);
                    mark(o[u].z);
                    mark(o[u].w);
                }
        }
        if (tid < n - n4 * 4)
            mark(lists[slab + n4 * 4 + tid]);
    }
    for (int o = 32; o; o >>= 1) {
        before += __shfl_xor(before, o);
        all += __shfl_xor(all, o);
        below += __shfl_xor(below, o);
    }
    if (lane == 0) {
        s_part[wave] = before;
        s_part[KE_THREADS / 64 + wave] = all;
        s_part[2 * (KE_THREADS / 64) + wave] = below;
    }
    __syncthreads();                                             // ... and the bit map is complete
    uint32_t pos = 0, everything = 0;
    for (uint32_t w = 0; w < KE_THREADS / 64; w++) {
        pos += s_part[w] + s_part[2 * (KE_THREADS / 64) + w];
        everything += s_part[KE_THREADS / 64 + w];
    }
    if (blockIdx.x == 0 && tid == 0)
        *n_listed = everything;
    // The map leaves in chunks of KE_THREADS words (8192 ids at most): thread t ranks the bits of
    // word t of the chunk, parks their offsets in LDS in rank order, and the chunk's ids go out as
    // one contiguous, coalesced run. (Every thread writing its own ids straight to `out` is 64
    // different cache lines per store instruction: 12 M partial-line writes, as bad as the byte map.)
    uint64_t at = pos;
    for (uint32_t c0 = 0; c0 < words; c0 += KE_THREADS) {
        uint32_t m = s_bits[c0 + tid];
        const uint32_t mine = __popc(m);
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += up;
        }
        __syncthreads();                  // s_scan and s_ids of the chunk before are done with
        if (lane == 63)
            s_scan[wave] = incl;
        __syncthreads();
        uint32_t r = incl - mine, chunk_total = 0;
        for (uint32_t w = 0; w < KE_THREADS / 64; w++) {
            r += w < wave ? s_scan[w] : 0u;
            chunk_total += s_scan[w];
        }
        while (m) {
            s_ids[r++] = (uint16_t)(((tid) << 5) + (__ffs((int)m) - 1));      // offset inside the chunk
            m &= m - 1;
        }
        __syncthreads();
        const uint64_t chunk_base = id_base + first + q_lo + ((uint64_t)c0 << 5);
        for (uint32_t j = tid; j < chunk_total; j += KE_THREADS)
            out[at + j] = chunk_base + s_ids[j];
        at += chunk_total;
    }
}

// (the cursors count from 0: list (b, s) starts at (b * subs + s) << bin_shift)
__global__ void kept_bin_starts_kernel(uint32_t n_lists, uint32_t bin_shift, uint32_t *__restrict__ cursor)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_lists)
        cursor[b] = 0u;
}

// ---- ascending id list = compaction of the window's byte map (bytes are 0 or 1) -------------
// A block covers WIN_BLOCK flags, 16 per thread (one uint4): counts per block, a scan of the
// block counts, then every thread writes the ids of its set bytes behind its block's offset.
constexpr uint32_t WIN_BLOCK = 256 * 16;

__device__ __forceinline__ uint4 window_load16(const uint8_t *__restrict__ flags, uint64_t at, uint64_t n)
{
    if (at + 16 <= n)
        return *reinterpret_cast<const uint4 *>(flags + at);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; j < 16 && at + j < n; j++)
        w[j >> 2] |= (uint32_t)flags[at + j] << (8 * (j & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void window_count_kernel(const uint8_t *__restrict__ flags, uint64_t n,
                                                           uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t wave_sum[4];
    const uint64_t at = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t c = 0;
    if (at < n) {
        const uint4 v = window_load16(flags, at, n);
        c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (int o = 32; o; o >>= 1)
        c += __shfl_xor(c, o);
    if (fqd_lane() == 0)
        wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        block_counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

__global__ __launch_bounds__(256) void window_emit_kernel(const uint8_t *__restrict__ flags, uint64_t n,
                                                          const uint32_t *__restrict__ block_incl, uint64_t id_base,
                                                          uint64_t *__restrict__ out)
{
    __shared__ uint32_t wave_sum[4];
    const uint64_t at = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at < n)
        v = window_load16(flags, at, n);
    const uint32_t mine = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    uint32_t incl = mine;                       // inclusive scan over the wave
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (fqd_lane() >= (uint32_t)o)
            incl += up;
    }
    if (fqd_lane() == 63)
        wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t pos = blockIdx.x ? block_incl[blockIdx.x - 1] : 0u;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
        pos += wave_sum[w];
    pos += incl - mine;
    const uint32_t word[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        uint32_t m = word[j];
        while (m) {
            const uint32_t b = __ffs((int)m) - 1;          // bit 8k of the word = byte k
            out[pos++] = id_base + at + j * 4 + (b >> 3);
            m &= m - 1;
        }
    }
}

__global__ void gather_kept_kernel(const uint32_t *__restrict__ kept_u32, const uint32_t *__restrict__ kept_scan,
                                   const uint64_t *__restrict__ ufirst, uint64_t U, uint64_t *out)
{
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < U && kept_u32[v])
        out[kept_scan[v] - 1] = ufirst[v];
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

namespace fqd {

hipError_t launch_uf_init(uint32_t *parent, uint64_t U, hipStream_t st)
{
    if (U)
        uf_init_kernel<<<grid_for(U), 256, 0, st>>>(parent, U);
    return hipGetLastError();
}

hipError_t launch_uf_union(uint32_t *parent, const uint32_t *edges, uint64_t E, unsigned long long *n_hooks,
                           hipStream_t st, bool sampled_first)
{
    if (E && !sampled_first)
        uf_union_kernel<<<grid_for(E), 256, 0, st>>>(parent, edges, E, n_hooks, 0u);
    if (E && sampled_first) {
        uf_union_kernel<<<grid_for(E), 256, 0, st>>>(parent, edges, E, n_hooks, 1u);
        uf_union_kernel<<<grid_for(E), 256, 0, st>>>(parent, edges, E, n_hooks, 2u);
    }
    return hipGetLastError();
}

hipError_t launch_union_directional(uint32_t *node, const uint32_t *edges, uint64_t E, const uint32_t *ucounts, uint32_t *list11,
                                    unsigned long long *list11_count, unsigned long long *n_hooks, hipStream_t st,
                                    bool sampled_first)
{
    if (!E)
        return hipSuccess;
    const unsigned grid = (unsigned)((E + 256 * UD_EPT - 1) / (256 * UD_EPT));
    if (!sampled_first) {
        union_directional_kernel<<<grid, 256, 0, st>>>(node, edges, E, ucounts, list11, list11_count, n_hooks, 0u);
    } else {
        union_directional_kernel<<<grid, 256, 0, st>>>(node, edges, E, ucounts, list11, list11_count, n_hooks, 1u);
        union_directional_kernel<<<grid, 256, 0, st>>>(node, edges, E, ucounts, list11, list11_count, n_hooks, 2u);
    }
    return hipGetLastError();
}

hipError_t launch_unzip_nodes(const uint32_t *node, uint32_t *parent, uint8_t *state, uint64_t U, hipStream_t st)
{
    if (U)
        unzip_nodes_kernel<<<(unsigned)std::max<uint64_t>(1, ((U + 3) / 4 + 255) / 256), 256, 0, st>>>(node, parent, state, U);
    return hipGetLastError();
}

hipError_t launch_graph_preinit_nodes(uint32_t *node, uint32_t *best, uint8_t *root_taint, const uint32_t *ucounts, uint64_t U,
                                      unsigned long long *hook_slots, uint32_t hook_words, hipStream_t st, uint32_t *zero32,
                                      uint32_t zero32_words, unsigned long long *zero64_a, unsigned long long *zero64_b,
                                      unsigned long long *zero64_c)
{
    const uint64_t quads = (U + 3) / 4;
    graph_preinit_nodes_kernel<<<(unsigned)std::max<uint64_t>(1, (quads + 255) / 256), 256, 0, st>>>(
        node, best, root_taint, ucounts, U, hook_slots, hook_words, zero32, zero32_words, zero64_a, zero64_b, zero64_c);
    return hipGetLastError();
}

hipError_t launch_mask_dead_edges(uint32_t *edges, uint64_t E, const uint8_t *alive, hipStream_t st)
{
    if (E)
        mask_dead_edges_kernel<<<grid_for(E), 256, 0, st>>>(edges, E, alive);
    return hipGetLastError();
}

hipError_t launch_hook_total(const unsigned long long *slots, uint64_t n_nodes, unsigned long long *n_components,
                             hipStream_t st, unsigned long long *n_second_walks)
{
    hook_total_kernel<<<1, 64, 0, st>>>(slots, n_nodes, n_components, n_second_walks);
    return hipGetLastError();
}

hipError_t launch_edge_roots(uint32_t *parent, const uint32_t *edges, uint64_t E, uint32_t *roots, hipStream_t st)
{
    if (E)
        edge_roots_kernel<<<grid_for(E), 256, 0, st>>>(parent, edges, E, roots);
    return hipGetLastError();
}

hipError_t launch_subgraph_mark(const uint32_t *uv, const uint32_t *roots, uint64_t E, uint32_t n_parts, uint32_t part,
                                uint32_t *flags, uint32_t *sub, unsigned long long *n_sub, hipStream_t st)
{
    if (E)
        subgraph_mark_kernel<<<(unsigned)((E + 1023) / 1024), 1024, 0, st>>>(uv, roots, E, n_parts, part, flags, sub, n_sub);
    return hipGetLastError();
}

hipError_t launch_subgraph_mark_home(const uint32_t *uv, const uint32_t *roots, uint64_t E, uint32_t n_parts, uint32_t part,
                                     UidBounds bounds, uint8_t *span, uint32_t *flags, uint32_t *sub,
                                     unsigned long long *n_sub, uint32_t *home, unsigned long long *n_home,
                                     unsigned long long *n_span, hipStream_t st)
{
    if (E) {
        span_mark_kernel<<<grid_for(E), 256, 0, st>>>(uv, roots, E, bounds, span);
        subgraph_mark_home_kernel<<<(unsigned)((E + 1023) / 1024), 1024, 0, st>>>(uv, roots, E, n_parts, part, bounds, span,
                                                                                  flags, sub, n_sub, home, n_home, n_span);
    }
    return hipGetLastError();
}

hipError_t launch_mark_dropped_after(uint8_t *state, uint32_t *best, uint64_t U, const uint32_t *dropped, uint64_t n,
                                     uint32_t *bad, hipStream_t st)
{
    if (n)
        mark_dropped_after_kernel<<<grid_for(n), 256, 0, st>>>(state, best, U, dropped, n, bad);
    return hipGetLastError();
}

hipError_t launch_subgraph_finish(const uint32_t *flags, const uint32_t *flags_incl, uint64_t n_nodes, uint64_t E,
                                  uint32_t *touched, uint32_t *sub, const unsigned long long *n_sub, hipStream_t st)
{
    if (n_nodes)
        subgraph_nodes_kernel<<<(unsigned)((n_nodes + 255) / 256), 256, 0, st>>>(flags, flags_incl, n_nodes, touched);
    if (E)
        subgraph_renumber_kernel<<<(unsigned)((2 * E + 255) / 256), 256, 0, st>>>(sub, n_sub, flags_incl);
    return hipGetLastError();
}

hipError_t launch_check_indices(const uint32_t *idx, uint64_t n, uint64_t limit, uint32_t *bad, hipStream_t st)
{
    if (n)
        check_indices_kernel<<<grid_for(n), 256, 0, st>>>(idx, n, limit, bad);
    return hipGetLastError();
}

hipError_t launch_mark_dropped(uint8_t *state, uint64_t U, const uint32_t *dropped, uint64_t n, uint32_t *bad,
                               hipStream_t st)
{
    if (n)
        mark_dropped_kernel<<<grid_for(n), 256, 0, st>>>(state, U, dropped, n, bad);
    return hipGetLastError();
}

hipError_t launch_uf_flatten(uint32_t *parent, uint64_t U, unsigned long long *n_roots, hipStream_t st)
{
    if (U) {
        unsigned g = grid_for(U);
        if (g > 2048)
            g = 2048;
        uf_flatten_kernel<<<g, 256, 0, st>>>(parent, U, n_roots);
    }
    return hipGetLastError();
}

hipError_t launch_graph_preinit(uint32_t *parent, uint32_t *best, uint8_t *state, uint8_t *root_taint,
                                uint64_t U, unsigned long long *hook_slots, uint32_t hook_words, hipStream_t st,
                                uint32_t *zero32, uint32_t zero32_words, unsigned long long *zero64_a,
                                unsigned long long *zero64_b, const uint32_t *ucounts, unsigned long long *zero64_c)
{
    const uint64_t quads = (U + 3) / 4;
    graph_preinit_kernel<<<(unsigned)std::max<uint64_t>(1, (quads + 255) / 256), 256, 0, st>>>(
        parent, best, state, root_taint, U, hook_slots, hook_words, zero32, zero32_words, zero64_a, zero64_b,
        zero64_c, ucounts);
    return hipGetLastError();
}

hipError_t launch_dstate_init(uint8_t *state, const uint32_t *ucounts, uint64_t U, hipStream_t st)
{
    if (U)
        dstate_init_kernel<<<grid_for(U), 256, 0, st>>>(state, ucounts, U);
    return hipGetLastError();
}

hipError_t launch_dissect_init(uint32_t *best, uint8_t *state, uint64_t U, hipStream_t st)
{
    if (U)
        dissect_init_kernel<<<grid_for(U), 256, 0, st>>>(best, state, U);
    return hipGetLastError();
}

hipError_t launch_highest_count(const uint32_t *labels, const uint32_t *ucounts, const uint32_t *urecs,
                                const uint32_t *ulens, KeyShape sh, uint64_t U, uint32_t *best, hipStream_t st)
{
    if (U)
        highest_count_kernel<<<grid_for(U), 256, 0, st>>>(labels, ucounts, urecs, ulens, sh, U, best);
    return hipGetLastError();
}

hipError_t launch_directional_round(const uint32_t *edges, uint64_t E, const uint32_t *ucounts,
                                    const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t *best,
                                    uint32_t *stamp, uint32_t round, uint32_t *changed, hipStream_t st)
{
    if (E)
        directional_round_kernel<<<grid_for(E), 256, 0, st>>>(edges, E, ucounts, urecs, ulens, sh, best, stamp, round,
                                                              changed);
    return hipGetLastError();
}

hipError_t launch_directional_closed(const uint32_t *edges, uint64_t E, const uint32_t *ucounts, const uint32_t *urecs,
                                     const uint32_t *ulens, KeyShape sh, const uint32_t *parent, uint8_t *state,
                                     uint32_t *list11, unsigned long long *list11_count, uint8_t *root_taint,
                                     uint32_t *best, int pass, hipStream_t st, uint32_t *roots, uint32_t *list2,
                                     unsigned long long *list2_count)
{
    if (!E)
        return hipSuccess;
    const unsigned list_grid = (unsigned)std::min<uint64_t>(grid_for(E), 8192);   // (the list's length is on the device)
    // parent: the COMPONENTS' union-find (uf_union_kernel), complete before pass 2 -- pass 1 does not look at it.
    // list2 != NULL: pass 1b keeps the listed edges pass 2 must look at (worth its launch where the edges between
    // count-1 keys are many: d >= 2, error variants of one molecule next to each other); else pass 2 walks the whole list
    if (pass == 1) {
        const unsigned grid = (unsigned)((E + 256 * DE_EPT - 1) / (256 * DE_EPT));
        directional_edges_kernel<<<grid, 256, 0, st>>>(edges, E, ucounts, state, list11, list11_count);
    } else {
        if (list2)
            directional_filter_kernel<<<(unsigned)std::min<uint64_t>(grid_for((E + 3) / 4), 8192), 256, 0, st>>>(
                edges, list11, list11_count, state, list2, list2_count);
        const uint32_t *list = list2 ? list2 : list11;
        const unsigned long long *count = list2 ? list2_count : list11_count;
        directional_roots_kernel<<<list_grid, 256, 0, st>>>(edges, list, count, parent, state, root_taint, roots);
        directional_best_kernel<<<list_grid, 256, 0, st>>>(edges, list, count, ucounts, urecs, ulens, sh, roots, root_taint,
                                                           best);
    }
    return hipGetLastError();
}

hipError_t launch_orient_edges(uint32_t *edges, uint64_t E, const uint32_t *ucounts, const uint32_t *urecs,
                               const uint32_t *ulens, KeyShape sh, hipStream_t st)
{
    if (E)
        orient_edges_kernel<<<grid_for(E), 256, 0, st>>>(edges, E, ucounts, urecs, ulens, sh);
    return hipGetLastError();
}

hipError_t launch_adjacency_live_edges(const uint32_t *edges, uint64_t E, const uint8_t *state, uint32_t *live,
                                       unsigned long long *live_count, hipStream_t st)
{
    if (E)
        adjacency_live_edges_kernel<<<(unsigned)((E + 1023) / 1024), 256, 0, st>>>(edges, E, state, live, live_count);
    return hipGetLastError();
}

hipError_t launch_adjacency_round(const uint32_t *edges, uint64_t E, uint64_t U, uint8_t *state, uint32_t *blocked,
                                  uint32_t round, uint32_t *changed, hipStream_t st)
{
    if (E)
        adjacency_edges_kernel<<<grid_for(E), 256, 0, st>>>(edges, E, state, blocked, round, changed);
    if (U)
        adjacency_nodes_kernel<<<grid_for(U), 256, 0, st>>>(U, state, blocked, round, changed);
    return hipGetLastError();
}

hipError_t launch_kept_flags(int method, const uint32_t *labels, const uint32_t *best, const uint8_t *state,
                             const uint64_t *ufirst, uint64_t id_lo, uint64_t id_hi, uint64_t U, uint8_t *kept,
                             uint32_t *kept_u32, uint8_t *window_flags, uint64_t window_size,
                             const uint32_t *ucounts, const uint32_t *parent1, const uint8_t *root_taint,
                             unsigned long long *n_kept_total, hipStream_t st)
{
    if (U) {
        unsigned g = grid_for(U);
        if (g > 2048)
            g = 2048;
        kept_flags_kernel<<<g, 256, 0, st>>>(method, labels, best, state, ufirst, id_lo, id_hi, U, kept, kept_u32,
                                             window_flags, window_size, ucounts, parent1, root_taint, n_kept_total);
    }
    return hipGetLastError();
}

// bins of 2^shift ids, shift >= 15 (the emit kernel's 1024 threads own >= one word each), <= 512 bins
uint32_t kept_bin_lists() { return kb_subs(); }

uint32_t kept_bin_shift(uint64_t window)
{
    uint32_t s = 15;
    while (((window + (1ull << s) - 1) >> s) > KB_MAX_BINS)
        s++;
    return s;
}

hipError_t launch_kept_bins(int method, const uint32_t *labels, const uint32_t *best, const uint8_t *state,
                            const uint64_t *ufirst, uint64_t id_lo, uint64_t window, uint64_t U, uint8_t *kept,
                            const uint32_t *ucounts, const uint32_t *parent1, const uint8_t *root_taint,
                            uint32_t *cursor, uint32_t *lists, unsigned long long *n_kept_total, uint64_t id_base,
                            uint64_t *out, uint32_t *n_listed, hipStream_t st, bool cursors_zeroed)
{
    const uint32_t shift = kept_bin_shift(window);
    const uint32_t n_bins = (uint32_t)((window + (1ull << shift) - 1) >> shift);
    const uint32_t KB_SUBS = kb_subs();
    if (!U || !n_bins || shift > 18 || ((uint64_t)n_bins * KB_SUBS << shift) > 0xFFFFFFFFull)   // (a 2^18-bit map is 32 KB of LDS)
        return hipErrorInvalidValue;
    if (!cursors_zeroed)       // (else: graph_preinit_kernel has cleared the whole cursor table)
        kept_bin_starts_kernel<<<(n_bins * KB_SUBS + 255) / 256, 256, 0, st>>>(n_bins * KB_SUBS, shift, cursor);
    // 1024-thread workgroups (16 Ki keys per tile) from 2^22 keys on: runs per (tile, bin) four times as long, a
    // quarter of the cursor atomics; small tables keep 256 threads (more workgroups than CUs)
    static const int kb_threads_env = getenv("FQD_KB_THREADS") ? atoi(getenv("FQD_KB_THREADS")) : 0;
    const uint32_t threads = kb_threads_env == 256 || kb_threads_env == 512 || kb_threads_env == 1024
                                 ? (uint32_t)kb_threads_env : (U >= (1ull << 22) ? 1024u : 256u);
    const unsigned grid = (unsigned)((U + (uint64_t)threads * KB_KPT - 1) / ((uint64_t)threads * KB_KPT));
#define FQD_KB_LAUNCH_T(M, T)                                                                                            \
    kept_bin_kernel<M, T><<<grid, T, 0, st>>>(KB_SUBS, labels, best, state, ufirst, id_lo, window, U, kept, ucounts,      \
                                              parent1, root_taint, shift, n_bins, cursor, lists, n_kept_total)
#define FQD_KB_LAUNCH(M)                                                                                                 \
    do {                                                                                                                 \
        if (threads == 1024)                                                                                             \
            FQD_KB_LAUNCH_T(M, 1024);                                                                                    \
        else if (threads == 512)                                                                                         \
            FQD_KB_LAUNCH_T(M, 512);                                                                                     \
        else                                                                                                             \
            FQD_KB_LAUNCH_T(M, 256);                                                                                     \
    } while (0)
    switch (method) {
    case 0: FQD_KB_LAUNCH(0); break;
    case 2: FQD_KB_LAUNCH(2); break;
    case 3: FQD_KB_LAUNCH(3); break;
    default: FQD_KB_LAUNCH(1); break;      // adjacency: state[v] == 1
    }
#undef FQD_KB_LAUNCH
#undef FQD_KB_LAUNCH_T
    kept_emit_kernel<<<n_bins * KE_Q, KE_THREADS, (1u << (shift - 5)) * 4 / KE_Q, st>>>(KB_SUBS, cursor, lists, shift, n_bins,
                                                                                       id_base, out, n_listed);
    return hipGetLastError();
}

uint32_t window_blocks(uint64_t n) { return (uint32_t)((n + WIN_BLOCK - 1) / WIN_BLOCK); }

hipError_t launch_window_count(const uint8_t *flags, uint64_t n, uint32_t *block_counts, hipStream_t st)
{
    if (n)
        window_count_kernel<<<window_blocks(n), 256, 0, st>>>(flags, n, block_counts);
    return hipGetLastError();
}

hipError_t launch_window_emit(const uint8_t *flags, uint64_t n, const uint32_t *block_incl, uint64_t id_base,
                              uint64_t *out, hipStream_t st)
{
    if (n)
        window_emit_kernel<<<window_blocks(n), 256, 0, st>>>(flags, n, block_incl, id_base, out);
    return hipGetLastError();
}

hipError_t launch_gather_kept(const uint32_t *kept_u32, const uint32_t *kept_scan, const uint64_t *ufirst, uint64_t U,
                              uint64_t *out, hipStream_t st)
{
    if (U)
        gather_kept_kernel<<<grid_for(U), 256, 0, st>>>(kept_u32, kept_scan, ufirst, U, out);
    return hipGetLastError();
}

}  // namespace fqd
