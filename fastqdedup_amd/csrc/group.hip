// group.hip -- stage 3 without a device-wide sort: the (segment hash, uid) pairs of one search
// pass are PARTITIONED into ~2^B buckets by the top B hash bits (two levels, LDS-aggregated,
// the scheme of collapse_lds.hip applied to 8-byte items); one wave per group of buckets then
// compares the hashes of each bucket's ~50-100 keys all against all out of LDS and lists the pairs
// whose full 32-bit segment hashes agree; a second kernel verifies those candidates, one thread
// per pair (record fetches, mismatch count, first-agreeing-segment rule). Same contract as
// bucket_pairs_kernel (edges.hip; reference _triemodule.c:807-895, the neighbour search of
// Trie.pop_cluster): every pair within max_distance is reported in the pass of the FIRST
// segment it agrees on.
//
// Against the radix sort it replaces (3 onesweep passes + histogram, 0.47 ms for 14 M pairs):
// two histogram passes and two scatter passes, every store part of a contiguous run (0.2 ms).
// What the sort gave for free -- equal hashes adjacent -- costs lambda/2 LDS hash compares per
// key here (lambda = keys per bucket), cheap next to a fourth pass over HBM.
#include <cstdlib>
#include "fqd_internal.h"
#include "partition.cuh"

// staging rounds of the pair partition's 4096-item tiles (partition.cuh ROUNDS): 2 halves the staging area (52 -> 33 KB
// of LDS, four workgroups per CU instead of three): 0.273 instead of 0.290 ms for both levels at config 3; 4: 0.271
#ifndef FQD_PAIR_ROUNDS
#define FQD_PAIR_ROUNDS 2
#endif
namespace {

constexpr uint32_t GP_THREADS = 256;
constexpr uint32_t GP_SLICE = 512;      // hashes of one bucket held in LDS per wave
constexpr uint32_t GP_ECAP = 2048;      // edges buffered per block of the verify kernel (>= 4 x 256 hits of a sweep + ECAP / 2 left over)

// What a (segment hash, uid) item looks like to the partition (partition.cuh): LEVEL 1 reads the
// hash array (uid = position); level 2 reads level-1's items. The key is the hash.
struct PairPolicy {
    using Item = uint2;
    static constexpr uint32_t EPT = 16;         // 4096-pair tiles
    static constexpr uint32_t ROUNDS = FQD_PAIR_ROUNDS;
    static constexpr bool MAY_SKIP = false;
    static constexpr bool CAN_SPILL = false;
    static __device__ __forceinline__ bool skip(const uint2 &) { return false; }
    struct Source {
        const uint32_t *hashes;   // level 1
        const uint2 *in;          // level 2
        const uint32_t *values;   // level 1: what travels with hash i (NULL: its position i)
    };
    using Raw = uint2;
    template <bool LEVEL1>
    static __device__ __forceinline__ uint2 fetch(const Source &s, uint32_t i)
    {
        return LEVEL1 ? make_uint2(s.hashes[i], s.values ? s.values[i] : i) : s.in[i];
    }
    struct Shared {};
    static __device__ __forceinline__ void init_shared(Shared &, uint32_t) {}
    static __device__ __forceinline__ void flush(const Source &, Shared &, uint32_t) {}
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t finish(const Source &, uint32_t, const uint2 &raw, uint2 &v, bool, uint32_t,
                                                      Shared &)
    {
        v = raw;
        return v.x;
    }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t load(const Source &s, uint32_t i, uint2 &v)
    {
        v = fetch<LEVEL1>(s, i);
        return v.x;
    }
    using KeyRaw = uint32_t;
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t key_fetch(const Source &s, uint32_t i)
    {
        return LEVEL1 ? s.hashes[i] : s.in[i].x;
    }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t key_finish(const Source &, uint32_t, uint32_t raw) { return raw; }
    static __device__ __forceinline__ uint32_t segment_tag(const Source &, uint32_t) { return 0u; }
    static __device__ __forceinline__ void apply_tag(uint2 &, uint32_t) {}
};

template <bool LEVEL1>
__global__ __launch_bounds__(fqd_partition::THREADS) void gp_hist_kernel(PairPolicy::Source src,
                                                                         const uint32_t *__restrict__ seg_start,
                                                                         const uint32_t *__restrict__ tile_start,
                                                                         uint32_t n_seg, uint32_t shift, uint32_t n_bins,
                                                                         uint32_t *__restrict__ hist)
{
    fqd_partition::hist_body<PairPolicy, LEVEL1>(src, seg_start, tile_start, n_seg, shift, n_bins, hist);
}

template <bool LEVEL1>
__global__ __launch_bounds__(fqd_partition::THREADS) void gp_scatter_kernel(PairPolicy::Source src,
                                                                            const uint32_t *__restrict__ seg_start,
                                                                            const uint32_t *__restrict__ tile_start,
                                                                            uint32_t n_seg, uint32_t shift,
                                                                            uint32_t n_bins, uint32_t *__restrict__ cursor,
                                                                            uint2 *__restrict__ out, uint32_t slab_cap,
                                                                            uint32_t *__restrict__ slab_overflow,
                                                                            const uint32_t *__restrict__ seg_end,
                                                                            uint32_t seg_mask, uint32_t l1_subs)
{
    fqd_partition::scatter_body<PairPolicy, LEVEL1>(src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out,
                                                    slab_cap, slab_overflow, seg_end, 0, seg_mask, l1_subs);
}

// tiles of the slab segments level 1 left (level 1 in slab mode, partition.cuh l1_subs)
__global__ __launch_bounds__(1024) void gp_slab_tile_starts_kernel(const uint32_t *__restrict__ seg_start,
                                                                   const uint32_t *__restrict__ seg_end, uint32_t n_seg,
                                                                   uint32_t *__restrict__ tile_start)
{
    fqd_partition::slab_tile_starts_body<fqd_partition::THREADS * PairPolicy::EPT>(seg_start, seg_end, n_seg, tile_start);
}

// start of a search pass: the level-1 segment / tile bounds and the zeroed candidate counters, in one
// small launch (two host-to-device copies and a fill, each a stop on the stream, did this before)
// `more` (a search inside fqd_find_edges): also the search's job counters (edges, candidate need, slab flag), its
// statistics slots, and the slab starts of both partition levels -- four fills and a launch less on the stream
__global__ void gp_pass_init_kernel(uint32_t *__restrict__ seg1, uint32_t *__restrict__ tiles1, uint32_t n_items,
                                    uint32_t n_tiles, unsigned long long *__restrict__ cand_ctr, uint32_t ctr_words,
                                    fqd::PassInitMore more)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        seg1[0] = 0;
        seg1[1] = n_items;
        tiles1[0] = 0;
        tiles1[1] = n_tiles;
        if (more.ctr64) {
            more.ctr64[more.zero_a] = 0ull;
            more.ctr64[more.zero_b] = 0ull;
            more.ctr64[more.zero_c] = 0ull;
        }
    }
    if (t < ctr_words)
        cand_ctr[t] = 0;
    if (t < more.stat_words)
        more.stats[t] = 0u;
    if (more.start1 && t <= more.n1) {
        more.start1[t] = t * more.cap1;
        if (t < more.n1)
            more.cursor1[t] = t * more.cap1;
    }
    if (more.start2 && t <= more.n2) {
        more.start2[t] = t * more.cap2;
        if (t < more.n2)
            more.cursor2[t] = t * more.cap2;
    }
}

// slab mode of level 2 (as in collapse_lds.hip): bucket b owns slots [b * cap, (b + 1) * cap)
__global__ void gp_slab_starts_kernel(uint32_t n_buckets, uint32_t cap, uint32_t *__restrict__ bucket_start,
                                      uint32_t *__restrict__ cursor)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets)
        return;
    bucket_start[b] = b * cap;
    if (b < n_buckets)
        cursor[b] = b * cap;
}

__global__ void gp_tile_starts_kernel(const uint32_t *__restrict__ seg_start, uint32_t n_seg,
                                      uint32_t *__restrict__ tile_start)
{
    fqd_partition::tile_starts_body<fqd_partition::THREADS * PairPolicy::EPT>(seg_start, n_seg, tile_start);
}

__global__ void gp_matrix_starts_kernel(const uint32_t *__restrict__ matrix_incl, uint32_t n_bins, uint32_t n_tiles,
                                        uint32_t *__restrict__ start)
{
    fqd_partition::matrix_starts_body(matrix_incl, n_bins, n_tiles, start);
}

__global__ void gp_bucket_starts_kernel(const uint32_t *__restrict__ hist_incl, uint32_t n_buckets,
                                        uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ cursor)
{
    fqd_partition::bucket_starts_body(hist_incl, n_buckets, bucket_start, cursor);
}

template <int K>
__device__ __forceinline__ bool gp_verify(const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens,
                                          const KeyShape &sh, uint32_t d, uint32_t seg, uint32_t nseg, uint32_t uid,
                                          uint32_t uj, uint32_t dw_first /* OR over the planes of word 0 of a ^ b */)
{
    const uint32_t len = fqd_key_len(sh, ulens, uid);
    if (fqd_key_len(sh, ulens, uj) != len)
        return false;
    const uint32_t W = sh.words;
    const uint32_t *my_rec = urecs + (uint64_t)uid * sh.stride, *other = urecs + (uint64_t)uj * sh.stride;
    // The first 32-base word alone -- a candidate that is no neighbour at all (a shared segment,
    // nothing else in common) is out after ONE sector per record -- then eight words per step, their
    // loads requested together: a true neighbour is read to the end, and a word-by-word loop with an
    // exit test pays a round trip per word (10 for a 300-nt key). Measured per launch, word by word /
    // eight at once from the start / this: config 2 0.092 / 0.136 / 0.092 ms, config 5 2.87 / 1.23 / 1.46 ms.
    uint32_t dist = W ? __popc(dw_first) : 0u;
    for (uint32_t w0 = 1; w0 < W && dist <= d; w0 += 8) {
        uint32_t dw[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            dw[j] = 0;
            if (w0 + j < W) {
#pragma unroll
                for (int k = 0; k < K; k++)
                    dw[j] |= my_rec[(w0 + j) * K + k] ^ other[(w0 + j) * K + k];
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            dist += __popc(dw[j]);
    }
    if (dist > d)
        return false;
    // emit only in the pass of the first truly agreeing segment
    for (uint32_t s2 = 0; s2 < seg; s2++) {
        uint32_t slo, shi;
        fqd_segment(len, s2, nseg, slo, shi);
        bool agree = true;
        if (shi > slo) {
            for (uint32_t w = slo >> 5; w <= ((shi - 1) >> 5) && agree; w++) {
                uint32_t dw = dw_first;
                if (w) {
                    dw = 0;
#pragma unroll
                    for (int k = 0; k < K; k++)
                        dw |= my_rec[w * K + k] ^ other[w * K + k];
                }
                if (dw & fqd_range_mask(w, slo, shi))
                    agree = false;
            }
        }
        if (agree)
            return false;
    }
    return true;
}

// Phase 1 -- one wave per bucket (~200 keys). The wave counting-sorts its bucket by the next 6
// hash bits inside LDS (64 sub-bins of ~3 keys: LDS atomics for the ranks, one shuffle scan), so a
// key is compared only with the few keys after it in its sub-bin -- one or two 128-bit LDS reads
// instead of ~45 compares per key against the whole bucket (that loop was VALU/LDS-throughput
// bound: 0.10 ms of the kernel's 0.15 ms for 14 M keys). Pairs whose 32-bit hashes agree are
// CANDIDATES, appended (uid, uid) to a device list through a per-wave LDS buffer that the wave
// flushes by itself. No record is touched here: verifying in place cost one HBM latency chain per
// pair (17 us per bucket). Waves share nothing, so there is no workgroup barrier at all.
constexpr uint32_t GP_WCAP = 128;   // candidates buffered per wave
constexpr uint32_t GP_LISTS = 64;   // candidate lists (one counter each, a cache line apart): flush atomics spread out

__global__ __launch_bounds__(GP_THREADS) void grouped_candidates_kernel(
    const uint2 *__restrict__ items, const uint32_t *__restrict__ bucket_start,
    const uint32_t *__restrict__ bucket_end /* NULL, or slab mode: where each bucket's cursor stopped */,
    uint32_t n_buckets, uint32_t sub_shift, uint2 *__restrict__ cands_all, unsigned long long *__restrict__ cand_counts, uint64_t list_cap,
    const uint8_t *__restrict__ skip /* NULL, or skip[b] != 0: bucket b is CROWDED -- its keys are matched on finer
                                      * segments instead (gp_refine_items_kernel) */,
    uint32_t give_up_over, unsigned long long *__restrict__ cand_need /* give_up_over != 0: a bucket of more items (the
                          * fine items of a family the pieces do not split) is not walked -- millions of pairs by ONE
                          * wave, 16 s for the 65 536-key ladder -- but reported as a need beyond every budget: the host
                          * then takes the crowded buckets all pairs in tiles */,
    uint32_t require_any /* != 0: only pairs one of whose values has one of these bits are listed (the Levenshtein
                          * search for pairs of DIFFERENT lengths: a pair of two index items is a pair of one length,
                          * which the Hamming passes have found already -- 11 M of 11.2 M candidates at config 5's
                          * variant, each of them fetched and dropped by the verification before) */)
{
    constexpr uint32_t WAVES = GP_THREADS / 64;
    // GP_LISTS lists of list_cap pairs each, counters 8 words apart. A wave starts at "its" list
    // and moves to the next one after every flush, so one crowded bucket fills all lists evenly.
    const uint32_t list0 = (blockIdx.x * WAVES + (threadIdx.x >> 6)) % GP_LISTS;
    const uint64_t cand_cap = list_cap;
    __shared__ __attribute__((aligned(16))) uint32_t s_hash[WAVES][GP_SLICE];
    __shared__ uint32_t s_uid[WAVES][GP_SLICE];
    __shared__ uint32_t s_off[WAVES][66];        // sub-bin counts, then sub-bin starts (64 + end)
    __shared__ uint2 s_wbuf[WAVES][GP_WCAP];
    __shared__ uint32_t s_wcnt[WAVES], s_wflush[WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t *hashes = s_hash[wave], *uids = s_uid[wave], *off = s_off[wave];
    uint2 *wbuf = s_wbuf[wave];
    // s_wcnt / s_wflush: written by the leader lane, read by all -- relaxed workgroup-scope atomics, never cached in a
    // register (a volatile pointer to them lost its LDS address space: FLAT loads and stores, which wait on both the
    // vector-memory and the LDS counters)
    auto ld_shared = [](uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto st_shared = [](uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    // the wave's buffer -> the next list (called by converged lanes: `n` of them, ranks 0..n-1)
    auto flush = [&](uint32_t have, uint32_t n, uint32_t rank, int leader) {
        const uint32_t turn = ld_shared(&s_wflush[wave]);
        const uint32_t list = (list0 + turn) % GP_LISTS;
        unsigned long long g = 0;
        if ((int)lane == leader) {
            g = atomicAdd(cand_counts + (size_t)list * 8, (unsigned long long)have);
            st_shared(&s_wflush[wave], turn + 1);
        }
        g = __shfl(g, leader);
        uint2 *dst = cands_all + (size_t)list * list_cap;
        for (uint32_t e = rank; e < have; e += n)
            if (g + e < cand_cap)
                dst[g + e] = wbuf[e];
    };

    // Called by whatever lanes are active at the call site (they are converged there): the active
    // lanes take consecutive slots; a full buffer is written out first, by the same lanes.
    auto note = [&](uint32_t a, uint32_t b) {
        const unsigned long long act = __ballot(1);
        const uint32_t n = (uint32_t)__popcll(act), rank = (uint32_t)__popcll(act & fqd_lanemask_lt());
        const int leader = __ffsll((long long)act) - 1;
        uint32_t have = ld_shared(&s_wcnt[wave]);
        if (have + n > GP_WCAP) {
            flush(have, n, rank, leader);
            have = 0;
        }
        wbuf[have + rank] = make_uint2(a, b);
        if ((int)lane == leader)
            st_shared(&s_wcnt[wave], have + n);
    };

    if (lane == 0) {
        st_shared(&s_wcnt[wave], 0u);
        st_shared(&s_wflush[wave], 0u);
    }

    for (uint32_t b = blockIdx.x * WAVES + wave; b < n_buckets; b += gridDim.x * WAVES) {
        uint32_t two = 0;
        if (lane < 2)
            two = bucket_start[b + lane];
        const uint32_t lo = __shfl(two, 0);
        uint32_t hi = __shfl(two, 1);
        if (bucket_end)
            hi = min(hi, bucket_end[b]);
        const uint32_t m = hi - lo;
        if (m < 2 || (skip && skip[b]))
            continue;
        if (give_up_over && m > give_up_over) {
            if (lane == 0)
                atomicMax(cand_need, 1ull << 62);
            continue;
        }
        const uint2 *bucket = items + lo;
        const uint32_t m_lds = m < GP_SLICE ? m : GP_SLICE;
        // ---- load (all of the lane's loads in flight before anything else), rank, scan, place
        uint2 v[GP_SLICE / 64];
        uint32_t rank[GP_SLICE / 64];
#pragma unroll
        for (uint32_t k = 0; k < GP_SLICE / 64; k++) {
            const uint32_t t = lane + 64 * k;
            v[k] = make_uint2(0, 0);
            if (t < m_lds)
                v[k] = bucket[t];
        }
        off[lane] = 0;
#pragma unroll
        for (uint32_t k = 0; k < GP_SLICE / 64; k++)
            if (lane + 64 * k < m_lds)
                rank[k] = atomicAdd(&off[(v[k].x >> sub_shift) & 63u], 1u);
        {
            const uint32_t c = off[lane];
            uint32_t incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if ((int)lane >= o)
                    incl += up;
            }
            off[lane] = incl - c;          // start of sub-bin `lane`
            if (lane == 63)
                off[64] = incl;            // = m_lds
        }
#pragma unroll
        for (uint32_t k = 0; k < GP_SLICE / 64; k++)
            if (lane + 64 * k < m_lds) {
                const uint32_t p = off[(v[k].x >> sub_shift) & 63u] + rank[k];
                hashes[p] = v[k].x;
                uids[p] = v[k].y;
            }
        // (same wave wrote what it now reads: the LDS keeps a wave's accesses in order)
        // ---- every key against the keys after it in its sub-bin
        for (uint32_t i = lane; i < m_lds; i += 64) {
            const uint32_t h = hashes[i];
            const uint32_t end = off[((h >> sub_shift) & 63u) + 1];
            if (i + 1 >= end)
                continue;
            const uint32_t ui = uids[i];
            // windows of 8 hashes, 16-byte aligned, no branch per read
            for (uint32_t jbase = (i + 1) & ~3u; jbase < end; jbase += 8) {
                const uint32_t j1 = jbase + 4 < GP_SLICE ? jbase + 4 : GP_SLICE - 4;   // stay inside the slice
                const uint4 a = *reinterpret_cast<const uint4 *>(hashes + jbase);
                const uint4 c4 = *reinterpret_cast<const uint4 *>(hashes + j1);
                uint32_t match = (a.x == h ? 1u : 0u) | (a.y == h ? 2u : 0u) | (a.z == h ? 4u : 0u) |
                                 (a.w == h ? 8u : 0u) | (c4.x == h ? 16u : 0u) | (c4.y == h ? 32u : 0u) |
                                 (c4.z == h ? 64u : 0u) | (c4.w == h ? 128u : 0u);
                const uint32_t first = i + 1 > jbase ? i + 1 - jbase : 0u;      // < 4
                const uint32_t stop = end - jbase;                              // >= 1
                match &= 0xFFu << first;
                if (stop < 8)
                    match &= (1u << stop) - 1u;
                while (match) {
                    const uint32_t bit = __ffs((int)match) - 1;
                    match &= match - 1;
                    const uint32_t uj = uids[jbase + bit];
                    if (!require_any || ((ui | uj) & require_any))
                        note(ui, uj);
                }
            }
        }
        // ---- a bucket larger than the slice: its tail against the sorted part and against itself
        for (uint32_t t = GP_SLICE + lane; t < m; t += 64) {
            const uint2 me = bucket[t];
            const uint32_t sb = (me.x >> sub_shift) & 63u;
            for (uint32_t j = off[sb]; j < off[sb + 1]; j++)
                if (hashes[j] == me.x && (!require_any || ((uids[j] | me.y) & require_any)))
                    note(uids[j], me.y);
            for (uint32_t t2 = t + 1; t2 < m; t2++) {
                const uint2 it = bucket[t2];
                if (it.x == me.x && (!require_any || ((me.y | it.y) & require_any)))
                    note(me.y, it.y);
            }
        }
    }
    // what is left in the wave's buffer (all lanes are back together here)
    {
        const uint32_t have = ld_shared(&s_wcnt[wave]);
        if (have)
            flush(have, 64, lane, 0);
    }
}

// Phase 2 -- one thread per candidate (persistent grid, gridDim.x a multiple of GP_LISTS; cand_cap
// is the capacity of ONE list): fetch both records, count mismatches,
// keep the pair only in the pass of the first segment it agrees on; hits leave through an LDS
// edge buffer. Tens of thousands of independent record fetches are in flight at once.
// COOP (fixed-length records longer than one uint4): the 64 candidates of a wave are verified by
// 64 / Q groups of Q lanes, one uint4 of either record per lane -- every request a whole record line
// instead of 16 bytes of it (one thread per candidate: 1.46 ms per launch at config 5). The XOR of
// the two records is staged in LDS, the group's lanes count the mismatches of the 32-base words
// between them and note which earlier segments disagree; lane 0 of the group adds that up.
template <int K, bool COOP>
__global__ __launch_bounds__(GP_THREADS) void verify_candidates_kernel(
    const uint2 *__restrict__ cands, const unsigned long long *__restrict__ cand_count, uint64_t cand_cap,
    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t d, uint32_t seg,
    uint32_t nseg, uint32_t *__restrict__ edges, unsigned long long *__restrict__ edge_count, uint64_t edge_cap,
    unsigned long long *__restrict__ cand_need, fqd::PairStats *__restrict__ stats,
    uint32_t fused_U /* != 0: ALL passes in one -- a candidate holds positions in the [nseg][U] hash array (segment
                        * U + uid), and the pass it belongs to is its segment */)
{
    __shared__ uint32_t s_edges[2 * GP_ECAP];
    __shared__ uint32_t s_ctl[4];
    __shared__ uint32_t s_x[COOP ? GP_THREADS / 64 : 1][COOP ? 256 : 1];        // XOR dwords of the groups' records
    __shared__ uint32_t s_part[COOP ? GP_THREADS / 64 : 1][COOP ? 64 : 1][2];   // per lane: mismatches, disagreeing segments
    __shared__ uint8_t s_hit[COOP ? GP_THREADS / 64 : 1][COOP ? 64 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t Q = sh.stride / 4, groups = COOP ? 64u / Q : 1u;
    const uint32_t gl = COOP ? lane / Q : 0u, ql = COOP ? lane - gl * Q : 0u;
    // blocks l, l + GP_LISTS, ... sweep list l
    const uint32_t list = blockIdx.x % GP_LISTS, part = blockIdx.x / GP_LISTS, parts = gridDim.x / GP_LISTS;
    const unsigned long long filled = cand_count[(size_t)list * 8];
    const unsigned long long total = filled < cand_cap ? filled : cand_cap;
    cands += (size_t)list * cand_cap;
    if (part == 0 && tid == 0 && filled > cand_cap)
        atomicMax(cand_need, filled * GP_LISTS);   // a list overflowed: the host grows them and searches again
    unsigned long long n_pairs = 0, n_hits = 0;
    if (tid == 0)
        s_ctl[0] = 0;
    __syncthreads();
    // One thread per candidate; without COOP a thread takes VE candidates per sweep: the candidates, then the first
    // words of all 2 * VE records are requested together (clamped, unconditional) -- one candidate per sweep was two
    // dependent round trips and a barrier per 256 candidates.
    constexpr uint32_t VE = COOP ? 1u : 4u;
    constexpr uint32_t CR = 2;      // COOP: rounds of groups whose record gathers are in flight together (4: 120 VGPRs, no faster)
    const unsigned long long step = (unsigned long long)parts * GP_THREADS * VE;
    for (unsigned long long base = (unsigned long long)part * GP_THREADS * VE; base < total; base += step) {
        bool hits[VE];
        uint2 prs[VE];
        if (!COOP) {
            bool lives[VE];
            uint32_t segs[VE], dw0[VE];
#pragma unroll
            for (uint32_t t = 0; t < VE; t++) {
                const unsigned long long idx = base + (unsigned long long)t * GP_THREADS + tid;
                lives[t] = idx < total;
                prs[t] = cands[idx < total ? idx : total - 1];
            }
#pragma unroll
            for (uint32_t t = 0; t < VE; t++) {
                segs[t] = seg;
                n_pairs += lives[t] ? 1ull : 0ull;
                if (fused_U) {
                    uint32_t sx = 0, sy = 0;
                    while (prs[t].x >= fused_U) {
                        prs[t].x -= fused_U;
                        sx++;
                    }
                    while (prs[t].y >= fused_U) {
                        prs[t].y -= fused_U;
                        sy++;
                    }
                    segs[t] = seg + sx;                  // (seg: the first pass of the fused ones)
                    lives[t] = lives[t] && sx == sy;     // (hashes of different segments meet only by collision)
                }
            }
#pragma unroll
            for (uint32_t t = 0; t < VE; t++) {
                const uint32_t *a = urecs + (uint64_t)prs[t].x * sh.stride, *b = urecs + (uint64_t)prs[t].y * sh.stride;
                uint32_t dw = 0;
#pragma unroll
                for (int k = 0; k < K; k++)
                    dw |= a[k] ^ b[k];
                dw0[t] = dw;
            }
#pragma unroll
            for (uint32_t t = 0; t < VE; t++)
                hits[t] = lives[t] && gp_verify<K>(urecs, ulens, sh, d, segs[t], nseg, prs[t].x, prs[t].y, dw0[t]);
        }
        const unsigned long long idx = base + tid;
        bool hit = false, live = false;
        uint2 pr = make_uint2(0, 0);
        uint32_t my_seg = seg;
        if (COOP && idx < total) {
            pr = cands[idx];
            n_pairs++;
            live = true;
            if (fused_U) {
                uint32_t sx = 0, sy = 0;
                while (pr.x >= fused_U) {
                    pr.x -= fused_U;
                    sx++;
                }
                while (pr.y >= fused_U) {
                    pr.y -= fused_U;
                    sy++;
                }
                my_seg = seg + sx;
                live = sx == sy;             // (hashes of different segments meet only by collision)
            }
        }
        if (COOP) {
            const uint4 *recs4 = reinterpret_cast<const uint4 *>(urecs);
            const uint32_t W = sh.words;
            const unsigned long long have = __ballot(live);
            // Keys of ONE length: which bits of this lane's 32-base words (ql, ql + Q: W <= 2 Q for records of up to
            // 16 uint4... else the general loop) lie in segment s2 does not depend on the candidate -- worked out once
            // per sweep, not per word, segment and candidate (two divisions by nseg each: the inner loop was mostly that)
            // Ragged keys whose rows hold their length in their last word (sh.ragged & 2; bits 8.. of sh.ragged: a likely
            // length, api.hip modal_len_hint): no look-ups in ulens[] -- two sectors per candidate -- and the masks of that
            // likely length serve every pair that has it (nearly all of a FASTQ file's keys); other pairs work theirs out.
            const bool pow2q = (Q & (Q - 1)) == 0;
            const bool len_pad = (sh.ragged & 2u) != 0 && pow2q;
            const uint32_t fixed_len = sh.ragged ? (len_pad ? sh.ragged >> 8 : 0u) : sh.max_len;
            const bool fixed = fixed_len != 0 && nseg <= 4 && W <= 2 * Q;
            uint32_t fm0[4] = {0, 0, 0, 0}, fm1[4] = {0, 0, 0, 0};
            if (fixed) {
#pragma unroll
                for (uint32_t s2 = 0; s2 < 4; s2++)
                    if (s2 < nseg) {
                        uint32_t slo, shi;
                        fqd_segment(fixed_len, s2, nseg, slo, shi);
                        fm0[s2] = ql < W ? fqd_range_mask(ql, slo, shi) : 0u;
                        fm1[s2] = ql + Q < W ? fqd_range_mask(ql + Q, slo, shi) : 0u;
                    }
            }
            const bool pow2 = (Q & (Q - 1)) == 0;       // (groups that a butterfly of shuffles adds up)
            for (uint32_t c0 = 0; c0 < 64 && (have >> c0); c0 += CR * groups) {
                // CR rounds of groups: their 2 * CR loads per lane are requested together
                uint4 xs[CR];
                bool on[CR];
                uint32_t lens2[CR];     // the pair's key length; keys of different lengths are no Hamming neighbours
#pragma unroll
                for (uint32_t t = 0; t < CR; t++) {
                    const uint32_t cnd = c0 + t * groups + gl;
                    on[t] = gl < groups && cnd < 64 && ((have >> cnd) & 1ull);
                    // (a lane without a live candidate holds the pair (0, 0): record 0 is readable, so the loads are
                    // unconditional and both rounds' four gathers are in flight together -- "if (on) load" compiled
                    // to a branch with its own wait per round)
                    const uint32_t pu = __shfl(pr.x, cnd & 63u), pv = __shfl(pr.y, cnd & 63u);
                    lens2[t] = sh.max_len;
                    uint32_t len_v = sh.max_len;
                    if (sh.ragged && !len_pad) {
                        lens2[t] = ulens[pu];
                        len_v = ulens[pv];
                    }
                    const uint32_t qc = min(ql, Q - 1);
                    const uint4 a = recs4[(size_t)pu * Q + qc], b = recs4[(size_t)pv * Q + qc];
                    if (len_pad) {                       // (the group's last lane holds the rows' last uint4)
                        lens2[t] = __shfl(a.w, (int)((lane - ql) + Q - 1));
                        len_v = __shfl(b.w, (int)((lane - ql) + Q - 1));
                    }
                    xs[t] = make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
                    on[t] = on[t] && len_v == lens2[t];
                }
#pragma unroll
                for (uint32_t t = 0; t < CR; t++) {
                    const uint32_t cnd = c0 + t * groups + gl;
                    const uint32_t cseg = __shfl(my_seg, cnd & 63u);     // the candidate's pass
                    __builtin_amdgcn_wave_barrier();
                    if (gl < groups)
                        *reinterpret_cast<uint4 *>(&s_x[wave][(gl * Q + ql) * 4]) = xs[t];
                    __builtin_amdgcn_wave_barrier();
                    uint32_t dist = 0, seg_mis = 0;
                    if (on[t] && fixed && lens2[t] == fixed_len) {
                        const uint32_t *x = &s_x[wave][gl * Q * 4];
                        uint32_t dw0 = 0, dw1 = 0;
                        if (ql < W) {
#pragma unroll
                            for (int k = 0; k < K; k++)
                                dw0 |= x[ql * K + k];
                        }
                        if (ql + Q < W) {
#pragma unroll
                            for (int k = 0; k < K; k++)
                                dw1 |= x[(ql + Q) * K + k];
                        }
                        dist = __popc(dw0) + __popc(dw1);
#pragma unroll
                        for (uint32_t s2 = 0; s2 < 4; s2++)
                            if (s2 < cseg && ((dw0 & fm0[s2]) | (dw1 & fm1[s2])))
                                seg_mis |= 1u << s2;
                    } else if (on[t]) {
                        const uint32_t *x = &s_x[wave][gl * Q * 4];
                        for (uint32_t w = ql; w < W; w += Q) {
                            uint32_t dw = 0;
#pragma unroll
                            for (int k = 0; k < K; k++)
                                dw |= x[w * K + k];
                            dist += __popc(dw);
                            for (uint32_t s2 = 0; s2 < cseg; s2++) {
                                uint32_t slo, shi;
                                fqd_segment(lens2[t], s2, nseg, slo, shi);
                                if (dw & fqd_range_mask(w, slo, shi))
                                    seg_mis |= 1u << s2;
                            }
                        }
                    }
                    if (pow2) {
                        // (the Q lanes of a group: a butterfly; lanes of groups that are off hold zeros)
                        for (uint32_t o = Q >> 1; o; o >>= 1) {
                            dist += __shfl_xor(dist, o);
                            seg_mis |= __shfl_xor(seg_mis, o);
                        }
                    } else {
                        s_part[wave][lane][0] = dist;
                        s_part[wave][lane][1] = seg_mis;
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (gl < groups && cnd < 64 && ql == 0)
                        s_hit[wave][cnd] = 0;          // (a pair of different lengths stays a miss)
                    if (on[t] && ql == 0) {
                        uint32_t dsum = dist, mis = seg_mis;
                        if (!pow2) {
                            dsum = 0;
                            mis = 0;
                            for (uint32_t q = 0; q < Q; q++) {
                                dsum += s_part[wave][lane + q][0];
                                mis |= s_part[wave][lane + q][1];
                            }
                        }
                        // a neighbour, reported in the pass of the FIRST segment the pair agrees on: all
                        // earlier segments must disagree (an empty segment agrees trivially, as in gp_verify)
                        const uint32_t earlier = cseg >= 32 ? 0xFFFFFFFFu : (1u << cseg) - 1u;
                        s_hit[wave][cnd] = (dsum <= d && (mis & earlier) == earlier) ? 1 : 0;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            hit = live && s_hit[wave][lane] != 0;
            hits[0] = hit;
            prs[0] = pr;
        }
#pragma unroll
        for (uint32_t t = 0; t < VE; t++) {
            const unsigned long long mask = __ballot(hits[t]);
            if (mask) {
                uint32_t at = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if ((int)lane == leader)
                    at = atomicAdd(&s_ctl[0], (uint32_t)__popcll(mask));
                at = __shfl(at, leader);
                if (hits[t]) {
                    at += __popcll(mask & fqd_lanemask_lt());
                    const uint32_t eu = prs[t].x < prs[t].y ? prs[t].x : prs[t].y, ev = prs[t].x < prs[t].y ? prs[t].y : prs[t].x;
                    s_edges[2 * at] = eu;          // at < VE * 256 hits per sweep + < GP_ECAP / 2 left over
                    s_edges[2 * at + 1] = ev;
                    n_hits++;
                }
            }
        }
        __syncthreads();
        const uint32_t buffered = s_ctl[0];
        const bool last = base + step >= total;
        if (buffered >= GP_ECAP / 2 || (last && buffered)) {
            if (tid == 0) {
                const unsigned long long gb = atomicAdd(edge_count, (unsigned long long)buffered);
                s_ctl[1] = (uint32_t)gb;
                s_ctl[2] = (uint32_t)(gb >> 32);
            }
            __syncthreads();
            const unsigned long long gb = ((unsigned long long)s_ctl[2] << 32) | s_ctl[1];
            for (uint32_t e = tid; e < buffered; e += GP_THREADS)
                if (gb + e < edge_cap) {
                    edges[2 * (gb + e)] = s_edges[2 * e];
                    edges[2 * (gb + e) + 1] = s_edges[2 * e + 1];
                }
            __syncthreads();
            if (tid == 0)
                s_ctl[0] = 0;
            __syncthreads();
        }
    }
    if (stats) {
        for (int o = 32; o; o >>= 1) {
            n_pairs += __shfl_xor(n_pairs, o);
            n_hits += __shfl_xor(n_hits, o);
        }
        __syncthreads();
        unsigned long long *red = reinterpret_cast<unsigned long long *>(s_edges);
        if (lane == 0) {
            red[(tid >> 6) * 2 + 0] = n_pairs;
            red[(tid >> 6) * 2 + 1] = n_hits;
        }
        __syncthreads();
        if (tid < 2) {
            unsigned long long tot = 0;
            for (uint32_t wv = 0; wv < GP_THREADS / 64; wv++)
                tot += red[wv * 2 + tid];
            fqd::PairStats *slot = stats + (blockIdx.x % FQD_STAT_SLOTS);
            if (tot) {
                if (tid == 0) {
                    atomicAdd(&slot->pairs_compared, tot);
                    atomicAdd(&slot->keys_gathered, 2 * tot);   // two records fetched per candidate
                } else {
                    atomicAdd(&slot->edges, tot);
                }
            }
        }
    }
}


// ---- crowded buckets: a segment value shared by thousands of keys (low-complexity sequence, a constant prefix, a
// family of keys that differ in a few positions) makes all of them pairwise candidates -- quadratic, and one wave's
// work. Such keys are matched on FINER pieces instead: the key is cut into k pieces, and a key files one item per
// set M of d pieces, hashed over the whole key with the pieces of M masked out. Two keys within distance d differ
// inside at most d pieces -- the set D -- and meet in the items of every M that contains D; the pair is reported under
// ONE of them, the numerically smallest such M (D plus the lowest pieces outside it). Groups are as small as "keys
// equal outside d pieces". d = 1: k = 16 pieces, 16 items per key (8 pieces made groups of 256 keys of the skewed
// workload's ladder: 16.7 M candidates, 3 ms of verification); d = 2: k = 16, 120 items; d = 3: k = 8, 56 items.
// The verification keeps a pair iff the FIRST main segment it agrees on falls into a crowded bucket (else the main
// pass of that segment has reported it).
// (d = 2 with 8 pieces of a 32-nt key: the skewed workload's ladder -- every value of the last eight bases -- is ONE
// group under the set of its two pieces, 17 000 keys, 150 M candidates, and the search fell back to the sort path;
// with 16 pieces its groups are 256 keys)
constexpr uint32_t GP_FINE_MAX_D = 3, GP_FINE_MAX_SETS = 120, GP_FINE_SET_SHIFT = 25;      // (a set's index rides above 25 uid bits)
__host__ __device__ constexpr uint32_t gp_fine_k(uint32_t d) { return d <= 2 ? 16u : 8u; }
__host__ __device__ constexpr uint32_t gp_fine_sets(uint32_t d) { return d == 1 ? 16u : d == 2 ? 120u : d == 3 ? 56u : 0u; }

// the idx-th set of d pieces out of k, as a bit mask: sets in increasing numeric order of their masks (the combinatorial
// number system: idx = C(c_d, d) + ... + C(c_1, 1), c_d > ... > c_1)
__device__ __forceinline__ uint32_t gp_choose(uint32_t c, uint32_t t)
{
    return t == 1 ? c : t == 2 ? c * (c - 1u) / 2u : c * (c - 1u) * (c - 2u) / 6u;      // (c < t: 0 -- c - 1, c - 2 wrap only when a factor is 0)
}
__device__ __forceinline__ uint32_t gp_fine_set(uint32_t d, uint32_t idx)
{
    uint32_t set = 0;
    for (uint32_t t = d; t >= 1; t--) {
        uint32_t c = t - 1;                         // (C(t - 1, t) = 0 <= idx)
        while (c + 1 < 32 && (c + 1 >= t ? gp_choose(c + 1, t) : 0u) <= idx)
            c++;
        idx -= c >= t ? gp_choose(c, t) : 0u;
        set |= 1u << c;
    }
    return set;
}

// crowded[b]: 0 = an ordinary bucket; 1 = crowded, its keys are matched on finer pieces (list / counts[0..3]);
// 2 = crowded and of at most tile_max items: all pairs in tiles (list2 / counts[4..6]) -- a family of a few thousand keys
// that are all near each other (a jackpot key's one- and two-substitution copies) is cheap there and costly on the fine
// pieces, where each of its keys meets hundreds of others under every set that holds its differences.
// counts: [0] buckets, [1] items, [2] (keys that filed fine items), [3] sum of items^2 of list; [4], [5], [6] of list2.
__global__ void gp_mark_crowded_kernel(const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ bucket_end,
                                       uint32_t n_buckets, uint32_t limit, uint32_t tile_max, uint8_t *__restrict__ crowded,
                                       uint32_t *__restrict__ list, uint32_t *__restrict__ list2,
                                       unsigned long long *__restrict__ counts)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets)
        return;
    uint32_t hi = bucket_start[b + 1];
    if (bucket_end)
        hi = min(hi, bucket_end[b]);
    const uint32_t m = hi - bucket_start[b];
    const bool big = m > limit, small = big && m <= tile_max;
    crowded[b] = big ? (small ? 2 : 1) : 0;
    if (big) {
        unsigned long long *cn = counts + (small ? 4 : 0);
        (small ? list2 : list)[atomicAdd(&cn[0], 1ull)] = b;
        atomicAdd(&cn[1], (unsigned long long)m);
        atomicAdd(&cn[small ? 2 : 3], (unsigned long long)m * m);      // (all pairs in tiles: that many compares, twice)
    }
}

// The hash of a key with the pieces of `set` (bits over k pieces) masked out = a mix of the SUM of the hashes of the
// pieces left in (each piece hashed with its words' numbers, so the sum is position-aware): a key's k piece hashes are
// worked out once, one pass over its record, and each of its 16 / 120 / 56 items is the total minus d of them. (Round 4's
// first version hashed the whole masked record per item: 120 passes over a 120-byte record per key, 5.4 ms for the 62 K
// crowded keys of the skewed config 4.)
constexpr uint32_t GP_FINE_K_MAX = 16;

// every key with an item in a crowded bucket files its fine items (once: seen[uid])
__global__ __launch_bounds__(256) void gp_refine_items_kernel(
    const uint2 *__restrict__ items, const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ bucket_end,
    const uint32_t *__restrict__ list, const unsigned long long *__restrict__ counts, uint32_t fused_U,
    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t *__restrict__ seen,
    uint32_t *__restrict__ out_hash, uint32_t *__restrict__ out_val, unsigned long long *__restrict__ n_keys, uint64_t key_cap,
    uint32_t d)
{
    __shared__ uint32_t s_set[GP_FINE_MAX_SETS];
    __shared__ uint32_t s_ph[256][GP_FINE_K_MAX + 1];
    const uint32_t n_sets = gp_fine_sets(d), k = gp_fine_k(d), K = sh.planes;
    if (threadIdx.x < n_sets)
        s_set[threadIdx.x] = gp_fine_set(d, threadIdx.x);
    __syncthreads();
    uint32_t *ph = s_ph[threadIdx.x];
    // (every workgroup walks the whole list and takes its share of each bucket's items: there may be ONE crowded
    // bucket with a hundred thousand items)
    const uint32_t n_list = (uint32_t)counts[0];
    for (uint32_t li = 0; li < n_list; li++) {
        const uint32_t b = list[li], lo = bucket_start[b];
        uint32_t hi = bucket_start[b + 1];
        if (bucket_end)
            hi = min(hi, bucket_end[b]);
        for (uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
            uint32_t uid = items[i].y;
            if (fused_U)
                uid %= fused_U;
            if (atomicOr(&seen[uid >> 5], 1u << (uid & 31u)) & (1u << (uid & 31u)))
                continue;                              // (a key crowded in several passes files its items once)
            const unsigned long long at = atomicAdd(n_keys, 1ull);
            if (at >= key_cap)
                continue;
            const uint32_t *rec = urecs + (uint64_t)uid * sh.stride;
            const uint32_t len = fqd_key_len(sh, ulens, uid);
            uint32_t total = 0;
            for (uint32_t p = 0; p < k; p++) {
                const uint32_t plo = len * p / k, phi = len * (p + 1) / k;
                uint32_t h = 0;
                for (uint32_t w = plo >> 5; plo < phi && w <= (phi - 1u) >> 5; w++) {
                    const uint32_t m = fqd_range_mask(w, plo, phi);
                    for (uint32_t kk = 0; kk < K; kk++)
                        h += fqd_mix32((rec[w * K + kk] & m) + ((w * K + kk) * GP_FINE_K_MAX + p + 1u) * 0x9E3779B1u);
                }
                ph[p] = h;
                total += h;
            }
            const uint32_t len_mix = len * 0x9E3779B1u + 0x27D4EB2Fu;
            for (uint32_t j = 0; j < n_sets; j++) {
                const uint32_t set = s_set[j];
                uint32_t part = total;
                for (uint32_t rest = set; rest; rest &= rest - 1u)
                    part -= ph[__ffs(rest) - 1];
                out_hash[at * n_sets + j] = fqd_mix32(part + fqd_mix32(len_mix + set * 0x85EBCA77u));
                out_val[at * n_sets + j] = uid | (j << GP_FINE_SET_SHIFT);
            }
        }
    }
}

// candidates of the fine items -> edges. One thread per candidate (they are few).
__global__ __launch_bounds__(256) void gp_verify_refined_kernel(
    const uint2 *__restrict__ cands, const unsigned long long *__restrict__ cand_count, uint64_t cand_cap,
    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t nseg,
    const uint32_t *__restrict__ seg_hashes /* [nseg][U] */, uint64_t U, uint32_t bucket_bits,
    const uint8_t *__restrict__ crowded, uint32_t *__restrict__ edges, unsigned long long *__restrict__ edge_count,
    uint64_t edge_cap, unsigned long long *__restrict__ cand_need, uint32_t d, uint32_t accept /* bits of crowded[] values whose buckets this refinement answers for */)
{
    const uint32_t list = blockIdx.x % GP_LISTS, part = blockIdx.x / GP_LISTS, parts = gridDim.x / GP_LISTS;
    const unsigned long long filled = cand_count[(size_t)list * 8];
    const unsigned long long total = filled < cand_cap ? filled : cand_cap;
    if (part == 0 && threadIdx.x == 0 && filled > cand_cap)
        atomicMax(cand_need, filled * GP_LISTS);
    cands += (size_t)list * cand_cap;
    // four candidates per thread and sweep: the candidates, then both records' first words of all four, are requested
    // together (clamped, unconditional)
    // The hits of one sweep of the workgroup leave through an LDS buffer and ONE addition to the edge counter (an
    // addition per hit -- which the compiler turns into one per wave and sweep -- queued 31 000 additions on one address
    // for the ladder of the skewed workload: 0.38 ms for 2 M candidates).
    constexpr uint32_t VR = 4;
    constexpr uint32_t UID_MASK = (1u << GP_FINE_SET_SHIFT) - 1u;
    __shared__ uint2 s_edge[256 * VR];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_at;
    __shared__ uint32_t s_set[GP_FINE_MAX_SETS];
    const uint32_t k = gp_fine_k(d);
    if (threadIdx.x < gp_fine_sets(d))
        s_set[threadIdx.x] = gp_fine_set(d, threadIdx.x);
    for (unsigned long long base = (unsigned long long)part * blockDim.x * VR; base < total;
         base += (unsigned long long)parts * blockDim.x * VR) {
        if (threadIdx.x == 0)
            s_n = 0;
        __syncthreads();
        uint2 pr[VR];
        bool live[VR];
#pragma unroll
        for (uint32_t t = 0; t < VR; t++) {
            const unsigned long long idx = base + (unsigned long long)t * blockDim.x + threadIdx.x;
            live[t] = idx < total;
            pr[t] = cands[idx < total ? idx : total - 1];
        }
        uint32_t dw0[VR];
#pragma unroll
        for (uint32_t t = 0; t < VR; t++) {
            const uint32_t ua = pr[t].x & UID_MASK, ub = pr[t].y & UID_MASK;
            const uint32_t *ra = urecs + (uint64_t)ua * sh.stride, *rb = urecs + (uint64_t)ub * sh.stride;
            uint32_t dw = 0;
            for (uint32_t kk = 0; kk < sh.planes; kk++)
                dw |= ra[kk] ^ rb[kk];
            dw0[t] = dw;
        }
#pragma unroll
        for (uint32_t t = 0; t < VR; t++) {
            const uint32_t ja = pr[t].x >> GP_FINE_SET_SHIFT, jb = pr[t].y >> GP_FINE_SET_SHIFT;
            const uint32_t ua = pr[t].x & UID_MASK, ub = pr[t].y & UID_MASK;
            if (!live[t] || ja != jb || ua == ub)
                continue;
            const uint32_t len = fqd_key_len(sh, ulens, ua);
            if (fqd_key_len(sh, ulens, ub) != len)
                continue;
            const uint32_t *ra = urecs + (uint64_t)ua * sh.stride, *rb = urecs + (uint64_t)ub * sh.stride;
            // the mismatching positions (at most d, else out): their pieces, and the main segments they fall into
            uint32_t dist = 0, pieces = 0, seg_mis = 0;
            for (uint32_t w = 0; w < sh.words && dist <= d; w++) {
                uint32_t dw = dw0[t];
                if (w) {
                    dw = 0;
                    for (uint32_t kk = 0; kk < sh.planes; kk++)
                        dw |= ra[w * sh.planes + kk] ^ rb[w * sh.planes + kk];
                }
                dist += __popc(dw);
                while (dw && dist <= d) {
                    const uint32_t pos = w * 32u + (uint32_t)(__ffs((int)dw) - 1);
                    dw &= dw - 1u;
                    uint32_t j = (uint32_t)((uint64_t)pos * k / len);         // the piece [len j / k, len (j + 1) / k) of pos
                    while (j && pos < len * j / k)
                        j--;
                    while (j + 1 < k && pos >= len * (j + 1) / k)
                        j++;
                    pieces |= 1u << j;
                    for (uint32_t s2 = 0; s2 < nseg; s2++) {
                        uint32_t slo, shi;
                        fqd_segment(len, s2, nseg, slo, shi);
                        if (pos >= slo && pos < shi)
                            seg_mis |= 1u << s2;
                    }
                }
            }
            if (dist == 0 || dist > d)
                continue;
            // the set both items left out must be THE set of this pair: its pieces and the lowest others (else the
            // pair is reported under another set, or the hashes collided)
            const uint32_t set = s_set[ja];
            if (pieces & ~set)
                continue;
            uint32_t want = pieces, others = ~pieces & ((1u << k) - 1u);
            for (uint32_t need = d - (uint32_t)__popc(pieces); need; need--) {
                want |= others & (0u - others);
                others &= others - 1u;
            }
            if (want != set)
                continue;
            // the first main segment the pair agrees on (an empty segment agrees trivially)
            uint32_t first = 0;
            while (first < nseg && (seg_mis >> first & 1u))
                first++;
            if (first >= nseg)
                continue;                                  // (no segment agrees: cannot be within d of d + 1 segments)
            if (!(crowded[seg_hashes[(size_t)first * U + ua] >> (32u - bucket_bits)] & accept))
                continue;                                  // the main pass of that segment has it
            s_edge[atomicAdd(&s_n, 1u)] = make_uint2(min(ua, ub), max(ua, ub));
        }
        __syncthreads();
        const uint32_t n_hit = s_n;
        if (threadIdx.x == 0 && n_hit)
            s_at = atomicAdd(edge_count, (unsigned long long)n_hit);
        __syncthreads();
        if (n_hit) {
            const unsigned long long at = s_at;
            for (uint32_t e = threadIdx.x; e < n_hit; e += blockDim.x)
                if (at + e < edge_cap)
                    reinterpret_cast<uint2 *>(edges)[at + e] = s_edge[e];
        }
    }
}

// ---- crowded buckets, the last resort: ALL PAIRS of a bucket, tiled over the whole GPU ---------------------------
// The fine pieces above split a crowded bucket only when its keys differ OUTSIDE a few pieces; a family that varies
// inside one piece (all 4^8 values of eight adjacent bases of a 300-nt key: 65 536 keys, each with 276 neighbours
// within distance 2) stays one group, its candidate pairs outgrow every budget, and until round 4 the search then took
// the sort path -- which walks such a run in ONE wave: 19 s for 25 M reads. All pairs of m keys are m^2 / 2 cheap
// compares when they are spread over the GPU: tiles of 128 x 128 rows, both tiles' records staged in LDS, a row per
// thread against every row of the other tile (a broadcast read per word), hash first (a bucket holds many segment
// values), then the exact distance and the first-agreeing-segment rule -- 2 G pairs of 128-byte records in a few ms.
constexpr uint32_t CT_TS = 128, CT_WCAP = 192;

__global__ void gp_tile_prefix_kernel(const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ bucket_end,
                                      const uint32_t *__restrict__ list, const unsigned long long *__restrict__ counts,
                                      unsigned long long *__restrict__ tile_prefix /* [n_list + 1] */)
{
    if (blockIdx.x || threadIdx.x)
        return;
    const uint32_t n_list = (uint32_t)counts[0];
    unsigned long long run = 0;
    for (uint32_t li = 0; li < n_list; li++) {
        const uint32_t b = list[li];
        uint32_t hi = bucket_start[b + 1];
        if (bucket_end)
            hi = min(hi, bucket_end[b]);
        const unsigned long long T = (hi - bucket_start[b] + CT_TS - 1) / CT_TS;
        tile_prefix[li] = run;
        run += T * (T + 1) / 2;
    }
    tile_prefix[n_list] = run;
}

// K planes, Q4 uint4 per record (1, 2, 4 or 8): a thread's own record sits in registers, the other tile's rows in LDS at
// a stride of Q4 * 4 + 4 words (16-byte aligned: a row is read by Q4 broadcast ds_read_b128, all in flight together).
// (First version: both tiles in LDS, word-by-word loops with run-time bounds and an early exit -- every word a dependent
// LDS round trip, two waves per SIMD to hide it: 450 ms for the skewed config 4 where this one takes __TILES_MS__.)
template <uint32_t K, uint32_t Q4>
__global__ __launch_bounds__(CT_TS) void gp_crowded_tiles_kernel(
    const uint2 *__restrict__ items, const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ bucket_end,
    const uint32_t *__restrict__ list, const unsigned long long *__restrict__ counts,
    const unsigned long long *__restrict__ tile_prefix, uint32_t fused_U, uint32_t seg0,
    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t d, uint32_t nseg,
    uint32_t *__restrict__ edges, unsigned long long *__restrict__ edge_count, uint64_t edge_cap)
{
    constexpr uint32_t ROW4 = Q4 + 1, WMAX = Q4 * 4 / K;        // (words of 32 bases a record of Q4 uint4 can hold)
    __shared__ uint4 rec_j[CT_TS * ROW4];
    __shared__ uint32_t h_j[CT_TS], uid_j[CT_TS], seg_j[CT_TS];
    __shared__ uint2 wbuf[CT_TS / 64][CT_WCAP];
    __shared__ uint32_t s_wn[CT_TS / 64];
    __shared__ uint32_t s_mask[WMAX][8];                  // keys of ONE length: the bits of word w inside main segment s
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t W = sh.words;
    if (lane == 0)
        s_wn[wave] = 0;
    for (uint32_t x = tid; x < WMAX * 8u; x += CT_TS) {
        uint32_t slo = 0, shi = 0;
        if ((x & 7u) < nseg)
            fqd_segment(sh.max_len, x & 7u, nseg, slo, shi);
        s_mask[x >> 3][x & 7u] = (x & 7u) < nseg ? fqd_range_mask(x >> 3, slo, shi) : 0u;
    }
    const uint32_t n_list = (uint32_t)counts[0];
    const unsigned long long total = tile_prefix[n_list];
    auto flush = [&](uint32_t have) {
        unsigned long long g = 0;
        if (lane == 0)
            g = atomicAdd(edge_count, (unsigned long long)have);
        g = __shfl(g, 0);
        for (uint32_t e = lane; e < have; e += 64)
            if (g + e < edge_cap)
                reinterpret_cast<uint2 *>(edges)[g + e] = wbuf[wave][e];
    };
    const uint4 *recs4 = reinterpret_cast<const uint4 *>(urecs);
    for (unsigned long long g = blockIdx.x; g < total; g += gridDim.x) {
        // which bucket, which pair of tiles
        uint32_t a = 0, z = n_list;
        while (z - a > 1) {
            const uint32_t mid = (a + z) >> 1;
            if (tile_prefix[mid] <= g)
                a = mid;
            else
                z = mid;
        }
        const uint32_t b = list[a], lo = bucket_start[b];
        uint32_t hi = bucket_start[b + 1];
        if (bucket_end)
            hi = min(hi, bucket_end[b]);
        const uint32_t m = hi - lo, T = (m + CT_TS - 1) / CT_TS;
        const unsigned long long idx = g - tile_prefix[a];
        // row-major over the upper triangle: row ti holds T - ti pairs (ti, ti .. T - 1)
        uint32_t ti = (uint32_t)(((2.0 * T + 1.0) - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)idx)) * 0.5);
        auto row_start = [&](uint32_t r) { return (unsigned long long)r * T - (unsigned long long)r * (r - 1) / 2; };
        ti = min(ti, T - 1);
        while (ti > 0 && row_start(ti) > idx)
            ti--;
        while (ti + 1 < T && row_start(ti + 1) <= idx)
            ti++;
        const uint32_t tj = ti + (uint32_t)(idx - row_start(ti));
        const uint32_t ni = min(CT_TS, m - ti * CT_TS), nj = min(CT_TS, m - tj * CT_TS);
        __syncthreads();                                  // (the tile of the pair before is done with)
        // my row of tile i: item and record (registers); row tid of tile j: item (LDS)
        const uint2 it_i = items[lo + ti * CT_TS + min(tid, ni - 1)];
        const uint32_t my_h = it_i.x, my_uid = fused_U ? it_i.y % fused_U : it_i.y;
        const uint32_t my_seg = seg0 + (fused_U ? it_i.y / fused_U : 0u);
        uint4 mine[Q4];
#pragma unroll
        for (uint32_t q = 0; q < Q4; q++)
            mine[q] = recs4[(size_t)my_uid * Q4 + q];
        if (tid < nj) {
            const uint2 it = items[lo + tj * CT_TS + tid];
            h_j[tid] = it.x;
            uid_j[tid] = fused_U ? it.y % fused_U : it.y;
            seg_j[tid] = seg0 + (fused_U ? it.y / fused_U : 0u);
        }
        __syncthreads();
        // tile j's records as one coalesced stream of uint4
        for (uint32_t x0 = tid; x0 < CT_TS * Q4; x0 += 4 * CT_TS) {
            uint4 vj[4];
#pragma unroll
            for (uint32_t t = 0; t < 4; t++) {             // (clamped, unconditional: in flight together)
                const uint32_t x = x0 + t * CT_TS, r = min(x / Q4, nj - 1), q = x % Q4;
                vj[t] = recs4[(size_t)uid_j[r] * Q4 + q];
            }
#pragma unroll
            for (uint32_t t = 0; t < 4; t++) {
                const uint32_t x = x0 + t * CT_TS, r = x / Q4, q = x % Q4;
                if (x < CT_TS * Q4 && r < nj)
                    rec_j[r * ROW4 + q] = vj[t];
            }
        }
        __syncthreads();
        const uint32_t my_len = sh.ragged ? fqd_key_len(sh, ulens, my_uid) : sh.max_len;
        const uint32_t *mw = reinterpret_cast<const uint32_t *>(mine);
        for (uint32_t k = 0; k < nj; k++) {
            const uint32_t uk = uid_j[k];
            const bool cand = tid < ni && h_j[k] == my_h && seg_j[k] == my_seg && (ti != tj || k > tid) && uk != my_uid;
            if (!__ballot(cand))
                continue;                                  // (nobody in the wave: the row is not even read)
            uint4 other[Q4];
#pragma unroll
            for (uint32_t q = 0; q < Q4; q++)
                other[q] = rec_j[k * ROW4 + q];
            const uint32_t *ow = reinterpret_cast<const uint32_t *>(other);
            uint32_t dist = 0;
#pragma unroll
            for (uint32_t w = 0; w < WMAX; w++) {
                uint32_t dw = 0;
#pragma unroll
                for (uint32_t kk = 0; kk < K; kk++)
                    dw |= mw[w * K + kk] ^ ow[w * K + kk];
                dist += w < W ? __popc(dw) : 0u;
            }
            bool hit = cand && dist && dist <= d && (!sh.ragged || fqd_key_len(sh, ulens, uk) == my_len);
            if (hit) {                                      // (rare: which main segments hold the differences)
                uint32_t seg_mis = 0;
#pragma unroll
                for (uint32_t w = 0; w < WMAX; w++) {
                    uint32_t dw = 0;
#pragma unroll
                    for (uint32_t kk = 0; kk < K; kk++)
                        dw |= mw[w * K + kk] ^ ow[w * K + kk];
                    if (w < W && dw) {
                        if (!sh.ragged) {
                            for (uint32_t s2 = 0; s2 < nseg; s2++)
                                seg_mis |= (dw & s_mask[w][s2]) ? 1u << s2 : 0u;
                        } else {
                            for (uint32_t s2 = 0; s2 < nseg; s2++) {
                                uint32_t slo, shi;
                                fqd_segment(my_len, s2, nseg, slo, shi);
                                seg_mis |= (dw & fqd_range_mask(w, slo, shi)) ? 1u << s2 : 0u;
                            }
                        }
                    }
                }
                uint32_t first = 0;
                while (first < nseg && (seg_mis >> first & 1u))
                    first++;
                hit = first == my_seg;                   // reported in the pass of the FIRST segment the pair agrees on
            }
            const unsigned long long hm = __ballot(hit);
            if (hm) {
                const uint32_t n = (uint32_t)__popcll(hm);
                uint32_t have = s_wn[wave];
                if (have + n > CT_WCAP) {
                    flush(have);
                    have = 0;
                }
                if (hit)
                    wbuf[wave][have + (uint32_t)__popcll(hm & fqd_lanemask_lt())] = make_uint2(min(my_uid, uk), max(my_uid, uk));
                if (lane == 0)
                    s_wn[wave] = have + n;
            }
        }
    }
    __syncthreads();
    const uint32_t left = s_wn[wave];
    if (left)
        flush(left);
}

}  // namespace

namespace fqd {

uint32_t group_tile_size() { return fqd_partition::THREADS * PairPolicy::EPT; }
uint32_t group_cand_lists() { return GP_LISTS; }
uint32_t group_max_bins() { return fqd_partition::MAX_BINS; }

hipError_t launch_group_hist(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                             const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                             uint32_t n_bins, uint32_t *hist, hipStream_t st)
{
    if (!max_tiles)
        return hipSuccess;
    const PairPolicy::Source src{hashes, reinterpret_cast<const uint2 *>(in), nullptr};
    if (level1)
        gp_hist_kernel<true><<<max_tiles, fqd_partition::THREADS, 0, st>>>(src, seg_start, tile_start, n_seg, shift,
                                                                           n_bins, hist);
    else
        gp_hist_kernel<false><<<max_tiles, fqd_partition::THREADS, 0, st>>>(src, seg_start, tile_start, n_seg, shift,
                                                                            n_bins, hist);
    return hipGetLastError();
}

// level 1 with l1_subs != 0: slab mode (slab_cap slots per (sub-part, bin)); level 2 with seg_end != NULL: over
// the slab segments such a level 1 left, segment -> part by seg_mask
hipError_t launch_group_scatter(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                                const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                                uint32_t n_bins, uint32_t *cursor, uint32_t *out, hipStream_t st, uint32_t slab_cap,
                                uint32_t *slab_overflow, const uint32_t *values, uint32_t l1_subs,
                                const uint32_t *seg_end, uint32_t seg_mask)
{
    if (!max_tiles)
        return hipSuccess;
    if (l1_subs & (l1_subs - 1))
        return hipErrorInvalidValue;
    const PairPolicy::Source src{hashes, reinterpret_cast<const uint2 *>(in), values};
    uint2 *out2 = reinterpret_cast<uint2 *>(out);
    if (level1)
        gp_scatter_kernel<true><<<max_tiles, fqd_partition::THREADS, 0, st>>>(
            src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out2, l1_subs ? slab_cap : 0u,
            l1_subs ? slab_overflow : nullptr, nullptr, 0xFFFFFFFFu, l1_subs);
    else
        gp_scatter_kernel<false><<<max_tiles, fqd_partition::THREADS, 0, st>>>(
            src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out2, slab_cap, slab_overflow, seg_end, seg_mask, 0u);
    return hipGetLastError();
}

hipError_t launch_group_slab_tile_starts(const uint32_t *seg_start, const uint32_t *seg_end, uint32_t n_seg,
                                         uint32_t *tile_start, hipStream_t st)
{
    gp_slab_tile_starts_kernel<<<1, 1024, 0, st>>>(seg_start, seg_end, n_seg, tile_start);
    return hipGetLastError();
}

hipError_t launch_group_tile_starts(const uint32_t *seg_start, uint32_t n_seg, uint32_t *tile_start, hipStream_t st)
{
    gp_tile_starts_kernel<<<1, 256, 0, st>>>(seg_start, n_seg, tile_start);
    return hipGetLastError();
}

hipError_t launch_group_matrix_starts(const uint32_t *matrix_incl, uint32_t n_bins, uint32_t n_tiles, uint32_t *start,
                                      hipStream_t st)
{
    gp_matrix_starts_kernel<<<(n_bins + 1 + 255) / 256, 256, 0, st>>>(matrix_incl, n_bins, n_tiles, start);
    return hipGetLastError();
}

hipError_t launch_group_bucket_starts(const uint32_t *hist_incl, uint32_t n_buckets, uint32_t *bucket_start,
                                      uint32_t *cursor, hipStream_t st)
{
    gp_bucket_starts_kernel<<<(n_buckets + 1 + 255) / 256, 256, 0, st>>>(hist_incl, n_buckets, bucket_start, cursor);
    return hipGetLastError();
}

hipError_t launch_group_pass_init(uint32_t *seg1, uint32_t *tiles1, uint32_t n_items, uint32_t n_tiles,
                                  unsigned long long *cand_ctr, uint32_t ctr_words, hipStream_t st, PassInitMore more)
{
    uint32_t most = std::max(ctr_words, more.stat_words);
    if (more.start1)
        most = std::max(most, more.n1 + 1);
    if (more.start2)
        most = std::max(most, more.n2 + 1);
    gp_pass_init_kernel<<<(most + 255) / 256 + 1, 256, 0, st>>>(seg1, tiles1, n_items, n_tiles, cand_ctr, ctr_words, more);
    return hipGetLastError();
}

hipError_t launch_group_slab_starts(uint32_t n_buckets, uint32_t cap, uint32_t *bucket_start, uint32_t *cursor,
                                    hipStream_t st)
{
    gp_slab_starts_kernel<<<(n_buckets + 1 + 255) / 256, 256, 0, st>>>(n_buckets, cap, bucket_start, cursor);
    return hipGetLastError();
}

hipError_t launch_grouped_candidates(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                     uint32_t n_buckets, uint32_t bucket_bits, uint64_t *cands,
                                     unsigned long long *cand_count, uint64_t cand_cap, hipStream_t st,
                                     uint32_t require_any, const uint8_t *skip, uint32_t give_up_over,
                                     unsigned long long *cand_need)
{
    if (!n_buckets)
        return hipSuccess;
    // the 6 hash bits below the bucket bits pick the sub-bin (bucket_bits <= 26)
    const uint32_t sub_shift = 32u - bucket_bits - 6u;
    const uint32_t blocks = (n_buckets + 3) / 4;
    const unsigned grid = blocks < 8192 ? blocks : 8192;
    grouped_candidates_kernel<<<grid, GP_THREADS, 0, st>>>(reinterpret_cast<const uint2 *>(items), bucket_start,
                                                           bucket_end, n_buckets, sub_shift, reinterpret_cast<uint2 *>(cands),
                                                           cand_count, cand_cap / GP_LISTS, skip,
                                                           cand_need ? give_up_over : 0u, cand_need, require_any);
    return hipGetLastError();
}

hipError_t launch_verify_candidates(const uint64_t *cands, const unsigned long long *cand_count, uint64_t cand_cap,
                                    const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t d, uint32_t seg,
                                    uint32_t nseg, uint32_t *edges, unsigned long long *edge_count, uint64_t edge_cap,
                                    unsigned long long *cand_need, PairStats *stats, hipStream_t st, uint32_t fused_U)
{
    // several lanes per candidate for fixed-length records of four uint4 and more (48-byte records,
    // config 2: 0.119 ms against 0.092 one thread per candidate -- most candidates there are no
    // neighbours and are out after one word; 128-byte records, config 5: 0.44 against 1.46 ms), and at
    // most 32 segments: the per-lane notes are a bit mask
    // (records of ONE 64-byte sector -- keys of 65-128 nt, padded to a 64-byte stride since round 4 -- take the
    // one-thread way again: 0.159 against 0.174 ms at config 2)
    const bool coop = sh.stride >= 20 && sh.stride <= 256 && !(sh.stride & 3u) && nseg <= 32 &&
                      !getenv("FQD_VERIFY_NO_COOP");
#define FQD_GP_CASE(KK)                                                                                              \
    case KK:                                                                                                         \
        if (coop)                                                                                                    \
            verify_candidates_kernel<KK, true><<<2048, GP_THREADS, 0, st>>>(                                         \
                reinterpret_cast<const uint2 *>(cands), cand_count, cand_cap / GP_LISTS, urecs, ulens, sh, d, seg, nseg,  \
                edges, edge_count, edge_cap, cand_need, stats, fused_U);                                             \
        else                                                                                                         \
            verify_candidates_kernel<KK, false><<<2048, GP_THREADS, 0, st>>>(                                        \
                reinterpret_cast<const uint2 *>(cands), cand_count, cand_cap / GP_LISTS, urecs, ulens, sh, d, seg, nseg,  \
                edges, edge_count, edge_cap, cand_need, stats, fused_U);                                             \
        break;
    switch (sh.planes) {
        FQD_GP_CASE(1)
        FQD_GP_CASE(2)
        FQD_GP_CASE(3)
        FQD_GP_CASE(4)
        FQD_GP_CASE(5)
        FQD_GP_CASE(6)
        FQD_GP_CASE(7)
    default:
        return hipErrorInvalidValue;
    }
#undef FQD_GP_CASE
    return hipGetLastError();
}

uint32_t group_fine_items(uint32_t d) { return gp_fine_sets(d); }       // fine items a crowded key files (0: no refinement at d)
uint32_t group_fine_uid_bits() { return GP_FINE_SET_SHIFT; }

hipError_t launch_group_mark_crowded(const uint32_t *bucket_start, const uint32_t *bucket_end, uint32_t n_buckets,
                                     uint32_t limit, uint32_t tile_max, uint8_t *crowded, uint32_t *list, uint32_t *list2,
                                     unsigned long long *counts, hipStream_t st)
{
    gp_mark_crowded_kernel<<<(n_buckets + 255) / 256, 256, 0, st>>>(bucket_start, bucket_end, n_buckets, limit, tile_max,
                                                                    crowded, list, list2, counts);
    return hipGetLastError();
}

hipError_t launch_group_refine_items(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                     const uint32_t *list, const unsigned long long *counts, uint32_t fused_U,
                                     const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t *seen,
                                     uint32_t *out_hash, uint32_t *out_val, unsigned long long *n_keys, uint64_t key_cap,
                                     hipStream_t st, uint32_t d)
{
    if (!gp_fine_sets(d))
        return hipErrorInvalidValue;
    gp_refine_items_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const uint2 *>(items), bucket_start, bucket_end, list,
                                                 counts, fused_U, urecs, ulens, sh, seen, out_hash, out_val, n_keys,
                                                 key_cap, d);
    return hipGetLastError();
}

// all pairs of the crowded buckets (marked by launch_group_mark_crowded; list / counts from there), tiled:
// tile_prefix has room for n_buckets + 1 words of 8 bytes. Records of up to 32 words; nseg <= 8.
hipError_t launch_group_crowded_tiles(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                      const uint32_t *list, const unsigned long long *counts, unsigned long long *tile_prefix,
                                      uint32_t fused_U, uint32_t seg0, const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                                      uint32_t d, uint32_t nseg, uint32_t *edges, unsigned long long *edge_count,
                                      uint64_t edge_cap, hipStream_t st)
{
    const uint32_t q4 = sh.stride / 4;
    if (sh.stride > 32 || (sh.stride & 3u) || (q4 & (q4 - 1)) || nseg > 8 || sh.planes < 2 || sh.planes > 3)
        return hipErrorInvalidValue;
    gp_tile_prefix_kernel<<<1, 64, 0, st>>>(bucket_start, bucket_end, list, counts, tile_prefix);
#define FQD_TILES(KK, QQ)                                                                                               \
    gp_crowded_tiles_kernel<KK, QQ><<<8192, CT_TS, 0, st>>>(reinterpret_cast<const uint2 *>(items), bucket_start,      \
                                                            bucket_end, list, counts, tile_prefix, fused_U, seg0, urecs, \
                                                            ulens, sh, d, nseg, edges, edge_count, edge_cap)
    if (sh.planes == 3) {
        if (q4 == 1) FQD_TILES(3, 1); else if (q4 == 2) FQD_TILES(3, 2); else if (q4 == 4) FQD_TILES(3, 4); else FQD_TILES(3, 8);
    } else {
        if (q4 == 1) FQD_TILES(2, 1); else if (q4 == 2) FQD_TILES(2, 2); else if (q4 == 4) FQD_TILES(2, 4); else FQD_TILES(2, 8);
    }
#undef FQD_TILES
    return hipGetLastError();
}

// (what the launcher above takes)
bool group_tiles_possible(KeyShape sh, uint32_t nseg)
{
    const uint32_t q4 = sh.stride / 4;
    return sh.stride <= 32 && !(sh.stride & 3u) && !(q4 & (q4 - 1)) && nseg <= 8 && sh.planes >= 2 && sh.planes <= 3;
}

hipError_t launch_group_verify_refined(const uint64_t *cands, const unsigned long long *cand_count, uint64_t cand_cap,
                                       const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t nseg,
                                       const uint32_t *seg_hashes, uint64_t U, uint32_t bucket_bits,
                                       const uint8_t *crowded, uint32_t *edges, unsigned long long *edge_count,
                                       uint64_t edge_cap, unsigned long long *cand_need, hipStream_t st, uint32_t d,
                                       uint32_t accept)
{
    if (!gp_fine_sets(d))
        return hipErrorInvalidValue;
    gp_verify_refined_kernel<<<GP_LISTS * 16, 256, 0, st>>>(reinterpret_cast<const uint2 *>(cands), cand_count,
                                                            cand_cap / GP_LISTS, urecs, ulens, sh, nseg, seg_hashes, U,
                                                            bucket_bits, crowded, edges, edge_count, edge_cap, cand_need, d,
                                                            accept);
    return hipGetLastError();
}

}  // namespace fqd
