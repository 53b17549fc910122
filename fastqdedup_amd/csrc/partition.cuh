// partition.cuh -- the LDS-aggregated hash partition used twice on the hot path: by the exact-
// duplicate collapse (16-byte records, collapse_lds.hip) and by the neighbour search ((segment
// hash, uid) pairs, group.hip). No reference counterpart: it replaces device-wide radix sorts.
//
// Items are partitioned by some bits of a 32-bit key in two levels. Each level is a histogram pass
// and a scatter pass over TILES of 2048 or 4096 items; tiles never straddle a segment (level 1: one
// segment = everything; level 2: the parts level 1 made). Bin counts and the ranks inside a bin
// are LDS atomics, so global memory sees one counter update per (tile, bin) instead of one per
// item (a scatter with one global atomic per item took 5.3 ms for 50 M reads; this takes ~1 ms).
// Level 1 has ONE segment -- thousands of tiles would hammer the same few hundred counters -- so
// its counts go to a (bin x tile) matrix whose scan in bin-major order IS every (tile, bin)'s
// output position: no atomics, deterministic placement. In the scatter pass the tile is counting-
// sorted by bin in LDS before it leaves, so every bin's share goes out as ONE contiguous run of
// stores (a lone 16-byte store costs a whole 64-byte HBM burst).
//
// A Policy names the item type and how an item and its key are fetched:
//   using Item = ...;  struct Source { ... };
//   using KeyRaw = ...;  key_fetch<LEVEL1>(src, i) -> KeyRaw;  key_finish<LEVEL1>(src, i, raw) -> key   // histogram pass
//   using Raw = ...;                                                                            // scatter pass, two steps:
//   template <bool LEVEL1> static __device__ Raw fetch(const Source &, uint32_t i);             //   the loads alone (no branch, no side effect)
//   template <bool LEVEL1> static __device__ uint32_t finish(const Source &, uint32_t i, const Raw &, Item &, bool valid, uint32_t seg_tag);  // -> key
//   struct Shared { ... };  per-workgroup LDS state of the scatter pass, handed to finish() and, after the tile has
//   left, to  static __device__ void flush(const Source &, Shared &, uint32_t tid)   (all threads call it)
//   (the loads of a whole tile must be in flight together: "if (i < hi) { load; hash }" compiles to one branch per
//   item with s_waitcnt vmcnt(0) inside -- EPT dependent round trips per tile; see tools/isa_skeleton.py. The bodies
//   below therefore fetch at min(i, hi - 1), unconditionally, and finish afterwards; valid = i < hi)
//   static __device__ uint32_t segment_tag(const Source &, uint32_t segment);  static __device__ void apply_tag(Item &, uint32_t tag);
//   static constexpr bool MAY_SKIP;  static __device__ bool skip(const Item &);   // scatter pass: an item load() has disposed of otherwise
//   static constexpr uint32_t EPT, ROUNDS;   // items per thread (tile = 256 * EPT); staging rounds of the scatter pass
//   static constexpr bool CAN_SPILL;   // slab mode: items behind a full slab's end go to a list of the Policy's --
//   spill_reserve(src, n) -> first index, spill_write(src, index, item) -- instead of raising the overflow flag
#pragma once
#include "fqd_internal.h"

namespace fqd_partition {

constexpr uint32_t THREADS = 256;
constexpr uint32_t MAX_BINS = 1024;
// Items per thread are the Policy's choice (Policy::EPT; a tile is THREADS * EPT items): longer
// tiles make longer runs per bin but the staged tile must fit the LDS -- 8 for 16-byte records
// (32 KB staged), 16 for 8-byte pairs (measured: 4096-item tiles are 15 % faster for pairs, no
// better for records).

// tile_start[s] = first tile of segment s, tile_start[n_seg] = tile count.
// seg_end == NULL: segment s ends where s + 1 starts; else segments are slabs with slack behind them.
template <uint32_t TILE>
__device__ __forceinline__ bool tile_of_block(const uint32_t *__restrict__ seg_start,
                                              const uint32_t *__restrict__ tile_start, uint32_t n_seg,
                                              uint32_t &seg, uint32_t &lo, uint32_t &hi,
                                              const uint32_t *__restrict__ seg_end = nullptr)
{
    const uint32_t t = blockIdx.x;
    if (!tile_start) {
        // slab segments of one capacity, no tile table: the grid is n_seg x (tiles a full slab has), a workgroup
        // whose tile lies behind its slab's cursor leaves at once. Worth it where slabs usually hold several tiles
        // (the collapse: 3 of 4; no single-workgroup kernel and its hand-overs between the pack and level 2); where
        // a second tile is rare (the search: 8192 empty workgroups, each holding 52 KB of LDS for ~2 us) it cost
        // 0.06 ms, and one workgroup per slab looping over its tiles doubled the registers of this function.
        const uint32_t per = gridDim.x / n_seg;
        seg = t / per;
        lo = seg_start[seg] + (t - seg * per) * TILE;
        hi = seg_start[seg + 1];
        if (seg_end)
            hi = min(hi, seg_end[seg]);
        if (lo >= hi)
            return false;
        hi = min(lo + TILE, hi);
        return true;
    }
    if (t >= tile_start[n_seg])
        return false;
    // (a table with every tile's segment instead of this search -- 13 dependent scalar loads over 8192 slabs --
    // changed nothing: 0.435 ms either way for level 2 at config 3)
    uint32_t a = 0, b = n_seg;  // last segment with tile_start <= t
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (tile_start[m] <= t)
            a = m;
        else
            b = m;
    }
    seg = a;
    lo = seg_start[a] + (t - tile_start[a]) * TILE;
    hi = seg_start[a + 1];
    if (seg_end)
        hi = min(hi, seg_end[a]);        // (a cursor that ran past its slab's end: the slab is full)
    hi = min(lo + TILE, hi);
    return true;
}

template <class Policy, bool LEVEL1>
__device__ __forceinline__ void hist_body(const typename Policy::Source &src, const uint32_t *__restrict__ seg_start,
                                          const uint32_t *__restrict__ tile_start, uint32_t n_seg, uint32_t shift,
                                          uint32_t n_bins, uint32_t *__restrict__ hist)
{
    constexpr uint32_t EPT = Policy::EPT, TILE = THREADS * EPT;
    __shared__ uint32_t s_hist[MAX_BINS];
    uint32_t seg, lo, hi;
    if (!tile_of_block<TILE>(seg_start, tile_start, n_seg, seg, lo, hi))
        return;
    for (uint32_t b = threadIdx.x; b < n_bins; b += THREADS)
        s_hist[b] = 0;
    uint32_t h[EPT];
    {
        typename Policy::KeyRaw raw[EPT];
#pragma unroll
        for (uint32_t e = 0; e < EPT; e++)   // all loads in flight before the first LDS atomic (clamped, not conditional)
            raw[e] = Policy::template key_fetch<LEVEL1>(src, min(lo + e * THREADS + threadIdx.x, hi - 1));
#pragma unroll
        for (uint32_t e = 0; e < EPT; e++)
            h[e] = Policy::template key_finish<LEVEL1>(src, min(lo + e * THREADS + threadIdx.x, hi - 1), raw[e]);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < EPT; e++)
        if (lo + e * THREADS + threadIdx.x < hi)
            atomicAdd(&s_hist[(h[e] >> shift) & (n_bins - 1)], 1u);
    __syncthreads();
    if (LEVEL1) {
        const uint32_t n_tiles = tile_start[n_seg];
        for (uint32_t b = threadIdx.x; b < n_bins; b += THREADS)
            hist[(size_t)b * n_tiles + blockIdx.x] = s_hist[b];
    } else {
        for (uint32_t b = threadIdx.x; b < n_bins; b += THREADS)
            if (s_hist[b])
                atomicAdd(&hist[seg * n_bins + b], s_hist[b]);
    }
}

// LEVEL1: cursor = inclusive scan of the (bin x tile) count matrix. Level 2: cursor[seg * n_bins + b]
// = next free position of that bucket (one atomic per (tile, bin)).
template <class Policy, bool LEVEL1, uint32_t MAXB = MAX_BINS, uint32_t NT = 1>
__device__ __forceinline__ void scatter_body(const typename Policy::Source &src,
                                             const uint32_t *__restrict__ seg_start,
                                             const uint32_t *__restrict__ tile_start, uint32_t n_seg, uint32_t shift,
                                             uint32_t n_bins, uint32_t *__restrict__ cursor,
                                             typename Policy::Item *__restrict__ out, uint32_t slab_cap = 0,
                                             uint32_t *__restrict__ slab_overflow = nullptr,
                                             const uint32_t *__restrict__ seg_end = nullptr, uint32_t seg_shift = 0,
                                             uint32_t seg_mask = 0xFFFFFFFFu, uint32_t l1_subs = 0)
{
    // l1_subs != 0 (LEVEL1 only; a power of two): level 1 in slab mode too, without the count matrix and its
    // histogram pass -- tile t adds to sub-part t % l1_subs of every bin: part (sub, bin) owns
    // out[(sub * n_bins + bin) * slab_cap, +slab_cap), cursor[sub * n_bins + bin] started at its first slot; the
    // parts are the slab segments of level 2 (part -> bin: segment & (n_bins - 1), its seg_mask).
    // seg_end / seg_shift (level 2 behind the fused pack, pack.hip): the input segments are slabs
    // [seg_start[s], seg_end[s]) and 2^seg_shift consecutive segments are sub-parts of ONE level-1
    // part -- they feed the same buckets. seg_mask: slabs received from several ranks come sender by
    // sender, each sender's in part order: part = (segment >> seg_shift) & seg_mask. Policy::segment_tag
    // / apply_tag mark every loaded item with something derived from its (unshifted) segment.
    // slab_cap != 0 (level 2 only): bucket k owns out[k * slab_cap, (k + 1) * slab_cap) and its cursor
    // started at k * slab_cap -- no histogram pass told us how full it gets. Items that would land
    // behind the slab's end are dropped and *slab_overflow gets bit 1: the caller redoes the level
    // exactly. What IS written stays gap-free ([start, min(cursor, end)) holds items of this run and
    // nothing else): the consumers of an overflowed attempt still run before the flag is read, and
    // stale bytes taken for (hash, uid) items or read indices would send them out of bounds.
    using Item = typename Policy::Item;
    constexpr uint32_t EPT = Policy::EPT, TILE = THREADS * EPT;
    // Policy::ROUNDS > 1: the tile is held in registers and leaves through a staging area of TILE / ROUNDS items,
    // ROUNDS ranges of sorted positions one after the other -- a tile twice as long (half the cursor atomics, runs
    // twice as long) without twice the LDS
    constexpr uint32_t ROUNDS = Policy::ROUNDS, STAGE = TILE / ROUNDS;
    static_assert(TILE % ROUNDS == 0 && EPT % ROUNDS == 0, "rounds must divide the tile");
    // MAXB bounds n_bins (the three bin tables): 256 instead of 1024 is one more workgroup per CU.
    // (Halving the staging area as well -- two phases, 21 KB, twice the workgroups -- changed
    // nothing: the kernel is bound by the write efficiency of its 128-byte runs, not by occupancy.)
    __shared__ uint32_t s_hist[MAXB];   // tile count per bin
    __shared__ uint32_t s_off[MAXB];    // first tile-local position of the bin
    __shared__ uint32_t s_base[MAXB];   // global position of the bin's run, minus s_off
    __shared__ uint32_t s_wave[THREADS / 64];
    __shared__ Item s_stage[STAGE];
    __shared__ uint16_t s_stage_bin[STAGE];
    __shared__ typename Policy::Shared s_policy;
    uint32_t seg, lo, hi;
    if (!tile_of_block<TILE * NT>(seg_start, tile_start, n_seg, seg, lo, hi, seg_end))
        return;
    Policy::init_shared(s_policy, threadIdx.x);       // (visible after the barrier behind the loads)
    const uint32_t seg_tag = Policy::segment_tag(src, seg);       // (per workgroup: a tile lies in ONE segment)
    seg = (seg >> seg_shift) & seg_mask;
    const bool matrix = LEVEL1 && !l1_subs;
    if (LEVEL1 && l1_subs)
        seg = blockIdx.x & (l1_subs - 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // NT tiles per workgroup (NT = 2: the collapse's level 2): the loads of ALL of them are requested up front, so
    // the second tile's bytes travel while the first one goes through its LDS phases and its stores -- a workgroup
    // that loads, sorts and stores one tile has nothing in flight two thirds of the time. Both tiles lie in one
    // segment: same bins, same cursors.
    const uint32_t all_lo = lo, all_hi = hi;
    typename Policy::Raw raw_all[NT][EPT];
#pragma unroll
    for (uint32_t k = 0; k < NT; k++)
#pragma unroll
        for (uint32_t e = 0; e < EPT; e++)      // every load in flight: clamped indices, no branch
            raw_all[k][e] = Policy::template fetch<LEVEL1>(src, min(all_lo + (k * EPT + e) * THREADS + threadIdx.x, all_hi - 1));
#pragma unroll
  for (uint32_t tile_k = 0; tile_k < NT; tile_k++) {
    lo = all_lo + tile_k * TILE;
    if (lo >= all_hi)
        break;                                   // (the same for every thread of the workgroup)
    hi = min(lo + TILE, all_hi);
    if (tile_k)
        __syncthreads();                         // (the tables of the tile before have been read)
    for (uint32_t b = tid; b < n_bins; b += THREADS)
        s_hist[b] = 0;
    Item v[EPT];
    uint32_t h[EPT], bin[EPT], rank[EPT];
    {
#pragma unroll
        for (uint32_t e = 0; e < EPT; e++) {
            const uint32_t i = lo + e * THREADS + tid;
            h[e] = Policy::template finish<LEVEL1>(src, min(i, hi - 1), raw_all[tile_k][e], v[e], i < hi, seg_tag, s_policy);
            Policy::apply_tag(v[e], seg_tag);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < EPT; e++) {
        bin[e] = 0xFFFFFFFFu;
        if (lo + e * THREADS + tid < hi && !(Policy::MAY_SKIP && Policy::skip(v[e]))) {
            bin[e] = (h[e] >> shift) & (n_bins - 1);
            rank[e] = atomicAdd(&s_hist[bin[e]], 1u);   // position inside the tile's share of the bin
        }
    }
    __syncthreads();
    // exclusive scan of the bin counts (each thread owns bpt consecutive bins) + global bases
    // (reserving bins 2k and 2k + 1 with ONE 64-bit add on their adjacent cursors -- half the cursor atomics --
    // changed nothing: 0.42-0.46 ms for level 2 of the collapse, 0.29 for the search's two levels, either way)
    const uint32_t bpt = (n_bins + THREADS - 1) / THREADS;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < bpt; k++) {
        const uint32_t b = tid * bpt + k;
        mine += b < n_bins ? s_hist[b] : 0u;
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
    uint32_t run = incl - mine;
    for (uint32_t wv = 0; wv < wave; wv++)
        run += s_wave[wv];
    uint32_t staged = 0;                                  // items of the tile that go out (all, unless the Policy skips some)
    for (uint32_t wv = 0; wv < THREADS / 64; wv++)
        staged += s_wave[wv];
    // The cursor reservations (one global atomic per (tile, bin), ~2 us until the answer is back) are ISSUED here
    // and consumed only after the tile has been staged: staging needs the tile-local offsets alone, so the
    // round trip runs under it instead of in front of it.
    constexpr uint32_t BPT_MAX = (MAXB + THREADS - 1) / THREADS;
    uint32_t g_base[BPT_MAX], g_cnt[BPT_MAX], g_run[BPT_MAX];
#pragma unroll
    for (uint32_t k = 0; k < BPT_MAX; k++) {
        const uint32_t b = tid * bpt + k;
        g_cnt[k] = 0;
        g_base[k] = 0;
        g_run[k] = run;
        if (k < bpt && b < n_bins) {
            const uint32_t c = s_hist[b];
            s_off[b] = run;
            g_cnt[k] = c;
            if (matrix)
                g_base[k] = cursor[(size_t)b * tile_start[n_seg] + blockIdx.x] - c;
            else
                g_base[k] = c ? atomicAdd(&cursor[seg * n_bins + b], c) : 0u;
            run += c;
        }
    }
    __syncthreads();
    const uint32_t count = Policy::MAY_SKIP ? staged : hi - lo;
    auto stage_round = [&](uint32_t r0) {
#pragma unroll
        for (uint32_t e = 0; e < EPT; e++)
            if (bin[e] != 0xFFFFFFFFu) {
                const uint32_t p = s_off[bin[e]] + rank[e] - r0;
                if (ROUNDS == 1 || p < STAGE) {
                    s_stage[p] = v[e];
                    s_stage_bin[p] = (uint16_t)bin[e];
                }
            }
    };
    stage_round(0);
#pragma unroll
    for (uint32_t k = 0; k < BPT_MAX; k++) {
        const uint32_t b = tid * bpt + k;
        if (k < bpt && b < n_bins) {
            s_base[b] = g_base[k] - g_run[k];
            if (!matrix && slab_cap && g_cnt[k]) {
                const uint64_t end = ((uint64_t)seg * n_bins + b + 1) * slab_cap;
                if ((uint64_t)g_base[k] + g_cnt[k] > end) {
                    if constexpr (Policy::CAN_SPILL) {
                        // the items behind the slab's end go to the Policy's spill list: position pos to
                        // spill[s_hist[bin] + pos] (the bin's count is in g_cnt by now: its word is free)
                        const uint32_t first = (uint32_t)max((uint64_t)g_base[k], end);
                        s_hist[b] = Policy::spill_reserve(src, g_base[k] + g_cnt[k] - first) - first;
                    } else {
                        atomicOr(slab_overflow, 2u);
                    }
                }
            }
        }
    }
    for (uint32_t r0 = 0; r0 < (ROUNDS > 1 ? count : 1u); r0 += STAGE) {
        if (r0)
            stage_round(r0);
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < EPT / ROUNDS; e++) {
            const uint32_t p = e * THREADS + tid;
            if (r0 + p < count) {
                const uint32_t bn = s_stage_bin[p];
                const uint32_t pos = s_base[bn] + r0 + p;   // consecutive p of one bin: consecutive addresses
                if (matrix || !slab_cap || pos < (seg * n_bins + bn + 1) * slab_cap)
                    out[pos] = s_stage[p];
                else if constexpr (Policy::CAN_SPILL)
                    Policy::spill_write(src, s_hist[bn] + pos, s_stage[p]);
            }
        }
        if (ROUNDS > 1)
            __syncthreads();
    }
  }
    Policy::flush(src, s_policy, tid);
}

// tile_start[] for the segments seg_start[0..n_seg] (single block; n_seg <= MAX_BINS)
template <uint32_t TILE>
__device__ __forceinline__ void tile_starts_body(const uint32_t *__restrict__ seg_start, uint32_t n_seg,
                                                 uint32_t *__restrict__ tile_start)
{
    __shared__ uint32_t s[MAX_BINS + 1];
    for (uint32_t t = threadIdx.x; t < n_seg; t += blockDim.x)
        s[t] = (seg_start[t + 1] - seg_start[t] + TILE - 1) / TILE;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < n_seg; i++) {
            const uint32_t c = s[i];
            s[i] = acc;
            acc += c;
        }
        s[n_seg] = acc;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t <= n_seg; t += blockDim.x)
        tile_start[t] = s[t];
}

// the same for any number of slab segments [seg_start[s], min(seg_end[s], seg_start[s + 1])) (single block of
// 1024 threads; seg_start has n_seg + 1 entries). Up to 8192 segments -- the fused pack's and the search's level
// 1 -- the tile counts are first fetched into LDS with coalesced, independent loads (thread t takes segments
// t, t + 1024, ...): a thread walking its own 8 consecutive segments waited for 8 dependent round trips, 13.7 us
// for the whole kernel, on the critical path between two partition levels.
template <uint32_t TILE>
__device__ __forceinline__ void slab_tile_starts_body(const uint32_t *__restrict__ seg_start,
                                                      const uint32_t *__restrict__ seg_end, uint32_t n_seg,
                                                      uint32_t *__restrict__ tile_start)
{
    constexpr uint32_t STAGED = 8192;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_tiles[STAGED];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (n_seg + blockDim.x - 1) / blockDim.x;
    const uint32_t s0 = tid * per, s1 = min(s0 + per, n_seg);
    auto tiles_of = [&](uint32_t t) { return (min(seg_end[t], seg_start[t + 1]) - seg_start[t] + TILE - 1) / TILE; };
    const bool staged = n_seg <= STAGED;
    if (staged) {
#pragma unroll
        for (uint32_t k = 0; k < STAGED / 1024; k++) {
            const uint32_t t = k * 1024 + tid;
            if (t < n_seg && blockDim.x == 1024)
                s_tiles[t] = tiles_of(t);
        }
        if (blockDim.x != 1024)
            for (uint32_t t = tid; t < n_seg; t += blockDim.x)
                s_tiles[t] = tiles_of(t);
        __syncthreads();
    }
    uint32_t mine = 0;
    for (uint32_t t = s0; t < s1; t++)
        mine += staged ? s_tiles[t] : tiles_of(t);
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t wv = 0; wv < wave; wv++)
        run += s_wave[wv];
    for (uint32_t t = s0; t < s1; t++) {
        tile_start[t] = run;
        run += staged ? s_tiles[t] : tiles_of(t);
    }
    if (s1 == n_seg && s0 < n_seg)
        tile_start[n_seg] = run;
}

// level 1: part p starts where the scan of the (bin x tile) matrix stood before row p
__device__ __forceinline__ void matrix_starts_body(const uint32_t *__restrict__ matrix_incl, uint32_t n_bins,
                                                   uint32_t n_tiles, uint32_t *__restrict__ start)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= n_bins)
        start[b] = b ? matrix_incl[(size_t)b * n_tiles - 1] : 0u;
}

// bucket_start[b] = items in buckets < b (from the inclusive scan of the level-2 histogram); the
// level-2 cursors start there
__device__ __forceinline__ void bucket_starts_body(const uint32_t *__restrict__ hist_incl, uint32_t n_buckets,
                                                   uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ cursor)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets)
        return;
    const uint32_t v = b ? hist_incl[b - 1] : 0u;
    bucket_start[b] = v;
    if (b < n_buckets)
        cursor[b] = v;
}

}  // namespace fqd_partition
