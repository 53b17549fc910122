// collapse_pairs.hip -- exact-duplicate collapse for records LONGER than one uint4 (keys above
// 32 nt: BASELINE configs 2, 4, 5), without a device-wide sort. Same contract as collapse.hip and
// collapse_lds.hip (reference _triemodule.c:235-239, :261-264: an identical key bumps a count;
// pass-2 rule __init__.py:201-206: the first holder is remembered).
//
// The records stay where the pack kernel put them. What moves is one (key hash, read position)
// pair per read: partitioned by the top hash bits into buckets of ~400-800 pairs (the two-level
// LDS-aggregated partition of group.hip / partition.cuh), then one workgroup per bucket runs the
// pairs through an LDS hash table keyed by the 32-bit hash. A slot is claimed with an LDS
// compare-and-swap by the first read of a hash, which parks its POSITION; every later read with
// that hash has its record compared with the parked read's record in HBM (a hash only proposes;
// the comparisons of a wave are queued in LDS and done by 64 / Q groups of Q lanes, one uint4 of
// either record per lane, so that every request is a whole record line), and then bumps the
// slot's count / min position with LDS atomics or probes on. Per
// read: 8 bytes through the partition twice and, for a read that is not the first of its key, two
// record gathers (its own and the parked one, which the reads before it have pulled into L2).
// bucket_pairs_compact_kernel then gathers one record per unique key into the unique table.
#include <algorithm>
#include <cstdlib>
#include "fqd_internal.h"

namespace {

constexpr uint32_t PD_SLOTS = 1024;        // LDS table slots per bucket (power of two)
constexpr uint32_t PD_THREADS = 256;
constexpr uint32_t PD_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t PD_AHEAD = 4;

template <bool RAGGED>
__global__ __launch_bounds__(PD_THREADS) void bucket_pairs_dedupe_kernel(
    const uint2 *__restrict__ items /* (hash, position) */, const uint32_t *__restrict__ bucket_start,
    const uint32_t *__restrict__ bucket_end /* NULL, or slab mode: where each bucket's cursor stopped */,
    const uint4 *__restrict__ recs4, uint32_t q_per_rec, const uint32_t *__restrict__ weights,
    uint32_t *__restrict__ tmp_rep, uint32_t *__restrict__ tmp_count, uint32_t *__restrict__ tmp_first,
    uint32_t *__restrict__ bucket_unique, uint32_t *__restrict__ overflow,
    uint32_t tag_mask /* ~0; tests: few bits => different keys share a tag */,
    const uint32_t *__restrict__ lens /* ragged keys: equal records of different length are different keys (the
                                         bases past a key's end hold code 0, like its first symbol might); else NULL */,
    fqd::PairsSlices sl /* slice != 0: a bucket's first `slice` items only; workgroups past n_buckets take the further
                           SLICES of longer buckets (pairs_slices_kernel), pairs_merge_kernel joins their rows */,
    uint32_t n_buckets)
{
    __shared__ uint32_t s_tag[PD_SLOTS], s_rep[PD_SLOTS], s_cnt[PD_SLOTS], s_min[PD_SLOTS];
    __shared__ uint32_t s_wave_tot[PD_THREADS / 64];
    // comparisons queued by a wave in a round: (position, parked position) in, "differs" out
    __shared__ uint32_t s_qa[PD_THREADS / 64][64 * PD_AHEAD], s_qb[PD_THREADS / 64][64 * PD_AHEAD];
    __shared__ uint8_t s_qdiff[PD_THREADS / 64][64 * PD_AHEAD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t groups = 64u / q_per_rec;            // comparisons a wave does at once (q_per_rec <= 64)
    const uint32_t gl = lane / q_per_rec, ql = lane - gl * q_per_rec;
    const uint32_t b = blockIdx.x;
    uint32_t lo, hi;
    bool sliced = b >= n_buckets;
    if (sliced) {
        if (b - n_buckets >= sl.ctr[1])
            return;
        const uint2 piece = sl.extra[b - n_buckets];
        lo = piece.x;
        hi = piece.y;
    } else {
        lo = bucket_start[b];
        hi = bucket_start[b + 1];
        if (bucket_end)
            hi = min(hi, bucket_end[b]);
        if (sl.slice && hi - lo > sl.slice) {
            hi = lo + sl.slice;
            sliced = true;
        }
    }
    for (uint32_t s = tid; s < PD_SLOTS; s += PD_THREADS)
        s_tag[s] = PD_EMPTY;
    __syncthreads();

    bool full = false;
    for (uint32_t base0 = lo; base0 < hi; base0 += PD_AHEAD * PD_THREADS) {
        uint2 it[PD_AHEAD];
        uint32_t w[PD_AHEAD], tag[PD_AHEAD], slot[PD_AHEAD], probes[PD_AHEAD];
        bool pending[PD_AHEAD];
#pragma unroll
        for (uint32_t k = 0; k < PD_AHEAD; k++) {
            const uint32_t i = base0 + k * PD_THREADS + tid;
            it[k] = make_uint2(0, 0);
            if (i < hi)
                it[k] = items[i];
        }
#pragma unroll
        for (uint32_t k = 0; k < PD_AHEAD; k++) {
            const uint32_t i = base0 + k * PD_THREADS + tid;
            pending[k] = i < hi;
            w[k] = pending[k] ? (weights ? weights[it[k].y] : 1u) : 0u;
            tag[k] = (it[k].x == PD_EMPTY ? 0u : it[k].x) & tag_mask;
            slot[k] = (tag[k] * 0x9E3779B1u) >> 22;    // top 10 bits of a re-mix: PD_SLOTS == 1024
            probes[k] = 0;
        }
        // a round: claim an empty slot or stop at a slot whose tag matches, for every pending pair of
        // the thread; barrier (the parked positions are visible); verify against the parked read's
        // record and count, or probe on
        bool any;
        do {
#pragma unroll
            for (uint32_t k = 0; k < PD_AHEAD; k++) {
                if (!pending[k])
                    continue;
                for (;;) {
                    const uint32_t old = atomicCAS(&s_tag[slot[k]], PD_EMPTY, tag[k]);
                    if (old == PD_EMPTY) {
                        s_rep[slot[k]] = it[k].y;
                        s_cnt[slot[k]] = w[k];
                        s_min[slot[k]] = it[k].y;
                        pending[k] = false;
                        break;
                    }
                    if (old == tag[k])
                        break;
                    slot[k] = (slot[k] + 1) & (PD_SLOTS - 1);
                    if (++probes[k] >= PD_SLOTS) {
                        full = true;
                        pending[k] = false;
                        break;
                    }
                }
            }
            __syncthreads();
            // queue this wave's comparisons ...
            uint32_t qn = 0, qi[PD_AHEAD];
#pragma unroll
            for (uint32_t k = 0; k < PD_AHEAD; k++) {
                const unsigned long long m = __ballot(pending[k]);
                qi[k] = qn + __popcll(m & fqd_lanemask_lt());
                if (pending[k]) {
                    s_qa[wave][qi[k]] = it[k].y;
                    s_qb[wave][qi[k]] = s_rep[slot[k]];
                }
                qn += __popcll(m);
            }
            __syncthreads();
            // ... do them with q_per_rec lanes each, four rounds of loads in flight together ...
            for (uint32_t base = 0; base < qn; base += 4 * groups) {
                uint4 xa[4], xb[4];
                uint32_t la[4], lb[4];
                bool valid[4];
#pragma unroll
                for (uint32_t t = 0; t < 4; t++) {
                    // (clamped and unconditional: "if (valid) load" compiles to a branch with its own wait per round,
                    // and the four rounds' gathers went out one after the other)
                    const uint32_t e = base + t * groups + gl;
                    valid[t] = gl < groups && e < qn;
                    const uint32_t ec = min(base + t * groups + min(gl, groups - 1), qn - 1);
                    xa[t] = recs4[(size_t)s_qa[wave][ec] * q_per_rec + ql];
                    xb[t] = recs4[(size_t)s_qb[wave][ec] * q_per_rec + ql];
                }
                if (RAGGED) {
                    // (ragged keys: the two lengths of a comparison, requested WITH its records -- behind them, under a
                    // branch of their own, they were a second dependent round trip per round: the ragged dedupe of config 5's
                    // variant took 1.71 ms where keys of one length take 1.14)
#pragma unroll
                    for (uint32_t t = 0; t < 4; t++) {
                        const uint32_t ec = min(base + t * groups + min(gl, groups - 1), qn - 1);
                        la[t] = lens[s_qa[wave][ec]];
                        lb[t] = lens[s_qb[wave][ec]];
                    }
                }
#pragma unroll
                for (uint32_t t = 0; t < 4; t++) {
                    const bool diff = valid[t] && (((xa[t].x ^ xb[t].x) | (xa[t].y ^ xb[t].y) | (xa[t].z ^ xb[t].z) |
                                                    (xa[t].w ^ xb[t].w)) != 0 || (RAGGED && la[t] != lb[t]));
                    const unsigned long long m = __ballot(diff);
                    if (valid[t] && ql == 0)      // the group's lanes are [lane, lane + q_per_rec)
                        s_qdiff[wave][base + t * groups + gl] =
                            (uint8_t)(((m >> lane) & ((q_per_rec < 64 ? (1ull << q_per_rec) : 0ull) - 1ull)) != 0);
                }
            }
            __syncthreads();
            // ... and act on the answers
            any = false;
#pragma unroll
            for (uint32_t k = 0; k < PD_AHEAD; k++) {
                if (!pending[k])
                    continue;
                if (!s_qdiff[wave][qi[k]]) {
                    atomicAdd(&s_cnt[slot[k]], w[k]);
                    atomicMin(&s_min[slot[k]], it[k].y);
                    pending[k] = false;
                } else {  // same hash, different key: keep probing
                    slot[k] = (slot[k] + 1) & (PD_SLOTS - 1);
                    if (++probes[k] >= PD_SLOTS) {
                        full = true;
                        pending[k] = false;
                    } else {
                        any = true;
                    }
                }
            }
        } while (__syncthreads_or(any));
    }
    if (full)
        atomicOr(overflow, 1u);
    __syncthreads();

    // live slots (count > 0: a key all of whose holders have weight 0 is not in the trie; a SLICE keeps such rows -- their
    // first position counts if another slice holds the key with a weight -- and the merge drops them) -> tmp[lo ...)
    uint32_t mine = 0;
    for (uint32_t s = tid; s < PD_SLOTS; s += PD_THREADS)
        mine += (s_tag[s] != PD_EMPTY && (s_cnt[s] > 0 || sliced)) ? 1u : 0u;
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t wv = 0; wv < wave; wv++)
        before += s_wave_tot[wv];
    uint32_t total = 0;
    for (uint32_t wv = 0; wv < PD_THREADS / 64; wv++)
        total += s_wave_tot[wv];
    uint32_t out = lo + before;
    for (uint32_t s = tid; s < PD_SLOTS; s += PD_THREADS)
        if (s_tag[s] != PD_EMPTY && (s_cnt[s] > 0 || sliced)) {
            tmp_rep[out] = s_rep[s];
            tmp_count[out] = s_cnt[s];
            tmp_first[out] = s_min[s];
            if (sliced)                     // (the slice's items are done with: a row's tag rides where they were)
                sl.tags[(size_t)out * 2] = s_tag[s];
            out++;
        }
    if (tid == 0)
        (b >= n_buckets ? sl.extra_unique[b - n_buckets] : bucket_unique[b]) = total;
}

// ---- a bucket of very many pairs (ONE key with half a million copies: the skewed model's hot molecule) -------------
// One workgroup walked it alone -- 500 K record comparisons at what 4 waves keep in flight: 5.4 ms of a 7 ms job. Such a
// bucket is cut into SLICES of PD_SLICE items, each a workgroup of the dedupe kernel above with its own LDS table and its
// rows where its items were; this kernel then joins the rows of a bucket's slices (a few hundred: the other keys of the
// bucket + one row of the hot key per slice) through one more LDS table, keyed by (tag, parked position) in ONE 64-bit
// word so that a claim and its position appear together, and leaves them where the compaction expects them.
constexpr uint32_t PD_SLICE = 4096;

__global__ void pairs_slices_kernel(const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ bucket_end,
                                    uint32_t n_buckets, fqd::PairsSlices sl)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets)
        return;
    const uint32_t lo = bucket_start[b];
    uint32_t hi = bucket_start[b + 1];
    if (bucket_end)
        hi = min(hi, bucket_end[b]);
    if (hi - lo <= sl.slice)
        return;
    const uint32_t k = (hi - lo + sl.slice - 1) / sl.slice - 1;        // slices beyond the first
    const uint32_t g = atomicAdd(&sl.ctr[0], 1u), base = atomicAdd(&sl.ctr[1], k);     // (sum of k <= items / slice <= cap)
    sl.big[g * 3] = b;
    sl.big[g * 3 + 1] = base;
    sl.big[g * 3 + 2] = k;
    for (uint32_t t = 0; t < k; t++)
        sl.extra[base + t] = make_uint2(lo + (t + 1) * sl.slice, min(hi, lo + (t + 2) * sl.slice));
}

__global__ __launch_bounds__(PD_THREADS) void pairs_merge_kernel(
    const uint32_t *__restrict__ bucket_start, const uint4 *__restrict__ recs4, uint32_t q_per_rec,
    uint32_t *__restrict__ tmp_rep, uint32_t *__restrict__ tmp_count, uint32_t *__restrict__ tmp_first,
    uint32_t *__restrict__ bucket_unique, uint32_t *__restrict__ overflow, const uint32_t *__restrict__ lens,
    fqd::PairsSlices sl)
{
    constexpr uint32_t PM_CHUNK = 1024;
    __shared__ unsigned long long s_key[PD_SLOTS];
    __shared__ uint32_t s_cnt[PD_SLOTS], s_min[PD_SLOTS];
    __shared__ uint32_t s_slo[PM_CHUNK], s_row0[PM_CHUNK + 1];
    __shared__ uint32_t s_wave_tot[PD_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_big = sl.ctr[0];
    for (uint32_t g = blockIdx.x; g < n_big; g += gridDim.x) {
        const uint32_t b = sl.big[g * 3], base = sl.big[g * 3 + 1], k = sl.big[g * 3 + 2], lo = bucket_start[b];
        __syncthreads();
        for (uint32_t s = tid; s < PD_SLOTS; s += PD_THREADS) {
            s_key[s] = ~0ull;
            s_cnt[s] = 0;
            s_min[s] = 0xFFFFFFFFu;
        }
        __syncthreads();
        bool full = false;
        // the slices' rows as one list (a slice leaves a handful of rows; walking the slices one after the other was a
        // chain of dependent round trips per slice: 0.28 ms for the 122 slices of a key with 500 K copies)
        for (uint32_t t0 = 0; t0 <= k; t0 += PM_CHUNK) {
            const uint32_t nt = min(PM_CHUNK, k + 1 - t0);
            __syncthreads();
            for (uint32_t x = tid; x < nt; x += PD_THREADS) {
                const uint32_t t = t0 + x;
                s_slo[x] = t ? sl.extra[base + t - 1].x : lo;
                s_row0[x] = t ? sl.extra_unique[base + t - 1] : bucket_unique[b];
            }
            __syncthreads();
            if (wave == 0) {                       // exclusive prefix of the row counts (nt <= PM_CHUNK, 64 lanes)
                uint32_t run = 0;
                for (uint32_t x0 = 0; x0 < nt; x0 += 64) {
                    const uint32_t v = x0 + lane < nt ? s_row0[x0 + lane] : 0u;
                    uint32_t inc = v;
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t up = __shfl_up(inc, o);
                        if ((int)lane >= o)
                            inc += up;
                    }
                    if (x0 + lane < nt)
                        s_row0[x0 + lane] = run + inc - v;
                    run += __shfl(inc, 63);
                }
                if (lane == 0)
                    s_row0[nt] = run;
            }
            __syncthreads();
            const uint32_t n_rows = s_row0[nt];
            for (uint32_t r = tid; r < n_rows; r += PD_THREADS) {
                uint32_t a = 0, z = nt;
                while (z - a > 1) {
                    const uint32_t mid = (a + z) >> 1;
                    if (s_row0[mid] <= r)
                        a = mid;
                    else
                        z = mid;
                }
                const uint32_t at = s_slo[a] + (r - s_row0[a]);
                const uint32_t tag = sl.tags[(size_t)at * 2], rep = tmp_rep[at];
                const uint32_t cnt = tmp_count[at], first = tmp_first[at];
                uint32_t slot = (tag * 0x9E3779B1u) >> 22, probes = 0;
                for (;;) {
                    const unsigned long long mine = (unsigned long long)tag << 32 | rep;
                    const unsigned long long old = atomicCAS(&s_key[slot], ~0ull, mine);
                    bool same = old == ~0ull;
                    if (!same && (uint32_t)(old >> 32) == tag) {
                        const uint32_t park = (uint32_t)old;
                        same = !lens || lens[park] == lens[rep];
                        for (uint32_t q = 0; q < q_per_rec && same; q++) {
                            const uint4 x = recs4[(size_t)rep * q_per_rec + q], y = recs4[(size_t)park * q_per_rec + q];
                            same = ((x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w)) == 0;
                        }
                    }
                    if (same) {
                        atomicAdd(&s_cnt[slot], cnt);
                        atomicMin(&s_min[slot], first);
                        break;
                    }
                    slot = (slot + 1) & (PD_SLOTS - 1);
                    if (++probes >= PD_SLOTS) {
                        full = true;
                        break;
                    }
                }
            }
        }
        if (full)
            atomicOr(overflow, 1u);
        __syncthreads();                         // (every row of every slice has been read: the bucket's rows may be overwritten)
        uint32_t mine = 0;
        for (uint32_t s = tid; s < PD_SLOTS; s += PD_THREADS)
            mine += (s_key[s] != ~0ull && s_cnt[s] > 0) ? 1u : 0u;
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += up;
        }
        if (lane == 63)
            s_wave_tot[wave] = incl;
        __syncthreads();
        uint32_t out = lo + incl - mine, total = 0;
        for (uint32_t wv = 0; wv < PD_THREADS / 64; wv++) {
            out += wv < wave ? s_wave_tot[wv] : 0u;
            total += s_wave_tot[wv];
        }
        for (uint32_t s = tid; s < PD_SLOTS; s += PD_THREADS)
            if (s_key[s] != ~0ull && s_cnt[s] > 0) {
                tmp_rep[out] = (uint32_t)s_key[s];
                tmp_count[out] = s_cnt[s];
                tmp_first[out] = s_min[s];
                out++;
            }
        if (tid == 0)
            bucket_unique[b] = total;
    }
}

// one wave per bucket: the record at tmp_rep[bucket_start[b] + j] -> urecs[unique offset of b + j]
__global__ __launch_bounds__(256) void bucket_pairs_compact_kernel(
    const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ unique_incl /* inclusive scan */,
    uint32_t n_buckets, const uint32_t *__restrict__ tmp_rep, const uint32_t *__restrict__ tmp_count,
    const uint32_t *__restrict__ tmp_first, const uint4 *__restrict__ recs4, uint32_t q_per_rec, IdSource read_ids,
    uint4 *__restrict__ urecs4, uint32_t *__restrict__ ucounts, uint64_t *__restrict__ ufirst,
    fqd::SegHashOut sho /* nseg != 0 (q_per_rec a power of two): also the segment hashes of the search that follows */,
    const uint32_t *__restrict__ lens, uint32_t *__restrict__ ulens /* ragged keys: their lengths travel too */,
    uint32_t row_cap /* the unique table has room for this many rows: a launch queued BEFORE the host knows the number of
                        unique keys (api.hip collapse_pairs) writes nothing at all when they are more */,
    uint32_t len_hint /* != 0 (ragged keys): a record's last word IS its key's length (pack.hip): no gathers out of lens[];
                         the value: a likely length, for the segment masks worked out once per wave */)
{
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_buckets || unique_incl[n_buckets - 1] > row_cap)
        return;
    const uint32_t end = unique_incl[b], begin = b ? unique_incl[b - 1] : 0u;
    const uint32_t src = bucket_start[b], cnt = end - begin;
    const bool len_in_rec = lens && len_hint;
    for (uint32_t j = fqd_lane(); j < cnt; j += 64) {
        ucounts[begin + j] = tmp_count[src + j];
        ufirst[begin + j] = read_ids.at(tmp_first[src + j]);
        if (lens && !len_in_rec)
            ulens[begin + j] = lens[tmp_rep[src + j]];
    }
    const uint32_t n_unique = unique_incl[n_buckets - 1];
    // four record quarters per lane and step: the four position loads go out together, then the four record
    // gathers (one after the other, every step waited for a chain of two dependent round trips -- 25-50 steps per
    // wave: 1.3 ms for 12.9 M records of 128 bytes, config 4)
    // (clamped indices, unconditional loads: "x < total ? load : 0" compiles to a branch per load with a wait inside,
    // and the CP gathers then go out one after the other -- DESIGN.md section 0, item 2)
#ifndef FQD_PAIRS_COMPACT_CP
#define FQD_PAIRS_COMPACT_CP 4
#endif
    constexpr uint32_t CP = FQD_PAIRS_COMPACT_CP;
    const uint32_t total = cnt * q_per_rec;
    // q_per_rec a power of two (always when hashes are asked for): x / q and x % q are a shift and a mask, and a lane
    // keeps its quarter q across its steps (64 is a multiple of q_per_rec) -- for keys of one length the segment masks
    // of its four words are computed once per wave, not once per record quarter
    const bool pow2 = (q_per_rec & (q_per_rec - 1)) == 0;
    const uint32_t q_shift = pow2 ? (uint32_t)__ffs((int)q_per_rec) - 1u : 0u;
    constexpr uint32_t MAX_SEG = 4;
    uint32_t seg_mask[MAX_SEG][4];
    // (ragged keys: the masks of the bucket's first key's length -- nearly every key of a FASTQ file has the modal
    // length --, a key of another length computes its own)
    const bool fixed_masks = sho.nseg && sho.nseg <= MAX_SEG && pow2 && cnt;
    const uint32_t mask_len = fixed_masks && lens ? (len_in_rec ? len_hint : lens[tmp_rep[src]]) : sho.len;
    if (fixed_masks) {
        const uint32_t q = fqd_lane() & (q_per_rec - 1);
        for (uint32_t sg = 0; sg < MAX_SEG; sg++) {
            uint32_t lo = 0, hi = 0;
            if (sg < sho.nseg)
                fqd_segment(mask_len, sg, sho.nseg, lo, hi);
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                const uint32_t jw = q * 4 + e;
                seg_mask[sg][e] = sg < sho.nseg && jw < sho.kw ? fqd_range_mask(jw / sho.planes, lo, hi) : 0u;
            }
        }
    }
    // (the positions of a step's records are requested a step AHEAD: a step was two dependent round trips -- positions,
    // then the record gathers -- six to twelve steps per bucket)
    uint32_t rep_next[CP];
    auto request_reps = [&](uint32_t x0) {
#pragma unroll
        for (uint32_t t = 0; t < CP; t++) {
            const uint32_t xc = min(x0 + t * 64, total ? total - 1 : 0u);
            rep_next[t] = tmp_rep[src + (pow2 ? xc >> q_shift : xc / q_per_rec)];
        }
    };
    if (total)
        request_reps(fqd_lane());
    for (uint32_t x0 = fqd_lane(); x0 < total; x0 += CP * 64) {
        uint32_t rep[CP], klen_of[CP];
        uint4 v[CP];
#pragma unroll
        for (uint32_t t = 0; t < CP; t++)
            rep[t] = rep_next[t];
        request_reps(x0 + CP * 64);
#pragma unroll
        for (uint32_t t = 0; t < CP; t++) {
            const uint32_t xc = min(x0 + t * 64, total - 1);
            v[t] = recs4[(size_t)rep[t] * q_per_rec + (pow2 ? xc & (q_per_rec - 1) : xc % q_per_rec)];
        }
        if (sho.nseg && len_in_rec) {        // (the length sits in the record's last word: the record's last lane has it)
#pragma unroll
            for (uint32_t t = 0; t < CP; t++)
                klen_of[t] = __shfl(v[t].w, (int)(fqd_lane() | (q_per_rec - 1u)));
        } else if (sho.nseg && lens) {       // (ragged keys: segments of the key's own length; one branch for the CP loads)
#pragma unroll
            for (uint32_t t = 0; t < CP; t++)
                klen_of[t] = lens[rep[t]];
        } else {
#pragma unroll
            for (uint32_t t = 0; t < CP; t++)
                klen_of[t] = sho.len;
        }
#pragma unroll
        for (uint32_t t = 0; t < CP; t++) {
            const uint32_t x = x0 + t * 64;
            if (x >= total)
                continue;              // (a record's q_per_rec lanes stay or leave together: 64 is a multiple of it when hashes are asked for)
            const uint32_t j = pow2 ? x >> q_shift : x / q_per_rec, q = x - j * q_per_rec;
            urecs4[(size_t)(begin + j) * q_per_rec + q] = v[t];
            if (len_in_rec && q == q_per_rec - 1u)
                ulens[begin + j] = v[t].w;
            if (sho.nseg) {
                // as segment_hashes_kernel (edges.hip): the record's q_per_rec lanes sit side by side in
                // the wave (q_per_rec divides 64), each sums its four words' share of a segment
                const uint32_t word[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
                const uint32_t klen = klen_of[t];
                for (uint32_t sg = 0; sg < sho.nseg; sg++) {
                    uint32_t part = 0;
                    if (fixed_masks && klen == mask_len) {
#pragma unroll
                        for (uint32_t e = 0; e < 4; e++) {
                            const uint32_t m = seg_mask[sg < MAX_SEG ? sg : 0][e];
                            if (m)
                                part += fqd_mix32((word[e] & m) + (q * 4 + e + 1u) * 0x9E3779B1u);
                        }
                    } else {
                        uint32_t lo, hi;
                        fqd_segment(klen, sg, sho.nseg, lo, hi);
#pragma unroll
                        for (uint32_t e = 0; e < 4; e++) {
                            const uint32_t jw = q * 4 + e;             // word index in the record
                            if (jw < sho.kw) {
                                const uint32_t m = fqd_range_mask(jw / sho.planes, lo, hi);
                                if (m)
                                    part += fqd_mix32((word[e] & m) + (jw + 1u) * 0x9E3779B1u);
                            }
                        }
                    }
                    for (uint32_t off = 1; off < q_per_rec; off <<= 1) {
                        const uint32_t other = __shfl_down(part, off);
                        if (q + off < q_per_rec)
                            part += other;
                    }
                    if (q == 0)
                        sho.out[(size_t)sg * n_unique + begin + j] =
                            fqd_mix32(part + fqd_mix32(klen * 0x9E3779B1u + sg * 0x85EBCA77u + 0x165667B1u));
                }
            }
        }
    }
}

}  // namespace

namespace fqd {

hipError_t launch_bucket_pairs_dedupe(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                      uint32_t n_buckets, const uint32_t *recs, uint32_t stride_words,
                                      const uint32_t *weights, uint32_t *tmp_rep, uint32_t *tmp_count,
                                      uint32_t *tmp_first, uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st,
                                      const uint32_t *lens, PairsSlices sl)
{
    if (!n_buckets || (stride_words & 3u))
        return n_buckets ? hipErrorInvalidValue : hipSuccess;
    if (sl.slice)
        pairs_slices_kernel<<<(n_buckets + 255) / 256, 256, 0, st>>>(bucket_start, bucket_end, n_buckets, sl);
    const uint32_t tag_mask = getenv("FQD_PAIRS_TAG_MASK") ? (uint32_t)strtoul(getenv("FQD_PAIRS_TAG_MASK"), nullptr, 0) : 0xFFFFFFFFu;
    if (lens)
        bucket_pairs_dedupe_kernel<true><<<n_buckets + (sl.slice ? sl.cap : 0u), PD_THREADS, 0, st>>>(
            reinterpret_cast<const uint2 *>(items), bucket_start, bucket_end, reinterpret_cast<const uint4 *>(recs),
            stride_words / 4, weights, tmp_rep, tmp_count, tmp_first, bucket_unique, overflow, tag_mask, lens, sl, n_buckets);
    else
        bucket_pairs_dedupe_kernel<false><<<n_buckets + (sl.slice ? sl.cap : 0u), PD_THREADS, 0, st>>>(
            reinterpret_cast<const uint2 *>(items), bucket_start, bucket_end, reinterpret_cast<const uint4 *>(recs),
            stride_words / 4, weights, tmp_rep, tmp_count, tmp_first, bucket_unique, overflow, tag_mask, lens, sl, n_buckets);
    if (sl.slice)
        pairs_merge_kernel<<<std::min(sl.cap, 256u), PD_THREADS, 0, st>>>(bucket_start, reinterpret_cast<const uint4 *>(recs),
                                                                         stride_words / 4, tmp_rep, tmp_count, tmp_first,
                                                                         bucket_unique, overflow, lens, sl);
    return hipGetLastError();
}

// rows whose last padding word holds the key's length (pack.hip) as rows without: the store compares whole rows of
// tables built either way
__global__ void clear_last_word_kernel(uint32_t *__restrict__ recs, uint64_t n, uint32_t stride)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        recs[i * stride + stride - 1u] = 0u;
}

hipError_t launch_clear_last_word(uint32_t *recs, uint64_t n, uint32_t stride, hipStream_t st)
{
    if (n && stride)
        clear_last_word_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(recs, n, stride);
    return hipGetLastError();
}

uint32_t pairs_slice_items()
{
    if (const char *e = getenv("FQD_PAIRS_SLICE"))       // tests: small slices; 0 = one workgroup per bucket whatever its size
        return (uint32_t)strtoul(e, nullptr, 10);
    return PD_SLICE;
}

hipError_t launch_bucket_pairs_compact(const uint32_t *bucket_start, const uint32_t *unique_incl, uint32_t n_buckets,
                                       const uint32_t *tmp_rep, const uint32_t *tmp_count, const uint32_t *tmp_first,
                                       const uint32_t *recs, uint32_t stride_words, IdSource read_ids, uint32_t *urecs,
                                       uint32_t *ucounts, uint64_t *ufirst, hipStream_t st, SegHashOut seg_hashes,
                                       const uint32_t *lens, uint32_t *ulens, uint32_t row_cap, uint32_t len_hint)
{
    if (!n_buckets)
        return hipSuccess;
    const uint32_t q = stride_words / 4;
    if (seg_hashes.nseg && (q == 0 || q > 64 || (q & (q - 1))))
        return hipErrorInvalidValue;       // the caller asks only for power-of-two records
    const uint64_t threads = (uint64_t)n_buckets * 64;
    bucket_pairs_compact_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(
        bucket_start, unique_incl, n_buckets, tmp_rep, tmp_count, tmp_first, reinterpret_cast<const uint4 *>(recs),
        stride_words / 4, read_ids, reinterpret_cast<uint4 *>(urecs), ucounts, ufirst, seg_hashes, lens, ulens, row_cap,
        len_hint);
    return hipGetLastError();
}

}  // namespace fqd
