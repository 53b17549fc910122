// edges.hip -- stage 3: neighbour search. Replaces TrieNode_FindNearest +
// within_hamming_distance / within_edit_distance (reference
// _triemodule.c:380-495, distances.h:8-88).
//
// Pigeonhole: cut every key into d+1 segments
// [len*s/(d+1), len*(s+1)/(d+1)); two keys of equal length within Hamming
// distance d agree exactly on at least one segment. Per segment s the unique
// keys are sorted by a 32-bit hash of (len, s, segment bits) (prims.hip), which
// makes every bucket a contiguous run; bucket_pairs_kernel compares all pairs
// inside a run. A hash only proposes candidates: the pair kernel counts the
// real mismatches on the full records. A pair is emitted in the pass of the
// FIRST segment on which it truly agrees, so every edge appears exactly once.
#include "fqd_internal.h"

namespace {

// Q = stride/4 lanes share one record, one uint4 (4 words) each, so a wave reads whole records
// with fully coalesced 16-byte loads whatever the record size. A segment's hash is
// fmix(len, s, SUM over its words of mix(word & segment mask, word index)): the sum is
// order-free, so the Q lanes add up their partial sums with shuffles.
// Segment bounds of fixed-length keys, worked out once on the host (two integer divisions per
// segment and thread otherwise: the kernel was ALU bound on them). n == 0: compute per key (ragged).
struct SegBounds {
    uint32_t n;
    uint32_t lo[8], hi[8];
};

// A wave takes SH_GROUPS consecutive groups of 64 / Q records, their loads requested together: one
// group per wave -- a single 16-byte load per lane, then ~200 ALU instructions and out -- left the
// kernel at 1.5 TB/s on 128-byte records (waves too short for their launch cost).
constexpr uint32_t SH_GROUPS = 4;

__global__ __launch_bounds__(256) void segment_hashes_kernel(const uint32_t *__restrict__ urecs,
                                                             const uint32_t *__restrict__ ulens, uint64_t U,
                                                             KeyShape sh, uint32_t nseg, uint32_t s_begin,
                                                             uint32_t s_end, uint32_t mod, SegBounds fixed,
                                                             uint32_t inv_q, uint32_t inv_k,
                                                             uint32_t *__restrict__ seg_hashes)
{
    const uint32_t Q = sh.stride / 4, KW = sh.planes * sh.words;
    const uint32_t rpw = 64u / Q;                       // records per wave and group (Q <= 64 checked by the host)
    const uint32_t lane = fqd_lane();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // x / Q and x / K by multiplication (inv = ceil(2^20 / divisor), exact for x < 2^14)
    const uint32_t rl = (lane * inv_q) >> 20, q = lane - rl * Q;
    uint4 vs[SH_GROUPS];
    uint32_t lens[SH_GROUPS];
    bool actives[SH_GROUPS];
#pragma unroll
    for (uint32_t g = 0; g < SH_GROUPS; g++) {
        const uint64_t u = (wave * SH_GROUPS + g) * rpw + rl;
        actives[g] = rl < rpw && u < U;
        vs[g] = make_uint4(0, 0, 0, 0);
        lens[g] = 0;
        if (actives[g]) {
            vs[g] = reinterpret_cast<const uint4 *>(urecs + u * sh.stride)[q];
            lens[g] = fqd_key_len(sh, ulens, u);
        }
    }
    uint32_t wi[4];   // 32-base word index of each of this lane's 4 record words
#pragma unroll
    for (uint32_t e = 0; e < 4; e++)
        wi[e] = ((q * 4 + e) * inv_k) >> 20;
#pragma unroll
    for (uint32_t g = 0; g < SH_GROUPS; g++) {
        const uint64_t u = (wave * SH_GROUPS + g) * rpw + rl;
        const bool active = actives[g];
        const uint32_t len = lens[g];
        const uint32_t word[4] = {vs[g].x, vs[g].y, vs[g].z, vs[g].w};
        // segments [s_begin, s_end) of the nseg-way split; row (s - s_begin) of the output. mod != 0
        // stores hash % mod (the owner rank of a segment-routed exchange).
        for (uint32_t s = s_begin; s < s_end; s++) {
            uint32_t lo, hi;
            if (fixed.n) {
                lo = fixed.lo[s - s_begin];
                hi = fixed.hi[s - s_begin];
            } else {
                fqd_segment(len, s, nseg, lo, hi);
            }
            uint32_t part = 0;
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                const uint32_t j = q * 4 + e;               // word index in the record
                if (j < KW) {
                    const uint32_t m = fqd_range_mask(wi[e], lo, hi);
                    if (m)
                        part += fqd_mix32((word[e] & m) + (j + 1u) * 0x9E3779B1u);
                }
            }
            for (uint32_t off = 1; off < Q; off <<= 1) {    // segmented sum over the record's Q lanes
                const uint32_t other = __shfl_down(part, off);
                if (q + off < Q)
                    part += other;
            }
            if (active && q == 0) {
                const uint32_t h = fqd_mix32(part + fqd_mix32(len * 0x9E3779B1u + s * 0x85EBCA77u + 0x165667B1u));
                seg_hashes[(uint64_t)(s - s_begin) * U + u] = mod ? h % mod : h;
            }
        }
    }
}

// Multi-GPU: keep only this rank's share of the buckets (bucket_hash % n_shards == shard)
// BEFORE the sort, so a rank sorts and searches 1/G of the unique table. One atomic per wave.
__global__ __launch_bounds__(256) void select_shard_kernel(const uint32_t *__restrict__ hashes, uint64_t U,
                                                           uint32_t shard, uint32_t n_shards,
                                                           uint32_t *__restrict__ out_hash,
                                                           uint32_t *__restrict__ out_uid,
                                                           unsigned long long *__restrict__ counter)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t h = u < U ? hashes[u] : 0u;
    const bool mine = u < U && (h % n_shards) == shard;
    const unsigned long long m = __ballot(mine);
    if (!m)
        return;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long at = 0;
    if ((int)fqd_lane() == leader)
        at = atomicAdd(counter, (unsigned long long)__popcll(m));
    at = __shfl(at, leader);
    if (mine) {
        at += __popcll(m & fqd_lanemask_lt());
        out_hash[at] = h;
        out_uid[at] = (uint32_t)u;
    }
}

// ---- the pair kernel -----------------------------------------------------------
// Persistent blocks walk tiles of T consecutive positions of the bucket-sorted
// order. Keys that sit in a bucket of >= 2 are gathered ONCE from HBM (one
// aligned record each) into a word-major LDS tile (word j of key t at
// tile[j*T + t]: lanes t, t+1, ... hit consecutive banks). Lane t then walks
// forward over the rest of its bucket, XOR/OR/popcount per 32-base word with
// early exit; partners beyond the tile are read from HBM. Hits go to an LDS
// edge buffer (wave ballot + popcount of the lanes below, one LDS atomic per
// wave); the buffer is flushed to HBM with ONE global atomic per flush, so the
// single edge counter is touched a few thousand times per launch instead of
// once per hit.
constexpr uint32_t PAIR_ECAP = 1024;  // edges buffered per block

template <int K, bool USE_LDS>
__global__ __launch_bounds__(256) void bucket_pairs_kernel(
    const uint32_t *__restrict__ sorted_hash, const uint32_t *__restrict__ sorted_uid, uint64_t U,
    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t d, uint32_t seg,
    uint32_t nseg, uint32_t shard, uint32_t n_shards, uint32_t *__restrict__ edges,
    unsigned long long *__restrict__ edge_count, uint64_t edge_cap, fqd::PairStats *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t T = blockDim.x;
    uint32_t *s_hash = smem;                    // T
    uint32_t *s_uid = smem + T;                 // T
    uint32_t *s_len = smem + 2 * T;             // T
    uint32_t *s_edges = smem + 3 * T;           // 2 * PAIR_ECAP
    uint32_t *s_ctl = s_edges + 2 * PAIR_ECAP;  // [0] buffered edges, [1..2] flush base (lo, hi)
    uint32_t *tile = s_ctl + 4;                 // KW * T when USE_LDS

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t W = sh.words, KW = W * K, stride = sh.stride;
    const uint64_t n_tiles = (U + T - 1) / T;
    unsigned long long n_pairs = 0, n_hits = 0, n_gathered = 0;

    if (tid == 0)
        s_ctl[0] = 0;
    __syncthreads();

    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * T;
        const uint64_t i = base + tid;
        const bool valid = i < U;
        const uint32_t h = valid ? sorted_hash[i] : 0u;
        const uint32_t uid = valid ? sorted_uid[i] : 0u;
        s_hash[tid] = h;
        s_uid[tid] = uid;
        __syncthreads();
        bool prev_same = false, next_same = false;
        if (valid) {
            if (i > 0)
                prev_same = (tid > 0 ? s_hash[tid - 1] : sorted_hash[i - 1]) == h;
            if (i + 1 < U)
                next_same = (tid + 1 < T ? s_hash[tid + 1] : sorted_hash[i + 1]) == h;
        }
        const bool mine = n_shards <= 1 || (h % n_shards) == shard;
        const bool multi = valid && mine && (prev_same || next_same);
        const uint32_t *my_rec = urecs + (uint64_t)uid * stride;
        uint32_t len = sh.max_len;
        if (multi) {
            n_gathered++;
            if (sh.ragged)
                len = ulens[uid];
        }
        if (USE_LDS && multi) {
            // only keys in a bucket of >= 2 (typically 10-40 %) fetch their record; a per-lane
            // fetch beats a cooperative one here because the idle lanes have nothing to loop over
            const uint4 *src = reinterpret_cast<const uint4 *>(my_rec);
            for (uint32_t q = 0; q < stride / 4; q++) {
                const uint4 v = src[q];
                const uint32_t j = q * 4;
                if (j + 0 < KW) tile[(j + 0) * T + tid] = v.x;
                if (j + 1 < KW) tile[(j + 1) * T + tid] = v.y;
                if (j + 2 < KW) tile[(j + 2) * T + tid] = v.z;
                if (j + 3 < KW) tile[(j + 3) * T + tid] = v.w;
            }
        }
        s_len[tid] = len;
        __syncthreads();

        if (multi && next_same) {
            for (uint64_t gj = i + 1; gj < U; gj++) {
                const uint32_t j = (uint32_t)(gj - base);
                const bool in_tile = j < T;
                const uint32_t hj = in_tile ? s_hash[j] : sorted_hash[gj];
                if (hj != h)
                    break;
                const uint32_t uj = in_tile ? s_uid[j] : sorted_uid[gj];
                const uint32_t lj = in_tile ? s_len[j] : fqd_key_len(sh, ulens, uj);
                n_pairs++;
                bool hit = false;
                if (lj == len) {
                    const uint32_t *other = urecs + (uint64_t)uj * stride;
                    uint32_t dist = 0;
                    for (uint32_t w = 0; w < W && dist <= d; w++) {
                        uint32_t dw = 0;
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            const uint32_t a = USE_LDS ? tile[(w * K + k) * T + tid] : my_rec[w * K + k];
                            const uint32_t b = (USE_LDS && in_tile) ? tile[(w * K + k) * T + j] : other[w * K + k];
                            dw |= a ^ b;
                        }
                        dist += __popc(dw);
                    }
                    if (dist <= d) {
                        // emit only in the pass of the first truly agreeing segment
                        hit = true;
                        for (uint32_t s2 = 0; s2 < seg && hit; s2++) {
                            uint32_t lo, hi;
                            fqd_segment(len, s2, nseg, lo, hi);
                            bool agree = true;
                            if (hi > lo) {
                                for (uint32_t w = lo >> 5; w <= ((hi - 1) >> 5) && agree; w++) {
                                    uint32_t dw = 0;
#pragma unroll
                                    for (int k = 0; k < K; k++) {
                                        const uint32_t a = USE_LDS ? tile[(w * K + k) * T + tid] : my_rec[w * K + k];
                                        const uint32_t b =
                                            (USE_LDS && in_tile) ? tile[(w * K + k) * T + j] : other[w * K + k];
                                        dw |= a ^ b;
                                    }
                                    if (dw & fqd_range_mask(w, lo, hi))
                                        agree = false;
                                }
                            }
                            if (agree)
                                hit = false;
                        }
                    }
                }
                const unsigned long long m = __ballot(hit);
                if (m) {
                    const int leader = __ffsll((long long)m) - 1;
                    uint32_t at = 0;
                    if ((int)lane == leader)
                        at = atomicAdd(&s_ctl[0], (uint32_t)__popcll(m));
                    at = __shfl(at, leader);
                    if (hit) {
                        at += __popcll(m & fqd_lanemask_lt());
                        const uint32_t eu = uid < uj ? uid : uj, ev = uid < uj ? uj : uid;
                        if (at < PAIR_ECAP) {
                            s_edges[2 * at] = eu;
                            s_edges[2 * at + 1] = ev;
                        } else {
                            // buffer full inside one tile (a very large bucket): straight to HBM
                            const unsigned long long g = atomicAdd(edge_count, 1ull);
                            if (g < edge_cap) {
                                edges[2 * g] = eu;
                                edges[2 * g + 1] = ev;
                            }
                        }
                        n_hits++;
                    }
                }
            }
        }
        __syncthreads();
        // flush when more than half full, and after the block's last tile
        const uint32_t buffered = s_ctl[0];
        const bool last = t + gridDim.x >= n_tiles;
        if (buffered >= PAIR_ECAP / 2 || (last && buffered)) {
            const uint32_t cnt = buffered < PAIR_ECAP ? buffered : PAIR_ECAP;
            if (tid == 0) {
                const unsigned long long g = atomicAdd(edge_count, (unsigned long long)cnt);
                s_ctl[1] = (uint32_t)g;
                s_ctl[2] = (uint32_t)(g >> 32);
            }
            __syncthreads();
            const unsigned long long g = ((unsigned long long)s_ctl[2] << 32) | s_ctl[1];
            for (uint32_t e = tid; e < cnt; e += T)
                if (g + e < edge_cap) {
                    edges[2 * (g + e)] = s_edges[2 * e];
                    edges[2 * (g + e) + 1] = s_edges[2 * e + 1];
                }
            __syncthreads();
            if (tid == 0)
                s_ctl[0] = 0;
            __syncthreads();
        }
    }
    if (stats) {
        // block totals land in one of FQD_STAT_SLOTS slots: no hot counter
        for (int o = 32; o; o >>= 1) {
            n_gathered += __shfl_xor(n_gathered, o);
            n_pairs += __shfl_xor(n_pairs, o);
            n_hits += __shfl_xor(n_hits, o);
        }
        __syncthreads();
        unsigned long long *red = reinterpret_cast<unsigned long long *>(s_edges);
        if (lane == 0) {
            red[(tid >> 6) * 3 + 0] = n_gathered;
            red[(tid >> 6) * 3 + 1] = n_pairs;
            red[(tid >> 6) * 3 + 2] = n_hits;
        }
        __syncthreads();
        if (tid < 3) {
            unsigned long long tot = 0;
            for (uint32_t wv = 0; wv < T / 64; wv++)
                tot += red[wv * 3 + tid];
            fqd::PairStats *slot = stats + (blockIdx.x % FQD_STAT_SLOTS);
            unsigned long long *dst = tid == 0 ? &slot->keys_gathered : (tid == 1 ? &slot->pairs_compared : &slot->edges);
            if (tot)
                atomicAdd(dst, tot);
        }
    }
}

// ---- raw-byte predicates (single calls of the reference surface) ---------------

// distances.h:8-31
__device__ bool bytes_within_hamming(const uint8_t *a, uint64_t la, const uint8_t *b, uint64_t lb, int d)
{
    if (la != lb)
        return false;
    int budget = d;
    for (uint64_t i = 0; i < la; i++)
        if (a[i] != b[i] && --budget < 0)
            return false;
    return true;
}

// distances.h:33-88 computes exact bounded Levenshtein; here as a banded DP
// (band 2d+1, values clamped at d+1), which gives the same predicate.
constexpr int EDIT_MAX_D = 64;
__device__ bool bytes_within_edit(const uint8_t *a, uint64_t la, const uint8_t *b, uint64_t lb, int d)
{
    const uint64_t gap = la > lb ? la - lb : lb - la;
    if (d < 0 || gap > (uint64_t)d)
        return false;
    if ((uint64_t)d >= (la > lb ? la : lb))
        return true;
    if (d > EDIT_MAX_D)
        d = EDIT_MAX_D;  // host refuses larger d unless trivially true
    const int INF = d + 1, B = 2 * d + 1;
    int prev[2 * EDIT_MAX_D + 2], cur[2 * EDIT_MAX_D + 2];
    for (int k = 0; k < B; k++)
        prev[k] = k >= d ? k - d : INF;  // row 0: D[0][j] = j
    for (uint64_t i = 1; i <= la; i++) {
        int row_min = INF;
        for (int k = 0; k < B; k++) {
            const int64_t j = (int64_t)i + k - d;
            int v = INF;
            if (j >= 0 && j <= (int64_t)lb) {
                if (j == 0) {
                    v = (int)(i > (uint64_t)INF ? INF : i);
                } else {
                    const int sub = prev[k] + (a[i - 1] != b[j - 1] ? 1 : 0);
                    const int del = k + 1 < B ? prev[k + 1] + 1 : INF;
                    const int ins = k > 0 ? cur[k - 1] + 1 : INF;
                    v = sub < del ? sub : del;
                    v = v < ins ? v : ins;
                    if (v > INF)
                        v = INF;
                }
            }
            cur[k] = v;
            row_min = v < row_min ? v : row_min;
        }
        if (row_min > d)
            return false;
        for (int k = 0; k < B; k++)
            prev[k] = cur[k];
    }
    const int k = (int)((int64_t)lb - (int64_t)la + d);
    return prev[k] <= d;
}

__global__ void pairs_within_kernel(const uint8_t *__restrict__ a, const uint64_t *__restrict__ ao,
                                    const uint8_t *__restrict__ b, const uint64_t *__restrict__ bo, uint64_t n,
                                    int d, int metric, uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint8_t *pa = a + ao[i], *pb = b + bo[i];
    const uint64_t la = ao[i + 1] - ao[i], lb = bo[i + 1] - bo[i];
    out[i] = metric ? bytes_within_edit(pa, la, pb, lb, d) : bytes_within_hamming(pa, la, pb, lb, d);
}

}  // namespace

namespace fqd {

hipError_t launch_segment_hashes(const uint32_t *urecs, const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t nseg,
                                 uint32_t s_begin, uint32_t s_end, uint32_t mod, uint32_t *seg_hashes,
                                 hipStream_t st)
{
    if (!U)
        return hipSuccess;
    const uint32_t Q = sh.stride / 4;
    if (Q > 64)
        return hipErrorInvalidValue;  // records above 1 KiB: not reachable within the pack tile limit
    const uint64_t rpw = 64 / Q, waves = (U + rpw * SH_GROUPS - 1) / (rpw * SH_GROUPS), blocks = (waves + 3) / 4;
    if (blocks > 0x7FFFFFull * 256)
        return hipErrorInvalidValue;
    SegBounds fixed{};
    if (!sh.ragged && s_end - s_begin <= 8) {
        fixed.n = s_end - s_begin;
        for (uint32_t s = s_begin; s < s_end; s++) {     // as fqd_segment
            fixed.lo[s - s_begin] = sh.max_len * s / nseg;
            fixed.hi[s - s_begin] = sh.max_len * (s + 1) / nseg;
        }
    }
    const uint32_t inv_q = ((1u << 20) + (uint32_t)Q - 1) / (uint32_t)Q, inv_k = ((1u << 20) + sh.planes - 1) / sh.planes;
    segment_hashes_kernel<<<(unsigned)blocks, 256, 0, st>>>(urecs, ulens, U, sh, nseg, s_begin, s_end, mod, fixed, inv_q,
                                                            inv_k, seg_hashes);
    return hipGetLastError();
}

hipError_t launch_bucket_pairs(const uint32_t *sorted_hash, const uint32_t *sorted_uid, uint64_t U,
                               const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t d, uint32_t seg,
                               uint32_t nseg, uint32_t shard, uint32_t n_shards, uint32_t *edges,
                               unsigned long long *edge_count, uint64_t edge_cap, PairStats *stats, hipStream_t st)
{
    if (!U)
        return hipSuccess;
    const uint32_t KW = sh.planes * sh.words;
    uint32_t T = 256;
    const uint32_t budget = 60 * 1024;
    const uint32_t fixed_words = 2 * PAIR_ECAP + 4;
    bool use_lds = true;
    while (T > 64 && ((3 + KW) * T + fixed_words) * 4 > budget)
        T >>= 1;
    if (((3 + KW) * T + fixed_words) * 4 > budget) {
        use_lds = false;
        T = 256;
    }
    const uint32_t lds = ((3 + (use_lds ? KW : 0)) * T + fixed_words) * 4;
    const uint64_t n_tiles = (U + T - 1) / T;
    // persistent grid: enough blocks to fill 256 CUs several times over, few enough that the
    // per-block flush/stat atomics stay in the thousands
    const unsigned grid = (unsigned)(n_tiles < 4096 ? n_tiles : 4096);
#define FQD_PAIRS_CASE(KK)                                                                                   \
    case KK:                                                                                                 \
        if (use_lds)                                                                                         \
            bucket_pairs_kernel<KK, true><<<grid, T, lds, st>>>(sorted_hash, sorted_uid, U, urecs, ulens, sh, d, \
                                                                seg, nseg, shard, n_shards, edges, edge_count,   \
                                                                edge_cap, stats);                                \
        else                                                                                                 \
            bucket_pairs_kernel<KK, false><<<grid, T, lds, st>>>(sorted_hash, sorted_uid, U, urecs, ulens, sh, d, \
                                                                 seg, nseg, shard, n_shards, edges, edge_count,   \
                                                                 edge_cap, stats);                                \
        break;
    switch (sh.planes) {
        FQD_PAIRS_CASE(1)
        FQD_PAIRS_CASE(2)
        FQD_PAIRS_CASE(3)
        FQD_PAIRS_CASE(4)
        FQD_PAIRS_CASE(5)
        FQD_PAIRS_CASE(6)
        FQD_PAIRS_CASE(7)
    default:
        return hipErrorInvalidValue;
    }
#undef FQD_PAIRS_CASE
    return hipGetLastError();
}

hipError_t launch_select_shard(const uint32_t *hashes, uint64_t U, uint32_t shard, uint32_t n_shards,
                               uint32_t *out_hash, uint32_t *out_uid, unsigned long long *counter, hipStream_t st)
{
    if (U)
        select_shard_kernel<<<(unsigned)((U + 255) / 256), 256, 0, st>>>(hashes, U, shard, n_shards, out_hash,
                                                                         out_uid, counter);
    return hipGetLastError();
}

hipError_t launch_pairs_within(const uint8_t *a, const uint64_t *ao, const uint8_t *b, const uint64_t *bo,
                               uint64_t n, int d, int metric, uint8_t *out, hipStream_t st)
{
    if (n)
        pairs_within_kernel<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(a, ao, b, bo, n, d, metric, out);
    return hipGetLastError();
}

}  // namespace fqd
