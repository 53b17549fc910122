// edit.hip -- the Levenshtein side of stage 3 and Trie.contains_sequence.
//
// Replaces the edit branches of TrieNode_FindNearest (reference
// _triemodule.c:410-413, :423-434, :456-464, :483-491) and within_edit_distance
// (distances.h:33-88), which together compute exact bounded Levenshtein.
//
// Candidate generation (pigeonhole for edit distance): cut key b (length lb) into
// d+1 segments [lb*s/(d+1), lb*(s+1)/(d+1)). d edits touch at most d segments, so
// if lev(a, b) <= d some segment of b occurs verbatim in a, starting at most d
// positions away from where it starts in b. Every unique key therefore files
//   * an INDEX record per own segment s:      hash(len, s, segment bits)
//   * a PROBE record per (length class l with |l - len| <= d that occurs in the
//     input, segment s of an l-long key, shift delta in [-d, d]):
//                                             hash(l, s, bits of own substring)
// (the probe of its own length class at shift 0 is its index record and is skipped).
// Records are sorted by hash; inside a run every (index, index) and (index, probe)
// pair of different keys becomes a candidate; candidates are sorted and made unique;
// each unique candidate is verified once by a banded DP on the two records.
#include "fqd_internal.h"

namespace {

// 32 bits (or fewer at the key's end) of plane k starting at base `start`.
__device__ __forceinline__ uint32_t plane_bits(const uint32_t *rec, uint32_t K, uint32_t W, uint32_t k,
                                               uint32_t start)
{
    const uint32_t w = start >> 5, sft = start & 31u;
    const uint32_t lo = w < W ? rec[w * K + k] : 0u;
    const uint32_t hi = (w + 1 < W) ? rec[(w + 1) * K + k] : 0u;
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sft);
}

// hash of (class length, segment number, the nbits bases starting at `start`)
__device__ __forceinline__ uint32_t substring_hash(const uint32_t *rec, uint32_t K, uint32_t W, uint32_t cls_len,
                                                   uint32_t seg, uint32_t start, uint32_t nbits)
{
    uint32_t h = fqd_mix32(cls_len * 0x9E3779B1u + seg * 0x85EBCA77u + 0x27D4EB2Fu);
    for (uint32_t off = 0; off < nbits; off += 32) {
        const uint32_t rem = nbits - off;
        const uint32_t mask = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
        for (uint32_t k = 0; k < K; k++) {
            h = (h + (plane_bits(rec, K, W, k, start + off) & mask)) * 0x9E3779B1u;
            h ^= h >> 15;
        }
    }
    return fqd_mix32(h);
}

constexpr uint32_t ROLE_PROBE = 0x80000000u;
constexpr uint32_t DEAD_HASH = 0xFFFFFFFFu;

// One thread per unique key; slots_per_key slots each; unused slots get DEAD_HASH and sort last.
__global__ __launch_bounds__(256) void edit_records_kernel(const uint32_t *__restrict__ urecs,
                                                           const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                           uint32_t d, const uint8_t *__restrict__ len_present,
                                                           uint32_t slots_per_key, uint32_t *__restrict__ out_hash,
                                                           uint32_t *__restrict__ out_payload)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U)
        return;
    const uint32_t K = sh.planes, W = sh.words, nseg = d + 1;
    const uint32_t len = fqd_key_len(sh, ulens, u);
    const uint32_t *rec = urecs + u * sh.stride;
    uint32_t *hs = out_hash + u * slots_per_key, *pl = out_payload + u * slots_per_key;
    uint32_t n = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        uint32_t lo, hi;
        fqd_segment(len, s, nseg, lo, hi);
        uint32_t h = substring_hash(rec, K, W, len, s, lo, hi - lo);
        hs[n] = h == DEAD_HASH ? 0u : h;
        pl[n] = (uint32_t)u;
        n++;
    }
    const uint32_t l0 = len > d ? len - d : 0u;
    for (uint32_t l = l0; l <= len + d && l <= sh.max_len; l++) {
        if (!len_present[l])
            continue;
        for (uint32_t s = 0; s < nseg; s++) {
            uint32_t lo, hi;
            fqd_segment(l, s, nseg, lo, hi);
            const uint32_t nb = hi - lo;
            for (int delta = -(int)d; delta <= (int)d; delta++) {
                if (l == len && delta == 0)
                    continue;  // that is the index record
                const int start = (int)lo + delta;
                if (start < 0 || (uint32_t)start + nb > len)
                    continue;
                uint32_t h = substring_hash(rec, K, W, l, s, (uint32_t)start, nb);
                hs[n] = h == DEAD_HASH ? 0u : h;
                pl[n] = (uint32_t)u | ROLE_PROBE;
                n++;
            }
        }
    }
    for (; n < slots_per_key; n++) {
        hs[n] = DEAD_HASH;
        pl[n] = 0;
    }
}

// One thread per sorted record; walks forward over its run and emits candidate pairs.
__global__ __launch_bounds__(256) void edit_candidates_kernel(const uint32_t *__restrict__ sorted_hash,
                                                              const uint32_t *__restrict__ sorted_payload, uint64_t R,
                                                              const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t d,
                                                              uint32_t shard, uint32_t n_shards,
                                                              uint64_t *__restrict__ cands,
                                                              unsigned long long *__restrict__ cand_count,
                                                              uint64_t cand_cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R)
        return;
    const uint32_t h = sorted_hash[i];
    if (h == DEAD_HASH || (n_shards > 1 && (h % n_shards) != shard))
        return;
    const uint32_t pi = sorted_payload[i];
    const uint32_t ui = pi & ~ROLE_PROBE;
    const bool probe_i = (pi & ROLE_PROBE) != 0;
    const uint32_t li = fqd_key_len(sh, ulens, ui);
    for (uint64_t j = i + 1; j < R && sorted_hash[j] == h; j++) {
        const uint32_t pj = sorted_payload[j];
        const uint32_t uj = pj & ~ROLE_PROBE;
        if (uj == ui || (probe_i && (pj & ROLE_PROBE)))
            continue;
        const uint32_t lj = fqd_key_len(sh, ulens, uj);
        if ((li > lj ? li - lj : lj - li) > d)
            continue;
        const uint64_t a = ui < uj ? ui : uj, b = ui < uj ? uj : ui;
        const unsigned long long at = atomicAdd(cand_count, 1ull);
        if (at < cand_cap)
            cands[at] = (a << 32) | b;
    }
}

// ---- bounded Levenshtein over symbol accessors -----------------------------------
struct RecSeq {
    const uint32_t *rec;
    uint32_t K;
    __device__ __forceinline__ uint32_t at(uint32_t p) const
    {
        uint32_t c = 0;
        for (uint32_t k = 0; k < K; k++)
            c |= ((rec[(p >> 5) * K + k] >> (p & 31u)) & 1u) << k;
        return c;
    }
};
struct DecodedRecSeq {  // record decoded to bytes through the alphabet (code -> symbol)
    const uint32_t *rec;
    uint32_t K;
    const uint8_t *alphabet;
    __device__ __forceinline__ uint32_t at(uint32_t p) const
    {
        uint32_t c = 0;
        for (uint32_t k = 0; k < K; k++)
            c |= ((rec[(p >> 5) * K + k] >> (p & 31u)) & 1u) << k;
        return alphabet[c];
    }
};
struct ByteSeq {
    const uint8_t *p;
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return p[i]; }
};

constexpr int EDIT_MAX_D = 64;

// distances.h:33-88 is exact bounded Levenshtein; this is the same predicate as a banded DP
// (band 2d+1 around the diagonal, values clamped at d+1, early exit when a row exceeds d).
template <typename SA, typename SB>
__device__ bool within_edit(const SA &a, uint32_t la, const SB &b, uint32_t lb, int d)
{
    const uint32_t gap = la > lb ? la - lb : lb - la;
    if (d < 0 || gap > (uint32_t)d)
        return false;
    if ((uint32_t)d >= (la > lb ? la : lb))
        return true;
    if (d > EDIT_MAX_D)
        d = EDIT_MAX_D;
    const int INF = d + 1, B = 2 * d + 1;
    int prev[2 * EDIT_MAX_D + 2], cur[2 * EDIT_MAX_D + 2];
    for (int k = 0; k < B; k++)
        prev[k] = k >= d ? k - d : INF;
    for (uint32_t i = 1; i <= la; i++) {
        const uint32_t ai = a.at(i - 1);
        int row_min = INF;
        for (int k = 0; k < B; k++) {
            const long long j = (long long)i + k - d;
            int v = INF;
            if (j >= 0 && j <= (long long)lb) {
                if (j == 0) {
                    v = i > (uint32_t)INF ? INF : (int)i;
                } else {
                    const int sub = prev[k] + (ai != b.at((uint32_t)j - 1) ? 1 : 0);
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
    return prev[(int)lb - (int)la + d] <= d;
}

template <typename SA, typename SB>
__device__ bool within_hamming(const SA &a, uint32_t la, const SB &b, uint32_t lb, int d)
{
    if (la != lb)
        return false;
    int budget = d;
    for (uint32_t i = 0; i < la; i++)
        if (a.at(i) != b.at(i) && --budget < 0)
            return false;
    return true;
}

// cands sorted ascending; a candidate equal to its predecessor is a repeat.
__global__ __launch_bounds__(256) void edit_verify_kernel(const uint64_t *__restrict__ cands, uint64_t C,
                                                          const uint32_t *__restrict__ urecs,
                                                          const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t d,
                                                          uint32_t *__restrict__ edges,
                                                          unsigned long long *__restrict__ edge_count, uint64_t edge_cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    uint32_t u = 0, v = 0;
    if (i < C) {
        const uint64_t c = cands[i];
        if (i == 0 || cands[i - 1] != c) {
            u = (uint32_t)(c >> 32);
            v = (uint32_t)c;
            const RecSeq a{urecs + (uint64_t)u * sh.stride, sh.planes}, b{urecs + (uint64_t)v * sh.stride, sh.planes};
            hit = within_edit(a, fqd_key_len(sh, ulens, u), b, fqd_key_len(sh, ulens, v), (int)d);
        }
    }
    const unsigned long long m = __ballot(hit);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long at = 0;
        if ((int)fqd_lane() == leader)
            at = atomicAdd(edge_count, (unsigned long long)__popcll(m));
        at = __shfl(at, leader);
        if (hit) {
            at += __popcll(m & fqd_lanemask_lt());
            if (at < edge_cap) {
                edges[2 * at] = u;
                edges[2 * at + 1] = v;
            }
        }
    }
}

__global__ void len_present_kernel(const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                   uint8_t *__restrict__ len_present)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < U)
        len_present[fqd_key_len(sh, ulens, u)] = 1;
}

// Trie.contains_sequence (_triemodule.c:730-758): one thread per (query, unique key).
__global__ __launch_bounds__(256) void contains_kernel(const uint8_t *__restrict__ q, const uint64_t *__restrict__ qo,
                                                       uint64_t nq, const uint32_t *__restrict__ urecs,
                                                       const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                       const uint8_t *__restrict__ alphabet, int d, int metric,
                                                       uint32_t *__restrict__ hit_flags,
                                                       const uint8_t *__restrict__ alive /* NULL: every row */)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t qi = blockIdx.y;
    if (u >= U || qi >= nq)
        return;
    if (alive && !alive[u])
        return;
    const ByteSeq a{q + qo[qi]};
    const uint64_t la64 = qo[qi + 1] - qo[qi];
    const uint32_t lb = fqd_key_len(sh, ulens, u);
    if (la64 > 0xFFFFFFFFull)
        return;
    const DecodedRecSeq b{urecs + u * sh.stride, sh.planes, alphabet};
    const bool ok = metric ? within_edit(a, (uint32_t)la64, b, lb, d) : within_hamming(a, (uint32_t)la64, b, lb, d);
    if (ok)
        hit_flags[qi] = 1;
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

namespace fqd {

hipError_t launch_len_present(const uint32_t *ulens, uint64_t U, KeyShape sh, uint8_t *len_present, hipStream_t st)
{
    if (U)
        len_present_kernel<<<grid_for(U), 256, 0, st>>>(ulens, U, sh, len_present);
    return hipGetLastError();
}

hipError_t launch_edit_records(const uint32_t *urecs, const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t d,
                               const uint8_t *len_present, uint32_t slots_per_key, uint32_t *out_hash,
                               uint32_t *out_payload, hipStream_t st)
{
    if (U)
        edit_records_kernel<<<grid_for(U), 256, 0, st>>>(urecs, ulens, U, sh, d, len_present, slots_per_key, out_hash,
                                                         out_payload);
    return hipGetLastError();
}

hipError_t launch_edit_candidates(const uint32_t *sorted_hash, const uint32_t *sorted_payload, uint64_t R,
                                  const uint32_t *ulens, KeyShape sh, uint32_t d, uint32_t shard, uint32_t n_shards,
                                  uint64_t *cands, unsigned long long *cand_count, uint64_t cand_cap, hipStream_t st)
{
    if (R)
        edit_candidates_kernel<<<grid_for(R), 256, 0, st>>>(sorted_hash, sorted_payload, R, ulens, sh, d, shard,
                                                            n_shards, cands, cand_count, cand_cap);
    return hipGetLastError();
}

hipError_t launch_edit_verify(const uint64_t *cands, uint64_t C, const uint32_t *urecs, const uint32_t *ulens,
                              KeyShape sh, uint32_t d, uint32_t *edges, unsigned long long *edge_count,
                              uint64_t edge_cap, hipStream_t st)
{
    if (C)
        edit_verify_kernel<<<grid_for(C), 256, 0, st>>>(cands, C, urecs, ulens, sh, d, edges, edge_count, edge_cap);
    return hipGetLastError();
}

hipError_t launch_contains(const uint8_t *q, const uint64_t *qo, uint64_t nq, const uint32_t *urecs,
                           const uint32_t *ulens, uint64_t U, KeyShape sh, const uint8_t *alphabet_dev, int d,
                           int metric, uint32_t *hit_flags, hipStream_t st, const uint8_t *alive)
{
    if (!U || !nq)
        return hipSuccess;
    if (nq > 65535)
        return hipErrorInvalidValue;
    dim3 grid(grid_for(U), (unsigned)nq);
    contains_kernel<<<grid, 256, 0, st>>>(q, qo, nq, urecs, ulens, U, sh, alphabet_dev, d, metric, hit_flags, alive);
    return hipGetLastError();
}

}  // namespace fqd
