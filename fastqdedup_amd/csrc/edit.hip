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

// The hash fqd_segment_hash gives segment seg of a key of cls_len bases whose segment holds the nb bases that THIS
// record has from base `start` on: the record is read through a window shifted by delta = start - lo, word by word
// in the indexed key's coordinates. With it a probe item meets the index items the Hamming passes have hashed
// already (segment_hashes_kernel), and the edit search need not hash the index side again.
__device__ __forceinline__ uint32_t shifted_segment_hash(const uint32_t *rec, uint32_t K, uint32_t W, uint32_t cls_len,
                                                         uint32_t seg, uint32_t nseg, int delta)
{
    uint32_t lo, hi;
    fqd_segment(cls_len, seg, nseg, lo, hi);
    uint32_t part = 0;
    for (uint32_t w = lo >> 5; w * 32u < hi; w++) {
        const uint32_t m = fqd_range_mask(w, lo, hi);
        if (!m)
            continue;
        const int at = (int)(w * 32u) + delta;                // the record's base under bit 0 of word w
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t bits = at >= 0 ? plane_bits(rec, K, W, k, (uint32_t)at)
                                          : plane_bits(rec, K, W, k, 0u) << (uint32_t)(-at);
            part += fqd_mix32((bits & m) + (w * K + k + 1u) * 0x9E3779B1u);
        }
    }
    return fqd_mix32(part + fqd_mix32(cls_len * 0x9E3779B1u + seg * 0x85EBCA77u + 0x165667B1u));
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

// ---- the same search without a device-wide sort ("grouped") --------------------------------------
// Items (substring hash, payload) instead of sorted records; payload = uid (26 bits) | segment (2) |
// shift + 3 (3) | probe role (1), so d <= 3 and fewer than 2^26 keys. Fewer items, too:
//   * an INDEX item per key and own segment, as before;
//   * PROBE items only where one side of a pair has to file them: a pair (a, b) is found as soon as
//     ONE of the two probes the other's length class (if lev(a, b) <= d, some segment of b occurs in
//     a within d positions -- whichever of the two is called a), so of two length classes the
//     smaller one probes the larger (ties: the shorter keys probe), and the own class is probed with
//     shifts != 0 only (for d >= 2; with d = 1 two keys of one length are one substitution apart or
//     not neighbours at all). 2 items per key instead of 20 for 300-nt keys with a 1 % indel tail.
// The items are partitioned by hash bits and matched inside LDS (group.hip, the kernels of the
// Hamming passes); a candidate pair carries both payloads. A pair usually matches under several
// (direction, segment, shift) configurations: it is verified and reported under the FIRST matching
// one only (directions in uid order, then segments, then shifts -- the substrings are compared,
// a hash only proposes), so every edge comes out exactly once without sorting the candidates.
constexpr uint32_t EG_UID_BITS = 26, EG_UID_MASK = (1u << EG_UID_BITS) - 1u;

__device__ __forceinline__ uint32_t eg_payload(uint32_t uid, uint32_t seg, int delta, bool probe)
{
    return uid | (seg << EG_UID_BITS) | ((uint32_t)(delta + 3) << (EG_UID_BITS + 2)) | (probe ? ROLE_PROBE : 0u);
}

// does a key of length la file probes for class lb? (probe_mask[la] bit (lb - la + d); same class: shifts != 0)
__device__ __forceinline__ bool eg_probes(const uint8_t *__restrict__ probe_mask, uint32_t la, uint32_t lb, uint32_t d)
{
    const int j = (int)lb - (int)la + (int)d;
    return j >= 0 && j <= 2 * (int)d && ((probe_mask[la] >> j) & 1u);
}

// keys per length. Almost all keys share one or two lengths (26 M single adds on one address took
// 290 ms, ~11 ns each): a workgroup counts its 4096 keys in an LDS histogram -- runs of equal
// lengths in a register first -- and adds the non-empty bins once. Lengths of 2048 and more (rare)
// go to the global counters directly.
constexpr uint32_t EG_LEN_BINS = 2048;

__global__ __launch_bounds__(256) void edit_len_counts_kernel(const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                              uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_hist[EG_LEN_BINS];
    for (uint32_t b = threadIdx.x; b < EG_LEN_BINS; b += 256)
        s_hist[b] = 0;
    __syncthreads();
    const uint64_t u0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t run_len = 0, run = 0;
    auto flush = [&]() {
        if (!run)
            return;
        if (run_len < EG_LEN_BINS)
            atomicAdd(&s_hist[run_len], run);
        else
            atomicAdd(&counts[run_len], run);
    };
    for (uint32_t t = 0; t < 16 && u0 + t < U; t++) {
        const uint32_t len = fqd_key_len(sh, ulens, u0 + t);
        if (run && len == run_len) {
            run++;
        } else {
            flush();
            run_len = len;
            run = 1;
        }
    }
    flush();
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < EG_LEN_BINS && b <= sh.max_len; b += 256)
        if (s_hist[b])
            atomicAdd(&counts[b], s_hist[b]);
}

// pass 0: per_key[u] = probe items key u files (by its length class). pass 1: the index items (slot
// u * (d + 1) + s) and the probe items (behind all index items, at the key's scanned offset).
// STAGED (pass 1, records of up to 47 words): the workgroup's 256 records come in with coalesced 16-byte loads
// and are read from LDS, one record per thread at a stride of stride + 1 words (no bank conflicts) -- a thread
// walking its own 128-byte record in global memory shares no line with its neighbours (config 5's variant:
// 4.5 ms of a 25.8 ms step for 14 M keys).
template <bool STAGED>
__global__ __launch_bounds__(256) void edit_items_kernel(const uint32_t *__restrict__ urecs,
                                                         const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                         uint32_t d, const uint8_t *__restrict__ probe_mask,
                                                         const uint32_t *__restrict__ probe_count,
                                                         uint32_t *__restrict__ per_key,
                                                         const uint32_t *__restrict__ per_key_incl,
                                                         uint32_t *__restrict__ hashes, uint32_t *__restrict__ payloads,
                                                         int pass,
                                                         const uint32_t *__restrict__ index_from /* NULL, or the
                                                         [d + 1][U] segment hashes of the Hamming passes: they ARE the index
                                                         items' hashes, the probe items are hashed to meet them
                                                         (shifted_segment_hash), and only the probing keys' records
                                                         are read -- never with STAGED */,
                                                         uint32_t *__restrict__ probers /* != NULL: pass 1 lists the keys
                                                         that file probe items here (any order) for edit_probe_items_kernel
                                                         instead of hashing them itself: one such key in a hundred kept
                                                         half of the waves in the probe loops */,
                                                         unsigned long long *__restrict__ n_probers)
{
    __shared__ uint32_t s_n, s_at;
    if (probers && pass == 1) {
        if (threadIdx.x == 0)
            s_n = 0;
        __syncthreads();
    }
    extern __shared__ uint32_t s_recs[];
    const uint64_t u0 = (uint64_t)blockIdx.x * blockDim.x;
    const uint64_t u = u0 + threadIdx.x;
    if (STAGED) {
        const uint32_t S = sh.stride, S4 = S / 4;               // (the stride is a multiple of four words)
        const uint32_t nk = (uint32_t)min((uint64_t)blockDim.x, U - u0);
        const uint4 *src = reinterpret_cast<const uint4 *>(urecs + u0 * S);
        for (uint32_t i = threadIdx.x; i < nk * S4; i += blockDim.x) {
            const uint4 q = src[i];
            const uint32_t r = i / S4, w = (i - r * S4) * 4;
            uint32_t *dst = s_recs + r * (S + 1) + w;
            dst[0] = q.x;
            dst[1] = q.y;
            dst[2] = q.z;
            dst[3] = q.w;
        }
        __syncthreads();
    }
    if (probers && pass == 1) {
        // index items by everybody, the probing keys of the workgroup appended to the list with one atomic
        const bool live = u < U;
        const uint32_t len = live ? fqd_key_len(sh, ulens, u) : 0u;
        const uint32_t K = sh.planes, W = sh.words, nseg = d + 1;
        const uint32_t *rec = STAGED ? s_recs + threadIdx.x * (sh.stride + 1) : urecs + u * sh.stride;
        for (uint32_t s = 0; live && s < nseg; s++) {
            uint32_t lo, hi;
            fqd_segment(len, s, nseg, lo, hi);
            hashes[u * nseg + s] = index_from ? index_from[(uint64_t)s * U + u] : substring_hash(rec, K, W, len, s, lo, hi - lo);
            payloads[u * nseg + s] = eg_payload((uint32_t)u, s, 0, false);
        }
        const bool files = live && probe_count[len] != 0;
        uint32_t slot = 0;
        if (files)
            slot = atomicAdd(&s_n, 1u);
        __syncthreads();
        if (threadIdx.x == 0 && s_n)
            s_at = (uint32_t)atomicAdd(n_probers, (unsigned long long)s_n);
        __syncthreads();
        if (files)
            probers[s_at + slot] = (uint32_t)u;
        return;
    }
    if (u >= U)
        return;
    const uint32_t len = fqd_key_len(sh, ulens, u);
    if (pass == 0) {
        per_key[u] = probe_count[len];
        return;
    }
    const uint32_t K = sh.planes, W = sh.words, nseg = d + 1;
    const uint32_t *rec = STAGED ? s_recs + threadIdx.x * (sh.stride + 1) : urecs + u * sh.stride;
    for (uint32_t s = 0; s < nseg; s++) {
        uint32_t lo, hi;
        fqd_segment(len, s, nseg, lo, hi);
        hashes[u * nseg + s] = index_from ? index_from[(uint64_t)s * U + u] : substring_hash(rec, K, W, len, s, lo, hi - lo);
        payloads[u * nseg + s] = eg_payload((uint32_t)u, s, 0, false);
    }
    const uint32_t mine = probe_count[len];
    if (!mine)
        return;
    uint64_t at = U * nseg + (per_key_incl[u] - mine);
    for (int j = 0; j <= 2 * (int)d; j++) {
        if (!((probe_mask[len] >> j) & 1u))
            continue;
        const uint32_t l = (uint32_t)((int)len - (int)d + j);
        for (uint32_t s = 0; s < nseg; s++) {
            uint32_t lo, hi;
            fqd_segment(l, s, nseg, lo, hi);
            const uint32_t nb = hi - lo;
            for (int delta = -(int)d; delta <= (int)d; delta++) {
                if (l == len && delta == 0)
                    continue;
                const int start = (int)lo + delta;
                if (start < 0 || (uint32_t)start + nb > len)
                    continue;
                hashes[at] = index_from ? shifted_segment_hash(rec, K, W, l, s, nseg, delta)
                                        : substring_hash(rec, K, W, l, s, (uint32_t)start, nb);
                payloads[at] = eg_payload((uint32_t)u, s, delta, true);
                at++;
            }
        }
    }
}

// Light versions of edit_items_kernel's two cheap jobs, four keys per thread with their loads in flight together
// (inside the big kernel -- many registers, few waves -- each of them took 0.64 ms for 26 M keys, a chain of two
// dependent loads per key): pass 0 (per_key[u] = probe items of u's length class), and pass 1 when the index
// hashes exist already (index_from): copy them into item order, add the payloads, list the probing keys.
constexpr uint32_t EL_KEYS = 4;

__global__ __launch_bounds__(256) void edit_probe_counts_kernel(const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                                const uint32_t *__restrict__ probe_count,
                                                                uint32_t *__restrict__ per_key)
{
    const uint64_t u0 = ((uint64_t)blockIdx.x * EL_KEYS) * blockDim.x + threadIdx.x;
    uint32_t len[EL_KEYS];
#pragma unroll
    for (uint32_t t = 0; t < EL_KEYS; t++) {
        const uint64_t u = u0 + (uint64_t)t * blockDim.x;
        len[t] = u < U ? fqd_key_len(sh, ulens, u) : 0u;
    }
#pragma unroll
    for (uint32_t t = 0; t < EL_KEYS; t++) {
        const uint64_t u = u0 + (uint64_t)t * blockDim.x;
        if (u < U)
            per_key[u] = probe_count[len[t]];
    }
}

__global__ __launch_bounds__(256) void edit_index_items_kernel(const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                               uint32_t d, const uint32_t *__restrict__ probe_count,
                                                               const uint32_t *__restrict__ index_from,
                                                               uint32_t *__restrict__ hashes, uint32_t *__restrict__ payloads,
                                                               uint32_t *__restrict__ probers,
                                                               unsigned long long *__restrict__ n_probers)
{
    __shared__ uint32_t s_n, s_at;
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    const uint32_t nseg = d + 1;
    const uint64_t u0 = ((uint64_t)blockIdx.x * EL_KEYS) * blockDim.x + threadIdx.x;
    uint32_t len[EL_KEYS], slot[EL_KEYS];
    bool files[EL_KEYS];
#pragma unroll
    for (uint32_t t = 0; t < EL_KEYS; t++) {
        const uint64_t u = u0 + (uint64_t)t * blockDim.x;
        len[t] = u < U ? fqd_key_len(sh, ulens, u) : 0u;
    }
    for (uint32_t s = 0; s < nseg; s++) {
        uint32_t h[EL_KEYS];
#pragma unroll
        for (uint32_t t = 0; t < EL_KEYS; t++) {
            const uint64_t u = u0 + (uint64_t)t * blockDim.x;
            h[t] = u < U ? index_from[(uint64_t)s * U + u] : 0u;
        }
#pragma unroll
        for (uint32_t t = 0; t < EL_KEYS; t++) {
            const uint64_t u = u0 + (uint64_t)t * blockDim.x;
            if (u < U) {
                hashes[u * nseg + s] = h[t];
                payloads[u * nseg + s] = eg_payload((uint32_t)u, s, 0, false);
            }
        }
    }
#pragma unroll
    for (uint32_t t = 0; t < EL_KEYS; t++) {
        const uint64_t u = u0 + (uint64_t)t * blockDim.x;
        files[t] = u < U && probe_count[len[t]] != 0;
        slot[t] = files[t] ? atomicAdd(&s_n, 1u) : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_n)
        s_at = (uint32_t)atomicAdd(n_probers, (unsigned long long)s_n);
    __syncthreads();
#pragma unroll
    for (uint32_t t = 0; t < EL_KEYS; t++)
        if (files[t])
            probers[s_at + slot[t]] = (uint32_t)(u0 + (uint64_t)t * blockDim.x);
}

// The probe items of the keys edit_items_kernel listed: one thread per probing key (dense waves).
__global__ __launch_bounds__(256) void edit_probe_items_kernel(const uint32_t *__restrict__ urecs,
                                                               const uint32_t *__restrict__ ulens, uint64_t U, KeyShape sh,
                                                               uint32_t d, const uint8_t *__restrict__ probe_mask,
                                                               const uint32_t *__restrict__ probe_count,
                                                               const uint32_t *__restrict__ per_key_incl,
                                                               const uint32_t *__restrict__ probers, uint64_t n_probers,
                                                               bool segment_hashes, uint32_t *__restrict__ hashes,
                                                               uint32_t *__restrict__ payloads)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_probers)
        return;
    const uint32_t u = probers[i];
    const uint32_t len = fqd_key_len(sh, ulens, u);
    const uint32_t K = sh.planes, W = sh.words, nseg = d + 1;
    const uint32_t *rec = urecs + (uint64_t)u * sh.stride;
    const uint32_t mine = probe_count[len];
    uint64_t at = U * nseg + (per_key_incl[u] - mine);
    for (int j = 0; j <= 2 * (int)d; j++) {
        if (!((probe_mask[len] >> j) & 1u))
            continue;
        const uint32_t l = (uint32_t)((int)len - (int)d + j);
        for (uint32_t s = 0; s < nseg; s++) {
            uint32_t lo, hi;
            fqd_segment(l, s, nseg, lo, hi);
            const uint32_t nb = hi - lo;
            for (int delta = -(int)d; delta <= (int)d; delta++) {
                if (l == len && delta == 0)
                    continue;
                const int start = (int)lo + delta;
                if (start < 0 || (uint32_t)start + nb > len)
                    continue;
                hashes[at] = segment_hashes ? shifted_segment_hash(rec, K, W, l, s, nseg, delta)
                                            : substring_hash(rec, K, W, l, s, (uint32_t)start, nb);
                payloads[at] = eg_payload(u, s, delta, true);
                at++;
            }
        }
    }
}

// bases [sa, sa + nb) of a  ==  bases [sb, sb + nb) of b ?
__device__ __forceinline__ bool eg_same_substring(const uint32_t *__restrict__ a, uint32_t sa,
                                                  const uint32_t *__restrict__ b, uint32_t sb, uint32_t nb, uint32_t K,
                                                  uint32_t W)
{
    for (uint32_t off = 0; off < nb; off += 32) {
        const uint32_t rem = nb - off;
        const uint32_t mask = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
        for (uint32_t k = 0; k < K; k++)
            if ((plane_bits(a, K, W, k, sa + off) ^ plane_bits(b, K, W, k, sb + off)) & mask)
                return false;
    }
    return true;
}

// Does configuration (prober a -> indexed b, segment s of b's class, shift delta) match?
__device__ __forceinline__ bool eg_config_matches(const uint32_t *__restrict__ ra, uint32_t la,
                                                  const uint32_t *__restrict__ rb, uint32_t lb, uint32_t s, int delta,
                                                  uint32_t d, uint32_t K, uint32_t W)
{
    uint32_t lo, hi;
    fqd_segment(lb, s, d + 1, lo, hi);
    const int start = (int)lo + delta;
    if (start < 0 || (uint32_t)start + (hi - lo) > la)
        return false;
    return eg_same_substring(ra, (uint32_t)start, rb, lo, hi - lo, K, W);
}

// bases a[i...] and b[j...] agree for how many positions? (bit planes: XOR, OR over the planes, count zeros)
__device__ __forceinline__ uint32_t eg_lcp(const uint32_t *__restrict__ a, uint32_t i, uint32_t la,
                                           const uint32_t *__restrict__ b, uint32_t j, uint32_t lb, uint32_t K, uint32_t W)
{
    const uint32_t n = (la - i) < (lb - j) ? la - i : lb - j;
    for (uint32_t off = 0; off < n; off += 32) {
        uint32_t x = 0;
        for (uint32_t k = 0; k < K; k++)
            x |= plane_bits(a, K, W, k, i + off) ^ plane_bits(b, K, W, k, j + off);
        if (x) {
            const uint32_t t = off + (uint32_t)__ffs((int)x) - 1u;
            return t < n ? t : n;
        }
    }
    return n;
}

// Levenshtein(a, b) <= d for d <= 3 by diagonals (Landau-Vishkin): with e edits, how far down
// diagonal k = j - i does a match run? Every extension is an eg_lcp over whole words -- a few dozen
// word operations per pair where the banded table of within_edit touches every base three times
// through a scratch array (272 ms instead of ~2 for the 11 M candidates of config 5's variant).
// The same predicate as distances.h:33-88.
__device__ bool eg_within_edit(const uint32_t *__restrict__ a, uint32_t la, const uint32_t *__restrict__ b, uint32_t lb,
                               int d, uint32_t K, uint32_t W)
{
    const int goal = (int)lb - (int)la;                  // the diagonal both keys end on
    if (goal > d || -goal > d)
        return false;
    constexpr int NONE = -1000000;
    int prev[7], cur[7];                                 // diagonals -3 .. 3 at index k + 3
#pragma unroll
    for (int t = 0; t < 7; t++)
        prev[t] = NONE;
    prev[3] = (int)eg_lcp(a, 0, la, b, 0, lb, K, W);
    if (goal == 0 && prev[3] >= (int)la)
        return true;
    for (int e = 1; e <= d; e++) {
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int k = t - 3;
            cur[t] = NONE;
            if (k < -e || k > e)
                continue;
            int row = NONE;
            if (prev[t] != NONE)
                row = prev[t] + 1;                                   // substitution
            if (t + 1 < 7 && prev[t + 1] != NONE && prev[t + 1] + 1 > row)
                row = prev[t + 1] + 1;                               // a base of a skipped
            if (t > 0 && prev[t - 1] != NONE && prev[t - 1] > row)
                row = prev[t - 1];                                   // a base of b skipped
            if (row == NONE)
                continue;
            if (row > (int)la)
                row = (int)la;
            if (row + k > (int)lb)
                row = (int)lb - k;
            if (row < 0 || row + k < 0)
                continue;
            row += (int)eg_lcp(a, (uint32_t)row, la, b, (uint32_t)(row + k), lb, K, W);
            cur[t] = row;
            if (k == goal && row >= (int)la)
                return true;
        }
#pragma unroll
        for (int t = 0; t < 7; t++)
            prev[t] = cur[t];
    }
    return false;
}

// One thread per candidate (payload, payload). n_lists lists of list_cap candidates, counters 8 words apart
// (group.hip's layout).
__global__ __launch_bounds__(256) void edit_grouped_verify_kernel(
    const uint2 *__restrict__ cands, const unsigned long long *__restrict__ cand_count, uint64_t list_cap,
    uint32_t n_lists, const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t d,
    const uint8_t *__restrict__ probe_mask, uint32_t *__restrict__ edges, unsigned long long *__restrict__ edge_count,
    uint64_t edge_cap, unsigned long long *__restrict__ cand_need, unsigned long long *__restrict__ n_verified,
    int cross_only /* pairs of ONE length are somebody else's (d = 1: the Hamming passes have reported them) */)
{
    const uint32_t list = blockIdx.x % n_lists, part = blockIdx.x / n_lists, parts = gridDim.x / n_lists;
    const unsigned long long filled = cand_count[(size_t)list * 8];
    const unsigned long long total = filled < list_cap ? filled : list_cap;
    if (part == 0 && threadIdx.x == 0 && filled > list_cap)
        atomicMax(cand_need, filled * n_lists);
    cands += (size_t)list * list_cap;
    const uint32_t K = sh.planes, W = sh.words;
    unsigned long long verified = 0;
    for (unsigned long long base = (unsigned long long)part * blockDim.x; base < total;
         base += (unsigned long long)parts * blockDim.x) {
        const unsigned long long idx = base + threadIdx.x;
        bool hit = false;
        uint32_t eu = 0, ev = 0;
        if (idx < total) {
            const uint2 c = cands[idx];
            uint32_t pa = c.x, pb = c.y;
            const bool probe_a = (pa & ROLE_PROBE) != 0, probe_b = (pb & ROLE_PROBE) != 0;
            uint32_t ua = pa & EG_UID_MASK, ub = pb & EG_UID_MASK;
            if (ua != ub && !(probe_a && probe_b)) {
                // (prober, indexed): the probe item's owner probes; two index items: the smaller uid "probes"
                if (probe_b || (!probe_a && ub < ua)) {
                    const uint32_t t = pa;
                    pa = pb;
                    pb = t;
                    const uint32_t tu = ua;
                    ua = ub;
                    ub = tu;
                }
                // (sh.ragged & 2: a row's last word is its key's length -- the rows are about to be read anyway, a look-up in
                // ulens[] is a sector of its own per key)
                const uint32_t la = (sh.ragged & 2u) ? urecs[((uint64_t)ua + 1) * sh.stride - 1] : fqd_key_len(sh, ulens, ua);
                const uint32_t lb = (sh.ragged & 2u) ? urecs[((uint64_t)ub + 1) * sh.stride - 1] : fqd_key_len(sh, ulens, ub);
                const uint32_t gap = la > lb ? la - lb : lb - la;
                const uint32_t seg = (pa >> EG_UID_BITS) & 3u;
                const int delta = (int)((pa >> (EG_UID_BITS + 2)) & 7u) - 3;
                const bool both_index = !((pa | pb) & ROLE_PROBE);
                // an index item of the other segment number cannot be this probe's partner (hash collision)
                const bool seg_ok = ((pb >> EG_UID_BITS) & 3u) == seg;
                if (gap <= d && seg_ok && (!both_index || la == lb) && !(cross_only && la == lb)) {
                    const uint32_t *ra = urecs + (uint64_t)ua * sh.stride, *rb = urecs + (uint64_t)ub * sh.stride;
                    // the first configuration under which the pair matches, in the order: direction
                    // (smaller uid probing first), segment, shift -- among the configurations that are filed
                    const uint32_t x = ua < ub ? ua : ub;
                    bool mine_is_first = false, found = false;
                    for (int dir = 0; dir < 2 && !found; dir++) {
                        const bool a_probes = (dir == 0) == (ua == x);        // this direction has `a` as the prober
                        const uint32_t *rp = a_probes ? ra : rb, *ri = a_probes ? rb : ra;
                        const uint32_t lp = a_probes ? la : lb, li = a_probes ? lb : la;
                        const bool same_class = lp == li;
                        if (!same_class && !eg_probes(probe_mask, lp, li, d))
                            continue;
                        for (uint32_t s = 0; s <= d && !found; s++)
                            for (int dl = -(int)d; dl <= (int)d && !found; dl++) {
                                if (same_class) {
                                    if (dl == 0 ? dir == 1 : !eg_probes(probe_mask, lp, li, d))
                                        continue;      // shift 0: the index-index match, counted once; shifts != 0: filed for d >= 2
                                }
                                if (eg_config_matches(rp, lp, ri, li, s, dl, d, K, W)) {
                                    found = true;
                                    mine_is_first = a_probes && s == seg && dl == delta;
                                }
                            }
                    }
                    if (mine_is_first) {
                        verified++;
                        hit = eg_within_edit(ra, la, rb, lb, (int)d, K, W);
                        eu = ua < ub ? ua : ub;
                        ev = ua < ub ? ub : ua;
                    }
                }
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
                    edges[2 * at] = eu;
                    edges[2 * at + 1] = ev;
                }
            }
        }
    }
    for (int o = 32; o; o >>= 1)
        verified += __shfl_xor(verified, o);
    if (fqd_lane() == 0 && verified)
        atomicAdd(n_verified, verified);
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

hipError_t launch_edit_len_counts(const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t *counts, hipStream_t st)
{
    if (U)
        edit_len_counts_kernel<<<(unsigned)((U + 4095) / 4096), 256, 0, st>>>(ulens, U, sh, counts);
    return hipGetLastError();
}

hipError_t launch_edit_items(const uint32_t *urecs, const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t d,
                             const uint8_t *probe_mask, const uint32_t *probe_count, uint32_t *per_key,
                             const uint32_t *per_key_incl, uint32_t *hashes, uint32_t *payloads, int pass, hipStream_t st,
                             const uint32_t *index_from, uint32_t *probers, unsigned long long *n_probers_dev,
                             uint64_t n_probers)
{
    if (!U)
        return hipSuccess;
    const unsigned light_grid = (unsigned)((U + 256 * EL_KEYS - 1) / (256 * EL_KEYS));
    if (pass == 0) {
        edit_probe_counts_kernel<<<light_grid, 256, 0, st>>>(ulens, U, sh, probe_count, per_key);
        return hipGetLastError();
    }
    if (index_from && probers) {
        edit_index_items_kernel<<<light_grid, 256, 0, st>>>(ulens, U, sh, d, probe_count, index_from, hashes, payloads,
                                                            probers, n_probers_dev);
        if (n_probers)
            edit_probe_items_kernel<<<grid_for(n_probers), 256, 0, st>>>(urecs, ulens, U, sh, d, probe_mask, probe_count,
                                                                        per_key_incl, probers, n_probers, true, hashes,
                                                                        payloads);
        return hipGetLastError();
    }
    const size_t lds = (size_t)256 * (sh.stride + 1) * 4;
    if (pass == 1 && !index_from && sh.stride % 4 == 0 && lds <= 48 * 1024)
        edit_items_kernel<true><<<grid_for(U), 256, lds, st>>>(urecs, ulens, U, sh, d, probe_mask, probe_count, per_key,
                                                               per_key_incl, hashes, payloads, pass, nullptr, probers,
                                                               n_probers_dev);
    else
        edit_items_kernel<false><<<grid_for(U), 256, 0, st>>>(urecs, ulens, U, sh, d, probe_mask, probe_count, per_key,
                                                              per_key_incl, hashes, payloads, pass, index_from, probers,
                                                              n_probers_dev);
    if (pass == 1 && probers && n_probers)
        edit_probe_items_kernel<<<grid_for(n_probers), 256, 0, st>>>(urecs, ulens, U, sh, d, probe_mask, probe_count,
                                                                    per_key_incl, probers, n_probers, index_from != nullptr,
                                                                    hashes, payloads);
    return hipGetLastError();
}

hipError_t launch_edit_grouped_verify(const uint64_t *cands, const unsigned long long *cand_count, uint64_t list_cap,
                                      uint32_t n_lists, const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                                      uint32_t d, const uint8_t *probe_mask, uint32_t *edges,
                                      unsigned long long *edge_count, uint64_t edge_cap, unsigned long long *cand_need,
                                      unsigned long long *n_verified, int cross_only, hipStream_t st)
{
    // (Staging both records of a candidate in LDS first -- 128 candidates per workgroup, 8 lanes fetching a record --
    // was slower: 4.2 instead of 3.3 ms at config 5's variant; at 34 KB per workgroup too few waves are left to hide
    // the gathers.)
    edit_grouped_verify_kernel<<<n_lists * 32, 256, 0, st>>>(reinterpret_cast<const uint2 *>(cands), cand_count, list_cap,
                                                             n_lists, urecs, ulens, sh, d, probe_mask, edges, edge_count,
                                                             edge_cap, cand_need, n_verified, cross_only);
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
