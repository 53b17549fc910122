// trieorder.hip -- what the reference's pointer trie can tell about itself, computed from the flat
// unique table instead (there is no trie on the device):
//
//   * the ORDER in which the reference visits keys: Trie.pop_cluster seeds every cluster with the
//     leftmost key (TrieNode_GetSequence, _triemodule.c:510-551: children in alphabet-index order,
//     a child before the node's own count, i.e. a longer key before its own prefix). Here: an LSD
//     radix sort of the unique keys on digits (alphabet index of the base, END = largest digit);
//   * the node census Trie.memory_size / Trie.raw_stats walk the trie for (_triemodule.c:553-594,
//     909-964), from the common-prefix lengths of neighbours in that order: an inner node exists
//     for every prefix shared by two distinct keys, its child array is as wide as its highest
//     child index + 1 (TrieNode_Resize :136-161 never shrinks), a key that is nobody's prefix ends
//     in a leaf one level below its deepest shared prefix (TrieNode_AddSequence :222-288);
//   * the clusters as a CSR in pop order (members in the same order), for a host that wants
//     pop_cluster's output from the ABI alone.
//
// Introspection and iteration, not the hot path: clarity over the last microsecond.
#include "fqd_internal.h"

namespace {

constexpr uint32_t NO_LCP = 0xFFFFFFFFu;     // "no neighbour on this side" (acts as -1)

// digit of position p of a key: alphabet index of its base, `end_digit` behind the key's end
__device__ __forceinline__ uint32_t digit_at(const uint32_t *__restrict__ rec, uint32_t len, uint32_t p, uint32_t K,
                                             const uint8_t *__restrict__ idx_of_code, uint32_t end_digit)
{
    if (p >= len)
        return end_digit;
    return idx_of_code[fqd_code_at(rec, p >> 5, p & 31u, K)];
}

// sort key of chunk [p0, p0 + P) of the key at position i of the current order: first base in the
// most significant digit
__global__ void chunk_keys_kernel(const uint32_t *__restrict__ order, uint64_t U, const uint32_t *__restrict__ urecs,
                                  const uint32_t *__restrict__ ulens, KeyShape sh, const uint8_t *__restrict__ idx_of_code,
                                  uint32_t end_digit, uint32_t bits, uint32_t p0, uint32_t P,
                                  unsigned long long *__restrict__ keys)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U)
        return;
    const uint32_t u = order[i];
    const uint32_t *rec = urecs + (uint64_t)u * sh.stride;
    const uint32_t len = fqd_key_len(sh, ulens, u);
    unsigned long long k = 0;
    for (uint32_t j = 0; j < P; j++)
        k = (k << bits) | digit_at(rec, len, p0 + j, sh.planes, idx_of_code, end_digit);
    keys[i] = k;
}

__global__ void iota_kernel(uint32_t *out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (uint32_t)i;
}

__global__ void rank_of_kernel(const uint32_t *__restrict__ order, uint64_t U, uint32_t *__restrict__ rank)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < U)
        rank[order[r]] = (uint32_t)r;
}

// common prefix (in bases) of two keys given as records
__device__ __forceinline__ uint32_t common_prefix(const uint32_t *__restrict__ a, uint32_t la,
                                                  const uint32_t *__restrict__ b, uint32_t lb, uint32_t K, uint32_t W)
{
    const uint32_t lim = la < lb ? la : lb;
    for (uint32_t w = 0; w * 32u < lim; w++) {
        const uint32_t d = fqd_diff_word_dyn(a, b, w, K);
        if (d) {
            const uint32_t pos = w * 32u + (uint32_t)__ffs((int)d) - 1u;
            return pos < lim ? pos : lim;
        }
    }
    (void)W;
    return lim;
}

// lcp[r] = common prefix of the keys at ranks r - 1 and r; lcp[0] = lcp[U] = NO_LCP
__global__ void lcp_kernel(const uint32_t *__restrict__ order, uint64_t U, const uint32_t *__restrict__ urecs,
                           const uint32_t *__restrict__ ulens, KeyShape sh, uint32_t *__restrict__ lcp)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > U)
        return;
    if (r == 0 || r == U) {
        lcp[r] = NO_LCP;
        return;
    }
    const uint32_t a = order[r - 1], b = order[r];
    lcp[r] = common_prefix(urecs + (uint64_t)a * sh.stride, fqd_key_len(sh, ulens, a), urecs + (uint64_t)b * sh.stride,
                           fqd_key_len(sh, ulens, b), sh.planes, sh.words);
}

// last_alive[r] + 1 = (rank of the last alive key at or before rank r) + 1, 0 = none: input of a
// running maximum
__global__ void alive_mark_kernel(const uint32_t *__restrict__ order, uint64_t U, const uint8_t *__restrict__ alive,
                                  uint32_t *__restrict__ mark)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < U)
        mark[r] = alive[order[r]] ? (uint32_t)r + 1u : 0u;
}

// The census. One thread per rank r (key k of length l); prev = lcp[r], next = lcp[r + 1] as
// signed numbers (-1 = no neighbour). With every key alive:
//   * k is a proper prefix of another key  <=>  prev == l (the longer keys sort before it): k is
//     counted on the inner node of its own depth and has no leaf. Otherwise k ends in a leaf at
//     depth m = max(prev, next) + 1 holding l - m suffix bytes (the only key: a root leaf, m = 0);
//   * the inner nodes CLOSED at r -- k is the last key below them -- are its prefixes of depth j,
//     next < j <= prev. The widest child slot of such a node belongs to the last key below it
//     that is longer than j: k itself, or k's predecessor when k IS the node (l == j).
// With an alive mask (keys removed by pop_cluster; TrieNode_DeleteSequence :301-363 frees a leaf,
// then every ancestor left without children, and turns a childless counted node into an empty
// leaf): leaves of removed keys are gone; an inner node survives iff some alive key strictly
// extends its prefix -- the last alive key at or before the node's widest child tells, by its
// common prefix with k; child arrays keep their width; an alive key counted on a node that did
// not survive sits in an empty leaf at its own depth.
__global__ __launch_bounds__(256) void census_kernel(
    const uint32_t *__restrict__ order, const uint32_t *__restrict__ lcp, uint64_t U,
    const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh,
    const uint8_t *__restrict__ idx_of_code, const uint8_t *__restrict__ alive /* by uid, or NULL */,
    const uint32_t *__restrict__ last_alive /* by rank, +1; NULL without alive */, uint32_t n_layers, uint32_t n_cols,
    unsigned long long *__restrict__ stats /* n_layers x n_cols */, unsigned long long *__restrict__ memory_size)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bytes = 0;
    if (r < U) {
        const uint32_t u = order[r];
        const uint32_t *rec = urecs + (uint64_t)u * sh.stride;
        const uint32_t l = fqd_key_len(sh, ulens, u), K = sh.planes;
        const int prev = (int)lcp[r], next = (int)lcp[r + 1];        // NO_LCP reads as -1
        const bool me_alive = !alive || alive[u];
        auto bump = [&](uint32_t layer, uint32_t col) {
            if (layer < n_layers && col < n_cols)
                atomicAdd(&stats[(uint64_t)layer * n_cols + col], 1ull);
        };
        // alive-key look-ups for the survival test
        const uint32_t *prec = nullptr;
        uint32_t pl = 0;
        if (r > 0) {
            const uint32_t pu = order[r - 1];
            prec = urecs + (uint64_t)pu * sh.stride;
            pl = fqd_key_len(sh, ulens, pu);
        }
        int reach_self = -1, reach_pred = -1;   // deepest prefix of k still extended by an alive key at/before r, r - 1
        if (alive) {
            const uint32_t a1 = last_alive[r];
            if (a1) {
                const uint32_t au = order[a1 - 1];
                reach_self = a1 - 1 == r ? (int)l
                                         : (int)common_prefix(rec, l, urecs + (uint64_t)au * sh.stride,
                                                              fqd_key_len(sh, ulens, au), K, sh.words);
            }
            const uint32_t a2 = r > 0 ? last_alive[r - 1] : 0u;
            if (a2) {
                const uint32_t au = order[a2 - 1];
                reach_pred = (int)common_prefix(rec, l, urecs + (uint64_t)au * sh.stride, fqd_key_len(sh, ulens, au), K,
                                                sh.words);
            }
        }
        // inner nodes closed here
        for (int j = next + 1; j <= prev; j++) {
            const bool self_is_node = (uint32_t)j == l;              // k is the node's own key: widest child = predecessor's
            bool survives = true;
            if (alive) {
                // some alive key strictly extends the prefix: one at or before the widest child's
                // key sharing >= j bases with k (the node's own key, if it is k, does not count)
                survives = self_is_node ? reach_pred >= j : reach_self >= j;
            }
            if (!survives)
                continue;
            const uint32_t *wrec = self_is_node ? prec : rec;
            const uint32_t wl = self_is_node ? pl : l;
            const uint32_t width = (uint32_t)digit_at(wrec, wl, (uint32_t)j, K, idx_of_code, 0u) + 1u;
            bump((uint32_t)j, width);
            bytes += 8ull + 8ull * width;
        }
        // the key's own leaf
        if (me_alive) {
            const int m = prev > next ? prev : next;
            if (prev == (int)l && U > 1) {
                // counted on the node of its own depth; an empty leaf if that node is gone
                if (alive && reach_pred < (int)l) {
                    bump(l, 0);
                    bytes += 8ull;
                }
            } else {
                const uint32_t depth = (uint32_t)(m + 1);
                bump(depth, 0);
                bytes += 8ull + (l - depth);
            }
        }
    }
    for (int o = 32; o; o >>= 1)
        bytes += __shfl_xor(bytes, o);
    __shared__ unsigned long long s_bytes[4];
    if (fqd_lane() == 0)
        s_bytes[threadIdx.x >> 6] = bytes;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long sum = s_bytes[0] + s_bytes[1] + s_bytes[2] + s_bytes[3];
        if (sum)
            atomicAdd(memory_size, sum);
    }
}

// ---- clusters in pop order -----------------------------------------------------------------
// seed[label] = smallest rank among the alive members of the component
__global__ void seed_rank_kernel(const uint32_t *__restrict__ labels, const uint32_t *__restrict__ rank, uint64_t U,
                                 const uint8_t *__restrict__ alive, uint32_t *__restrict__ seed)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < U && (!alive || alive[u]))
        atomicMin(&seed[labels[u]], rank[u]);
}

// (seed rank of the key's cluster, the key's own rank): sorting these lists the clusters in pop
// order and every cluster's members in trie order; removed keys sort behind everything
__global__ void member_keys_kernel(const uint32_t *__restrict__ labels, const uint32_t *__restrict__ rank, uint64_t U,
                                   const uint8_t *__restrict__ alive, const uint32_t *__restrict__ seed,
                                   unsigned long long *__restrict__ keys)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U)
        return;
    keys[u] = (!alive || alive[u]) ? ((unsigned long long)seed[labels[u]] << 32) | rank[u] : ~0ull;
}

// heads[i] = 1 where a cluster starts in the sorted member list; members[i] = uid
__global__ void member_heads_kernel(const unsigned long long *__restrict__ sorted, uint64_t U,
                                    const uint32_t *__restrict__ order, uint32_t *__restrict__ heads,
                                    uint32_t *__restrict__ members)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U)
        return;
    const unsigned long long k = sorted[i];
    if (k == ~0ull) {
        heads[i] = 0;
        members[i] = 0xFFFFFFFFu;
        return;
    }
    members[i] = order[(uint32_t)k];
    heads[i] = (i == 0 || (uint32_t)(sorted[i - 1] >> 32) != (uint32_t)(k >> 32)) ? 1u : 0u;
}

// offsets[c] = position of the head of cluster c; offsets[n_clusters] = number of alive members
__global__ void cluster_offsets_kernel(const unsigned long long *__restrict__ sorted, const uint32_t *__restrict__ heads,
                                       const uint32_t *__restrict__ heads_incl, uint64_t U,
                                       unsigned long long *__restrict__ offsets)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= U)
        return;
    if (heads[i])
        offsets[heads_incl[i] - 1] = i;
    const bool last_alive = sorted[i] != ~0ull && (i + 1 == U || sorted[i + 1] == ~0ull);
    if (last_alive)
        offsets[heads_incl[i]] = i + 1;
}

// One thread per (record, 32-base word of the NEW layout): the bases' codes are looked up in the
// old record, mapped to the new alphabet's codes and laid out over the new planes. dst is zeroed
// beforehand (padding words, words past the old key's end).
__global__ void transcode_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ src_lens, uint64_t n,
                                 KeyShape from, KeyShape to, const uint8_t *__restrict__ code_map,
                                 uint32_t *__restrict__ dst, uint32_t *__restrict__ dst_lens)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t / to.words;
    const uint32_t w = (uint32_t)(t % to.words);
    if (i >= n)
        return;
    const uint32_t len = fqd_key_len(from, src_lens, i);
    if (w == 0 && to.ragged && dst_lens)
        dst_lens[i] = len;
    if (w * 32u >= len)
        return;
    const uint32_t *rec = src + i * from.stride;
    uint32_t plane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t hi = len - w * 32u < 32u ? len - w * 32u : 32u;
    for (uint32_t b = 0; b < hi; b++) {
        const uint32_t code = code_map[fqd_code_at(rec, w, b, from.planes)];
        for (uint32_t k = 0; k < to.planes; k++)
            plane[k] |= ((code >> k) & 1u) << b;
    }
    for (uint32_t k = 0; k < to.planes; k++)
        dst[i * to.stride + w * to.planes + k] = plane[k];
}

// ---- the lazy alphabet ------------------------------------------------------------------------
// The reference registers a symbol outside the constructor alphabet when an INNER node first looks
// it up (TrieNode_AddSequence, _triemodule.c:266-273): base j of key k is looked up at the moment
// some other key sharing k's first j bases is in the trie together with k -- not when k is added,
// and never for bases in a leaf's suffix. For a symbol c only the FIRST position j0 of c in a key
// can win (the deeper ones are looked up no earlier), and only in keys whose j0 is at most their
// longest common prefix with another key. A round of the search (api_trie.hip):
//   symbol_candidates_kernel   per wanted symbol: the earliest-added such key later than `after`
//   symbol_depth_kernel        j0 of the chosen key
//   symbol_partner_kernel      the earliest-added OTHER key sharing its first j0 bases
__device__ __forceinline__ uint32_t first_position_of(const uint32_t *__restrict__ rec, uint32_t len, uint32_t code,
                                                      uint32_t K, uint32_t W)
{
    for (uint32_t w = 0; w < W && w * 32u < len; w++) {
        uint32_t m = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t plane = rec[w * K + k];
            m &= ((code >> k) & 1u) ? plane : ~plane;
        }
        const uint32_t left = len - w * 32u;
        if (left < 32u)
            m &= (1u << left) - 1u;
        if (m)
            return w * 32u + (uint32_t)__ffs((int)m) - 1u;
    }
    return 0xFFFFFFFFu;
}

constexpr unsigned long long NO_KEY = ~0ull;

__global__ void symbol_candidates_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ lcp, uint64_t U,
                                         const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens,
                                         KeyShape sh, const unsigned long long *__restrict__ ufirst,
                                         const uint8_t *__restrict__ alive, const uint8_t *__restrict__ codes,
                                         uint32_t n_symbols, const unsigned long long *__restrict__ after,
                                         unsigned long long *__restrict__ cand /* (first id << 32) | uid */)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= U)
        return;
    const uint32_t u = order[r];
    if (alive && !alive[u])
        return;
    const int prev = (int)lcp[r], next = (int)lcp[r + 1];
    const int reach = prev > next ? prev : next;          // deepest base of this key an inner node looks up
    if (reach < 0)
        return;
    const uint32_t *rec = urecs + (uint64_t)u * sh.stride;
    const uint32_t len = fqd_key_len(sh, ulens, u);
    const unsigned long long fid = ufirst[u];
    for (uint32_t s = 0; s < n_symbols; s++) {
        if (codes[s] == 0xFFu || (after[s] != NO_KEY && fid <= after[s]))
            continue;
        const uint32_t j0 = first_position_of(rec, len, codes[s], sh.planes, sh.words);
        if (j0 != 0xFFFFFFFFu && (int)j0 <= reach)
            atomicMin(&cand[s], (fid << 32) | u);
    }
}

__global__ void symbol_depth_kernel(const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens, KeyShape sh,
                                    const uint8_t *__restrict__ codes, uint32_t n_symbols,
                                    const unsigned long long *__restrict__ cand, uint32_t *__restrict__ depth)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_symbols)
        return;
    depth[s] = 0xFFFFFFFFu;
    if (cand[s] == NO_KEY)
        return;
    const uint32_t u = (uint32_t)cand[s];
    depth[s] = first_position_of(urecs + (uint64_t)u * sh.stride, fqd_key_len(sh, ulens, u), codes[s], sh.planes, sh.words);
}

__global__ void symbol_partner_kernel(uint64_t U, const uint32_t *__restrict__ urecs, const uint32_t *__restrict__ ulens,
                                      KeyShape sh, const unsigned long long *__restrict__ ufirst,
                                      const uint8_t *__restrict__ alive, uint32_t n_symbols,
                                      const unsigned long long *__restrict__ cand, const uint32_t *__restrict__ depth,
                                      unsigned long long *__restrict__ partner)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U || (alive && !alive[u]))
        return;
    const uint32_t *rec = urecs + u * sh.stride;
    const uint32_t len = fqd_key_len(sh, ulens, u);
    for (uint32_t s = 0; s < n_symbols; s++) {
        if (cand[s] == NO_KEY)
            continue;
        const uint32_t q = (uint32_t)cand[s];
        if (q == u)
            continue;
        const uint32_t cp = common_prefix(rec, len, urecs + (uint64_t)q * sh.stride, fqd_key_len(sh, ulens, q), sh.planes,
                                          sh.words);
        if (cp >= depth[s])
            atomicMin(&partner[s], ufirst[u]);
    }
}

__global__ void store_remove_kernel(const uint32_t *__restrict__ uids, uint64_t n, uint64_t U, uint8_t *__restrict__ alive,
                                    uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t u = uids[i];
    if (u >= U)
        atomicOr(bad, 1u);
    else
        alive[u] = 0;
}

__global__ void store_weights_kernel(const uint32_t *__restrict__ counts, const uint8_t *__restrict__ alive, uint64_t U,
                                     uint32_t *__restrict__ out)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < U)
        out[u] = (!alive || alive[u]) ? counts[u] : 0u;
}

__global__ void fill_ids_kernel(uint64_t *__restrict__ out, uint64_t n, uint64_t base)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = base + i;
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

namespace fqd {

hipError_t launch_trie_iota(uint32_t *out, uint64_t n, hipStream_t st)
{
    if (n)
        iota_kernel<<<grid_for(n), 256, 0, st>>>(out, n);
    return hipGetLastError();
}

hipError_t launch_trie_chunk_keys(const uint32_t *order, uint64_t U, const uint32_t *urecs, const uint32_t *ulens,
                                  KeyShape sh, const uint8_t *idx_of_code, uint32_t end_digit, uint32_t bits, uint32_t p0,
                                  uint32_t P, unsigned long long *keys, hipStream_t st)
{
    if (U)
        chunk_keys_kernel<<<grid_for(U), 256, 0, st>>>(order, U, urecs, ulens, sh, idx_of_code, end_digit, bits, p0, P,
                                                        keys);
    return hipGetLastError();
}

hipError_t launch_trie_rank_of(const uint32_t *order, uint64_t U, uint32_t *rank, hipStream_t st)
{
    if (U)
        rank_of_kernel<<<grid_for(U), 256, 0, st>>>(order, U, rank);
    return hipGetLastError();
}

hipError_t launch_trie_lcp(const uint32_t *order, uint64_t U, const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                           uint32_t *lcp, hipStream_t st)
{
    lcp_kernel<<<grid_for(U + 1), 256, 0, st>>>(order, U, urecs, ulens, sh, lcp);
    return hipGetLastError();
}

hipError_t launch_trie_alive_mark(const uint32_t *order, uint64_t U, const uint8_t *alive, uint32_t *mark,
                                  hipStream_t st)
{
    if (U)
        alive_mark_kernel<<<grid_for(U), 256, 0, st>>>(order, U, alive, mark);
    return hipGetLastError();
}

hipError_t launch_trie_census(const uint32_t *order, const uint32_t *lcp, uint64_t U, const uint32_t *urecs,
                              const uint32_t *ulens, KeyShape sh, const uint8_t *idx_of_code, const uint8_t *alive,
                              const uint32_t *last_alive, uint32_t n_layers, uint32_t n_cols, unsigned long long *stats,
                              unsigned long long *memory_size, hipStream_t st)
{
    if (U)
        census_kernel<<<grid_for(U), 256, 0, st>>>(order, lcp, U, urecs, ulens, sh, idx_of_code, alive, last_alive,
                                                    n_layers, n_cols, stats, memory_size);
    return hipGetLastError();
}

hipError_t launch_trie_seed_ranks(const uint32_t *labels, const uint32_t *rank, uint64_t U, const uint8_t *alive,
                                  uint32_t *seed, hipStream_t st)
{
    if (U)
        seed_rank_kernel<<<grid_for(U), 256, 0, st>>>(labels, rank, U, alive, seed);
    return hipGetLastError();
}

hipError_t launch_trie_member_keys(const uint32_t *labels, const uint32_t *rank, uint64_t U, const uint8_t *alive,
                                   const uint32_t *seed, unsigned long long *keys, hipStream_t st)
{
    if (U)
        member_keys_kernel<<<grid_for(U), 256, 0, st>>>(labels, rank, U, alive, seed, keys);
    return hipGetLastError();
}

hipError_t launch_trie_member_heads(const unsigned long long *sorted, uint64_t U, const uint32_t *order, uint32_t *heads,
                                    uint32_t *members, hipStream_t st)
{
    if (U)
        member_heads_kernel<<<grid_for(U), 256, 0, st>>>(sorted, U, order, heads, members);
    return hipGetLastError();
}

hipError_t launch_trie_cluster_offsets(const unsigned long long *sorted, const uint32_t *heads, const uint32_t *heads_incl,
                                       uint64_t U, unsigned long long *offsets, hipStream_t st)
{
    if (U)
        cluster_offsets_kernel<<<grid_for(U), 256, 0, st>>>(sorted, heads, heads_incl, U, offsets);
    return hipGetLastError();
}

hipError_t launch_symbol_round(const uint32_t *order, const uint32_t *lcp, uint64_t U, const uint32_t *urecs,
                               const uint32_t *ulens, KeyShape sh, const uint64_t *ufirst, const uint8_t *alive,
                               const uint8_t *codes, uint32_t n_symbols, const unsigned long long *after,
                               unsigned long long *cand, uint32_t *depth, unsigned long long *partner, hipStream_t st)
{
    if (!U || !n_symbols)
        return hipSuccess;
    const unsigned long long *first = reinterpret_cast<const unsigned long long *>(ufirst);
    symbol_candidates_kernel<<<grid_for(U), 256, 0, st>>>(order, lcp, U, urecs, ulens, sh, first, alive, codes, n_symbols,
                                                           after, cand);
    symbol_depth_kernel<<<grid_for(n_symbols), 256, 0, st>>>(urecs, ulens, sh, codes, n_symbols, cand, depth);
    symbol_partner_kernel<<<grid_for(U), 256, 0, st>>>(U, urecs, ulens, sh, first, alive, n_symbols, cand, depth, partner);
    return hipGetLastError();
}

hipError_t launch_store_remove(const uint32_t *uids, uint64_t n, uint64_t U, uint8_t *alive, uint32_t *bad, hipStream_t st)
{
    if (n)
        store_remove_kernel<<<grid_for(n), 256, 0, st>>>(uids, n, U, alive, bad);
    return hipGetLastError();
}

hipError_t launch_store_weights(const uint32_t *counts, const uint8_t *alive, uint64_t U, uint32_t *out, hipStream_t st)
{
    if (U)
        store_weights_kernel<<<grid_for(U), 256, 0, st>>>(counts, alive, U, out);
    return hipGetLastError();
}

hipError_t launch_store_fill_ids(uint64_t *out, uint64_t n, uint64_t base, hipStream_t st)
{
    if (n)
        fill_ids_kernel<<<grid_for(n), 256, 0, st>>>(out, n, base);
    return hipGetLastError();
}

hipError_t launch_transcode_records(const uint32_t *src, const uint32_t *src_lens, uint64_t n, KeyShape from, KeyShape to,
                                    const uint8_t *code_map, uint32_t *dst, uint32_t *dst_lens, hipStream_t st)
{
    if (!n || !to.words)
        return hipSuccess;
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)n * to.stride * 4, st);
    if (e != hipSuccess)
        return e;
    transcode_kernel<<<grid_for(n * to.words), 256, 0, st>>>(src, src_lens, n, from, to, code_map, dst, dst_lens);
    return hipGetLastError();
}

}  // namespace fqd
