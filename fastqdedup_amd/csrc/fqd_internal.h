// fqd_internal.h -- shared between the translation units of libfqdedup_hip.so.
//
// Record layout (DESIGN.md "data layout"): a key of `len` bases over an alphabet
// of A symbols is stored as K = ceil(log2 A) bit planes. Symbol codes are the
// ranks of the symbols in ASCII order, so comparing codes compares bytes the way
// Python's str ordering does (reference __init__.py:68,99,111 sort (count, key)).
// Base p lives in bit (p & 31) of 32-base word (p >> 5): LSB-first, which is what
// a wave64 __ballot produces. Word w of plane k is rec[w * K + k]; positions
// >= len hold code 0; words past K * W up to `stride` (a multiple of 4 u32, so
// records are 16-byte aligned) are 0 -- except that a RAGGED record with such a padding word may carry its key's
// length in the LAST one (pack.hip, fqd_ctx::recs_len_pad: comparing two whole records then compares the keys, no
// look-up in lens[]), and a record on its way to another rank the sender's read index in the FIRST one (IdSource).
// Nothing that reads key words looks past K * W.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct KeyShape {
    uint32_t planes;     // K
    uint32_t words;      // W = ceil(max_len / 32)
    uint32_t stride;     // u32 words per record, multiple of 4
    uint32_t max_len;
    uint32_t ragged;     // 1: per-key lengths in lens[]; 0: every key is max_len long
};

#define FQD_WAVE 64

// ---- device helpers ---------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ uint64_t fqd_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t fqd_mix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}

__device__ __forceinline__ uint32_t fqd_key_len(const KeyShape &sh, const uint32_t *lens, uint64_t i)
{
    return sh.ragged ? lens[i] : sh.max_len;
}

// The 32-bit key hash the collapse sorts by (and that picks a key's owner rank). It only
// PROPOSES equality: runs of equal hash are verified record by record (collapse.hip), so
// its quality trades collision-run fix-ups against cycles, never exactness.
__device__ __forceinline__ uint32_t fqd_hash_record(const uint32_t *rec, uint32_t n_words, uint32_t len)
{
    uint32_t h = 0x9E3779B9u ^ (len * 0x85EBCA6Bu);
    for (uint32_t j = 0; j < n_words; j++) {
        h = (h ^ rec[j]) * 0x9E3779B1u;
        h ^= h >> 15;
    }
    return fqd_mix32(h);
}

// Mismatching positions of word w between records a and b (one bit per base).
__device__ __forceinline__ uint32_t fqd_diff_word_dyn(const uint32_t *a, const uint32_t *b, uint32_t w,
                                                      uint32_t K)
{
    uint32_t d = 0;
    for (uint32_t k = 0; k < K; k++)
        d |= a[w * K + k] ^ b[w * K + k];
    return d;
}

__device__ __forceinline__ uint32_t fqd_code_at(const uint32_t *rec, uint32_t w, uint32_t bit, uint32_t K)
{
    uint32_t c = 0;
    for (uint32_t k = 0; k < K; k++)
        c |= ((rec[w * K + k] >> bit) & 1u) << k;
    return c;
}

// Python str order of two DIFFERENT keys given as records: >0 when a > b.
__device__ __forceinline__ int fqd_key_cmp(const uint32_t *a, uint32_t la, const uint32_t *b, uint32_t lb,
                                           uint32_t K, uint32_t W)
{
    for (uint32_t w = 0; w < W; w++) {
        uint32_t d = fqd_diff_word_dyn(a, b, w, K);
        if (d) {
            uint32_t bit = (uint32_t)__ffs((int)d) - 1u;
            uint32_t pos = w * 32u + bit;
            // a position past the end of the shorter key: the shorter key is a proper
            // prefix up to here only if no earlier difference -- which is the case.
            if (pos >= la || pos >= lb)
                return la < lb ? -1 : 1;
            return fqd_code_at(a, w, bit, K) < fqd_code_at(b, w, bit, K) ? -1 : 1;
        }
    }
    return la < lb ? -1 : (la > lb ? 1 : 0);
}

// Bits of 32-base word w that fall inside base range [lo, hi).
__device__ __forceinline__ uint32_t fqd_range_mask(uint32_t w, uint32_t lo, uint32_t hi)
{
    uint32_t w0 = w * 32u, w1 = w0 + 32u;
    uint32_t a = lo > w0 ? lo : w0, b = hi < w1 ? hi : w1;
    if (a >= b)
        return 0u;
    uint32_t n = b - a, s = a - w0;
    uint32_t m = n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u);
    return m << s;
}

// Segment s of d+1 for a key of len bases: [len*s/(d+1), len*(s+1)/(d+1))  
__device__ __forceinline__ void fqd_segment(uint32_t len, uint32_t s, uint32_t nseg, uint32_t &lo, uint32_t &hi)
{
    // 32-bit on purpose (a 64-bit divide is a long software routine): len <= ~150 k bases
    // (pack tile limit) and nseg <= 65, so len * (s + 1) < 2^32
    lo = len * s / nseg;
    hi = len * (s + 1) / nseg;
}

// Hash of segment s of the nseg-way split of one record, by ONE thread; the same value as
// segment_hashes_kernel's cooperative sum (edges.hip): fmix(len, s, SUM over the record's words
// of mix(word & segment mask, word index)).
__device__ __forceinline__ uint32_t fqd_segment_hash(const uint32_t *rec, uint32_t K, uint32_t KW, uint32_t len,
                                                     uint32_t s, uint32_t nseg)
{
    uint32_t lo, hi;
    fqd_segment(len, s, nseg, lo, hi);
    uint32_t part = 0;
    for (uint32_t j = 0; j < KW; j++) {
        const uint32_t m = fqd_range_mask(j / K, lo, hi);
        if (m)
            part += fqd_mix32((rec[j] & m) + (j + 1u) * 0x9E3779B1u);
    }
    return fqd_mix32(part + fqd_mix32(len * 0x9E3779B1u + s * 0x85EBCA77u + 0x165667B1u));
}

__device__ __forceinline__ uint32_t fqd_lane() { return threadIdx.x & 63u; }

// Where the id of the read at position `pos` of the packed buffer comes from: an explicit id
// array, or (reads received from several ranks, fqd_collapse_received) the local index that the
// sender stamped into the record's first padding word plus the id base of the sender's segment,
// or the position itself.
struct IdSource {
    const uint64_t *ids64 = nullptr;
    const uint32_t *stamped = nullptr;    // packed records; word `spare_word` of each holds a local index
    uint32_t stride = 0, spare_word = 0;
    const uint32_t *seg_rows = nullptr;   // n_seg + 1 row offsets of the segments (device)
    const uint64_t *seg_id0 = nullptr;    // n_seg id bases (device)
    uint32_t n_seg = 0;
    // LDS collapse of received reads without weights: the partition stamps (segment << packed_bits |
    // local index) into the travelling record instead of the position -- the same order, since the
    // segments arrive in rank order -- and the id needs no look-up in the packed buffer afterwards.
    uint32_t packed_bits = 0;             // 0: not possible / not used
    // slabs received from several senders (fqd_collapse_owner_slabs): the number stamped for sender v of the
    // buffer (NULL: v itself) -- the senders' ranks by id base, so that the smallest stamped word of a key is its
    // first holder also when a rank's reads arrive as several senders out of id order (chunk by chunk)
    const uint32_t *stamp_map = nullptr;
    // first rows of segments 1..7 (up to 8 ranks; unused = ~0): SCALAR members on purpose -- with
    // an array member the partition kernel's argument struct was no longer split into registers
    // and the whole kernel went through scratch memory (0.49 -> 0.70 ms)
    uint32_t row1 = ~0u, row2 = ~0u, row3 = ~0u, row4 = ~0u, row5 = ~0u, row6 = ~0u, row7 = ~0u;
    __device__ __forceinline__ uint32_t packed_segment_of(uint32_t pos) const
    {
        return (pos >= row1 ? 1u : 0u) + (pos >= row2 ? 1u : 0u) + (pos >= row3 ? 1u : 0u) + (pos >= row4 ? 1u : 0u) +
               (pos >= row5 ? 1u : 0u) + (pos >= row6 ? 1u : 0u) + (pos >= row7 ? 1u : 0u);
    }
    __device__ __forceinline__ uint32_t segment_of(uint32_t pos) const
    {
        uint32_t a = 0, b = n_seg;        // last segment with seg_rows[a] <= pos
        while (b - a > 1) {
            const uint32_t m = (a + b) >> 1;
            if (seg_rows[m] <= pos)
                a = m;
            else
                b = m;
        }
        return a;
    }
    __device__ __forceinline__ uint64_t from_packed(uint32_t tag) const
    {
        return seg_id0[tag >> packed_bits] + (tag & ((1u << packed_bits) - 1u));
    }
    __device__ __forceinline__ uint64_t at(uint32_t pos) const
    {
        if (ids64)
            return ids64[pos];
        if (!stamped)
            return pos;
        return seg_id0[segment_of(pos)] + stamped[(uint64_t)pos * stride + spare_word];
    }
};

__device__ __forceinline__ uint64_t fqd_lanemask_lt()
{
    return (1ull << fqd_lane()) - 1ull;
}

#endif  // __HIPCC__

// ---- launchers (one per translation unit; all asynchronous on `st`) ---------
namespace fqd {

// prims.hip -- rocPRIM device-wide primitives
size_t sort_pairs_u32_u32_temp(uint64_t n, int begin_bit, int end_bit);
hipError_t sort_pairs_u32_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, uint64_t n, int begin_bit,
                              int end_bit, hipStream_t st);
size_t sort_keys_u64_temp(uint64_t n);
hipError_t sort_keys_u64(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, uint64_t n,
                         int end_bit, hipStream_t st);
size_t scan_u32_temp(uint64_t n);
hipError_t inclusive_scan_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n,
                              hipStream_t st);
size_t sort_pairs_u64_u32_temp(uint64_t n, int end_bit);
hipError_t sort_pairs_u64_u32(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                              uint32_t *vout, uint64_t n, int end_bit, hipStream_t st);
size_t scan_max_u32_temp(uint64_t n);
hipError_t inclusive_scan_max_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n,
                                  hipStream_t st);

// pack.hip
hipError_t launch_scan_bytes(const uint8_t *bytes, uint64_t n_bytes, uint32_t *present128_dev,
                             hipStream_t st);
hipError_t launch_scan_lens(const uint64_t *offsets, uint64_t n, uint32_t *minmax_dev, hipStream_t st);
// owner rule (own_parts != 0): owners[i] = hash(segment own_seg of own_nseg) % own_parts, in the
// same pass (the record is in LDS anyway); owners may be NULL.
struct OwnerRule {
    uint32_t parts = 0, nseg = 1, seg = 0;
};
// pack fused with level 1 of the LDS collapse (pack.hip, FUSED): the tile goes out partitioned
struct PackScatter {
    uint32_t *cursor;      // n_bins * subs part cursors, cursor[p] starts at p * cap
    uint4 *out;            // n_bins * subs * cap records
    uint32_t *overflow;    // bit 4: some part's slab was full
    uint32_t shift, n_bins, subs, cap;
    uint32_t tables_at;    // set by launch_pack: LDS word offsets of the partition tables (off | base | wave | bin16)
    uint32_t hist_at;      // ... and of the bin counts, which live across the workgroup's tiles
    // owner-major bins (multi-GPU): bin = owner * owner_hb + top log2(owner_hb) hash bits, owner by the
    // OwnerRule handed to launch_pack; n_bins = owner_parts * owner_hb. 0: plain hash bins.
    uint32_t owner_parts = 0, owner_hb = 0;
    // != 0: the records are binned by a hash of SEGMENT 0 of the key alone (route_mask: its bits in the 32-base word)
    // instead of the whole key -- every key sharing segment 0 then ends in one bucket of the collapse, whose
    // compaction finds the pairs of search pass 0 on the spot (Pass0 below)
    uint32_t route_mask = 0;
    // the SPILL list (pack_kernel<.., 2>; a context that has met keys with very many copies, api.hip heavy_keys): a
    // record that finds its part's slab full goes to spill[atomicAdd(spill_cursor, ..)] instead of ending the
    // attempt, and the bin is marked in l1_over[bin]; the list is collapsed with the side path's keys
    // (launch_side_insert_spill) and the dedupe of the bin's buckets merges its rows into those (bucket_dedupe12 MERGE)
    uint4 *spill = nullptr;
    uint32_t *spill_cursor = nullptr;
    uint32_t spill_cap = 0;
    uint32_t *l1_over = nullptr;
    // the reads of this launch are reads id_base .. of the job (a job packed in pieces, each behind its own
    // host-to-device copy: the cursors go on from piece to piece)
    uint32_t id_base = 0;
    uint32_t sub_rot = 0;      // set by launch_pack: the job-wide number of this launch's first workgroup (sub-part = workgroup % subs)
    // compact records (Rec12, squeeze 1): a key with an N leaves HERE for the side slabs (SideSlabs below: side_slabs
    // slabs of side_cap records, cursor[s] starting at s * side_cap; a full slab raises bit 16 of *overflow) instead
    // of travelling through level 1 to be taken out by level 2 -- the side path then runs beside level 2 and has
    // finished (the head of the unique table, the probe lists of search pass 0) before the dedupe starts.
    // NULL: every record goes to its bin.
    uint4 *side_recs = nullptr;
    uint32_t *side_cursor = nullptr;
    uint32_t side_slabs = 0, side_cap = 0;
    uint32_t rare_at = 0;      // set by launch_pack: LDS word offset of the parked keys with an N
};
// The route hash of a one-word key in (a, b) form -- a = p0 | p2, b = p1 | p2 for the three planes of "ACGNT", the two
// planes themselves for a two-plane alphabet -- over the bits of segment 0: what the fused pack (level 1), level 2
// and the side path's probe lists must agree on.
__device__ __forceinline__ uint32_t fqd_route_hash(uint32_t a, uint32_t b, uint32_t route_mask);
// Compact records of the LDS collapse (collapse_lds.hip; single GPU, one 32-base word per plane): level 2 turns the
// pack kernel's uint4 records into 12 bytes -- two key words + the read index -- which the dedupe then reads.
//   squeeze 1: the "ACGNT" code table (A 0, C 1, G 2, N 3, T 4): planes (p0, p1, p2) -> (p0 | p2, p1 | p2), a
//              bijection on N-free keys; a key with an N (p0 & p1 != 0) leaves level 2 as the uint4 it is, through
//              one of the side slabs (SideSlabs), and is collapsed apart (launch_side_collapse)
//   squeeze 2: two planes: the record's two words as they are, no side path
struct Rec12 {
    uint32_t a, b, id;
};
__device__ __forceinline__ uint32_t fqd_hash_rec12(uint32_t a, uint32_t b)
{
    uint32_t h = (a ^ 0x9E3779B9u) * 0x9E3779B1u;
    h ^= h >> 15;
    h = (h ^ b) * 0x85EBCA6Bu;
    return fqd_mix32(h);
}
__device__ __forceinline__ uint32_t fqd_route_hash(uint32_t a, uint32_t b, uint32_t route_mask)
{
    return fqd_hash_rec12(a & route_mask, b & route_mask);
}
// Search pass 0 inside the collapse's compaction (collapse_lds.hip bucket_compact12_kernel): with the reads routed
// by segment 0, a bucket holds EVERY unique key sharing its segment-0 values, so the pairs that agree on segment 0
// and lie within distance d are found among the bucket's rows while they are in registers -- no (hash, uid) items,
// no partition, no candidate list, no record gather for that pass. Keys with an N (side path, head of the unique
// table) are filed in per-bucket probe lists by side_emit_kernel and compared with the bucket's rows there.
constexpr uint32_t FQD_P0_PROBE_CAP = 48;     // side keys per bucket (<= 64; more: *flag, the search runs pass 0 itself). The side
                                              // slabs hold n / 64 keys, ~12 per bucket of 800 reads: 48 is ten sigma above that
struct PairStats;
struct Pass0 {
    uint32_t mask = 0;                  // bits of segment 0 in the key word; 0: no pass 0
    uint32_t d = 0;                     // pairs within this distance
    uint32_t bucket_bits = 0;           // bucket of a key = route hash >> (32 - bucket_bits)
    uint32_t max_rows = 0;              // buckets of more rows than this raise *flag (<= the 512 a wave holds in LDS)
    uint32_t *probe_n = nullptr;        // [n_buckets], zeroed before the side path runs
    uint32_t *probe = nullptr;          // [n_buckets][FQD_P0_PROBE_CAP] uids of side keys
    uint32_t *edges = nullptr;          // (u, v), u < v, appended behind *edge_count
    unsigned long long *edge_count = nullptr;
    uint64_t edge_cap = 0;
    uint32_t *flag = nullptr;           // |= 1: pass 0 is incomplete (a bucket of more rows than the wave's LDS holds, a full probe list)
    PairStats *stats = nullptr;         // stats[slot].edges += pairs reported
};
// Dedupe + compaction of compact records in one persistent kernel (collapse_lds.hip bucket_collapse12_kernel): the
// words its workgroups meet on. Grid = 64 * teams_per_round workgroups, all resident at once; round r = buckets
// [r * grid, (r + 1) * grid).
struct CollapseSync {
    unsigned long long *team = nullptr;   // [n_rounds * teams_per_round], zeroed: arrived workgroups << 32 | their unique keys
    uint32_t *base = nullptr;             // [n_rounds * teams_per_round], zeroed: 1 + the first row of the team's keys (0: not known yet)
    uint32_t *abort = nullptr;            // zeroed; != 0 afterwards: a wait ran into its limit (1) or a bucket had more unique
                                          // keys than the kernel takes (2): nothing of the output counts
    uint32_t *result = nullptr;           // zeroed: the row counter -- afterwards the job's unique keys without the side path's
    uint32_t team_size = 0, teams_per_round = 0, n_rounds = 0;      // grid = team_size * teams_per_round workgroups
    uint64_t wait_ticks = 0;              // limit of one wait in wall_clock64 ticks (100 MHz)
    unsigned long long *prof = nullptr;   // (diagnostic builds, -DFQD_FC_PROF: 16 words of per-phase ticks, zeroed)
};
hipError_t launch_pack(const uint8_t *bytes, uint64_t n_bytes, const uint64_t *offsets, uint64_t n,
                       uint32_t fixed_len, KeyShape sh, const uint8_t *lut_dev, const uint8_t *lut_host,
                       uint32_t *recs, uint32_t *lens, uint32_t *hashes, uint32_t *owners, OwnerRule rule,
                       uint32_t *bad_flag, hipStream_t st, const PackScatter *fused = nullptr);
hipError_t launch_hash_records(const uint32_t *recs, const uint32_t *lens, uint64_t n, KeyShape sh,
                               uint32_t *hashes, hipStream_t st);

// collapse.hip
hipError_t launch_iota_u32(uint32_t *out, uint64_t n, hipStream_t st);
hipError_t launch_head_flags(const uint32_t *hs, const uint32_t *ids, const uint32_t *recs,
                             const uint32_t *lens, uint64_t n, KeyShape sh, uint32_t hash_mask,
                             uint32_t *flags, uint32_t *n_collision_runs, uint32_t *collision_runs,
                             uint32_t cap, hipStream_t st);
hipError_t launch_fix_collision_runs(const uint32_t *hs, uint32_t *ids, const uint32_t *recs,
                                     const uint32_t *lens, uint64_t n, KeyShape sh, uint32_t hash_mask,
                                     uint32_t *flags, const uint32_t *collision_runs, uint32_t n_runs,
                                     hipStream_t st);
hipError_t launch_run_starts(const uint32_t *flags, const uint32_t *run_idx, uint64_t n,
                             uint32_t *run_start, hipStream_t st);
hipError_t launch_run_weights(const uint32_t *run_start, uint32_t n_runs, uint64_t n, const uint32_t *ids,
                              const uint32_t *weights, uint32_t *run_weight, uint32_t *live_flag,
                              hipStream_t st);
hipError_t launch_write_unique(const uint32_t *run_start, const uint32_t *run_weight,
                               const uint32_t *live_flag, const uint32_t *live_idx, uint32_t n_runs,
                               const uint32_t *ids, const uint32_t *recs, const uint32_t *lens,
                               const uint64_t *read_ids, KeyShape sh, uint32_t *urecs, uint32_t *ulens,
                               uint32_t *ucounts, uint64_t *ufirst, hipStream_t st);
hipError_t launch_sum_u32(const uint32_t *in, uint64_t n, unsigned long long *out, hipStream_t st);
hipError_t launch_max_u64(const uint64_t *in, uint64_t n, unsigned long long *out, hipStream_t st);

// collapse_lds.hip -- sort-free collapse for one-uint4 records
hipError_t launch_tile_starts(const uint32_t *seg_start, uint32_t n_seg, uint32_t *tile_start, hipStream_t st);
hipError_t launch_part_hist(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                            const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                            uint32_t n_bins, uint32_t kw, uint32_t len, uint32_t *hist, hipStream_t st);
hipError_t launch_part_scatter(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                               const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                               uint32_t n_bins, uint32_t kw, uint32_t len, uint32_t *cursor, uint32_t *out,
                               hipStream_t st, IdSource packed = IdSource(), uint32_t slab_cap = 0,
                               uint32_t *slab_overflow = nullptr, const uint32_t *seg_end = nullptr,
                               uint32_t seg_shift = 0, uint32_t seg_mask = 0xFFFFFFFFu, uint32_t stamp_div = 0);
// slab segments of reads received from n_senders ranks (fqd_collapse_owner_slabs)
hipError_t launch_owner_slab_bounds(const uint32_t *cursors, uint32_t n_senders, uint32_t parts_per_owner,
                                    uint32_t my_part, uint32_t cap, uint32_t *seg_start, uint32_t *seg_end,
                                    hipStream_t st);
hipError_t launch_slab_tile_starts(const uint32_t *seg_start, const uint32_t *seg_end, uint32_t n_seg,
                                   uint32_t *tile_start, hipStream_t st, bool tiles12 = false /* part_tile_size12() */);
uint32_t part_tile_size();
// bucket_compact_kernel may write the segment hashes of the search that follows (nseg = 0: no)
struct SegHashOut {
    uint32_t *out = nullptr;      // [nseg - first][n_unique]: segments first .. nseg - 1
    uint32_t nseg = 0, planes = 0, kw = 0, len = 0;
    uint32_t first = 0;           // (1: pass 0 is done by the compaction itself, see Pass0)
};
// collapse_pairs.hip -- sort-free collapse for records longer than one uint4
// buckets of more than `slice` pairs are cut into slices (0: never)
struct PairsSlices {
    uint32_t slice = 0, cap = 0;       // items per slice; slices beyond the buckets' first (and long buckets) there is room for
    uint2 *extra = nullptr;            // [cap] (first item, end) of each further slice
    uint32_t *extra_unique = nullptr;  // [cap] rows each of them left
    uint32_t *big = nullptr;           // [cap][3] bucket, its first further slice, how many
    uint32_t *ctr = nullptr;           // [2] long buckets, further slices (zero at launch)
    uint32_t *tags = nullptr;          // the items' array: a row's tag is written over the hash of item (slice start + row)
};
hipError_t launch_bucket_pairs_dedupe(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                      uint32_t n_buckets, const uint32_t *recs, uint32_t stride_words,
                                      const uint32_t *weights, uint32_t *tmp_rep, uint32_t *tmp_count,
                                      uint32_t *tmp_first, uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st,
                                      const uint32_t *lens = nullptr, PairsSlices sl = PairsSlices());
uint32_t pairs_slice_items();
hipError_t launch_clear_last_word(uint32_t *recs, uint64_t n, uint32_t stride, hipStream_t st);
hipError_t launch_bucket_pairs_compact(const uint32_t *bucket_start, const uint32_t *unique_incl, uint32_t n_buckets,
                                       const uint32_t *tmp_rep, const uint32_t *tmp_count, const uint32_t *tmp_first,
                                       const uint32_t *recs, uint32_t stride_words, IdSource read_ids, uint32_t *urecs,
                                       uint32_t *ucounts, uint64_t *ufirst, hipStream_t st,
                                       SegHashOut seg_hashes = SegHashOut(), const uint32_t *lens = nullptr,
                                       uint32_t *ulens = nullptr, uint32_t row_cap = 0xFFFFFFFFu /* more unique keys than
                                       this: the launch writes nothing (the caller queued it before it knew) */,
                                       uint32_t len_hint = 0 /* != 0: ragged records hold their key's length in their last word */);
hipError_t launch_matrix_starts(const uint32_t *matrix_incl, uint32_t n_bins, uint32_t n_tiles, uint32_t *start,
                                hipStream_t st);
hipError_t launch_bucket_starts(const uint32_t *hist_incl, uint32_t n_buckets, uint32_t *bucket_start,
                                uint32_t *cursor, hipStream_t st);
hipError_t launch_slab_starts(uint32_t n_buckets, uint32_t cap, uint32_t *bucket_start, uint32_t *cursor, hipStream_t st);
// up to three slab sets in one launch (start2 / start3 NULL: fewer)
hipError_t launch_slab_starts3(uint32_t n1, uint32_t cap1, uint32_t *start1, uint32_t *cursor1, uint32_t n2, uint32_t cap2,
                               uint32_t *start2, uint32_t *cursor2, uint32_t n3, uint32_t cap3, uint32_t *start3,
                               uint32_t *cursor3, hipStream_t st, uint32_t n_zero = 0, uint32_t *zero = nullptr,
                               uint32_t *zero_b = nullptr, uint32_t last_b = 0, uint32_t *zero_c = nullptr,
                               uint32_t last_c = 0);   // zero[0 .. n_zero], zero_b[0 .. last_b], zero_c[0 .. last_c] = 0
// Buckets of the uint4 dedupe that hold tens of thousands of reads and more (ONE key with very many copies): cut into
// chunks that workgroups reduce side by side, their rows merged by the bucket's own workgroup (collapse_lds.hip)
#define FQD_HUGE_MAX 32        // huge buckets handled that way per launch (the others: one workgroup each, as before)
#define FQD_HUGE_CHUNKS 64     // chunks per huge bucket
struct HugeBuckets {
    uint8_t *slot = nullptr;       // [n_buckets]: 0, or 1 + the bucket's number among the huge ones
    uint32_t *vlo = nullptr, *vhi = nullptr, *vunique = nullptr;   // [FQD_HUGE_MAX * FQD_HUGE_CHUNKS] (+ 1 counter behind vunique)
    uint32_t mode = 0;             // set by launch_bucket_dedupe
};
hipError_t launch_bucket_dedupe(const uint32_t *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                uint32_t n_buckets, const uint32_t *weights, uint32_t *tmp_rec, uint32_t *tmp_count,
                                uint32_t *tmp_first, uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st,
                                HugeBuckets huge = HugeBuckets());
hipError_t launch_bucket_compact(const uint32_t *bucket_start, const uint32_t *unique_incl, uint32_t n_buckets,
                                 const uint32_t *tmp_rec, const uint32_t *tmp_count, const uint32_t *tmp_first,
                                 IdSource read_ids, uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst,
                                 hipStream_t st, SegHashOut seg_hashes = SegHashOut());
// the same three steps with Rec12 items: level 2 over the pack kernel's slabs (uint4 in, Rec12 out), dedupe (tmp
// rows are one uint4: a, b, count, first index), compaction behind the *side_unique keys that
// launch_side_collapse put at the head of the unique table
struct SideSlabs {
    uint4 *recs = nullptr;         // n_slabs * cap records
    uint32_t *cursor = nullptr;    // cursor[s] starts at s * cap (launch_slab_starts)
    uint32_t n_slabs = 0, cap = 0; // n_slabs: a power of two
    uint32_t *overflow = nullptr;  // bit 16: a side slab was full
    // the spill list (PackScatter::spill): recs[spill_at, spill_at + spill_cap), filled up to *spill_cursor; level 2
    // appends the items that find their bucket's slab full (as uint4 records again). spill_cursor == NULL: no list,
    // a full slab ends the attempt.
    uint32_t spill_at = 0, spill_cap = 0;
    uint32_t *spill_cursor = nullptr;
};
hipError_t launch_part_scatter12(const uint32_t *in /* uint4 records */, uint32_t squeeze, SideSlabs side,
                                 const uint32_t *seg_start, const uint32_t *tile_start, uint32_t n_seg,
                                 uint32_t max_tiles, uint32_t shift, uint32_t n_bins, uint32_t *cursor, Rec12 *out,
                                 hipStream_t st, uint32_t slab_cap, uint32_t *slab_overflow, const uint32_t *seg_end,
                                 uint32_t seg_shift, uint32_t route_mask = 0xFFFFFFFFu, uint32_t seg_mask = 0xFFFFFFFFu,
                                 uint32_t stamp_div = 0, uint32_t stamp_shift = 0, const uint32_t *stamp_map = nullptr);
uint32_t part_tile_size12();
uint32_t pass0_max_rows();      // rows of a bucket the compaction's search pass 0 takes (Pass0::max_rows at most)
hipError_t launch_bucket_dedupe12(const Rec12 *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                  uint32_t n_buckets, const uint32_t *weights, uint32_t *tmp_rec,
                                  uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st,
                                  uint32_t *group_total = nullptr, bool big_table = false /* 2048 slots per bucket */);
hipError_t launch_bucket_compact12(const uint32_t *bucket_start, const uint32_t *unique_incl, uint32_t n_buckets,
                                   const uint32_t *tmp_rec, uint32_t squeeze, const uint32_t *side_unique,
                                   uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst, hipStream_t st,
                                   SegHashOut seg_hashes = SegHashOut(), const uint32_t *bucket_unique = nullptr,
                                   const uint32_t *group_total = nullptr, Pass0 pass0 = Pass0(),
                                   IdSource read_ids = IdSource());
// dedupe + compaction (+ search pass 0) in one persistent kernel (CollapseSync above). collapse12_resident(): workgroups
// of it the device holds at once (0: the kernel cannot be used) -- the grid must not be larger, and
// sync.n_rounds * grid >= n_buckets. The side path must have finished (*side_unique,
// the probe lists of pass0); at most ONE array of segment hashes (seg_hashes.nseg - seg_hashes.first <= 1).
uint32_t collapse12_resident();
hipError_t launch_bucket_collapse12(const Rec12 *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                    uint32_t n_buckets, const uint32_t *weights, uint32_t squeeze,
                                    const uint32_t *side_unique, uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst,
                                    hipStream_t st, SegHashOut seg_hashes, Pass0 pass0, IdSource read_ids,
                                    CollapseSync sync, uint32_t *overflow);
// the keys of the side slabs (few: the reads with an N) collapsed through a hash table in global memory and
// written to the head of the unique table; table: side_table_words(table_slots) words (table_slots a power of
// two), cleared here; block_counts = table + 3 * table_slots
uint32_t side_table_words(uint32_t table_slots);
hipError_t launch_side_collapse(const uint4 *side, const uint32_t *cursor /* of the subs side slabs */,
                                uint32_t first_part /* the cursors started at (first_part + sub) * cap */, uint32_t subs, uint32_t cap,
                                const uint32_t *weights, uint32_t *table, uint32_t table_slots, uint32_t *block_counts,
                                uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst, uint32_t *side_unique,
                                uint32_t *overflow, hipStream_t st, Pass0 pass0 = Pass0(), IdSource read_ids = IdSource());

// a context with a spill list (SideSlabs::spill_cursor, PackScatter::spill): launch_side_begin (side slabs + spill
// list -> table), launch_bucket_dedupe12_merge (rows whose key is in the table merge into it), launch_side_finish
// (table -> head of the unique table)
hipError_t launch_side_begin(SideSlabs side, const uint32_t *weights, uint32_t *table, uint32_t table_slots, hipStream_t st);
hipError_t launch_bucket_dedupe12_merge(const Rec12 *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                        uint32_t n_buckets, const uint32_t *weights, uint32_t *tmp_rec,
                                        uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st, uint32_t *group_total,
                                        const uint4 *side, uint32_t *table, uint32_t table_slots, const uint32_t *l1_over,
                                        uint32_t l1_shift);
hipError_t launch_side_finish(const uint4 *side, uint32_t *table, uint32_t table_slots, uint32_t *block_counts,
                              uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst, uint32_t *side_unique, hipStream_t st,
                              Pass0 pass0 = Pass0(), IdSource read_ids = IdSource());

// edges.hip
hipError_t launch_segment_hashes(const uint32_t *urecs, const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t nseg,
                                 uint32_t s_begin, uint32_t s_end, uint32_t mod, uint32_t *seg_hashes,
                                 hipStream_t st);
struct PairStats {
    unsigned long long keys_gathered, pairs_compared, edges;
};
#define FQD_STAT_SLOTS 64
#define FQD_HOOK_SLOTS 256   // uf_union_kernel's hook counters (one 64-byte line each)  // the pair kernel spreads its block totals over this many PairStats
hipError_t launch_bucket_pairs(const uint32_t *sorted_hash, const uint32_t *sorted_uid, uint64_t U,
                               const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t d,
                               uint32_t seg, uint32_t nseg, uint32_t shard, uint32_t n_shards,
                               uint32_t *edges, unsigned long long *edge_count, uint64_t edge_cap,
                               PairStats *stats, hipStream_t st);
hipError_t launch_select_shard(const uint32_t *hashes, uint64_t U, uint32_t shard, uint32_t n_shards,
                               uint32_t *out_hash, uint32_t *out_uid, unsigned long long *counter, hipStream_t st);
hipError_t launch_pairs_within(const uint8_t *a, const uint64_t *ao, const uint8_t *b, const uint64_t *bo,
                               uint64_t n, int d, int metric, uint8_t *out, hipStream_t st);

// edit.hip -- Levenshtein neighbour search + contains
hipError_t launch_len_present(const uint32_t *ulens, uint64_t U, KeyShape sh, uint8_t *len_present, hipStream_t st);
hipError_t launch_edit_records(const uint32_t *urecs, const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t d,
                               const uint8_t *len_present, uint32_t slots_per_key, uint32_t *out_hash,
                               uint32_t *out_payload, hipStream_t st);
hipError_t launch_edit_candidates(const uint32_t *sorted_hash, const uint32_t *sorted_payload, uint64_t R,
                                  const uint32_t *ulens, KeyShape sh, uint32_t d, uint32_t shard, uint32_t n_shards,
                                  uint64_t *cands, unsigned long long *cand_count, uint64_t cand_cap, hipStream_t st);
hipError_t launch_edit_verify(const uint64_t *cands, uint64_t C, const uint32_t *urecs, const uint32_t *ulens,
                              KeyShape sh, uint32_t d, uint32_t *edges, unsigned long long *edge_count,
                              uint64_t edge_cap, hipStream_t st);
// the bucketed edit search without a sort (edit.hip, "grouped"): items = (substring hash, payload)
hipError_t launch_edit_len_counts(const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t *counts, hipStream_t st);
hipError_t launch_edit_items(const uint32_t *urecs, const uint32_t *ulens, uint64_t U, KeyShape sh, uint32_t d,
                             const uint8_t *probe_mask, const uint32_t *probe_count, uint32_t *per_key,
                             const uint32_t *per_key_incl, uint32_t *hashes, uint32_t *payloads, int pass, hipStream_t st,
                             const uint32_t *index_from = nullptr /* [d + 1][U] segment hashes: the index items' hashes */,
                             uint32_t *probers = nullptr /* pass 1: room for the n_probers keys that file probe items; their
                                                            items come from a second, dense kernel */,
                             unsigned long long *n_probers_dev = nullptr /* zeroed by the caller */, uint64_t n_probers = 0);
hipError_t launch_edit_grouped_verify(const uint64_t *cands, const unsigned long long *cand_count, uint64_t list_cap,
                                      uint32_t n_lists, const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                                      uint32_t d, const uint8_t *probe_mask, uint32_t *edges,
                                      unsigned long long *edge_count, uint64_t edge_cap, unsigned long long *cand_need,
                                      unsigned long long *n_verified, int cross_only, hipStream_t st);
hipError_t launch_contains(const uint8_t *q, const uint64_t *qo, uint64_t nq, const uint32_t *urecs,
                           const uint32_t *ulens, uint64_t U, KeyShape sh, const uint8_t *alphabet_dev, int d,
                           int metric, uint32_t *hit_flags, hipStream_t st, const uint8_t *alive = nullptr);

// group.hip -- one search pass without a sort: two-level partition of (segment hash, uid) + one wave per bucket
uint32_t group_tile_size();
uint32_t group_max_bins();
uint32_t group_cand_lists();   // cand_count of the two launchers below: this many counters, 8 words apart
hipError_t launch_group_hist(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                             const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                             uint32_t n_bins, uint32_t *hist, hipStream_t st);
hipError_t launch_group_scatter(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                                const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                                uint32_t n_bins, uint32_t *cursor, uint32_t *out, hipStream_t st, uint32_t slab_cap = 0,
                                uint32_t *slab_overflow = nullptr, const uint32_t *values = nullptr, uint32_t l1_subs = 0,
                                const uint32_t *seg_end = nullptr, uint32_t seg_mask = 0xFFFFFFFFu);
hipError_t launch_group_slab_tile_starts(const uint32_t *seg_start, const uint32_t *seg_end, uint32_t n_seg,
                                         uint32_t *tile_start, hipStream_t st);
hipError_t launch_group_tile_starts(const uint32_t *seg_start, uint32_t n_seg, uint32_t *tile_start, hipStream_t st);
hipError_t launch_group_matrix_starts(const uint32_t *matrix_incl, uint32_t n_bins, uint32_t n_tiles, uint32_t *start,
                                      hipStream_t st);
hipError_t launch_group_bucket_starts(const uint32_t *hist_incl, uint32_t n_buckets, uint32_t *bucket_start,
                                      uint32_t *cursor, hipStream_t st);
// what else the first launch of a search inside fqd_find_edges sets up (all optional)
struct PassInitMore {
    unsigned long long *ctr64 = nullptr;     // ctr64[zero_a] = ctr64[zero_b] = ctr64[zero_c] = 0
    uint32_t zero_a = 0, zero_b = 0, zero_c = 0;
    uint32_t *stats = nullptr;               // stat_words words of zeros
    uint32_t stat_words = 0;
    uint32_t *start1 = nullptr, *cursor1 = nullptr, n1 = 0, cap1 = 0;    // slab starts of partition level 1 ...
    uint32_t *start2 = nullptr, *cursor2 = nullptr, n2 = 0, cap2 = 0;    // ... and level 2
};
hipError_t launch_group_pass_init(uint32_t *seg1, uint32_t *tiles1, uint32_t n_items, uint32_t n_tiles,
                                  unsigned long long *cand_ctr, uint32_t ctr_words, hipStream_t st,
                                  PassInitMore more = PassInitMore());
hipError_t launch_group_slab_starts(uint32_t n_buckets, uint32_t cap, uint32_t *bucket_start, uint32_t *cursor,
                                    hipStream_t st);
hipError_t launch_grouped_candidates(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                     uint32_t n_buckets, uint32_t bucket_bits, uint64_t *cands,
                                     unsigned long long *cand_count, uint64_t cand_cap, hipStream_t st,
                                     uint32_t require_any = 0 /* list only pairs with one of these bits in a value */,
                                     const uint8_t *skip = nullptr /* skip[b] != 0: bucket b is left out */,
                                     uint32_t give_up_over = 0, unsigned long long *cand_need = nullptr /* a bucket of more
                                     items is not walked; *cand_need is raised beyond every budget instead */);
bool group_tiles_possible(KeyShape sh, uint32_t nseg);
// crowded buckets of a search at distance d <= 3 (group.hip "crowded buckets"): marked, skipped by the candidate kernel,
// their keys matched on finer pieces. group_fine_items(d): items a crowded key files (0: no refinement at that distance);
// uids must fit group_fine_uid_bits() bits
uint32_t group_fine_items(uint32_t d);
uint32_t group_fine_uid_bits();
hipError_t launch_group_mark_crowded(const uint32_t *bucket_start, const uint32_t *bucket_end, uint32_t n_buckets,
                                     uint32_t limit, uint32_t tile_max, uint8_t *crowded, uint32_t *list, uint32_t *list2,
                                     unsigned long long *counts /* [8] */, hipStream_t st);
hipError_t launch_group_refine_items(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                     const uint32_t *list, const unsigned long long *counts, uint32_t fused_U,
                                     const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t *seen,
                                     uint32_t *out_hash, uint32_t *out_val, unsigned long long *n_keys, uint64_t key_cap,
                                     hipStream_t st, uint32_t d);
hipError_t launch_group_crowded_tiles(const uint32_t *items, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                      const uint32_t *list, const unsigned long long *counts, unsigned long long *tile_prefix,
                                      uint32_t fused_U, uint32_t seg0, const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                                      uint32_t d, uint32_t nseg, uint32_t *edges, unsigned long long *edge_count,
                                      uint64_t edge_cap, hipStream_t st);
hipError_t launch_group_verify_refined(const uint64_t *cands, const unsigned long long *cand_count, uint64_t cand_cap,
                                       const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t nseg,
                                       const uint32_t *seg_hashes, uint64_t U, uint32_t bucket_bits,
                                       const uint8_t *crowded, uint32_t *edges, unsigned long long *edge_count,
                                       uint64_t edge_cap, unsigned long long *cand_need, hipStream_t st, uint32_t d, uint32_t accept = 3);
hipError_t launch_verify_candidates(const uint64_t *cands, const unsigned long long *cand_count, uint64_t cand_cap,
                                    const uint32_t *urecs, const uint32_t *ulens, KeyShape sh, uint32_t d, uint32_t seg,
                                    uint32_t nseg, uint32_t *edges, unsigned long long *edge_count, uint64_t edge_cap,
                                    unsigned long long *cand_need, PairStats *stats, hipStream_t st,
                                    uint32_t fused_U = 0);

// exchange.hip -- group packed reads by owner rank
hipError_t launch_owner(const uint32_t *hashes, uint64_t n, uint32_t parts, uint32_t *owner, hipStream_t st);
hipError_t launch_gather_by_owner(const uint32_t *order, uint64_t n, KeyShape sh, const uint32_t *recs,
                                  const uint32_t *lens, const uint32_t *weights, uint64_t id0, uint32_t *recs_out,
                                  uint32_t *lens_out, uint64_t *ids_out, uint32_t *ids32_out, uint32_t *weights_out,
                                  hipStream_t st, uint32_t stamp_word = 0);
// ids64[pos] = src.at(pos) for every read, and the stamped word is cleared (the sort-based collapse
// compares whole records, padding included)
hipError_t launch_extract_ids(IdSource src, uint32_t *recs, uint64_t n, uint64_t *ids64, hipStream_t st);
uint32_t split_tiles(uint64_t n);
uint32_t split_max_parts();
hipError_t launch_split_count(const uint32_t *owner, uint64_t n, uint32_t parts, uint32_t *matrix, hipStream_t st);
hipError_t launch_split_order(const uint32_t *owner, uint64_t n, uint32_t parts, const uint32_t *matrix,
                              const uint32_t *matrix_incl, uint32_t *order, uint64_t *counts, hipStream_t st);
hipError_t launch_owner_counts(const uint32_t *owner_sorted, uint64_t n, uint32_t parts, uint64_t *counts,
                               hipStream_t st);

// quality.hip -- per-read mean error rate gate
hipError_t launch_quality(const uint8_t *bytes, uint64_t n_bytes, const uint64_t *offsets, uint64_t n,
                          uint32_t fixed_len, uint32_t max_len, const double *table_dev, uint32_t phred_offset,
                          uint32_t max_score, double threshold, uint32_t *pass, double *means, uint32_t *bad_flag,
                          hipStream_t st);

// graph.hip -- union-find + dissection
hipError_t launch_uf_init(uint32_t *parent, uint64_t U, hipStream_t st);
// components and pass 1 of the closed-form directional dissection in one sweep over the edges, on node records
// node[x] = (parent, state byte) -- graph.hip union_directional_kernel; then the records taken apart again
hipError_t launch_graph_preinit_nodes(uint32_t *node, uint32_t *best, uint8_t *root_taint, const uint32_t *ucounts, uint64_t U,
                                      unsigned long long *hook_slots, uint32_t hook_words, hipStream_t st, uint32_t *zero32,
                                      uint32_t zero32_words, unsigned long long *zero64_a, unsigned long long *zero64_b,
                                      unsigned long long *zero64_c);
hipError_t launch_union_directional(uint32_t *node, const uint32_t *edges, uint64_t E, const uint32_t *ucounts, uint32_t *list11,
                                    unsigned long long *list11_count, unsigned long long *n_hooks, hipStream_t st,
                                    bool sampled_first);
hipError_t launch_unzip_nodes(const uint32_t *node, uint32_t *parent, uint8_t *state, uint64_t U, hipStream_t st);
hipError_t launch_mask_dead_edges(uint32_t *edges, uint64_t E, const uint8_t *alive, hipStream_t st);
hipError_t launch_uf_union(uint32_t *parent, const uint32_t *edges, uint64_t E, unsigned long long *n_hooks,
                           hipStream_t st, bool sampled_first = false);
uint32_t kept_bin_shift(uint64_t window);
uint32_t kept_bin_lists();
hipError_t launch_kept_bins(int method, const uint32_t *labels, const uint32_t *best, const uint8_t *state,
                            const uint64_t *ufirst, uint64_t id_lo, uint64_t window, uint64_t U, uint8_t *kept,
                            const uint32_t *ucounts, const uint32_t *parent1, const uint8_t *root_taint,
                            uint32_t *cursor, uint32_t *lists, unsigned long long *n_kept_total, uint64_t id_base,
                            uint64_t *out, uint32_t *n_listed, hipStream_t st, bool cursors_zeroed = false);
uint32_t window_blocks(uint64_t n);
hipError_t launch_window_count(const uint8_t *flags, uint64_t n, uint32_t *block_counts, hipStream_t st);
hipError_t launch_window_emit(const uint8_t *flags, uint64_t n, const uint32_t *block_incl, uint64_t id_base,
                              uint64_t *out, hipStream_t st);
hipError_t launch_hook_total(const unsigned long long *slots, uint64_t n_nodes, unsigned long long *n_components,
                             hipStream_t st, unsigned long long *n_second_walks = nullptr);
hipError_t launch_edge_roots(uint32_t *parent, const uint32_t *edges, uint64_t E, uint32_t *roots, hipStream_t st);
hipError_t launch_subgraph_mark(const uint32_t *uv, const uint32_t *roots, uint64_t E, uint32_t n_parts, uint32_t part,
                                uint32_t *flags, uint32_t *sub, unsigned long long *n_sub, hipStream_t st);
// home clusters (graph.hip): rank r holds the unique keys [lo[r], lo[r + 1]) of the job-wide numbering
#define FQD_MAX_HOME_RANKS 16
struct UidBounds {
    uint32_t n;                                 // ranks
    uint32_t lo[FQD_MAX_HOME_RANKS + 1];
};
hipError_t launch_subgraph_mark_home(const uint32_t *uv, const uint32_t *roots, uint64_t E, uint32_t n_parts, uint32_t part,
                                     UidBounds bounds, uint8_t *span, uint32_t *flags, uint32_t *sub,
                                     unsigned long long *n_sub, uint32_t *home, unsigned long long *n_home,
                                     unsigned long long *n_span, hipStream_t st);
hipError_t launch_mark_dropped_after(uint8_t *state, uint32_t *best, uint64_t U, const uint32_t *dropped, uint64_t n,
                                     uint32_t *bad, hipStream_t st);
// owner slabs without their slack (collapse_lds.hip): fills + exclusive starts of n slabs (cap == 0: `in` holds fills),
// and the filled prefixes of the slabs copied to dense[start[p] ..)
hipError_t launch_fill_scan(const uint32_t *in, uint32_t n, uint32_t cap, uint32_t *fills, uint32_t *start, uint32_t *end,
                            hipStream_t st);
hipError_t launch_slab_dense_rows(const uint32_t *slabs, const uint32_t *start, uint32_t n_slabs, uint32_t cap,
                                  uint32_t *dense, hipStream_t st, uint64_t dense_rows = ~0ull /* room in dense: rows behind it are
                                  * dropped */, uint32_t *over = nullptr /* set to 1 when a row was dropped */);
hipError_t launch_subgraph_finish(const uint32_t *flags, const uint32_t *flags_incl, uint64_t n_nodes, uint64_t E,
                                  uint32_t *touched, uint32_t *sub, const unsigned long long *n_sub, hipStream_t st);
hipError_t launch_check_indices(const uint32_t *idx, uint64_t n, uint64_t limit, uint32_t *bad, hipStream_t st);
hipError_t launch_mark_dropped(uint8_t *state, uint64_t U, const uint32_t *dropped, uint64_t n, uint32_t *bad,
                               hipStream_t st);
hipError_t launch_uf_flatten(uint32_t *parent, uint64_t U, unsigned long long *n_roots, hipStream_t st);
hipError_t launch_dissect_init(uint32_t *best, uint8_t *state, uint64_t U, hipStream_t st);
hipError_t launch_graph_preinit(uint32_t *parent, uint32_t *best, uint8_t *state, uint8_t *root_taint /* != NULL: the closed-form
                                directional dissection follows */,
                                uint64_t U, unsigned long long *hook_slots, uint32_t hook_words, hipStream_t st,
                                uint32_t *zero32 = nullptr, uint32_t zero32_words = 0,
                                unsigned long long *zero64_a = nullptr, unsigned long long *zero64_b = nullptr,
                                const uint32_t *ucounts = nullptr /* with root_taint: state[i] starts as the count nibble of
                                                                   * the closed-form directional dissection */,
                                unsigned long long *zero64_c = nullptr);
hipError_t launch_dstate_init(uint8_t *state, const uint32_t *ucounts, uint64_t U, hipStream_t st);
hipError_t launch_highest_count(const uint32_t *labels, const uint32_t *ucounts, const uint32_t *urecs,
                                const uint32_t *ulens, KeyShape sh, uint64_t U, uint32_t *best,
                                hipStream_t st);
hipError_t launch_directional_round(const uint32_t *edges, uint64_t E, const uint32_t *ucounts,
                                    const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                                    uint32_t *best, uint32_t *stamp, uint32_t round, uint32_t *changed,
                                    hipStream_t st);
hipError_t launch_orient_edges(uint32_t *edges, uint64_t E, const uint32_t *ucounts, const uint32_t *urecs,
                               const uint32_t *ulens, KeyShape sh, hipStream_t st);
hipError_t launch_adjacency_live_edges(const uint32_t *edges, uint64_t E, const uint8_t *state, uint32_t *live,
                                       unsigned long long *live_count, hipStream_t st);
hipError_t launch_adjacency_round(const uint32_t *edges, uint64_t E, uint64_t U, uint8_t *state, uint32_t *blocked,
                                  uint32_t round, uint32_t *changed, hipStream_t st);
hipError_t launch_kept_flags(int method, const uint32_t *labels, const uint32_t *best, const uint8_t *state,
                             const uint64_t *ufirst, uint64_t id_lo, uint64_t id_hi, uint64_t U, uint8_t *kept,
                             uint32_t *kept_u32, uint8_t *window_flags, uint64_t window_size,
                             const uint32_t *ucounts, const uint32_t *parent1, const uint8_t *root_taint,
                             unsigned long long *n_kept_total, hipStream_t st);
// directional dissection without rounds: pass 1 (edges: in-arcs, taint, unions of count-1 keys), pass 2
// (count-1 keys report to their set's root); method 3 of launch_kept_flags reads the verdicts
hipError_t launch_directional_closed(const uint32_t *edges, uint64_t E, const uint32_t *ucounts, const uint32_t *urecs,
                                     const uint32_t *ulens, KeyShape sh, const uint32_t *parent /* the components' (complete
                                     before pass 2) */, uint8_t *state,
                                     uint32_t *list11, unsigned long long *list11_count, uint8_t *root_taint,
                                     uint32_t *best, int pass, hipStream_t st, uint32_t *roots, uint32_t *list2,
                                     unsigned long long *list2_count);
hipError_t launch_gather_kept(const uint32_t *kept_u32, const uint32_t *kept_scan, const uint64_t *ufirst,
                              uint64_t U, uint64_t *out, hipStream_t st);

// trieorder.hip -- the reference trie's key order, node census and clusters in pop order
hipError_t launch_trie_iota(uint32_t *out, uint64_t n, hipStream_t st);
hipError_t launch_trie_chunk_keys(const uint32_t *order, uint64_t U, const uint32_t *urecs, const uint32_t *ulens,
                                  KeyShape sh, const uint8_t *idx_of_code, uint32_t end_digit, uint32_t bits, uint32_t p0,
                                  uint32_t P, unsigned long long *keys, hipStream_t st);
hipError_t launch_trie_rank_of(const uint32_t *order, uint64_t U, uint32_t *rank, hipStream_t st);
hipError_t launch_trie_lcp(const uint32_t *order, uint64_t U, const uint32_t *urecs, const uint32_t *ulens, KeyShape sh,
                           uint32_t *lcp, hipStream_t st);
hipError_t launch_trie_alive_mark(const uint32_t *order, uint64_t U, const uint8_t *alive, uint32_t *mark,
                                  hipStream_t st);
hipError_t launch_trie_census(const uint32_t *order, const uint32_t *lcp, uint64_t U, const uint32_t *urecs,
                              const uint32_t *ulens, KeyShape sh, const uint8_t *idx_of_code, const uint8_t *alive,
                              const uint32_t *last_alive, uint32_t n_layers, uint32_t n_cols, unsigned long long *stats,
                              unsigned long long *memory_size, hipStream_t st);
hipError_t launch_trie_seed_ranks(const uint32_t *labels, const uint32_t *rank, uint64_t U, const uint8_t *alive,
                                  uint32_t *seed, hipStream_t st);
hipError_t launch_trie_member_keys(const uint32_t *labels, const uint32_t *rank, uint64_t U, const uint8_t *alive,
                                   const uint32_t *seed, unsigned long long *keys, hipStream_t st);
hipError_t launch_trie_member_heads(const unsigned long long *sorted, uint64_t U, const uint32_t *order, uint32_t *heads,
                                    uint32_t *members, hipStream_t st);
hipError_t launch_trie_cluster_offsets(const unsigned long long *sorted, const uint32_t *heads, const uint32_t *heads_incl,
                                       uint64_t U, unsigned long long *offsets, hipStream_t st);
// one round of the lazy-alphabet search (see trieorder.hip)
hipError_t launch_symbol_round(const uint32_t *order, const uint32_t *lcp, uint64_t U, const uint32_t *urecs,
                               const uint32_t *ulens, KeyShape sh, const uint64_t *ufirst, const uint8_t *alive,
                               const uint8_t *codes, uint32_t n_symbols, const unsigned long long *after,
                               unsigned long long *cand, uint32_t *depth, unsigned long long *partner, hipStream_t st);
// the unique table as a store: rows removed, merge weights (count, 0 for removed rows), ids base + i
hipError_t launch_store_remove(const uint32_t *uids, uint64_t n, uint64_t U, uint8_t *alive, uint32_t *bad, hipStream_t st);
hipError_t launch_store_weights(const uint32_t *counts, const uint8_t *alive, uint64_t U, uint32_t *out, hipStream_t st);
hipError_t launch_store_fill_ids(uint64_t *out, uint64_t n, uint64_t base, hipStream_t st);
// records of one geometry re-encoded into another (a store whose alphabet or key length grew)
hipError_t launch_transcode_records(const uint32_t *src, const uint32_t *src_lens, uint64_t n, KeyShape from, KeyShape to,
                                    const uint8_t *code_map, uint32_t *dst, uint32_t *dst_lens, hipStream_t st);

// synth.hip
hipError_t launch_synth(uint8_t *out, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length,
                        uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub,
                        hipStream_t st, uint64_t thr_hot = 0, uint64_t thr_ladder = 0, uint32_t lowc_every = 0,
                        uint32_t skew = 0);
hipError_t launch_copy16(const void *src, void *dst, uint64_t bytes, hipStream_t st);
hipError_t launch_synth_indels(uint64_t n_total, uint64_t start, uint64_t count, uint32_t length, uint32_t umi,
                               uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub, uint64_t thr_indel,
                               unsigned long long *lens, const unsigned long long *offsets, uint8_t *out, hipStream_t st);

}  // namespace fqd
