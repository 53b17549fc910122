// api_search.hip -- C ABI, part 2: the neighbour search (stage 3): Hamming passes by partition
// (group.hip) or radix sort (edges.hip), the bucketed edit search (edit.hip).
#include "api_ctx.h"

namespace {

int find_edges_edit(fqd_ctx *c, uint32_t d, uint32_t shard, uint32_t n_shards)
{
    const uint64_t U = c->U;
    const KeyShape sh = c->ks;
    if (d > 64)
        return fail(c, FQD_E_VALUE, "edit distance bound above 64 is not supported on device");
    HIP_TRY(c, c->len_present.reserve((size_t)sh.max_len + 16));
    HIP_TRY(c, hipMemsetAsync(c->len_present.p, 0, (size_t)sh.max_len + 1, c->st));
    HIP_TRY(c, fqd::launch_len_present(c->ulens.as<uint32_t>(), U, sh, c->len_present.as<uint8_t>(), c->st));
    std::vector<uint8_t> present((size_t)sh.max_len + 1);
    HIP_TRY(c, hipMemcpyAsync(present.data(), c->len_present.p, present.size(), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    uint32_t n_lengths = 0;
    for (uint8_t f : present)
        n_lengths += f ? 1 : 0;
    const uint32_t classes = std::min<uint32_t>(2 * d + 1, std::max<uint32_t>(n_lengths, 1));
    const uint32_t slots = (d + 1) * (1 + classes * (2 * d + 1));
    const uint64_t R = U * slots;
    if (R >= 0xFFFFFF00ull)
        return fail(c, FQD_E_VALUE, "edit search: more than 2^32 index/probe records; lower max_distance or shard the job");
    HIP_TRY(c, c->ed_hash.reserve(R * 4 + 16));
    HIP_TRY(c, c->ed_payload.reserve(R * 4 + 16));
    HIP_TRY(c, c->ed_hash_sorted.reserve(R * 4 + 16));
    HIP_TRY(c, c->ed_payload_sorted.reserve(R * 4 + 16));
    HIP_TRY(c, fqd::launch_edit_records(c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), U, sh, d,
                                        c->len_present.as<uint8_t>(), slots, c->ed_hash.as<uint32_t>(),
                                        c->ed_payload.as<uint32_t>(), c->st));
    FQD_TRY(sort_u32_pairs(c, c->ed_hash.as<uint32_t>(), c->ed_hash_sorted.as<uint32_t>(),
                           c->ed_payload.as<uint32_t>(), c->ed_payload_sorted.as<uint32_t>(), R));
    uint64_t cap = std::max<uint64_t>(c->ed_cands.cap / 8, std::max<uint64_t>(4096, 4 * U));
    unsigned long long n_cand = 0;
    for (;;) {
        HIP_TRY(c, c->ed_cands.reserve(cap * 8));
        FQD_TRY(zero_ctr64(c, C64_SUM));
        HIP_TRY(c, fqd::launch_edit_candidates(c->ed_hash_sorted.as<uint32_t>(), c->ed_payload_sorted.as<uint32_t>(), R,
                                               c->ulens.as<uint32_t>(), sh, d, shard, n_shards,
                                               c->ed_cands.as<uint64_t>(), c->d_ctr64.as<unsigned long long>() + C64_SUM,
                                               cap, c->st));
        FQD_TRY(read_ctr64(c, C64_SUM, &n_cand));
        if (n_cand <= cap)
            break;
        cap = n_cand + n_cand / 8 + 1024;
    }
    c->last_stats.pairs_compared = n_cand;
    if (!n_cand)
        return FQD_OK;
    HIP_TRY(c, c->ed_cands_sorted.reserve(n_cand * 8 + 16));
    {
        const size_t need = fqd::sort_keys_u64_temp(n_cand);
        HIP_TRY(c, c->tmp.reserve(need + 16));
        HIP_TRY(c, fqd::sort_keys_u64(c->tmp.p, need, c->ed_cands.as<uint64_t>(), c->ed_cands_sorted.as<uint64_t>(),
                                      n_cand, 64, c->st));
    }
    HIP_TRY(c, c->edges.reserve(n_cand * 8 + 16));  // every unique candidate yields at most one edge
    c->edge_cap = c->edges.cap / 8;
    HIP_TRY(c, fqd::launch_edit_verify(c->ed_cands_sorted.as<uint64_t>(), n_cand, c->urecs.as<uint32_t>(),
                                       c->ulens.as<uint32_t>(), sh, d, c->edges.as<uint32_t>(),
                                       c->d_ctr64.as<unsigned long long>() + C64_EDGES, c->edge_cap, c->st));
    unsigned long long ne = 0;
    FQD_TRY(read_ctr64(c, C64_EDGES, &ne));
    c->E = ne;
    c->last_stats.edges = ne;
    return FQD_OK;
}

}  // namespace

extern "C" int fqd_api_partition_pairs(fqd_ctx *c, const uint32_t *keys, uint64_t N, uint32_t B, bool slabs,
                                       const uint32_t **items_out, const uint32_t **bucket_end_out,
                                       const uint32_t *values);

namespace {

// The same search without a device-wide sort (edit.hip, "grouped"): few items (index items, probe
// items only from the smaller of two length classes), partitioned by hash bits and matched in LDS
// by the kernels of the Hamming passes, every pair verified once under its first matching
// configuration. *done = false: not applicable (d > 3, 2^26 keys or more) -- the sorted way runs.
// cross_only (d = 1): pairs of keys of ONE length are Hamming neighbours or no neighbours -- the
// Hamming passes have reported them into c->edges already (c->E of them); this search appends the
// pairs of different lengths behind them.
// index_ready (with cross_only): c->seg_hashes holds the [d + 1][U] segment hashes the Hamming passes grouped by --
// they are the index items' hashes, no key is hashed twice and only the probing keys' records are read.
int find_edges_edit_grouped(fqd_ctx *c, uint32_t d, bool *done, bool cross_only = false, bool index_ready = false)
{
    *done = false;
    const uint64_t base_edges = cross_only ? c->E : 0;
    const uint64_t U = c->U;
    const KeyShape sh = c->ks;
    const char *pin = getenv("FQD_EDIT");                  // "sort" / "grouped": tests pin a path
    if (d > 3 || d == 0 || U >= (1ull << 26) || (pin && !strcmp(pin, "sort")))
        return FQD_OK;
    if (U < 32768 && !(pin && !strcmp(pin, "grouped")))
        return FQD_OK;
    const uint32_t nseg = d + 1, L = sh.max_len;
    // ---- length classes: who probes whom --------------------------------------------------
    HIP_TRY(c, c->eg_tables.reserve(((size_t)L + 1) * 12 + 64));
    uint32_t *d_counts = c->eg_tables.as<uint32_t>();          // [L + 1] keys per length, later probe items per key of that length
    uint8_t *d_mask = reinterpret_cast<uint8_t *>(d_counts + (L + 1));
    std::vector<uint32_t> counts((size_t)L + 1, 0), probe_count((size_t)L + 1, 0);
    std::vector<uint8_t> mask((size_t)L + 1, 0);
    if (sh.ragged) {
        HIP_TRY(c, hipMemsetAsync(d_counts, 0, ((size_t)L + 1) * 4, c->st));
        HIP_TRY(c, fqd::launch_edit_len_counts(c->ulens.as<uint32_t>(), U, sh, d_counts, c->st));
        HIP_TRY(c, hipMemcpyAsync(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(c, stream_wait(c->st));
    } else {
        counts[L] = (uint32_t)U;               // one length class
    }
    uint64_t n_probe = 0, n_probers = 0;       // probe items; keys that file any
    for (uint32_t la = 0; la <= L; la++) {
        if (!counts[la])
            continue;
        for (int j = 0; j <= 2 * (int)d; j++) {
            const long lb = (long)la - (long)d + j;
            if (lb < 0 || lb > (long)L || !counts[(size_t)lb])
                continue;
            // the smaller class probes the larger (ties: the shorter keys probe); the own class with
            // shifts != 0 only, which d = 1 never needs
            const bool probes = (uint32_t)lb == la ? d >= 2
                                                   : (counts[la] < counts[(size_t)lb] ||
                                                      (counts[la] == counts[(size_t)lb] && la < (uint32_t)lb));
            if (!probes)
                continue;
            mask[la] |= (uint8_t)(1u << j);
            for (uint32_t s = 0; s < nseg; s++) {
                const uint32_t lo = (uint32_t)lb * s / nseg, hi = (uint32_t)lb * (s + 1) / nseg;
                for (int delta = -(int)d; delta <= (int)d; delta++) {
                    if ((uint32_t)lb == la && delta == 0)
                        continue;
                    const long start = (long)lo + delta;
                    if (start >= 0 && (uint64_t)start + (hi - lo) <= la)
                        probe_count[la]++;
                }
            }
        }
        n_probe += (uint64_t)probe_count[la] * counts[la];
        n_probers += probe_count[la] ? counts[la] : 0u;
    }
    const uint64_t R = U * nseg + n_probe;
    if (R >= 0xFFFFFF00ull)
        return FQD_OK;
    HIP_TRY(c, hipMemcpyAsync(d_counts, probe_count.data(), probe_count.size() * 4, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipMemcpyAsync(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, stream_wait(c->st));          // (host vectors)
    // ---- items -------------------------------------------------------------------------
    HIP_TRY(c, c->ed_hash.reserve(R * 4 + 16));
    HIP_TRY(c, c->ed_payload.reserve(R * 4 + 16));
    HIP_TRY(c, c->eg_per_key.reserve(U * 4 + 16));
    HIP_TRY(c, c->eg_per_key_incl.reserve(U * 4 + 16));
    if (n_probe) {
        HIP_TRY(c, fqd::launch_edit_items(c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), U, sh, d, d_mask, d_counts,
                                          c->eg_per_key.as<uint32_t>(), nullptr, nullptr, nullptr, 0, c->st));
        FQD_TRY(scan_u32(c, c->eg_per_key.as<uint32_t>(), c->eg_per_key_incl.as<uint32_t>(), U));
    } else {
        HIP_TRY(c, hipMemsetAsync(c->eg_per_key_incl.p, 0, U * 4, c->st));
    }
    // (the per-key counts have been scanned: their array now takes the list of the probing keys)
    FQD_TRY(zero_ctr64(c, C64_SUM));
    HIP_TRY(c, fqd::launch_edit_items(c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), U, sh, d, d_mask, d_counts,
                                      c->eg_per_key.as<uint32_t>(), c->eg_per_key_incl.as<uint32_t>(),
                                      c->ed_hash.as<uint32_t>(), c->ed_payload.as<uint32_t>(), 1, c->st,
                                      cross_only && index_ready && !getenv("FQD_EDIT_OWN_INDEX_HASHES")
                                          ? c->seg_hashes.as<uint32_t>() : nullptr,
                                      c->eg_per_key.as<uint32_t>(), c->d_ctr64.as<unsigned long long>() + C64_SUM, n_probers));
    // ---- partition, candidates, verification; buffers grow and the passes run again if needed -----
    uint32_t B = 8;
    while (B < 20 && (R >> B) > 320)
        B++;
    if (const char *e = getenv("FQD_GROUP_BUCKET_BITS"))
        B = (uint32_t)std::max(1, std::min(20, atoi(e)));
    const uint32_t n_buckets = 1u << B;
    if (c->gp_cand_cap < 1024 || !c->gp_cands.p) {
        c->gp_cand_cap = std::max<uint64_t>(1u << 20, 2 * R);
        HIP_TRY(c, c->gp_cands.reserve(c->gp_cand_cap * 8));
    }
    c->gp_cand_cap = c->gp_cands.cap / 8;
    if (!cross_only && (c->edge_cap < 1024 || !c->edges.p)) {
        c->edge_cap = std::max<uint64_t>(1024, U);
        HIP_TRY(c, c->edges.reserve(c->edge_cap * 8));
    }
    c->edge_cap = c->edges.cap / 8;
    unsigned long long *ctr = c->d_ctr64.as<unsigned long long>();
    for (int attempt = 0;; attempt++) {
        c->h_extra64[7] = base_edges;                  // the edge counter starts behind the edges already there
        HIP_TRY(c, hipMemcpyAsync(ctr + C64_EDGES, &c->h_extra64[7], 8, hipMemcpyHostToDevice, c->st));
        FQD_TRY(zero_ctr64(c, C64_CAND_NEED, 2));      // ... and C64_SLAB
        FQD_TRY(zero_ctr64(c, C64_SUM));
        const uint32_t *items = nullptr, *bucket_end = nullptr;
        FQD_TRY(fqd_api_partition_pairs(c, c->ed_hash.as<uint32_t>(), R, B,
                                        !c->gp_slab_off && !getenv("FQD_GROUP_NO_SLABS"), &items, &bucket_end,
                                        c->ed_payload.as<uint32_t>()));
        unsigned long long *cand_ctr = reinterpret_cast<unsigned long long *>(c->gp_small.as<uint32_t>() + 4096);
        // (pairs of different lengths only: one of the two items is then a probe item -- bit 31 of its payload)
        KTIME(c, FQD_K_PAIRS, fqd::launch_grouped_candidates(items, c->ld_start.as<uint32_t>(), bucket_end, n_buckets, B,
                                                             c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap, c->st,
                                                             cross_only ? 0x80000000u : 0u));
        KTIME(c, FQD_K_VERIFY, fqd::launch_edit_grouped_verify(
                  c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap / fqd::group_cand_lists(), fqd::group_cand_lists(),
                  c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), shape_with_row_lengths(c), d, d_mask, c->edges.as<uint32_t>(),
                  ctr + C64_EDGES, c->edge_cap, ctr + C64_CAND_NEED, ctr + C64_SUM, cross_only ? 1 : 0, c->st));
        unsigned long long ctrs[C64_SLAB + 1] = {0};
        FQD_TRY(read_ctr64(c, 0, ctrs, C64_SLAB + 1));
        const unsigned long long now = ctrs[C64_EDGES], cand_need = ctrs[C64_CAND_NEED];
        const bool slab_over = ctrs[C64_SLAB] != 0;
        if (slab_over)
            c->gp_slab_off = true;
        if (!slab_over && now <= c->edge_cap && cand_need <= c->gp_cand_cap) {
            c->E = now;
            c->last_stats.pairs_compared += ctrs[C64_SUM];
            c->last_stats.edges = now;
            break;
        }
        if (attempt > 3)
            return fail(c, FQD_E_RUNTIME, "edit search: buffers kept overflowing");
        if (now > c->edge_cap) {
            // a larger edge buffer; the edges that were there before this search move over
            DevBuf bigger;
            HIP_TRY(c, bigger.reserve((size_t)(now + now / 8 + 1024) * 8));
            if (base_edges)
                HIP_TRY(c, hipMemcpyAsync(bigger.p, c->edges.p, (size_t)base_edges * 8, hipMemcpyDeviceToDevice, c->st));
            HIP_TRY(c, stream_wait(c->st));
            c->edges.release();
            c->edges = bigger;
            c->edge_cap = c->edges.cap / 8;
        }
        if (cand_need > c->gp_cand_cap) {
            c->gp_cands.release();
            HIP_TRY(c, c->gp_cands.reserve((size_t)(cand_need + cand_need / 8 + 1024) * 8));
            c->gp_cand_cap = c->gp_cands.cap / 8;
        }
    }
    *done = true;
    return FQD_OK;
}

}  // namespace

extern "C" {

// (key, index) pairs of `keys[0..N)` partitioned into 2^B buckets by the top B key bits (group.hip,
// partition.cuh): level 1 by a count matrix, level 2 on fixed slabs unless `slabs` is false (an
// overfull slab raises C64_SLAB; the caller then asks again without slabs). Queues work only.
// *items: the partitioned (key, index) pairs; bucket b is [start[b], min(start[b + 1], end[b])) with
// start = c->ld_start and end = *bucket_end (NULL: start[b + 1]). Also zeroes the candidate counters
// behind the small tables (the search pass uses them).
int fqd_api_partition_pairs(fqd_ctx *c, const uint32_t *keys, uint64_t N, uint32_t B, bool slabs,
                            const uint32_t **items_out, const uint32_t **bucket_end_out, const uint32_t *values)
{
    const uint32_t B1 = B <= 18 ? std::min<uint32_t>(B, 8) : B - 10, B2 = B - B1;
    const uint32_t bins1 = 1u << B1, bins2 = 1u << B2, n_buckets = 1u << B;
    const uint32_t tile = fqd::group_tile_size();
    const uint32_t tiles1 = (uint32_t)((N + tile - 1) / tile), max_tiles2 = tiles1 + bins1;
    HIP_TRY(c, c->gp_a.reserve(N * 8 + 16));
    HIP_TRY(c, c->gp_small.reserve(4096 * 4 + (size_t)fqd::group_cand_lists() * 64));
    HIP_TRY(c, c->ld_start.reserve(((size_t)n_buckets + 1) * 4 + 16));
    uint32_t *small = c->gp_small.as<uint32_t>();
    uint32_t *seg1 = small, *tiles1_d = small + 8, *start1 = small + 16, *tiles2_d = small + 2048;
    // the candidate counters (one per list, a cache line apart) live behind the small tables
    unsigned long long *cand_ctr = reinterpret_cast<unsigned long long *>(small + 4096);
    // (what this first launch also sets up, see fqd::PassInitMore: the slab starts of both levels when level 1 runs
    // in slab mode, and for a search inside fqd_find_edges its counters and statistics)
    fqd::PassInitMore more;
    if (c->search_zero_pending) {
        more.ctr64 = c->d_ctr64.as<unsigned long long>();
        more.zero_a = c->search_keeps_edges ? C64_CAND_NEED : C64_EDGES;     // (pass 0's pairs and statistics stay)
        more.zero_b = C64_CAND_NEED;
        more.zero_c = C64_SLAB;
        if (!c->search_keeps_edges) {
            more.stats = c->d_stats.as<uint32_t>();
            more.stat_words = (uint32_t)(FQD_STAT_SLOTS * sizeof(fqd::PairStats) / 4);
        }
        c->search_zero_pending = false;
    }
    // ---- level 1. Many items (>= 1024 tiles), slabs allowed and a level 2 to follow: slab mode as in the fused
    // pack -- 32 sub-parts per bin (tile t feeds sub-part t % 32), one atomic per (tile, bin) on the part's cursor,
    // no histogram pass over the keys, no count matrix (0.04 ms + a scan at config 3); an overfull part raises
    // C64_SLAB like an overfull level-2 slab. Else: (bin x tile) count matrix, scan, placement without atomics.
    // level 2 in slab mode (as the collapse): no histogram pass; an overfull slab is flagged in C64_SLAB
    uint32_t slab_cap = 0;
    if (B2 && slabs) {
        slab_cap = (uint32_t)(((N >> B) * 3 / 2 + 64 + 3) & ~3ull);
        if ((uint64_t)slab_cap * n_buckets + N >= 0xFFFFFF00ull)
            slab_cap = 0;
    }
    uint32_t l1_subs = 0, cap1 = 0, parts = 0;
    uint32_t l1_min_tiles = 1024;
    if (const char *e = getenv("FQD_GROUP_L1_SLABS_MIN_TILES"))   // tests: small jobs through the slab mode; 0: never
        l1_min_tiles = (uint32_t)strtoul(e, nullptr, 10);
    if (slab_cap && l1_min_tiles && tiles1 >= l1_min_tiles && bins1 <= 256) {
        l1_subs = 32;
        parts = bins1 * l1_subs;
        cap1 = (uint32_t)(((N / parts) * 5 / 4 + 256 + 3) & ~3ull);
        if ((uint64_t)parts * cap1 + N >= 0xFFFFFF00ull)
            l1_subs = 0;
    }
    // level-1 slab tables: seg_start (parts + 1) | cursor = seg_end (parts) | tile_start (parts + 1), 4 words apart
    uint32_t *l1_start = nullptr, *l1_cursor = nullptr, *l1_tiles = nullptr;
    if (l1_subs) {
        HIP_TRY(c, c->gp_a.reserve((size_t)parts * cap1 * 8 + 16));
        HIP_TRY(c, c->ld_seg.reserve((size_t)3 * (parts + 4) * 4));
        l1_start = c->ld_seg.as<uint32_t>();
        l1_cursor = l1_start + (parts + 4);
        l1_tiles = l1_cursor + (parts + 4);
        // the slab starts of both levels with the pass's first launch, ahead of level 1 (l1_subs implies slab mode at level 2)
        HIP_TRY(c, c->ld_cursor.reserve((size_t)n_buckets * 4 + 16));
        more.start1 = l1_start;
        more.cursor1 = l1_cursor;
        more.n1 = parts;
        more.cap1 = cap1;
        more.start2 = c->ld_start.as<uint32_t>();
        more.cursor2 = c->ld_cursor.as<uint32_t>();
        more.n2 = n_buckets;
        more.cap2 = slab_cap;
        HIP_TRY(c, fqd::launch_group_pass_init(seg1, tiles1_d, (uint32_t)N, tiles1, cand_ctr, fqd::group_cand_lists() * 8,
                                               c->st, more));
        KTIME(c, FQD_K_GROUP_SCATTER, fqd::launch_group_scatter(
                  true, keys, nullptr, seg1, tiles1_d, 1, tiles1, 32 - B1, bins1, l1_cursor, c->gp_a.as<uint32_t>(), c->st,
                  cap1, reinterpret_cast<uint32_t *>(c->d_ctr64.as<unsigned long long>() + C64_SLAB), values, l1_subs));
    } else {
        HIP_TRY(c, fqd::launch_group_pass_init(seg1, tiles1_d, (uint32_t)N, tiles1, cand_ctr, fqd::group_cand_lists() * 8,
                                               c->st, more));
        const size_t matrix = (size_t)bins1 * tiles1;
        HIP_TRY(c, c->ld_matrix.reserve(matrix * 4 + 16));
        HIP_TRY(c, c->ld_matrix_incl.reserve(matrix * 4 + 16));
        KTIME(c, FQD_K_GROUP_HIST, fqd::launch_group_hist(true, keys, nullptr, seg1, tiles1_d, 1, tiles1, 32 - B1, bins1,
                                                          c->ld_matrix.as<uint32_t>(), c->st));
        FQD_TRY(scan_u32(c, c->ld_matrix.as<uint32_t>(), c->ld_matrix_incl.as<uint32_t>(), matrix));
        HIP_TRY(c, fqd::launch_group_matrix_starts(c->ld_matrix_incl.as<uint32_t>(), bins1, tiles1, start1, c->st));
        KTIME(c, FQD_K_GROUP_SCATTER, fqd::launch_group_scatter(true, keys, nullptr, seg1, tiles1_d, 1, tiles1, 32 - B1,
                                                                bins1, c->ld_matrix_incl.as<uint32_t>(),
                                                                c->gp_a.as<uint32_t>(), c->st, 0, nullptr, values));
    }
    const uint32_t *items = c->gp_a.as<uint32_t>();
    const uint32_t *bucket_end = nullptr;
    if (B2 == 0) {
        HIP_TRY(c, hipMemcpyAsync(c->ld_start.p, start1, ((size_t)bins1 + 1) * 4, hipMemcpyDeviceToDevice, c->st));
    } else {
        // ---- level 2: every part into 2^B2 buckets by the next key bits
        HIP_TRY(c, c->gp_b.reserve((slab_cap ? (uint64_t)slab_cap * n_buckets : N) * 8 + 16));
        HIP_TRY(c, c->ld_hist.reserve((size_t)n_buckets * 4 + 16));
        HIP_TRY(c, c->ld_hist_incl.reserve((size_t)n_buckets * 4 + 16));
        HIP_TRY(c, c->ld_cursor.reserve((size_t)n_buckets * 4 + 16));
        if (l1_subs)
            HIP_TRY(c, fqd::launch_group_slab_tile_starts(l1_start, l1_cursor, parts, l1_tiles, c->st));
        else
            HIP_TRY(c, fqd::launch_group_tile_starts(start1, bins1, tiles2_d, c->st));
        if (slab_cap) {
            if (!l1_subs)
                HIP_TRY(c, fqd::launch_group_slab_starts(n_buckets, slab_cap, c->ld_start.as<uint32_t>(),
                                                         c->ld_cursor.as<uint32_t>(), c->st));
            bucket_end = c->ld_cursor.as<uint32_t>();
        } else {
            HIP_TRY(c, hipMemsetAsync(c->ld_hist.p, 0, (size_t)n_buckets * 4, c->st));
            KTIME(c, FQD_K_GROUP_HIST, fqd::launch_group_hist(false, nullptr, c->gp_a.as<uint32_t>(), start1, tiles2_d, bins1,
                                                              max_tiles2, 32 - B, bins2, c->ld_hist.as<uint32_t>(), c->st));
            FQD_TRY(scan_u32(c, c->ld_hist.as<uint32_t>(), c->ld_hist_incl.as<uint32_t>(), n_buckets));
            HIP_TRY(c, fqd::launch_group_bucket_starts(c->ld_hist_incl.as<uint32_t>(), n_buckets,
                                                       c->ld_start.as<uint32_t>(), c->ld_cursor.as<uint32_t>(), c->st));
        }
        if (l1_subs)
            KTIME(c, FQD_K_GROUP_SCATTER, fqd::launch_group_scatter(
                      false, nullptr, c->gp_a.as<uint32_t>(), l1_start, l1_tiles, parts, tiles1 + parts, 32 - B, bins2,
                      c->ld_cursor.as<uint32_t>(), c->gp_b.as<uint32_t>(), c->st, slab_cap,
                      reinterpret_cast<uint32_t *>(c->d_ctr64.as<unsigned long long>() + C64_SLAB), nullptr, 0, l1_cursor,
                      bins1 - 1));
        else
        KTIME(c, FQD_K_GROUP_SCATTER, fqd::launch_group_scatter(
                  false, nullptr, c->gp_a.as<uint32_t>(), start1, tiles2_d, bins1, max_tiles2, 32 - B, bins2,
                  c->ld_cursor.as<uint32_t>(), c->gp_b.as<uint32_t>(), c->st, slab_cap,
                  reinterpret_cast<uint32_t *>(c->d_ctr64.as<unsigned long long>() + C64_SLAB)));
        items = c->gp_b.as<uint32_t>();
    }
    *items_out = items;
    *bucket_end_out = bucket_end;
    return FQD_OK;
}

// One Hamming search pass without a device-wide sort (group.hip): the (segment hash, uid) pairs
// are partitioned into 2^B buckets of ~200 keys by the top hash bits, then one wave per bucket
// sub-sorts them in LDS and lists the pairs with equal hashes; a second kernel verifies those.
// Queues work only (no host round trip).
// fused_U != 0: `hashes` holds the hashes of ALL nseg passes, pass s at [s * fused_U, (s + 1) * fused_U):
// one partition, one candidate kernel and one verification for the whole search (the hash of a
// segment mixes its number in, so items of different passes do not meet; a candidate's pass is its
// position / fused_U). Half the launches of two passes, and kernels twice as long.
static bool tiles_possible(const fqd_ctx *c, uint32_t nseg);

// what a grouped_pass leaves about its crowded buckets (one buffer: fqd_ctx::gp_crowded)
struct CrowdedBuf {
    uint8_t *flags;                     // [n_buckets] 0 / 1 (fine pieces) / 2 (small: tiles)
    uint32_t *list, *list2;             // the buckets flagged 1, flagged 2
    unsigned long long *counts;         // [8] see gp_mark_crowded_kernel
    unsigned long long *tile_prefix;    // [n_buckets + 2] scratch of the tiles
    static size_t bytes(uint32_t n_buckets) { return (size_t)n_buckets * 17 + 256; }
};
static CrowdedBuf crowded_buf(fqd_ctx *c, uint32_t n_buckets)
{
    CrowdedBuf b;
    b.flags = c->gp_crowded.as<uint8_t>();
    b.list = reinterpret_cast<uint32_t *>(b.flags + (((size_t)n_buckets + 63) & ~(size_t)63));
    b.list2 = b.list + n_buckets;
    b.counts = reinterpret_cast<unsigned long long *>(b.list2 + n_buckets + (n_buckets & 1u));
    b.tile_prefix = b.counts + 8;
    return b;
}

static int grouped_pass(fqd_ctx *c, const uint32_t *hashes, uint64_t U, uint32_t d, uint32_t s, uint32_t nseg,
                        uint32_t fused_U = 0)
{
    const KeyShape sh = c->ks;
    uint32_t B = 8;
    // ~160-320 keys per bucket: a wave sorts them into 64 sub-bins in LDS. (All passes in one: up to 480 -- at
    // config 3, 28 M items in 2^16 buckets of ~430 take 0.64 ms, in 2^17 buckets 0.67: level 2 of the
    // partition then has 512 bins and half as long runs.)
    while (B < 20 && (U >> B) > (fused_U ? 480u : 320u))
        B++;
    if (const char *e = getenv("FQD_GROUP_BUCKET_BITS"))   // tests: few, crowded buckets
        B = (uint32_t)std::max(1, std::min(20, atoi(e)));
    const uint32_t n_buckets = 1u << B;
    const uint32_t *items = nullptr, *bucket_end = nullptr;
    // (an overfull level-2 slab -- many keys sharing a segment -- is flagged in C64_SLAB and the caller
    // searches again with exact bucket sizes)
    FQD_TRY(fqd_api_partition_pairs(c, hashes, U, B, !c->gp_slab_off && !getenv("FQD_GROUP_NO_SLABS"), &items,
                                    &bucket_end, nullptr));
    unsigned long long *cand_ctr = reinterpret_cast<unsigned long long *>(c->gp_small.as<uint32_t>() + 4096);
    // candidates (pairs with equal segment hashes) -> device list -> verification, one thread per pair
    if (c->gp_cand_cap < 1024 || !c->gp_cands.p) {
        c->gp_cand_cap = std::max<uint64_t>(1u << 20, 2 * U);   // split evenly over the lists: leave slack
        HIP_TRY(c, c->gp_cands.reserve(c->gp_cand_cap * 8));
    }
    c->gp_cand_cap = c->gp_cands.cap / 8;
    unsigned long long *ctr = c->d_ctr64.as<unsigned long long>();
    // Distance 1 to 3, every pass in this one partition, exact bucket sizes (a context whose slabs have overflowed before:
    // data with crowded segment values): buckets of more than 1024 items are marked and SKIPPED here -- all their keys
    // are pairwise candidates -- and grouped_refine() below matches those keys on finer segments.
    const uint8_t *skip = nullptr;
    c->gp_crowded_bits = 0;
    c->gp_last_items = items;
    c->gp_fine_ok = fqd::group_fine_items(d) && fused_U < (1u << fqd::group_fine_uid_bits());
    if ((c->gp_fine_ok || tiles_possible(c, nseg)) && fused_U && s == 0 && nseg == d + 1 && !bucket_end &&
        !getenv("FQD_GROUP_NO_REFINE")) {
        HIP_TRY(c, c->gp_crowded.reserve(CrowdedBuf::bytes(n_buckets)));
        const CrowdedBuf cb = crowded_buf(c, n_buckets);
        uint8_t *flags = cb.flags;
        HIP_TRY(c, hipMemsetAsync(cb.counts, 0, 64, c->st));
        // twice the average bucket, between 512 and 1024 (round 4 began with 1024 throughout: at config 4's skewed shape a
        // family of 1000 keys sharing a segment is 500 K pairs for ONE wave of grouped_candidates -- 1.6 ms against 0.65
        // with 566; 512 throughout marked ordinary buckets of config 3's 430-item average: 4.7 ms against 3.4)
        uint32_t limit = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(512, 2 * (U >> B)));
        if (const char *e = getenv("FQD_GROUP_CROWDED_LIMIT"))
            limit = (uint32_t)std::max(2, atoi(e));
        // crowded buckets of up to 16 K items go all pairs in tiles whatever the others do (a context that has given up
        // on the fine pieces, or has none at this distance: every crowded bucket)
        uint32_t tile_max = !tiles_possible(c, nseg) ? 0u : (c->gp_tiles || !c->gp_fine_ok) ? 0xFFFFFFFFu : 16384u;
        if (const char *e = getenv("FQD_GROUP_TILE_MAX_BUCKET"))
            tile_max = tiles_possible(c, nseg) ? (uint32_t)strtoul(e, nullptr, 10) : 0u;
        HIP_TRY(c, fqd::launch_group_mark_crowded(c->ld_start.as<uint32_t>(), bucket_end, n_buckets, limit, tile_max, flags,
                                                  cb.list, cb.list2, cb.counts, c->st));
        skip = flags;
        c->gp_crowded_bits = B;
    }
    KTIME(c, FQD_K_PAIRS, fqd::launch_grouped_candidates(items, c->ld_start.as<uint32_t>(), bucket_end, n_buckets, B,
                                                         c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap,
                                                         c->st, 0, skip));
    KTIME(c, FQD_K_VERIFY, fqd::launch_verify_candidates(c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap,
                                                         c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), shape_with_row_lengths(c), d, s, nseg,
                                                         c->edges.as<uint32_t>(), ctr + C64_EDGES, c->edge_cap,
                                                         ctr + C64_CAND_NEED, c->d_stats.as<fqd::PairStats>(), c->st,
                                                         fused_U));
    return FQD_OK;
}

// The other way behind a grouped_pass that marked crowded buckets (group.hip "the last resort"): all pairs of each
// crowded bucket, tiled. No host round trip; nothing to do is a launch of empty workgroups.
static bool tiles_possible(const fqd_ctx *c, uint32_t nseg)
{
    return fqd::group_tiles_possible(c->ks, nseg) && !getenv("FQD_GROUP_NO_TILES");
}

static int grouped_tiles(fqd_ctx *c, uint64_t U, uint32_t d, uint32_t seg0, uint32_t nseg, uint32_t B, int which)
{
    const CrowdedBuf cb = crowded_buf(c, 1u << B);
    unsigned long long *ctr = c->d_ctr64.as<unsigned long long>();
    c->route |= FQD_ROUTE_SEARCH_TILES;
    KTIME(c, FQD_K_PAIRS, fqd::launch_group_crowded_tiles(c->gp_last_items, c->ld_start.as<uint32_t>(), nullptr,
                                                          which == 2 ? cb.list2 : cb.list, cb.counts + (which == 2 ? 4 : 0),
                                                          cb.tile_prefix, (uint32_t)U, seg0, c->urecs.as<uint32_t>(),
                                                          c->ulens.as<uint32_t>(), c->ks, d, nseg, c->edges.as<uint32_t>(),
                                                          ctr + C64_EDGES, c->edge_cap, c->st));
    return FQD_OK;
}

// Behind a grouped_pass that marked crowded buckets: their keys are matched on finer segments (group.hip "crowded
// buckets"). One host round trip to learn whether there are any (only contexts that met crowded data get here).
// *ok = false: more crowded keys than the fine-item buffers hold -- the caller searches again the plain way.
static int grouped_refine(fqd_ctx *c, const uint32_t *seg_hashes, uint64_t U, uint32_t nseg, bool *ok)
{
    const uint32_t d = nseg - 1;         // (the refinement runs behind a search whose d + 1 passes were all in one partition)
    *ok = true;
    const uint32_t B = c->gp_crowded_bits, n_buckets = 1u << B;
    c->gp_crowded_bits_last = B;
    c->gp_crowded_bits = 0;
    const KeyShape sh = c->ks;
    const CrowdedBuf cb = crowded_buf(c, n_buckets);
    uint8_t *flags = cb.flags;
    unsigned long long *counts = cb.counts;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, counts, 64, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    if (!h[0] && !h[4])
        return FQD_OK;
    // The small crowded buckets (list 2) go all pairs in tiles; so do the others when their pairs are few (no items, no
    // lists: a bucket of 62 K poly-A keys of 300 nt is 1.9 G compares, about what filing and matching 120 fine items per
    // key costs); beyond that the fine items, whose cost grows with the keys and not with their square. (The tiles read
    // the partition this pass left in ld_start: they are queued BEFORE the refinement partitions its items.)
    unsigned long long tile_budget = 8000000000ull;
    if (const char *e = getenv("FQD_GROUP_TILE_BUDGET"))
        tile_budget = strtoull(e, nullptr, 10);
    const bool tiles_ok = tiles_possible(c, nseg);
    const bool small_tiles = tiles_ok && h[4] && (h[6] <= tile_budget || !c->gp_fine_ok || c->gp_tiles);
    const bool fine_tiles = tiles_ok && h[0] && h[3] + (small_tiles ? h[6] : 0) <= tile_budget;
    if (getenv("FQD_DEBUG"))
        fprintf(stderr, "[fqd] crowded buckets: %llu with %llu items (%llu pairs twice over)%s; small ones: %llu with %llu items "
                        "(%llu)%s\n", h[0], h[1], h[3], fine_tiles ? " in tiles" : "", h[4], h[5], h[6],
                small_tiles ? " in tiles" : "");
    if (small_tiles)
        FQD_TRY(grouped_tiles(c, U, d, 0, nseg, B, 2));
    if (fine_tiles)
        FQD_TRY(grouped_tiles(c, U, d, 0, nseg, B, 1));
    const uint32_t accept = (h[0] && !fine_tiles ? 1u : 0u) | (h[4] && !small_tiles ? 2u : 0u);
    if (!accept)
        return FQD_OK;
    const uint32_t pieces = fqd::group_fine_items(d);      // fine items per crowded key
    // (every crowded item could be a key of its own)
    const uint64_t key_cap = std::min<uint64_t>((accept & 1u ? h[1] : 0) + (accept & 2u ? h[5] : 0), U);
    uint64_t fine_limit = 0xFFFFFF00ull;            // (positions in the fine-item arrays are 32-bit)
    if (const char *e = getenv("FQD_GROUP_FINE_LIMIT"))      // tests: as if the crowded keys were too many
        fine_limit = strtoull(e, nullptr, 10);
    if (!c->gp_fine_ok || !pieces || key_cap * pieces >= fine_limit) {
        if (!tiles_ok) {
            *ok = false;
            return FQD_OK;
        }
        // more crowded keys than the fine items can address: all pairs of those buckets too, in tiles
        if (accept & 2u)
            FQD_TRY(grouped_tiles(c, U, d, 0, nseg, B, 2));
        if (accept & 1u)
            FQD_TRY(grouped_tiles(c, U, d, 0, nseg, B, 1));
        return FQD_OK;
    }
    HIP_TRY(c, c->gp_fine_hash.reserve(key_cap * pieces * 4 + 16));
    HIP_TRY(c, c->gp_fine_val.reserve(key_cap * pieces * 4 + 16));
    HIP_TRY(c, c->gp_seen.reserve(U / 8 + 64));          // one bit per key
    HIP_TRY(c, hipMemsetAsync(c->gp_seen.p, 0, U / 8 + 16, c->st));
    const uint32_t *items = c->gp_last_items;
    for (int which = 1; which <= 2; which++)
        if (accept & (uint32_t)which)
            HIP_TRY(c, fqd::launch_group_refine_items(items, c->ld_start.as<uint32_t>(), nullptr, which == 2 ? cb.list2 : cb.list,
                                                      counts + (which == 2 ? 4 : 0), (uint32_t)U, c->urecs.as<uint32_t>(),
                                                      c->ulens.as<uint32_t>(), sh, c->gp_seen.as<uint32_t>(),
                                                      c->gp_fine_hash.as<uint32_t>(), c->gp_fine_val.as<uint32_t>(),
                                                      counts + 7, key_cap, c->st, d));
    HIP_TRY(c, hipMemcpyAsync(h, counts, 64, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    const uint64_t n_keys = std::min<uint64_t>(h[7], key_cap), N = n_keys * pieces;
    if (getenv("FQD_DEBUG"))
        fprintf(stderr, "[fqd] crowded buckets: %llu keys matched on %u finer segments\n", (unsigned long long)n_keys, pieces);
    if (N < 2)
        return FQD_OK;
    c->route |= FQD_ROUTE_SEARCH_REFINED;
    c->gp_fine_used = true;
    uint32_t B2 = 8;
    while (B2 < 20 && (N >> B2) > 320u)
        B2++;
    const uint32_t *items2 = nullptr, *bucket_end2 = nullptr;
    FQD_TRY(fqd_api_partition_pairs(c, c->gp_fine_hash.as<uint32_t>(), N, B2, false, &items2, &bucket_end2,
                                    c->gp_fine_val.as<uint32_t>()));
    unsigned long long *cand_ctr = reinterpret_cast<unsigned long long *>(c->gp_small.as<uint32_t>() + 4096);
    unsigned long long *ctr = c->d_ctr64.as<unsigned long long>();
    // (a fine bucket of thousands of items is a family the pieces do not split: not walked by its one wave when the
    // tiles can take it -- the need it reports sends this context to them)
    HIP_TRY(c, fqd::launch_grouped_candidates(items2, c->ld_start.as<uint32_t>(), bucket_end2, 1u << B2, B2,
                                              c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap, c->st, 0, nullptr,
                                              tiles_possible(c, nseg) ? 8192u : 0u, ctr + C64_CAND_NEED));
    HIP_TRY(c, fqd::launch_group_verify_refined(c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap, c->urecs.as<uint32_t>(),
                                                c->ulens.as<uint32_t>(), sh, nseg, seg_hashes, U, B, flags,
                                                c->edges.as<uint32_t>(), ctr + C64_EDGES, c->edge_cap, ctr + C64_CAND_NEED,
                                                c->st, d, accept));
    return FQD_OK;
}

// Shared body of fqd_find_edges / fqd_find_edges_segments: passes [seg_lo, seg_hi) of the
// (max_distance+1)-way pigeonhole split (the whole range for a plain search).
static int find_edges_impl(fqd_ctx *c, int max_distance, int metric, uint32_t shard, uint32_t n_shards,
                           uint32_t seg_lo, uint32_t seg_hi, uint64_t *n_edges)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "fqd_find_edges before fqd_collapse/fqd_import_unique");
    if (max_distance < 0)
        return fail(c, FQD_E_VALUE, "max_distance should be non-negative");
    c->last_search_d = max_distance;
    if (n_shards == 0 || shard >= n_shards)
        return fail(c, FQD_E_VALUE, "bad shard");
    const KeyShape sh = c->ks;
    // Levenshtein <= 1 between keys of ONE length is Hamming <= 1 (an indel changes the length):
    // that case shares the Hamming search; everything else takes the bucketed edit search.
    bool edit_general = metric == FQD_METRIC_EDIT && !(max_distance <= 1 && !sh.ragged);
    // Levenshtein d = 1 over several key lengths: pairs of ONE length are the Hamming passes' (one edit
    // that keeps the length is a substitution); only pairs of different lengths need the edit search
    // proper, which then runs behind the passes and appends to their edges (FQD_EDIT pins a path).
    bool cross_after = false;
    if (edit_general && max_distance == 1 && n_shards == 1 && c->collapsed && c->U >= 32768 &&
        c->U < (1ull << 26) && seg_lo == 0 && seg_hi == 2 && !getenv("FQD_EDIT")) {
        edit_general = false;
        cross_after = true;
    }
    if (seg_hi > (uint32_t)max_distance + 1 || seg_lo > seg_hi)
        return fail(c, FQD_E_VALUE, "segment range outside [0, max_distance + 1]");
    if (edit_general && (seg_lo != 0 || seg_hi != (uint32_t)max_distance + 1))
        return fail(c, FQD_E_VALUE, "the bucketed edit search has no per-segment passes");
    c->stage = ST_UNIQUE;
    const uint64_t U = c->U;
    StageTimer timer(c, FQD_T_EDGES);
    c->E = 0;
    c->ms[FQD_T_PAIRS_KERNEL] = 0;
    c->launches[FQD_T_PAIRS_KERNEL] = 0;
    c->last_stats = fqd::PairStats{0, 0, 0};
    c->stats_pending = false;
    // (zeroed by the first launch of the partitioned Hamming search when that is what runs: see search_zero_pending)
    const char *pin_path = getenv("FQD_EDGES");
    const bool grouped_first = !edit_general && U >= 2 && (max_distance > 0 || !c->collapsed) && n_shards == 1 &&
                               U < 0xFFFFFF00ull && (pin_path ? !strcmp(pin_path, "grouped") : U >= 65536) &&
                               !c->search_force_sort;
    c->search_zero_pending = false;
    // The routed collapse has done pass 0 of exactly this search (fqd::Pass0): its pairs are in the edge list, the
    // segment hashes it wrote start at segment 1, and the passes below start there too.
    // (seg_hi == 1: a rank of a sharded job asks for pass 0 alone -- fqd_find_edges_segments(0, 1) behind
    // fqd_collapse_owner_slabs with owner routing: nothing is left to search, the counters are read and that is it)
    const bool pass0_held = c->pass0_done && !edit_general && seg_lo == 0 && n_shards == 1 && max_distance >= 1 &&
                            c->pass0_nseg == (uint32_t)max_distance + 1 &&
                            (seg_hi == (uint32_t)max_distance + 1 || seg_hi == 1) &&
                            c->seg_hashes_nseg == c->pass0_nseg && c->seg_hashes_first == 1;
    c->pass0_done = false;
    if (pass0_held) {
        seg_lo = 1;
        c->route |= FQD_ROUTE_PASS0_CONTINUED;
    }
    if (!c->search_is_retry)
        c->route &= ~(FQD_ROUTE_SEARCH_GROUPED | FQD_ROUTE_SEARCH_SORT | FQD_ROUTE_SEARCH_EDIT | FQD_ROUTE_SEARCH_RETRIED);
    if (!grouped_first && !pass0_held) {
        FQD_TRY(zero_ctr64(c, C64_EDGES));
        HIP_TRY(c, hipMemsetAsync(c->d_stats.p, 0, FQD_STAT_SLOTS * sizeof(fqd::PairStats), c->st));
    }
    if (edit_general && U >= 2 && (max_distance > 0 || !c->collapsed)) {
        bool grouped_done = false;
        c->route |= FQD_ROUTE_SEARCH_EDIT;
        if (n_shards == 1 && c->collapsed)
            FQD_TRY(find_edges_edit_grouped(c, (uint32_t)max_distance, &grouped_done));
        if (!grouped_done)
            FQD_TRY(find_edges_edit(c, (uint32_t)max_distance, shard, n_shards));
    } else if (U >= 2 && (max_distance > 0 || !c->collapsed)) {
        // with d >= max_len every segment split has empty segments: still correct (all keys of a
        // length share the empty segment's bucket), just quadratic.
        const uint32_t d = (uint32_t)max_distance;
        const uint32_t nseg = d + 1;
        HIP_TRY(c, c->seg_hashes.reserve((size_t)(seg_hi - seg_lo) * U * 4 + 16));
        HIP_TRY(c, c->sorted_hash.reserve(U * 4 + 16));
        HIP_TRY(c, c->sorted_uid.reserve(U * 4 + 16));
        HIP_TRY(c, c->uid_iota.reserve(U * 4 + 16));
        // (the LDS collapse inside fqd_cluster[_keys] has written them already, on its way out)
        if (!pass0_held && !(c->seg_hashes_nseg == nseg && c->seg_hashes_first == 0 && seg_lo == 0))
            KTIME(c, FQD_K_SEG_HASH, fqd::launch_segment_hashes(c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), U, sh,
                                                                nseg, seg_lo, seg_hi, 0, c->seg_hashes.as<uint32_t>(),
                                                                c->st));
        c->seg_hashes_nseg = 0;
        c->seg_hashes_first = 0;
        if (c->edge_cap < 1024 || !c->edges.p) {
            c->edge_cap = std::max<uint64_t>(1024, U);
            HIP_TRY(c, c->edges.reserve(c->edge_cap * 8));
        }
        c->edge_cap = c->edges.cap / 8;
        if (pass0_held)
            c->edge_cap = std::min<uint64_t>(c->edge_cap, c->pass0_edge_cap);     // (what pass 0 wrote against)
        unsigned long long have = 0;
        if (n_shards > 1) {
            HIP_TRY(c, c->sel_hash.reserve(U * 4 + 16));
            HIP_TRY(c, c->sel_uid.reserve(U * 4 + 16));
        }
        // Grouping by partition (group.hip) unless a bucket shard was asked for or the table is small
        // (FQD_EDGES=sort|grouped pins the path for tests).
        const char *pin = getenv("FQD_EDGES");
        bool grouped = n_shards == 1 && U < 0xFFFFFF00ull && (pin ? !strcmp(pin, "grouped") : U >= 65536) &&
                       !c->search_force_sort;
        // Candidate pairs are listed before they are verified; a segment value shared by very many
        // keys (all of them pairwise candidates) would need a list beyond this budget: the search
        // then runs again on the sort path, which verifies in place and needs no list.
        uint64_t cand_budget = std::max<uint64_t>(8 * U, 1ull << 24);
        if (const char *e = getenv("FQD_GROUP_CAND_BUDGET"))
            cand_budget = strtoull(e, nullptr, 10);
        bool iota_ready = false;
        c->search_keeps_edges = pass0_held;
        if (grouped_first && grouped && seg_hi > seg_lo)
            c->search_zero_pending = true;           // edges, candidate need, slab flag, statistics: with the partition's first launch
        else
            FQD_TRY(zero_ctr64(c, C64_CAND_NEED, 2));    // ... and C64_SLAB
        // All d+1 passes are queued without a host round trip; the edge count is read ONCE at the
        // end. If the passes overflowed the edge buffer (the count still says how many edges there
        // are), the buffer is grown to the known need and the whole search runs again.
        // all passes of a plain search in one (see grouped_pass): positions in the hash array must fit 32 bits
        const uint32_t n_pass = seg_hi - seg_lo;
        const bool fuse_passes = grouped && (seg_lo == 0 || pass0_held) && seg_hi == nseg && n_pass >= 2 && nseg <= 8 &&
                                 (uint64_t)n_pass * U < 0xFFFFFF00ull && !getenv("FQD_GROUP_NO_FUSED_PASSES");
        bool used_refine = false;
        for (int attempt = 0;; attempt++) {
            c->route |= (grouped ? FQD_ROUTE_SEARCH_GROUPED : FQD_ROUTE_SEARCH_SORT) | (attempt ? FQD_ROUTE_SEARCH_RETRIED : 0u);
            if (fuse_passes && grouped) {
                FQD_TRY(grouped_pass(c, c->seg_hashes.as<uint32_t>(), (uint64_t)n_pass * U, d, seg_lo, nseg, (uint32_t)U));
                if (c->gp_crowded_bits && (c->gp_tiles || !c->gp_fine_ok) && tiles_possible(c, nseg) &&
                    !getenv("FQD_GROUP_TILE_MAX_BUCKET")) {
                    // (every crowded bucket was put on list 2: no host round trip)
                    const uint32_t B_marked = c->gp_crowded_bits;
                    c->gp_crowded_bits = 0;
                    FQD_TRY(grouped_tiles(c, U, d, seg_lo, nseg, B_marked, 2));
                } else if (c->gp_crowded_bits) {
                    bool refined = true;
                    c->gp_fine_used = false;
                    FQD_TRY(grouped_refine(c, c->seg_hashes.as<uint32_t>(), U, nseg, &refined));
                    used_refine = used_refine || c->gp_fine_used;
                    if (!refined) {
                        // more crowded keys than the fine items can address: the whole search once more on the sort path,
                        // which compares a crowded bucket's keys in place and needs no items at all (the reference's trie
                        // takes any distribution, _triemodule.c:380-495: this must never be an error)
                        if (getenv("FQD_DEBUG"))
                            fprintf(stderr, "[fqd] crowded buckets: too many keys for the fine items; the search runs again on the sort path\n");
                        c->route = (c->route & ~FQD_ROUTE_PASS0_CONTINUED) | FQD_ROUTE_SEARCH_RETRIED;
                        c->search_keeps_edges = false;
                        c->seg_hashes_nseg = 0;
                        timer.stop();
                        c->search_is_retry = true;
                        c->search_force_sort = true;
                        const int rc_again = find_edges_impl(c, max_distance, metric, shard, n_shards,
                                                             pass0_held ? 0u : seg_lo, seg_hi, n_edges);
                        c->search_force_sort = false;
                        c->search_is_retry = false;
                        return rc_again;
                    }
                }
            }
            for (uint32_t s = seg_lo; s < seg_hi && !(fuse_passes && grouped); s++) {
                const uint32_t *pass_hashes = c->seg_hashes.as<uint32_t>() + (size_t)(s - seg_lo) * U;
                uint64_t m = U;  // entries this rank sorts and searches in this pass
                if (n_shards > 1) {
                    // only this rank's buckets go through the sort and the pair kernel
                    FQD_TRY(zero_ctr64(c, C64_SUM));
                    HIP_TRY(c, fqd::launch_select_shard(pass_hashes, U, shard,
                                                        n_shards, c->sel_hash.as<uint32_t>(),
                                                        c->sel_uid.as<uint32_t>(),
                                                        c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
                    unsigned long long got = 0;
                    FQD_TRY(read_ctr64(c, C64_SUM, &got));
                    m = got;
                    FQD_TRY(sort_u32_pairs(c, c->sel_hash.as<uint32_t>(), c->sorted_hash.as<uint32_t>(),
                                           c->sel_uid.as<uint32_t>(), c->sorted_uid.as<uint32_t>(), m));
                } else if (grouped) {
                    FQD_TRY(grouped_pass(c, pass_hashes, U, d, s, nseg));
                    continue;
                } else {
                    if (!iota_ready) {
                        HIP_TRY(c, fqd::launch_iota_u32(c->uid_iota.as<uint32_t>(), U, c->st));
                        iota_ready = true;
                    }
                    FQD_TRY(sort_u32_pairs(c, pass_hashes,
                                           c->sorted_hash.as<uint32_t>(), c->uid_iota.as<uint32_t>(),
                                           c->sorted_uid.as<uint32_t>(), U));
                }
                KTIME(c, FQD_K_PAIRS, fqd::launch_bucket_pairs(
                               c->sorted_hash.as<uint32_t>(), c->sorted_uid.as<uint32_t>(), m,
                               c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), shape_with_row_lengths(c), d, s, nseg, 0, 1,
                               c->edges.as<uint32_t>(), c->d_ctr64.as<unsigned long long>() + C64_EDGES, c->edge_cap,
                               c->d_stats.as<fqd::PairStats>(), c->st));
            }
            unsigned long long ctrs[C64_SLAB + 1] = {0};
            // the counters come back while the GPU sets up what fqd_cluster runs next on the unique
            // table (nothing of that depends on the edges)
            FQD_TRY(queue_read_ctr64(c, 0, C64_SLAB + 1));
            if (pass0_held)          // (the compaction raised it AFTER the collapse's own read-back was queued)
                FQD_TRY(queue_read_u32(c, c->d_ctr32.as<uint32_t>() + C_P0, 0));
            FQD_TRY(queued_reads_mark(c));
            if (c->preinit_method >= 0) {
                FQD_TRY(fqd_api_graph_preinit(c, c->preinit_method));
                c->preinit_method = -1;
            }
            FQD_TRY(queued_reads_wait(c));
            taken_ctr64(c, ctrs, C64_SLAB + 1);
            const unsigned long long now = ctrs[C64_EDGES], cand_need = grouped ? ctrs[C64_CAND_NEED] : 0;
            const bool slab_over = grouped && ctrs[C64_SLAB] != 0;
            if (slab_over) {             // a level-2 slab overflowed: exact bucket sizes from now on
                c->gp_slab_off = true;
                FQD_TRY(zero_ctr64(c, C64_SLAB));
            }
            // pass 0 by the collapse's compaction, found incomplete now: a bucket of more rows than a wave holds
            const bool pass0_short = pass0_held && taken_u32(c, 0) != 0;
            if (!slab_over && now <= c->edge_cap && cand_need <= c->gp_cand_cap && !pass0_short) {
                have = now;
                break;
            }
            if (attempt > 2)
                return fail(c, FQD_E_RUNTIME, "edge buffer kept overflowing");
            if (now > c->edge_cap) {
                c->edges.release();
                HIP_TRY(c, c->edges.reserve((size_t)(now + now / 8 + 1024) * 8));
                c->edge_cap = c->edges.cap / 8;
            }
            if (pass0_held) {
                // the pairs of pass 0 are gone with the counters: the whole search once more, every pass here
                c->route = (c->route & ~FQD_ROUTE_PASS0_CONTINUED) | FQD_ROUTE_SEARCH_RETRIED;
                c->search_keeps_edges = false;
                c->seg_hashes_nseg = 0;
                timer.stop();
                c->search_is_retry = true;
                const int rc_again = find_edges_impl(c, max_distance, metric, shard, n_shards, 0, seg_hi, n_edges);
                c->search_is_retry = false;
                return rc_again;
            }
            if (cand_need > c->gp_cand_cap) {
                if (cand_need > cand_budget && used_refine && !c->gp_tiles && tiles_possible(c, nseg)) {
                    // the finer pieces left a crowded family in one piece (its candidate pairs are beyond every
                    // budget): from now on this context compares crowded buckets all pairs, tiled
                    c->gp_tiles = true;
                } else if (cand_need > cand_budget) {
                    grouped = false;
                } else {
                    c->gp_cands.release();
                    HIP_TRY(c, c->gp_cands.reserve((size_t)(cand_need + cand_need / 8 + 1024) * 8));
                    c->gp_cand_cap = c->gp_cands.cap / 8;
                }
            }
            FQD_TRY(zero_ctr64(c, C64_CAND_NEED));
            FQD_TRY(zero_ctr64(c, C64_EDGES));
            HIP_TRY(c, hipMemsetAsync(c->d_stats.p, 0, FQD_STAT_SLOTS * sizeof(fqd::PairStats), c->st));
        }
        c->E = have;
        c->stats_pending = true;     // the 64 stat slots are summed when fqd_edge_stats asks
    }
    if (cross_after && U >= 2) {
        bool cross_done = false;
        c->pre_zero_tail = false;        // (the edit search uses the counters the graph set-up has cleared)
        c->route |= FQD_ROUTE_SEARCH_EDIT;
        FQD_TRY(find_edges_edit_grouped(c, 1, &cross_done, true, true));
        if (!cross_done) {           // (cannot happen for d = 1 below 2^26 keys; the sorted search redoes everything)
            FQD_TRY(zero_ctr64(c, C64_EDGES));
            c->E = 0;
            FQD_TRY(find_edges_edit(c, 1, shard, n_shards));
        }
    }
    timer.stop();
    c->stage = ST_EDGES;
    if (n_edges)
        *n_edges = c->E;
    return FQD_OK;
}

int fqd_find_edges(fqd_ctx *c, int max_distance, int metric, uint32_t shard, uint32_t n_shards, uint64_t *n_edges)
{
    if (max_distance < 0)
        return fail(c, FQD_E_VALUE, "max_distance should be non-negative");
    return find_edges_impl(c, max_distance, metric, shard, n_shards, 0, (uint32_t)max_distance + 1, n_edges);
}

int fqd_find_edges_segments(fqd_ctx *c, int max_distance, uint32_t seg_lo, uint32_t seg_hi, uint64_t *n_edges)
{
    if (max_distance < 0)
        return fail(c, FQD_E_VALUE, "max_distance should be non-negative");
    return find_edges_impl(c, max_distance, FQD_METRIC_HAMMING, 0, 1, seg_lo, seg_hi, n_edges);
}

}  // extern "C"
