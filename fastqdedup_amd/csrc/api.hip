// api.hip -- C ABI, part 1: context life cycle, packing, the exact-duplicate collapse, the
// whole-path call fqd_cluster, single calls of the reference surface, the quality gate and the
// measurement getters. The search is api_search.hip, components/dissection/kept list api_graph.hip,
// the multi-GPU exchange entry points api_exchange.hip.
#include "api_ctx.h"

namespace {

int upload_lut(fqd_ctx *c, const uint8_t *lut256)
{
    if (c->lut_valid && !memcmp(c->lut_on_device, lut256, 256))
        return FQD_OK;
    memcpy(c->lut_on_device, lut256, 256);
    HIP_TRY(c, hipMemcpyAsync(c->d_lut.p, c->lut_on_device, 256, hipMemcpyHostToDevice, c->st));
    c->lut_valid = true;
    return FQD_OK;
}

void build_alphabet(fqd_ctx *c, const uint8_t *present128, uint8_t *lut256)
{
    memset(lut256, 0xFF, 256);
    memset(&c->shape.alphabet, 0, sizeof c->shape.alphabet);
    uint32_t a = 0;
    for (int b = 0; b < 128; b++)
        if (present128[b]) {
            lut256[b] = (uint8_t)a;
            c->shape.alphabet[a] = (uint8_t)b;
            a++;
        }
    c->shape.alphabet_size = a;
    uint32_t k = 1;
    while ((1u << k) < a)
        k++;
    c->shape.planes = k;
}

int set_geometry(fqd_ctx *c, uint32_t max_len, int ragged)
{
    c->shape.max_len = max_len;
    c->shape.ragged = ragged ? 1 : 0;
    c->shape.words = std::max<uint32_t>(1, (max_len + 31) / 32);
    const uint64_t kw = (uint64_t)c->shape.planes * c->shape.words;
    if (kw > 14000)
        return fail(c, FQD_E_VALUE, "key of " + std::to_string(max_len) + " bases is too long for the LDS pack tile");
    c->shape.stride_words = (uint32_t)((kw + 3) & ~3ull);
    // Records of 3 uint4 (keys of 65-128 nt over "ACGNT": 48 bytes) take a 64-byte stride: a record gather is then ONE
    // 64-byte sector instead of 1.75 on average, and the record is a power-of-two number of uint4 -- the cooperative
    // verification and the segment hashes inside the compaction apply (config 2: 10 M x 100 nt). 16 bytes more per
    // read out of the pack kernel; records of 5-7 uint4 are left alone (up to 60 % more bytes for the same effect).
    if (c->shape.stride_words == 12 && !getenv("FQD_NO_STRIDE_PADDING"))
        c->shape.stride_words = 16;
    c->ks.planes = c->shape.planes;
    c->ks.words = c->shape.words;
    c->ks.stride = c->shape.stride_words;
    c->ks.max_len = max_len;
    c->ks.ragged = c->shape.ragged;
    return FQD_OK;
}

int scan_keys_device(fqd_ctx *c, const uint8_t *d_bytes, const uint64_t *d_offsets, uint64_t n, uint64_t n_bytes,
                     uint32_t fixed_len, uint8_t *present128, uint32_t *max_len, int *ragged)
{
    HIP_TRY(c, c->d_present.reserve(256 * 4));
    HIP_TRY(c, hipMemsetAsync(c->d_present.p, 0, 256 * 4, c->st));
    HIP_TRY(c, fqd::launch_scan_bytes(d_bytes, n_bytes, c->d_present.as<uint32_t>(), c->st));
    uint32_t mm[2] = {fixed_len, fixed_len};
    if (d_offsets && n) {
        const uint32_t init[2] = {0xFFFFFFFFu, 0u};
        HIP_TRY(c, hipMemcpyAsync(c->d_ctr32.as<uint32_t>() + C_MINLEN, init, 8, hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, fqd::launch_scan_lens(d_offsets, n, c->d_ctr32.as<uint32_t>() + C_MINLEN, c->st));
        HIP_TRY(c, hipMemcpyAsync(mm, c->d_ctr32.as<uint32_t>() + C_MINLEN, 8, hipMemcpyDeviceToHost, c->st));
    } else if (!n) {
        mm[0] = mm[1] = 0;
    }
    uint32_t seen[256];
    HIP_TRY(c, hipMemcpyAsync(seen, c->d_present.p, sizeof seen, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    for (int b = 128; b < 256; b++)
        if (seen[b])
            return fail(c, FQD_E_VALUE, "Sequence must consist only of ASCII characters");
    for (int b = 0; b < 128; b++)
        present128[b] = seen[b] ? 1 : 0;
    *max_len = mm[1];
    *ragged = mm[0] != mm[1];
    return FQD_OK;
}

uint64_t total_bytes_of(const uint64_t *offsets_host_or_null, uint64_t n, uint32_t fixed_len)
{
    return offsets_host_or_null ? offsets_host_or_null[n] : n * (uint64_t)fixed_len;
}

int hash_bits_from_env()
{
    const char *e = getenv("FQD_HASH_BITS");  // tests narrow the hash to force collisions
    if (!e)
        return 32;
    int b = atoi(e);
    return b < 1 ? 1 : (b > 32 ? 32 : b);
}

// Sort-free collapse for records of one uint4 (collapse_lds.hip). Returns FQD_OK with
// *done = false when it does not apply or a bucket's table overflowed (caller falls back).
// Record hashes of the packed reads, computed on demand after an import (the LDS collapse works
// from the records alone; the sort-based collapse and the exports need the array).
int ensure_hashes(fqd_ctx *c)
{
    if (c->hashes_valid)
        return FQD_OK;
    HIP_TRY(c, c->hashes.reserve((size_t)c->n * 4 + 16));
    HIP_TRY(c, fqd::launch_hash_records(c->recs.as<uint32_t>(), c->lens.as<uint32_t>(), c->n, c->ks,
                                        c->hashes.as<uint32_t>(), c->st));
    c->hashes_valid = true;
    return FQD_OK;
}

// `fused` != NULL: level 1 was done by the pack kernel itself (fqd_cluster_keys): c->ld_part holds
// fused->parts slab segments [seg_start[s], cursor[s]) in c->ld_seg, 2^sub_bits of them per
// level-1 bin, and the reads exist nowhere else -- any overflow then ends the attempt (*done =
// false) and the caller starts over with the plain pack.
struct FusedLevel1 {
    uint32_t parts, sub_bits, B;
    uint32_t pack_bad = 0;     // out: the pack kernel met a byte outside the alphabet
    // slabs received from several ranks (fqd_collapse_owner_slabs): level 1 has level1_bits hash bits
    // (0: the usual 8), segment -> part is (segment >> sub_bits) & part_mask, and level 2 stamps the
    // sender (segment / stamp_div) above the read index every record carries; the segment tables in
    // c->ld_seg are filled by the caller (tables_ready)
    uint32_t level1_bits = 0, part_mask = 0xFFFFFFFFu, stamp_div = 0;
    bool tables_ready = false;
    // compact records (fqd_internal.h Rec12; squeeze code 1 or 2): level 2 writes 12-byte items, keys with an N
    // take the side path (c->ld_side / c->ld_side_table, sized by the caller: side_slabs x side_cap records, a
    // hash table of side_slots slots)
    uint32_t compact = 0, side_slabs = 0, side_cap = 0, side_slots = 0;
    uint32_t slab_cap1 = 0;    // the capacity of every level-1 slab when they all have one (the pack kernel's own slabs)
    // the caller has queued the slab starts of level 2 (c->ld_start / c->ld_cursor for 2^B buckets of the capacity
    // collapse_lds computes) and of the side path already -- fqd_cluster_keys does, BEFORE the pack kernel, so that
    // nothing but the pack kernel stands between the key bytes and level 2
    bool starts_ready = false;
    // the routed collapse: bins by segment 0 (route_mask), search pass 0 in the compaction (p0.mask != 0)
    uint32_t route_mask = 0, group_at = 0;      // group_at: the dedupe's group totals start at c->ld_hist + group_at
    fqd::Pass0 p0;
    // the spill list (fqd::PackScatter::spill; compact == 1, one sender): spill_cap records behind the side slabs in
    // c->ld_side, its cursor and the level-1 marks behind the side cursors. can_spill: this attempt could have had one
    // -- a full slab then turns c->heavy_keys on instead of the fused path off.
    uint32_t spill_cap = 0;
    uint32_t *spill_cursor = nullptr, *l1_over = nullptr;
    bool can_spill = false;
    uint32_t spill_used = 0;   // out: records that went to the spill list
    bool side_parked = false;  // the pack kernel has sent the keys with an N to the side slabs itself (PackScatter::side_recs):
                               // the side path is queued BEFORE level 2 and runs beside it
};

// Bucket bits of the LDS collapse for n reads: ~400-800 reads per bucket (2x fewer workgroups than
// at 200-400, longer runs)
uint32_t lds_bucket_bits(uint64_t n)
{
    uint32_t B = 8;
    while (B < 18 && (n >> B) > 800)
        B++;
    if (const char *e = getenv("FQD_LDS_BUCKET_BITS"))  // tests: few buckets => table overflow => fallback
        B = (uint32_t)std::max(1, std::min(18, atoi(e)));
    return B;
}

int collapse_lds(fqd_ctx *c, const uint32_t *d_w, IdSource d_ids, bool *done, FusedLevel1 *fused = nullptr)
{
    *done = false;
    c->seg_hashes_nseg = 0;
    c->seg_hashes_first = 0;
    c->pass0_done = false;
    const uint64_t n = c->n;
    const KeyShape sh = c->ks;
    const char *force = getenv("FQD_COLLAPSE");  // "sort" / "lds": tests pin a path
    if (force && !strcmp(force, "sort"))
        return FQD_OK;
    if (sh.ragged || sh.stride != 4 || sh.planes * sh.words > 3 || n >= 0xFFFFFF00ull)
        return FQD_OK;
    if (n < 32768 && !(force && !strcmp(force, "lds")))
        return FQD_OK;
    const uint32_t B = fused ? fused->B : lds_bucket_bits(n);
    const uint32_t B1 = fused && fused->level1_bits ? fused->level1_bits : std::min<uint32_t>(B, 8), B2 = B - B1;
    const uint32_t bins1 = 1u << B1, bins2 = 1u << B2, n_buckets = 1u << B;
    const uint32_t compact = fused ? fused->compact : 0;
    const uint32_t kw = sh.planes * sh.words, tile = compact ? fqd::part_tile_size12() : fqd::part_tile_size();
    const uint32_t tiles1 = (uint32_t)((n + tile - 1) / tile), max_tiles2 = tiles1 + bins1;
    HIP_TRY(c, c->ld_hist.reserve((size_t)n_buckets * 4 + 1024 * 4));
    HIP_TRY(c, c->ld_hist_incl.reserve((size_t)n_buckets * 4 + 1024 * 4));
    HIP_TRY(c, c->ld_start.reserve(((size_t)n_buckets + 1) * 4 + 16));
    HIP_TRY(c, c->ld_cursor.reserve((size_t)n_buckets * 4 + 1024 * 4));
    HIP_TRY(c, c->ld_unique.reserve((size_t)n_buckets * 4 + 16));
    HIP_TRY(c, c->ld_unique_incl.reserve((size_t)n_buckets * 4 + 16));
    HIP_TRY(c, c->ld_small.reserve(4096 * 4));
    // Level 2 in SLAB mode: every bucket gets a fixed slab of 1.5 x the mean + 64 slots and an atomic
    // cursor, so no histogram pass over the 16-byte records is needed (0.2 ms of 1.85). A key with
    // hundreds of copies overfills its slab: the scatter notices, and level 2 runs again with the
    // exact histogram (and stays exact for this context).
    uint32_t slab_cap = 0;
    if (B2 && (fused || (!c->slab_off && !getenv("FQD_LDS_NO_SLABS")))) {
        slab_cap = (uint32_t)(((n >> B) * 3 / 2 + 64 + 3) & ~3ull);
        if ((uint64_t)slab_cap * n_buckets + n >= 0xFFFFFF00ull)   // (a cursor may run n past its slab)
            slab_cap = 0;
    }
    const uint64_t slots = slab_cap ? (uint64_t)slab_cap * n_buckets : n;
    if (fused && !slab_cap)
        return FQD_OK;
    if (!fused)
        HIP_TRY(c, c->ld_part.reserve(n * 16 + 16));
    HIP_TRY(c, c->ld_tmp_rec.reserve(slots * 16 + 16));
    HIP_TRY(c, c->ld_part2.reserve(slots * (compact ? 12 : 16) + 16));
    if (!compact) {
        HIP_TRY(c, c->ld_tmp_count.reserve(slots * 4 + 16));
        HIP_TRY(c, c->ld_tmp_first.reserve(slots * 4 + 16));
    }
    // small device tables: [0] seg_start1 (2) | [8] tile_start1 (2) | [16] start1 (257) | [512] tile_start2 (257)
    uint32_t *small = c->ld_small.as<uint32_t>();
    uint32_t *seg1 = small, *tiles1_d = small + 8, *start1 = small + 16, *tiles2_d = small + 512;
    const uint32_t seg1_h[2] = {0u, (uint32_t)n}, tiles1_h[2] = {0u, tiles1};
    if (!fused) {
        HIP_TRY(c, hipMemcpyAsync(seg1, seg1_h, 8, hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, hipMemcpyAsync(tiles1_d, tiles1_h, 8, hipMemcpyHostToDevice, c->st));
        // ---- level 1: 2^B1 parts by the top B1 hash bits. Counts go to a (bin x tile) matrix whose
        // scan gives every (tile, bin) its output position (no atomics on 2^B1 hot counters).
        const size_t matrix = (size_t)bins1 * tiles1;
        HIP_TRY(c, c->ld_matrix.reserve(matrix * 4 + 16));
        HIP_TRY(c, c->ld_matrix_incl.reserve(matrix * 4 + 16));
        KTIME(c, FQD_K_PART_HIST1, fqd::launch_part_hist(true, c->hashes_valid ? c->hashes.as<uint32_t>() : nullptr, c->recs.as<uint32_t>(), seg1, tiles1_d, 1, tiles1,
                                         32 - B1, bins1, kw, sh.max_len, c->ld_matrix.as<uint32_t>(), c->st));
        FQD_TRY(scan_u32(c, c->ld_matrix.as<uint32_t>(), c->ld_matrix_incl.as<uint32_t>(), matrix));
        HIP_TRY(c, fqd::launch_matrix_starts(c->ld_matrix_incl.as<uint32_t>(), bins1, tiles1, start1, c->st));
        // received reads without weights: (segment, local index) travels in the record (IdSource)
        IdSource packed;
        if (d_ids.packed_bits && !d_w)
            packed = d_ids;
        else
            d_ids.packed_bits = 0;
        KTIME(c, FQD_K_PART_SCATTER1, fqd::launch_part_scatter(true, c->hashes.as<uint32_t>(), c->recs.as<uint32_t>(), seg1, tiles1_d, 1,
                                            tiles1, 32 - B1, bins1, kw, sh.max_len, c->ld_matrix_incl.as<uint32_t>(),
                                            c->ld_part.as<uint32_t>(), c->st, packed));
    }
    // the fused pack's slab segments: [0] seg_start (parts + 1) | cursor = seg_end (parts) | tile_start (parts + 1)
    const uint32_t f_parts = fused ? fused->parts : 0;
    uint32_t *f_seg_start = fused ? c->ld_seg.as<uint32_t>() : nullptr;
    uint32_t *f_seg_end = fused ? f_seg_start + (f_parts + 4) : nullptr;
    uint32_t *f_tiles = fused ? f_seg_end + (f_parts + 4) : nullptr;
    // compact records, squeeze 1: the side slabs' cursors (and their starts, unused) live behind the hash table
    fqd::SideSlabs side;
    if (compact == 1) {
        uint32_t *tail = c->ld_side_table.as<uint32_t>() + fqd::side_table_words(fused->side_slots);
        side.recs = c->ld_side.as<uint4>();
        side.cursor = tail;
        side.n_slabs = fused->side_slabs;
        side.cap = fused->side_cap;
        side.overflow = c->d_ctr32.as<uint32_t>() + C_BAD;
        if (fused->spill_cap) {
            side.spill_at = side.n_slabs * side.cap;
            side.spill_cap = fused->spill_cap;
            side.spill_cursor = fused->spill_cursor;
        }
    }
    const bool spill = side.spill_cursor != nullptr;
    const uint32_t *parted = c->ld_part.as<uint32_t>();
    uint32_t U32 = 0, overflow = 0, early_nseg = 0;
    bool side_pending = false;
    for (;;) {
        const uint32_t *bucket_end = nullptr;
        if (!fused)          // (the fused pack has already run and may have raised bit 4)
            FQD_TRY(zero_ctr32(c, C_BAD));
        if (B2 == 0) {
            HIP_TRY(c, hipMemcpyAsync(c->ld_start.p, start1, ((size_t)bins1 + 1) * 4, hipMemcpyDeviceToDevice, c->st));
        } else {
            // ---- level 2: every part into 2^B2 buckets by the next B2 hash bits
            // (the pack kernel left seg_start / seg_end = cursors there; received slabs: the caller did.)
            // Slabs of one capacity (fqd_cluster_keys): no tile table -- every slab gets the tiles a full one has
            const uint32_t f_cap = fused && !fused->tables_ready ? fused->slab_cap1 : 0u;
            const uint32_t f_grid = f_cap ? f_parts * ((f_cap + tile - 1) / tile) : tiles1 + f_parts;
            if (fused && f_cap)
                f_tiles = nullptr;
            else if (fused)
                HIP_TRY(c, fqd::launch_slab_tile_starts(f_seg_start, f_seg_end, f_parts, f_tiles, c->st, compact != 0));
            else
                HIP_TRY(c, fqd::launch_tile_starts(start1, bins1, tiles2_d, c->st));
            if (slab_cap) {
                if (!(fused && fused->starts_ready))
                    HIP_TRY(c, fqd::launch_slab_starts(n_buckets, slab_cap, c->ld_start.as<uint32_t>(),
                                                       c->ld_cursor.as<uint32_t>(), c->st));
                bucket_end = c->ld_cursor.as<uint32_t>();
            } else {
                HIP_TRY(c, hipMemsetAsync(c->ld_hist.p, 0, (size_t)n_buckets * 4, c->st));
                KTIME(c, FQD_K_PART_HIST2, fqd::launch_part_hist(false, nullptr, c->ld_part.as<uint32_t>(), start1, tiles2_d, bins1, max_tiles2,
                                                 32 - B, bins2, kw, sh.max_len, c->ld_hist.as<uint32_t>(), c->st));
                FQD_TRY(scan_u32(c, c->ld_hist.as<uint32_t>(), c->ld_hist_incl.as<uint32_t>(), n_buckets));
                HIP_TRY(c, fqd::launch_bucket_starts(c->ld_hist_incl.as<uint32_t>(), n_buckets, c->ld_start.as<uint32_t>(),
                                                     c->ld_cursor.as<uint32_t>(), c->st));
            }
            if (compact) {
                if (compact == 1 && !fused->starts_ready)
                    HIP_TRY(c, fqd::launch_slab_starts(side.n_slabs, side.cap, side.cursor + side.n_slabs, side.cursor,
                                                       c->st));
                const bool side_early = compact == 1 && !spill && fused->side_parked;
                auto queue_side_path = [&]() -> int {
                    HIP_TRY(c, hipEventRecord(c->ev_fork, c->st));
                    HIP_TRY(c, hipStreamWaitEvent(c->st_side, c->ev_fork, 0));
                    HIP_TRY(c, fqd::launch_side_collapse(
                                   side.recs, side.cursor, 0, side.n_slabs, side.cap, d_w, c->ld_side_table.as<uint32_t>(),
                                   fused->side_slots, c->ld_side_table.as<uint32_t>() + 3 * (size_t)fused->side_slots,
                                   c->urecs.as<uint32_t>(), c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(),
                                   c->d_ctr32.as<uint32_t>() + C_SIDE, c->d_ctr32.as<uint32_t>() + C_BAD, c->st_side,
                                   fused->p0, fused->stamp_div ? d_ids : IdSource()));
                    HIP_TRY(c, hipEventRecord(c->ev_join, c->st_side));
                    side_pending = true;
                    return FQD_OK;
                };
                // (the pack kernel has filled the side slabs: their keys are collapsed beside level 2)
                if (side_early)
                    FQD_TRY(queue_side_path());
                KTIME(c, FQD_K_PART_SCATTER12, fqd::launch_part_scatter12(
                          c->ld_part.as<uint32_t>(), compact, side, f_seg_start, f_tiles, f_parts, f_grid,
                          32 - B, bins2, c->ld_cursor.as<uint32_t>(), c->ld_part2.as<fqd::Rec12>(), c->st, slab_cap,
                          c->d_ctr32.as<uint32_t>() + C_BAD, f_seg_end, fused->sub_bits, fused->route_mask,
                          fused->part_mask, fused->stamp_div, fused->stamp_div ? d_ids.packed_bits : 0u,
                          fused->stamp_div ? d_ids.stamp_map : nullptr));
                // the keys with an N: collapsed apart, to the head of the unique table (few: a table in global
                // memory) -- on the context's second stream, beside the dedupe of the other keys: four short
                // kernels (0.05 ms in a row) that the compaction, not the dedupe, waits for
                if (spill) {
                    // (the dedupe below merges into the table: the side slabs and the spill list go in first)
                    HIP_TRY(c, fqd::launch_side_begin(side, d_w, c->ld_side_table.as<uint32_t>(), fused->side_slots, c->st));
                } else if (compact == 1 && !side_early) {
                    FQD_TRY(queue_side_path());
                }
            } else if (fused)
                KTIME(c, FQD_K_PART_SCATTER2, fqd::launch_part_scatter(
                          false, nullptr, c->ld_part.as<uint32_t>(), f_seg_start, f_tiles, f_parts, f_grid,
                          32 - B, bins2, kw, sh.max_len, c->ld_cursor.as<uint32_t>(), c->ld_part2.as<uint32_t>(), c->st,
                          fused->stamp_div ? d_ids : IdSource(), slab_cap, c->d_ctr32.as<uint32_t>() + C_BAD, f_seg_end,
                          fused->sub_bits, fused->part_mask, fused->stamp_div));
            else
                KTIME(c, FQD_K_PART_SCATTER2, fqd::launch_part_scatter(
                          false, nullptr, c->ld_part.as<uint32_t>(), start1, tiles2_d, bins1, max_tiles2, 32 - B, bins2,
                          kw, sh.max_len, c->ld_cursor.as<uint32_t>(), c->ld_part2.as<uint32_t>(), c->st, IdSource(),
                          slab_cap, c->d_ctr32.as<uint32_t>() + C_BAD));
            parted = c->ld_part2.as<uint32_t>();
        }
        // compact records behind a fused pack that zeroed the group totals (c->ld_hist) with its slab starts: the
        // dedupe adds every bucket's unique count to its group of 256 buckets, the compaction finds its offsets from
        // those -- no scan kernel (18 us on one workgroup, and its two hand-overs) between the two
        const uint32_t n_groups = std::max(n_buckets >> 8, 1u);
        uint32_t *group_total = compact && fused->starts_ready && slab_cap && c->h_pin_big && n_groups <= 4096
                                    ? c->ld_hist.as<uint32_t>() + fused->group_at : nullptr;
        // ---- dedupe + compaction (+ search pass 0) in ONE persistent kernel (collapse_lds.hip bucket_collapse12_kernel):
        // compact records behind a routed fused pack on one GPU, no spill list, buckets of the usual size, and at
        // most one array of segment hashes to write (its stride would be the unique count, which nobody knows yet).
        // The side path's keys lie at the head of the unique table and its probe lists feed pass 0: it must have
        // finished before the kernel starts.
        fqd::SegHashOut sho1;
        bool one_kernel = false;
        fqd::CollapseSync csync;
        if (compact && fused && !spill && group_total && fused->p0.mask != 0 && !fused->stamp_div && !d_w &&
            !c->one_kernel_off && (n >> B) <= 1000 && (n_buckets & 63u) == 0 && getenv("FQD_ONE_KERNEL_COLLAPSE")) {
            // (OFF unless asked for: measured at config 3 it takes 0.63-0.67 ms where the two kernels take 0.22 + 0.30 --
            // DESIGN "the one-kernel collapse")
            const bool want_hashes = c->seg_hint != 0;
            const uint32_t resident = fqd::collapse12_resident();
            uint32_t team = 16;                                       // workgroups that share one reservation of rows
            if (const char *e = getenv("FQD_ONE_KERNEL_TEAM"))
                team = (uint32_t)std::max(1, std::min(64, atoi(e)));
            if (resident >= team && (!want_hashes || c->seg_hint - 1u <= 1u)) {
                uint32_t G = std::min<uint32_t>(resident, n_buckets);
                if (const char *e = getenv("FQD_ONE_KERNEL_GRID"))     // (experiments / tests: fewer resident workgroups)
                    G = (uint32_t)std::max(1, std::min((int)G, atoi(e)));
                G -= G % team;
                const uint32_t TPR = G / team, R = (n_buckets + G - 1) / G;
                const size_t n_teams = (size_t)R * TPR;
                const size_t bytes = n_teams * 8 + ((n_teams * 4 + 7) & ~(size_t)7) + 16 + 16 * 8;
                HIP_TRY(c, c->ld_sync.reserve(bytes + 64));
                HIP_TRY(c, hipMemsetAsync(c->ld_sync.p, 0, bytes, c->st));
                csync.team = c->ld_sync.as<unsigned long long>();
                csync.base = reinterpret_cast<uint32_t *>(csync.team + n_teams);
                csync.abort = csync.base + ((n_teams + 1) & ~(size_t)1);
                csync.result = csync.abort + 1;
                csync.prof = reinterpret_cast<unsigned long long *>(csync.abort + 4);      // (16 words; only a -DFQD_FC_PROF build writes them)
                csync.team_size = team;
                csync.teams_per_round = TPR;
                csync.n_rounds = R;
                csync.wait_ticks = 20000000ull;                  // 0.2 s of the 100 MHz wall clock
                if (const char *e = getenv("FQD_ONE_KERNEL_WAIT_TICKS"))
                    csync.wait_ticks = strtoull(e, nullptr, 10);
                one_kernel = true;
            }
        }
        if (one_kernel) {
            if (side_pending) {
                HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join, 0));
                side_pending = false;
            }
            HIP_TRY(c, c->urecs.reserve(n * 16 + 16));
            HIP_TRY(c, c->ucounts.reserve(n * 4 + 16));
            HIP_TRY(c, c->ufirst.reserve(n * 8 + 64));
            if (c->seg_hint) {
                HIP_TRY(c, c->seg_hashes.reserve((size_t)c->seg_hint * n * 4 + 16));
                sho1.out = c->seg_hashes.as<uint32_t>();
                sho1.nseg = c->seg_hint;
                sho1.planes = sh.planes;
                sho1.kw = kw;
                sho1.len = sh.max_len;
                sho1.first = 1u;                     // (pass 0 happens in the kernel itself)
            }
            KTIME(c, FQD_K_DEDUPE12, fqd::launch_bucket_collapse12(
                      reinterpret_cast<const fqd::Rec12 *>(parted), c->ld_start.as<uint32_t>(), bucket_end, n_buckets, d_w,
                      compact, compact == 1 ? c->d_ctr32.as<uint32_t>() + C_SIDE : nullptr, c->urecs.as<uint32_t>(),
                      c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(), c->st, sho1, fused->p0, IdSource(), csync,
                      c->d_ctr32.as<uint32_t>() + C_BAD));
        } else if (compact && spill) {
            KTIME(c, FQD_K_DEDUPE12, fqd::launch_bucket_dedupe12_merge(
                      reinterpret_cast<const fqd::Rec12 *>(parted), c->ld_start.as<uint32_t>(), bucket_end, n_buckets, d_w,
                      c->ld_tmp_rec.as<uint32_t>(), c->ld_unique.as<uint32_t>(), c->d_ctr32.as<uint32_t>() + C_BAD, c->st,
                      group_total, side.recs, c->ld_side_table.as<uint32_t>(), fused->side_slots, fused->l1_over, B2));
            HIP_TRY(c, fqd::launch_side_finish(side.recs, c->ld_side_table.as<uint32_t>(), fused->side_slots,
                                               c->ld_side_table.as<uint32_t>() + 3 * (size_t)fused->side_slots,
                                               c->urecs.as<uint32_t>(), c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(),
                                               c->d_ctr32.as<uint32_t>() + C_SIDE, c->st, fused->p0, IdSource()));
        } else if (compact)
            KTIME(c, FQD_K_DEDUPE12, fqd::launch_bucket_dedupe12(reinterpret_cast<const fqd::Rec12 *>(parted),
                                                               c->ld_start.as<uint32_t>(), bucket_end, n_buckets, d_w,
                                                               c->ld_tmp_rec.as<uint32_t>(), c->ld_unique.as<uint32_t>(),
                                                               c->d_ctr32.as<uint32_t>() + C_BAD, c->st, group_total,
                                                               (n >> B) > 1000 /* (an owner's buckets at 5-8 ranks) */));
        else
        {
            // (exact bucket sizes: a key with very many copies is one huge bucket -- cut into chunks, see fqd::HugeBuckets)
            fqd::HugeBuckets huge;
            if (!bucket_end && !getenv("FQD_NO_HUGE_BUCKETS")) {
                const size_t nv = (size_t)FQD_HUGE_MAX * FQD_HUGE_CHUNKS;
                HIP_TRY(c, c->ld_huge.reserve(((size_t)n_buckets + 63) / 64 * 64 + (3 * nv + 4) * 4 + 64));
                huge.slot = c->ld_huge.as<uint8_t>();
                huge.vlo = reinterpret_cast<uint32_t *>(huge.slot + ((size_t)n_buckets + 63) / 64 * 64);
                huge.vhi = huge.vlo + nv;
                huge.vunique = huge.vhi + nv;
            }
            KTIME(c, FQD_K_DEDUPE, fqd::launch_bucket_dedupe(parted, c->ld_start.as<uint32_t>(), bucket_end, n_buckets, d_w,
                                                 c->ld_tmp_rec.as<uint32_t>(), c->ld_tmp_count.as<uint32_t>(),
                                                 c->ld_tmp_first.as<uint32_t>(), c->ld_unique.as<uint32_t>(),
                                                 c->d_ctr32.as<uint32_t>() + C_BAD, c->st, huge));
        }
        if (!group_total && !one_kernel)
            FQD_TRY(scan_u32(c, c->ld_unique.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets));
        if (side_pending) {                // (the read-back below takes the side path's count and flag too)
            HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join, 0));
            side_pending = false;
        }
        // One read-back: the unique count, the pack kernel's foreign-byte flag and every overflow flag
        // (bit 4: a level-1 slab of the fused pack, bit 2: a level-2 slab, bit 1: a bucket's LDS
        // table). The compaction is queued BEHIND the read-back and before the host waits for it: it
        // reads the unique count on the device and writes into tables sized for the worst case (as
        // many unique keys as reads), so the GPU works through the ~50 us the host needs to see the
        // numbers and react. If a flag says the attempt failed, what it wrote is simply not used.
        if (one_kernel)        // (result, abort: the two words behind the team words)
            FQD_TRY(queue_read_u32n(c, csync.abort, 2, 13));
        else if (group_total)       // (the unique count is the sum of the group totals: the host adds them up)
            HIP_TRY(c, hipMemcpyAsync(c->h_pin_big, group_total, (size_t)n_groups * 4, hipMemcpyDeviceToHost, c->st));
        else
            FQD_TRY(queue_read_u32(c, c->ld_unique_incl.as<uint32_t>() + (n_buckets - 1), 0));
        FQD_TRY(queue_read_u32n(c, c->d_ctr32.as<uint32_t>(), C_P0 + 1, 1));
        if (spill)
            FQD_TRY(queue_read_u32(c, side.spill_cursor, 12));
        FQD_TRY(queued_reads_mark(c));
        HIP_TRY(c, c->urecs.reserve(n * 16 + 16));
        HIP_TRY(c, c->ucounts.reserve(n * 4 + 16));
        HIP_TRY(c, c->ufirst.reserve(n * 8 + 64));
        // fqd_cluster[_keys] announced a Hamming search with nseg segments: the compaction writes its
        // segment hashes on the way (the records are fixed-length here)
        fqd::SegHashOut sho;
        const bool routed = fused && fused->p0.mask != 0;
        if (one_kernel) {
            sho = sho1;
        } else if (c->seg_hint && (routed || !getenv("FQD_NO_EARLY_SEG_HASHES"))) {
            HIP_TRY(c, c->seg_hashes.reserve((size_t)c->seg_hint * n * 4 + 16));
            sho.out = c->seg_hashes.as<uint32_t>();
            sho.nseg = c->seg_hint;
            sho.planes = sh.planes;
            sho.kw = kw;
            sho.len = sh.max_len;
            sho.first = routed ? 1u : 0u;        // (pass 0 happens in the compaction itself)
        }
        if (one_kernel)
            ;                  // (the rows are in the unique table already)
        else if (compact)
            KTIME(c, FQD_K_COMPACT, fqd::launch_bucket_compact12(
                      c->ld_start.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets,
                      c->ld_tmp_rec.as<uint32_t>(), compact, compact == 1 ? c->d_ctr32.as<uint32_t>() + C_SIDE : nullptr,
                      c->urecs.as<uint32_t>(), c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(), c->st, sho,
                      c->ld_unique.as<uint32_t>(), group_total, fused ? fused->p0 : fqd::Pass0(),
                      fused && fused->stamp_div ? d_ids : IdSource()));
        else
        KTIME(c, FQD_K_COMPACT, fqd::launch_bucket_compact(
                  c->ld_start.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets, c->ld_tmp_rec.as<uint32_t>(),
                  c->ld_tmp_count.as<uint32_t>(), c->ld_tmp_first.as<uint32_t>(), d_ids, c->urecs.as<uint32_t>(),
                  c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(), c->st, sho));
        early_nseg = sho.nseg;
        c->seg_hashes_first = sho.first;
        FQD_TRY(queued_reads_wait(c));
        uint32_t main_unique = 0;
        if (one_kernel) {
            if (taken_u32(c, 13) != 0) {
                // a wait inside the kernel ran into its limit (workgroups that were not resident: another process on
                // the GPU?): nothing it wrote counts -- once more, and from now on, with the two kernels
                c->one_kernel_off = true;
                if (getenv("FQD_DEBUG"))
                    fprintf(stderr, "[fqd] one-kernel collapse: a wait ran into its limit; two kernels from now on\n");
                return FQD_OK;             // (pack_collapse_fused sees one_kernel_off change and runs the attempt again)
            }
            main_unique = taken_u32(c, 14);
            c->route |= FQD_ROUTE_ONE_KERNEL_COLLAPSE;
            if (getenv("FQD_FC_PROF")) {       // (a -DFQD_FC_PROF build: thread 0's ticks per phase, summed over the workgroups)
                unsigned long long prof[16];
                HIP_TRY(c, hipMemcpyAsync(prof, csync.prof, sizeof prof, hipMemcpyDeviceToHost, c->st));
                HIP_TRY(c, stream_wait(c->st));
                const double per = 1.0 / (100.0 * csync.teams_per_round * csync.team_size);      // ticks of 10 ns -> microseconds per workgroup
                fprintf(stderr, "[fqd] one-kernel collapse, us per workgroup (%u rounds of %u workgroups):", csync.n_rounds,
                        csync.teams_per_round * csync.team_size);
                for (int i = 0; i < 11; i++)
                    fprintf(stderr, " p%d=%.1f", i, prof[i] * per);
                fprintf(stderr, "\n");
            }
        } else if (group_total)
            for (uint32_t g = 0; g < n_groups; g++)
                main_unique += static_cast<const uint32_t *>(c->h_pin_big)[g];
        else
            main_unique = taken_u32(c, 0);
        U32 = main_unique + (compact == 1 ? taken_u32(c, 1 + C_SIDE) : 0u);
        overflow = taken_u32(c, 1 + C_BAD);
        if (fused) {
            fused->pack_bad = taken_u32(c, 1 + C_PACKBAD);
            fused->spill_used = spill ? taken_u32(c, 12) : 0u;
            if (overflow && getenv("FQD_DEBUG"))
                fprintf(stderr, "[fqd] fused collapse: overflow flags 0x%x (1: a bucket's LDS table, 2: a level-2 slab, 4: a level-1 slab or the spill list, 16: the side path)\n", overflow);
            if (overflow & 16u)
                c->compact_off = true;
            if ((overflow & 6u) && !(overflow & 16u) && fused->can_spill && !spill && !c->heavy_keys) {
                // a full slab -- a key with hundreds of copies in a bucket, or with a share of all reads in a level-1
                // part: once more, and from then on, with the spill list
                c->heavy_keys = true;
            } else if ((overflow & 6u) && !(overflow & 16u) && fused->route_mask && !c->route_off) {
                // a ROUTED attempt without a spill list to turn to (two-plane alphabets): the slab may be full of keys
                // that share a segment-0 value, not of copies of one key -- whole-key hashing gets one attempt
                // (route_off below) before the fused path is given up for the context
            } else {
                if (overflow & 4u)
                    c->fused_off = true;
                if (overflow & 2u)
                    c->slab_off = true;
            }
            if (overflow && fused->route_mask)
                c->route_off = true;       // (keys crowding on one segment-0 value: whole-key hashing from now on)
            if (overflow || fused->pack_bad)
                return FQD_OK;
            // the compaction does search pass 0 -- unless a probe list was full (known now) or it meets a bucket too
            // large for that (known when the search reads C_P0 with its own counters: the compaction runs behind
            // this read-back)
            c->pass0_done = fused->p0.mask != 0 && taken_u32(c, 1 + C_P0) == 0 && sho.first == 1;
            c->pass0_nseg = c->pass0_done ? sho.nseg : 0;
            break;
        }
        if (slab_cap && (overflow & 2u)) {
            // a slab was too small (a key with hundreds of copies): once more with exact bucket sizes
            c->slab_off = true;
            slab_cap = 0;
            continue;
        }
        break;
    }
    if (overflow)
        return FQD_OK;  // some bucket held more distinct keys than the LDS table: sort-based path
    const uint64_t U = U32;
    HIP_TRY(c, c->ulens.reserve(U * 4 + 16));
    c->seg_hashes_nseg = U ? early_nseg : 0;
    unsigned long long counted = n;
    if (d_w) {
        FQD_TRY(zero_ctr64(c, C64_SUM));
        HIP_TRY(c, fqd::launch_sum_u32(d_w, n, c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
        FQD_TRY(read_ctr64(c, C64_SUM, &counted));
    }
    c->U = U;
    c->n_counted = counted;
    *done = true;
    return FQD_OK;
}

// Sort-free collapse for fixed-length records of any size (collapse_pairs.hip): (hash, position)
// pairs are partitioned into buckets, a workgroup per bucket matches them in an LDS table and
// verifies against the records where they lie. *done = false: not applicable, or a bucket / slab
// overflowed -- the caller takes the sort-based path.
int collapse_pairs(fqd_ctx *c, const uint32_t *d_w, IdSource d_ids, bool *done)
{
    *done = false;
    const uint64_t n = c->n;
    const KeyShape sh = c->ks;
    const char *force = getenv("FQD_COLLAPSE");  // "sort" / "lds" / "pairs": tests pin a path
    if (force && (!strcmp(force, "sort") || !strcmp(force, "lds")))
        return FQD_OK;
    if ((sh.stride & 3u) || sh.stride > 256 || n >= 0xFFFFFF00ull)
        return FQD_OK;
    if (n < 65536 && !(force && !strcmp(force, "pairs")))
        return FQD_OK;
    // ragged keys (trimmed reads, --check-lengths past a read's end): the lengths are compared with the records
    const uint32_t *d_lens = sh.ragged ? c->lens.as<uint32_t>() : nullptr;
    // (the records themselves hold the lengths: comparing records compares them, no gathers out of lens[])
    const bool len_in_rec = d_lens && c->recs_len_pad && c->recs_valid;
    FQD_TRY(ensure_hashes(c));
    const uint32_t B = lds_bucket_bits(n);
    const uint32_t n_buckets = 1u << B;
    uint32_t U32 = 0, overflow = 0;
    unsigned long long slab_over = 0;
    const uint32_t q_per_rec = sh.stride / 4;
    // fqd_cluster[_keys] announced a Hamming search with nseg segments: the compaction writes its
    // segment hashes on the way (records whose uint4 count divides 64: their lanes sit side by side)
    const bool want_seg = c->seg_hint && (!sh.ragged || d_lens) && q_per_rec <= 64 && !(q_per_rec & (q_per_rec - 1)) &&
                          !getenv("FQD_NO_EARLY_SEG_HASHES");
    uint32_t seg_written = 0;
    // the compaction into the unique table as it stands (rows 0 .. row_cap); seg_out: the segment hashes too
    auto compaction = [&](uint32_t n_buckets_, uint32_t row_cap, bool seg_out) -> int {
        fqd::SegHashOut sho;
        if (seg_out) {
            sho.out = c->seg_hashes.as<uint32_t>();
            sho.nseg = c->seg_hint;
            sho.planes = sh.planes;
            sho.kw = sh.planes * sh.words;
            sho.len = sh.max_len;
        }
        KTIME(c, FQD_K_COMPACT, fqd::launch_bucket_pairs_compact(
                  c->ld_start.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets_, c->ld_tmp_rec.as<uint32_t>(),
                  c->ld_tmp_count.as<uint32_t>(), c->ld_tmp_first.as<uint32_t>(), c->recs.as<uint32_t>(), sh.stride, d_ids,
                  c->urecs.as<uint32_t>(), c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(), c->st, sho, d_lens,
                  c->ulens.as<uint32_t>(), row_cap, len_in_rec ? c->modal_len_hint : 0u));
        seg_written = sho.nseg;
        return FQD_OK;
    };
    uint64_t queued_cap = 0;           // != 0: the compaction of the last attempt is queued already, for this many rows
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool slabs = attempt == 0 && !c->pairs_slab_off && !getenv("FQD_LDS_NO_SLABS");
        const uint32_t *items = nullptr, *bucket_end = nullptr;
        queued_cap = 0;
        FQD_TRY(zero_ctr32(c, C_BAD, 3));         // ... C_COLLISIONS, C_CHANGED: the slices' counters below
        FQD_TRY(zero_ctr64(c, C64_SLAB));
        FQD_TRY(fqd_api_partition_pairs(c, c->hashes.as<uint32_t>(), n, B, slabs, &items, &bucket_end, nullptr));
        // tmp rows of a bucket start where its items start (unique keys <= reads of the bucket)
        const uint64_t slots = c->gp_b.cap >= 16 && items == c->gp_b.as<uint32_t>() ? (c->gp_b.cap - 16) / 8 : n;
        HIP_TRY(c, c->ld_tmp_rec.reserve(slots * 4 + 16));        // here: the parked read's position
        HIP_TRY(c, c->ld_tmp_count.reserve(slots * 4 + 16));
        HIP_TRY(c, c->ld_tmp_first.reserve(slots * 4 + 16));
        HIP_TRY(c, c->ld_unique.reserve((size_t)n_buckets * 4 + 16));
        HIP_TRY(c, c->ld_unique_incl.reserve((size_t)n_buckets * 4 + 16));
        // a bucket of very many pairs (one key with 100 000 copies) is cut into slices, each a workgroup of its own
        fqd::PairsSlices sl;
        sl.slice = fqd::pairs_slice_items();
        if (sl.slice) {
            sl.cap = (uint32_t)(n / sl.slice + 1);
            HIP_TRY(c, c->pairs_slices.reserve((size_t)sl.cap * 24 + 64));
            sl.extra = c->pairs_slices.as<uint2>();
            sl.extra_unique = reinterpret_cast<uint32_t *>(sl.extra + sl.cap);
            sl.big = sl.extra_unique + sl.cap;
            sl.ctr = c->d_ctr32.as<uint32_t>() + C_COLLISIONS;       // (and C_CHANGED: neither has another user during a collapse)
            sl.tags = const_cast<uint32_t *>(items);
        }
        KTIME(c, FQD_K_DEDUPE, fqd::launch_bucket_pairs_dedupe(
                  items, c->ld_start.as<uint32_t>(), bucket_end, n_buckets, c->recs.as<uint32_t>(), sh.stride, d_w,
                  c->ld_tmp_rec.as<uint32_t>(), c->ld_tmp_count.as<uint32_t>(), c->ld_tmp_first.as<uint32_t>(),
                  c->ld_unique.as<uint32_t>(), c->d_ctr32.as<uint32_t>() + C_BAD, c->st, len_in_rec ? nullptr : d_lens, sl));
        FQD_TRY(scan_u32(c, c->ld_unique.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets));
        FQD_TRY(queue_read_u32(c, c->ld_unique_incl.as<uint32_t>() + (n_buckets - 1), 0));
        FQD_TRY(queue_read_u32(c, c->d_ctr32.as<uint32_t>() + C_BAD, 1));
        FQD_TRY(queue_read_ctr64(c, C64_SLAB, 1));
        FQD_TRY(queued_reads_mark(c));
        // The compaction goes out BEHIND the read-back without waiting for it, into the unique table the context's last
        // job left, when that has room for as many keys as that job had (the kernel itself writes nothing if this job has
        // more than fit): the GPU stood idle for ~50 us here -- three small copies, the host waking up, its launch --
        // at every job of a long-key workload. (FQD_NO_OPTIMISTIC_COMPACT=1: wait first, as before.)
        if (c->pairs_last_U && !getenv("FQD_NO_OPTIMISTIC_COMPACT")) {
            HIP_TRY(c, c->urecs.reserve(16));      // (a borrowed table is let go of)
            HIP_TRY(c, c->ulens.reserve(16));
            HIP_TRY(c, c->ucounts.reserve(16));
            HIP_TRY(c, c->ufirst.reserve(64));
            uint64_t room = std::min<uint64_t>({c->urecs.cap / ((size_t)sh.stride * 4), c->ucounts.cap / 4, c->ufirst.cap / 8,
                                                c->ulens.cap / 4});
            if (want_seg) {
                HIP_TRY(c, c->seg_hashes.reserve(16));
                room = std::min<uint64_t>(room, c->seg_hashes.cap / ((size_t)c->seg_hint * 4));
            }
            room = room > 16 ? room - 16 : 0;
            if (room >= c->pairs_last_U && room < 0xFFFFFFF0ull) {
                FQD_TRY(compaction(n_buckets, (uint32_t)room, want_seg));
                queued_cap = room;
            }
        }
        FQD_TRY(queued_reads_wait(c));
        taken_ctr64(c, &slab_over, 1);
        U32 = taken_u32(c, 0);
        overflow = taken_u32(c, 1);
        if (!slab_over)
            break;
        c->pairs_slab_off = true;      // a key with hundreds of copies: exact bucket sizes from now on
        if (attempt == 1)
            return FQD_OK;
    }
    if (overflow)
        return FQD_OK;                 // a bucket with more distinct keys than the LDS table holds
    const uint64_t U = U32;
    c->pairs_last_U = U;
    if (!(queued_cap && U <= queued_cap)) {
        HIP_TRY(c, c->urecs.reserve(U * sh.stride * 4 + 16));
        HIP_TRY(c, c->ulens.reserve(U * 4 + 16));
        HIP_TRY(c, c->ucounts.reserve(U * 4 + 16));
        HIP_TRY(c, c->ufirst.reserve(U * 8 + 64));
        if (want_seg && U)
            HIP_TRY(c, c->seg_hashes.reserve((size_t)c->seg_hint * U * 4 + 16));
        FQD_TRY(compaction(n_buckets, 0xFFFFFFFFu, want_seg && U));
    }
    c->seg_hashes_nseg = U ? seg_written : 0;
    unsigned long long counted = n;
    if (d_w) {
        FQD_TRY(zero_ctr64(c, C64_SUM));
        HIP_TRY(c, fqd::launch_sum_u32(d_w, n, c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
        FQD_TRY(read_ctr64(c, C64_SUM, &counted));
    }
    c->U = U;
    c->n_counted = counted;
    *done = true;
    return FQD_OK;
}

// Levenshtein neighbour search for the general case (edit.hip): index/probe records ->
// sort -> candidate pairs -> sort/unique -> banded-DP verification.

}  // namespace

int fqd_api_ensure_hashes(fqd_ctx *c) { return ensure_hashes(c); }

extern "C" {


int fqd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *fqd_global_error(void) { return g_global_error.c_str(); }

int fqd_create(int device, fqd_ctx **out)
{
    if (!out)
        return FQD_E_VALUE;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        g_global_error = "no HIP device visible: libfqdedup_hip needs an MI355X (gfx950); there is no CPU fallback";
        return FQD_E_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_global_error = "device ordinal out of range";
        return FQD_E_VALUE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        g_global_error = "hipGetDeviceProperties failed";
        return FQD_E_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_global_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return FQD_E_DEVICE;
    }
    fqd_ctx *c = new fqd_ctx();
    c->device = device;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&c->st) == hipSuccess &&
              [&] { for (hipEvent_t &e : c->tev) if (hipEventCreate(&e) != hipSuccess) return false; return true; }() &&
              [&] { for (hipEvent_t &e : c->kev) if (hipEventCreate(&e) != hipSuccess) return false; return true; }() &&
              c->d_ctr32.reserve(C_N32 * 4) == hipSuccess && c->d_ctr64.reserve(C64_N * 8) == hipSuccess &&
              c->d_lut.reserve(256) == hipSuccess &&
              c->d_stats.reserve(FQD_STAT_SLOTS * sizeof(fqd::PairStats)) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_rb, hipEventDisableTiming) == hipSuccess;
    // (the side stream at the highest priority: its four short kernels share the GPU with the dedupe kernel and,
    // queueing for CUs behind that kernel's 65 536 workgroups, would finish after it -- one of them was seen to
    // take 0.27 ms that way)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    ok = ok && hipStreamCreateWithPriority(&c->st_side, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    if (ok && hipHostMalloc(&c->h_pin, 256, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_pin = nullptr;            // read-backs then go through pageable memory
    }
    if (ok && hipHostMalloc(&c->h_pin_big, 1u << 20, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_pin_big = nullptr;
    }
    // event pairs around every kernel launch are a diagnostic: ~0.12 ms of a 2.4 ms job at config 3 (the stream
    // stops at every record), so they are off unless asked for
    if (const char *e = getenv("FQD_KERNEL_TIMERS"))
        c->ktime_mask = atoi(e) ? 0xFFFFFFFFu : 0u;
    if (!ok) {
        g_global_error = "could not create stream/events/buffers on the device";
        fqd_destroy(c);
        return FQD_E_DEVICE;
    }
    *out = c;
    return FQD_OK;
}

void fqd_destroy(fqd_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->st)
        (void)hipStreamSynchronize(c->st);
    DevBuf *bufs[] = {&c->d_lut, &c->d_ctr32, &c->d_ctr64, &c->d_present, &c->d_stats, &c->in_bytes, &c->in_offsets, &c->recs,
                      &c->lens, &c->hashes, &c->in_weights, &c->in_read_ids, &c->hs_sorted, &c->ids, &c->ids_sorted,
                      &c->flags, &c->run_idx, &c->run_start, &c->run_weight, &c->live_flag, &c->live_idx,
                      &c->collision_runs, &c->urecs, &c->ulens, &c->ucounts, &c->ufirst, &c->ld_hist, &c->ld_hist_incl, &c->ld_start,
                      &c->ld_cursor, &c->ld_part, &c->ld_part2, &c->ld_small, &c->ld_matrix, &c->ld_matrix_incl, &c->ld_tmp_rec, &c->ld_tmp_count, &c->ld_tmp_first, &c->ld_unique,
                      &c->ld_unique_incl, &c->seg_hashes,
                      &c->sorted_hash, &c->sorted_uid, &c->uid_iota, &c->edges, &c->sel_hash, &c->sel_uid, &c->q_table, &c->q_pass, &c->q_means, &c->q_bytes, &c->q_offsets,
                      &c->len_present, &c->ed_hash,
                      &c->ed_payload, &c->ed_hash_sorted, &c->ed_payload_sorted, &c->ed_cands, &c->ed_cands_sorted,
                      &c->d_alphabet, &c->labels, &c->best, &c->state,
                      &c->blocked, &c->kept, &c->kept_u32, &c->kept_scan, &c->kept_ids, &c->kept_ids_sorted, &c->tmp,
                      &c->stage_a, &c->stage_b, &c->stage_c, &c->stage_d, &c->hook_slots, &c->owners, &c->taint, &c->root_taint, &c->gp_a, &c->gp_b, &c->gp_small, &c->gp_cands, &c->seg_tab, &c->ld_seg, &c->kept_lists,
                      &c->t_idx, &c->t_order, &c->t_order_b, &c->t_rank, &c->t_keys, &c->t_keys_b, &c->t_lcp, &c->t_mark,
                      &c->t_mark_incl, &c->t_stats, &c->t_seed, &c->t_heads, &c->t_heads_incl, &c->t_member_uids,
                      &c->t_offsets, &c->store_alive, &c->st_recs, &c->st_lens, &c->st_counts, &c->st_first,
                      &c->st_comb_recs, &c->st_comb_lens, &c->st_comb_w, &c->st_comb_ids, &c->eg_tables, &c->eg_per_key,
                      &c->eg_per_key_incl, &c->ld_side, &c->ld_side_table};
    for (DevBuf *b : bufs)
        b->release();
    if (c->ev_rb)
        (void)hipEventDestroy(c->ev_rb);
    if (c->ev_fork)
        (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join)
        (void)hipEventDestroy(c->ev_join);
    if (c->st_side)
        (void)hipStreamDestroy(c->st_side);
    if (c->h_pin)
        (void)hipHostFree(c->h_pin);
    if (c->h_pin_big)
        (void)hipHostFree(c->h_pin_big);
    for (hipEvent_t e : c->tev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->kev)
        if (e) (void)hipEventDestroy(e);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

const char *fqd_last_error(const fqd_ctx *c) { return c ? c->err.c_str() : "null context"; }

int fqd_get_route(const fqd_ctx *c, uint32_t *route)
{
    if (!c || !route)
        return FQD_E_VALUE;
    *route = c->route;
    return FQD_OK;
}

int fqd_synchronize(fqd_ctx *c)
{
    FQD_TRY(bind(c));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

void *fqd_get_stream(fqd_ctx *c) { return c ? (void *)c->st : nullptr; }

int fqd_configure(fqd_ctx *c, const uint8_t *present128, uint32_t max_len, int ragged)
{
    if (!present128) {
        c->forced = false;
        return FQD_OK;
    }
    for (int b = 0; b < 128; b++)
        c->forced_present[b] = present128[b] ? 1 : 0;
    c->forced = true;
    c->forced_max_len = max_len;
    c->forced_ragged = ragged;
    // The geometry applies at once, so a context that only IMPORTS records (a routed search
    // pass, a cluster dissected away from its owner) needs no pack call first.
    FQD_TRY(bind(c));
    uint8_t lut[256];
    build_alphabet(c, c->forced_present, lut);
    FQD_TRY(set_geometry(c, max_len, ragged));
    FQD_TRY(upload_lut(c, lut));
    HIP_TRY(c, stream_wait(c->st));
    c->stage = ST_EMPTY;
    return FQD_OK;
}

int fqd_scan_keys(fqd_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len, int mem,
                  uint8_t *present128, uint32_t *max_len, int *ragged)
{
    FQD_TRY(bind(c));
    uint64_t n_bytes;
    if (offsets) {
        if (mem == FQD_HOST) {
            n_bytes = offsets[n];
        } else {
            HIP_TRY(c, hipMemcpyAsync(&n_bytes, offsets + n, 8, hipMemcpyDeviceToHost, c->st));
            HIP_TRY(c, stream_wait(c->st));
        }
    } else {
        n_bytes = n * (uint64_t)fixed_len;
    }
    const uint8_t *d_bytes;
    const uint64_t *d_off;
    FQD_TRY(to_device(c, bytes, (size_t)n_bytes, mem, c->in_bytes, &d_bytes));
    FQD_TRY(to_device(c, offsets, offsets ? (size_t)n + 1 : 0, mem, c->in_offsets, &d_off));
    return scan_keys_device(c, d_bytes, offsets ? d_off : nullptr, n, n_bytes, fixed_len, present128, max_len, ragged);
}

int fqd_get_shape(const fqd_ctx *c, fqd_shape *out)
{
    if (!c || !out)
        return FQD_E_VALUE;
    *out = c->shape;
    return FQD_OK;
}

int fqd_pack_keys(fqd_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len, int mem)
{
    FQD_TRY(bind(c));
    c->stage = ST_EMPTY;
    if (n >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "at most 2^32-16 keys per context");
    if (mem == FQD_DEVICE && ((uintptr_t)bytes & 15u))
        return fail(c, FQD_E_VALUE, "device key buffer must be 16-byte aligned");
    uint64_t n_bytes;
    if (offsets) {
        if (mem == FQD_HOST) {
            n_bytes = n ? offsets[n] : 0;
        } else {
            HIP_TRY(c, hipMemcpyAsync(&n_bytes, offsets + n, 8, hipMemcpyDeviceToHost, c->st));
            HIP_TRY(c, stream_wait(c->st));
        }
    } else {
        n_bytes = n * (uint64_t)fixed_len;
    }
    StageTimer timer(c, FQD_T_PACK);
    const uint8_t *d_bytes;
    const uint64_t *d_off = nullptr;
    FQD_TRY(to_device(c, bytes, (size_t)n_bytes, mem, c->in_bytes, &d_bytes));
    if (offsets)
        FQD_TRY(to_device(c, offsets, (size_t)n + 1, mem, c->in_offsets, &d_off));

    uint8_t present[128], lut[256];
    uint32_t max_len = fixed_len;
    int ragged = 0;
    // Auto mode is optimistic: pack with the DNA alphabet "ACGNT" first (no pass over the
    // bytes just to learn the alphabet); only if the pack kernel meets a byte outside it
    // are the bytes scanned and the keys packed again with the exact alphabet.
    bool optimistic = !c->forced;
    if (c->forced) {
        memcpy(present, c->forced_present, 128);
        max_len = c->forced_max_len;
        ragged = c->forced_ragged;
        c->modal_len_hint = offsets ? max_len : fixed_len;
        if (!offsets && fixed_len != max_len)
            ragged = 1;
    } else {
        memset(present, 0, sizeof present);
        for (const char *p = "ACGNT"; *p; p++)
            present[(int)*p] = 1;
        if (offsets && n) {
            uint32_t mm[2] = {0xFFFFFFFFu, 0u};
            HIP_TRY(c, hipMemcpyAsync(c->d_ctr32.as<uint32_t>() + C_MINLEN, mm, 8, hipMemcpyHostToDevice, c->st));
            HIP_TRY(c, fqd::launch_scan_lens(d_off, n, c->d_ctr32.as<uint32_t>() + C_MINLEN, c->st));
            HIP_TRY(c, hipMemcpyAsync(mm, c->d_ctr32.as<uint32_t>() + C_MINLEN, 8, hipMemcpyDeviceToHost, c->st));
            HIP_TRY(c, stream_wait(c->st));
            max_len = mm[1];
            ragged = mm[0] != mm[1];
            c->modal_len_hint = (uint32_t)(((uint64_t)mm[0] + mm[1] + 1) / 2);
        } else if (!n) {
            max_len = 0;
        }
    }
    for (int attempt = 0;; attempt++) {
        build_alphabet(c, present, lut);
        FQD_TRY(set_geometry(c, max_len, ragged));
        FQD_TRY(upload_lut(c, lut));
        const KeyShape sh = c->ks;
        HIP_TRY(c, c->recs.reserve((size_t)n * sh.stride * 4 + 16));
        HIP_TRY(c, c->hashes.reserve((size_t)n * 4 + 16));
        if (sh.ragged)
            HIP_TRY(c, c->lens.reserve((size_t)n * 4 + 16));
        if (c->owner_rule.parts)
            HIP_TRY(c, c->owners.reserve((size_t)n * 4 + 16));
        c->owners_done = fqd::OwnerRule{};
        FQD_TRY(zero_ctr32(c, C_BAD));
        // ragged records with a padding word carry their key's length in the last one (not for the store, whose rows are
        // of both kinds; not with an owner rule: the sender's read index rides there)
        c->recs_len_pad = sh.ragged && sh.stride > sh.planes * sh.words && !c->owner_rule.parts && !c->no_len_pad &&
                          c->modal_len_hint && !getenv("FQD_NO_LEN_IN_RECORD");
        KeyShape shp = sh;
        if (c->recs_len_pad)
            shp.ragged |= 2u;
        StageTimer kernel_timer(c, FQD_T_PACK_KERNEL);
        KTIME(c, FQD_K_PACK, fqd::launch_pack(d_bytes, n_bytes, d_off, n, fixed_len, shp, c->d_lut.as<uint8_t>(), lut,
                                    c->recs.as<uint32_t>(), sh.ragged ? c->lens.as<uint32_t>() : nullptr,
                                    c->hashes.as<uint32_t>(), c->owner_rule.parts ? c->owners.as<uint32_t>() : nullptr,
                                    c->owner_rule, c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
        kernel_timer.stop();
        uint32_t bad = 0;
        FQD_TRY(read_ctr32(c, C_BAD, &bad));
        if (getenv("FQD_DEBUG"))
            fprintf(stderr, "[fqd] pack attempt %d: n=%llu len=%u K=%u W=%u stride=%u alphabet=%.*s bad=%u\n",
                    attempt, (unsigned long long)n, max_len, sh.planes, sh.words, sh.stride, (int)c->shape.alphabet_size,
                    (const char *)c->shape.alphabet, bad);
        if (!bad)
            break;
        if (!optimistic || attempt > 0) {
            timer.stop();
            return fail(c, FQD_E_VALUE,
                        c->forced ? "a key holds a byte outside the configured alphabet"
                                  : "Sequence must consist only of ASCII characters");
        }
        // a byte outside "ACGNT": learn the real alphabet (and refuse non-ASCII there)
        FQD_TRY(scan_keys_device(c, d_bytes, d_off, n, n_bytes, fixed_len, present, &max_len, &ragged));
    }
    timer.stop();
    c->n = n;
    c->owners_done = c->owner_rule;
    c->hashes_valid = true;
    c->recs_valid = true;
    c->stage = ST_PACKED;
    return FQD_OK;
}

int fqd_set_owner_rule(fqd_ctx *c, uint32_t n_parts, uint32_t n_segments, uint32_t segment)
{
    if (n_parts > 65536 || (n_parts && (n_segments == 0 || segment >= n_segments)))
        return fail(c, FQD_E_VALUE, "bad owner rule");
    c->owner_rule.parts = n_parts;
    c->owner_rule.nseg = n_parts ? n_segments : 1;
    c->owner_rule.seg = n_parts ? segment : 0;
    return FQD_OK;
}

static void set_id_range(fqd_ctx *c, uint64_t limit)
{
    c->id_limit = limit;
    c->id_bits = 64;
    if (limit != ~0ull) {
        c->id_bits = 1;
        while (c->id_bits < 64 && ((limit ? limit - 1 : 0) >> c->id_bits))
            c->id_bits++;
    }
}

// Shared body of fqd_collapse / fqd_collapse_received. `ids` says where a read's id comes from
// (device pointers); id_limit bounds every id (~0: unknown).
static int collapse_impl(fqd_ctx *c, const uint32_t *weights, int mem, IdSource ids, uint64_t id_limit,
                         uint64_t *n_unique)
{
    const bool read_ids = ids.ids64 || ids.stamped;   // false: a read's id is its position
    c->stage = ST_PACKED;
    const uint64_t n = c->n;
    const KeyShape sh = c->ks;
    StageTimer timer(c, FQD_T_COLLAPSE);
    c->U = 0;
    c->n_counted = 0;
    if (n == 0) {
        timer.stop();
        c->stage = ST_UNIQUE;
        if (n_unique)
            *n_unique = 0;
        return FQD_OK;
    }
    const uint32_t *d_w;
    FQD_TRY(to_device(c, weights, (size_t)n, mem, c->in_weights, &d_w));

    c->urecs_len_pad = false;
    bool lds_done = false;
    FQD_TRY(collapse_lds(c, weights ? d_w : nullptr, ids, &lds_done));
    if (lds_done) {
        timer.stop();
        c->route |= FQD_ROUTE_COLLAPSE_LDS;
        c->collapse_path = 1;
        c->urecs_len_pad = false;
        c->collapsed = true;
        c->first_distinct = true;
        set_id_range(c, read_ids ? id_limit : n);
        c->stage = ST_UNIQUE;
        if (n_unique)
            *n_unique = c->U;
        return FQD_OK;
    }
    if (ids.stamped) {
        // The sort-based path compares whole records, padding included: take the ids out of the
        // padding word (explicit array from here on) and clear it. (The packed buffer may be one the
        // caller lent with FQD_DEVICE_BORROW: its padding words are cleared in place.)
        HIP_TRY(c, c->in_read_ids.reserve((size_t)n * 8 + 16));
        HIP_TRY(c, fqd::launch_extract_ids(ids, c->recs.as<uint32_t>(), n, c->in_read_ids.as<uint64_t>(), c->st));
        IdSource plain;
        plain.ids64 = c->in_read_ids.as<uint64_t>();
        ids = plain;
    }
    bool pairs_done = false;
    FQD_TRY(collapse_pairs(c, weights ? d_w : nullptr, ids, &pairs_done));
    if (pairs_done) {
        timer.stop();
        c->urecs_len_pad = c->ks.ragged && c->recs_len_pad && c->recs_valid && c->modal_len_hint < (1u << 24);
        c->route |= FQD_ROUTE_COLLAPSE_PAIRS;
        c->collapse_path = 3;
        c->collapsed = true;
        c->first_distinct = true;
        set_id_range(c, read_ids ? id_limit : n);
        c->stage = ST_UNIQUE;
        if (n_unique)
            *n_unique = c->U;
        return FQD_OK;
    }
    c->collapse_path = 2;
    c->route |= FQD_ROUTE_COLLAPSE_SORT;
    const int bits = hash_bits_from_env();
    const uint32_t mask = bits >= 32 ? ~0u : ((1u << bits) - 1u);
    HIP_TRY(c, c->hs_sorted.reserve(n * 4 + 16));
    HIP_TRY(c, c->ids.reserve(n * 4 + 16));
    HIP_TRY(c, c->ids_sorted.reserve(n * 4 + 16));
    HIP_TRY(c, c->flags.reserve(n * 4 + 16));
    HIP_TRY(c, c->run_idx.reserve(n * 4 + 16));
    HIP_TRY(c, fqd::launch_iota_u32(c->ids.as<uint32_t>(), n, c->st));
    FQD_TRY(ensure_hashes(c));
    FQD_TRY(sort_u32_pairs(c, c->hashes.as<uint32_t>(), c->hs_sorted.as<uint32_t>(), c->ids.as<uint32_t>(),
                           c->ids_sorted.as<uint32_t>(), n, bits));

    uint32_t cap = (uint32_t)std::max<size_t>(1024, c->collision_runs.cap / 4);
    for (;;) {
        HIP_TRY(c, c->collision_runs.reserve((size_t)cap * 4));
        FQD_TRY(zero_ctr32(c, C_COLLISIONS));
        KTIME(c, FQD_K_HEAD_FLAGS, fqd::launch_head_flags(c->hs_sorted.as<uint32_t>(), c->ids_sorted.as<uint32_t>(),
                                          c->recs.as<uint32_t>(), c->lens.as<uint32_t>(), n, sh, mask,
                                          c->flags.as<uint32_t>(), c->d_ctr32.as<uint32_t>() + C_COLLISIONS,
                                          c->collision_runs.as<uint32_t>(), cap, c->st));
        uint32_t n_coll = 0;
        FQD_TRY(read_ctr32(c, C_COLLISIONS, &n_coll));
        if (n_coll > cap) {
            cap = n_coll + 1024;
            continue;
        }
        HIP_TRY(c, fqd::launch_fix_collision_runs(c->hs_sorted.as<uint32_t>(), c->ids_sorted.as<uint32_t>(),
                                                  c->recs.as<uint32_t>(), c->lens.as<uint32_t>(), n, sh, mask,
                                                  c->flags.as<uint32_t>(), c->collision_runs.as<uint32_t>(), n_coll,
                                                  c->st));
        break;
    }
    FQD_TRY(scan_u32(c, c->flags.as<uint32_t>(), c->run_idx.as<uint32_t>(), n));
    uint32_t n_runs = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_runs, c->run_idx.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));

    HIP_TRY(c, c->run_start.reserve(((size_t)n_runs + 1) * 4 + 16));
    HIP_TRY(c, c->run_weight.reserve((size_t)n_runs * 4 + 16));
    HIP_TRY(c, c->live_flag.reserve((size_t)n_runs * 4 + 16));
    HIP_TRY(c, c->live_idx.reserve((size_t)n_runs * 4 + 16));
    HIP_TRY(c, fqd::launch_run_starts(c->flags.as<uint32_t>(), c->run_idx.as<uint32_t>(), n,
                                      c->run_start.as<uint32_t>(), c->st));
    HIP_TRY(c, fqd::launch_run_weights(c->run_start.as<uint32_t>(), n_runs, n, c->ids_sorted.as<uint32_t>(),
                                       weights ? d_w : nullptr, c->run_weight.as<uint32_t>(),
                                       c->live_flag.as<uint32_t>(), c->st));
    FQD_TRY(scan_u32(c, c->live_flag.as<uint32_t>(), c->live_idx.as<uint32_t>(), n_runs));
    uint32_t U32 = 0;
    HIP_TRY(c, hipMemcpyAsync(&U32, c->live_idx.as<uint32_t>() + (n_runs - 1), 4, hipMemcpyDeviceToHost, c->st));
    FQD_TRY(zero_ctr64(c, C64_SUM));
    HIP_TRY(c, fqd::launch_sum_u32(c->run_weight.as<uint32_t>(), n_runs, c->d_ctr64.as<unsigned long long>() + C64_SUM,
                                   c->st));
    unsigned long long counted = 0;
    FQD_TRY(read_ctr64(c, C64_SUM, &counted));
    const uint64_t U = U32;
    HIP_TRY(c, c->urecs.reserve(U * sh.stride * 4 + 16));
    HIP_TRY(c, c->ulens.reserve(U * 4 + 16));
    HIP_TRY(c, c->ucounts.reserve(U * 4 + 16));
    HIP_TRY(c, c->ufirst.reserve(U * 8 + 64));
    KTIME(c, FQD_K_WRITE_UNIQUE, fqd::launch_write_unique(c->run_start.as<uint32_t>(), c->run_weight.as<uint32_t>(),
                                        c->live_flag.as<uint32_t>(), c->live_idx.as<uint32_t>(), n_runs,
                                        c->ids_sorted.as<uint32_t>(), c->recs.as<uint32_t>(), c->lens.as<uint32_t>(),
                                        ids.ids64, sh, c->urecs.as<uint32_t>(),
                                        c->ulens.as<uint32_t>(), c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(),
                                        c->st));
    timer.stop();
    c->U = U;
    c->n_counted = counted;
    c->urecs_len_pad = false;
    c->collapsed = true;
    c->first_distinct = true;
    set_id_range(c, read_ids ? id_limit : n);
    c->stage = ST_UNIQUE;
    if (n_unique)
        *n_unique = U;
    return FQD_OK;
}

// Pack + collapse for fqd_cluster_keys with the pack kernel writing straight into level 1 of the
// LDS collapse (pack.hip FUSED): the packed reads never exist in read order, which saves one write
// and one read of all records, the hash array and the level-1 histogram pass (1.0 -> 0.55 ms of a
// 3.9 ms job at 50 M reads). *done = false: not applicable, or some slab / table overflowed or the
// keys hold a byte outside "ACGNT" -- the caller then goes the plain way, which handles all that.
static int pack_collapse_fused_once(fqd_ctx *c, const uint8_t *bytes, uint64_t n, uint32_t fixed_len, int mem,
                                    const uint32_t *weights, int aux_mem, bool *done);

// Compact records first (12 bytes per read through the partition, keys with an N on a side path); when their side
// slabs overflow -- the data has many keys with an N -- once more with uint4 records, and so from then on.
static int pack_collapse_fused(fqd_ctx *c, const uint8_t *bytes, uint64_t n, uint32_t fixed_len, int mem,
                               const uint32_t *weights, int aux_mem, bool *done)
{
    // ... and when a slab is full -- keys with very many copies -- once more with the spill list (c->heavy_keys: the
    // routed collapse is off then), and so from then on.
    // A context that gave up a fast path (heavy_keys, route_off) tries it again after fast_retry_after jobs that did
    // not need the slow one -- a file with PCR jackpots or a poly-A library should not cost every later file of a
    // long-lived context 0.3-0.6 ms per step; a retry that fails doubles the wait (8, 16, ... 1024 jobs).
    const bool gave_up_before = c->heavy_keys || c->route_off;
    if (gave_up_before && c->clean_jobs >= c->fast_retry_after && !getenv("FQD_NO_FAST_PATH_RETRY")) {
        c->heavy_keys = c->route_off = false;
        c->gp_slab_off = false;        // (the search's slabs and the one-launch union-find with it: each finds out
        c->uf_sampled = false;         //  again within the job if the data still needs the careful way)
        c->clean_jobs = 0;
        c->fast_probe = true;
    }
    const bool heavy_at_start = c->heavy_keys, route_off_at_start = c->route_off;
    c->last_spill_used = 0;
    for (int attempt = 0; attempt < 4; attempt++) {
        const bool was_off = c->compact_off, was_heavy = c->heavy_keys, was_routed = !c->route_off, was_one = !c->one_kernel_off;
        c->route &= ~FQD_ROUTE_RESTARTED;          // (raised by an attempt that ended early; the last one counts)
        FQD_TRY(pack_collapse_fused_once(c, bytes, n, fixed_len, mem, weights, aux_mem, done));
        if (*done || c->fused_off ||
            (was_off == c->compact_off && was_heavy == c->heavy_keys && was_routed == !c->route_off &&
             was_one == !c->one_kernel_off))
            break;
    }
    if ((c->heavy_keys && !heavy_at_start) || (c->route_off && !route_off_at_start)) {
        // the data of this job needed the slow way (again)
        if (c->fast_probe)
            c->fast_retry_after = std::min<uint32_t>(c->fast_retry_after * 2, 1024u);
        c->clean_jobs = 0;
    } else if (c->heavy_keys || c->route_off) {
        // (with the spill list: clean when nothing was spilled; route_off alone: the collapse cannot tell, every job counts)
        c->clean_jobs = (c->heavy_keys && c->last_spill_used) ? 0u : c->clean_jobs + 1u;
    }
    c->fast_probe = false;
    return FQD_OK;
}

static int pack_collapse_fused_once(fqd_ctx *c, const uint8_t *bytes, uint64_t n, uint32_t fixed_len, int mem,
                                    const uint32_t *weights, int aux_mem, bool *done)
{
    *done = false;
    const char *force = getenv("FQD_COLLAPSE");
    if (c->fused_off || getenv("FQD_NO_FUSED_PACK") || c->owner_rule.parts || (force && !strcmp(force, "sort")))
        return FQD_OK;
    uint64_t min_reads = 8ull << 20;   // below that the 8192 level-1 parts are too small to be worth it
    if (const char *e = getenv("FQD_FUSED_MIN_READS"))
        min_reads = strtoull(e, nullptr, 10);
    if (n < min_reads || n < 32768 || n >= 0xFFFFFF00ull || !fixed_len)
        return FQD_OK;
    if (mem == FQD_DEVICE && ((uintptr_t)bytes & 15u))
        return FQD_OK;                 // fqd_pack_keys reports it
    uint8_t present[128], lut[256];
    if (c->forced) {
        if (c->forced_ragged || c->forced_max_len != fixed_len)
            return FQD_OK;
        memcpy(present, c->forced_present, 128);
    } else {
        memset(present, 0, sizeof present);
        for (const char *p = "ACGNT"; *p; p++)
            present[(int)*p] = 1;
    }
    c->stage = ST_EMPTY;
    build_alphabet(c, present, lut);
    FQD_TRY(set_geometry(c, fixed_len, 0));
    const KeyShape sh = c->ks;
    if (sh.ragged || sh.stride != 4 || sh.planes * sh.words > 3)
        return FQD_OK;
    const uint32_t B = lds_bucket_bits(n);
    if (B <= 8)
        return FQD_OK;
    // 256 level-1 bins x 32 sub-parts: tile t adds to sub-part t % 32 of its bins. (Measured at 50 M
    // reads: up to 16 sub-parts per bin the tiles queue on the part cursors and the kernel takes
    // 1.0 ms; from 32 on, 0.55-0.6 ms.)
    uint32_t l1_bits = 8, sub_bits = 5;
    if (const char *e = getenv("FQD_FUSED_L1_BITS"))       // (experiments: 128 bins make the pack faster and level 2 slower)
        l1_bits = (uint32_t)std::max(4, std::min(8, atoi(e)));
    if (const char *e = getenv("FQD_FUSED_SUB_BITS"))
        sub_bits = (uint32_t)std::max(0, std::min(6, atoi(e)));
    if (B > l1_bits + 10)                                  // level 2 has at most 1024 bins
        l1_bits = B - 10;
    const uint32_t parts = (1u << l1_bits) << sub_bits;
    const uint32_t cap1 = (uint32_t)(((n / parts) * 5 / 4 + 256 + 3) & ~3ull);
    if ((uint64_t)parts * cap1 + n >= 0xFFFFFF00ull)
        return FQD_OK;
    FQD_TRY(upload_lut(c, lut));
    // compact records (fqd_internal.h Rec12): one 32-base word per plane, and either two planes or
    // the three of exactly "ACGNT" (a key with an N -- code 3 -- then takes the side path)
    uint32_t compact = 0;
    if (!c->compact_off && !getenv("FQD_NO_COMPACT_RECORDS") && sh.words == 1) {
        if (sh.planes == 2) {
            compact = 2;
        } else if (sh.planes == 3) {
            uint32_t others = present[0] ? 1u : 0u;
            for (int b = 1; b < 128; b++)
                others += present[b] && !strchr("ACGNT", b) ? 1u : 0u;
            if (!others && present['A'] && present['C'] && present['G'] && present['N'] && present['T'])
                compact = 1;
        }
    }
    // side path of the compact records: 256 slabs for n / 64 + 16 Ki keys with an N in all (more: uint4 records
    // from then on), a hash table of twice as many slots
    uint32_t side_slabs = 0, side_cap = 0, side_slots = 0;
    // ... and, in a context that has met a full slab before (c->heavy_keys), the spill list behind the side slabs:
    // room for an eighth of the reads, collapsed through the same table (twice its distinct keys at most: the table
    // is sized for a sixteenth of the reads on top)
    const bool can_spill = compact == 1 && !getenv("FQD_NO_SPILL_LIST");
    const bool heavy = can_spill && c->heavy_keys;
    uint32_t spill_cap = 0;
    if (compact == 1) {
        side_slabs = 256;
        side_cap = (uint32_t)((((n >> 6) + 16384) / side_slabs + 3) & ~3ull);
        if (heavy)
            spill_cap = (uint32_t)(((n >> 3) + 65536 + 3) & ~3ull);
        side_slots = 1024;
        while (side_slots < 2ull * side_slabs * side_cap + (heavy ? (n >> 3) : 0ull))
            side_slots *= 2;
        if ((uint64_t)side_slabs * side_cap + spill_cap >= 0xFFFFFF00ull)
            return FQD_OK;
    }
    const uint64_t n_bytes = n * (uint64_t)fixed_len;
    const uint8_t *d_bytes;
    StageTimer pack_timer(c, FQD_T_PACK);
    // Keys in HOST memory, a large job: the bytes come over in pieces on the second stream and the pack kernel of
    // piece k runs under the copy of piece k + 1 (it is the one part of the step that needs nothing but the bytes:
    // 0.5 ms of a 32 ms PCIe-inclusive step at config 3).
    uint32_t pieces = 1;
    if (mem == FQD_HOST && n_bytes >= (256ull << 20) && n >= (1u << 20) && !getenv("FQD_NO_CHUNKED_UPLOAD"))
        pieces = 8;
    if (const char *e = getenv("FQD_UPLOAD_PIECES"))
        pieces = mem == FQD_HOST ? (uint32_t)std::max(1, std::min(64, atoi(e))) : 1u;
    if (pieces > 1) {
        HIP_TRY(c, c->in_bytes.reserve((size_t)n_bytes + 16));
        d_bytes = c->in_bytes.as<uint8_t>();
    } else {
        FQD_TRY(to_device(c, bytes, (size_t)n_bytes, mem, c->in_bytes, &d_bytes));
    }
    HIP_TRY(c, c->ld_part.reserve((size_t)parts * cap1 * 16 + 16));
    HIP_TRY(c, c->ld_seg.reserve((size_t)3 * (parts + 4) * 4));
    uint32_t *seg_start = c->ld_seg.as<uint32_t>(), *cursor = seg_start + (parts + 4);
    FQD_TRY(zero_ctr32(c, 0, C_N32));
    if (compact == 1) {
        HIP_TRY(c, c->ld_side.reserve(((size_t)side_slabs * side_cap + spill_cap) * 16 + 16));
        // (behind the table: the side cursors, the side starts, 4 spare words | the spill cursor, 3 spare words, 256 level-1 marks)
        HIP_TRY(c, c->ld_side_table.reserve(((size_t)fqd::side_table_words(side_slots) + 2 * side_slabs + 8 + 256) * 4 + 16));
        // (the side path writes the head of the unique table before the compaction is queued)
        HIP_TRY(c, c->urecs.reserve(n * 16 + 16));
        HIP_TRY(c, c->ucounts.reserve(n * 4 + 16));
        HIP_TRY(c, c->ufirst.reserve(n * 8 + 64));
    }
    // The routed collapse: a Hamming search with nseg = d + 1 segments follows (c->seg_hint) -- the reads are binned by
    // a hash of segment 0 alone, so that every bucket of the collapse holds ALL keys sharing its segment-0 values and
    // the compaction does search pass 0 on the spot (fqd::Pass0): half of the (hash, uid) items never exist.
    const uint32_t n_buckets = 1u << B;
    const uint32_t n_groups = std::max(n_buckets >> 8, 1u);
    uint32_t route_mask = 0;
    fqd::Pass0 p0;
    if (compact && c->seg_hint >= 2 && !c->route_off && !heavy && !getenv("FQD_NO_ROUTED_COLLAPSE") && sh.words == 1) {
        const uint32_t seg0_len = fixed_len / c->seg_hint;         // fqd_segment(len, 0, nseg): [0, len / nseg)
        if (seg0_len >= 8) {                                       // (a shorter segment 0 has too few values to spread the reads)
            route_mask = seg0_len >= 32 ? 0xFFFFFFFFu : ((1u << seg0_len) - 1u);
            HIP_TRY(c, c->p0_probe.reserve((size_t)n_buckets * fqd::FQD_P0_PROBE_CAP * 4 + 16));
            const uint64_t want_edges = std::max<uint64_t>(c->edges.cap / 8, std::max<uint64_t>(1u << 20, n / 8));
            if (c->edge_cap < want_edges || !c->edges.p) {
                HIP_TRY(c, c->edges.reserve(want_edges * 8));
                c->edge_cap = c->edges.cap / 8;
            }
            p0.mask = route_mask;
            p0.d = c->seg_hint - 1;
            p0.bucket_bits = B;
            p0.max_rows = fqd::pass0_max_rows();
            if (const char *e = getenv("FQD_P0_MAX_ROWS"))         // tests: buckets "too large" for pass 0
                p0.max_rows = (uint32_t)std::max(1, std::min((int)fqd::pass0_max_rows(), atoi(e)));
            p0.probe = c->p0_probe.as<uint32_t>();
            p0.edges = c->edges.as<uint32_t>();
            p0.edge_count = c->d_ctr64.as<unsigned long long>() + C64_EDGES;
            p0.edge_cap = c->edge_cap;
            if (const char *e = getenv("FQD_P0_EDGE_CAP"))         // tests: an edge list too short for pass 0's pairs
                p0.edge_cap = c->edge_cap = std::min<uint64_t>(c->edge_cap, strtoull(e, nullptr, 10));
            c->pass0_edge_cap = p0.edge_cap;
            p0.flag = c->d_ctr32.as<uint32_t>() + C_P0;
            p0.stats = c->d_stats.as<fqd::PairStats>();
        }
    }
    // the slab starts of the pack kernel's parts and, ahead of the pack, of level 2 (the geometry collapse_lds will
    // compute) and of the side path: one launch
    bool starts_ready = false;
    uint32_t group_at = 0;
    {
        const uint32_t slab_cap2 = (uint32_t)(((n >> B) * 3 / 2 + 64 + 3) & ~3ull);
        uint32_t *tail = compact == 1 ? c->ld_side_table.as<uint32_t>() + fqd::side_table_words(side_slots) : nullptr;
        if ((uint64_t)slab_cap2 * n_buckets + n < 0xFFFFFF00ull) {
            HIP_TRY(c, c->ld_start.reserve(((size_t)n_buckets + 1) * 4 + 16));
            HIP_TRY(c, c->ld_cursor.reserve((size_t)n_buckets * 4 + 1024 * 4));
            // ... and the zeros of the dedupe's group totals and of the probe counts of pass 0 (c->ld_hist, free in slab
            // mode: [0, n_buckets) probe counts, behind them the group totals), of the edge counter and the statistics
            HIP_TRY(c, c->ld_hist.reserve(((size_t)n_buckets + n_groups) * 4 + 1024 * 4));
            group_at = p0.mask ? n_buckets : 0u;
            p0.probe_n = p0.mask ? c->ld_hist.as<uint32_t>() : nullptr;
            HIP_TRY(c, fqd::launch_slab_starts3(parts, cap1, seg_start, cursor, n_buckets, slab_cap2,
                                                c->ld_start.as<uint32_t>(), c->ld_cursor.as<uint32_t>(), side_slabs, side_cap,
                                                tail ? tail + side_slabs : nullptr, tail, c->st,
                                                compact ? group_at + n_groups : 0u,
                                                compact ? c->ld_hist.as<uint32_t>() : nullptr,
                                                p0.mask ? reinterpret_cast<uint32_t *>(p0.edge_count) : nullptr, 1u,
                                                p0.mask ? reinterpret_cast<uint32_t *>(p0.stats) : nullptr,
                                                (uint32_t)(FQD_STAT_SLOTS * sizeof(fqd::PairStats) / 4) - 1u));
            starts_ready = true;
        } else {
            HIP_TRY(c, fqd::launch_slab_starts(parts, cap1, seg_start, cursor, c->st));
            p0 = fqd::Pass0();
            route_mask = 0;
        }
    }
    fqd::PackScatter fs{cursor, reinterpret_cast<uint4 *>(c->ld_part.p), c->d_ctr32.as<uint32_t>() + C_BAD,
                        32 - l1_bits, 1u << l1_bits, 1u << sub_bits, cap1, 0u, 0u, 0u, 0u, route_mask};
    // compact records, one GPU: the pack kernel takes the keys with an N out itself (FQD_NO_PACK_PARKING=1: level 2 does)
    const bool side_parked = compact == 1 && starts_ready && !getenv("FQD_NO_PACK_PARKING");
    if (side_parked) {
        uint32_t *tail = c->ld_side_table.as<uint32_t>() + fqd::side_table_words(side_slots);
        fs.side_recs = c->ld_side.as<uint4>();
        fs.side_cursor = tail;
        fs.side_slabs = side_slabs;
        fs.side_cap = side_cap;
    }
    uint32_t *spill_words = nullptr;
    if (heavy) {
        spill_words = c->ld_side_table.as<uint32_t>() + fqd::side_table_words(side_slots) + 2 * side_slabs + 4;
        HIP_TRY(c, hipMemsetAsync(spill_words, 0, (4 + 256) * 4, c->st));
        fs.spill = c->ld_side.as<uint4>() + (size_t)side_slabs * side_cap;
        fs.spill_cursor = spill_words;
        fs.spill_cap = spill_cap;
        fs.l1_over = spill_words + 4;
    }
    if (pieces > 1) {
        // piece boundaries on multiples of 8192 reads: whole workgroups of the pack kernel, 16-byte aligned bytes
        const uint64_t per = ((n + pieces - 1) / pieces + 8191) & ~8191ull;
        for (uint64_t r0 = 0; r0 < n; r0 += per) {
            const uint64_t r1 = std::min<uint64_t>(n, r0 + per);
            HIP_TRY(c, hipMemcpyAsync(const_cast<uint8_t *>(d_bytes) + r0 * fixed_len, bytes + r0 * fixed_len,
                                      (size_t)((r1 - r0) * fixed_len), hipMemcpyHostToDevice, c->st_side));
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->st_side));
            HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_fork, 0));
            fs.id_base = (uint32_t)r0;
            KTIME(c, FQD_K_PACK, fqd::launch_pack(d_bytes + r0 * fixed_len, (r1 - r0) * fixed_len, nullptr, r1 - r0, fixed_len,
                                                  sh, c->d_lut.as<uint8_t>(), lut, nullptr, nullptr, nullptr, nullptr,
                                                  fqd::OwnerRule{}, c->d_ctr32.as<uint32_t>() + C_PACKBAD, c->st, &fs));
        }
    } else {
        StageTimer kernel_timer(c, FQD_T_PACK_KERNEL);
        KTIME(c, FQD_K_PACK, fqd::launch_pack(d_bytes, n_bytes, nullptr, n, fixed_len, sh, c->d_lut.as<uint8_t>(), lut,
                                              nullptr, nullptr, nullptr, nullptr, fqd::OwnerRule{},
                                              c->d_ctr32.as<uint32_t>() + C_PACKBAD, c->st, &fs));
        kernel_timer.stop();
    }
    pack_timer.stop();
    c->n = n;
    c->hashes_valid = false;
    c->recs_valid = false;
    c->owners_done = fqd::OwnerRule{};
    StageTimer timer(c, FQD_T_COLLAPSE);
    c->U = 0;
    c->n_counted = 0;
    const uint32_t *d_w;
    FQD_TRY(to_device(c, weights, (size_t)n, aux_mem, c->in_weights, &d_w));
    FusedLevel1 f{parts, sub_bits, B};
    f.level1_bits = l1_bits;
    f.part_mask = (1u << l1_bits) - 1;
    f.slab_cap1 = cap1;
    f.starts_ready = starts_ready;
    f.compact = compact;
    f.side_slabs = side_slabs;
    f.side_cap = side_cap;
    f.side_slots = side_slots;
    f.route_mask = route_mask;
    f.group_at = group_at;
    f.p0 = p0;
    f.can_spill = can_spill;
    f.side_parked = side_parked;
    if (heavy) {
        f.spill_cap = spill_cap;
        f.spill_cursor = spill_words;
        f.l1_over = spill_words + 4;
    }
    bool ok = false;
    FQD_TRY(collapse_lds(c, weights ? d_w : nullptr, IdSource(), &ok, &f));
    c->last_spill_used = f.spill_used;
    timer.stop();
    if (getenv("FQD_DEBUG"))
        fprintf(stderr, "[fqd] fused pack + collapse: n=%llu parts=%u cap=%u compact=%u done=%d pack_bad=%u fused_off=%d compact_off=%d routed=%d spill_list=%d heavy_keys=%d\n",
                (unsigned long long)n, parts, cap1, compact, (int)ok, f.pack_bad, (int)c->fused_off, (int)c->compact_off,
                (int)(route_mask != 0), (int)heavy, (int)c->heavy_keys);
    if (!ok) {
        c->route |= FQD_ROUTE_RESTARTED;
        return FQD_OK;
    }
    c->route |= FQD_ROUTE_FUSED_PACK | FQD_ROUTE_COLLAPSE_LDS | (compact ? FQD_ROUTE_COMPACT_RECORDS : 0u) |
                (c->pass0_done ? FQD_ROUTE_PASS0_IN_COLLAPSE : 0u) | (heavy ? FQD_ROUTE_SPILL_LIST : 0u);
    c->collapse_path = 1;
    c->urecs_len_pad = false;
    c->collapsed = true;
    c->first_distinct = true;
    set_id_range(c, n);
    c->stage = ST_UNIQUE;
    *done = true;
    return FQD_OK;
}

// ---- multi-GPU: the fused way in, across ranks -----------------------------------------------
// A rank packs its reads straight into owner-major slabs (bin = owner * hash_bins + top hash bits,
// `subs` slabs per bin); an owner's share is one contiguous range of slabs that the all-to-all
// moves as it is -- capacity included: the slack is what contiguity costs, 12 standard deviations
// of a Poisson slab + 64 records, ~1.16 x at 50 M reads -- and the receiving rank's collapse starts
// at level 2, reading the slabs of all senders in place. Per read and rank: one pass over the key
// bytes and one partition pass, like the single-GPU path (the general way -- fqd_pack_keys, grouping
// by owner, level 1 and level 2 at the receiver -- makes four).
// The routed collapse across ranks: every sender bins its owner slabs by a hash of SEGMENT 0 of the key (the segment
// the owner rule looks at anyway) instead of the whole key, so that the owner's compaction does search pass 0 on the
// spot as on one GPU (fqd::Pass0). All ranks must bin alike: the caller asks every rank (fqd_owner_routing_possible)
// and switches it on everywhere or nowhere (fqd_set_owner_routing).
static uint32_t owner_route_mask(uint32_t key_len, uint32_t n_segments)
{
    if (n_segments < 2 || n_segments > 4 || !key_len || key_len > 32)
        return 0;
    const uint32_t seg0_len = key_len / n_segments;            // fqd_segment(len, 0, nseg): [0, len / nseg)
    if (seg0_len < 8)
        return 0;
    return seg0_len >= 32 ? 0xFFFFFFFFu : ((1u << seg0_len) - 1u);
}

int fqd_owner_routing_possible(const fqd_ctx *c, uint32_t key_len, uint32_t n_segments, int *possible)
{
    if (!possible)
        return FQD_E_VALUE;
    *possible = !c->route_off && !c->compact_off && !getenv("FQD_NO_ROUTED_COLLAPSE") && !getenv("FQD_NO_COMPACT_RECORDS") &&
                owner_route_mask(key_len, n_segments) != 0;
    return FQD_OK;
}

int fqd_set_owner_routing(fqd_ctx *c, int enable)
{
    c->owner_routed = enable != 0;
    return FQD_OK;
}

int fqd_owner_slab_geometry(uint64_t n_max, uint32_t n_parts, uint32_t *hash_bins, uint32_t *subs, uint32_t *cap)
{
    if (!n_parts || n_parts > 256 || !hash_bins || !subs || !cap)
        return FQD_E_VALUE;
    uint32_t hb = 1;
    while (hb * 2 * n_parts <= 256)
        hb *= 2;
    *hash_bins = hb;
    *subs = 32;
    const uint64_t parts = (uint64_t)n_parts * hb * 32;
    const double mean = (double)n_max / (double)parts;
    *cap = (uint32_t)(((uint64_t)(mean + 12.0 * std::sqrt(mean) + 64.0) + 3) & ~3ull);
    return FQD_OK;
}

int fqd_pack_to_owner_slabs(fqd_ctx *c, const uint8_t *bytes, uint64_t n, uint32_t fixed_len, int mem, uint32_t n_parts,
                            uint32_t n_segments, uint32_t segment, uint32_t hash_bins, uint32_t subs, uint32_t cap,
                            uint32_t *slabs_out, uint32_t *cursors_out, uint64_t *counts, int *done)
{
    FQD_TRY(bind(c));
    if (!done || !counts || !slabs_out || !cursors_out)
        return fail(c, FQD_E_VALUE, "fqd_pack_to_owner_slabs: missing output");
    *done = 0;
    const uint32_t n_bins = n_parts * hash_bins, parts = n_bins * subs;
    if (!n_parts || n_bins > 256 || (hash_bins & (hash_bins - 1)) || subs != 32 || !cap || !n_segments ||
        segment >= n_segments || ((uintptr_t)slabs_out & 15u))
        return fail(c, FQD_E_VALUE, "fqd_pack_to_owner_slabs: bad geometry");
    std::fill(counts, counts + n_parts, 0ull);
    if (getenv("FQD_NO_FUSED_PACK") || !fixed_len || n >= 0xFFFFFF00ull || (uint64_t)parts * cap + n >= 0xFFFFFF00ull)
        return FQD_OK;
    if (mem == FQD_DEVICE && ((uintptr_t)bytes & 15u))
        return FQD_OK;
    uint8_t present[128], lut[256];
    if (c->forced) {
        if (c->forced_ragged || c->forced_max_len != fixed_len)
            return FQD_OK;
        memcpy(present, c->forced_present, 128);
    } else {
        memset(present, 0, sizeof present);
        for (const char *p = "ACGNT"; *p; p++)
            present[(int)*p] = 1;
    }
    c->stage = ST_EMPTY;
    build_alphabet(c, present, lut);
    FQD_TRY(set_geometry(c, fixed_len, 0));
    const KeyShape sh = c->ks;
    if (sh.stride != 4 || sh.planes * sh.words > 3)
        return FQD_OK;                       // longer records travel the general way
    FQD_TRY(upload_lut(c, lut));
    const uint64_t n_bytes = n * (uint64_t)fixed_len;
    const uint8_t *d_bytes;
    StageTimer pack_timer(c, FQD_T_PACK);
    FQD_TRY(to_device(c, bytes, (size_t)n_bytes, mem, c->in_bytes, &d_bytes));
    HIP_TRY(c, c->ld_seg.reserve((size_t)(parts + 4) * 4));
    FQD_TRY(zero_ctr32(c, 0, C_N32));
    // cursor p starts at the first slot of slab p (in the caller's buffers)
    HIP_TRY(c, fqd::launch_slab_starts(parts, cap, c->ld_seg.as<uint32_t>(), cursors_out, c->st));
    uint32_t hb_bits = 0;
    while ((1u << hb_bits) < hash_bins)
        hb_bits++;
    fqd::PackScatter fs{cursors_out, reinterpret_cast<uint4 *>(slabs_out), c->d_ctr32.as<uint32_t>() + C_BAD,
                        32 - hb_bits, n_bins, subs, cap};
    fs.owner_parts = n_parts;
    fs.owner_hb = hash_bins;
    if (c->owner_routed && segment == 0)
        fs.route_mask = owner_route_mask(fixed_len, n_segments);      // (the hash bins follow segment 0 too)
    const fqd::OwnerRule rule{n_parts, n_segments, segment};
    if (n) {
        StageTimer kernel_timer(c, FQD_T_PACK_KERNEL);
        KTIME(c, FQD_K_PACK, fqd::launch_pack(d_bytes, n_bytes, nullptr, n, fixed_len, sh, c->d_lut.as<uint8_t>(), lut,
                                              nullptr, nullptr, nullptr, nullptr, rule,
                                              c->d_ctr32.as<uint32_t>() + C_PACKBAD, c->st, &fs));
        kernel_timer.stop();
    }
    // the fill of every slab (for the part sizes), the overflow and foreign-byte flags: one wait
    if (!c->h_pin_big && hipHostMalloc(&c->h_pin_big, 1u << 20, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_pin_big = nullptr;
    }
    std::vector<uint32_t> cur_pageable;
    uint32_t *cur, *flags;
    if (c->h_pin_big && ((size_t)parts + C_N32) * 4 <= (1u << 20)) {      // (a pageable target waits on its own)
        cur = static_cast<uint32_t *>(c->h_pin_big);
        flags = cur + parts;
    } else {
        cur_pageable.resize((size_t)parts + C_N32);
        cur = cur_pageable.data();
        flags = cur + parts;
    }
    HIP_TRY(c, hipMemcpyAsync(cur, cursors_out, (size_t)parts * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipMemcpyAsync(flags, c->d_ctr32.p, (C_PACKBAD + 1) * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    pack_timer.stop();
    c->n = n;
    c->hashes_valid = false;
    c->recs_valid = false;
    c->owners_done = fqd::OwnerRule{};
    if ((flags[C_BAD] & 4u) || flags[C_PACKBAD])
        return FQD_OK;                       // a slab overflowed, or a byte outside the alphabet: the general way
    for (uint32_t p = 0; p < parts; p++)
        counts[p / (hash_bins * subs)] += std::min<uint64_t>(cur[p] - (uint64_t)p * cap, cap);
    *done = 1;
    return FQD_OK;
}

// The filled prefixes of a sender's owner slabs, back to back: what travels when the slack shall stay home (an
// all-to-all-v by rows instead of equal slab ranges: 1.16 x fewer bytes at 50 M reads, 1.35 x at four chunks).
// rows_out (device): room for the n reads packed; fills_out (device, n_parts * hash_bins * subs words): every slab's
// fill -- the owner needs the fills of its slabs (they are equal splits of this array) to find the slabs in the rows.
// rows_capacity: rows_out's room in rows. The fills come from the cursors, not from the caller's count: rows behind the
// capacity are not written (cursors of a pack that gave up -- the rows are not used then -- must not reach past the
// caller's buffer).
int fqd_dense_owner_slabs(fqd_ctx *c, const uint32_t *slabs, const uint32_t *cursors, uint32_t n_parts, uint32_t hash_bins,
                          uint32_t subs, uint32_t cap, uint32_t *rows_out, uint64_t rows_capacity, uint32_t *fills_out)
{
    FQD_TRY(bind(c));
    const uint64_t parts64 = (uint64_t)n_parts * hash_bins * subs;
    if (!slabs || !cursors || !rows_out || !fills_out || !n_parts || !hash_bins || subs != 32 || !cap || parts64 > 65536 ||
        parts64 * cap >= 0xFFFFFF00ull || ((uintptr_t)slabs & 15u) || ((uintptr_t)rows_out & 15u))
        return fail(c, FQD_E_VALUE, "fqd_dense_owner_slabs: bad arguments");
    const uint32_t parts = (uint32_t)parts64;
    HIP_TRY(c, c->ld_seg.reserve((size_t)3 * (parts + 4) * 4));
    uint32_t *start = c->ld_seg.as<uint32_t>();
    HIP_TRY(c, fqd::launch_fill_scan(cursors, parts, cap, fills_out, start, nullptr, c->st));
    HIP_TRY(c, fqd::launch_slab_dense_rows(slabs, start, parts, cap, rows_out, c->st, rows_capacity));
    return FQD_OK;
}

int fqd_collapse_owner_slabs(fqd_ctx *c, const uint32_t *slabs, const uint32_t *cursors, uint32_t n_senders,
                             uint32_t my_part, uint32_t hash_bins, uint32_t subs, uint32_t cap, const uint64_t *sender_id0,
                             uint64_t id_limit, uint64_t n_reads, uint32_t search_segments, uint64_t *n_unique, int *done)
{
    FQD_TRY(bind(c));
    if (!done || !slabs || !cursors || !sender_id0)
        return fail(c, FQD_E_VALUE, "fqd_collapse_owner_slabs: missing argument");
    *done = 0;
    const uint32_t ppo = hash_bins * subs;              // slabs per (sender, owner)
    const uint64_t parts64 = (uint64_t)n_senders * ppo;
    // cap == 0: DENSE rows (fqd_dense_owner_slabs on every sender): `slabs` holds the senders' filled prefixes back to
    // back, `cursors` the fill of every slab
    const bool dense = cap == 0;
    if (!n_senders || !hash_bins || (hash_bins & (hash_bins - 1)) || subs != 32 || parts64 > 65536 ||
        ((uintptr_t)slabs & 15u))
        return fail(c, FQD_E_VALUE, "fqd_collapse_owner_slabs: bad geometry");
    if (!c->shape.planes || c->ks.stride != 4 || c->ks.ragged || c->ks.planes * c->ks.words > 3)
        return fail(c, FQD_E_STATE, "fqd_collapse_owner_slabs needs the geometry of one-uint4 records");
    const uint32_t parts = (uint32_t)parts64;
    if ((dense ? n_reads : parts64 * cap) + n_reads >= 0xFFFFFF00ull || n_reads >= 0xFFFFFF00ull)
        return FQD_OK;
    // (sender, read index on the sender) must fit the record's spare word
    // (senders in ANY order -- a rank's reads may arrive as several senders, chunk by chunk: a sender's reads lie
    // below the next larger id base)
    uint64_t max_local = 0;
    {
        if (id_limit == ~0ull)
            return FQD_OK;
        std::vector<uint64_t> bases(sender_id0, sender_id0 + n_senders);
        std::sort(bases.begin(), bases.end());
        for (uint32_t s = 0; s < n_senders; s++) {
            const uint64_t next = s + 1 < n_senders ? bases[s + 1] : id_limit;
            if (next < bases[s])
                return FQD_OK;
            max_local = std::max(max_local, next - bases[s]);
        }
    }
    uint32_t lb = 1, sb = 0;
    while (lb < 32 && (max_local >> lb))
        lb++;
    while ((1u << sb) < n_senders)
        sb++;
    if (lb + sb > 32 || lb >= 32)
        return FQD_OK;
    uint32_t hb_bits = 0;
    while ((1u << hb_bits) < hash_bins)
        hb_bits++;
    uint32_t B = std::max<uint32_t>(lds_bucket_bits(n_reads), hb_bits + 1);
    B = std::min<uint32_t>(B, hb_bits + 10);             // level 2 has at most 1024 bins
    if (!hb_bits)
        return FQD_OK;                                   // (more than 128 ranks: the general way)
    c->stage = ST_EMPTY;
    c->n = n_reads;
    c->U = 0;
    c->n_counted = 0;
    c->hashes_valid = false;
    c->recs_valid = false;
    c->owners_done = fqd::OwnerRule{};
    StageTimer timer(c, FQD_T_COLLAPSE);
    if (!n_reads) {
        timer.stop();
        c->urecs_len_pad = false;
        c->collapsed = true;
        c->first_distinct = true;
        set_id_range(c, id_limit);
        c->stage = ST_UNIQUE;
        if (n_unique)
            *n_unique = 0;
        *done = 1;
        return FQD_OK;
    }
    // segment tables: seg_start (parts + 1) | seg_end (parts) | tile_start (parts + 1), 4 words apart
    HIP_TRY(c, c->ld_seg.reserve((size_t)3 * (parts + 4) * 4));
    uint32_t *seg_start = c->ld_seg.as<uint32_t>(), *seg_end = seg_start + (parts + 4);
    if (dense) {
        HIP_TRY(c, fqd::launch_fill_scan(cursors, parts, 0u, nullptr, seg_start, seg_end, c->st));
        c->ld_part.borrow(slabs, (size_t)n_reads * 16);
    } else {
        HIP_TRY(c, fqd::launch_owner_slab_bounds(cursors, n_senders, ppo, my_part, cap, seg_start, seg_end, c->st));
        c->ld_part.borrow(slabs, (size_t)parts * cap * 16);
    }
    // id bases of the senders on the device, in ascending order, and every sender's rank in that order: the rank is
    // what level 2 stamps above the read index, so stamped words compare like the ids they stand for
    std::vector<uint32_t> order(n_senders), rank_of(n_senders);
    for (uint32_t s = 0; s < n_senders; s++)
        order[s] = s;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sender_id0[a] < sender_id0[b]; });
    std::vector<uint64_t> sorted_id0(n_senders);
    for (uint32_t r = 0; r < n_senders; r++) {
        sorted_id0[r] = sender_id0[order[r]];
        rank_of[order[r]] = r;
    }
    HIP_TRY(c, c->seg_tab.reserve((size_t)n_senders * 12 + 16));
    uint32_t *d_rank_of = reinterpret_cast<uint32_t *>(c->seg_tab.as<uint64_t>() + n_senders);
    HIP_TRY(c, hipMemcpyAsync(c->seg_tab.p, sorted_id0.data(), (size_t)n_senders * 8, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipMemcpyAsync(d_rank_of, rank_of.data(), (size_t)n_senders * 4, hipMemcpyHostToDevice, c->st));
    if (dense)       // (the fills the senders told us must add up to the rows the caller received: level 2 walks them)
        FQD_TRY(queue_read_u32(c, seg_start + parts, 13));
    HIP_TRY(c, stream_wait(c->st));          // (host vectors)
    if (dense && taken_u32(c, 13) != n_reads) {
        c->ld_part.unborrow();
        return fail(c, FQD_E_VALUE, "fqd_collapse_owner_slabs: the slabs' fills do not add up to n_reads");
    }
    IdSource ids;
    ids.seg_id0 = c->seg_tab.as<uint64_t>();
    ids.n_seg = n_senders;
    ids.packed_bits = lb;
    ids.stamp_map = d_rank_of;
    FQD_TRY(zero_ctr32(c, 0, C_N32));
    FusedLevel1 f{parts, 5, B};
    f.level1_bits = hb_bits;
    f.part_mask = hash_bins - 1;
    f.stamp_div = ppo;
    f.tables_ready = true;
    // compact records as on one GPU (pack_collapse_fused_once): 12-byte items out of level 2, keys with an N on the
    // side path -- the alphabet must be exactly "ACGNT" (three planes) or have two planes
    if (!c->compact_off && !getenv("FQD_NO_COMPACT_RECORDS") && c->ks.words == 1) {
        if (c->ks.planes == 2) {
            f.compact = 2;
        } else if (c->ks.planes == 3 && c->shape.alphabet_size == 5 && !memcmp(c->shape.alphabet, "ACGNT", 5)) {
            f.compact = 1;
            f.side_slabs = 256;
            f.side_cap = (uint32_t)((((n_reads >> 6) + 16384) / f.side_slabs + 3) & ~3ull);
            f.side_slots = 1024;
            while (f.side_slots < 2ull * f.side_slabs * f.side_cap)
                f.side_slots *= 2;
            HIP_TRY(c, c->ld_side.reserve((size_t)f.side_slabs * f.side_cap * 16 + 16));
            HIP_TRY(c, c->ld_side_table.reserve(((size_t)fqd::side_table_words(f.side_slots) + 2 * f.side_slabs + 4) * 4 + 16));
            HIP_TRY(c, c->urecs.reserve(n_reads * 16 + 16));
            HIP_TRY(c, c->ucounts.reserve(n_reads * 4 + 16));
            HIP_TRY(c, c->ufirst.reserve(n_reads * 8 + 64));
        }
    }
    c->seg_hint = search_segments <= 4 ? search_segments : 0;
    // the senders binned by segment 0 (fqd_set_owner_routing, on every rank): level 2 does too, and the compaction
    // reports the pairs of search pass 0 (as pack_collapse_fused_once sets it up on one GPU)
    // (not where a bucket holds more reads than pass 0 is good for -- 5 to 8 ranks at 50 M reads each leave ~1500 reads
    // in each of the 2^15 buckets level 2 can make: a wave holds 512 unique rows, and a bucket beyond that sends the
    // whole pass back to the search. The senders' bins still follow segment 0; level 2 then hashes whole keys.)
    if (c->owner_routed && f.compact && c->seg_hint >= 2 && (n_reads >> B) <= 1200) {
        const uint32_t mask = owner_route_mask(c->ks.max_len, c->seg_hint);
        if (!mask)
            return fail(c, FQD_E_STATE, "fqd_set_owner_routing is on, but keys of this length cannot be routed");
        const uint32_t n_buckets = 1u << B;
        HIP_TRY(c, c->p0_probe.reserve((size_t)n_buckets * fqd::FQD_P0_PROBE_CAP * 4 + 16));
        HIP_TRY(c, c->ld_hist.reserve(((size_t)n_buckets + std::max(n_buckets >> 8, 1u)) * 4 + 1024 * 4));   // (collapse_lds asks for less)
        const uint64_t want_edges = std::max<uint64_t>(c->edges.cap / 8, std::max<uint64_t>(1u << 20, n_reads / 8));
        if (c->edge_cap < want_edges || !c->edges.p) {
            HIP_TRY(c, c->edges.reserve(want_edges * 8));
            c->edge_cap = c->edges.cap / 8;
        }
        fqd::Pass0 &p0 = f.p0;
        p0.mask = mask;
        p0.d = c->seg_hint - 1;
        p0.bucket_bits = B;
        p0.max_rows = fqd::pass0_max_rows();
        if (const char *e = getenv("FQD_P0_MAX_ROWS"))
            p0.max_rows = (uint32_t)std::max(1, std::min((int)fqd::pass0_max_rows(), atoi(e)));
        p0.probe = c->p0_probe.as<uint32_t>();
        p0.probe_n = c->ld_hist.as<uint32_t>();
        p0.edges = c->edges.as<uint32_t>();
        p0.edge_count = c->d_ctr64.as<unsigned long long>() + C64_EDGES;
        p0.edge_cap = c->edge_cap;
        c->pass0_edge_cap = p0.edge_cap;
        p0.flag = c->d_ctr32.as<uint32_t>() + C_P0;
        p0.stats = c->d_stats.as<fqd::PairStats>();
        HIP_TRY(c, hipMemsetAsync(p0.probe_n, 0, (size_t)n_buckets * 4, c->st));
        FQD_TRY(zero_ctr64(c, C64_EDGES));
        HIP_TRY(c, hipMemsetAsync(c->d_stats.p, 0, FQD_STAT_SLOTS * sizeof(fqd::PairStats), c->st));
        f.route_mask = mask;
    }
    // (routing on but no compact records here: the senders' level-1 bins follow the segment-0 hash, level 2 hashes the
    // whole record -- every copy of a key still meets in one bucket: correct, just without pass 0)
    bool ok = false;
    const bool compact_was_off = c->compact_off;
    int rc = collapse_lds(c, nullptr, ids, &ok, &f);
    if (rc == FQD_OK && !ok && f.compact && !compact_was_off && c->compact_off) {
        // the side slabs of the compact records overflowed (many keys with an N): once more with uint4 records -- the
        // received slabs are untouched
        f.compact = f.side_slabs = f.side_cap = f.side_slots = 0;
        f.route_mask = 0;
        f.p0 = fqd::Pass0();
        FQD_TRY(zero_ctr32(c, 0, C_N32));
        rc = collapse_lds(c, nullptr, ids, &ok, &f);
    }
    c->seg_hint = 0;
    c->ld_part.unborrow();
    timer.stop();
    FQD_TRY(rc);
    if (!ok)
        return FQD_OK;
    c->route = FQD_ROUTE_COLLAPSE_LDS | (f.compact ? FQD_ROUTE_COMPACT_RECORDS : 0u) |
               (c->pass0_done ? FQD_ROUTE_PASS0_IN_COLLAPSE : 0u);
    c->collapse_path = 1;
    c->urecs_len_pad = false;
    c->collapsed = true;
    c->first_distinct = true;
    set_id_range(c, id_limit);
    c->stage = ST_UNIQUE;
    if (n_unique)
        *n_unique = c->U;
    *done = 1;
    return FQD_OK;
}

int fqd_collapse(fqd_ctx *c, const uint32_t *weights, const uint64_t *read_ids, int mem, uint64_t *n_unique)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED || !c->recs_valid)
        return fail(c, FQD_E_STATE, "fqd_collapse before fqd_pack_keys/fqd_import_packed");
    IdSource ids;
    if (read_ids && c->n)
        FQD_TRY(to_device(c, read_ids, (size_t)c->n, mem, c->in_read_ids, &ids.ids64));
    return collapse_impl(c, weights, mem, ids, ~0ull, n_unique);
}

int fqd_collapse_received(fqd_ctx *c, const uint32_t *weights, const uint64_t *seg_rows, const uint64_t *seg_id0,
                          uint32_t n_seg, uint64_t id_limit, int mem, uint64_t *n_unique)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED || !c->recs_valid)
        return fail(c, FQD_E_STATE, "fqd_collapse_received before fqd_import_packed");
    const KeyShape sh = c->ks;
    if (sh.stride <= sh.planes * sh.words)
        return fail(c, FQD_E_VALUE, "records of this geometry have no padding word to carry an index");
    if (n_seg == 0 || n_seg > 65536 || !seg_rows || !seg_id0 || seg_rows[0] != 0 || seg_rows[n_seg] != c->n)
        return fail(c, FQD_E_VALUE, "segments must tile the packed reads");
    for (uint32_t s = 0; s < n_seg; s++)
        if (seg_rows[s] > seg_rows[s + 1])
            return fail(c, FQD_E_VALUE, "segment offsets must not decrease");
    // small tables on the device: n_seg + 1 row offsets (u32) behind n_seg id bases (u64)
    HIP_TRY(c, c->seg_tab.reserve((size_t)n_seg * 8 + ((size_t)n_seg + 1) * 4 + 16));
    std::vector<uint32_t> rows32((size_t)n_seg + 1);
    for (uint32_t s = 0; s <= n_seg; s++)
        rows32[s] = (uint32_t)seg_rows[s];
    uint64_t *d_id0 = c->seg_tab.as<uint64_t>();
    uint32_t *d_rows = reinterpret_cast<uint32_t *>(d_id0 + n_seg);
    HIP_TRY(c, hipMemcpyAsync(d_id0, seg_id0, (size_t)n_seg * 8, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipMemcpyAsync(d_rows, rows32.data(), ((size_t)n_seg + 1) * 4, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, stream_wait(c->st));      // rows32 goes out of scope
    IdSource ids;
    ids.stamped = c->recs.as<uint32_t>();
    ids.stride = sh.stride;
    ids.spare_word = sh.planes * sh.words;
    ids.seg_rows = d_rows;
    ids.seg_id0 = d_id0;
    ids.n_seg = n_seg;
    {   // (segment, local index) in 32 bits? local indices of segment s lie below the gap to the next id base
        uint64_t max_local = 0;
        bool known = id_limit != ~0ull;
        for (uint32_t s = 0; s < n_seg && known; s++) {
            const uint64_t next = s + 1 < n_seg ? seg_id0[s + 1] : id_limit;
            if (next < seg_id0[s])
                known = false;
            else
                max_local = std::max(max_local, next - seg_id0[s]);
        }
        uint32_t lb = 1, sb = 0;
        while (lb < 32 && (max_local >> lb))
            lb++;
        while ((1u << sb) < n_seg)
            sb++;
        if (known && n_seg <= 8 && lb + sb <= 32 && lb < 32) {
            ids.packed_bits = lb;
            // empty segments share their successor's first row: ">=" then counts them all at once
            uint32_t *row[7] = {&ids.row1, &ids.row2, &ids.row3, &ids.row4, &ids.row5, &ids.row6, &ids.row7};
            for (uint32_t s = 1; s < n_seg; s++)
                *row[s - 1] = rows32[s];
        }
    }
    return collapse_impl(c, weights, mem, ids, id_limit, n_unique);
}

static int cluster_tail(fqd_ctx *c, int max_distance, int metric, int method, fqd_summary *out);

// Segments of the pigeonhole search that follows a collapse inside fqd_cluster[_keys] (0: none whose
// hashes the collapse could prepare: the bucketed edit search, or more segments than pay off)
static uint32_t search_segments_hint(int max_distance, int metric)
{
    if (max_distance < 0 || max_distance > 3)
        return 0;
    if (metric == FQD_METRIC_EDIT && max_distance > 1)
        return 0;
    return (uint32_t)max_distance + 1;
}

int fqd_cluster(fqd_ctx *c, const uint32_t *weights, const uint64_t *read_ids, int mem, int max_distance, int metric,
                int method, fqd_summary *out)
{
    if (max_distance < 0)
        return fail(c, FQD_E_VALUE, "max_distance should be non-negative");
    c->seg_hint = search_segments_hint(max_distance, metric);
    const int rc = fqd_collapse(c, weights, read_ids, mem, nullptr);
    c->seg_hint = 0;
    FQD_TRY(rc);
    return cluster_tail(c, max_distance, metric, method, out);
}

int fqd_pack_collapse(fqd_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len, int mem,
                      const uint32_t *weights, const uint64_t *read_ids, int aux_mem, uint32_t search_segments,
                      uint64_t *n_unique)
{
    FQD_TRY(bind(c));
    bool done = false;
    c->seg_hint = search_segments <= 4 ? search_segments : 0;
    int rc = FQD_OK;
    if (!offsets && !read_ids)
        rc = pack_collapse_fused(c, bytes, n, fixed_len, mem, weights, aux_mem, &done);
    if (rc == FQD_OK && !done) {
        rc = fqd_pack_keys(c, bytes, offsets, n, fixed_len, mem);
        if (rc == FQD_OK)
            rc = fqd_collapse(c, weights, read_ids, aux_mem, nullptr);
    }
    c->seg_hint = 0;
    FQD_TRY(rc);
    if (n_unique)
        *n_unique = c->U;
    return FQD_OK;
}

int fqd_cluster_keys(fqd_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len, int mem,
                     const uint32_t *weights, const uint64_t *read_ids, int aux_mem, int max_distance, int metric,
                     int method, fqd_summary *out)
{
    FQD_TRY(bind(c));
    if (max_distance < 0)
        return fail(c, FQD_E_VALUE, "max_distance should be non-negative");
    bool done = false;
    c->route = 0;
    c->seg_hint = search_segments_hint(max_distance, metric);
    int rc = FQD_OK;
    if (!offsets && !read_ids)
        rc = pack_collapse_fused(c, bytes, n, fixed_len, mem, weights, aux_mem, &done);
    if (rc == FQD_OK && !done) {
        rc = fqd_pack_keys(c, bytes, offsets, n, fixed_len, mem);
        if (rc == FQD_OK)
            rc = fqd_collapse(c, weights, read_ids, aux_mem, nullptr);
    }
    c->seg_hint = 0;
    FQD_TRY(rc);
    return cluster_tail(c, max_distance, metric, method, out);
}

static int cluster_tail(fqd_ctx *c, int max_distance, int metric, int method, fqd_summary *out)
{
    if (c->join_pending) {           // (an earlier job that failed between the fork and the join)
        HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join, 0));
        c->join_pending = false;
    }
    c->pre_init = c->pre_init_closed = false;
    c->preinit_method = (method >= 0 && method <= 2 && method != FQD_METHOD_HIGHEST_COUNT && !getenv("FQD_NO_PREINIT")) ? method : -1;
    const int rc_search = fqd_find_edges(c, max_distance, metric, 0, 1, nullptr);
    c->preinit_method = -1;
    FQD_TRY(rc_search);
    // no host round trip between components and dissection; labels are flattened only if read
    // (directional, closed form: the dissection never reads the labels -- the union-find runs beside its first pass)
    const bool beside = method == FQD_METHOD_DIRECTIONAL && c->collapsed && !getenv("FQD_DIRECTIONAL_ROUNDS") &&
                        !getenv("FQD_NO_GRAPH_OVERLAP");
    FQD_TRY(fqd_api_components_queue(c, method == FQD_METHOD_HIGHEST_COUNT, beside));
    c->stage = ST_LABELS;
    c->ms[FQD_T_COMPONENTS] = 0;
    c->tpending[FQD_T_COMPONENTS] = false;
    FQD_TRY(fqd_dissect(c, method, nullptr));
    if (c->U) {
        c->n_clusters = c->roots_seen;   // read together with the kept counters
    } else {
        unsigned long long roots = 0;
        FQD_TRY(read_ctr64(c, C64_ROOTS, &roots));
        c->n_clusters = roots;
    }
    if (out) {
        out->n_reads = c->n;
        out->n_counted = c->n_counted;
        out->n_unique = c->U;
        out->n_edges = c->E;
        out->n_clusters = c->n_clusters;
        out->n_kept = c->n_kept;
    }
    return FQD_OK;
}

// ---- single calls ---------------------------------------------------------------
int fqd_within_distance(fqd_ctx *c, const uint8_t *a_bytes, const uint64_t *a_offsets, const uint8_t *b_bytes,
                        const uint64_t *b_offsets, uint64_t n, int max_distance, int metric, uint8_t *out, int mem)
{
    FQD_TRY(bind(c));
    if (!n)
        return FQD_OK;
    if (mem != FQD_HOST)
        return fail(c, FQD_E_VALUE, "fqd_within_distance takes host buffers");
    if (metric == FQD_METRIC_EDIT && max_distance > 64) {
        // only refuse when the answer is not trivially decided by the lengths
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t la = a_offsets[i + 1] - a_offsets[i], lb = b_offsets[i + 1] - b_offsets[i];
            if ((uint64_t)max_distance < std::max(la, lb) && (la > lb ? la - lb : lb - la) <= (uint64_t)max_distance)
                return fail(c, FQD_E_VALUE, "edit distance bound above 64 is not supported on device");
        }
    }
    const uint8_t *da, *db;
    const uint64_t *dao, *dbo;
    FQD_TRY(to_device(c, a_bytes, (size_t)a_offsets[n], FQD_HOST, c->stage_a, &da));
    FQD_TRY(to_device(c, a_offsets, (size_t)n + 1, FQD_HOST, c->stage_b, &dao));
    FQD_TRY(to_device(c, b_bytes, (size_t)b_offsets[n], FQD_HOST, c->stage_c, &db));
    FQD_TRY(to_device(c, b_offsets, (size_t)n + 1, FQD_HOST, c->stage_d, &dbo));
    HIP_TRY(c, c->tmp.reserve(n + 16));
    HIP_TRY(c, fqd::launch_pairs_within(da, dao, db, dbo, n, max_distance, metric, c->tmp.as<uint8_t>(), c->st));
    return from_device(c, out, c->tmp.p, (size_t)n, FQD_HOST);
}

int fqd_contains(fqd_ctx *c, const uint8_t *q_bytes, const uint64_t *q_offsets, uint64_t n, int max_distance,
                 int metric, uint8_t *out, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "fqd_contains before fqd_collapse/fqd_import_unique");
    if (mem != FQD_HOST)
        return fail(c, FQD_E_VALUE, "fqd_contains takes host buffers");
    if (!n)
        return FQD_OK;
    if (n > 65535)
        return fail(c, FQD_E_VALUE, "at most 65535 queries per call");
    if (metric == FQD_METRIC_EDIT && max_distance > 64 && (uint32_t)max_distance < c->ks.max_len)
        return fail(c, FQD_E_VALUE, "edit distance bound above 64 is not supported on device");
    for (uint64_t i = 0; i < n; i++)
        out[i] = 0;
    if (!c->U || max_distance < 0)
        return FQD_OK;
    const uint8_t *dq;
    const uint64_t *dqo;
    FQD_TRY(to_device(c, q_bytes, (size_t)q_offsets[n] + 1, FQD_HOST, c->stage_a, &dq));
    FQD_TRY(to_device(c, q_offsets, (size_t)n + 1, FQD_HOST, c->stage_b, &dqo));
    HIP_TRY(c, c->d_alphabet.reserve(128));
    HIP_TRY(c, hipMemcpyAsync(c->d_alphabet.p, c->shape.alphabet, 128, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, c->stage_c.reserve(n * 4 + 16));
    HIP_TRY(c, hipMemsetAsync(c->stage_c.p, 0, n * 4, c->st));
    // rows taken out by fqd_store_remove (popped clusters) are not in the trie any more
    const uint8_t *alive = (c->store_removed && c->store_table_U == c->U) ? c->store_alive.as<uint8_t>() : nullptr;
    HIP_TRY(c, fqd::launch_contains(dq, dqo, n, c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), c->U, c->ks,
                                    c->d_alphabet.as<uint8_t>(), max_distance, metric, c->stage_c.as<uint32_t>(),
                                    c->st, alive));
    std::vector<uint32_t> flags((size_t)n);
    HIP_TRY(c, hipMemcpyAsync(flags.data(), c->stage_c.p, n * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    for (uint64_t i = 0; i < n; i++)
        out[i] = flags[i] ? 1 : 0;
    return FQD_OK;
}

// ---- quality gate -----------------------------------------------------------------
int fqd_quality_filter(fqd_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len,
                       uint32_t phred_offset, double threshold, const double *table128, uint32_t *pass_out,
                       double *means_out, uint64_t *n_discarded, int mem)
{
    FQD_TRY(bind(c));
    if (phred_offset > 126)
        return fail(c, FQD_E_VALUE, "phred_offset out of range");
    if (n_discarded)
        *n_discarded = 0;
    if (!n)
        return FQD_OK;
    if (n >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "at most 2^32-16 reads per call");
    if (mem == FQD_DEVICE && ((uintptr_t)bytes & 15u))
        return fail(c, FQD_E_VALUE, "device buffer must be 16-byte aligned");
    uint64_t n_bytes;
    if (offsets) {
        if (mem == FQD_HOST) {
            n_bytes = offsets[n];
        } else {
            HIP_TRY(c, hipMemcpyAsync(&n_bytes, offsets + n, 8, hipMemcpyDeviceToHost, c->st));
            HIP_TRY(c, stream_wait(c->st));
        }
    } else {
        n_bytes = n * (uint64_t)fixed_len;
    }
    const uint8_t *d_bytes;
    const uint64_t *d_off = nullptr;
    FQD_TRY(to_device(c, bytes, (size_t)n_bytes, mem, c->q_bytes, &d_bytes));
    if (offsets)
        FQD_TRY(to_device(c, offsets, (size_t)n + 1, mem, c->q_offsets, &d_off));
    uint32_t max_len = fixed_len;
    if (offsets) {
        uint32_t mm[2] = {0xFFFFFFFFu, 0u};
        HIP_TRY(c, hipMemcpyAsync(c->d_ctr32.as<uint32_t>() + C_MINLEN, mm, 8, hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, fqd::launch_scan_lens(d_off, n, c->d_ctr32.as<uint32_t>() + C_MINLEN, c->st));
        HIP_TRY(c, hipMemcpyAsync(mm, c->d_ctr32.as<uint32_t>() + C_MINLEN, 8, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(c, stream_wait(c->st));
        max_len = mm[1];
    }
    double table[128];
    for (int i = 0; i < 128; i++)
        table[i] = table128 ? table128[i] : std::pow(10.0, -((double)i / 10.0));  // score_to_error_rate.py
    HIP_TRY(c, c->q_table.reserve(sizeof table));
    HIP_TRY(c, hipMemcpyAsync(c->q_table.p, table, sizeof table, hipMemcpyHostToDevice, c->st));
    uint32_t *d_pass = pass_out;
    double *d_means = means_out;
    if (mem == FQD_HOST) {
        HIP_TRY(c, c->q_pass.reserve(n * 4 + 16));
        d_pass = c->q_pass.as<uint32_t>();
        if (means_out) {
            HIP_TRY(c, c->q_means.reserve(n * 8 + 16));
            d_means = c->q_means.as<double>();
        }
    }
    FQD_TRY(zero_ctr32(c, C_BAD));
    HIP_TRY(c, fqd::launch_quality(d_bytes, n_bytes, d_off, n, fixed_len, max_len, c->q_table.as<double>(),
                                   phred_offset, 126u - phred_offset, threshold, d_pass, d_means,
                                   c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
    uint32_t bad = 0;
    FQD_TRY(read_ctr32(c, C_BAD, &bad));
    if (bad)
        return fail(c, FQD_E_VALUE, "a phred string holds a character outside the valid phred range");
    FQD_TRY(zero_ctr64(c, C64_SUM));
    HIP_TRY(c, fqd::launch_sum_u32(d_pass, n, c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
    unsigned long long passed = 0;
    FQD_TRY(read_ctr64(c, C64_SUM, &passed));
    if (n_discarded)
        *n_discarded = n - passed;
    if (mem == FQD_HOST) {
        FQD_TRY(from_device(c, pass_out, d_pass, (size_t)n, FQD_HOST));
        if (means_out)
            FQD_TRY(from_device(c, means_out, d_means, (size_t)n, FQD_HOST));
    }
    return FQD_OK;
}

// ---- measurement ----------------------------------------------------------------
int fqd_set_timing(fqd_ctx *c, int stage_timers, uint32_t kernel_mask)
{
    FQD_TRY(bind(c));
    stage_times_resolve(c);
    ktime_collect(c);
    c->stage_timing = stage_timers != 0;
    c->ktime_mask = kernel_mask;
    return FQD_OK;
}

int fqd_stage_times(fqd_ctx *c, float *ms, uint32_t *launches)
{
    FQD_TRY(bind(c));
    stage_times_resolve(c);
    if (ms)
        memcpy(ms, c->ms, sizeof c->ms);
    if (launches)
        memcpy(launches, c->launches, sizeof c->launches);
    return FQD_OK;
}

int fqd_kernel_times(fqd_ctx *c, float *ms, uint32_t *launches, int reset)
{
    ktime_collect(c);
    if (ms)
        memcpy(ms, c->kms, sizeof c->kms);
    if (launches)
        memcpy(launches, c->klaunches, sizeof c->klaunches);
    if (reset) {
        memset(c->kms, 0, sizeof c->kms);
        memset(c->klaunches, 0, sizeof c->klaunches);
    }
    return FQD_OK;
}

int fqd_edge_stats(fqd_ctx *c, uint64_t *keys_gathered, uint64_t *pairs_compared, uint64_t *edges_emitted)
{
    if (c->stats_pending) {
        FQD_TRY(bind(c));
        fqd::PairStats slots[FQD_STAT_SLOTS];
        HIP_TRY(c, hipMemcpyAsync(slots, c->d_stats.p, sizeof slots, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(c, stream_wait(c->st));
        c->last_stats = fqd::PairStats{0, 0, 0};
        for (const fqd::PairStats &p : slots) {
            c->last_stats.keys_gathered += p.keys_gathered;
            c->last_stats.pairs_compared += p.pairs_compared;
            c->last_stats.edges += p.edges;
        }
        c->stats_pending = false;
    }
    if (keys_gathered)
        *keys_gathered = c->last_stats.keys_gathered;
    if (pairs_compared)
        *pairs_compared = c->last_stats.pairs_compared;
    if (edges_emitted)
        *edges_emitted = c->last_stats.edges;
    return FQD_OK;
}

int fqd_synth_keys(fqd_ctx *c, uint8_t *out_device, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length,
                   uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub)
{
    FQD_TRY(bind(c));
    if (!copies)
        return fail(c, FQD_E_VALUE, "copies must be positive");
    HIP_TRY(c, fqd::launch_synth(out_device, n_total, start, count, length, umi, seed, copies, thr_n, thr_sub, c->st));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

int fqd_synth_keys_skewed(fqd_ctx *c, uint8_t *out_device, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length,
                          uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub, uint64_t thr_hot,
                          uint64_t thr_ladder, uint32_t lowc_every)
{
    FQD_TRY(bind(c));
    if (!copies)
        return fail(c, FQD_E_VALUE, "copies must be positive");
    if (length < 8)
        return fail(c, FQD_E_VALUE, "the ladder of the skewed model needs keys of 8 bases or more");
    HIP_TRY(c, fqd::launch_synth(out_device, n_total, start, count, length, umi, seed, copies, thr_n, thr_sub, c->st,
                                 thr_hot, thr_ladder, lowc_every, 1u));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

int fqd_copy_bandwidth(fqd_ctx *c, const void *src_device, void *dst_device, uint64_t bytes, uint32_t reps, double *gb_per_s)
{
    FQD_TRY(bind(c));
    if (!gb_per_s || !src_device || !dst_device || (bytes & 15u) || (((uintptr_t)src_device | (uintptr_t)dst_device) & 15u))
        return fail(c, FQD_E_VALUE, "fqd_copy_bandwidth: 16-byte aligned device buffers, a multiple of 16 bytes");
    *gb_per_s = 0.0;
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    HIP_TRY(c, fqd::launch_copy16(src_device, dst_device, bytes, c->st));       // warm-up
    for (uint32_t r = 0; r < std::max<uint32_t>(reps, 1); r++) {
        HIP_TRY(c, hipEventRecord(e0, c->st));
        HIP_TRY(c, fqd::launch_copy16(src_device, dst_device, bytes, c->st));
        HIP_TRY(c, hipEventRecord(e1, c->st));
        HIP_TRY(c, hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
        if (ms > 0)
            *gb_per_s = std::max(*gb_per_s, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return FQD_OK;
}

int fqd_synth_indel_keys(fqd_ctx *c, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length, uint32_t umi,
                         uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub, uint64_t thr_indel,
                         uint64_t *lens_out_device, const uint64_t *offsets_device, uint8_t *out_device)
{
    FQD_TRY(bind(c));
    if (!copies)
        return fail(c, FQD_E_VALUE, "copies must be positive");
    if (!lens_out_device && !(offsets_device && out_device))
        return fail(c, FQD_E_VALUE, "fqd_synth_indel_keys: either lens_out or offsets + out");
    HIP_TRY(c, fqd::launch_synth_indels(n_total, start, count, length, umi, seed, copies, thr_n, thr_sub, thr_indel,
                                        reinterpret_cast<unsigned long long *>(lens_out_device),
                                        reinterpret_cast<const unsigned long long *>(offsets_device), out_device, c->st));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

}  // extern "C"

// the collapse over c->recs with weights and ids that already sit on the device (api_trie.hip: the
// resident unique table merged with new keys)
int fqd_api_collapse_device(fqd_ctx *c, const uint32_t *d_weights, IdSource ids, uint64_t id_limit, uint64_t *n_unique)
{
    return collapse_impl(c, d_weights, FQD_DEVICE, ids, id_limit, n_unique);
}

