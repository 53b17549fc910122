// api_exchange.hip -- C ABI, part 4: what a multi-GPU caller moves between ranks (sharded.py):
// packed reads, unique tables and edge lists in and out, grouped by owner rank (exchange.hip holds
// the kernels).
#include "api_ctx.h"

extern "C" {

// ---- exchange -------------------------------------------------------------------
int fqd_export_packed(fqd_ctx *c, uint32_t *recs, uint32_t *lens, uint32_t *hashes, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED || !c->recs_valid)
        return fail(c, FQD_E_STATE, "nothing packed");
    FQD_TRY(from_device(c, recs, c->recs.p, (size_t)c->n * c->ks.stride, mem));
    if (lens) {
        if (c->ks.ragged) {
            FQD_TRY(from_device(c, lens, c->lens.p, (size_t)c->n, mem));
        } else if (mem == FQD_HOST) {
            std::fill(lens, lens + c->n, c->ks.max_len);
        } else if (c->n) {
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)lens, (int)c->ks.max_len, (size_t)c->n, c->st));
        }
    }
    if (hashes)
        FQD_TRY(fqd_api_ensure_hashes(c));
    FQD_TRY(from_device(c, hashes, c->hashes.p, (size_t)c->n, mem));
    return FQD_OK;
}

// Rows 0..n-1 of a record table grouped by owner[] (values < n_parts; part 0 first, stable):
// one radix pass over ceil(log2 parts) bits, one coalesced gather, the part sizes to the host.
static int export_grouped(fqd_ctx *c, uint64_t n, uint32_t n_parts, const uint32_t *owner, const uint32_t *src_recs,
                          const uint32_t *src_lens, const uint32_t *weights, uint64_t id0, uint32_t *recs,
                          uint32_t *lens, uint64_t *ids, uint32_t *ids32, uint32_t *weights_out, uint64_t *counts,
                          uint32_t stamp_word = 0)
{
    const KeyShape sh = c->ks;
    HIP_TRY(c, c->ids_sorted.reserve(n * 4 + 16));
    HIP_TRY(c, c->stage_d.reserve((size_t)n_parts * 8 + 16));
    uint32_t *order = c->ids_sorted.as<uint32_t>();
    if (n == 0) {
        std::fill(counts, counts + n_parts, 0ull);
        return FQD_OK;
    }
    if (n_parts <= fqd::split_max_parts() && !getenv("FQD_GROUP_BY_SORT")) {
        // stable multi-split: count per (part, tile), scan, place
        const size_t cells = (size_t)n_parts * fqd::split_tiles(n);
        HIP_TRY(c, c->ld_matrix.reserve(cells * 4 + 16));
        HIP_TRY(c, c->ld_matrix_incl.reserve(cells * 4 + 16));
        HIP_TRY(c, fqd::launch_split_count(owner, n, n_parts, c->ld_matrix.as<uint32_t>(), c->st));
        FQD_TRY(scan_u32(c, c->ld_matrix.as<uint32_t>(), c->ld_matrix_incl.as<uint32_t>(), cells));
        HIP_TRY(c, fqd::launch_split_order(owner, n, n_parts, c->ld_matrix.as<uint32_t>(),
                                           c->ld_matrix_incl.as<uint32_t>(), order, c->stage_d.as<uint64_t>(), c->st));
    } else {
        int bits = 1;
        while ((1u << bits) < n_parts)
            bits++;
        HIP_TRY(c, c->ids.reserve(n * 4 + 16));
        HIP_TRY(c, c->run_idx.reserve(n * 4 + 16));
        uint32_t *owner_sorted = c->run_idx.as<uint32_t>();
        HIP_TRY(c, fqd::launch_iota_u32(c->ids.as<uint32_t>(), n, c->st));
        FQD_TRY(sort_u32_pairs(c, owner, owner_sorted, c->ids.as<uint32_t>(), order, n, bits));
        HIP_TRY(c, fqd::launch_owner_counts(owner_sorted, n, n_parts, c->stage_d.as<uint64_t>(), c->st));
    }
    HIP_TRY(c, fqd::launch_gather_by_owner(order, n, sh, src_recs, src_lens, weights, id0, recs, lens, ids, ids32,
                                           weights_out, c->st, stamp_word));
    HIP_TRY(c, hipMemcpyAsync(counts, c->stage_d.p, (size_t)n_parts * 8, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

static int check_parts(fqd_ctx *c, uint32_t n_parts, int mem, const char *who)
{
    if (mem != FQD_DEVICE)
        return fail(c, FQD_E_VALUE, std::string(who) + " works on device buffers (counts: host)");
    if (n_parts == 0 || n_parts > 65536)
        return fail(c, FQD_E_VALUE, "1..65536 parts");
    return FQD_OK;
}

int fqd_export_packed_by_owner(fqd_ctx *c, uint32_t n_parts, uint64_t id0, const uint32_t *weights, uint32_t *recs,
                               uint32_t *lens, uint64_t *ids, uint32_t *weights_out, uint64_t *counts, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED || !c->recs_valid)
        return fail(c, FQD_E_STATE, "nothing packed");
    FQD_TRY(check_parts(c, n_parts, mem, "fqd_export_packed_by_owner"));
    const uint64_t n = c->n;
    HIP_TRY(c, c->flags.reserve(n * 4 + 16));
    FQD_TRY(fqd_api_ensure_hashes(c));
    HIP_TRY(c, fqd::launch_owner(c->hashes.as<uint32_t>(), n, n_parts, c->flags.as<uint32_t>(), c->st));
    return export_grouped(c, n, n_parts, c->flags.as<uint32_t>(), c->recs.as<uint32_t>(), c->lens.as<uint32_t>(),
                          weights, id0, recs, lens, ids, nullptr, weights_out, counts);
}

int fqd_export_packed_by_segment(fqd_ctx *c, uint32_t n_parts, uint32_t n_segments, uint32_t segment, uint64_t id0,
                                 const uint32_t *weights, uint32_t *recs, uint32_t *lens, uint64_t *ids,
                                 uint32_t *weights_out, uint64_t *counts, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED || !c->recs_valid)
        return fail(c, FQD_E_STATE, "nothing packed");
    FQD_TRY(check_parts(c, n_parts, mem, "fqd_export_packed_by_segment"));
    if (n_segments == 0 || segment >= n_segments)
        return fail(c, FQD_E_VALUE, "bad segment");
    const uint64_t n = c->n;
    const uint32_t *owner;
    if (c->owners_done.parts == n_parts && c->owners_done.nseg == n_segments && c->owners_done.seg == segment) {
        owner = c->owners.as<uint32_t>();      // fqd_pack_keys already worked them out
    } else {
        HIP_TRY(c, c->flags.reserve(n * 4 + 16));
        HIP_TRY(c, fqd::launch_segment_hashes(c->recs.as<uint32_t>(), c->lens.as<uint32_t>(), n, c->ks, n_segments,
                                              segment, segment + 1, n_parts, c->flags.as<uint32_t>(), c->st));
        owner = c->flags.as<uint32_t>();
    }
    // no id array asked for: the read's index on this rank rides in the record's first padding word
    uint32_t stamp_word = 0;
    if (!ids) {
        stamp_word = c->ks.planes * c->ks.words;
        if (stamp_word >= c->ks.stride)
            return fail(c, FQD_E_VALUE, "records of this geometry have no padding word: pass an ids buffer");
    }
    return export_grouped(c, n, n_parts, owner, c->recs.as<uint32_t>(), c->lens.as<uint32_t>(),
                          weights, id0, recs, lens, ids, nullptr, weights_out, counts, stamp_word);
}

int fqd_export_unique_by_segment(fqd_ctx *c, uint32_t n_parts, uint32_t n_segments, uint32_t segment,
                                 uint32_t uid_base, uint32_t *recs, uint32_t *lens, uint32_t *uids, uint64_t *counts,
                                 int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "no unique table yet");
    FQD_TRY(check_parts(c, n_parts, mem, "fqd_export_unique_by_segment"));
    if (n_segments == 0 || segment >= n_segments)
        return fail(c, FQD_E_VALUE, "bad segment");
    const uint64_t U = c->U;
    if ((uint64_t)uid_base + U > 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "global unique ids must stay below 2^32-16");
    HIP_TRY(c, c->flags.reserve(U * 4 + 16));
    HIP_TRY(c, fqd::launch_segment_hashes(c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), U, c->ks, n_segments,
                                          segment, segment + 1, n_parts, c->flags.as<uint32_t>(), c->st));
    return export_grouped(c, U, n_parts, c->flags.as<uint32_t>(), c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(),
                          nullptr, uid_base, recs, lens, nullptr, uids, nullptr, counts);
}

// Rows idx[0..n) of the unique table (an owner answering another rank's request for key data).
int fqd_gather_unique(fqd_ctx *c, const uint32_t *idx, uint64_t n, uint32_t *recs, uint32_t *lens, uint32_t *counts,
                      int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "no unique table yet");
    if (mem != FQD_DEVICE)
        return fail(c, FQD_E_VALUE, "fqd_gather_unique works on device buffers");
    if (n >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "too many rows");
    // an index past the table would be a wild read: check on the device first
    FQD_TRY(zero_ctr32(c, C_BAD));
    HIP_TRY(c, fqd::launch_check_indices(idx, n, c->U, c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
    uint32_t bad = 0;
    FQD_TRY(read_ctr32(c, C_BAD, &bad));
    if (bad)
        return fail(c, FQD_E_VALUE, "row index outside the unique table");
    HIP_TRY(c, fqd::launch_gather_by_owner(idx, n, c->ks, c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(),
                                           c->ucounts.as<uint32_t>(), 0, recs, lens, nullptr, nullptr, counts, c->st));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

int fqd_import_packed(fqd_ctx *c, const uint32_t *recs, const uint32_t *lens, uint64_t n, int mem)
{
    FQD_TRY(bind(c));
    if (!c->shape.planes || !c->ks.stride)
        return fail(c, FQD_E_STATE, "fqd_import_packed needs a geometry (fqd_configure + fqd_pack_keys first)");
    if (n >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "at most 2^32-16 keys per context");
    const KeyShape sh = c->ks;
    const hipMemcpyKind kind = mem == FQD_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const bool borrow = mem == FQD_DEVICE_BORROW && n > 0;
    if (sh.ragged && !lens)
        return fail(c, FQD_E_VALUE, "ragged geometry needs lens");
    HIP_TRY(c, c->hashes.reserve((size_t)n * 4 + 16));
    if (borrow) {
        if ((uintptr_t)recs & 15u)
            return fail(c, FQD_E_VALUE, "borrowed record buffer must be 16-byte aligned");
        c->recs.borrow(recs, (size_t)n * sh.stride * 4);
        if (sh.ragged)
            c->lens.borrow(lens, (size_t)n * 4);
    } else {
        HIP_TRY(c, c->recs.reserve((size_t)n * sh.stride * 4 + 16));
        if (n)
            HIP_TRY(c, hipMemcpyAsync(c->recs.p, recs, (size_t)n * sh.stride * 4, kind, c->st));
        if (sh.ragged) {
            HIP_TRY(c, c->lens.reserve((size_t)n * 4 + 16));
            if (n)
                HIP_TRY(c, hipMemcpyAsync(c->lens.p, lens, (size_t)n * 4, kind, c->st));
        }
    }
    if (mem == FQD_HOST)           // (device sources: the copy is ordered on the context's stream, nothing to wait for)
        HIP_TRY(c, stream_wait(c->st));
    c->n = n;
    c->hashes_valid = false;       // computed when somebody needs them (ensure_hashes)
    c->owners_done = fqd::OwnerRule{};
    c->recs_len_pad = false;
    c->recs_valid = true;
    c->stage = ST_PACKED;
    return FQD_OK;
}

int fqd_export_unique(fqd_ctx *c, uint32_t *recs, uint32_t *lens, uint32_t *counts, uint64_t *first_ids, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "no unique table yet");
    FQD_TRY(from_device(c, recs, c->urecs.p, (size_t)c->U * c->ks.stride, mem));
    if (lens) {
        if (c->ks.ragged) {
            FQD_TRY(from_device(c, lens, c->ulens.p, (size_t)c->U, mem));
        } else if (mem == FQD_HOST) {
            std::fill(lens, lens + c->U, c->ks.max_len);
        } else if (c->U) {
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)lens, (int)c->ks.max_len, (size_t)c->U, c->st));
        }
    }
    FQD_TRY(from_device(c, counts, c->ucounts.p, (size_t)c->U, mem));
    FQD_TRY(from_device(c, first_ids, c->ufirst.p, (size_t)c->U, mem));
    return FQD_OK;
}

int fqd_import_unique(fqd_ctx *c, const uint32_t *recs, const uint32_t *lens, const uint32_t *counts,
                      const uint64_t *first_ids, uint64_t U, int mem)
{
    FQD_TRY(bind(c));
    if (!c->shape.planes || !c->ks.stride)
        return fail(c, FQD_E_STATE, "fqd_import_unique needs a geometry (fqd_configure + fqd_pack_keys first)");
    if (U >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "at most 2^32-16 unique keys per context");
    const KeyShape sh = c->ks;
    const hipMemcpyKind kind = mem == FQD_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const bool borrow = mem == FQD_DEVICE_BORROW && U > 0;   // records and lengths stay where they are
    if (sh.ragged && !lens && U)
        return fail(c, FQD_E_VALUE, "ragged geometry needs lens");
    if (borrow) {
        if ((uintptr_t)recs & 15u)
            return fail(c, FQD_E_VALUE, "borrowed record buffer must be 16-byte aligned");
        c->urecs.borrow(recs, U * sh.stride * 4);
        if (sh.ragged)
            c->ulens.borrow(lens, U * 4);
        else
            HIP_TRY(c, c->ulens.reserve(16));
    } else {
        HIP_TRY(c, c->urecs.reserve(U * sh.stride * 4 + 16));
        HIP_TRY(c, c->ulens.reserve(U * 4 + 16));
    }
    HIP_TRY(c, c->ucounts.reserve(U * 4 + 16));
    HIP_TRY(c, c->ufirst.reserve(U * 8 + 64));
    if (U) {
        if (!borrow) {
            HIP_TRY(c, hipMemcpyAsync(c->urecs.p, recs, U * sh.stride * 4, kind, c->st));
            if (sh.ragged)
                HIP_TRY(c, hipMemcpyAsync(c->ulens.p, lens, U * 4, kind, c->st));
        }
        // a table used only for a neighbour search (a routed pass) needs neither column
        if (counts)
            HIP_TRY(c, hipMemcpyAsync(c->ucounts.p, counts, U * 4, kind, c->st));
        else
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->ucounts.p, 1, U, c->st));
        if (first_ids)
            HIP_TRY(c, hipMemcpyAsync(c->ufirst.p, first_ids, U * 8, kind, c->st));
        else
            HIP_TRY(c, hipMemsetAsync(c->ufirst.p, 0, U * 8, c->st));
    }
    if (mem == FQD_HOST)
        HIP_TRY(c, stream_wait(c->st));
    c->U = U;
    c->seg_hashes_nseg = 0;
    c->urecs_len_pad = false;
    c->collapsed = false;  // imported rows may repeat a key (dissection of a caller's list)
    c->first_distinct = first_ids != nullptr;
    c->id_bits = 1;
    c->id_limit = 1;
    if (first_ids) {   // width of the largest first-holder id: the kept-id sort runs over that many bits only
        unsigned long long mx = 0;
        FQD_TRY(zero_ctr64(c, C64_SUM));
        HIP_TRY(c, fqd::launch_max_u64(c->ufirst.as<uint64_t>(), U, c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
        FQD_TRY(read_ctr64(c, C64_SUM, &mx));
        c->id_bits = 1;
        c->id_limit = mx + 1;
        while (c->id_bits < 64 && (mx >> c->id_bits))
            c->id_bits++;
    }
    c->stage = ST_UNIQUE;
    return FQD_OK;
}

int fqd_declare_distinct_keys(fqd_ctx *c)
{
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "no unique table yet");
    c->collapsed = true;
    return FQD_OK;
}

int fqd_export_edges(fqd_ctx *c, uint32_t *uv, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_EDGES)
        return fail(c, FQD_E_STATE, "no edges yet");
    return from_device(c, uv, c->edges.p, (size_t)c->E * 2, mem);
}

int fqd_import_edges(fqd_ctx *c, const uint32_t *uv, uint64_t E, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "edges need a unique table first");
    HIP_TRY(c, c->edges.reserve(E * 8 + 16));
    c->edge_cap = c->edges.cap / 8;
    if (E)
        HIP_TRY(c, hipMemcpyAsync(c->edges.p, uv, E * 8,
                                  mem == FQD_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->st));
    if (mem != FQD_DEVICE)
        HIP_TRY(c, stream_wait(c->st));
    c->E = E;
    c->stage = ST_EDGES;
    return FQD_OK;
}

}  // extern "C"
