// api_trie.hip -- C ABI, part 5: the Trie OBJECT of the reference as a device-resident store.
//   fqd_store_add_keys / fqd_store_remove   Trie.add_sequence in batches, the removal half of
//                                           Trie.pop_cluster (_triemodule.c:677-706, :830-831, :875-876)
//   fqd_get_clusters / fqd_read_clusters    every remaining cluster in the order pop_cluster would
//                                           return them (_triemodule.c:778-897, seeds :510-551)
//   fqd_trie_order, fqd_trie_stats          Trie.memory_size / Trie.raw_stats (:553-594, :909-964)
// Kernels: trieorder.hip.
#include "api_ctx.h"

namespace {

// code (rank of the symbol in the record alphabet) -> index of the symbol in the caller's alphabet
// (Trie.alphabet: constructor symbols, then symbols in the order the trie met them)
int code_index_table(fqd_ctx *c, const uint8_t *alphabet, uint32_t n_alpha, uint8_t *idx_of_code /* 128 */)
{
    if (n_alpha > 254)
        return fail(c, FQD_E_VALUE, "Maximum alphabet length exceeded");
    int pos[256];
    for (int b = 0; b < 256; b++)
        pos[b] = -1;
    for (uint32_t i = 0; i < n_alpha; i++) {
        if (pos[alphabet[i]] >= 0)
            return fail(c, FQD_E_VALUE, "Alphabet should consist of unique characters.");
        pos[alphabet[i]] = (int)i;
    }
    for (uint32_t code = 0; code < 128; code++) {
        const int p = code < c->shape.alphabet_size ? pos[c->shape.alphabet[code]] : -1;
        idx_of_code[code] = (uint8_t)(p >= 0 ? p : (int)n_alpha);   // a symbol the caller's alphabet lacks sorts last
    }
    return FQD_OK;
}

// c->t_order[r] = uid of the key at rank r of the reference's key order; c->t_rank = its inverse
int trie_order(fqd_ctx *c, const uint8_t *alphabet, uint32_t n_alpha)
{
    const uint64_t U = c->U;
    const KeyShape sh = c->ks;
    uint8_t idx[128];
    FQD_TRY(code_index_table(c, alphabet, n_alpha, idx));
    HIP_TRY(c, c->t_idx.reserve(128));
    HIP_TRY(c, hipMemcpyAsync(c->t_idx.p, idx, 128, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, stream_wait(c->st));          // idx lives on this stack frame
    HIP_TRY(c, c->t_order.reserve(U * 4 + 16));
    HIP_TRY(c, c->t_order_b.reserve(U * 4 + 16));
    HIP_TRY(c, c->t_rank.reserve(U * 4 + 16));
    HIP_TRY(c, c->t_keys.reserve(U * 8 + 16));
    HIP_TRY(c, c->t_keys_b.reserve(U * 8 + 16));
    HIP_TRY(c, fqd::launch_trie_iota(c->t_order.as<uint32_t>(), U, c->st));
    // digits 0 .. n_alpha (END, and symbols outside the caller's alphabet): `bits` each, P per 64-bit sort key
    uint32_t bits = 1;
    while ((1u << bits) < n_alpha + 1u)
        bits++;
    const uint32_t P = 64u / bits;
    const uint32_t positions = sh.max_len, chunks = (positions + P - 1) / P;
    const size_t need = fqd::sort_pairs_u64_u32_temp(U, (int)(bits * P));
    HIP_TRY(c, c->tmp.reserve(need + 16));
    uint32_t *cur = c->t_order.as<uint32_t>(), *nxt = c->t_order_b.as<uint32_t>();
    for (uint32_t ch = chunks; ch-- > 0 && U > 1;) {      // least significant chunk first; the sort is stable
        HIP_TRY(c, fqd::launch_trie_chunk_keys(cur, U, c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), sh,
                                               c->t_idx.as<uint8_t>(), n_alpha, bits, ch * P, P,
                                               c->t_keys.as<unsigned long long>(), c->st));
        HIP_TRY(c, fqd::sort_pairs_u64_u32(c->tmp.p, need, c->t_keys.as<uint64_t>(), c->t_keys_b.as<uint64_t>(), cur, nxt,
                                           U, (int)(bits * P), c->st));
        std::swap(cur, nxt);
    }
    if (cur != c->t_order.as<uint32_t>() && U)
        HIP_TRY(c, hipMemcpyAsync(c->t_order.p, cur, U * 4, hipMemcpyDeviceToDevice, c->st));
    HIP_TRY(c, fqd::launch_trie_rank_of(c->t_order.as<uint32_t>(), U, c->t_rank.as<uint32_t>(), c->st));
    return FQD_OK;
}

const uint8_t *alive_or_null(const fqd_ctx *c)
{
    return c->store_removed ? c->store_alive.as<uint8_t>() : nullptr;
}

int reset_alive(fqd_ctx *c)
{
    HIP_TRY(c, c->store_alive.reserve(c->U + 16));
    if (c->U)
        HIP_TRY(c, hipMemsetAsync(c->store_alive.p, 1, c->U, c->st));
    c->store_removed = 0;
    c->store_table_U = c->U;
    return FQD_OK;
}

}  // namespace

extern "C" {

int fqd_trie_order(fqd_ctx *c, const uint8_t *alphabet, uint32_t n_alpha, uint32_t *order_out, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "fqd_trie_order before fqd_collapse/fqd_import_unique");
    if (!alphabet && n_alpha)
        return fail(c, FQD_E_VALUE, "alphabet missing");
    FQD_TRY(trie_order(c, alphabet, n_alpha));
    if (order_out)
        return from_device(c, order_out, c->t_order.p, (size_t)c->U, mem);
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

int fqd_trie_stats(fqd_ctx *c, const uint8_t *alphabet, uint32_t n_alpha, uint32_t n_layers, uint64_t *memory_size,
                   uint64_t *stats)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "fqd_trie_stats before fqd_collapse/fqd_import_unique");
    if (!alphabet && n_alpha)
        return fail(c, FQD_E_VALUE, "alphabet missing");
    const uint64_t U = c->U;
    const uint32_t n_cols = n_alpha + 1;
    const size_t cells = (size_t)n_layers * n_cols;
    if (stats)
        std::fill(stats, stats + cells, 0ull);
    if (memory_size)
        *memory_size = 0;
    if (!U)
        return FQD_OK;
    if (c->store_removed && c->store_table_U != U)
        return fail(c, FQD_E_STATE, "removed rows belong to another unique table");
    FQD_TRY(trie_order(c, alphabet, n_alpha));
    HIP_TRY(c, c->t_lcp.reserve((U + 1) * 4 + 16));
    HIP_TRY(c, fqd::launch_trie_lcp(c->t_order.as<uint32_t>(), U, c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), c->ks,
                                    c->t_lcp.as<uint32_t>(), c->st));
    const uint8_t *alive = alive_or_null(c);
    const uint32_t *last_alive = nullptr;
    if (alive) {
        HIP_TRY(c, c->t_mark.reserve(U * 4 + 16));
        HIP_TRY(c, c->t_mark_incl.reserve(U * 4 + 16));
        HIP_TRY(c, fqd::launch_trie_alive_mark(c->t_order.as<uint32_t>(), U, alive, c->t_mark.as<uint32_t>(), c->st));
        const size_t need = fqd::scan_max_u32_temp(U);
        HIP_TRY(c, c->tmp.reserve(need + 16));
        HIP_TRY(c, fqd::inclusive_scan_max_u32(c->tmp.p, need, c->t_mark.as<uint32_t>(), c->t_mark_incl.as<uint32_t>(), U,
                                               c->st));
        last_alive = c->t_mark_incl.as<uint32_t>();
    }
    HIP_TRY(c, c->t_stats.reserve((cells + 1) * 8 + 16));
    HIP_TRY(c, hipMemsetAsync(c->t_stats.p, 0, (cells + 1) * 8, c->st));
    unsigned long long *d_stats = c->t_stats.as<unsigned long long>();
    HIP_TRY(c, fqd::launch_trie_census(c->t_order.as<uint32_t>(), c->t_lcp.as<uint32_t>(), U, c->urecs.as<uint32_t>(),
                                       c->ulens.as<uint32_t>(), c->ks, c->t_idx.as<uint8_t>(), alive, last_alive, n_layers,
                                       n_cols, d_stats + 1, d_stats, c->st));
    std::vector<unsigned long long> host(cells + 1);
    HIP_TRY(c, hipMemcpyAsync(host.data(), c->t_stats.p, (cells + 1) * 8, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    if (memory_size)
        *memory_size = host[0];
    if (stats)
        for (size_t i = 0; i < cells; i++)
            stats[i] = host[i + 1];
    return FQD_OK;
}

int fqd_get_clusters(fqd_ctx *c, const uint8_t *alphabet, uint32_t n_alpha, uint64_t *n_clusters, uint64_t *n_members)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_LABELS)
        return fail(c, FQD_E_STATE, "fqd_get_clusters before fqd_components");
    if (!alphabet && n_alpha)
        return fail(c, FQD_E_VALUE, "alphabet missing");
    const uint64_t U = c->U;
    c->t_clusters = c->t_members = 0;
    c->t_clusters_valid = false;
    if (c->store_removed && c->store_table_U != U)
        return fail(c, FQD_E_STATE, "removed rows belong to another unique table");
    if (U) {
        FQD_TRY(fqd_api_flat_labels(c));
        FQD_TRY(trie_order(c, alphabet, n_alpha));
        const uint8_t *alive = alive_or_null(c);
        HIP_TRY(c, c->t_seed.reserve(U * 4 + 16));
        HIP_TRY(c, c->t_heads.reserve(U * 4 + 16));
        HIP_TRY(c, c->t_heads_incl.reserve(U * 4 + 16));
        HIP_TRY(c, c->t_member_uids.reserve(U * 4 + 16));
        HIP_TRY(c, c->t_offsets.reserve((U + 1) * 8 + 16));
        HIP_TRY(c, hipMemsetAsync(c->t_seed.p, 0xFF, U * 4, c->st));
        HIP_TRY(c, fqd::launch_trie_seed_ranks(c->labels.as<uint32_t>(), c->t_rank.as<uint32_t>(), U, alive,
                                               c->t_seed.as<uint32_t>(), c->st));
        HIP_TRY(c, fqd::launch_trie_member_keys(c->labels.as<uint32_t>(), c->t_rank.as<uint32_t>(), U, alive,
                                                c->t_seed.as<uint32_t>(), c->t_keys.as<unsigned long long>(), c->st));
        const size_t need = fqd::sort_keys_u64_temp(U);
        HIP_TRY(c, c->tmp.reserve(need + 16));
        HIP_TRY(c, fqd::sort_keys_u64(c->tmp.p, need, c->t_keys.as<uint64_t>(), c->t_keys_b.as<uint64_t>(), U, 64, c->st));
        HIP_TRY(c, fqd::launch_trie_member_heads(c->t_keys_b.as<unsigned long long>(), U, c->t_order.as<uint32_t>(),
                                                 c->t_heads.as<uint32_t>(), c->t_member_uids.as<uint32_t>(), c->st));
        FQD_TRY(scan_u32(c, c->t_heads.as<uint32_t>(), c->t_heads_incl.as<uint32_t>(), U));
        HIP_TRY(c, hipMemsetAsync(c->t_offsets.p, 0, 8, c->st));
        HIP_TRY(c, fqd::launch_trie_cluster_offsets(c->t_keys_b.as<unsigned long long>(), c->t_heads.as<uint32_t>(),
                                                    c->t_heads_incl.as<uint32_t>(), U,
                                                    c->t_offsets.as<unsigned long long>(), c->st));
        uint32_t n_cl = 0;
        HIP_TRY(c, hipMemcpyAsync(&n_cl, c->t_heads_incl.as<uint32_t>() + (U - 1), 4, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(c, stream_wait(c->st));
        unsigned long long members = 0;
        if (n_cl) {
            HIP_TRY(c, hipMemcpyAsync(&members, c->t_offsets.as<unsigned long long>() + n_cl, 8, hipMemcpyDeviceToHost,
                                      c->st));
            HIP_TRY(c, stream_wait(c->st));
        }
        c->t_clusters = n_cl;
        c->t_members = members;
    }
    c->t_clusters_valid = true;
    if (n_clusters)
        *n_clusters = c->t_clusters;
    if (n_members)
        *n_members = c->t_members;
    return FQD_OK;
}

int fqd_read_clusters(fqd_ctx *c, uint64_t *offsets, uint32_t *member_uids, int mem)
{
    FQD_TRY(bind(c));
    if (!c->t_clusters_valid)
        return fail(c, FQD_E_STATE, "fqd_read_clusters before fqd_get_clusters");
    if (offsets) {
        if (c->t_clusters || c->t_members) {
            FQD_TRY(from_device(c, offsets, c->t_offsets.p, (size_t)c->t_clusters + 1, mem));
        } else if (mem == FQD_HOST) {
            offsets[0] = 0;
        } else {
            HIP_TRY(c, hipMemsetAsync(offsets, 0, 8, c->st));
            HIP_TRY(c, stream_wait(c->st));
        }
    }
    if (member_uids && c->t_members)
        FQD_TRY(from_device(c, member_uids, c->t_member_uids.p, (size_t)c->t_members, mem));
    return FQD_OK;
}

int fqd_store_symbol_events(fqd_ctx *c, const uint8_t *alphabet, uint32_t n_alpha, int reuse_order,
                            const uint8_t *symbols, uint32_t n_symbols, const uint64_t *after, uint64_t *cand_first,
                            uint32_t *cand_depth, uint64_t *partner_first)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "fqd_store_symbol_events: no unique table");
    if (n_symbols > 128 || (!symbols && n_symbols) || (!alphabet && n_alpha))
        return fail(c, FQD_E_VALUE, "fqd_store_symbol_events: bad symbol list");
    const uint64_t U = c->U;
    for (uint32_t s = 0; s < n_symbols; s++) {
        cand_first[s] = ~0ull;
        cand_depth[s] = 0xFFFFFFFFu;
        partner_first[s] = ~0ull;
    }
    if (!U || !n_symbols)
        return FQD_OK;
    if (c->id_limit != ~0ull && c->id_limit > (1ull << 32))
        return fail(c, FQD_E_VALUE, "fqd_store_symbol_events needs first-holder ids below 2^32");
    if (c->store_removed && c->store_table_U != U)
        return fail(c, FQD_E_STATE, "removed rows belong to another unique table");
    if (!reuse_order) {
        FQD_TRY(trie_order(c, alphabet, n_alpha));
        HIP_TRY(c, c->t_lcp.reserve((U + 1) * 4 + 16));
        HIP_TRY(c, fqd::launch_trie_lcp(c->t_order.as<uint32_t>(), U, c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(),
                                        c->ks, c->t_lcp.as<uint32_t>(), c->st));
    }
    // symbols -> codes of the record alphabet (0xFF: not in any key)
    uint8_t codes[128];
    for (uint32_t s = 0; s < n_symbols; s++) {
        codes[s] = 0xFF;
        for (uint32_t a = 0; a < c->shape.alphabet_size; a++)
            if (c->shape.alphabet[a] == symbols[s])
                codes[s] = (uint8_t)a;
    }
    // small device block: codes (128 B) | after (n x 8) | cand (n x 8) | partner (n x 8) | depth (n x 4)
    const size_t n = n_symbols;
    HIP_TRY(c, c->t_stats.reserve(128 + n * 28 + 64));
    uint8_t *d_codes = c->t_stats.as<uint8_t>();
    unsigned long long *d_after = reinterpret_cast<unsigned long long *>(d_codes + 128);
    unsigned long long *d_cand = d_after + n, *d_partner = d_cand + n;
    uint32_t *d_depth = reinterpret_cast<uint32_t *>(d_partner + n);
    HIP_TRY(c, hipMemcpyAsync(d_codes, codes, 128, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipMemcpyAsync(d_after, after, n * 8, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipMemsetAsync(d_cand, 0xFF, n * 16, c->st));
    HIP_TRY(c, fqd::launch_symbol_round(c->t_order.as<uint32_t>(), c->t_lcp.as<uint32_t>(), U, c->urecs.as<uint32_t>(),
                                        c->ulens.as<uint32_t>(), c->ks, c->ufirst.as<uint64_t>(), alive_or_null(c), d_codes,
                                        n_symbols, d_after, d_cand, d_depth, d_partner, c->st));
    std::vector<unsigned long long> h_cand(n), h_partner(n);
    HIP_TRY(c, hipMemcpyAsync(h_cand.data(), d_cand, n * 8, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipMemcpyAsync(h_partner.data(), d_partner, n * 8, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipMemcpyAsync(cand_depth, d_depth, n * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    for (uint32_t s = 0; s < n_symbols; s++) {
        cand_first[s] = h_cand[s] == ~0ull ? ~0ull : h_cand[s] >> 32;
        partner_first[s] = h_partner[s];
    }
    return FQD_OK;
}

int fqd_store_remove(fqd_ctx *c, const uint32_t *uids, uint64_t n, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "fqd_store_remove: no unique table");
    if (!c->store_alive.p || c->store_table_U != c->U)
        FQD_TRY(reset_alive(c));
    if (!n)
        return FQD_OK;
    const uint32_t *d_uids;
    FQD_TRY(to_device(c, uids, (size_t)n, mem, c->stage_a, &d_uids));
    FQD_TRY(zero_ctr32(c, C_BAD));
    HIP_TRY(c, fqd::launch_store_remove(d_uids, n, c->U, c->store_alive.as<uint8_t>(), c->d_ctr32.as<uint32_t>() + C_BAD,
                                        c->st));
    uint32_t bad = 0;
    FQD_TRY(read_ctr32(c, C_BAD, &bad));
    if (bad)
        return fail(c, FQD_E_VALUE, "fqd_store_remove: a uid lies outside the unique table");
    c->store_removed += n;
    c->t_clusters_valid = false;
    return FQD_OK;
}

int fqd_store_removed_count(const fqd_ctx *c, uint64_t *n_removed)
{
    if (!c || !n_removed)
        return FQD_E_VALUE;
    *n_removed = (c->stage >= ST_UNIQUE && c->store_table_U == c->U) ? c->store_removed : 0;
    return FQD_OK;
}

int fqd_store_add_keys(fqd_ctx *c, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, uint32_t fixed_len, int mem,
                       const uint32_t *weights, const uint64_t *read_ids, int aux_mem, uint64_t *n_unique)
{
    FQD_TRY(bind(c));
    const bool resident = c->stage >= ST_UNIQUE && c->U > 0;
    if (!resident) {
        // first batch: the plain way in
        c->no_len_pad = true;
        const int rc_pack = fqd_pack_keys(c, bytes, offsets, n, fixed_len, mem);
        c->no_len_pad = false;
        FQD_TRY(rc_pack);
        FQD_TRY(fqd_collapse(c, weights, read_ids, aux_mem, nullptr));
        FQD_TRY(reset_alive(c));
        if (!read_ids)
            c->store_next_id = n;           // ids of a first batch without ids are its positions
        if (n_unique)
            *n_unique = c->U;
        return FQD_OK;
    }
    if (c->store_removed && c->store_table_U != c->U)
        return fail(c, FQD_E_STATE, "removed rows belong to another unique table");
    if (c->U + n >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "at most 2^32-16 keys per context");
    const uint64_t U0 = c->U;
    const KeyShape old_ks = c->ks;
    const fqd_shape old_shape = c->shape;
    // ---- what the new keys need: symbols, longest key, raggedness ----
    uint8_t present[128];
    memset(present, 0, sizeof present);
    for (uint32_t a = 0; a < old_shape.alphabet_size; a++)
        present[old_shape.alphabet[a]] = 1;
    uint32_t max_len = old_shape.max_len;
    int ragged = (int)old_shape.ragged;
    if (n) {
        uint8_t p_new[128];
        uint32_t ml = 0;
        int rg = 0;
        FQD_TRY(fqd_scan_keys(c, bytes, offsets, n, fixed_len, mem, p_new, &ml, &rg));
        for (int b = 0; b < 128; b++)
            present[b] |= p_new[b];
        if (rg || ml != max_len)
            ragged = 1;
        max_len = std::max(max_len, ml);
    }
    // ---- the resident rows step aside; the new keys are packed with the common geometry ----
    if (c->urecs_len_pad && U0) {
        // (a table a one-call job left: its rows hold their lengths in their last padding word, the store's do not)
        HIP_TRY(c, fqd::launch_clear_last_word(c->urecs.as<uint32_t>(), U0, old_ks.stride, c->st));
        c->urecs_len_pad = false;
    }
    std::swap(c->urecs, c->st_recs);
    std::swap(c->ulens, c->st_lens);
    std::swap(c->ucounts, c->st_counts);
    std::swap(c->ufirst, c->st_first);
    const bool was_forced = c->forced;
    uint8_t forced_present[128];
    memcpy(forced_present, c->forced_present, 128);
    const uint32_t forced_max_len = c->forced_max_len;
    const int forced_ragged = c->forced_ragged;
    int rc = fqd_configure(c, present, max_len, ragged);
    if (rc == FQD_OK) {
        c->no_len_pad = true;
        rc = fqd_pack_keys(c, bytes, offsets, n, fixed_len, mem);
        c->no_len_pad = false;
    }
    c->forced = was_forced;
    memcpy(c->forced_present, forced_present, 128);
    c->forced_max_len = forced_max_len;
    c->forced_ragged = forced_ragged;
    if (rc != FQD_OK) {
        // nothing was merged: the resident table comes back (in its own geometry)
        std::swap(c->urecs, c->st_recs);
        std::swap(c->ulens, c->st_lens);
        std::swap(c->ucounts, c->st_counts);
        std::swap(c->ufirst, c->st_first);
        const std::string msg = c->err;
        uint8_t old_present[128];
        memset(old_present, 0, sizeof old_present);
        for (uint32_t a = 0; a < old_shape.alphabet_size; a++)
            old_present[old_shape.alphabet[a]] = 1;
        (void)fqd_configure(c, old_present, old_shape.max_len, (int)old_shape.ragged);
        c->forced = was_forced;
        memcpy(c->forced_present, forced_present, 128);
        c->forced_max_len = forced_max_len;
        c->forced_ragged = forced_ragged;
        c->U = U0;
        c->stage = ST_UNIQUE;
        c->err = msg;
        return rc;
    }
    const KeyShape ks = c->ks;
    // ---- one packed buffer: resident rows (weight = count, or 0 once removed), then the new keys ----
    const uint64_t n_all = U0 + n;
    HIP_TRY(c, c->st_comb_recs.reserve(n_all * ks.stride * 4 + 16));
    HIP_TRY(c, c->st_comb_lens.reserve(n_all * 4 + 16));
    HIP_TRY(c, c->st_comb_w.reserve(n_all * 4 + 16));
    HIP_TRY(c, c->st_comb_ids.reserve(n_all * 8 + 16));
    uint32_t *comb_recs = c->st_comb_recs.as<uint32_t>(), *comb_lens = c->st_comb_lens.as<uint32_t>();
    uint32_t *comb_w = c->st_comb_w.as<uint32_t>();
    uint64_t *comb_ids = c->st_comb_ids.as<uint64_t>();
    const bool same_geometry = ks.planes == old_ks.planes && ks.words == old_ks.words && ks.stride == old_ks.stride &&
                               ks.ragged == old_ks.ragged &&
                               !memcmp(old_shape.alphabet, c->shape.alphabet, sizeof old_shape.alphabet);
    if (same_geometry) {
        HIP_TRY(c, hipMemcpyAsync(comb_recs, c->st_recs.p, U0 * ks.stride * 4, hipMemcpyDeviceToDevice, c->st));
        if (ks.ragged)
            HIP_TRY(c, hipMemcpyAsync(comb_lens, c->st_lens.p, U0 * 4, hipMemcpyDeviceToDevice, c->st));
    } else {
        uint8_t code_map[128];
        memset(code_map, 0, sizeof code_map);
        for (uint32_t a = 0; a < old_shape.alphabet_size; a++)
            for (uint32_t b = 0; b < c->shape.alphabet_size; b++)
                if (c->shape.alphabet[b] == old_shape.alphabet[a])
                    code_map[a] = (uint8_t)b;
        HIP_TRY(c, c->t_idx.reserve(128));
        HIP_TRY(c, hipMemcpyAsync(c->t_idx.p, code_map, 128, hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, stream_wait(c->st));
        HIP_TRY(c, fqd::launch_transcode_records(c->st_recs.as<uint32_t>(), c->st_lens.as<uint32_t>(), U0, old_ks, ks,
                                                 c->t_idx.as<uint8_t>(), comb_recs, ks.ragged ? comb_lens : nullptr,
                                                 c->st));
    }
    HIP_TRY(c, fqd::launch_store_weights(c->st_counts.as<uint32_t>(), alive_or_null(c), U0, comb_w, c->st));
    HIP_TRY(c, hipMemcpyAsync(comb_ids, c->st_first.p, U0 * 8, hipMemcpyDeviceToDevice, c->st));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(comb_recs + U0 * ks.stride, c->recs.p, n * ks.stride * 4, hipMemcpyDeviceToDevice, c->st));
        if (ks.ragged)
            HIP_TRY(c, hipMemcpyAsync(comb_lens + U0, c->lens.p, n * 4, hipMemcpyDeviceToDevice, c->st));
        const hipMemcpyKind kind = aux_mem == FQD_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        if (weights)
            HIP_TRY(c, hipMemcpyAsync(comb_w + U0, weights, n * 4, kind, c->st));
        else
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)(comb_w + U0), 1, n, c->st));
        if (read_ids)
            HIP_TRY(c, hipMemcpyAsync(comb_ids + U0, read_ids, n * 8, kind, c->st));
        else
            HIP_TRY(c, fqd::launch_store_fill_ids(comb_ids + U0, n, c->store_next_id, c->st));
        HIP_TRY(c, stream_wait(c->st));     // (host weights / ids may be pageable)
    }
    std::swap(c->recs, c->st_comb_recs);
    std::swap(c->lens, c->st_comb_lens);
    c->n = n_all;
    c->hashes_valid = false;
    c->recs_len_pad = false;
    c->recs_valid = true;
    c->owners_done = fqd::OwnerRule{};
    c->stage = ST_PACKED;
    IdSource ids;
    ids.ids64 = comb_ids;
    FQD_TRY(fqd_api_collapse_device(c, comb_w, ids, ~0ull, nullptr));
    c->recs_valid = false;                  // c->recs holds table rows + new keys, not a caller's reads
    if (!read_ids)
        c->store_next_id += n;
    FQD_TRY(reset_alive(c));
    if (n_unique)
        *n_unique = c->U;
    return FQD_OK;
}

}  // extern "C"
