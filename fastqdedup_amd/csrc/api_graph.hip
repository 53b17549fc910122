// api_graph.hip -- C ABI, part 3: components, dissection, the kept-id list and the getters of the
// unique table (stages 4 and 5; graph.hip holds the kernels).
#include "api_ctx.h"

extern "C" {

static int ensure_flat_labels(fqd_ctx *c)
{
    if (c->join_pending) {           // (a dissection that failed before its own wait)
        HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join, 0));
        c->join_pending = false;
    }
    if (c->labels_flat)
        return FQD_OK;
    HIP_TRY(c, c->tmp.reserve(64));
    KTIME(c, FQD_K_UF_FLATTEN, fqd::launch_uf_flatten(c->labels.as<uint32_t>(), c->U,
                                                      c->tmp.as<unsigned long long>(), c->st));
    c->labels_flat = true;
    return FQD_OK;
}

// Queue the union-find kernels; the component count stays on the device (C64_ROOTS) until
// somebody asks for it (fqd_cluster asks after the dissection, so the GPU never waits for the
// host in between). flatten = false leaves the parent forest unflattened: components = nodes -
// hooks needs no sweep over the nodes, and only highest_count and the label export read labels
// (ensure_flat_labels does the sweep then).
// on_side (fqd_cluster[_keys] before a dissection that does not read the labels): the union-find runs on the
// context's second stream, BESIDE the dissection's first pass over the same edges -- both are chains of dependent
// random accesses that leave most of the memory system idle (0.10 + 0.10 ms one after the other at config 3); the
// main stream waits for ev_join before the component count is read (c->join_pending).
static int components_queue(fqd_ctx *c, bool flatten, bool on_side = false)
{
    const uint64_t U = c->U;
    c->pass1_done = false;
    if (c->pre_init && c->pre_nodes) {
        // The set-up wrote node records (parent, state byte): components AND pass 1 of the closed-form directional
        // dissection in one sweep over the edges (graph.hip union_directional_kernel), then the records are taken apart
        // into the labels and state arrays everything downstream reads. On the context's own stream: there is nothing
        // left to run beside it.
        c->pre_nodes = false;
        const uint64_t E = c->E;
        if (E >= 0xFFFFFFFFull)
            return fail(c, FQD_E_VALUE, "more than 2^32 edges");
        HIP_TRY(c, c->taint.reserve(E * 8 + 16));          // the edges between count-1 keys, and behind them those pass 2 looks at
        if (c->store_removed && c->store_table_U == U)
            HIP_TRY(c, fqd::launch_mask_dead_edges(c->edges.as<uint32_t>(), E, c->store_alive.as<uint8_t>(), c->st));
        KTIME(c, FQD_K_UF_UNION, fqd::launch_union_directional(
                  c->nodes.as<uint32_t>(), c->edges.as<uint32_t>(), E, c->ucounts.as<uint32_t>(), c->taint.as<uint32_t>(),
                  c->d_ctr64.as<unsigned long long>() + C64_CANDS, c->hook_slots.as<unsigned long long>(), c->st,
                  c->uf_sampled && !getenv("FQD_UF_NO_SAMPLING")));
        HIP_TRY(c, fqd::launch_unzip_nodes(c->nodes.as<uint32_t>(), c->labels.as<uint32_t>(), c->state.as<uint8_t>(), U, c->st));
        HIP_TRY(c, fqd::launch_hook_total(c->hook_slots.as<unsigned long long>(), U,
                                          c->d_ctr64.as<unsigned long long>() + C64_ROOTS, c->st,
                                          c->d_ctr64.as<unsigned long long>() + C64_UF_AGAIN));
        c->pass1_done = E != 0;
        c->labels_flat = false;
        if (flatten)
            FQD_TRY(ensure_flat_labels(c));
        return FQD_OK;
    }
    if (!c->pre_init) {                // (else: queued by fqd_api_graph_preinit while the host waited for the edge count)
        HIP_TRY(c, c->labels.reserve(U * 4 + 16));
        HIP_TRY(c, c->hook_slots.reserve(FQD_HOOK_SLOTS * 64));
        HIP_TRY(c, hipMemsetAsync(c->hook_slots.p, 0, FQD_HOOK_SLOTS * 64, c->st));
        HIP_TRY(c, fqd::launch_uf_init(c->labels.as<uint32_t>(), U, c->st));
    }
    if (c->store_removed && c->store_table_U == U)     // keys of popped clusters link nothing any more
        HIP_TRY(c, fqd::launch_mask_dead_edges(c->edges.as<uint32_t>(), c->E, c->store_alive.as<uint8_t>(), c->st));
    on_side = on_side && !flatten && c->st_side && c->E;
    hipStream_t st = c->st;
    if (on_side) {
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->st));
        HIP_TRY(c, hipStreamWaitEvent(c->st_side, c->ev_fork, 0));
        st = c->st_side;
    }
    KTIME_ON(c, FQD_K_UF_UNION, st, fqd::launch_uf_union(c->labels.as<uint32_t>(), c->edges.as<uint32_t>(), c->E,
                                                         c->hook_slots.as<unsigned long long>(), st,
                                                         c->uf_sampled && !getenv("FQD_UF_NO_SAMPLING")));
    HIP_TRY(c, fqd::launch_hook_total(c->hook_slots.as<unsigned long long>(), U,
                                      c->d_ctr64.as<unsigned long long>() + C64_ROOTS, st,
                                      c->d_ctr64.as<unsigned long long>() + C64_UF_AGAIN));
    if (on_side) {
        HIP_TRY(c, hipEventRecord(c->ev_join, c->st_side));
        c->join_pending = true;
    }
    c->labels_flat = false;
    if (flatten)
        FQD_TRY(ensure_flat_labels(c));
    return FQD_OK;
}

// Everything of stages 4 and 5 that depends on the unique table but not on the edges: parents and hook
// counters of the components, best / state of the dissection, and for the closed-form directional
// dissection the root taints and the parents of the count-1 sets. fqd_cluster queues this behind the
// search's read-back of the edge count, so the GPU has work while the host waits for that number.
static int graph_preinit(fqd_ctx *c, int method)
{
    const uint64_t U = c->U;
    c->pre_init = c->pre_init_closed = false;
    c->pre_zero_tail = false;
    if (!U)
        return FQD_OK;
    HIP_TRY(c, c->labels.reserve(U * 4 + 16));
    HIP_TRY(c, c->hook_slots.reserve(FQD_HOOK_SLOTS * 64));
    HIP_TRY(c, c->best.reserve(U * 4 + 16));
    HIP_TRY(c, c->state.reserve(U + 16));
    const bool closed = method == FQD_METHOD_DIRECTIONAL && c->collapsed && !getenv("FQD_DIRECTIONAL_ROUNDS");
    if (closed)
        HIP_TRY(c, c->root_taint.reserve(U + 16));
    c->pre_nodes = false;
    // (from distance 2 on -- edges are many per key there: config 4 has 0.72 per key and the one sweep takes 0.76 ms where
    // the two kernels side by side took 0.80, step 7.07 -> 7.00 ms; at distance 1, config 3 with 0.12 edges per key, taking
    // the records apart again costs more than the sweep saves: 2.16 against 2.11 ms. FQD_NODE_RECORDS=1 / FQD_NO_NODE_RECORDS=1 pin it)
    const bool want_nodes = getenv("FQD_NODE_RECORDS") || (c->last_search_d >= 2 && !getenv("FQD_NO_NODE_RECORDS"));
    if (closed && want_nodes && !getenv("FQD_NO_GRAPH_OVERLAP")) {
        // components and pass 1 will run as ONE sweep on node records (components_queue): the set-up writes those
        HIP_TRY(c, c->nodes.reserve(U * 8 + 64));
        HIP_TRY(c, c->kept_u32.reserve(std::max<size_t>(U * 4 + 16, (size_t)512 * fqd::kept_bin_lists() * 4 + 16)));
        HIP_TRY(c, fqd::launch_graph_preinit_nodes(c->nodes.as<uint32_t>(), c->best.as<uint32_t>(), c->root_taint.as<uint8_t>(),
                                                   c->ucounts.as<uint32_t>(), U, c->hook_slots.as<unsigned long long>(),
                                                   FQD_HOOK_SLOTS * 8, c->st, c->kept_u32.as<uint32_t>(),
                                                   512 * fqd::kept_bin_lists(), c->d_ctr64.as<unsigned long long>() + C64_SUM,
                                                   c->d_ctr64.as<unsigned long long>() + C64_CANDS,
                                                   c->d_ctr64.as<unsigned long long>() + C64_CAND_NEED));
        c->pre_zero_tail = true;
        c->pre_init_closed = true;
        c->pre_init = true;
        c->pre_nodes = true;
        return FQD_OK;
    }
    // (the cursor table of the kept-id bins, at the size fqd_dissect will ask for -- no reallocation behind this)
    HIP_TRY(c, c->kept_u32.reserve(std::max<size_t>(U * 4 + 16, (size_t)512 * fqd::kept_bin_lists() * 4 + 16)));
    // one launch for all of it (graph.hip graph_preinit_kernel)
    HIP_TRY(c, fqd::launch_graph_preinit(c->labels.as<uint32_t>(), c->best.as<uint32_t>(), c->state.as<uint8_t>(),
                                         closed ? c->root_taint.as<uint8_t>() : nullptr, U,
                                         c->hook_slots.as<unsigned long long>(), FQD_HOOK_SLOTS * 8, c->st,
                                         c->kept_u32.as<uint32_t>(), 512 * fqd::kept_bin_lists(),
                                         c->d_ctr64.as<unsigned long long>() + C64_SUM,
                                         c->d_ctr64.as<unsigned long long>() + C64_CANDS,
                                         closed ? c->ucounts.as<uint32_t>() : nullptr,
                                         c->d_ctr64.as<unsigned long long>() + C64_CAND_NEED));
    c->pre_zero_tail = true;
    c->pre_init_closed = closed;
    c->pre_init = true;
    return FQD_OK;
}

int fqd_components(fqd_ctx *c, uint64_t *n_clusters)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_EDGES)
        return fail(c, FQD_E_STATE, "fqd_components before fqd_find_edges/fqd_import_edges");
    c->stage = ST_EDGES;
    StageTimer timer(c, FQD_T_COMPONENTS);
    FQD_TRY(components_queue(c, false));   // labels are flattened when somebody reads them
    unsigned long long roots = 0;
    FQD_TRY(read_ctr64(c, C64_ROOTS, &roots));
    timer.stop();
    c->n_clusters = roots;
    c->stage = ST_LABELS;
    if (n_clusters)
        *n_clusters = roots;
    return FQD_OK;
}

// Verdicts (best / state) -> kept flags, the counters and the ascending list of kept first-holder
// ids inside the id window.
static int list_kept(fqd_ctx *c, int method)
{
    const uint64_t U = c->U;
    c->n_kept = 0;
    c->n_listed = 0;
    c->kept_in_out = false;
    c->kept_list_lost = false;
    if (!U)
        return FQD_OK;
    // First-holder ids are distinct and bounded (by the id window, or by id_limit): when that
    // range is not much larger than the table, the ascending list is a compaction of a byte map of
    // the range -- cheaper than scan + gather + a radix sort of the ids.
    uint64_t base = 0, window = c->id_limit;
    if (c->id_hi != ~0ull) {
        base = c->id_lo;
        window = std::min(window > base ? window - base : 0, c->id_hi - c->id_lo);
    }
    const bool by_map = c->first_distinct && window <= 32 * U && window < 0xFFFFFFF0ull &&
                        !getenv("FQD_KEPT_BY_SORT");
    const bool tail_zeroed = c->pre_zero_tail;
    c->pre_zero_tail = false;
    if (!tail_zeroed)
        FQD_TRY(zero_ctr64(c, C64_SUM));
    // ... and cheaper still through id bins, without the map (graph.hip kept_bin_kernel): windows of
    // up to 512 bins x 2^18 ids
    if (by_map && window && fqd::kept_bin_shift(window) <= 18 &&
        window * fqd::kept_bin_lists() < 0xF0000000ull && !getenv("FQD_KEPT_BY_MAP")) {
        const uint32_t shift = fqd::kept_bin_shift(window);
        const uint64_t slots = (((window + (1ull << shift) - 1) >> shift) << shift) * fqd::kept_bin_lists();
        HIP_TRY(c, c->kept_lists.reserve(slots * 4 + 16));
        HIP_TRY(c, c->kept_u32.reserve((size_t)512 * fqd::kept_bin_lists() * 4 + 16));
        HIP_TRY(c, c->kept_scan.reserve(64));
        c->kept_in_out = c->kept_out && c->kept_out_cap >= std::min(window, U);
        uint64_t *list_out = c->kept_out;
        if (!c->kept_in_out) {
            HIP_TRY(c, c->kept_ids_sorted.reserve(std::min(window, U) * 8 + 16));
            list_out = c->kept_ids_sorted.as<uint64_t>();
        }
        KTIME(c, FQD_K_KEPT_FLAGS, fqd::launch_kept_bins(
                  method, c->labels.as<uint32_t>(), c->best.as<uint32_t>(), c->state.as<uint8_t>(),
                  c->ufirst.as<uint64_t>(), c->id_lo, window, U, c->kept.as<uint8_t>(), c->ucounts.as<uint32_t>(),
                  c->labels.as<uint32_t>(), c->root_taint.as<uint8_t>(), c->kept_u32.as<uint32_t>(),
                  c->kept_lists.as<uint32_t>(), c->d_ctr64.as<unsigned long long>() + C64_SUM, base, list_out,
                  c->kept_scan.as<uint32_t>(), c->st, tail_zeroed));
        FQD_TRY(queue_read_u32(c, c->kept_scan.as<uint32_t>(), 0));
        unsigned long long both[7] = {0, 0, 0, 0, 0, 0, 0};      // C64_ROOTS, C64_SUM ... C64_UF_AGAIN: one read for fqd_cluster
        FQD_TRY(read_ctr64(c, C64_ROOTS, both, C64_UF_AGAIN - C64_ROOTS + 1));
        c->roots_seen = both[0];
        c->n_kept = both[1];
        c->n_listed = taken_u32(c, 0);
        // a union-find with a second walk for every fifth edge has met a giant component: from now on this context
        // hooks every 16th edge first. (Measured, walks per edge: config 3 0.013, config 2 0.035, config 4 -- d = 2,
        // millions of clusters of 3-4 keys whose neighbours hook first -- 0.16, where the two launches cost 0.03 ms
        // and gain nothing; the skewed model with its 65 536-key component 0.27-0.29, where they gain 0.7 ms.)
        if (!c->uf_sampled && c->E >= 65536 && both[C64_UF_AGAIN - C64_ROOTS] > c->E / 5)
            c->uf_sampled = true;
        if (getenv("FQD_DEBUG"))
            fprintf(stderr, "[fqd] kept list by id bins: U=%llu base=%llu window=%llu shift=%u kept=%llu listed=%llu; union-find: %llu second walks for %llu edges, sampled=%d\n",
                    (unsigned long long)U, (unsigned long long)base, (unsigned long long)window, shift,
                    (unsigned long long)c->n_kept, (unsigned long long)c->n_listed,
                    (unsigned long long)both[C64_UF_AGAIN - C64_ROOTS], (unsigned long long)c->E, (int)c->uf_sampled);
        return FQD_OK;
    }
    if (by_map) {
        HIP_TRY(c, c->stage_c.reserve(window + 16));
        if (window)
            HIP_TRY(c, hipMemsetAsync(c->stage_c.p, 0, window, c->st));
        KTIME(c, FQD_K_KEPT_FLAGS, fqd::launch_kept_flags(method, c->labels.as<uint32_t>(), c->best.as<uint32_t>(),
                                          c->state.as<uint8_t>(), c->ufirst.as<uint64_t>(), c->id_lo, c->id_hi, U,
                                          c->kept.as<uint8_t>(), nullptr, c->stage_c.as<uint8_t>(), window,
                                          c->ucounts.as<uint32_t>(), c->labels.as<uint32_t>(),
                                          c->root_taint.as<uint8_t>(),
                                          c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
        // the list goes straight into the caller's buffer when one was announced and is large enough
        // for any outcome (fqd_set_kept_output): no copy, no extra round trip afterwards
        c->kept_in_out = c->kept_out && c->kept_out_cap >= std::min(window, U);
        uint64_t *list_out = c->kept_out;
        if (!c->kept_in_out) {
            HIP_TRY(c, c->kept_ids_sorted.reserve(std::min(window, U) * 8 + 16));
            list_out = c->kept_ids_sorted.as<uint64_t>();
        }
        const uint32_t blocks = fqd::window_blocks(window);
        uint32_t listed = 0;
        if (blocks) {
            HIP_TRY(c, c->kept_u32.reserve((size_t)blocks * 4 + 16));
            HIP_TRY(c, c->kept_scan.reserve((size_t)blocks * 4 + 16));
            HIP_TRY(c, fqd::launch_window_count(c->stage_c.as<uint8_t>(), window, c->kept_u32.as<uint32_t>(), c->st));
            FQD_TRY(scan_u32(c, c->kept_u32.as<uint32_t>(), c->kept_scan.as<uint32_t>(), blocks));
            HIP_TRY(c, fqd::launch_window_emit(c->stage_c.as<uint8_t>(), window, c->kept_scan.as<uint32_t>(), base,
                                               list_out, c->st));
            FQD_TRY(queue_read_u32(c, c->kept_scan.as<uint32_t>() + (blocks - 1), 0));
        }
        unsigned long long both[2] = {0, 0};      // C64_ROOTS, C64_SUM: one read for fqd_cluster
        FQD_TRY(read_ctr64(c, C64_ROOTS, both, 2));
        if (blocks)
            listed = taken_u32(c, 0);
        c->roots_seen = both[0];
        const unsigned long long total = both[1];
        c->n_kept = total;
        c->n_listed = listed;
        if (getenv("FQD_DEBUG"))
            fprintf(stderr, "[fqd] kept list by map: U=%llu base=%llu window=%llu id_limit=%llu kept=%llu listed=%u\n",
                    (unsigned long long)U, (unsigned long long)base, (unsigned long long)window,
                    (unsigned long long)c->id_limit, total, listed);
        return FQD_OK;
    }
    KTIME(c, FQD_K_KEPT_FLAGS, fqd::launch_kept_flags(method, c->labels.as<uint32_t>(), c->best.as<uint32_t>(),
                                      c->state.as<uint8_t>(), c->ufirst.as<uint64_t>(), c->id_lo, c->id_hi, U,
                                      c->kept.as<uint8_t>(), c->kept_u32.as<uint32_t>(), nullptr, 0,
                                      c->ucounts.as<uint32_t>(), c->labels.as<uint32_t>(), c->root_taint.as<uint8_t>(),
                                      c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
    c->kept_in_out = false;
    FQD_TRY(scan_u32(c, c->kept_u32.as<uint32_t>(), c->kept_scan.as<uint32_t>(), U));
    uint32_t nk = 0;
    FQD_TRY(queue_read_u32(c, c->kept_scan.as<uint32_t>() + (U - 1), 0));
    unsigned long long both[2] = {0, 0};
    FQD_TRY(read_ctr64(c, C64_ROOTS, both, 2));
    nk = taken_u32(c, 0);
    c->roots_seen = both[0];
    const unsigned long long total = both[1];
    c->n_kept = total;
    c->n_listed = nk;
    HIP_TRY(c, c->kept_ids.reserve((size_t)nk * 8 + 16));
    HIP_TRY(c, c->kept_ids_sorted.reserve((size_t)nk * 8 + 16));
    HIP_TRY(c, fqd::launch_gather_kept(c->kept_u32.as<uint32_t>(), c->kept_scan.as<uint32_t>(),
                                       c->ufirst.as<uint64_t>(), U, c->kept_ids.as<uint64_t>(), c->st));
    if (nk) {
        int sort_bits = c->id_bits;   // listed ids lie below id_hi: fewer radix passes
        if (c->id_hi != ~0ull) {
            int wb = 1;
            while (wb < 64 && (c->id_hi >> wb))
                wb++;
            sort_bits = std::min(sort_bits, wb);
        }
        const size_t need = fqd::sort_keys_u64_temp(nk);
        HIP_TRY(c, c->tmp.reserve(need + 16));
        HIP_TRY(c, fqd::sort_keys_u64(c->tmp.p, need, c->kept_ids.as<uint64_t>(),
                                      c->kept_ids_sorted.as<uint64_t>(), nk, sort_bits, c->st));
    }
    return FQD_OK;
}

int fqd_dissect(fqd_ctx *c, int method, uint64_t *n_kept)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_LABELS)
        return fail(c, FQD_E_STATE, "fqd_dissect before fqd_components");
    if (method < 0 || method > 2)
        return fail(c, FQD_E_VALUE, "unknown cluster dissection method");
    c->stage = ST_LABELS;
    const uint64_t U = c->U, E = c->E;
    const KeyShape sh = c->ks;
    StageTimer timer(c, FQD_T_DISSECT);
    HIP_TRY(c, c->best.reserve(U * 4 + 16));
    HIP_TRY(c, c->state.reserve(U + 16));
    HIP_TRY(c, c->kept.reserve(U + 16));
    HIP_TRY(c, c->kept_u32.reserve(U * 4 + 16));
    HIP_TRY(c, c->kept_scan.reserve(U * 4 + 16));
    const bool pre = c->pre_init, pre_closed = c->pre_init && c->pre_init_closed;
    c->pre_init = c->pre_init_closed = false;          // (one job's worth)
    const bool pass1_done = c->pass1_done && pre_closed;      // (components_queue did pass 1 of THIS dissection, on node records)
    c->pass1_done = false;
    if (!pre)
        HIP_TRY(c, fqd::launch_dissect_init(c->best.as<uint32_t>(), c->state.as<uint8_t>(), U, c->st));
    const bool closed_form = method == FQD_METHOD_DIRECTIONAL && c->collapsed && !getenv("FQD_DIRECTIONAL_ROUNDS") && E;
    if (pre_closed && !closed_form)      // (the set-up launch wrote count nibbles into the state bytes: not for these)
        HIP_TRY(c, hipMemsetAsync(c->state.p, 0, U, c->st));
    uint32_t *d_changed = c->d_ctr32.as<uint32_t>() + C_CHANGED;
    int list_method = method;      // how list_kept reads the verdicts
    if (method == FQD_METHOD_HIGHEST_COUNT) {
        FQD_TRY(ensure_flat_labels(c));
        HIP_TRY(c, fqd::launch_highest_count(c->labels.as<uint32_t>(), c->ucounts.as<uint32_t>(),
                                             c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), sh, U,
                                             c->best.as<uint32_t>(), c->st));
    } else if (method == FQD_METHOD_DIRECTIONAL && c->collapsed && !getenv("FQD_DIRECTIONAL_ROUNDS")) {
        // closed form (graph.hip): two passes over the edges, no rounds, no host round trips. It
        // relies on a strict order of the keys, so a caller's list with repeated keys
        // (fqd_import_unique) takes the relaxation rounds below.
        // (the sets of count-1 keys are whole components: their parents are the components' -- c->labels)
        HIP_TRY(c, c->taint.reserve(E * 8 + 16));       // here: the edges between count-1 keys (edge indices), and behind them those pass 2 looks at
        HIP_TRY(c, c->stage_a.reserve(E * 8 + 16));     // ... and the roots of their ends
        HIP_TRY(c, c->root_taint.reserve(U + 16));
        if (E) {
            if (E >= 0xFFFFFFFFull)
                return fail(c, FQD_E_VALUE, "more than 2^32 edges");
            if (!c->pre_zero_tail)
                FQD_TRY(zero_ctr64(c, C64_CANDS, 2));      // (the search's candidate counters, free here: the two lists' lengths)
            if (!pre_closed) {
                HIP_TRY(c, hipMemsetAsync(c->root_taint.p, 0, U, c->st));
                // (the state byte of this dissection starts as the key's count nibble, graph.hip dstate_init)
                HIP_TRY(c, fqd::launch_dstate_init(c->state.as<uint8_t>(), c->ucounts.as<uint32_t>(), U, c->st));
            }
            // (pass 1b -- graph.hip directional_unions_kernel -- from distance 2 on, or when the edges came from elsewhere)
            const bool split_unions = (c->last_search_d >= 2 || getenv("FQD_DIRECTIONAL_SPLIT_UNIONS")) &&
                                      !getenv("FQD_DIRECTIONAL_NO_SPLIT_UNIONS");
            for (int pass = pass1_done ? 2 : 1; pass <= 2; pass++) {
                if (pass == 2 && c->join_pending) {
                    // pass 2 walks the components' parents: the union-find that ran beside pass 1 must be through
                    HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join, 0));
                    c->join_pending = false;
                }
                KTIME(c, FQD_K_DISSECT_ROUND, fqd::launch_directional_closed(
                          c->edges.as<uint32_t>(), E, c->ucounts.as<uint32_t>(), c->urecs.as<uint32_t>(),
                          c->ulens.as<uint32_t>(), shape_with_row_lengths(c), c->labels.as<uint32_t>(), c->state.as<uint8_t>(),
                          c->taint.as<uint32_t>(), c->d_ctr64.as<unsigned long long>() + C64_CANDS,
                          c->root_taint.as<uint8_t>(), c->best.as<uint32_t>(), pass, c->st, c->stage_a.as<uint32_t>(),
                          split_unions ? c->taint.as<uint32_t>() + E : nullptr,
                          c->d_ctr64.as<unsigned long long>() + C64_CAND_NEED));
            }
            list_method = 3;
        }
    } else if (method == FQD_METHOD_DIRECTIONAL) {
        HIP_TRY(c, c->blocked.reserve(U * 4 + 16));   // here: round stamps of the nodes
        if (E)
            HIP_TRY(c, hipMemsetAsync(c->blocked.p, 0, U * 4, c->st));
        // two sweeps per host check: the flag is read back half as often, and a sweep over edges
        // whose ends did not move is cheap (stamps)
        for (uint64_t round = 1; E && round <= U + 2; round += 2) {
            FQD_TRY(zero_ctr32(c, C_CHANGED));
            for (uint32_t k = 0; k < 2; k++)
                KTIME(c, FQD_K_DISSECT_ROUND, fqd::launch_directional_round(c->edges.as<uint32_t>(), E, c->ucounts.as<uint32_t>(),
                                                     c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), sh,
                                                     c->best.as<uint32_t>(), c->blocked.as<uint32_t>(),
                                                     (uint32_t)(round + k), d_changed, c->st));
            uint32_t changed = 0;
            FQD_TRY(read_ctr32(c, C_CHANGED, &changed));
            if (!changed)
                break;
        }
    } else {
        HIP_TRY(c, c->blocked.reserve(U * 4 + 16));
        HIP_TRY(c, hipMemsetAsync(c->blocked.p, 0, U * 4, c->st));
        // edges become (higher rank, lower rank); union-find and the other methods do not care
        HIP_TRY(c, fqd::launch_orient_edges(c->edges.as<uint32_t>(), E, c->ucounts.as<uint32_t>(),
                                            c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), shape_with_row_lengths(c), c->st));
        // After the first two sweeps only the edges that can still matter are swept (their list is
        // made while the host waits for the "anything changed?" flag of those sweeps).
        const uint32_t *sweep_edges = c->edges.as<uint32_t>();
        uint64_t sweep_E = E;
        const bool shrink = E >= 65536 && !getenv("FQD_ADJACENCY_ALL_EDGES");
        for (uint64_t round = 1; round <= U + 2; round += 2) {
            FQD_TRY(zero_ctr32(c, C_CHANGED));
            for (uint32_t k = 0; k < 2; k++)
                KTIME(c, FQD_K_DISSECT_ROUND, fqd::launch_adjacency_round(sweep_edges, sweep_E, U, c->state.as<uint8_t>(),
                                                   c->blocked.as<uint32_t>(), (uint32_t)(round + k), d_changed, c->st));
            uint32_t changed = 0;
            if (round == 1 && shrink) {
                HIP_TRY(c, c->taint.reserve(E * 8 + 16));        // here: the live edges
                FQD_TRY(zero_ctr64(c, C64_CANDS));               // (the search's candidate counter, free here)
                FQD_TRY(queue_read_u32(c, c->d_ctr32.as<uint32_t>() + C_CHANGED, 0));
                FQD_TRY(queued_reads_mark(c));
                HIP_TRY(c, fqd::launch_adjacency_live_edges(c->edges.as<uint32_t>(), E, c->state.as<uint8_t>(),
                                                            c->taint.as<uint32_t>(),
                                                            c->d_ctr64.as<unsigned long long>() + C64_CANDS, c->st));
                FQD_TRY(queued_reads_wait(c));
                changed = taken_u32(c, 0);
                if (changed) {
                    unsigned long long live = 0;
                    FQD_TRY(read_ctr64(c, C64_CANDS, &live));
                    sweep_edges = c->taint.as<uint32_t>();
                    sweep_E = live;
                }
            } else {
                FQD_TRY(read_ctr32(c, C_CHANGED, &changed));
            }
            if (!changed)
                break;
        }
    }
    if (c->join_pending) {           // the components ran beside all this on the second stream
        HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join, 0));
        c->join_pending = false;
    }
    if (c->drop_after_n) {
        // (fqd_dissect_except: keys whose cluster was dissected elsewhere stood alone in this graph)
        FQD_TRY(zero_ctr32(c, C_BAD));
        HIP_TRY(c, fqd::launch_mark_dropped_after(c->state.as<uint8_t>(), c->best.as<uint32_t>(), U, c->drop_after,
                                                  c->drop_after_n, c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
    }
    FQD_TRY(list_kept(c, list_method));
    if (c->drop_after_n) {
        uint32_t bad = 0;
        FQD_TRY(read_ctr32(c, C_BAD, &bad));
        if (bad)
            return fail(c, FQD_E_VALUE, "dropped row outside the unique table");
    }
    timer.stop();
    c->stage = ST_KEPT;
    if (n_kept)
        *n_kept = c->n_kept;
    return FQD_OK;
}

// fqd_dissect for a rank of a sharded job (sharded.py, home clusters): the edges in the context are those of the
// clusters that live on this rank alone; `dropped` (device) lists the rows that the dissection of the OTHER clusters
// -- done elsewhere -- dropped. Those rows have no edge here, so they are kept by this dissection and then overruled.
int fqd_dissect_except(fqd_ctx *c, int method, const uint32_t *dropped, uint64_t n_dropped, int mem, uint64_t *n_kept)
{
    if (mem != FQD_DEVICE && n_dropped)
        return fail(c, FQD_E_VALUE, "fqd_dissect_except works on device buffers");
    c->drop_after = dropped;
    c->drop_after_n = n_dropped;
    const int rc = fqd_dissect(c, method, n_kept);
    c->drop_after = nullptr;
    c->drop_after_n = 0;
    return rc;
}

int fqd_set_id_window(fqd_ctx *c, uint64_t lo, uint64_t hi)
{
    c->id_lo = lo;
    c->id_hi = hi;
    return FQD_OK;
}

int fqd_get_kept_count(fqd_ctx *c, uint64_t *n_kept, uint64_t *n_listed)
{
    if (c->stage < ST_KEPT)
        return fail(c, FQD_E_STATE, "no dissection result yet");
    if (n_kept)
        *n_kept = c->n_kept;
    if (n_listed)
        *n_listed = c->n_listed;
    return FQD_OK;
}

int fqd_get_kept_read_ids(fqd_ctx *c, uint64_t *out, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_KEPT)
        return fail(c, FQD_E_STATE, "no dissection result yet");
    if (c->kept_list_lost)
        return fail(c, FQD_E_STATE, "the kept list was written to a caller's buffer that has been withdrawn");
    if (c->kept_in_out) {
        if (out == c->kept_out && mem == FQD_DEVICE)
            return FQD_OK;            // already there
        return from_device(c, out, c->kept_out, (size_t)c->n_listed, mem);
    }
    return from_device(c, out, c->kept_ids_sorted.p, (size_t)c->n_listed, mem);
}

int fqd_set_kept_output(fqd_ctx *c, uint64_t *out_device, uint64_t capacity)
{
    c->kept_out = out_device;
    c->kept_out_cap = out_device ? capacity : 0;
    if (!out_device && c->kept_in_out) {
        c->kept_in_out = false;      // the list lived in the buffer we are letting go of
        c->kept_list_lost = true;
    }
    return FQD_OK;
}

int fqd_get_unique_table(fqd_ctx *c, uint64_t *first_ids, uint32_t *counts, uint32_t *labels, uint8_t *kept, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "no unique table yet");
    FQD_TRY(from_device(c, first_ids, c->ufirst.p, (size_t)c->U, mem));
    FQD_TRY(from_device(c, counts, c->ucounts.p, (size_t)c->U, mem));
    if (labels) {
        if (c->stage < ST_LABELS)
            return fail(c, FQD_E_STATE, "no component labels yet");
        FQD_TRY(ensure_flat_labels(c));
        FQD_TRY(from_device(c, labels, c->labels.p, (size_t)c->U, mem));
    }
    if (kept) {
        if (c->stage < ST_KEPT)
            return fail(c, FQD_E_STATE, "no dissection result yet");
        FQD_TRY(from_device(c, kept, c->kept.p, (size_t)c->U, mem));
    }
    return FQD_OK;
}

// Union-find over n_nodes nodes and a caller's edge list (device): roots[e] = smallest node of
// edge e's component; *n_components = n_nodes - successful hooks.
int fqd_edge_labels(fqd_ctx *c, const uint32_t *uv, uint64_t E, uint64_t n_nodes, uint32_t *roots,
                    uint64_t *n_components, int mem)
{
    FQD_TRY(bind(c));
    if (mem != FQD_DEVICE)
        return fail(c, FQD_E_VALUE, "fqd_edge_labels works on device buffers");
    if (n_nodes >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "at most 2^32-16 nodes");
    if ((uintptr_t)uv & 7u)
        return fail(c, FQD_E_VALUE, "the edge list must be 8-byte aligned");
    FQD_TRY(zero_ctr32(c, C_BAD));
    HIP_TRY(c, fqd::launch_check_indices(uv, 2 * E, n_nodes, c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
    uint32_t bad = 0;
    FQD_TRY(read_ctr32(c, C_BAD, &bad));
    if (bad)
        return fail(c, FQD_E_VALUE, "edge end outside [0, n_nodes)");
    HIP_TRY(c, c->stage_a.reserve(n_nodes * 4 + 16));
    HIP_TRY(c, c->stage_b.reserve(FQD_HOOK_SLOTS * 64));
    uint32_t *parent = c->stage_a.as<uint32_t>();
    HIP_TRY(c, hipMemsetAsync(c->stage_b.p, 0, FQD_HOOK_SLOTS * 64, c->st));
    HIP_TRY(c, fqd::launch_uf_init(parent, n_nodes, c->st));
    HIP_TRY(c, fqd::launch_uf_union(parent, uv, E, c->stage_b.as<unsigned long long>(), c->st));
    HIP_TRY(c, fqd::launch_edge_roots(parent, uv, E, roots, c->st));
    // (the slots added up on the device: one 8-byte read instead of FQD_HOOK_SLOTS x 64 bytes into pageable memory)
    HIP_TRY(c, fqd::launch_hook_total(c->stage_b.as<unsigned long long>(), n_nodes,
                                      c->d_ctr64.as<unsigned long long>() + C64_ROOTS, c->st));
    unsigned long long comps = 0;
    FQD_TRY(read_ctr64(c, C64_ROOTS, &comps));
    if (n_components)
        *n_components = comps;
    return FQD_OK;
}

// The clusters a rank dissects, cut out of the job-wide edge list: edges whose root (fqd_edge_labels)
// is congruent to `part` modulo n_parts, their distinct ends in ascending order (touched_out), and
// the edges again with every end replaced by its position in touched_out. DEVICE buffers:
// touched_out min(2 E, n_nodes) words, sub_edges_out 2 E words. Replaces a mask + unique + inverse
// chain on the caller's side (three sorts' worth of work for 3 M ends).
int fqd_cluster_subgraph(fqd_ctx *c, const uint32_t *uv, const uint32_t *roots, uint64_t E, uint64_t n_nodes,
                         uint32_t n_parts, uint32_t part, uint32_t *touched_out, uint32_t *sub_edges_out,
                         uint64_t *n_touched, uint64_t *n_sub, int mem)
{
    FQD_TRY(bind(c));
    if (mem != FQD_DEVICE)
        return fail(c, FQD_E_VALUE, "fqd_cluster_subgraph works on device buffers");
    if (!n_parts || part >= n_parts || n_nodes >= 0xFFFFFFF0ull)
        return fail(c, FQD_E_VALUE, "fqd_cluster_subgraph: bad arguments");
    if (n_touched)
        *n_touched = 0;
    if (n_sub)
        *n_sub = 0;
    if (!E || !n_nodes)
        return FQD_OK;
    HIP_TRY(c, c->stage_a.reserve(n_nodes * 4 + 16));       // flags
    HIP_TRY(c, c->stage_b.reserve(n_nodes * 4 + 16));       // their inclusive scan
    HIP_TRY(c, hipMemsetAsync(c->stage_a.p, 0, n_nodes * 4, c->st));
    FQD_TRY(zero_ctr64(c, C64_SUM));
    unsigned long long *d_n = c->d_ctr64.as<unsigned long long>() + C64_SUM;
    HIP_TRY(c, fqd::launch_subgraph_mark(uv, roots, E, n_parts, part, c->stage_a.as<uint32_t>(), sub_edges_out, d_n, c->st));
    FQD_TRY(scan_u32(c, c->stage_a.as<uint32_t>(), c->stage_b.as<uint32_t>(), n_nodes));
    HIP_TRY(c, fqd::launch_subgraph_finish(c->stage_a.as<uint32_t>(), c->stage_b.as<uint32_t>(), n_nodes, E, touched_out,
                                           sub_edges_out, d_n, c->st));
    FQD_TRY(queue_read_u32(c, c->stage_b.as<uint32_t>() + (n_nodes - 1), 0));
    unsigned long long ns = 0;
    FQD_TRY(read_ctr64(c, C64_SUM, &ns));
    if (n_touched)
        *n_touched = taken_u32(c, 0);
    if (n_sub)
        *n_sub = ns;
    return FQD_OK;
}

// fqd_cluster_subgraph with HOME clusters apart (graph.hip): uid_bounds[r] .. uid_bounds[r + 1] is rank r's range of the
// job-wide key numbering (n_parts + 1 host values, n_parts <= FQD_MAX_HOME_RANKS). home_edges_out (device, 2 E words):
// the edges of the clusters that live on this rank alone, ends as rows of ITS table; touched_out / sub_edges_out: this
// rank's share of the clusters that span ranks.
int fqd_cluster_subgraph_home(fqd_ctx *c, const uint32_t *uv, const uint32_t *roots, uint64_t E, uint64_t n_nodes,
                              uint32_t n_parts, uint32_t part, const uint64_t *uid_bounds, uint32_t *touched_out,
                              uint32_t *sub_edges_out, uint32_t *home_edges_out, uint64_t *n_touched, uint64_t *n_sub,
                              uint64_t *n_home, uint64_t *n_spanning_edges, int mem)
{
    FQD_TRY(bind(c));
    if (mem != FQD_DEVICE)
        return fail(c, FQD_E_VALUE, "fqd_cluster_subgraph_home works on device buffers");
    if (!n_parts || part >= n_parts || n_parts > FQD_MAX_HOME_RANKS || n_nodes >= 0xFFFFFFF0ull || !uid_bounds ||
        ((uintptr_t)uv & 7u))
        return fail(c, FQD_E_VALUE, "fqd_cluster_subgraph_home: bad arguments");
    fqd::UidBounds bounds{};
    bounds.n = n_parts;
    for (uint32_t r = 0; r <= n_parts; r++) {
        if (uid_bounds[r] > n_nodes || (r && uid_bounds[r] < uid_bounds[r - 1]))
            return fail(c, FQD_E_VALUE, "fqd_cluster_subgraph_home: bad key ranges");
        bounds.lo[r] = (uint32_t)uid_bounds[r];
    }
    if (n_touched)
        *n_touched = 0;
    if (n_sub)
        *n_sub = 0;
    if (n_home)
        *n_home = 0;
    if (n_spanning_edges)
        *n_spanning_edges = 0;
    if (!E || !n_nodes)
        return FQD_OK;
    HIP_TRY(c, c->stage_a.reserve(n_nodes * 4 + 16));       // flags
    HIP_TRY(c, c->stage_b.reserve(n_nodes * 4 + 16));       // their inclusive scan
    HIP_TRY(c, c->span.reserve(n_nodes + 16));
    HIP_TRY(c, hipMemsetAsync(c->stage_a.p, 0, n_nodes * 4, c->st));
    HIP_TRY(c, hipMemsetAsync(c->span.p, 0, n_nodes, c->st));
    FQD_TRY(zero_ctr64(c, C64_SUM, 3));     // C64_SUM: edges of my share, + 1: home edges, + 2: edges of spanning clusters
    unsigned long long *d_n = c->d_ctr64.as<unsigned long long>() + C64_SUM;
    HIP_TRY(c, fqd::launch_subgraph_mark_home(uv, roots, E, n_parts, part, bounds, c->span.as<uint8_t>(),
                                              c->stage_a.as<uint32_t>(), sub_edges_out, d_n, home_edges_out, d_n + 1, d_n + 2,
                                              c->st));
    FQD_TRY(scan_u32(c, c->stage_a.as<uint32_t>(), c->stage_b.as<uint32_t>(), n_nodes));
    HIP_TRY(c, fqd::launch_subgraph_finish(c->stage_a.as<uint32_t>(), c->stage_b.as<uint32_t>(), n_nodes, E, touched_out,
                                           sub_edges_out, d_n, c->st));
    FQD_TRY(queue_read_u32(c, c->stage_b.as<uint32_t>() + (n_nodes - 1), 0));
    unsigned long long both[3] = {0, 0, 0};
    FQD_TRY(read_ctr64(c, C64_SUM, both, 3));
    if (n_touched)
        *n_touched = taken_u32(c, 0);
    if (n_sub)
        *n_sub = both[0];
    if (n_home)
        *n_home = both[1];
    if (n_spanning_edges)
        *n_spanning_edges = both[2];
    return FQD_OK;
}

// The dissection's verdicts came from elsewhere (the rank that held the cluster): every key of
// the unique table is kept except the listed rows. Fills the kept list like fqd_dissect.
int fqd_list_kept_except(fqd_ctx *c, const uint32_t *dropped, uint64_t n_dropped, int mem, uint64_t *n_kept)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_UNIQUE)
        return fail(c, FQD_E_STATE, "no unique table yet");
    if (mem != FQD_DEVICE && n_dropped)
        return fail(c, FQD_E_VALUE, "fqd_list_kept_except works on device buffers");
    const uint64_t U = c->U;
    StageTimer timer(c, FQD_T_DISSECT);
    HIP_TRY(c, c->state.reserve(U + 16));
    HIP_TRY(c, c->kept.reserve(U + 16));
    HIP_TRY(c, c->kept_u32.reserve(U * 4 + 16));
    HIP_TRY(c, c->kept_scan.reserve(U * 4 + 16));
    if (U)
        HIP_TRY(c, hipMemsetAsync(c->state.p, 1, U, c->st));
    FQD_TRY(zero_ctr32(c, C_BAD));
    HIP_TRY(c, fqd::launch_mark_dropped(c->state.as<uint8_t>(), U, dropped, n_dropped,
                                        c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
    uint32_t bad = 0;
    FQD_TRY(read_ctr32(c, C_BAD, &bad));
    if (bad)
        return fail(c, FQD_E_VALUE, "dropped row outside the unique table");
    FQD_TRY(list_kept(c, FQD_METHOD_ADJACENCY));   // "state == 1" is the verdict
    timer.stop();
    c->stage = ST_KEPT;
    if (n_kept)
        *n_kept = c->n_kept;
    return FQD_OK;
}

}  // extern "C"

int fqd_api_components_queue(fqd_ctx *c, bool flatten, bool on_side) { return components_queue(c, flatten, on_side); }
int fqd_api_flat_labels(fqd_ctx *c) { return ensure_flat_labels(c); }
int fqd_api_graph_preinit(fqd_ctx *c, int method) { return graph_preinit(c, method); }
