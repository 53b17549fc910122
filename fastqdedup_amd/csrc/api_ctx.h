// api_ctx.h -- PRIVATE to the api_*.hip files: the context behind the C ABI of
// libfqdedup_hip.so (include/fqdedup_hip.h), its device buffers, and the small helpers every
// entry point uses (error strings, host<->device staging, counter read-backs, HIP-event timing).
// No kernels here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fqdedup_hip.h"
#include "fqd_internal.h"

namespace {

std::string g_global_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;   // p belongs to the caller (FQD_DEVICE_BORROW): never freed, never reused
    void *own_p = nullptr;   // the context's own allocation, parked while p is borrowed (no
    size_t own_cap = 0;      // hipFree/hipMalloc per job: a job that borrows every time would churn)
    void unborrow()
    {
        if (borrowed) {
            p = own_p;
            cap = own_cap;
            own_p = nullptr;
            own_cap = 0;
            borrowed = false;
        }
    }
    hipError_t reserve(size_t bytes)
    {
        unborrow();
        if (bytes <= cap)
            return hipSuccess;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = ((bytes + (bytes >> 4)) + 4095) & ~(size_t)4095;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    void release()
    {
        unborrow();
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // Use the caller's device buffer in place (no copy); the next reserve() lets go of it.
    void borrow(const void *ptr, size_t bytes)
    {
        if (!borrowed) {
            own_p = p;
            own_cap = cap;
        }
        p = const_cast<void *>(ptr);
        cap = bytes;
        borrowed = true;
    }
    template <typename T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

enum Stage { ST_EMPTY = 0, ST_PACKED = 1, ST_UNIQUE = 2, ST_EDGES = 3, ST_LABELS = 4, ST_KEPT = 5 };

// small device-side words read back by the host
enum Ctr { C_BAD = 0, C_COLLISIONS = 1, C_CHANGED = 2, C_MINLEN = 3, C_MAXLEN = 4, C_PACKBAD = 5, C_SIDE = 6, C_P0 = 7, C_N32 = 8 };
enum Ctr64 { C64_EDGES = 0, C64_ROOTS = 1, C64_SUM = 2, C64_STATS = 3, C64_CANDS = 4, C64_CAND_NEED = 5, C64_SLAB = 6, C64_UF_AGAIN = 7, C64_N = 8 };

}  // namespace

struct fqd_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    int stage = ST_EMPTY;

    bool forced = false;
    uint8_t forced_present[128];
    uint32_t forced_max_len = 0;
    int forced_ragged = 0;

    fqd_shape shape{};
    KeyShape ks{};
    DevBuf d_lut, d_ctr32, d_ctr64, d_present, d_stats;
    uint8_t lut_on_device[256];    // what d_lut holds (a pageable H2D copy per pack call is a host stall)
    bool lut_valid = false;
    uint64_t *kept_out = nullptr;  // fqd_set_kept_output: the kept-id list is written here directly
    uint64_t kept_out_cap = 0;
    bool kept_in_out = false;      // the current kept list lives in kept_out, not in kept_ids_sorted
    bool kept_list_lost = false;   // ... and kept_out has been withdrawn since

    // stage 1
    uint64_t n = 0;
    DevBuf in_bytes, in_offsets, recs, lens, hashes, owners;
    bool compact_off = false;      // fqd_cluster_keys: the side slabs of the compact records overflowed once (many keys with an N) -- uint4 records from now on
    bool fused_off = false;        // fqd_cluster_keys: a level-1 slab of the fused pack overflowed once -- plain pack from now on
    bool urecs_len_pad = false;    // ... and the unique table's rows still do (the long-key collapse copies whole rows): the search's
                                   // verification and the graph's key comparisons take lengths from the rows they fetch anyway
    bool recs_len_pad = false;     // ragged records whose last padding word holds the key's length (fqd_pack_keys; see pack.hip)
    bool no_len_pad = false;       // (the store: its rows come from jobs of either kind and are compared with lens[])
    uint32_t modal_len_hint = 0;   // ... and a likely length of such keys: (shortest + longest + 1) / 2
    bool recs_valid = false;       // c->recs holds the packed reads in read order (not after the fused pack)
    bool pairs_slab_off = false;   // long-record collapse: same, for its (hash, position) partition
    // crowded buckets of a distance-1 search (group.hip): flags + list + counters, the fine items, the seen marks
    DevBuf gp_crowded, gp_fine_hash, gp_fine_val, gp_seen;
    const uint32_t *gp_last_items = nullptr;   // the partitioned items of the last grouped pass
    uint32_t gp_crowded_bits_last = 0;
    uint32_t gp_crowded_bits = 0;      // != 0: the last grouped pass marked crowded buckets (2^bits buckets) and skipped them
    bool search_is_retry = false;       // find_edges calling itself after pass 0's pairs were lost: the route bits stay
    bool search_force_sort = false;     // ... or after the crowded-bucket refinement gave up: that run takes the sort path
    bool search_keeps_edges = false;    // ... except the edge counter and the statistics: pass 0 of this search has run (fqd::Pass0)
    bool search_zero_pending = false;   // find_edges: the job counters and statistics are zeroed by the partition's first launch
    uint64_t pairs_last_U = 0;     // collapse_pairs: the unique keys of the context's last job (the compaction is queued before this job's are known)
    DevBuf pairs_slices;           // collapse_pairs: the slices of very long buckets (fqd::PairsSlices)
    bool gp_fine_ok = false;       // the last grouped pass: its crowded keys can be matched on finer pieces
    bool gp_fine_used = false;     // the last grouped_refine filed fine items (its candidates count towards the budget)
    bool gp_tiles = false;         // grouped search: crowded buckets go all pairs in tiles (the finer pieces did not split them)
    bool gp_slab_off = false;      // grouped search: same, for the (hash, uid) partition
    bool slab_off = false;         // LDS collapse: a slab of level 2 overflowed once, use exact bucket sizes
    bool hashes_valid = false;     // `hashes` holds the record hashes of the packed reads (lazy after an import)
    fqd::OwnerRule owner_rule;     // fqd_set_owner_rule: fqd_pack_keys also writes each read's owner rank
    fqd::OwnerRule owners_done;    // the rule `owners` was filled with (parts == 0: not filled)
    // stage 2
    uint64_t U = 0, n_counted = 0;
    int id_bits = 64;  // bits needed to sort first-holder ids (read ids 0..n-1 need few)
    uint64_t id_limit = ~0ull;  // every first-holder id is below this (~0: unknown)
    bool collapsed = false;  // unique table came from fqd_collapse (keys are pairwise distinct)
    bool first_distinct = true;  // first-holder ids are pairwise distinct (false: imported without ids)
    DevBuf seg_tab;   // fqd_collapse_received: id bases and row offsets of the senders' segments
    DevBuf in_weights, in_read_ids, hs_sorted, ids, ids_sorted, flags, run_idx, run_start, run_weight, live_flag,
        live_idx, collision_runs;
    DevBuf urecs, ulens, ucounts, ufirst;
    DevBuf ld_small, ld_part2, ld_matrix, ld_matrix_incl, ld_seg, ld_side, ld_side_table;
    DevBuf ld_hist, ld_hist_incl, ld_start, ld_cursor, ld_part, ld_tmp_rec, ld_tmp_count, ld_tmp_first, ld_unique,
        ld_unique_incl;
    uint32_t route = 0;     // FQD_ROUTE_* bits of the job in progress / the last one (fqd_get_route)
    int collapse_path = 0;  // 1: LDS bucket dedupe, 2: sort + verify (last fqd_collapse)
    // stage 3
    uint64_t E = 0, edge_cap = 0;
    DevBuf seg_hashes, sorted_hash, sorted_uid, uid_iota, edges, sel_hash, sel_uid;
    uint32_t seg_hint = 0;          // fqd_cluster[_keys]: segments of the search that follows the collapse
    uint32_t seg_hashes_nseg = 0;   // != 0: seg_hashes already holds the segment hashes of the unique table, [nseg - first][U]
    uint32_t seg_hashes_first = 0;  // ... from this segment on (1: the collapse did search pass 0 itself)
    // the routed collapse (fqd_cluster_keys, compact records): reads binned by segment 0 of the key, search pass 0
    // inside the compaction (fqd::Pass0). pass0_done: the edge list holds pass 0 of a search with pass0_nseg
    // segments, for the next fqd_find_edges to continue; route_off: a bucket's LDS table overflowed under the
    // routing (many keys share a segment-0 value) -- this context keeps to whole-key hashing.
    bool pass0_done = false, route_off = false;
    bool owner_routed = false;     // fqd_set_owner_routing: the owner slabs are binned by segment 0 on EVERY rank of the job
    uint32_t pass0_nseg = 0;
    uint64_t pass0_edge_cap = 0;    // the edge list's capacity pass 0 wrote against (pairs behind it were counted, not written)
    DevBuf p0_probe;
    DevBuf ld_sync;                 // the words the workgroups of the one-kernel dedupe + compaction meet on (fqd::CollapseSync)
    bool one_kernel_off = false;    // ... a wait of that kernel ran into its limit once (CUs held by another process?): two kernels from now on
    DevBuf ld_huge;                 // the uint4 dedupe's huge-bucket plan (fqd::HugeBuckets)
    DevBuf gp_a, gp_b, gp_small, gp_cands;   // grouped search pass: items after level 1 / level 2, small tables, candidate pairs
    uint64_t gp_cand_cap = 0;
    DevBuf q_table, q_pass, q_means, q_bytes, q_offsets;
    DevBuf len_present, ed_hash, ed_payload, ed_hash_sorted, ed_payload_sorted, ed_cands, ed_cands_sorted, d_alphabet;
    DevBuf eg_tables, eg_per_key, eg_per_key_incl;   // grouped edit search: class tables, probe items per key
    fqd::PairStats last_stats{};
    bool stats_pending = false;   // d_stats holds the slots of the last search, not yet summed into last_stats
    // stage 4
    uint64_t n_clusters = 0, roots_seen = 0;
    DevBuf labels, hook_slots;
    bool labels_flat = false;
    bool pre_zero_tail = false;     // ... and it cleared the kept-bin cursors, C64_SUM and C64_CANDS too (the tail launches no fills)
    bool pre_init = false, pre_init_closed = false;   // fqd_api_graph_preinit ran for this job (its closed-form part too)
    DevBuf nodes;                   // node records (parent, state byte) of the one-sweep components + directional pass 1
    bool pre_nodes = false;         // ... the set-up wrote THEM (not the labels / state arrays): components_queue runs the one sweep
    bool pass1_done = false;        // ... which did pass 1 of the closed-form directional dissection: fqd_dissect starts at pass 2
    int preinit_method = -1;        // >= 0: the search queues fqd_api_graph_preinit(method) behind its read-back
    // stage 5
    uint64_t n_kept = 0, n_listed = 0;          // kept keys; kept keys whose first holder is in the id window
    uint64_t id_lo = 0, id_hi = ~0ull;
    DevBuf best, state, blocked, taint, root_taint, kept, kept_u32, kept_scan, kept_ids, kept_ids_sorted, kept_lists;
    // the reference trie's view of the unique table (api_trie.hip): key order, census, clusters in pop order
    DevBuf t_idx, t_order, t_order_b, t_rank, t_keys, t_keys_b, t_lcp, t_mark, t_mark_incl, t_stats, t_seed, t_heads,
        t_heads_incl, t_member_uids, t_offsets;
    uint64_t t_clusters = 0, t_members = 0;
    bool t_clusters_valid = false;
    // the unique table as a store (fqd_store_*): rows removed since the last merge, ids handed out
    DevBuf store_alive, st_recs, st_lens, st_counts, st_first, st_comb_recs, st_comb_lens, st_comb_w, st_comb_ids;
    uint64_t store_removed = 0, store_table_U = 0, store_next_id = 0;
    // scratch
    DevBuf tmp, stage_a, stage_b, stage_c, stage_d;
    DevBuf span;                   // fqd_cluster_subgraph_home: one byte per node, 1 = its cluster has keys on two ranks
    const uint32_t *drop_after = nullptr;   // fqd_dissect_except: rows whose verdict came from elsewhere (device)
    uint64_t drop_after_n = 0;

    // stage timers: one event pair per stage, recorded while the work is queued and resolved when
    // fqd_stage_times asks (a stage end is NOT a host synchronisation point)
    // the fast paths a context has given up (heavy_keys: spill list, no routing; route_off) are tried again after
    // fast_retry_after jobs that did not need them (api.hip pack_collapse_fused); a retry that fails doubles the wait
    uint32_t clean_jobs = 0, fast_retry_after = 8;
    bool fast_probe = false;
    uint32_t last_spill_used = 0;  // records the last fused job sent to its spill list
    bool heavy_keys = false;       // a fused attempt ended on a full slab: from now on with the spill list (api.hip pack_collapse_fused_once)
    int last_search_d = 0;         // max_distance of the last neighbour search (the closed-form dissection: pass 1b from d = 2 on)
    bool uf_sampled = false;       // the union-find met a giant component on this context: every 16th edge first (graph.hip uf_union_kernel)
    bool join_pending = false;     // the components were queued on st_side: ev_join must be waited for before their counter is read
    hipStream_t st_side = nullptr; // the side path of the compact collapse runs here, beside the dedupe (ev_fork / ev_join order it)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_rb = nullptr;    // marks queued read-backs: the host can wait for THEM while later work runs
    void *h_pin = nullptr;         // 256 pinned host bytes: where counter read-backs land
    void *h_pin_big = nullptr;     // 1 MiB of pinned host memory for small tables (allocated on first use)
    uint32_t h_extra[16] = {0};    // (without pinned memory)
    unsigned long long h_extra64[8] = {0};
    hipEvent_t tev[2 * FQD_T_COUNT] = {nullptr};
    bool tpending[FQD_T_COUNT] = {false};
    bool stage_timing = true;
    // per-kernel timing (fqd_kernel_times): a pool of event pairs for the kernels in ktime_mask,
    // folded into the sums at fqd_kernel_times or when the pool runs low
    uint32_t ktime_mask = 0;       // (fqd_create: all kernels with FQD_KERNEL_TIMERS=1 in the environment; fqd_set_timing)
    static constexpr int KPOOL = 512;
    hipEvent_t kev[2 * KPOOL] = {nullptr};
    int kslot[KPOOL] = {0};
    int kused = 0;
    float kms[FQD_K_COUNT] = {0};
    uint32_t klaunches[FQD_K_COUNT] = {0};
    float ms[FQD_T_COUNT] = {0};
    uint32_t launches[FQD_T_COUNT] = {0};
};

namespace {

int fail(fqd_ctx *c, int code, const std::string &msg)
{
    if (c)
        c->err = msg;
    return code;
}

int hip_fail(fqd_ctx *c, hipError_t e, const char *what)
{
    // clear the sticky error so later calls report their own
    (void)hipGetLastError();
    return fail(c, e == hipErrorOutOfMemory ? FQD_E_NOMEM : FQD_E_DEVICE,
                std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(c, call)                         \
    do {                                         \
        hipError_t e_ = (call);                  \
        if (e_ != hipSuccess)                    \
            return hip_fail((c), e_, #call);     \
    } while (0)

#define FQD_TRY(call)        \
    do {                     \
        int rc_ = (call);    \
        if (rc_ != FQD_OK)   \
            return rc_;      \
    } while (0)

int bind(fqd_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    return FQD_OK;
}

// Returns a device pointer for a caller buffer: the buffer itself (FQD_DEVICE)
// or a staged copy (FQD_HOST).
template <typename T>
int to_device(fqd_ctx *c, const T *src, size_t count, int mem, DevBuf &staging, const T **out)
{
    if (!src) {
        *out = nullptr;
        return FQD_OK;
    }
    if (mem == FQD_DEVICE) {
        *out = src;
        return FQD_OK;
    }
    HIP_TRY(c, staging.reserve(count * sizeof(T) + 16));
    if (count)
        HIP_TRY(c, hipMemcpyAsync(staging.p, src, count * sizeof(T), hipMemcpyHostToDevice, c->st));
    *out = staging.as<T>();
    return FQD_OK;
}

// Wait for the context's stream. (Polling hipStreamQuery instead gained 1.4 % while the counter
// read-backs went through pageable memory and nothing once they were pinned: not worth a spinning core.)
// the key shape as the kernels that can take a row's length from the row itself get it: bit 1 of `ragged` says so, the
// bits from 8 up hold a likely length (see group.hip verify_candidates_kernel, graph.hip key_greater_inflight)
inline KeyShape shape_with_row_lengths(const fqd_ctx *c)
{
    KeyShape sh = c->ks;
    if (sh.ragged && c->urecs_len_pad && !getenv("FQD_NO_LEN_IN_RECORD"))
        sh.ragged = 1u | 2u | (c->modal_len_hint << 8);
    return sh;
}

inline hipError_t stream_wait(hipStream_t st) { return hipStreamSynchronize(st); }

template <typename T>
int from_device(fqd_ctx *c, T *dst, const void *src, size_t count, int mem)
{
    if (!dst || !count)
        return FQD_OK;
    HIP_TRY(c, hipMemcpyAsync(dst, src, count * sizeof(T),
                              mem == FQD_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    return FQD_OK;
}

// Counter read-backs land in a small pinned buffer (no staging copy on the host side).
int read_ctr32n(fqd_ctx *c, int idx, uint32_t *v, int count)
{
    void *dst = c->h_pin ? c->h_pin : (void *)v;
    HIP_TRY(c, hipMemcpyAsync(dst, c->d_ctr32.as<uint32_t>() + idx, 4 * (size_t)count, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    if (dst != (void *)v)
        memcpy(v, dst, 4 * (size_t)count);
    return FQD_OK;
}

int read_ctr32(fqd_ctx *c, int idx, uint32_t *v) { return read_ctr32n(c, idx, v, 1); }

// One more device word to come back with the NEXT read_ctr* (which waits): queued now, taken after
// that wait. (A copy to pageable memory would wait on its own: two round trips instead of one.)
int queue_read_u32(fqd_ctx *c, const uint32_t *dev, int slot)
{
    uint32_t *dst = c->h_pin ? reinterpret_cast<uint32_t *>(static_cast<char *>(c->h_pin) + 192) + slot
                             : &c->h_extra[slot];
    HIP_TRY(c, hipMemcpyAsync(dst, dev, 4, hipMemcpyDeviceToHost, c->st));
    return FQD_OK;
}

// `count` consecutive device words into slots [slot, slot + count) (16 slots)
int queue_read_u32n(fqd_ctx *c, const uint32_t *dev, int count, int slot)
{
    uint32_t *dst = c->h_pin ? reinterpret_cast<uint32_t *>(static_cast<char *>(c->h_pin) + 192) + slot
                             : &c->h_extra[slot];
    HIP_TRY(c, hipMemcpyAsync(dst, dev, 4 * (size_t)count, hipMemcpyDeviceToHost, c->st));
    return FQD_OK;
}

// Wait for the read-backs queued so far -- not for work queued after this call's event.
int queued_reads_mark(fqd_ctx *c)
{
    HIP_TRY(c, hipEventRecord(c->ev_rb, c->st));
    return FQD_OK;
}

int queued_reads_wait(fqd_ctx *c)
{
    HIP_TRY(c, hipEventSynchronize(c->ev_rb));
    return FQD_OK;
}

uint32_t taken_u32(const fqd_ctx *c, int slot)
{
    return c->h_pin ? reinterpret_cast<const uint32_t *>(static_cast<const char *>(c->h_pin) + 192)[slot]
                    : c->h_extra[slot];
}

// 64-bit counters into the first bytes of the pinned buffer, to be taken after queued_reads_wait()
int queue_read_ctr64(fqd_ctx *c, int idx, int count)
{
    void *dst = c->h_pin ? c->h_pin : (void *)c->h_extra64;
    HIP_TRY(c, hipMemcpyAsync(dst, c->d_ctr64.as<unsigned long long>() + idx, 8 * (size_t)count, hipMemcpyDeviceToHost,
                              c->st));
    return FQD_OK;
}

void taken_ctr64(const fqd_ctx *c, unsigned long long *v, int count)
{
    memcpy(v, c->h_pin ? c->h_pin : (const void *)c->h_extra64, 8 * (size_t)count);
}

int read_ctr64(fqd_ctx *c, int idx, unsigned long long *v, int count = 1)
{
    void *dst = c->h_pin ? c->h_pin : (void *)v;
    HIP_TRY(c, hipMemcpyAsync(dst, c->d_ctr64.as<unsigned long long>() + idx, 8 * (size_t)count,
                              hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, stream_wait(c->st));
    if (dst != (void *)v)
        memcpy(v, dst, 8 * (size_t)count);
    return FQD_OK;
}

int zero_ctr32(fqd_ctx *c, int idx, int count = 1)
{
    HIP_TRY(c, hipMemsetAsync(c->d_ctr32.as<uint32_t>() + idx, 0, 4 * (size_t)count, c->st));
    return FQD_OK;
}

int zero_ctr64(fqd_ctx *c, int idx, int count = 1)
{
    HIP_TRY(c, hipMemsetAsync(c->d_ctr64.as<unsigned long long>() + idx, 0, 8 * (size_t)count, c->st));
    return FQD_OK;
}

void ktime_collect(fqd_ctx *c);

void stage_times_resolve(fqd_ctx *c)
{
    bool any = false;
    for (int t = 0; t < FQD_T_COUNT; t++)
        any |= c->tpending[t];
    if (!any)
        return;
    (void)hipStreamSynchronize(c->st);
    for (int t = 0; t < FQD_T_COUNT; t++) {
        if (!c->tpending[t])
            continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->tev[2 * t], c->tev[2 * t + 1]) == hipSuccess)
            c->ms[t] = ms;
        c->launches[t] = 1;
        c->tpending[t] = false;
    }
}

struct StageTimer {
    fqd_ctx *c;
    int slot;
    bool on;
    StageTimer(fqd_ctx *ctx, int s) : c(ctx), slot(s), on(ctx->stage_timing)
    {
        c->tpending[slot] = false;
        if (on)
            (void)hipEventRecord(c->tev[2 * slot], c->st);
        else
            c->ms[slot] = 0, c->launches[slot] = 0;
    }
    void stop()
    {
        if (on) {
            (void)hipEventRecord(c->tev[2 * slot + 1], c->st);
            c->tpending[slot] = true;
        }
        if (c->kused > fqd_ctx::KPOOL - 40)   // keep room for the next stage's kernels
            ktime_collect(c);
    }
};

int ktime_begin(fqd_ctx *c, int slot, hipStream_t st = nullptr)
{
    if (!((c->ktime_mask >> slot) & 1u) || c->kused >= fqd_ctx::KPOOL)
        return -1;
    const int i = c->kused++;
    c->kslot[i] = slot;
    (void)hipEventRecord(c->kev[2 * i], st ? st : c->st);
    return i;
}

void ktime_end(fqd_ctx *c, int i, hipStream_t st = nullptr)
{
    if (i >= 0)
        (void)hipEventRecord(c->kev[2 * i + 1], st ? st : c->st);
}

// after the stage's final synchronisation: fold the recorded pairs into the per-kernel sums
void ktime_collect(fqd_ctx *c)
{
    if (!c->kused)
        return;
    (void)hipStreamSynchronize(c->st);
    for (int i = 0; i < c->kused; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->kev[2 * i], c->kev[2 * i + 1]) == hipSuccess) {
            c->kms[c->kslot[i]] += ms;
            c->klaunches[c->kslot[i]] += 1;
        }
    }
    c->kused = 0;
}

// time one kernel launch with HIP events on the context's stream
#define KTIME(c, slot, call)                 \
    do {                                     \
        const int kt_ = ktime_begin((c), (slot)); \
        HIP_TRY((c), call);                  \
        ktime_end((c), kt_);                 \
    } while (0)

// ... on another stream of the context (the events must be recorded where the kernel runs)
#define KTIME_ON(c, slot, st, call)                      \
    do {                                                 \
        const int kt_ = ktime_begin((c), (slot), (st));  \
        HIP_TRY((c), call);                              \
        ktime_end((c), kt_, (st));                       \
    } while (0)

int sort_u32_pairs(fqd_ctx *c, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, uint64_t n,
                   int bits = 32)
{
    if (!n)
        return FQD_OK;
    const size_t need = fqd::sort_pairs_u32_u32_temp(n, 0, bits);
    HIP_TRY(c, c->tmp.reserve(need + 16));
    HIP_TRY(c, fqd::sort_pairs_u32_u32(c->tmp.p, need, kin, kout, vin, vout, n, 0, bits, c->st));
    return FQD_OK;
}

int scan_u32(fqd_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n)
{
    const size_t need = fqd::scan_u32_temp(n);
    HIP_TRY(c, c->tmp.reserve(need + 16));
    HIP_TRY(c, fqd::inclusive_scan_u32(c->tmp.p, need, in, out, n, c->st));
    return FQD_OK;
}

}  // namespace

// helpers shared between the api_*.hip files (defined in the file named)
int fqd_api_ensure_hashes(fqd_ctx *c);                 // api.hip
int fqd_api_components_queue(fqd_ctx *c, bool flatten, bool on_side = false);   // api_graph.hip
int fqd_api_flat_labels(fqd_ctx *c);                       // api_graph.hip
// the collapse over c->recs with DEVICE weights and ids (api.hip)
int fqd_api_collapse_device(fqd_ctx *c, const uint32_t *d_weights, IdSource ids, uint64_t id_limit, uint64_t *n_unique);
int fqd_api_graph_preinit(fqd_ctx *c, int method);          // api_graph.hip
extern "C" int fqd_api_partition_pairs(fqd_ctx *c, const uint32_t *keys, uint64_t N, uint32_t B, bool slabs,
                                       const uint32_t **items_out, const uint32_t **bucket_end_out,
                                       const uint32_t *values);   // api_search.hip
