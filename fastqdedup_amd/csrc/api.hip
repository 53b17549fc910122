// api.hip -- the C ABI of libfqdedup_hip.so (include/fqdedup_hip.h): context,
// device buffers, stage orchestration, HIP-event timing. No kernels here.
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
enum Ctr { C_BAD = 0, C_COLLISIONS = 1, C_CHANGED = 2, C_MINLEN = 3, C_MAXLEN = 4, C_N32 = 8 };
enum Ctr64 { C64_EDGES = 0, C64_ROOTS = 1, C64_SUM = 2, C64_STATS = 3, C64_CANDS = 4, C64_CAND_NEED = 5, C64_N = 8 };

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

    // stage 1
    uint64_t n = 0;
    DevBuf in_bytes, in_offsets, recs, lens, hashes, owners;
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
    DevBuf ld_small, ld_part2, ld_matrix, ld_matrix_incl;
    DevBuf ld_hist, ld_hist_incl, ld_start, ld_cursor, ld_part, ld_tmp_rec, ld_tmp_count, ld_tmp_first, ld_unique,
        ld_unique_incl;
    int collapse_path = 0;  // 1: LDS bucket dedupe, 2: sort + verify (last fqd_collapse)
    // stage 3
    uint64_t E = 0, edge_cap = 0;
    DevBuf seg_hashes, sorted_hash, sorted_uid, uid_iota, edges, sel_hash, sel_uid;
    DevBuf gp_a, gp_b, gp_small, gp_cands;   // grouped search pass: items after level 1 / level 2, small tables, candidate pairs
    uint64_t gp_cand_cap = 0;
    DevBuf q_table, q_pass, q_means, q_bytes, q_offsets;
    DevBuf len_present, ed_hash, ed_payload, ed_hash_sorted, ed_payload_sorted, ed_cands, ed_cands_sorted, d_alphabet;
    fqd::PairStats last_stats{};
    bool stats_pending = false;   // d_stats holds the slots of the last search, not yet summed into last_stats
    // stage 4
    uint64_t n_clusters = 0, roots_seen = 0;
    DevBuf labels, hook_slots;
    bool labels_flat = false;
    // stage 5
    uint64_t n_kept = 0, n_listed = 0;          // kept keys; kept keys whose first holder is in the id window
    uint64_t id_lo = 0, id_hi = ~0ull;
    DevBuf best, state, blocked, taint, root_taint, kept, kept_u32, kept_scan, kept_ids, kept_ids_sorted;
    // scratch
    DevBuf tmp, stage_a, stage_b, stage_c, stage_d;

    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
    // per-kernel timing (fqd_kernel_times): a pool of event pairs, drained at every stage end
    static constexpr int KPOOL = 96;
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

template <typename T>
int from_device(fqd_ctx *c, T *dst, const void *src, size_t count, int mem)
{
    if (!dst || !count)
        return FQD_OK;
    HIP_TRY(c, hipMemcpyAsync(dst, src, count * sizeof(T),
                              mem == FQD_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    return FQD_OK;
}

int read_ctr32(fqd_ctx *c, int idx, uint32_t *v)
{
    HIP_TRY(c, hipMemcpyAsync(v, c->d_ctr32.as<uint32_t>() + idx, 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    return FQD_OK;
}

int read_ctr64(fqd_ctx *c, int idx, unsigned long long *v, int count = 1)
{
    HIP_TRY(c, hipMemcpyAsync(v, c->d_ctr64.as<unsigned long long>() + idx, 8 * (size_t)count,
                              hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
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

struct StageTimer {
    fqd_ctx *c;
    int slot;
    StageTimer(fqd_ctx *ctx, int s) : c(ctx), slot(s) { (void)hipEventRecord(c->ev0, c->st); }
    void stop()
    {
        (void)hipEventRecord(c->ev1, c->st);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess)
            c->ms[slot] = ms;
        c->launches[slot] = 1;
        ktime_collect(c);
    }
};

int ktime_begin(fqd_ctx *c, int slot)
{
    if (c->kused >= fqd_ctx::KPOOL)
        return -1;
    const int i = c->kused++;
    c->kslot[i] = slot;
    (void)hipEventRecord(c->kev[2 * i], c->st);
    return i;
}

void ktime_end(fqd_ctx *c, int i)
{
    if (i >= 0)
        (void)hipEventRecord(c->kev[2 * i + 1], c->st);
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
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

int collapse_lds(fqd_ctx *c, const uint32_t *d_w, IdSource d_ids, bool *done)
{
    *done = false;
    const uint64_t n = c->n;
    const KeyShape sh = c->ks;
    const char *force = getenv("FQD_COLLAPSE");  // "sort" / "lds": tests pin a path
    if (force && !strcmp(force, "sort"))
        return FQD_OK;
    if (sh.ragged || sh.stride != 4 || sh.planes * sh.words > 3 || n >= 0xFFFFFF00ull)
        return FQD_OK;
    if (n < 32768 && !(force && !strcmp(force, "lds")))
        return FQD_OK;
    uint32_t B = 8;
    while (B < 18 && (n >> B) > 800)   // ~400-800 reads per bucket: 2x fewer workgroups than at 200-400, longer runs
        B++;
    if (const char *e = getenv("FQD_LDS_BUCKET_BITS"))  // tests: few buckets => table overflow => fallback
        B = (uint32_t)std::max(1, std::min(18, atoi(e)));
    const uint32_t B1 = std::min<uint32_t>(B, 8), B2 = B - B1;
    const uint32_t bins1 = 1u << B1, bins2 = 1u << B2, n_buckets = 1u << B;
    const uint32_t kw = sh.planes * sh.words, tile = fqd::part_tile_size();
    const uint32_t tiles1 = (uint32_t)((n + tile - 1) / tile), max_tiles2 = tiles1 + bins1;
    HIP_TRY(c, c->ld_hist.reserve((size_t)n_buckets * 4 + 1024 * 4));
    HIP_TRY(c, c->ld_hist_incl.reserve((size_t)n_buckets * 4 + 1024 * 4));
    HIP_TRY(c, c->ld_start.reserve(((size_t)n_buckets + 1) * 4 + 16));
    HIP_TRY(c, c->ld_cursor.reserve((size_t)n_buckets * 4 + 1024 * 4));
    HIP_TRY(c, c->ld_unique.reserve((size_t)n_buckets * 4 + 16));
    HIP_TRY(c, c->ld_unique_incl.reserve((size_t)n_buckets * 4 + 16));
    HIP_TRY(c, c->ld_small.reserve(4096 * 4));
    HIP_TRY(c, c->ld_part.reserve(n * 16 + 16));
    HIP_TRY(c, c->ld_tmp_rec.reserve(n * 16 + 16));   // level-2 output first, then the dedupe's tmp
    HIP_TRY(c, c->ld_part2.reserve(n * 16 + 16));
    HIP_TRY(c, c->ld_tmp_count.reserve(n * 4 + 16));
    HIP_TRY(c, c->ld_tmp_first.reserve(n * 4 + 16));
    // small device tables: [0] seg_start1 (2) | [8] tile_start1 (2) | [16] start1 (257) | [512] tile_start2 (257)
    uint32_t *small = c->ld_small.as<uint32_t>();
    uint32_t *seg1 = small, *tiles1_d = small + 8, *start1 = small + 16, *tiles2_d = small + 512;
    const uint32_t seg1_h[2] = {0u, (uint32_t)n}, tiles1_h[2] = {0u, tiles1};
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
    KTIME(c, FQD_K_PART_SCATTER1, fqd::launch_part_scatter(true, c->hashes.as<uint32_t>(), c->recs.as<uint32_t>(), seg1, tiles1_d, 1,
                                        tiles1, 32 - B1, bins1, kw, sh.max_len, c->ld_matrix_incl.as<uint32_t>(),
                                        c->ld_part.as<uint32_t>(), c->st));
    const uint32_t *parted = c->ld_part.as<uint32_t>();
    if (B2 == 0) {
        HIP_TRY(c, hipMemcpyAsync(c->ld_start.p, start1, ((size_t)bins1 + 1) * 4, hipMemcpyDeviceToDevice, c->st));
    } else {
        // ---- level 2: every part into 2^B2 buckets by the next B2 hash bits
        HIP_TRY(c, fqd::launch_tile_starts(start1, bins1, tiles2_d, c->st));
        HIP_TRY(c, hipMemsetAsync(c->ld_hist.p, 0, (size_t)n_buckets * 4, c->st));
        KTIME(c, FQD_K_PART_HIST2, fqd::launch_part_hist(false, nullptr, c->ld_part.as<uint32_t>(), start1, tiles2_d, bins1, max_tiles2,
                                         32 - B, bins2, kw, sh.max_len, c->ld_hist.as<uint32_t>(), c->st));
        FQD_TRY(scan_u32(c, c->ld_hist.as<uint32_t>(), c->ld_hist_incl.as<uint32_t>(), n_buckets));
        HIP_TRY(c, fqd::launch_bucket_starts(c->ld_hist_incl.as<uint32_t>(), n_buckets, c->ld_start.as<uint32_t>(),
                                             c->ld_cursor.as<uint32_t>(), c->st));
        KTIME(c, FQD_K_PART_SCATTER2, fqd::launch_part_scatter(false, nullptr, c->ld_part.as<uint32_t>(), start1, tiles2_d, bins1,
                                            max_tiles2, 32 - B, bins2, kw, sh.max_len, c->ld_cursor.as<uint32_t>(),
                                            c->ld_part2.as<uint32_t>(), c->st));
        parted = c->ld_part2.as<uint32_t>();
    }
    FQD_TRY(zero_ctr32(c, C_BAD));
    KTIME(c, FQD_K_DEDUPE, fqd::launch_bucket_dedupe(parted, c->ld_start.as<uint32_t>(), n_buckets, d_w,
                                         c->ld_tmp_rec.as<uint32_t>(), c->ld_tmp_count.as<uint32_t>(),
                                         c->ld_tmp_first.as<uint32_t>(), c->ld_unique.as<uint32_t>(),
                                         c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
    FQD_TRY(scan_u32(c, c->ld_unique.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets));
    uint32_t U32 = 0, overflow = 0;
    HIP_TRY(c, hipMemcpyAsync(&U32, c->ld_unique_incl.as<uint32_t>() + (n_buckets - 1), 4, hipMemcpyDeviceToHost,
                              c->st));
    FQD_TRY(read_ctr32(c, C_BAD, &overflow));
    if (overflow)
        return FQD_OK;  // some bucket held more distinct keys than the LDS table: sort-based path
    const uint64_t U = U32;
    HIP_TRY(c, c->urecs.reserve(U * 16 + 16));
    HIP_TRY(c, c->ulens.reserve(U * 4 + 16));
    HIP_TRY(c, c->ucounts.reserve(U * 4 + 16));
    HIP_TRY(c, c->ufirst.reserve(U * 8 + 16));
    KTIME(c, FQD_K_COMPACT, fqd::launch_bucket_compact(c->ld_start.as<uint32_t>(), c->ld_unique_incl.as<uint32_t>(), n_buckets,
                                          c->ld_tmp_rec.as<uint32_t>(), c->ld_tmp_count.as<uint32_t>(),
                                          c->ld_tmp_first.as<uint32_t>(), d_ids, c->urecs.as<uint32_t>(),
                                          c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(), c->st));
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
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

// =============================================================================
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
              hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
              hipEventCreate(&c->evk0) == hipSuccess && hipEventCreate(&c->evk1) == hipSuccess &&
              [&] { for (hipEvent_t &e : c->kev) if (hipEventCreate(&e) != hipSuccess) return false; return true; }() &&
              c->d_ctr32.reserve(C_N32 * 4) == hipSuccess && c->d_ctr64.reserve(C64_N * 8) == hipSuccess &&
              c->d_lut.reserve(256) == hipSuccess &&
              c->d_stats.reserve(FQD_STAT_SLOTS * sizeof(fqd::PairStats)) == hipSuccess;
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
                      &c->stage_a, &c->stage_b, &c->stage_c, &c->stage_d, &c->hook_slots, &c->owners, &c->taint, &c->root_taint, &c->gp_a, &c->gp_b, &c->gp_small, &c->gp_cands, &c->seg_tab};
    for (DevBuf *b : bufs)
        b->release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evk0) (void)hipEventDestroy(c->evk0);
    if (c->evk1) (void)hipEventDestroy(c->evk1);
    for (hipEvent_t e : c->kev)
        if (e) (void)hipEventDestroy(e);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

const char *fqd_last_error(const fqd_ctx *c) { return c ? c->err.c_str() : "null context"; }

int fqd_synchronize(fqd_ctx *c)
{
    FQD_TRY(bind(c));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    return FQD_OK;
}

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
    HIP_TRY(c, hipMemcpyAsync(c->d_lut.p, lut, 256, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
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
            HIP_TRY(c, hipStreamSynchronize(c->st));
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
            HIP_TRY(c, hipStreamSynchronize(c->st));
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
            HIP_TRY(c, hipStreamSynchronize(c->st));
            max_len = mm[1];
            ragged = mm[0] != mm[1];
        } else if (!n) {
            max_len = 0;
        }
    }
    for (int attempt = 0;; attempt++) {
        build_alphabet(c, present, lut);
        FQD_TRY(set_geometry(c, max_len, ragged));
        HIP_TRY(c, hipMemcpyAsync(c->d_lut.p, lut, 256, hipMemcpyHostToDevice, c->st));
        const KeyShape sh = c->ks;
        HIP_TRY(c, c->recs.reserve((size_t)n * sh.stride * 4 + 16));
        HIP_TRY(c, c->hashes.reserve((size_t)n * 4 + 16));
        if (sh.ragged)
            HIP_TRY(c, c->lens.reserve((size_t)n * 4 + 16));
        if (c->owner_rule.parts)
            HIP_TRY(c, c->owners.reserve((size_t)n * 4 + 16));
        c->owners_done = fqd::OwnerRule{};
        FQD_TRY(zero_ctr32(c, C_BAD));
        (void)hipEventRecord(c->evk0, c->st);
        KTIME(c, FQD_K_PACK, fqd::launch_pack(d_bytes, n_bytes, d_off, n, fixed_len, sh, c->d_lut.as<uint8_t>(), lut,
                                    c->recs.as<uint32_t>(), sh.ragged ? c->lens.as<uint32_t>() : nullptr,
                                    c->hashes.as<uint32_t>(), c->owner_rule.parts ? c->owners.as<uint32_t>() : nullptr,
                                    c->owner_rule, c->d_ctr32.as<uint32_t>() + C_BAD, c->st));
        (void)hipEventRecord(c->evk1, c->st);
        uint32_t bad = 0;
        FQD_TRY(read_ctr32(c, C_BAD, &bad));
        float kms = 0;
        if (hipEventElapsedTime(&kms, c->evk0, c->evk1) == hipSuccess) {
            c->ms[FQD_T_PACK_KERNEL] = kms;
            c->launches[FQD_T_PACK_KERNEL] = 1;
        }
        if (getenv("FQD_DEBUG"))
            fprintf(stderr, "[fqd] pack attempt %d: n=%llu len=%u K=%u W=%u stride=%u alphabet=%.*s bad=%u kernel=%.3f ms\n",
                    attempt, (unsigned long long)n, max_len, sh.planes, sh.words, sh.stride, (int)c->shape.alphabet_size,
                    (const char *)c->shape.alphabet, bad, kms);
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

    bool lds_done = false;
    FQD_TRY(collapse_lds(c, weights ? d_w : nullptr, ids, &lds_done));
    if (lds_done) {
        timer.stop();
        c->collapse_path = 1;
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
    c->collapse_path = 2;
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
    HIP_TRY(c, hipStreamSynchronize(c->st));

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
    HIP_TRY(c, c->ufirst.reserve(U * 8 + 16));
    KTIME(c, FQD_K_WRITE_UNIQUE, fqd::launch_write_unique(c->run_start.as<uint32_t>(), c->run_weight.as<uint32_t>(),
                                        c->live_flag.as<uint32_t>(), c->live_idx.as<uint32_t>(), n_runs,
                                        c->ids_sorted.as<uint32_t>(), c->recs.as<uint32_t>(), c->lens.as<uint32_t>(),
                                        ids.ids64, sh, c->urecs.as<uint32_t>(),
                                        c->ulens.as<uint32_t>(), c->ucounts.as<uint32_t>(), c->ufirst.as<uint64_t>(),
                                        c->st));
    timer.stop();
    c->U = U;
    c->n_counted = counted;
    c->collapsed = true;
    c->first_distinct = true;
    set_id_range(c, read_ids ? id_limit : n);
    c->stage = ST_UNIQUE;
    if (n_unique)
        *n_unique = U;
    return FQD_OK;
}

int fqd_collapse(fqd_ctx *c, const uint32_t *weights, const uint64_t *read_ids, int mem, uint64_t *n_unique)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED)
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
    if (c->stage < ST_PACKED)
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
    HIP_TRY(c, hipStreamSynchronize(c->st));      // rows32 goes out of scope
    IdSource ids;
    ids.stamped = c->recs.as<uint32_t>();
    ids.stride = sh.stride;
    ids.spare_word = sh.planes * sh.words;
    ids.seg_rows = d_rows;
    ids.seg_id0 = d_id0;
    ids.n_seg = n_seg;
    return collapse_impl(c, weights, mem, ids, id_limit, n_unique);
}

// One Hamming search pass without a device-wide sort (group.hip): the (segment hash, uid) pairs
// are partitioned into 2^B buckets of ~200 keys by the top hash bits, then one wave per bucket
// sub-sorts them in LDS and lists the pairs with equal hashes; a second kernel verifies those.
// Queues work only (no host round trip).
static int grouped_pass(fqd_ctx *c, const uint32_t *hashes, uint64_t U, uint32_t d, uint32_t s, uint32_t nseg)
{
    const KeyShape sh = c->ks;
    uint32_t B = 8;
    while (B < 20 && (U >> B) > 320)   // ~160-320 keys per bucket: a wave sorts them into 64 sub-bins in LDS
        B++;
    if (const char *e = getenv("FQD_GROUP_BUCKET_BITS"))   // tests: few, crowded buckets
        B = (uint32_t)std::max(1, std::min(20, atoi(e)));
    const uint32_t B1 = B <= 18 ? std::min<uint32_t>(B, 8) : B - 10, B2 = B - B1;
    const uint32_t bins1 = 1u << B1, bins2 = 1u << B2, n_buckets = 1u << B;
    const uint32_t tile = fqd::group_tile_size();
    const uint32_t tiles1 = (uint32_t)((U + tile - 1) / tile), max_tiles2 = tiles1 + bins1;
    HIP_TRY(c, c->gp_a.reserve(U * 8 + 16));
    HIP_TRY(c, c->gp_small.reserve(4096 * 4 + (size_t)fqd::group_cand_lists() * 64));
    HIP_TRY(c, c->ld_start.reserve(((size_t)n_buckets + 1) * 4 + 16));
    uint32_t *small = c->gp_small.as<uint32_t>();
    uint32_t *seg1 = small, *tiles1_d = small + 8, *start1 = small + 16, *tiles2_d = small + 2048;
    const uint32_t seg1_h[2] = {0u, (uint32_t)U}, tiles1_h[2] = {0u, tiles1};
    HIP_TRY(c, hipMemcpyAsync(seg1, seg1_h, 8, hipMemcpyHostToDevice, c->st));
    HIP_TRY(c, hipMemcpyAsync(tiles1_d, tiles1_h, 8, hipMemcpyHostToDevice, c->st));
    // ---- level 1: (bin x tile) count matrix, scan, placement without atomics
    const size_t matrix = (size_t)bins1 * tiles1;
    HIP_TRY(c, c->ld_matrix.reserve(matrix * 4 + 16));
    HIP_TRY(c, c->ld_matrix_incl.reserve(matrix * 4 + 16));
    KTIME(c, FQD_K_GROUP_HIST, fqd::launch_group_hist(true, hashes, nullptr, seg1, tiles1_d, 1, tiles1, 32 - B1, bins1,
                                                      c->ld_matrix.as<uint32_t>(), c->st));
    FQD_TRY(scan_u32(c, c->ld_matrix.as<uint32_t>(), c->ld_matrix_incl.as<uint32_t>(), matrix));
    HIP_TRY(c, fqd::launch_group_matrix_starts(c->ld_matrix_incl.as<uint32_t>(), bins1, tiles1, start1, c->st));
    KTIME(c, FQD_K_GROUP_SCATTER, fqd::launch_group_scatter(true, hashes, nullptr, seg1, tiles1_d, 1, tiles1, 32 - B1,
                                                            bins1, c->ld_matrix_incl.as<uint32_t>(),
                                                            c->gp_a.as<uint32_t>(), c->st));
    const uint32_t *items = c->gp_a.as<uint32_t>();
    if (B2 == 0) {
        HIP_TRY(c, hipMemcpyAsync(c->ld_start.p, start1, ((size_t)bins1 + 1) * 4, hipMemcpyDeviceToDevice, c->st));
    } else {
        // ---- level 2: every part into 2^B2 buckets by the next hash bits
        HIP_TRY(c, c->gp_b.reserve(U * 8 + 16));
        HIP_TRY(c, c->ld_hist.reserve((size_t)n_buckets * 4 + 16));
        HIP_TRY(c, c->ld_hist_incl.reserve((size_t)n_buckets * 4 + 16));
        HIP_TRY(c, c->ld_cursor.reserve((size_t)n_buckets * 4 + 16));
        HIP_TRY(c, fqd::launch_group_tile_starts(start1, bins1, tiles2_d, c->st));
        HIP_TRY(c, hipMemsetAsync(c->ld_hist.p, 0, (size_t)n_buckets * 4, c->st));
        KTIME(c, FQD_K_GROUP_HIST, fqd::launch_group_hist(false, nullptr, c->gp_a.as<uint32_t>(), start1, tiles2_d, bins1,
                                                          max_tiles2, 32 - B, bins2, c->ld_hist.as<uint32_t>(), c->st));
        FQD_TRY(scan_u32(c, c->ld_hist.as<uint32_t>(), c->ld_hist_incl.as<uint32_t>(), n_buckets));
        HIP_TRY(c, fqd::launch_group_bucket_starts(c->ld_hist_incl.as<uint32_t>(), n_buckets,
                                                   c->ld_start.as<uint32_t>(), c->ld_cursor.as<uint32_t>(), c->st));
        KTIME(c, FQD_K_GROUP_SCATTER, fqd::launch_group_scatter(false, nullptr, c->gp_a.as<uint32_t>(), start1, tiles2_d,
                                                                bins1, max_tiles2, 32 - B, bins2,
                                                                c->ld_cursor.as<uint32_t>(), c->gp_b.as<uint32_t>(),
                                                                c->st));
        items = c->gp_b.as<uint32_t>();
    }
    // candidates (pairs with equal segment hashes) -> device list -> verification, one thread per pair
    if (c->gp_cand_cap < 1024 || !c->gp_cands.p) {
        c->gp_cand_cap = std::max<uint64_t>(1u << 20, 2 * U);   // split evenly over the lists: leave slack
        HIP_TRY(c, c->gp_cands.reserve(c->gp_cand_cap * 8));
    }
    c->gp_cand_cap = c->gp_cands.cap / 8;
    // the candidate counters (one per list, a cache line apart) live behind the small tables
    unsigned long long *cand_ctr = reinterpret_cast<unsigned long long *>(small + 4096);
    HIP_TRY(c, hipMemsetAsync(cand_ctr, 0, (size_t)fqd::group_cand_lists() * 64, c->st));
    unsigned long long *ctr = c->d_ctr64.as<unsigned long long>();
    KTIME(c, FQD_K_PAIRS, fqd::launch_grouped_candidates(items, c->ld_start.as<uint32_t>(), n_buckets, B,
                                                         c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap,
                                                         c->st));
    KTIME(c, FQD_K_VERIFY, fqd::launch_verify_candidates(c->gp_cands.as<uint64_t>(), cand_ctr, c->gp_cand_cap,
                                                         c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), sh, d, s, nseg,
                                                         c->edges.as<uint32_t>(), ctr + C64_EDGES, c->edge_cap,
                                                         ctr + C64_CAND_NEED, c->d_stats.as<fqd::PairStats>(), c->st));
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
    if (n_shards == 0 || shard >= n_shards)
        return fail(c, FQD_E_VALUE, "bad shard");
    const KeyShape sh = c->ks;
    // Levenshtein <= 1 between keys of ONE length is Hamming <= 1 (an indel changes the length):
    // that case shares the Hamming search; everything else takes the bucketed edit search.
    const bool edit_general = metric == FQD_METRIC_EDIT && !(max_distance <= 1 && !sh.ragged);
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
    FQD_TRY(zero_ctr64(c, C64_EDGES));
    HIP_TRY(c, hipMemsetAsync(c->d_stats.p, 0, FQD_STAT_SLOTS * sizeof(fqd::PairStats), c->st));
    if (edit_general && U >= 2 && (max_distance > 0 || !c->collapsed)) {
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
        KTIME(c, FQD_K_SEG_HASH, fqd::launch_segment_hashes(c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), U, sh, nseg,
                                              seg_lo, seg_hi, 0, c->seg_hashes.as<uint32_t>(), c->st));
        if (c->edge_cap < 1024 || !c->edges.p) {
            c->edge_cap = std::max<uint64_t>(1024, U);
            HIP_TRY(c, c->edges.reserve(c->edge_cap * 8));
        }
        c->edge_cap = c->edges.cap / 8;
        unsigned long long have = 0;
        if (n_shards > 1) {
            HIP_TRY(c, c->sel_hash.reserve(U * 4 + 16));
            HIP_TRY(c, c->sel_uid.reserve(U * 4 + 16));
        }
        // Grouping by partition (group.hip) unless a bucket shard was asked for or the table is small
        // (FQD_EDGES=sort|grouped pins the path for tests).
        const char *pin = getenv("FQD_EDGES");
        bool grouped = n_shards == 1 && U < 0xFFFFFF00ull && (pin ? !strcmp(pin, "grouped") : U >= 65536);
        // Candidate pairs are listed before they are verified; a segment value shared by very many
        // keys (all of them pairwise candidates) would need a list beyond this budget: the search
        // then runs again on the sort path, which verifies in place and needs no list.
        uint64_t cand_budget = std::max<uint64_t>(8 * U, 1ull << 24);
        if (const char *e = getenv("FQD_GROUP_CAND_BUDGET"))
            cand_budget = strtoull(e, nullptr, 10);
        bool iota_ready = false;
        FQD_TRY(zero_ctr64(c, C64_CAND_NEED));
        // All d+1 passes are queued without a host round trip; the edge count is read ONCE at the
        // end. If the passes overflowed the edge buffer (the count still says how many edges there
        // are), the buffer is grown to the known need and the whole search runs again.
        for (int attempt = 0;; attempt++) {
            for (uint32_t s = seg_lo; s < seg_hi; s++) {
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
                               c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), sh, d, s, nseg, 0, 1,
                               c->edges.as<uint32_t>(), c->d_ctr64.as<unsigned long long>() + C64_EDGES, c->edge_cap,
                               c->d_stats.as<fqd::PairStats>(), c->st));
            }
            unsigned long long ctrs[C64_CAND_NEED + 1] = {0};
            FQD_TRY(read_ctr64(c, 0, ctrs, C64_CAND_NEED + 1));
            const unsigned long long now = ctrs[C64_EDGES], cand_need = grouped ? ctrs[C64_CAND_NEED] : 0;
            if (now <= c->edge_cap && cand_need <= c->gp_cand_cap) {
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
            if (cand_need > c->gp_cand_cap) {
                if (cand_need > cand_budget) {
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

static int ensure_flat_labels(fqd_ctx *c)
{
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
static int components_queue(fqd_ctx *c, bool flatten)
{
    const uint64_t U = c->U;
    HIP_TRY(c, c->labels.reserve(U * 4 + 16));
    HIP_TRY(c, c->hook_slots.reserve(FQD_HOOK_SLOTS * 64));
    HIP_TRY(c, hipMemsetAsync(c->hook_slots.p, 0, FQD_HOOK_SLOTS * 64, c->st));
    HIP_TRY(c, fqd::launch_uf_init(c->labels.as<uint32_t>(), U, c->st));
    KTIME(c, FQD_K_UF_UNION, fqd::launch_uf_union(c->labels.as<uint32_t>(), c->edges.as<uint32_t>(), c->E,
                                                  c->hook_slots.as<unsigned long long>(), c->st));
    HIP_TRY(c, fqd::launch_hook_total(c->hook_slots.as<unsigned long long>(), U,
                                      c->d_ctr64.as<unsigned long long>() + C64_ROOTS, c->st));
    c->labels_flat = false;
    if (flatten)
        FQD_TRY(ensure_flat_labels(c));
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
    FQD_TRY(zero_ctr64(c, C64_SUM));
    if (by_map) {
        HIP_TRY(c, c->stage_c.reserve(window + 16));
        if (window)
            HIP_TRY(c, hipMemsetAsync(c->stage_c.p, 0, window, c->st));
        HIP_TRY(c, fqd::launch_kept_flags(method, c->labels.as<uint32_t>(), c->best.as<uint32_t>(),
                                          c->state.as<uint8_t>(), c->ufirst.as<uint64_t>(), c->id_lo, c->id_hi, U,
                                          c->kept.as<uint8_t>(), nullptr, c->stage_c.as<uint8_t>(), window,
                                          c->ucounts.as<uint32_t>(), c->blocked.as<uint32_t>(),
                                          c->root_taint.as<uint8_t>(),
                                          c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
        HIP_TRY(c, c->kept_ids_sorted.reserve(std::min(window, U) * 8 + 16));
        const uint32_t blocks = fqd::window_blocks(window);
        uint32_t listed = 0;
        if (blocks) {
            HIP_TRY(c, c->kept_u32.reserve((size_t)blocks * 4 + 16));
            HIP_TRY(c, c->kept_scan.reserve((size_t)blocks * 4 + 16));
            HIP_TRY(c, fqd::launch_window_count(c->stage_c.as<uint8_t>(), window, c->kept_u32.as<uint32_t>(), c->st));
            FQD_TRY(scan_u32(c, c->kept_u32.as<uint32_t>(), c->kept_scan.as<uint32_t>(), blocks));
            HIP_TRY(c, fqd::launch_window_emit(c->stage_c.as<uint8_t>(), window, c->kept_scan.as<uint32_t>(), base,
                                               c->kept_ids_sorted.as<uint64_t>(), c->st));
            HIP_TRY(c, hipMemcpyAsync(&listed, c->kept_scan.as<uint32_t>() + (blocks - 1), 4, hipMemcpyDeviceToHost,
                                      c->st));
        }
        unsigned long long both[2] = {0, 0};      // C64_ROOTS, C64_SUM: one read for fqd_cluster
        FQD_TRY(read_ctr64(c, C64_ROOTS, both, 2));
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
    HIP_TRY(c, fqd::launch_kept_flags(method, c->labels.as<uint32_t>(), c->best.as<uint32_t>(),
                                      c->state.as<uint8_t>(), c->ufirst.as<uint64_t>(), c->id_lo, c->id_hi, U,
                                      c->kept.as<uint8_t>(), c->kept_u32.as<uint32_t>(), nullptr, 0,
                                      c->ucounts.as<uint32_t>(), c->blocked.as<uint32_t>(), c->root_taint.as<uint8_t>(),
                                      c->d_ctr64.as<unsigned long long>() + C64_SUM, c->st));
    FQD_TRY(scan_u32(c, c->kept_u32.as<uint32_t>(), c->kept_scan.as<uint32_t>(), U));
    uint32_t nk = 0;
    HIP_TRY(c, hipMemcpyAsync(&nk, c->kept_scan.as<uint32_t>() + (U - 1), 4, hipMemcpyDeviceToHost, c->st));
    unsigned long long both[2] = {0, 0};
    FQD_TRY(read_ctr64(c, C64_ROOTS, both, 2));
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
    HIP_TRY(c, fqd::launch_dissect_init(c->best.as<uint32_t>(), c->state.as<uint8_t>(), U, c->st));
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
        HIP_TRY(c, c->blocked.reserve(U * 4 + 16));   // here: union-find over the count-1 keys
        HIP_TRY(c, c->taint.reserve(U + 16));
        HIP_TRY(c, c->root_taint.reserve(U + 16));
        if (E) {
            HIP_TRY(c, hipMemsetAsync(c->taint.p, 0, U, c->st));
            HIP_TRY(c, hipMemsetAsync(c->root_taint.p, 0, U, c->st));
            HIP_TRY(c, fqd::launch_uf_init(c->blocked.as<uint32_t>(), U, c->st));
            for (int pass = 1; pass <= 2; pass++)
                KTIME(c, FQD_K_DISSECT_ROUND, fqd::launch_directional_closed(
                          c->edges.as<uint32_t>(), E, c->ucounts.as<uint32_t>(), c->urecs.as<uint32_t>(),
                          c->ulens.as<uint32_t>(), sh, c->blocked.as<uint32_t>(), c->state.as<uint8_t>(),
                          c->taint.as<uint8_t>(), c->root_taint.as<uint8_t>(), c->best.as<uint32_t>(), pass, c->st));
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
                                            c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), sh, c->st));
        for (uint64_t round = 1; round <= U + 2; round += 2) {
            FQD_TRY(zero_ctr32(c, C_CHANGED));
            for (uint32_t k = 0; k < 2; k++)
                KTIME(c, FQD_K_DISSECT_ROUND, fqd::launch_adjacency_round(c->edges.as<uint32_t>(), E, U, c->state.as<uint8_t>(),
                                                   c->blocked.as<uint32_t>(), (uint32_t)(round + k), d_changed, c->st));
            uint32_t changed = 0;
            FQD_TRY(read_ctr32(c, C_CHANGED, &changed));
            if (!changed)
                break;
        }
    }
    FQD_TRY(list_kept(c, list_method));
    timer.stop();
    c->stage = ST_KEPT;
    if (n_kept)
        *n_kept = c->n_kept;
    return FQD_OK;
}

int fqd_cluster(fqd_ctx *c, const uint32_t *weights, const uint64_t *read_ids, int mem, int max_distance, int metric,
                int method, fqd_summary *out)
{
    if (max_distance < 0)
        return fail(c, FQD_E_VALUE, "max_distance should be non-negative");
    FQD_TRY(fqd_collapse(c, weights, read_ids, mem, nullptr));
    FQD_TRY(fqd_find_edges(c, max_distance, metric, 0, 1, nullptr));
    // no host round trip between components and dissection; labels are flattened only if read
    FQD_TRY(components_queue(c, method == FQD_METHOD_HIGHEST_COUNT));
    c->stage = ST_LABELS;
    c->ms[FQD_T_COMPONENTS] = 0;
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
    return from_device(c, out, c->kept_ids_sorted.p, (size_t)c->n_listed, mem);
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

// ---- exchange -------------------------------------------------------------------
int fqd_export_packed(fqd_ctx *c, uint32_t *recs, uint32_t *lens, uint32_t *hashes, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED)
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
        FQD_TRY(ensure_hashes(c));
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
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
    if (c->stage < ST_PACKED)
        return fail(c, FQD_E_STATE, "nothing packed");
    FQD_TRY(check_parts(c, n_parts, mem, "fqd_export_packed_by_owner"));
    const uint64_t n = c->n;
    HIP_TRY(c, c->flags.reserve(n * 4 + 16));
    FQD_TRY(ensure_hashes(c));
    HIP_TRY(c, fqd::launch_owner(c->hashes.as<uint32_t>(), n, n_parts, c->flags.as<uint32_t>(), c->st));
    return export_grouped(c, n, n_parts, c->flags.as<uint32_t>(), c->recs.as<uint32_t>(), c->lens.as<uint32_t>(),
                          weights, id0, recs, lens, ids, nullptr, weights_out, counts);
}

int fqd_export_packed_by_segment(fqd_ctx *c, uint32_t n_parts, uint32_t n_segments, uint32_t segment, uint64_t id0,
                                 const uint32_t *weights, uint32_t *recs, uint32_t *lens, uint64_t *ids,
                                 uint32_t *weights_out, uint64_t *counts, int mem)
{
    FQD_TRY(bind(c));
    if (c->stage < ST_PACKED)
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
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
    std::vector<unsigned long long> slots((size_t)FQD_HOOK_SLOTS * 8);
    HIP_TRY(c, hipMemcpyAsync(slots.data(), c->stage_b.p, FQD_HOOK_SLOTS * 64, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    unsigned long long hooks = 0;
    for (size_t i = 0; i < slots.size(); i += 8)
        hooks += slots[i];
    if (n_components)
        *n_components = n_nodes - hooks;
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
    c->n = n;
    c->hashes_valid = false;       // computed when somebody needs them (ensure_hashes)
    c->owners_done = fqd::OwnerRule{};
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
    HIP_TRY(c, c->ufirst.reserve(U * 8 + 16));
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
    c->U = U;
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
    c->E = E;
    c->stage = ST_EDGES;
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
    HIP_TRY(c, fqd::launch_contains(dq, dqo, n, c->urecs.as<uint32_t>(), c->ulens.as<uint32_t>(), c->U, c->ks,
                                    c->d_alphabet.as<uint8_t>(), max_distance, metric, c->stage_c.as<uint32_t>(),
                                    c->st));
    std::vector<uint32_t> flags((size_t)n);
    HIP_TRY(c, hipMemcpyAsync(flags.data(), c->stage_c.p, n * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
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
            HIP_TRY(c, hipStreamSynchronize(c->st));
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
        HIP_TRY(c, hipStreamSynchronize(c->st));
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
int fqd_stage_times(fqd_ctx *c, float *ms, uint32_t *launches)
{
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
        HIP_TRY(c, hipStreamSynchronize(c->st));
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
    HIP_TRY(c, hipStreamSynchronize(c->st));
    return FQD_OK;
}

}  // extern "C"
