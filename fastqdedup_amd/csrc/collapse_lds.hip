// collapse_lds.hip -- stage 2 for short keys (a record is one uint4: <= 3 plane words + a
// spare word): exact-duplicate collapse WITHOUT a device-wide sort and WITHOUT gathers.
// Same contract as collapse.hip (reference _triemodule.c:235-239, :261-264: an identical
// key bumps a count; pass-2 rule __init__.py:201-206: the first holder is remembered).
//
//   1+2. two-level partition by the top B = 8 + B2 bits of the 32-bit key hash: level 1
//      splits all reads into 256 parts, level 2 splits every part into 2^B2 buckets. Each
//      level is a histogram pass and a scatter pass over TILES of 2048 reads; bin counts
//      and the ranks inside a bin are LDS atomics, so the global cursor table sees ONE
//      atomic per (tile, bin) instead of one per read (a one-level scatter with one global
//      atomic per read took 5.3 ms for 50 M reads; this takes ~1 ms). The record travels
//      as a uint4 whose spare word carries the read index; level 2 re-derives the hash.
//   3. one workgroup per bucket (~400 reads) streams the bucket through an LDS hash
//      table keyed by the full record: a slot's tag is claimed with ds_cmpswap, the
//      claimer parks its record, later copies are VERIFIED word by word against the
//      parked record (a tag only proposes), then bump the slot's count / min index with
//      LDS atomics. Equal keys meet because they share a hash, hence a bucket.
//   4. per-bucket unique counts are scanned and the parked records are copied out.
//
// Every read is moved twice, always as a 16-byte store; nothing is gathered. Buckets hold whatever falls into them: a key with a
// million copies makes one long bucket (slow for that workgroup, still exact); a bucket
// with more distinct keys than the table holds raises `overflow` and the caller falls
// back to the sort-based collapse.
#include <algorithm>
#include "fqd_internal.h"
#include "partition.cuh"

// Items per thread of part_scatter12_kernel and the rounds its tile is staged in (partition.cuh ROUNDS). Measured at
// config 3: 8 / 1 (2048-record tiles) 0.42-0.45 ms; 16 / 2 (4096-record tiles, half the cursor atomics, 16-record
// runs, the same LDS) 0.49 ms, 16 / 4 0.50 ms; 8 / 2 (half the staging area, eight waves per SIMD instead of
// five) 0.427 ms -- neither occupancy nor run length nor the number of cursor atomics moves this kernel (its waves
// are parked 80 % of their cycles, profiles/r02_sq_per_kernel_config3.json).
#ifndef FQD_SCATTER12_EPT
#define FQD_SCATTER12_EPT 8
#endif
// 256-record chunks of a bucket the dedupe requests before it hashes the first (3: 0.344 ms instead of 0.32 at config 3 --
// a bucket of ~763 records then needs a second trip)
#ifndef FQD_DD12_AHEAD
#define FQD_DD12_AHEAD 4
#endif
#ifndef FQD_SCATTER12_ROUNDS
#define FQD_SCATTER12_ROUNDS 1
#endif

namespace {

constexpr uint32_t DD_SLOTS = 1024;        // LDS table slots per bucket (power of two)
constexpr uint32_t DD_THREADS = 256;
constexpr uint32_t DD_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t DD_HUGE = 16384;       // reads of a bucket from which on a wave merges its equal records first

// The partition itself (tiles, LDS counting sort, count matrix) is partition.cuh; here is what a
// 16-byte record looks like to it. LEVEL 1 reads the packed reads and stamps the read index into
// the spare word; level 2 reads level-1 output. The key is the record hash: recomputed from the
// record (a dozen VALU ops) wherever the record is loaded anyway -- reading the stored hash would
// add 4 B to the 32 B the scatter moves per read; only the level-1 HISTOGRAM reads the hash array
// (4 B instead of the 16-B record), when there is one (reads imported from other ranks have none).
struct RecordPolicy {
    using Item = uint4;
    static constexpr uint32_t EPT = 8;          // 2048-record tiles
    static constexpr uint32_t ROUNDS = 1;
    static constexpr bool MAY_SKIP = false;
    static constexpr bool CAN_SPILL = false;
    static __device__ __forceinline__ bool skip(const uint4 &) { return false; }
    struct Source {
        const uint32_t *hashes;   // level 1 only, may be NULL
        const uint4 *in;
        uint32_t kw, len;
        IdSource ids;             // ids.packed_bits != 0: level 1 stamps (segment, local index), see IdSource
        uint32_t stamp_div;       // != 0 (level 2 over slabs received from several ranks): segment / stamp_div is
                                  // the sender, stamped above the read index the record carries
    };
    static __device__ __forceinline__ uint32_t segment_tag(const Source &s, uint32_t seg)
    {
        if (!s.stamp_div)
            return 0u;
        const uint32_t sender = seg / s.stamp_div;
        return (s.ids.stamp_map ? s.ids.stamp_map[sender] : sender) << s.ids.packed_bits;
    }
    static __device__ __forceinline__ void apply_tag(uint4 &v, uint32_t tag) { v.w |= tag; }
    using Raw = uint4;
    template <bool LEVEL1>
    static __device__ __forceinline__ uint4 fetch(const Source &s, uint32_t i) { return s.in[i]; }
    struct Shared {};
    static __device__ __forceinline__ void init_shared(Shared &, uint32_t) {}
    static __device__ __forceinline__ void flush(const Source &, Shared &, uint32_t) {}
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t load(const Source &s, uint32_t i, uint4 &v)
    {
        Shared none;
        return finish<LEVEL1>(s, i, s.in[i], v, true, 0u, none);
    }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t finish(const Source &s, uint32_t i, const uint4 &raw, uint4 &v, bool,
                                                      uint32_t, Shared &)
    {
        v = raw;
        if (LEVEL1) {
            // words past the key are padding: zero on this rank's own packed reads, but a sender's
            // local index on reads received from other ranks (fqd_collapse_received) -- never part
            // of the key
            // (masks, not "kw == 3 ? v.w : ...": the compiler turns that select chain into an indexed
            // access and parks the whole register array of the tile in scratch memory: 0.49 -> 0.70 ms)
            const uint32_t is3 = 0u - (uint32_t)(s.kw == 3), is2 = 0u - (uint32_t)(s.kw == 2),
                           is1 = 0u - (uint32_t)(s.kw == 1);
            uint32_t tag = i;
            if (s.ids.packed_bits) {
                const uint32_t local = (v.w & is3) | (v.z & is2) | (v.y & is1);   // word kw of the record
                tag = (s.ids.packed_segment_of(i) << s.ids.packed_bits) | local;
            }
            v.z &= is3;                // words past the key are not key
            v.y &= is3 | is2;
            v.w = tag;
        }
        const uint32_t rec[3] = {v.x, v.y, v.z};
        return fqd_hash_record(rec, s.kw, s.len);
    }
    using KeyRaw = uint4;
    template <bool LEVEL1>
    static __device__ __forceinline__ uint4 key_fetch(const Source &s, uint32_t i)
    {
        if (LEVEL1 && s.hashes)
            return make_uint4(s.hashes[i], 0u, 0u, 0u);
        return s.in[i];
    }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t key_finish(const Source &s, uint32_t i, const uint4 &raw)
    {
        if (LEVEL1 && s.hashes)
            return raw.x;
        uint4 v;
        Shared none;
        return finish<false>(s, i, raw, v, true, 0u, none);
    }
};

template <bool LEVEL1>
__global__ __launch_bounds__(fqd_partition::THREADS) void part_hist_kernel(RecordPolicy::Source src,
                                                                           const uint32_t *__restrict__ seg_start,
                                                                           const uint32_t *__restrict__ tile_start,
                                                                           uint32_t n_seg, uint32_t shift,
                                                                           uint32_t n_bins, uint32_t *__restrict__ hist)
{
    fqd_partition::hist_body<RecordPolicy, LEVEL1>(src, seg_start, tile_start, n_seg, shift, n_bins, hist);
}

template <bool LEVEL1, uint32_t MAXB>
__global__ __launch_bounds__(fqd_partition::THREADS) void part_scatter_kernel(RecordPolicy::Source src,
                                                                              const uint32_t *__restrict__ seg_start,
                                                                              const uint32_t *__restrict__ tile_start,
                                                                              uint32_t n_seg, uint32_t shift,
                                                                              uint32_t n_bins,
                                                                              uint32_t *__restrict__ cursor,
                                                                              uint4 *__restrict__ out,
                                                                              uint32_t slab_cap,
                                                                              uint32_t *__restrict__ slab_overflow,
                                                                              const uint32_t *__restrict__ seg_end,
                                                                              uint32_t seg_shift, uint32_t seg_mask)
{
    fqd_partition::scatter_body<RecordPolicy, LEVEL1, MAXB>(src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out,
                                                            slab_cap, slab_overflow, seg_end, seg_shift, seg_mask);
}

// slabs received from n_senders ranks, sender by sender: segment s = sender * ppo + j holds the
// sender's slab my_part * ppo + j, whose cursor started at that slab's first slot ON THE SENDER
__global__ void owner_slab_bounds_kernel(const uint32_t *__restrict__ cursors, uint32_t n_senders, uint32_t ppo,
                                         uint32_t my_part, uint32_t cap, uint32_t *__restrict__ seg_start,
                                         uint32_t *__restrict__ seg_end)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, n = n_senders * ppo;
    if (s > n)
        return;
    seg_start[s] = s * cap;
    if (s < n) {
        const uint32_t first = (my_part * ppo + s % ppo) * cap;
        const uint32_t fill = cursors[s] > first ? cursors[s] - first : 0u;
        seg_end[s] = s * cap + (fill < cap ? fill : cap);
    }
}

// ---- owner slabs without their slack on the wire (multi-GPU, sharded.py) ------------------------------
// fill[i] of n slabs and its exclusive prefix sum, ONE block: cap != 0: in[] are the cursors of slabs that start at
// i * cap (a sender's own slabs); cap == 0: in[] ARE the fills (what a sender told the owner). start has n + 1 entries.
__global__ __launch_bounds__(1024) void fill_scan_kernel(const uint32_t *__restrict__ in, uint32_t n, uint32_t cap,
                                                         uint32_t *__restrict__ fills /* may be NULL */,
                                                         uint32_t *__restrict__ start, uint32_t *__restrict__ end /* may be NULL */)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (n + 1023u) / 1024u, i0 = tid * per, i1 = min(i0 + per, n);
    auto fill_of = [&](uint32_t i) {
        const uint32_t v = in[i];
        if (!cap)
            return v;
        const uint32_t first = i * cap;
        return v > first ? min(v - first, cap) : 0u;
    };
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; i++)
        mine += fill_of(i);
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t w = 0; w < wave; w++)
        run += s_wave[w];
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t f = fill_of(i);
        start[i] = run;
        if (fills)
            fills[i] = f;
        if (end)
            end[i] = run + f;
        run += f;
    }
    if ((i0 < n && i1 == n) || (n == 0 && tid == 0))      // (the one thread whose range reaches the end)
        start[n] = run;
}

// rows [p * cap, p * cap + fill[p]) of every slab -> dense[start[p] ..): one workgroup per 1024 rows of a slab.
// dense has room for dense_rows rows: a row behind that is not written (the caller's count and the cursors
// disagree -- cursors of a pack that gave up, say; the rows are not used then) and *over is raised.
__global__ __launch_bounds__(256) void slab_dense_rows_kernel(const uint4 *__restrict__ slabs, const uint32_t *__restrict__ start,
                                                              uint32_t cap, uint32_t chunks_per_slab, uint4 *__restrict__ dense,
                                                              uint64_t dense_rows, uint32_t *__restrict__ over)
{
    const uint32_t p = blockIdx.x / chunks_per_slab, chunk = blockIdx.x - p * chunks_per_slab;
    const uint32_t lo = start[p], fill = start[p + 1] - lo;
    if (chunk * 1024u >= fill)
        return;
    uint4 v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)      // (clamped, unconditional: in flight together)
        v[k] = slabs[(size_t)p * cap + min(chunk * 1024u + k * 256u + threadIdx.x, fill - 1)];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = chunk * 1024u + k * 256u + threadIdx.x;
        if (r < fill) {
            if ((uint64_t)lo + r < dense_rows)
                dense[(size_t)lo + r] = v[k];
            else if (over)
                *over = 1u;          // (the same value from every thread that gets here)
        }
    }
}

__global__ __launch_bounds__(1024) void slab_tile_starts_kernel(const uint32_t *__restrict__ seg_start,
                                                                const uint32_t *__restrict__ seg_end, uint32_t n_seg,
                                                                uint32_t *__restrict__ tile_start)
{
    fqd_partition::slab_tile_starts_body<fqd_partition::THREADS * RecordPolicy::EPT>(seg_start, seg_end, n_seg,
                                                                                    tile_start);
}

// slab mode of level 2: bucket b owns slots [b * cap, (b + 1) * cap); its cursor starts there
// three slab sets at once (the pack kernel's parts, level 2's buckets, the side path's slabs: one launch at the head
// of fqd_cluster_keys instead of three)
struct SlabSet {
    uint32_t n, cap;
    uint32_t *start, *cursor;
};
__global__ void slab_starts3_kernel(SlabSet a, SlabSet b, SlabSet c3, SlabSet d4, SlabSet e5, SlabSet f6)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const SlabSet sets[6] = {a, b, c3, d4, e5, f6};      // (a set of capacity 0 is a table of zeros)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (sets[k].start && i <= sets[k].n) {
            sets[k].start[i] = i * sets[k].cap;
            if (i < sets[k].n)
                sets[k].cursor[i] = i * sets[k].cap;
        }
    }
}

__global__ void slab_starts_kernel(uint32_t n_buckets, uint32_t cap, uint32_t *__restrict__ bucket_start,
                                   uint32_t *__restrict__ cursor)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets)
        return;
    bucket_start[b] = b * cap;
    if (b < n_buckets)
        cursor[b] = b * cap;
}

__global__ void tile_starts_kernel(const uint32_t *__restrict__ seg_start, uint32_t n_seg,
                                   uint32_t *__restrict__ tile_start)
{
    fqd_partition::tile_starts_body<fqd_partition::THREADS * RecordPolicy::EPT>(seg_start, n_seg, tile_start);
}

__device__ __forceinline__ uint32_t rec_tag(const uint4 &v)
{
    uint32_t h = (v.x ^ 0x9E3779B9u) * 0x85EBCA6Bu;
    h = (h ^ (h >> 15) ^ v.y) * 0xC2B2AE35u;
    h = (h ^ (h >> 13) ^ v.z) * 0x27D4EB2Fu;
    h ^= h >> 16;
    return h == DD_EMPTY ? 0u : h;
}

__global__ __launch_bounds__(DD_THREADS) void bucket_dedupe_kernel(
    const uint4 *__restrict__ part, const uint32_t *__restrict__ bucket_start /* n_buckets + 1 */,
    const uint32_t *__restrict__ bucket_end /* NULL, or slab mode: where each bucket's cursor stopped */,
    const uint32_t *__restrict__ weights, uint4 *__restrict__ tmp_rec, uint32_t *__restrict__ tmp_count,
    uint32_t *__restrict__ tmp_first, uint32_t *__restrict__ bucket_unique, uint32_t *__restrict__ overflow,
    fqd::HugeBuckets huge)
{
    __shared__ uint32_t s_tag[DD_SLOTS], s_x[DD_SLOTS], s_y[DD_SLOTS], s_z[DD_SLOTS], s_cnt[DD_SLOTS], s_min[DD_SLOTS];
    __shared__ uint32_t s_wave_tot[DD_THREADS / 64];
    __shared__ uint32_t s_chunk[FQD_HUGE_CHUNKS + 1];       // huge bucket, second pass: where each chunk's rows start
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    // huge.mode 1: the workgroup takes CHUNK b of a huge bucket ([huge.vlo[b], huge.vhi[b]), rows to tmp[vlo ...), their
    // number to huge.vunique[b]); mode 2: a bucket marked in huge.slot reads the ROWS its chunks left -- (key, count,
    // first index) triples -- instead of its reads: a key with a million copies is then streamed by FQD_HUGE_CHUNKS
    // workgroups side by side and merged by one (it was ONE workgroup's 4.4 ms).
    const bool chunk_pass = huge.mode == 1;
    uint32_t lo = chunk_pass ? huge.vlo[b] : bucket_start[b];
    uint32_t hi = chunk_pass ? huge.vhi[b] : bucket_start[b + 1];
    if (!chunk_pass && bucket_end)
        hi = min(hi, bucket_end[b]);
    const uint32_t hslot = huge.mode == 2 && huge.slot ? huge.slot[b] : 0u;
    const uint32_t out_lo = lo;
    if (hslot) {
        // logical positions 0 .. rows of all chunks: chunk c's rows lie at tmp[vlo[c] ...)
        if (tid == 0) {
            uint32_t acc = 0;
            for (uint32_t c = 0; c < FQD_HUGE_CHUNKS; c++) {
                s_chunk[c] = acc;
                acc += huge.vunique[(hslot - 1) * FQD_HUGE_CHUNKS + c];
            }
            s_chunk[FQD_HUGE_CHUNKS] = acc;
        }
    }
    for (uint32_t s = tid; s < DD_SLOTS; s += DD_THREADS)
        s_tag[s] = DD_EMPTY;
    __syncthreads();
    if (hslot) {
        lo = 0;
        hi = s_chunk[FQD_HUGE_CHUNKS];
    }
    // (second pass over a huge bucket) logical position -> the row's place in tmp
    auto row_at = [&](uint32_t p) {
        uint32_t a = 0, z = FQD_HUGE_CHUNKS;        // last chunk with s_chunk[a] <= p
        while (z - a > 1) {
            const uint32_t m = (a + z) >> 1;
            if (s_chunk[m] <= p)
                a = m;
            else
                z = m;
        }
        return huge.vlo[(hslot - 1) * FQD_HUGE_CHUNKS + a] + (p - s_chunk[a]);
    };

    bool full = false;
    // Four 256-read chunks are fetched before the first one is hashed: a bucket (~760 reads) then
    // pays ONE HBM latency instead of one per chunk (the kernel is latency bound: 42 buckets per
    // resident workgroup, ~10 us each).
    constexpr uint32_t DD_AHEAD = 4;
    for (uint32_t base0 = lo; base0 < hi; base0 += DD_AHEAD * DD_THREADS) {
      uint4 ahead[DD_AHEAD];
      uint32_t ahead_w[DD_AHEAD];
#pragma unroll
      for (uint32_t k = 0; k < DD_AHEAD; k++) {
          const uint32_t i = base0 + k * DD_THREADS + tid;
          ahead[k] = make_uint4(0, 0, 0, 0);
          ahead_w[k] = 0;
          if (i < hi && !hslot)
              ahead[k] = part[i];
          if (i < hi && hslot) {
              const uint32_t at = row_at(i);
              ahead[k] = tmp_rec[at];
              ahead[k].w = tmp_first[at];
              ahead_w[k] = tmp_count[at];
          }
      }
      if (!hslot) {
#pragma unroll
          for (uint32_t k = 0; k < DD_AHEAD; k++) {
              const uint32_t i = base0 + k * DD_THREADS + tid;
              ahead_w[k] = i < hi ? (weights ? weights[ahead[k].w] : 1u) : 0u;
          }
      }
      // All DD_AHEAD records of the thread go through the table TOGETHER: a round is "claim an
      // empty slot or stop at a slot whose tag matches" for every pending record, a barrier (the
      // parked records of the round are visible), "verify against the parked record and count, or
      // probe on". Usually one round per bucket -- two barriers instead of two per chunk.
      uint32_t tag[DD_AHEAD], slot[DD_AHEAD], probes[DD_AHEAD];
      bool pending[DD_AHEAD];
#pragma unroll
      for (uint32_t k = 0; k < DD_AHEAD; k++) {
          tag[k] = rec_tag(ahead[k]);
          slot[k] = (tag[k] * 0x9E3779B1u) >> 22;  // top 10 bits of a re-mix: DD_SLOTS == 1024
          probes[k] = 0;
          pending[k] = base0 + k * DD_THREADS + tid < hi;
      }
      // A HUGE bucket is a key with very many copies (its reads all hash here): the 64 records a wave holds are then
      // mostly ONE key, and 64 lanes adding to one LDS slot take turns -- one workgroup needed 4.4 ms for a key with
      // 10^6 copies. There the wave first merges its equal records: the first lane of a key keeps it with the summed
      // weight and the smallest read index, the others drop out. (A loop per DISTINCT key of the wave: not for
      // ordinary buckets, whose 64 records are 64 keys.)
      if (hi - lo > DD_HUGE || chunk_pass) {
#pragma unroll
          for (uint32_t k = 0; k < DD_AHEAD; k++) {
              unsigned long long left = __ballot(pending[k]);
              while (left) {
                  const int lead = __ffsll((long long)left) - 1;
                  const uint32_t kx = __shfl(ahead[k].x, lead), ky = __shfl(ahead[k].y, lead), kz = __shfl(ahead[k].z, lead);
                  const bool mine = pending[k] && ahead[k].x == kx && ahead[k].y == ky && ahead[k].z == kz;
                  const unsigned long long grp = __ballot(mine);
                  uint32_t w_sum = mine ? ahead_w[k] : 0u, id_min = mine ? ahead[k].w : 0xFFFFFFFFu;
                  for (int o = 32; o; o >>= 1) {
                      w_sum += __shfl_xor(w_sum, o);
                      id_min = min(id_min, (uint32_t)__shfl_xor(id_min, o));
                  }
                  if ((int)lane == lead) {
                      ahead_w[k] = w_sum;
                      ahead[k].w = id_min;
                  } else if (mine) {
                      pending[k] = false;
                  }
                  left &= ~grp;
              }
          }
      }
      bool any;
      do {
#pragma unroll
          for (uint32_t k = 0; k < DD_AHEAD; k++) {
              if (!pending[k])
                  continue;
              const uint4 v = ahead[k];
              for (;;) {
                  const uint32_t old = atomicCAS(&s_tag[slot[k]], DD_EMPTY, tag[k]);
                  if (old == DD_EMPTY) {
                      s_x[slot[k]] = v.x;
                      s_y[slot[k]] = v.y;
                      s_z[slot[k]] = v.z;
                      s_cnt[slot[k]] = ahead_w[k];
                      s_min[slot[k]] = v.w;
                      pending[k] = false;
                      break;
                  }
                  if (old == tag[k])
                      break;
                  slot[k] = (slot[k] + 1) & (DD_SLOTS - 1);
                  if (++probes[k] >= DD_SLOTS) {
                      full = true;
                      pending[k] = false;
                      break;
                  }
              }
          }
          __syncthreads();  // parked records of this round are visible
          any = false;
#pragma unroll
          for (uint32_t k = 0; k < DD_AHEAD; k++) {
              if (!pending[k])
                  continue;
              const uint4 v = ahead[k];
              if (s_x[slot[k]] == v.x && s_y[slot[k]] == v.y && s_z[slot[k]] == v.z) {
                  atomicAdd(&s_cnt[slot[k]], ahead_w[k]);
                  atomicMin(&s_min[slot[k]], v.w);
                  pending[k] = false;
              } else {  // same tag, different key: keep probing
                  slot[k] = (slot[k] + 1) & (DD_SLOTS - 1);
                  if (++probes[k] >= DD_SLOTS) {
                      full = true;
                      pending[k] = false;
                  } else {
                      any = true;
                  }
              }
          }
      } while (__syncthreads_or(any));
    }
    if (full)
        atomicOr(overflow, 1u);
    __syncthreads();

    // compact the live slots (count > 0: a key all of whose holders have weight 0 is not
    // in the trie) to tmp[lo ...): unique keys <= reads of the bucket, so they fit. (A CHUNK of a huge bucket keeps
    // its weight-0 rows: their read index may be the key's first holder -- the merge decides.)
    uint32_t mine = 0;
    for (uint32_t s = tid; s < DD_SLOTS; s += DD_THREADS)
        mine += (s_tag[s] != DD_EMPTY && (s_cnt[s] > 0 || chunk_pass)) ? 1u : 0u;
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t wv = 0; wv < wave; wv++)
        before += s_wave_tot[wv];
    uint32_t total = 0;
    for (uint32_t wv = 0; wv < DD_THREADS / 64; wv++)
        total += s_wave_tot[wv];
    uint32_t out = out_lo + before;
    for (uint32_t s = tid; s < DD_SLOTS; s += DD_THREADS)
        if (s_tag[s] != DD_EMPTY && (s_cnt[s] > 0 || chunk_pass)) {
            tmp_rec[out] = make_uint4(s_x[s], s_y[s], s_z[s], 0u);
            tmp_count[out] = s_cnt[s];
            tmp_first[out] = s_min[s];
            out++;
        }
    if (tid == 0)
        (chunk_pass ? huge.vunique : bucket_unique)[b] = total;
}

// buckets of more than DD_HUGE reads (exact bucket sizes): up to FQD_HUGE_MAX of them are cut into FQD_HUGE_CHUNKS
// chunks each (multiples of 1024 reads) for bucket_dedupe_kernel's chunk pass
__global__ void huge_plan_kernel(const uint32_t *__restrict__ bucket_start, uint32_t n_buckets, fqd::HugeBuckets huge,
                                 uint32_t *__restrict__ n_huge)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets)
        return;
    const uint32_t lo = bucket_start[b], hi = bucket_start[b + 1];
    uint32_t slot = 0;
    if (hi - lo > DD_HUGE) {
        const uint32_t h = atomicAdd(n_huge, 1u);
        if (h < FQD_HUGE_MAX) {
            slot = h + 1;
            const uint32_t per = (((hi - lo) + FQD_HUGE_CHUNKS - 1) / FQD_HUGE_CHUNKS + 1023u) & ~1023u;
            for (uint32_t c = 0; c < FQD_HUGE_CHUNKS; c++) {
                huge.vlo[h * FQD_HUGE_CHUNKS + c] = min(lo + c * per, hi);
                huge.vhi[h * FQD_HUGE_CHUNKS + c] = min(lo + (c + 1) * per, hi);
            }
        }
    }
    huge.slot[b] = (uint8_t)slot;
}

// (the chunks of slots nobody took: empty)
__global__ void huge_clear_kernel(fqd::HugeBuckets huge, uint32_t *__restrict__ n_huge)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < FQD_HUGE_MAX * FQD_HUGE_CHUNKS) {
        huge.vlo[v] = 0;
        huge.vhi[v] = 0;
        huge.vunique[v] = 0;
    }
    if (v == 0)
        *n_huge = 0;
}

// one wave per bucket: tmp[bucket_start[b] + j] -> out[uoff[b] + j], j < unique count
__global__ __launch_bounds__(256) void bucket_compact_kernel(
    const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ unique_incl /* inclusive scan */,
    uint32_t n_buckets, const uint4 *__restrict__ tmp_rec, const uint32_t *__restrict__ tmp_count,
    const uint32_t *__restrict__ tmp_first, IdSource read_ids, uint4 *__restrict__ urecs,
    uint32_t *__restrict__ ucounts, uint64_t *__restrict__ ufirst, fqd::SegHashOut sho)
{
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_buckets)
        return;
    const uint32_t end = unique_incl[b], begin = b ? unique_incl[b - 1] : 0u;
    const uint32_t src = bucket_start[b];
    const uint32_t n_unique = unique_incl[n_buckets - 1];    // (the host may not know it yet)
    // four rows per lane and step, their loads requested together (a bucket has ~210 unique keys:
    // usually one step)
    const uint32_t cnt = end - begin;
    for (uint32_t j0 = fqd_lane(); j0 < cnt; j0 += 4 * 64) {
        uint4 rec[4];
        uint32_t c4[4], f4[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) {
            const uint32_t j = j0 + t * 64;
            rec[t] = make_uint4(0, 0, 0, 0);
            c4[t] = f4[t] = 0;
            if (j < cnt) {
                rec[t] = tmp_rec[src + j];
                c4[t] = tmp_count[src + j];
                f4[t] = tmp_first[src + j];
            }
        }
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) {
            const uint32_t j = j0 + t * 64;
            if (j >= cnt)
                continue;
            // the neighbour search that follows groups the keys by hashes of their d + 1 segments: while
            // the record is in registers anyway (a later segment_hashes_kernel would read it once more)
            if (sho.nseg) {
                const uint32_t w[3] = {rec[t].x, rec[t].y, rec[t].z};
                for (uint32_t sg = 0; sg < sho.nseg; sg++)
                    sho.out[(size_t)sg * n_unique + begin + j] =
                        fqd_segment_hash(w, sho.planes, sho.kw, sho.len, sg, sho.nseg);
            }
            urecs[begin + j] = rec[t];
            ucounts[begin + j] = c4[t];
            ufirst[begin + j] = read_ids.packed_bits ? read_ids.from_packed(f4[t]) : read_ids.at(f4[t]);
        }
    }
}

__global__ void matrix_starts_kernel(const uint32_t *__restrict__ matrix_incl, uint32_t n_bins, uint32_t n_tiles,
                                     uint32_t *__restrict__ start)
{
    fqd_partition::matrix_starts_body(matrix_incl, n_bins, n_tiles, start);
}

__global__ void bucket_starts_kernel(const uint32_t *__restrict__ hist_incl, uint32_t n_buckets,
                                     uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ cursor)
{
    fqd_partition::bucket_starts_body(hist_incl, n_buckets, bucket_start, cursor);
}


// ---- compact records (fqd_internal.h Rec12): 12-byte items out of level 2 ------------------------------
// What differs from the uint4 pipeline above: level 2 WRITES Rec12 items (two key words + read index; a key with
// an N leaves through a side slab instead), the dedupe reads 12 instead of 16 bytes per read and keeps no third
// key word in its LDS table (20 KB, one more workgroup per CU), its tmp rows are ONE uint4 per unique key (a, b,
// count, first index), and the compaction turns the two words back into the three planes of the unique table,
// behind the few keys of the side path at its head.
// (The pack kernel itself writing Rec12 items into level 1 was measured: 0.98 ms instead of 0.57 -- its 4-record
// runs become 48 bytes at any 4-byte offset, and such writes cost more than the quarter of the bytes saves;
// level 2 writes 8-record runs.)
// SPILL: an item that finds its bucket's slab full goes, as the uint4 record it came from, to the spill list behind
// the side slabs (fqd::SideSlabs) instead of ending the attempt -- a context that has met keys with hundreds of
// copies (api.hip heavy_keys); the list is collapsed with the side path's keys.
template <bool SPILL>
struct CompactPolicyT {
    using Item = fqd::Rec12;
    static constexpr uint32_t EPT = FQD_SCATTER12_EPT;
    static constexpr uint32_t ROUNDS = FQD_SCATTER12_ROUNDS;
    static constexpr bool MAY_SKIP = true;
    static constexpr bool CAN_SPILL = SPILL;
    struct Source;
    static __device__ __forceinline__ uint32_t spill_reserve(const Source &s, uint32_t n) { return atomicAdd(s.side.spill_cursor, n); }
    static __device__ __forceinline__ void spill_write(const Source &s, uint32_t at, const fqd::Rec12 &v)
    {
        // (squeeze 1, a key without an N: back to its three planes, see rec12_planes)
        if (at < s.side.spill_cap)
            s.side.recs[(size_t)s.side.spill_at + at] = make_uint4(v.a & ~v.b, v.b & ~v.a, v.a & v.b, v.id);
        else
            atomicOr(s.side.overflow, 4u);           // (the spill list is full as well: the attempt ends)
    }
    struct Source {
        const uint4 *in;          // the pack kernel's records: planes + read index
        uint32_t squeeze;
        fqd::SideSlabs side;
        uint32_t route_mask;      // the key bits the bucket hash looks at (all ones: the whole key; else segment 0)
        // slabs received from several ranks (fqd_collapse_owner_slabs): segment / stamp_div is the sender, whose rank
        // by id base (stamp_map) is stamped above the read index the record carries, stamp_shift bits up
        uint32_t stamp_div, stamp_shift;
        const uint32_t *stamp_map;
    };
    static __device__ __forceinline__ uint32_t segment_tag(const Source &s, uint32_t seg)
    {
        if (!s.stamp_div)
            return 0u;
        const uint32_t sender = seg / s.stamp_div;
        return (s.stamp_map ? s.stamp_map[sender] : sender) << s.stamp_shift;
    }
    // (a record that left through the side slabs has id 0xFFFFFFFF and stays so: it was stamped on its way there)
    static __device__ __forceinline__ void apply_tag(fqd::Rec12 &v, uint32_t tag) { v.id |= tag; }
    // a record that left through the side slabs: not staged, not written
    static __device__ __forceinline__ bool skip(const fqd::Rec12 &v) { return v.id == 0xFFFFFFFFu; }
    using Raw = uint4;
    template <bool LEVEL1>
    static __device__ __forceinline__ uint4 fetch(const Source &s, uint32_t i) { return s.in[i]; }
    // The keys with an N of a tile (a few of 2048) are parked in LDS and leave for the side slabs behind ONE cursor
    // reservation per workgroup, after the tile itself has left (flush) -- a global atomic with its answer per rare
    // record, in the middle of the tile's loads, was a round trip on the critical path of four tiles in five.
    static constexpr uint32_t RARE_CAP = 32;     // (64 made the kernel 32 800 B of LDS: four workgroups per CU instead of five)
    struct Shared {
        uint32_t n;
        uint32_t base;
        uint4 rec[RARE_CAP];
    };
    static __device__ __forceinline__ void init_shared(Shared &sh, uint32_t tid)
    {
        if (tid == 0)
            sh.n = 0;
    }
    static __device__ __forceinline__ void flush(const Source &s, Shared &sh, uint32_t tid)
    {
        __syncthreads();
        const uint32_t n = min(sh.n, RARE_CAP);
        if (!n)
            return;
        const uint32_t slab = blockIdx.x & (s.side.n_slabs - 1);
        if (tid == 0)
            sh.base = atomicAdd(&s.side.cursor[slab], n);
        __syncthreads();
        if (tid < n) {
            const uint32_t pos = sh.base + tid;
            if (pos < (slab + 1) * s.side.cap)
                s.side.recs[pos] = sh.rec[tid];
            else
                atomicOr(s.side.overflow, 16u);
        }
    }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t load(const Source &s, uint32_t i, fqd::Rec12 &v)
    {
        Shared none;
        none.n = RARE_CAP;           // (no workgroup state here: straight to the slabs)
        return finish<LEVEL1>(s, i, s.in[i], v, true, 0u, none);
    }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t finish(const Source &s, uint32_t, const uint4 &r, fqd::Rec12 &v, bool valid,
                                                      uint32_t seg_tag, Shared &sh)
    {
        const bool squeeze = s.squeeze == 1;
        const bool rare = squeeze && (r.x & r.y) != 0u;
        if (rare && valid) {
            const uint4 stamped = make_uint4(r.x, r.y, r.z, r.w | seg_tag);
            const uint32_t k = atomicAdd(&sh.n, 1u);
            if (k < RARE_CAP) {
                sh.rec[k] = stamped;
            } else {
                // (more keys with an N in one tile than the parking space holds: one global atomic each)
                const uint32_t slab = blockIdx.x & (s.side.n_slabs - 1);
                const uint32_t pos = atomicAdd(&s.side.cursor[slab], 1u);
                if (pos < (slab + 1) * s.side.cap)
                    s.side.recs[pos] = stamped;
                else
                    atomicOr(s.side.overflow, 16u);
            }
        }
        // (the item's words as selects, not as assignments on the two sides of the branch above: with 16 items
        // per thread hipcc 7.2 merged those into "id = 0xFFFFFFFF" for every lane and the kernel skipped all items)
        v.a = squeeze ? r.x | r.z : r.x;
        v.b = squeeze ? r.y | r.z : r.y;
        v.id = rare ? 0xFFFFFFFFu : r.w;
        return rare ? 0u : fqd::fqd_route_hash(v.a, v.b, s.route_mask);
    }
    using KeyRaw = uint4;
    template <bool LEVEL1>
    static __device__ __forceinline__ uint4 key_fetch(const Source &s, uint32_t i) { return s.in[i]; }
    template <bool LEVEL1>
    static __device__ __forceinline__ uint32_t key_finish(const Source &s, uint32_t i, const uint4 &raw)
    {
        fqd::Rec12 v;
        Shared none;
        return finish<false>(s, i, raw, v, false, 0u, none);     // (a histogram pass puts nothing on the side path)
    }
};
using CompactPolicy = CompactPolicyT<false>;

// (the tiles of part_scatter12_kernel)
__global__ __launch_bounds__(1024) void slab_tile_starts12_kernel(const uint32_t *__restrict__ seg_start,
                                                                  const uint32_t *__restrict__ seg_end, uint32_t n_seg,
                                                                  uint32_t *__restrict__ tile_start)
{
    fqd_partition::slab_tile_starts_body<fqd_partition::THREADS * CompactPolicy::EPT>(seg_start, seg_end, n_seg,
                                                                                     tile_start);
}

template <uint32_t MAXB, uint32_t NT = 1, bool SPILL = false>
__global__ __launch_bounds__(fqd_partition::THREADS) void part_scatter12_kernel(
    typename CompactPolicyT<SPILL>::Source src, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ tile_start,
    uint32_t n_seg, uint32_t shift, uint32_t n_bins, uint32_t *__restrict__ cursor, fqd::Rec12 *__restrict__ out,
    uint32_t slab_cap, uint32_t *__restrict__ slab_overflow, const uint32_t *__restrict__ seg_end, uint32_t seg_shift,
    uint32_t seg_mask)
{
    fqd_partition::scatter_body<CompactPolicyT<SPILL>, false, MAXB, NT>(src, seg_start, tile_start, n_seg, shift, n_bins, cursor,
                                                                        out, slab_cap, slab_overflow, seg_end, seg_shift,
                                                                        seg_mask);
}

__device__ __forceinline__ uint32_t rec12_tag(uint32_t a, uint32_t b)
{
    uint32_t h = (a ^ 0x9E3779B9u) * 0x85EBCA6Bu;
    h = (h ^ (h >> 15) ^ b) * 0xC2B2AE35u;
    h = (h ^ (h >> 13)) * 0x27D4EB2Fu;
    h ^= h >> 16;
    return h == DD_EMPTY ? 0u : h;
}


// ---- the side table (see "the side path" below): helpers shared by its kernels and by the merging dedupe ----
constexpr uint32_t SIDE_EMPTY = 0xFFFFFFFFu, SIDE_BLOCK = 1024;
constexpr uint32_t SIDE_MAX_PROBES = 8192;     // (a table that full is given up: overflow bit 16)

__device__ __forceinline__ uint32_t side_hash(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t h = (x ^ 0x9E3779B9u) * 0x85EBCA6Bu;
    h = (h ^ (h >> 15) ^ y) * 0xC2B2AE35u;
    h = (h ^ (h >> 13) ^ z) * 0x27D4EB2Fu;
    return h ^ (h >> 16);
}

// one record (planes x, y, z; weight w, read index id) into the table; pos: the position in `side` of a record with
// this key, which claims the slot if the key is new
__device__ __forceinline__ void side_insert_one(const uint4 *__restrict__ side, uint32_t *table, uint32_t table_slots,
                                                uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t id, uint32_t pos,
                                                uint32_t *__restrict__ overflow)
{
    uint32_t slot = side_hash(x, y, z) & (table_slots - 1);
    const uint32_t max_probes = min(table_slots, SIDE_MAX_PROBES);
    for (uint32_t probes = 0; probes < max_probes; probes++) {
        uint32_t owner = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (owner == SIDE_EMPTY) {
            owner = atomicCAS(&table[slot], SIDE_EMPTY, pos);
            if (owner == SIDE_EMPTY)
                owner = pos;
        }
        bool same = owner == pos;
        if (!same) {
            const uint4 o = side[owner];
            same = o.x == x && o.y == y && o.z == z;
        }
        if (same) {
            atomicAdd(&table[table_slots + slot], w);
            atomicMin(&table[2 * table_slots + slot], id);
            return;
        }
        slot = (slot + 1) & (table_slots - 1);
    }
    atomicOr(overflow, 16u);
}

// the slot of a key in the COMPLETE table (no insert runs beside this), or SIDE_EMPTY
__device__ __forceinline__ uint32_t side_find(const uint4 *__restrict__ side, const uint32_t *__restrict__ table,
                                              uint32_t table_slots, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t slot = side_hash(x, y, z) & (table_slots - 1);
    const uint32_t max_probes = min(table_slots, SIDE_MAX_PROBES);
    for (uint32_t probes = 0; probes < max_probes; probes++) {
        const uint32_t owner = table[slot];
        if (owner == SIDE_EMPTY)
            return SIDE_EMPTY;
        const uint4 o = side[owner];
        if (o.x == x && o.y == y && o.z == z)
            return slot;
        slot = (slot + 1) & (table_slots - 1);
    }
    return SIDE_EMPTY;
}

// what the merging dedupe needs to know (bucket_dedupe12_kernel<true>)
struct SideMerge {
    const uint4 *side;
    uint32_t *table;
    uint32_t table_slots;
    const uint32_t *l1_over;     // [bucket >> l1_shift] != 0: a level-1 slab of the bucket's bin spilled
    uint32_t l1_shift;
};

// bucket_dedupe_kernel for Rec12 items. The two key words of an item ARE the key, so a slot is one 64-bit word
// and ONE 64-bit LDS compare-and-swap both claims an empty slot and recognises the key in a taken one: no tag,
// no parked record to verify against, hence no workgroup barrier between claiming and verifying -- the rounds of
// the uint4 kernel above (claim, barrier, verify, barrier) were what bounded this kernel (0.32 ms at config 3:
// 65 536 workgroups of ~10 us, parked on barriers half of their cycles). The all-ones key (32 x 'T') doubles as
// the EMPTY mark and is counted in a slot of its own behind the table. A live slot leaves as ONE uint4
// (a, b, count, first index) at tmp[lo + rank].
constexpr unsigned long long DD_EMPTY64 = ~0ull;

// MERGE (a context with a spill list, squeeze 1): copies of this bucket's keys may lie in the side table -- the spill
// list has been collapsed into it BEFORE this kernel -- when the bucket's slab overflowed at level 2 or a slab of its
// level-1 bin did in the pack kernel. Such a bucket looks every one of its keys up there; a key that is found adds its
// count and first index to the table's entry and leaves no row of its own.
// SLOTS: 1024, or 2048 for buckets of more than ~1000 reads (the owner's collapse of a 5-8 rank job: 2^15 buckets at
// most for 50 M received reads, ~1500 reads each -- mostly-unique data overfilled the 1024-slot table there and sent
// every rank back to the general way)
template <bool MERGE, uint32_t SLOTS = DD_SLOTS>
__global__ __launch_bounds__(DD_THREADS) void bucket_dedupe12_kernel(
    const fqd::Rec12 *__restrict__ part, const uint32_t *__restrict__ bucket_start,
    const uint32_t *__restrict__ bucket_end, const uint32_t *__restrict__ weights, uint4 *__restrict__ tmp,
    uint32_t *__restrict__ bucket_unique, uint32_t *__restrict__ overflow,
    uint32_t *__restrict__ group_total /* NULL, or [b >> 8] += unique keys of bucket b: with it the compaction finds
                                        * its offsets itself (bucket_compact12_kernel) and no scan runs in between */,
    SideMerge merge)
{
    __shared__ unsigned long long s_key[SLOTS + 1];
    __shared__ uint32_t s_cnt[SLOTS + 1], s_min[SLOTS + 1];
    __shared__ uint32_t s_wave_tot[DD_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // With group totals, consecutive workgroups take buckets of DIFFERENT groups (workgroup i: group i % G): the 256
    // buckets of one group in 256 workgroups that run side by side queued on the group's counter (+0.12 ms).
    uint32_t b = blockIdx.x;
    if (group_total && gridDim.x >= 512) {
        const uint32_t G = gridDim.x >> 8;
        b = ((blockIdx.x % G) << 8) + blockIdx.x / G;
    }
    const uint32_t lo = bucket_start[b];
    uint32_t hi = bucket_start[b + 1];
    if (bucket_end)
        hi = min(hi, bucket_end[b]);
    constexpr uint32_t DD_AHEAD = FQD_DD12_AHEAD;
    bool full = false;
    fqd::Rec12 ahead[DD_AHEAD];
    const uint32_t last = max(hi, lo + 1) - 1;       // (an empty bucket still owns its slab: part[lo] is readable)
    auto fetch = [&](uint32_t base0) {
#pragma unroll
        for (uint32_t k = 0; k < DD_AHEAD; k++)      // clamped, not conditional: the loads are in flight together
            ahead[k] = part[min(base0 + k * DD_THREADS + tid, last)];
    };
    fetch(lo);                                // (the table is cleared while the first items are on their way)
    for (uint32_t s = tid; s <= SLOTS; s += DD_THREADS) {
        s_key[s] = DD_EMPTY64;
        s_cnt[s] = 0u;
        s_min[s] = 0xFFFFFFFFu;
    }
    __syncthreads();
    for (uint32_t base0 = lo; base0 < hi; base0 += DD_AHEAD * DD_THREADS) {
        if (base0 != lo)
            fetch(base0);
        uint32_t ahead_w[DD_AHEAD];
#pragma unroll
        for (uint32_t k = 0; k < DD_AHEAD; k++) {
            const uint32_t i = base0 + k * DD_THREADS + tid;
            ahead_w[k] = i < hi ? (weights ? weights[ahead[k].id] : 1u) : 0u;
        }
        unsigned long long key[DD_AHEAD];
        uint32_t slot[DD_AHEAD];
        uint32_t pend = 0;
#pragma unroll
        for (uint32_t k = 0; k < DD_AHEAD; k++) {
            key[k] = ((unsigned long long)ahead[k].b << 32) | ahead[k].a;
            slot[k] = (rec12_tag(ahead[k].a, ahead[k].b) * 0x9E3779B1u) >> (SLOTS == 2048 ? 21 : 22);   // top 10 / 11 bits of a re-mix
            if (base0 + k * DD_THREADS + tid < hi) {
                if (key[k] == DD_EMPTY64) {
                    atomicAdd(&s_cnt[SLOTS], ahead_w[k]);
                    atomicMin(&s_min[SLOTS], ahead[k].id);
                } else {
                    pend |= 1u << k;
                }
            }
        }
        for (uint32_t probes = 0; pend && probes < SLOTS; probes++) {
            unsigned long long old[DD_AHEAD];
#pragma unroll
            for (uint32_t k = 0; k < DD_AHEAD; k++)       // the compare-and-swaps of all pending items in flight together
                if (pend >> k & 1u)
                    old[k] = atomicCAS(&s_key[slot[k]], DD_EMPTY64, key[k]);
#pragma unroll
            for (uint32_t k = 0; k < DD_AHEAD; k++)
                if (pend >> k & 1u) {
                    if (old[k] == DD_EMPTY64 || old[k] == key[k]) {
                        atomicAdd(&s_cnt[slot[k]], ahead_w[k]);
                        atomicMin(&s_min[slot[k]], ahead[k].id);
                        pend &= ~(1u << k);
                    } else {
                        slot[k] = (slot[k] + 1) & (SLOTS - 1);
                    }
                }
        }
        full = full || pend;
    }
    if (full)
        atomicOr(overflow, 1u);
    __syncthreads();
    constexpr uint32_t PER = (SLOTS + DD_THREADS) / DD_THREADS;     // slots per thread, table + 1
    if constexpr (MERGE) {
        const bool spilled = (bucket_end && bucket_end[b] > bucket_start[b + 1]) || merge.l1_over[b >> merge.l1_shift];
        if (spilled) {                              // (the same for every thread of the workgroup)
#pragma unroll
            for (uint32_t k = 0; k < PER; k++) {
                const uint32_t sl = tid + k * DD_THREADS;
                // (a taken slot, whatever its count: the first index of a weight-0 holder counts as well)
                if (sl <= SLOTS && (sl == SLOTS ? s_min[sl] != 0xFFFFFFFFu : s_key[sl] != DD_EMPTY64)) {
                    const unsigned long long kk = s_key[sl];      // (a, b) -> the three planes (rec12_planes, squeeze 1)
                    const uint32_t ka = (uint32_t)kk, kb = (uint32_t)(kk >> 32);
                    const uint32_t at = side_find(merge.side, merge.table, merge.table_slots, ka & ~kb, kb & ~ka, ka & kb);
                    if (at != SIDE_EMPTY) {
                        atomicAdd(&merge.table[merge.table_slots + at], s_cnt[sl]);
                        atomicMin(&merge.table[2 * merge.table_slots + at], s_min[sl]);
                        s_cnt[sl] = 0u;
                    }
                }
            }
            __syncthreads();
        }
    }

    // live slots (count > 0: a key all of whose holders have weight 0 is not in the trie), the slot behind the
    // table included
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t s = tid + k * DD_THREADS;
        mine += (s <= SLOTS && s_cnt[s] > 0) ? 1u : 0u;
    }
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t wv = 0; wv < wave; wv++)
        before += s_wave_tot[wv];
    uint32_t total = 0;
    for (uint32_t wv = 0; wv < DD_THREADS / 64; wv++)
        total += s_wave_tot[wv];
    uint32_t out = lo + before;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t s = tid + k * DD_THREADS;
        if (s <= SLOTS && s_cnt[s] > 0) {
            const unsigned long long kk = s_key[s];     // (the extra slot's key was never written: it IS the EMPTY pattern)
            tmp[out++] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), s_cnt[s], s_min[s]);
        }
    }
    if (tid == 0) {
        bucket_unique[b] = total;
        if (group_total && total)
            atomicAdd(&group_total[b >> 8], total);
    }
}

// the planes of a compact key (squeeze 1: (p0 | p2, p1 | p2) of an N-free "ACGNT" key; 2: the two planes themselves)
__device__ __forceinline__ void rec12_planes(uint32_t squeeze, uint32_t a, uint32_t b, uint32_t w[3])
{
    if (squeeze == 1) {
        w[0] = a & ~b;
        w[1] = b & ~a;
        w[2] = a & b;
    } else {
        w[0] = a;
        w[1] = b;
        w[2] = 0;
    }
}

#ifndef FQD_P0_ROWS
#define FQD_P0_ROWS 512
#endif
constexpr uint32_t P0_ROWS = FQD_P0_ROWS;   // rows of a bucket a wave holds in LDS for search pass 0 (more: the search does that pass)
constexpr uint32_t P0_WCAP = 256;   // edges buffered per wave (it keeps them across its buckets: one atomic on the job's edge counter per flush,
                                    // and ONE word takes ~88 atomics per microsecond -- a flush per bucket, 65 536 of them, was 0.6 ms)
constexpr uint32_t P0_SIDE_WAVES = 64;   // virtual buckets behind the last one: the segment hashes of the side path's keys

// one wave per bucket: tmp[bucket_start[b] + j] -> row side + uoff[b] + j of the unique table, side = *side_unique
// keys that launch_side_collapse has put at the head of the table already; the waves behind the last bucket
// write the segment hashes of those keys (their total was not known when they were written)
template <bool PASS0>
__global__ __launch_bounds__(256) void bucket_compact12_kernel(
    const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ unique_incl, uint32_t n_buckets,
    const uint4 *__restrict__ tmp, uint32_t squeeze, const uint32_t *__restrict__ side_unique,
    uint4 *__restrict__ urecs, uint32_t *__restrict__ ucounts, uint64_t *__restrict__ ufirst, fqd::SegHashOut sho,
    const uint32_t *__restrict__ bucket_unique, const uint32_t *__restrict__ group_total, fqd::Pass0 p0,
    IdSource read_ids /* packed_bits != 0: the rows' index words are (sender rank, read index on the sender) */)
{
    // search pass 0 (fqd::Pass0): the wave's rows, counting-sorted by six more bits of their route hash
    // (PASS0 = false: none of this is compiled in -- p0.mask is 0 then)
    __shared__ uint32_t s_pa[PASS0 ? 4 : 1][PASS0 ? P0_ROWS : 1], s_pb[PASS0 ? 4 : 1][PASS0 ? P0_ROWS : 1];
    __shared__ uint16_t s_pj[PASS0 ? 4 : 1][PASS0 ? P0_ROWS : 1];
    __shared__ uint32_t s_poff[PASS0 ? 4 : 1][66];
    __shared__ uint2 s_pe[PASS0 ? 4 : 1][PASS0 ? P0_WCAP : 1];
    __shared__ uint32_t s_pn[4];
    if (!PASS0)
        p0.mask = 0;
    const uint32_t side = side_unique ? *side_unique : 0u;
    const uint32_t lane = fqd_lane(), wave = threadIdx.x >> 6;
    // With pass 0 the grid is a fixed number of waves, each taking buckets w, w + waves, ...; else one wave per bucket.
    const uint32_t waves_total = (gridDim.x * blockDim.x) >> 6, wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t side_waves = side_unique && sho.nseg ? (p0.mask ? P0_SIDE_WAVES : waves_total - n_buckets) : 0u;
    unsigned long long reported = 0;
    if (p0.mask && lane == 0)
        s_pn[wave] = 0;
    // (group_total != NULL) the job's unique keys and, per lane, the totals of up to 256 groups: once per wave, not once
    // per bucket -- and a bucket's row count and slab start are requested one bucket AHEAD, so that its rows can be
    // requested at the top of its turn instead of behind a round trip for the count
    const uint32_t n_groups = max(n_buckets >> 8, 1u);
    uint32_t g_all = 0, gt[4] = {0, 0, 0, 0};
    if (group_total) {
        for (uint32_t g0 = 0; g0 < n_groups; g0 += 64) {
            const uint32_t g = g0 + lane, t = g < n_groups ? group_total[g] : 0u;
            g_all += t;
            if (g0 < 256)
                gt[g0 >> 6] = t;
        }
        for (int o = 32; o; o >>= 1)
            g_all += __shfl_xor(g_all, o);
    }
    uint32_t nxt_cnt = 0, nxt_src = 0;
    if (group_total && wave_global < n_buckets) {
        nxt_cnt = bucket_unique[wave_global];
        nxt_src = bucket_start[wave_global];
    }
  for (uint32_t b = wave_global; b < n_buckets + side_waves; b += waves_total) {
    // group_total != NULL: no scan of the bucket counts has run -- the wave adds up the totals of the groups of 256
    // buckets before its own and the counts of the buckets before it inside its group (2 KB of L2-resident words)
    uint32_t g_begin = 0;
    const uint32_t g_cnt = nxt_cnt, g_src = nxt_src;
    if (group_total) {
        const uint32_t my_group = min(b, n_buckets - 1) >> 8;
        if (b + waves_total < n_buckets) {
            nxt_cnt = bucket_unique[b + waves_total];
            nxt_src = bucket_start[b + waves_total];
        }
        if (n_groups <= 256) {
#pragma unroll
            for (uint32_t c4 = 0; c4 < 4; c4++)
                g_begin += c4 * 64 + lane < my_group ? gt[c4] : 0u;
        } else {
            for (uint32_t g0 = 0; g0 < n_groups; g0 += 64) {
                const uint32_t g = g0 + lane;
                g_begin += g < my_group ? group_total[g] : 0u;
            }
        }
        if (b < n_buckets) {
#pragma unroll
            for (uint32_t j0 = 0; j0 < 256; j0 += 64) {
                const uint32_t x = (my_group << 8) + j0 + lane;
                g_begin += x < b ? bucket_unique[x] : 0u;
            }
        }
    }
    const uint32_t n_unique = side + (group_total ? g_all : unique_incl[n_buckets - 1]);
    if (b >= n_buckets) {
        if (!sho.nseg)
            continue;
        const uint32_t n_waves = side_waves;
        for (uint32_t j = (b - n_buckets) * 64 + lane; j < side; j += n_waves * 64) {
            const uint4 r = urecs[j];
            const uint32_t w[3] = {r.x, r.y, r.z};
            for (uint32_t sg = sho.first; sg < sho.nseg; sg++)
                sho.out[(size_t)(sg - sho.first) * n_unique + j] = fqd_segment_hash(w, sho.planes, sho.kw, sho.len, sg, sho.nseg);
        }
        continue;
    }
    // (g_begin: still a per-lane share here -- added up BEHIND the row loads below, which need the count alone)
    const uint32_t begin_part = group_total ? g_begin : (b ? unique_incl[b - 1] : 0u);
    const uint32_t cnt = group_total ? g_cnt : unique_incl[b] - begin_part;
    const uint32_t src = group_total ? g_src : bucket_start[b];
    const bool pass0 = PASS0 && p0.mask != 0 && cnt <= p0.max_rows && cnt > 0;
    if (p0.mask && cnt > p0.max_rows && lane == 0)
        atomicOr(p0.flag, 1u);                 // more rows than the wave's LDS holds: the search does pass 0 itself
    // the bucket's first row: the lanes' shares added up (called behind the row loads: it needs their count alone)
    uint32_t begin = begin_part;
    auto sum_begin = [&]() {
        if (group_total)
            for (int o = 32; o; o >>= 1)
                begin += __shfl_xor(begin, o);
    };
    // one row -> the unique table (and the segment hashes of the search passes that follow)
    auto write_row = [&](const uint4 &row, uint32_t j) {
        uint32_t w[3];
        rec12_planes(squeeze, row.x, row.y, w);
        const uint32_t u = side + begin + j;
        if (sho.nseg)
            for (uint32_t sg = sho.first; sg < sho.nseg; sg++)
                sho.out[(size_t)(sg - sho.first) * n_unique + u] = fqd_segment_hash(w, sho.planes, sho.kw, sho.len, sg, sho.nseg);
        urecs[u] = make_uint4(w[0], w[1], w[2], 0u);
        ucounts[u] = row.z;
        ufirst[u] = read_ids.packed_bits ? read_ids.from_packed(row.w) : (uint64_t)row.w;
    };
    if (!pass0) {
        sum_begin();
        for (uint32_t j0 = lane; j0 < cnt; j0 += 4 * 64) {
            uint4 row[4];
#pragma unroll
            for (uint32_t t = 0; t < 4; t++)       // (clamped, unconditional: the four loads are in flight together)
                row[t] = tmp[src + min(j0 + t * 64, cnt - 1)];
#pragma unroll
            for (uint32_t t = 0; t < 4; t++)
                if (j0 + t * 64 < cnt)
                    write_row(row[t], j0 + t * 64);
        }
        continue;
    }
    // ---- with search pass 0 (fqd::Pass0). The wave is a chain of dependent steps, and 65 536 of them run: every
    // load is requested as early as its address is known -- the rows, the bucket's probe list and the probes' records
    // -- and the stores of the unique table go out while the edge list's reservation is on its way.
    constexpr uint32_t RPL = P0_ROWS / 64;      // rows per lane
    uint4 row[RPL];
#pragma unroll
    for (uint32_t q = 0; q < RPL; q++)
        row[q] = tmp[src + min(lane + 64 * q, cnt - 1)];
    const uint32_t np = p0.probe_n ? min(p0.probe_n[b], fqd::FQD_P0_PROBE_CAP) : 0u;
    const uint32_t my_probe = p0.probe_n ? p0.probe[(size_t)b * fqd::FQD_P0_PROBE_CAP + (lane % fqd::FQD_P0_PROBE_CAP)] : 0u;
    uint4 my_probe_rec = make_uint4(0, 0, 0, 0);      // lane l < np: the record of probe l
    if (lane < np)
        my_probe_rec = urecs[my_probe];
    sum_begin();
    const uint32_t sub_shift = 32u - p0.bucket_bits - 6u;
    s_poff[wave][lane] = 0;
    uint32_t sub_rank[RPL];                     // sub-bin << 16 | rank inside it
#pragma unroll
    for (uint32_t q = 0; q < RPL; q++) {
        sub_rank[q] = 0;
        if (lane + 64 * q < cnt) {
            const uint32_t sub = (fqd::fqd_route_hash(row[q].x, row[q].y, p0.mask) >> sub_shift) & 63u;
            sub_rank[q] = (sub << 16) | atomicAdd(&s_poff[wave][sub], 1u);
        }
    }
    // (the same wave wrote what it now reads: the LDS keeps a wave's accesses in order)
    {
        const uint32_t c = s_poff[wave][lane];
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += up;
        }
        s_poff[wave][lane] = incl - c;          // start of sub-bin `lane`
        if (lane == 63)
            s_poff[wave][64] = incl;            // = cnt
    }
#pragma unroll
    for (uint32_t q = 0; q < RPL; q++) {
        const uint32_t j = lane + 64 * q;
        if (j < cnt) {
            const uint32_t p = s_poff[wave][sub_rank[q] >> 16] + (sub_rank[q] & 0xFFFFu);
            s_pa[wave][p] = row[q].x;
            s_pb[wave][p] = row[q].y;
            s_pj[wave][p] = (uint16_t)(j | (sub_rank[q] >> 16) << 9);      // (row number < 512 | sub-bin: no second hash in the compare loop)
        }
    }
    static_assert(P0_ROWS <= 512, "s_pj: nine bits of row number");
    // Pairs go to the wave's LDS buffer; a full buffer is written out by the lanes that are active (converged) at the
    // call. Pairs behind the edge list's end are counted, not written: the search sees the count and starts over.
    auto write_out = [&](uint32_t have, uint32_t n, uint32_t rank, int leader) {
        unsigned long long g = 0;
        if ((int)lane == leader)
            g = atomicAdd(p0.edge_count, (unsigned long long)have);
        g = __shfl(g, leader);
        for (uint32_t e = rank; e < have; e += n)
            if (g + e < p0.edge_cap)
                reinterpret_cast<uint2 *>(p0.edges)[g + e] = s_pe[wave][e];
        reported += (int)lane == leader ? have : 0u;
    };
    auto note = [&](uint32_t u, uint32_t v) {
        const unsigned long long act = __ballot(1);
        const uint32_t n = (uint32_t)__popcll(act), rank = (uint32_t)__popcll(act & fqd_lanemask_lt());
        const int leader = __ffsll((long long)act) - 1;
        uint32_t have = __hip_atomic_load(&s_pn[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (have + n > P0_WCAP) {
            write_out(have, n, rank, leader);
            have = 0;
        }
        s_pe[wave][have + rank] = make_uint2(min(u, v), max(u, v));
        if ((int)lane == leader)
            __hip_atomic_store(&s_pn[wave], have + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const uint32_t uid0 = side + begin;
    // every row against the rows behind it in its sub-bin: same segment 0, at most d mismatches elsewhere
    for (uint32_t i = lane; i < cnt; i += 64) {
        const uint32_t ai = s_pa[wave][i], bi = s_pb[wave][i], ji = s_pj[wave][i];
        const uint32_t end = s_poff[wave][(ji >> 9) + 1];
        for (uint32_t k = i + 1; k < end; k++) {
            const uint32_t x = (ai ^ s_pa[wave][k]) | (bi ^ s_pb[wave][k]);     // mismatching positions
            if (!(x & p0.mask) && (uint32_t)__popc(x) <= p0.d)
                note(uid0 + (ji & 511u), uid0 + (s_pj[wave][k] & 511u));
        }
    }
    // the keys with an N that were routed here: against every row, and against each other
    for (uint32_t p = 0; p < np; p++) {
        const uint32_t sp = __shfl(my_probe, p);
        const uint4 pr = make_uint4(__shfl(my_probe_rec.x, p), __shfl(my_probe_rec.y, p), __shfl(my_probe_rec.z, p), 0u);
        for (uint32_t i = lane; i < cnt; i += 64) {
            uint32_t w[3];
            rec12_planes(squeeze, s_pa[wave][i], s_pb[wave][i], w);
            const uint32_t x = (pr.x ^ w[0]) | (pr.y ^ w[1]) | (pr.z ^ w[2]);
            if (!(x & p0.mask) && (uint32_t)__popc(x) <= p0.d)
                note(sp, uid0 + (s_pj[wave][i] & 511u));
        }
        if (lane > p && lane < np) {            // (lane l holds probe l)
            const uint32_t x = (pr.x ^ my_probe_rec.x) | (pr.y ^ my_probe_rec.y) | (pr.z ^ my_probe_rec.z);
            if (!(x & p0.mask) && (uint32_t)__popc(x) <= p0.d)
                note(sp, my_probe);
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < RPL; q++)
        if (lane + 64 * q < cnt)
            write_row(row[q], lane + 64 * q);
  }
    // what is left in the wave's edge buffer
    if (p0.mask) {
        const uint32_t left = __hip_atomic_load(&s_pn[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (left) {
            unsigned long long g_left = 0;
            if (lane == 0)
                g_left = atomicAdd(p0.edge_count, (unsigned long long)left);
            g_left = __shfl(g_left, 0);
            for (uint32_t e = lane; e < left; e += 64)
                if (g_left + e < p0.edge_cap)
                    reinterpret_cast<uint2 *>(p0.edges)[g_left + e] = s_pe[wave][e];
            reported += lane == 0 ? left : 0u;
        }
        if (p0.stats && lane == 0 && reported)
            atomicAdd(&p0.stats[wave_global % FQD_STAT_SLOTS].edges, reported);
    }
}

// ---- dedupe + compaction (+ search pass 0) of compact records in ONE kernel (fqd::CollapseSync) ----------------
// bucket_dedupe12_kernel and bucket_compact12_kernel<true> fused: a bucket's unique rows never leave the workgroup
// between the LDS table and the unique table -- no tmp rows (16 B written and read back per unique key), no second
// launch of 65 536 latency chains. What stood in the way is the row's place in the unique table, a prefix sum over
// ALL buckets. Here:
//   * The grid is PERSISTENT and resident at once: G workgroups in teams of TEAM, workgroup w taking buckets w, w + G,
//     w + 2 G, ... -- "round" r is the buckets [r G, (r + 1) G).
//   * Having counted a bucket's unique keys, a workgroup adds (1 << 32 | count) to its team's word of the round with ONE
//     64-bit atomic: the old value's low half is its offset inside the team (arrival order -- any order will do, rows of
//     a bucket stay together), the high half tells the last arriver that the team is complete. That one reserves the
//     team's rows with ONE addition to the job's row counter (G / TEAM per round: a few thousand on that word in all,
//     where one per bucket would be 65 536 -- 0.7 ms at ~88 per microsecond) and leaves the base for the others.
//     Teams take their place in the order they complete: no workgroup waits for any but the TEAM - 1 others of its team
//     (a prefix over the teams of a round made every round a barrier across all resident workgroups: 9 of 29 us).
//   * The rows of round r are WRITTEN half a round later, behind the insert phase of round r + 1 (they wait in LDS):
//     by then the team's base has all but always been written, and the word was requested at the top of the round.
//   * The next bucket's items are requested as soon as the current ones are in the table, so no wave waits for HBM.
//   * Search pass 0 runs on the rows while they are in LDS, as in bucket_compact12_kernel<true> but with one ROW per
//     thread instead of eight per lane; a pair is buffered with bucket-local row numbers (bit 31) until base is known.
// Every wait is bounded (a wall-clock limit, then *abort: all workgroups leave and the host runs the two kernels
// instead) -- a workgroup that is not resident (another process holding CUs) must not hang the others.
constexpr uint32_t FC_ECAP = 384, FC_FLUSH_AT = 128;     // pair buffer of a workgroup / flushed when more than that many are final
constexpr uint32_t FC_LOCAL = 0x80000000u;               // a pair's end that is still a bucket-local row number
constexpr uint32_t FC_AHEAD = 4;                         // 256-item chunks of a bucket held in registers (more: fetched in place)
#ifndef FQD_FC_SLOTS
#define FQD_FC_SLOTS 1024
#endif
#ifndef FQD_FC_WAVES
#define FQD_FC_WAVES 5
#endif
constexpr uint32_t FC_SLOTS = FQD_FC_SLOTS;              // slots of the LDS table (512: 20 KB of LDS per workgroup instead of 28)
constexpr uint32_t FC_ROWS = P0_ROWS;                    // unique keys of a bucket the kernel takes (more: *abort |= 2, the two kernels run)
__global__ __launch_bounds__(DD_THREADS, FQD_FC_WAVES) void bucket_collapse12_kernel(
    const fqd::Rec12 *__restrict__ part, const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ bucket_end,
    const uint32_t *__restrict__ weights, uint32_t n_buckets, uint32_t squeeze, const uint32_t *__restrict__ side_unique,
    uint4 *__restrict__ urecs, uint32_t *__restrict__ ucounts, uint64_t *__restrict__ ufirst, fqd::SegHashOut sho,
    fqd::Pass0 p0, IdSource read_ids, fqd::CollapseSync sync, uint32_t *__restrict__ overflow)
{
    constexpr uint32_t SLOTS = FC_SLOTS, PER = (SLOTS + DD_THREADS) / DD_THREADS;
    static_assert(SLOTS == 1024 || SLOTS == 512, "slot = top 10 / 9 bits");
    __shared__ unsigned long long s_key[SLOTS + 1];
    __shared__ uint32_t s_cnt[SLOTS + 1], s_min[SLOTS + 1];
    // a bucket's rows, densely and in pass 0's order: they wait here until their place in the unique table is known
    __shared__ uint32_t s_ra[FC_ROWS], s_rb[FC_ROWS], s_rc[FC_ROWS], s_rm[FC_ROWS];
    __shared__ uint4 s_probe[fqd::FQD_P0_PROBE_CAP];
    __shared__ uint2 s_edge[FC_ECAP];
    __shared__ uint32_t s_poff[66];
    __shared__ uint32_t s_wave_tot[DD_THREADS / 64];
    __shared__ uint32_t s_en, s_base, s_g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t TPR = sync.teams_per_round, TEAM = sync.team_size, G = TPR * TEAM, R = sync.n_rounds, w = blockIdx.x,
                   my_team = w / TEAM;
    const uint32_t side = side_unique ? *side_unique : 0u;
    const uint32_t sub_shift = 32u - p0.bucket_bits - 6u;

    auto bounds = [&](uint32_t b, uint32_t &lo, uint32_t &hi) {
        lo = hi = 0;
        if (b < n_buckets) {
            lo = bucket_start[b];
            hi = bucket_start[b + 1];
            if (bucket_end)
                hi = min(hi, bucket_end[b]);
        }
    };
    fqd::Rec12 cur[FC_AHEAD];
    auto fetch = [&](uint32_t lo, uint32_t hi) {
        const uint32_t last = max(hi, lo + 1) - 1;       // (an empty bucket still owns its slab: part[lo] is readable)
#pragma unroll
        for (uint32_t k = 0; k < FC_AHEAD; k++)          // clamped, not conditional: the loads are in flight together
            cur[k] = part[min(lo + k * DD_THREADS + tid, last)];
    };
    auto clear_table = [&]() {
        for (uint32_t sl = tid; sl <= SLOTS; sl += DD_THREADS) {
            s_key[sl] = DD_EMPTY64;
            s_cnt[sl] = 0u;
            s_min[sl] = 0xFFFFFFFFu;
        }
    };
    // FC_AHEAD items per thread into the table (bucket_dedupe12_kernel's probe loop)
    // (returns != 0 when an item found no slot)
    auto insert_chunk = [&](const fqd::Rec12 (&it)[FC_AHEAD], uint32_t base0, uint32_t hi) -> uint32_t {
        uint32_t wgt[FC_AHEAD];
        uint32_t slot[FC_AHEAD];
        uint32_t pend = 0;
#pragma unroll
        for (uint32_t k = 0; k < FC_AHEAD; k++) {
            const bool valid = base0 + k * DD_THREADS + tid < hi;
            wgt[k] = valid ? (weights ? weights[it[k].id] : 1u) : 0u;
            slot[k] = (rec12_tag(it[k].a, it[k].b) * 0x9E3779B1u) >> (SLOTS == 1024 ? 22 : 23);   // top 10 / 9 bits of a re-mix
            if (valid) {
                if ((it[k].a & it[k].b) == 0xFFFFFFFFu) {      // the all-ones key IS the EMPTY pattern: a slot of its own
                    atomicAdd(&s_cnt[SLOTS], wgt[k]);
                    atomicMin(&s_min[SLOTS], it[k].id);
                } else {
                    pend |= 1u << k;
                }
            }
        }
        for (uint32_t probes = 0; pend && probes < SLOTS; probes++) {
            unsigned long long old[FC_AHEAD];
#pragma unroll
            for (uint32_t k = 0; k < FC_AHEAD; k++)       // the compare-and-swaps of all pending items in flight together
                if (pend >> k & 1u)
                    old[k] = atomicCAS(&s_key[slot[k]], DD_EMPTY64, ((unsigned long long)it[k].b << 32) | it[k].a);
#pragma unroll
            for (uint32_t k = 0; k < FC_AHEAD; k++)
                if (pend >> k & 1u) {
                    if (old[k] == DD_EMPTY64 || old[k] == (((unsigned long long)it[k].b << 32) | it[k].a)) {
                        atomicAdd(&s_cnt[slot[k]], wgt[k]);
                        atomicMin(&s_min[slot[k]], it[k].id);
                        pend &= ~(1u << k);
                    } else {
                        slot[k] = (slot[k] + 1) & (SLOTS - 1);
                    }
                }
        }
        return pend;
    };

    // the rows that wait for their place (round prev_r) are s_ra .. s_rm[0, prev_total)
    uint32_t prev_total = 0, prev_off = 0, prev_r = 0;
    uint32_t have_prev = 0;
    uint32_t n_final = 0;             // pairs [0, n_final) of s_edge are job-wide uids; [n_final, s_en) wait for prev's base
    unsigned long long reported = 0;
    if (tid == 0)
        s_en = 0;

    // the pairs whose ends are final leave: one addition to the job's edge counter for all of them
    // (all threads, between barriers; n_final == s_en here)
    auto flush_edges = [&]() {
        if (tid == 0) {
            const unsigned long long g0 = atomicAdd(p0.edge_count, (unsigned long long)n_final);
            reported += n_final;
            s_g = (uint32_t)g0;                   // (64 bits through two LDS words; s_base has been read by everybody)
            s_base = (uint32_t)(g0 >> 32);
        }
        __syncthreads();
        const unsigned long long g = ((unsigned long long)s_base << 32) | s_g;
        for (uint32_t e = tid; e < n_final; e += DD_THREADS)
            if (g + e < p0.edge_cap)
                reinterpret_cast<uint2 *>(p0.edges)[g + e] = s_edge[e];
        __syncthreads();
        if (tid == 0)
            s_en = 0;
        n_final = 0;
        __syncthreads();
    };

    // rows of round prev_r -> the unique table; its pairs get their uids (returns false when a wait ran into its limit).
    // pre_v: the team's base word as it was at the top of this round (requested there, so that its round trip lies under
    // the insert phase); 0: the team's closer has not written it yet -- wait for it.
    auto write_prev = [&](uint32_t pre_v) -> bool {
        if (tid == 0) {
            const uint32_t *at = sync.base + (size_t)prev_r * TPR + my_team;
            uint32_t v = pre_v;
            bool aborted = false;
#ifdef FQD_FC_PROF
            const long long t00 = wall_clock64();
#endif
            if (!v) {
                const long long t0 = wall_clock64();
                for (uint32_t spin = 0;; spin++) {
                    v = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v)
                        break;
                    if ((spin & 15u) == 15u) {
                        if (wall_clock64() - t0 > (long long)sync.wait_ticks)
                            atomicOr(sync.abort, 1u);
                        if (__hip_atomic_load(sync.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                            aborted = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#ifdef FQD_FC_PROF
            if (sync.prof)
                atomicAdd(&sync.prof[10], (unsigned long long)(wall_clock64() - t00));
#endif
            s_base = aborted ? 0xFFFFFFFFu : side + (v - 1u) + prev_off;
        }
        __syncthreads();
        const uint32_t base = s_base;
        if (base == 0xFFFFFFFFu)
            return false;
        for (uint32_t j = tid; j < prev_total; j += DD_THREADS) {
            const uint32_t u = base + j;
            uint32_t wv[3];
            rec12_planes(squeeze, s_ra[j], s_rb[j], wv);
            if (sho.nseg)
                for (uint32_t sg = sho.first; sg < sho.nseg; sg++)      // (one array at most: the host saw to it)
                    sho.out[u] = fqd_segment_hash(wv, sho.planes, sho.kw, sho.len, sg, sho.nseg);
            urecs[u] = make_uint4(wv[0], wv[1], wv[2], 0u);
            ucounts[u] = s_rc[j];
            const uint32_t first_word = s_rm[j];
            ufirst[u] = read_ids.packed_bits ? read_ids.from_packed(first_word) : (uint64_t)first_word;
        }
        // the pairs of that bucket: bucket-local row numbers -> uids, smaller first
        const uint32_t en = min(s_en, FC_ECAP);
        for (uint32_t e = n_final + tid; e < en; e += DD_THREADS) {
            uint2 pr = s_edge[e];
            if (pr.x & FC_LOCAL)
                pr.x = base + (pr.x & ~FC_LOCAL);
            if (pr.y & FC_LOCAL)
                pr.y = base + (pr.y & ~FC_LOCAL);
            s_edge[e] = make_uint2(min(pr.x, pr.y), max(pr.x, pr.y));
        }
        n_final = en;
        __syncthreads();
        if (tid == 0 && s_en != en)
            s_en = en;
        if (n_final > FC_FLUSH_AT)
            flush_edges();
        return true;
    };

    auto note = [&](uint32_t x, uint32_t y) {
        const uint32_t at = atomicAdd(&s_en, 1u);
        if (at < FC_ECAP)
            s_edge[at] = make_uint2(x, y);
        else
            atomicOr(p0.flag, 1u);          // more pairs in one bucket than the buffer holds: the search does pass 0 itself
    };

    uint32_t np_next = 0, probe_next = 0;
    auto load_probes = [&](uint32_t b) {
        np_next = 0;
        if (p0.probe_n && b < n_buckets) {
            np_next = min(p0.probe_n[b], fqd::FQD_P0_PROBE_CAP);
            probe_next = p0.probe[(size_t)b * fqd::FQD_P0_PROBE_CAP + (tid % fqd::FQD_P0_PROBE_CAP)];   // (unconditional: no wait for the count)
        }
    };
    uint32_t lo, hi;
    bounds(w, lo, hi);
    fetch(lo, hi);
    load_probes(w);
    clear_table();
    uint32_t aborted = 0;
#ifdef FQD_FC_PROF
    // (a diagnostic build: thread 0's wall-clock ticks per phase of the round, summed over the workgroups into sync.prof)
    unsigned long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_last = wall_clock64();
#define FC_PROF(i) do { if (tid == 0) { const long long t_ = wall_clock64(); prof_acc[i] += (unsigned long long)(t_ - prof_last); prof_last = t_; } } while (0)
#else
#define FC_PROF(i) do { } while (0)
#endif
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t b = r * G + w;
        uint32_t lo_n, hi_n;
        bounds(r + 1 < R ? b + G : n_buckets, lo_n, hi_n);
        // the keys with an N that were routed to this bucket (fqd::Pass0 probe lists; the side path has finished): their
        // uids were requested a round ago
        const uint32_t np = np_next, my_probe = probe_next;
        load_probes(r + 1 < R ? b + G : n_buckets);
        uint32_t pre_v = 0;                       // (thread 0: has my team of the round before got its base yet?)
        if (have_prev && tid == 0)
            pre_v = __hip_atomic_load(sync.base + (size_t)prev_r * TPR + my_team, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                          // the table is clear; pass 0 of the bucket before is over
        FC_PROF(0);
        if (tid < 66)
            s_poff[tid] = 0;                      // (the sub-bin counters: used behind the next barrier)
        // ---- the bucket's items into the table
        uint32_t full = insert_chunk(cur, lo, hi);
        for (uint32_t base0 = lo + FC_AHEAD * DD_THREADS; base0 < hi; base0 += FC_AHEAD * DD_THREADS) {
            fqd::Rec12 more[FC_AHEAD];             // (a bucket of more than 1024 items: the rest where it lies)
            const uint32_t last = hi - 1;
#pragma unroll
            for (uint32_t k = 0; k < FC_AHEAD; k++)
                more[k] = part[min(base0 + k * DD_THREADS + tid, last)];
            full |= insert_chunk(more, base0, hi);
        }
        // ---- the next bucket's items and this bucket's probe records are on their way while the bucket is finished
        fetch(lo_n, hi_n);
        uint4 my_probe_rec = make_uint4(0, 0, 0, 0);
        if (tid < np)
            my_probe_rec = urecs[my_probe];
        FC_PROF(1);
        __syncthreads();                          // every item of the bucket is in the table
        FC_PROF(2);
        // ---- the rows of the round before leave the row arrays: their place is known by now
        if (have_prev && !write_prev(pre_v)) {
            aborted = 1;
            break;
        }
        FC_PROF(3);
        // ---- live slots (count > 0: a key all of whose holders have weight 0 is not in the trie): counted, and ranked
        // inside the sub-bins pass 0 sorts them by (six more bits of the route hash; LDS atomics)
        const bool sorted = p0.mask != 0;
        uint32_t live = 0, sub_rank[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t sl = tid + k * DD_THREADS;
            sub_rank[k] = 0;
            if (sl <= SLOTS && s_cnt[sl] > 0) {
                live |= 1u << k;
                if (sorted) {
                    const unsigned long long kk = s_key[sl];       // (the slot behind the table: never written, it IS the EMPTY pattern)
                    const uint32_t sub = (fqd::fqd_route_hash((uint32_t)kk, (uint32_t)(kk >> 32), p0.mask) >> sub_shift) & 63u;
                    sub_rank[k] = (sub << 16) | atomicAdd(&s_poff[sub], 1u);
                }
            }
        }
        const uint32_t mine = (uint32_t)__popc(live);
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += up;
        }
        if (lane == 63)
            s_wave_tot[wave] = incl;
        __syncthreads();
        FC_PROF(4);
        uint32_t before = incl - mine, total = 0;
        for (uint32_t wv = 0; wv < DD_THREADS / 64; wv++) {
            before += wv < wave ? s_wave_tot[wv] : 0u;
            total += s_wave_tot[wv];
        }
        // ---- the bucket's count to its team; what comes back (my offset inside the team; am I the last?) is looked at
        // at the END of the round: the atomic's round trip lies under pass 0
        unsigned long long pub_old = 0;
        if (tid == 0) {
            pub_old = __hip_atomic_fetch_add(sync.team + (size_t)r * TPR + my_team, (1ull << 32) | total, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
            if (total > FC_ROWS)
                atomicOr(sync.abort, 2u);         // more unique keys than the row arrays hold: the two kernels run instead
        }
        FC_PROF(5);
        if (full)
            atomicOr(overflow, 1u);               // more distinct keys than slots: the caller takes another way
        if (sorted && wave == 0) {
            const uint32_t c = s_poff[lane];
            uint32_t in2 = c;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(in2, o);
                if ((int)lane >= o)
                    in2 += up;
            }
            s_poff[lane] = in2 - c;                // start of sub-bin `lane`
            if (lane == 63)
                s_poff[64] = in2;                  // = total
        }
        __syncthreads();
        FC_PROF(6);
        // ---- the rows densely into the row arrays, in pass 0's order (or in slot order)
        {
            uint32_t j = before;
#pragma unroll
            for (uint32_t k = 0; k < PER; k++)
                if (live >> k & 1u) {
                    const uint32_t sl = tid + k * DD_THREADS;
                    const uint32_t pp = sorted ? s_poff[sub_rank[k] >> 16] + (sub_rank[k] & 0xFFFFu) : j;
                    if (pp < FC_ROWS) {
                        const unsigned long long kk = s_key[sl];
                        s_ra[pp] = (uint32_t)kk;
                        s_rb[pp] = (uint32_t)(kk >> 32);
                        s_rc[pp] = s_cnt[sl];
                        s_rm[pp] = s_min[sl];
                    }
                    j++;
                }
        }
        const bool pass0 = p0.mask != 0 && total > 0 && total <= min(p0.max_rows, FC_ROWS);
        if (p0.mask && total > p0.max_rows && tid == 0)
            atomicOr(p0.flag, 1u);                // more rows than pass 0 takes: the search does that pass itself
        if (pass0 && tid < np)
            s_probe[tid] = make_uint4(my_probe_rec.x, my_probe_rec.y, my_probe_rec.z, my_probe);
        __syncthreads();                          // the table's slots have been read: it can be cleared
        FC_PROF(7);
        clear_table();
        // ---- search pass 0: every row against the rows behind it in its sub-bin (same segment 0, at most d mismatches elsewhere)
        if (pass0) {
            for (uint32_t i = tid; i < total; i += DD_THREADS) {
                const uint32_t ai = s_ra[i], bi = s_rb[i];
                const uint32_t end = s_poff[((fqd::fqd_route_hash(ai, bi, p0.mask) >> sub_shift) & 63u) + 1];
                for (uint32_t k = i + 1; k < end; k++) {
                    const uint32_t x = (ai ^ s_ra[k]) | (bi ^ s_rb[k]);     // mismatching positions
                    if (!(x & p0.mask) && (uint32_t)__popc(x) <= p0.d)
                        note(FC_LOCAL | i, FC_LOCAL | k);
                }
            }
            // the keys with an N that were routed here: against every row, and against each other
            for (uint32_t p = 0; p < np; p++) {
                const uint4 pr = s_probe[p];
                for (uint32_t i = tid; i < total; i += DD_THREADS) {
                    uint32_t wv[3];
                    rec12_planes(squeeze, s_ra[i], s_rb[i], wv);
                    const uint32_t x = (pr.x ^ wv[0]) | (pr.y ^ wv[1]) | (pr.z ^ wv[2]);
                    if (!(x & p0.mask) && (uint32_t)__popc(x) <= p0.d)
                        note(pr.w, FC_LOCAL | i);
                }
                if (tid > p && tid < np) {          // (thread t holds probe t)
                    const uint32_t x = (pr.x ^ my_probe_rec.x) | (pr.y ^ my_probe_rec.y) | (pr.z ^ my_probe_rec.z);
                    if (!(x & p0.mask) && (uint32_t)__popc(x) <= p0.d)
                        note(pr.w, my_probe);
                }
            }
        }
        FC_PROF(8);
        prev_total = min(total, FC_ROWS);
        prev_r = r;
        have_prev = 1;
        if (tid == 0) {
            prev_off = (uint32_t)pub_old;
            if ((uint32_t)(pub_old >> 32) == TEAM - 1u) {
                // the team is complete and I am its closer: ONE reservation in the unique table for all of it (teams in
                // the order they complete -- any order will do), then the base for the team to read
                const uint32_t team_sum = (uint32_t)pub_old + total;
                const uint32_t at = __hip_atomic_fetch_add(sync.result, team_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync.base + (size_t)r * TPR + my_team, at + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        lo = lo_n;
        hi = hi_n;
    }
    if (have_prev && !aborted) {
        __syncthreads();                          // pass 0 of the last bucket is over
        if (!write_prev(0u))
            aborted = 1;
    }
    if (!aborted && p0.mask) {
        __syncthreads();
        if (n_final)
            flush_edges();
        if (p0.stats && tid == 0 && reported)
            atomicAdd(&p0.stats[w % FQD_STAT_SLOTS].edges, reported);
    }
#ifdef FQD_FC_PROF
    if (tid == 0 && sync.prof)
        for (int i = 0; i < 10; i++)
            atomicAdd(&sync.prof[i], prof_acc[i]);
#endif
    // the segment hashes of the side path's keys (head of the unique table; one array: its stride does not matter)
    if (!aborted && sho.nseg && side)
        for (uint32_t j = w * DD_THREADS + tid; j < side; j += G * DD_THREADS) {
            const uint4 rr = urecs[j];
            const uint32_t wv[3] = {rr.x, rr.y, rr.z};
            for (uint32_t sg = sho.first; sg < sho.nseg; sg++)
                sho.out[j] = fqd_segment_hash(wv, sho.planes, sho.kw, sho.len, sg, sho.nseg);
        }
}

// ---- the side path: keys with the rare symbol (uint4 records in `subs` slabs) -----------------------------
// An open-addressing table in global memory, one word triple per slot: the POSITION of the record that claimed
// the slot, the count and the smallest read index. A later record compares itself with the claimer's record in
// the slabs (written by an earlier kernel) -- no record is parked in the table, so nothing can be read half
// written. Then the live slots are counted per block of 1024 and written, in table order, to the head of the
// unique table.
__global__ void side_clear_kernel(uint32_t *__restrict__ table, uint32_t table_slots)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < table_slots) {
        table[i] = SIDE_EMPTY;                       // claimer
        table[table_slots + i] = 0u;                 // count
        table[2 * table_slots + i] = 0xFFFFFFFFu;    // first index
    }
}

// cursor[sub]: the pack kernel's cursor of side slab `sub`; it started at (first_part + sub) * cap and may have run
// past the slab's end (the pack kernel has raised the overflow flag then)
__global__ void side_insert_kernel(const uint4 *__restrict__ side, const uint32_t *__restrict__ cursor,
                                   uint32_t first_part, uint32_t cap, const uint32_t *__restrict__ weights,
                                   uint32_t *table, uint32_t table_slots, uint32_t *__restrict__ overflow)
{
    const uint32_t sub = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t filled = cursor[sub] - (first_part + sub) * cap;
    if (i >= min(filled, cap))
        return;
    const uint32_t pos = sub * cap + i;
    const uint4 v = side[pos];
    side_insert_one(side, table, table_slots, v.x, v.y, v.z, weights ? weights[v.w] : 1u, v.w, pos, overflow);
}

// The spill list (SideSlabs::spill_at ..): records the pack kernel or level 2 could not place because a slab was full
// -- mostly copies of a few keys with very many holders. A workgroup collapses chunks of 1024 records in an LDS table
// first (64-bit compare-and-swap on the two key words, like bucket_dedupe12_kernel) and sends one insert per distinct
// key of the chunk to the table in global memory: a record each would queue a million additions on one word for a
// key with a million copies (~88 per microsecond). Keys with an N and the all-T key (the LDS table's EMPTY pattern) go
// straight to the global table.
constexpr uint32_t SP_THREADS = 256, SP_R = 4, SP_CHUNK = SP_THREADS * SP_R, SP_SLOTS = 2 * SP_CHUNK;

__global__ __launch_bounds__(SP_THREADS) void side_insert_spill_kernel(
    const uint4 *__restrict__ side, uint32_t spill_at, uint32_t spill_cap, const uint32_t *__restrict__ spill_cursor,
    const uint32_t *__restrict__ weights, uint32_t *table, uint32_t table_slots, uint32_t *__restrict__ overflow)
{
    __shared__ unsigned long long s_key[SP_SLOTS];
    __shared__ uint32_t s_cnt[SP_SLOTS], s_min[SP_SLOTS], s_pos[SP_SLOTS];
    const uint32_t filled = min(*spill_cursor, spill_cap), tid = threadIdx.x;
    for (uint32_t lo = blockIdx.x * SP_CHUNK; lo < filled; lo += gridDim.x * SP_CHUNK) {
        uint4 v[SP_R];
#pragma unroll
        for (uint32_t r = 0; r < SP_R; r++)            // clamped, unconditional: in flight together
            v[r] = side[(size_t)spill_at + min(lo + r * SP_THREADS + tid, filled - 1)];
        for (uint32_t sl = tid; sl < SP_SLOTS; sl += SP_THREADS) {
            s_key[sl] = DD_EMPTY64;
            s_cnt[sl] = 0u;
            s_min[sl] = 0xFFFFFFFFu;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < SP_R; r++) {
            const uint32_t i = lo + r * SP_THREADS + tid;
            if (i >= filled)
                continue;
            const uint32_t w = weights ? weights[v[r].w] : 1u;
            const uint32_t a = v[r].x | v[r].z, b = v[r].y | v[r].z;
            const unsigned long long key = ((unsigned long long)b << 32) | a;
            if ((v[r].x & v[r].y) != 0u || key == DD_EMPTY64) {
                side_insert_one(side, table, table_slots, v[r].x, v[r].y, v[r].z, w, v[r].w, spill_at + i, overflow);
                continue;
            }
            uint32_t slot = (rec12_tag(a, b) * 0x9E3779B1u) >> 21;           // top 11 bits: SP_SLOTS == 2048
            for (;;) {                                 // (half as many records as slots: a free one is found)
                const unsigned long long old = atomicCAS(&s_key[slot], DD_EMPTY64, key);
                if (old == DD_EMPTY64)
                    s_pos[slot] = spill_at + i;        // (one winner per slot; read behind the barrier)
                if (old == DD_EMPTY64 || old == key) {
                    atomicAdd(&s_cnt[slot], w);
                    atomicMin(&s_min[slot], v[r].w);
                    break;
                }
                slot = (slot + 1) & (SP_SLOTS - 1);
            }
        }
        __syncthreads();
        for (uint32_t sl = tid; sl < SP_SLOTS; sl += SP_THREADS) {
            const unsigned long long kk = s_key[sl];
            if (kk != DD_EMPTY64) {
                const uint32_t ka = (uint32_t)kk, kb = (uint32_t)(kk >> 32);
                side_insert_one(side, table, table_slots, ka & ~kb, kb & ~ka, ka & kb, s_cnt[sl], s_min[sl], s_pos[sl],
                                overflow);
            }
        }
        __syncthreads();
    }
}

// (256 threads for 1024 slots: these kernels run beside the dedupe kernel, whose workgroups fill every CU -- a
// 1024-thread workgroup waited 0.27 ms for sixteen free wave slots on one CU)
constexpr uint32_t SIDE_THREADS = 256, SIDE_ROUNDS = SIDE_BLOCK / SIDE_THREADS;

__global__ __launch_bounds__(SIDE_THREADS) void side_count_kernel(const uint32_t *__restrict__ table, uint32_t table_slots,
                                                                  uint32_t *__restrict__ block_counts)
{
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < SIDE_ROUNDS; k++) {
        const uint32_t i = blockIdx.x * SIDE_BLOCK + k * SIDE_THREADS + threadIdx.x;
        mine += i < table_slots && table[i] != SIDE_EMPTY && table[table_slots + i] > 0 ? 1u : 0u;
    }
    __shared__ uint32_t s_part[SIDE_THREADS / 64];
    for (int o = 32; o; o >>= 1)
        mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63u) == 0)
        s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (uint32_t k = 0; k < SIDE_THREADS / 64; k++)
            n += s_part[k];
        block_counts[blockIdx.x] = n;
    }
}

__global__ __launch_bounds__(SIDE_THREADS) void side_emit_kernel(const uint4 *__restrict__ side,
                                                                 const uint32_t *__restrict__ table, uint32_t table_slots,
                                                                 const uint32_t *__restrict__ block_counts,
                                                                 uint4 *__restrict__ urecs, uint32_t *__restrict__ ucounts,
                                                                 uint64_t *__restrict__ ufirst,
                                                                 uint32_t *__restrict__ side_unique, fqd::Pass0 p0,
                                                                 IdSource read_ids)
{
    __shared__ uint32_t s_part[SIDE_THREADS / 64], s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // rows before this block: the counts of the blocks before it (at most a few thousand)
    uint32_t before = 0;
    for (uint32_t k = tid; k < blockIdx.x; k += SIDE_THREADS)
        before += block_counts[k];
    for (int o = 32; o; o >>= 1)
        before += __shfl_xor(before, o);
    if (lane == 0)
        s_part[wave] = before;
    __syncthreads();
    if (tid == 0) {
        uint32_t sum = 0;
        for (uint32_t k = 0; k < SIDE_THREADS / 64; k++)
            sum += s_part[k];
        s_base = sum;
    }
    __syncthreads();
    uint32_t base = s_base;
    // the block's slots in table order: round k holds slots [k * 256, (k + 1) * 256) of the block
    for (uint32_t k = 0; k < SIDE_ROUNDS; k++) {
        __syncthreads();                                         // (s_part is read above / in the round before)
        const uint32_t i = blockIdx.x * SIDE_BLOCK + k * SIDE_THREADS + tid;
        const bool live = i < table_slots && table[i] != SIDE_EMPTY && table[table_slots + i] > 0;
        const unsigned long long m = __ballot(live);
        if (lane == 0)
            s_part[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), round_total = 0;
        for (uint32_t w = 0; w < SIDE_THREADS / 64; w++) {
            rank += w < wave ? s_part[w] : 0u;
            round_total += s_part[w];
        }
        if (live) {
            const uint4 v = side[table[i]];
            const uint32_t u = base + rank;
            urecs[u] = make_uint4(v.x, v.y, v.z, 0u);
            ucounts[u] = table[table_slots + i];
            const uint32_t first_word = table[2 * table_slots + i];
            ufirst[u] = read_ids.packed_bits ? read_ids.from_packed(first_word) : (uint64_t)first_word;
            if (p0.mask) {
                // search pass 0 happens in the compaction of the bucket this key's segment 0 routes to (fqd::Pass0)
                const uint32_t bkt = fqd::fqd_route_hash(v.x | v.z, v.y | v.z, p0.mask) >> (32u - p0.bucket_bits);
                const uint32_t at = atomicAdd(&p0.probe_n[bkt], 1u);
                if (at < fqd::FQD_P0_PROBE_CAP)
                    p0.probe[(size_t)bkt * fqd::FQD_P0_PROBE_CAP + at] = u;
                else
                    atomicOr(p0.flag, 1u);
            }
        }
        base += round_total;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0)
        *side_unique = base;
}

}  // namespace

namespace fqd {

hipError_t launch_tile_starts(const uint32_t *seg_start, uint32_t n_seg, uint32_t *tile_start, hipStream_t st)
{
    tile_starts_kernel<<<1, 256, 0, st>>>(seg_start, n_seg, tile_start);
    return hipGetLastError();
}

// max_tiles bounds the tile count (n / TILE + n_seg); surplus blocks exit at once
hipError_t launch_part_hist(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                            const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                            uint32_t n_bins, uint32_t kw, uint32_t len, uint32_t *hist, hipStream_t st)
{
    if (n_bins > fqd_partition::MAX_BINS)
        return hipErrorInvalidValue;
    const RecordPolicy::Source src{hashes, reinterpret_cast<const uint4 *>(in), kw, len, IdSource(), 0u};
    if (level1)
        part_hist_kernel<true><<<max_tiles, fqd_partition::THREADS, 0, st>>>(src, seg_start, tile_start, n_seg, shift,
                                                                             n_bins, hist);
    else
        part_hist_kernel<false><<<max_tiles, fqd_partition::THREADS, 0, st>>>(src, seg_start, tile_start, n_seg, shift,
                                                                              n_bins, hist);
    return hipGetLastError();
}

hipError_t launch_part_scatter(bool level1, const uint32_t *hashes, const uint32_t *in, const uint32_t *seg_start,
                               const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                               uint32_t n_bins, uint32_t kw, uint32_t len, uint32_t *cursor, uint32_t *out,
                               hipStream_t st, IdSource packed, uint32_t slab_cap, uint32_t *slab_overflow,
                               const uint32_t *seg_end, uint32_t seg_shift, uint32_t seg_mask, uint32_t stamp_div)
{
    if (n_bins > fqd_partition::MAX_BINS)
        return hipErrorInvalidValue;
    const RecordPolicy::Source src{hashes, reinterpret_cast<const uint4 *>(in), kw, len, packed, stamp_div};
    uint4 *out4 = reinterpret_cast<uint4 *>(out);
    // few bins (the usual 256): small bin tables, one more workgroup per CU
#define FQD_SCATTER(L1, MB)                                                                                    \
    part_scatter_kernel<L1, MB><<<max_tiles, fqd_partition::THREADS, 0, st>>>(src, seg_start, tile_start, n_seg, \
                                                                              shift, n_bins, cursor, out4, slab_cap, \
                                                                              slab_overflow, seg_end, seg_shift, seg_mask)
    if (n_bins <= 256) {
        if (level1) FQD_SCATTER(true, 256); else FQD_SCATTER(false, 256);
    } else {
        if (level1) FQD_SCATTER(true, fqd_partition::MAX_BINS); else FQD_SCATTER(false, fqd_partition::MAX_BINS);
    }
#undef FQD_SCATTER
    return hipGetLastError();
}

uint32_t part_tile_size() { return fqd_partition::THREADS * RecordPolicy::EPT; }

hipError_t launch_owner_slab_bounds(const uint32_t *cursors, uint32_t n_senders, uint32_t parts_per_owner,
                                    uint32_t my_part, uint32_t cap, uint32_t *seg_start, uint32_t *seg_end,
                                    hipStream_t st)
{
    const uint32_t n = n_senders * parts_per_owner + 1;
    owner_slab_bounds_kernel<<<(n + 255) / 256, 256, 0, st>>>(cursors, n_senders, parts_per_owner, my_part, cap, seg_start,
                                                             seg_end);
    return hipGetLastError();
}

hipError_t launch_fill_scan(const uint32_t *in, uint32_t n, uint32_t cap, uint32_t *fills, uint32_t *start, uint32_t *end,
                            hipStream_t st)
{
    fill_scan_kernel<<<1, 1024, 0, st>>>(in, n, cap, fills, start, end);
    return hipGetLastError();
}

hipError_t launch_slab_dense_rows(const uint32_t *slabs, const uint32_t *start, uint32_t n_slabs, uint32_t cap,
                                  uint32_t *dense, hipStream_t st, uint64_t dense_rows, uint32_t *over)
{
    const uint32_t chunks = (cap + 1023u) / 1024u;
    if ((uint64_t)n_slabs * chunks > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    if (n_slabs)
        slab_dense_rows_kernel<<<n_slabs * chunks, 256, 0, st>>>(reinterpret_cast<const uint4 *>(slabs), start, cap, chunks,
                                                                reinterpret_cast<uint4 *>(dense), dense_rows, over);
    return hipGetLastError();
}

hipError_t launch_matrix_starts(const uint32_t *matrix_incl, uint32_t n_bins, uint32_t n_tiles, uint32_t *start,
                                hipStream_t st)
{
    matrix_starts_kernel<<<(n_bins + 1 + 255) / 256, 256, 0, st>>>(matrix_incl, n_bins, n_tiles, start);
    return hipGetLastError();
}

hipError_t launch_bucket_starts(const uint32_t *hist_incl, uint32_t n_buckets, uint32_t *bucket_start,
                                uint32_t *cursor, hipStream_t st)
{
    bucket_starts_kernel<<<(n_buckets + 1 + 255) / 256, 256, 0, st>>>(hist_incl, n_buckets, bucket_start, cursor);
    return hipGetLastError();
}

hipError_t launch_slab_starts3(uint32_t n1, uint32_t cap1, uint32_t *start1, uint32_t *cursor1, uint32_t n2, uint32_t cap2,
                               uint32_t *start2, uint32_t *cursor2, uint32_t n3, uint32_t cap3, uint32_t *start3,
                               uint32_t *cursor3, hipStream_t st, uint32_t n_zero, uint32_t *zero, uint32_t *zero_b,
                               uint32_t last_b, uint32_t *zero_c, uint32_t last_c)
{
    // zero[0 .. n_zero] = 0 on the way (the dedupe's group totals, the probe counts of search pass 0), and two more
    // small tables of zeros (the edge counter, the search statistics)
    uint32_t most = std::max(std::max(n1, zero ? n_zero : 0u), std::max(start2 ? n2 : 0u, start3 ? n3 : 0u));
    most = std::max(most, std::max(zero_b ? last_b : 0u, zero_c ? last_c : 0u));
    slab_starts3_kernel<<<(most + 1 + 255) / 256, 256, 0, st>>>(SlabSet{n1, cap1, start1, cursor1},
                                                               SlabSet{n2, cap2, start2, cursor2},
                                                               SlabSet{n3, cap3, start3, cursor3},
                                                               SlabSet{zero ? n_zero : 0u, 0u, zero, zero},
                                                               SlabSet{zero_b ? last_b : 0u, 0u, zero_b, zero_b},
                                                               SlabSet{zero_c ? last_c : 0u, 0u, zero_c, zero_c});
    return hipGetLastError();
}

hipError_t launch_slab_starts(uint32_t n_buckets, uint32_t cap, uint32_t *bucket_start, uint32_t *cursor, hipStream_t st)
{
    slab_starts_kernel<<<(n_buckets + 1 + 255) / 256, 256, 0, st>>>(n_buckets, cap, bucket_start, cursor);
    return hipGetLastError();
}

hipError_t launch_slab_tile_starts(const uint32_t *seg_start, const uint32_t *seg_end, uint32_t n_seg,
                                   uint32_t *tile_start, hipStream_t st, bool tiles12)
{
    if (tiles12)
        slab_tile_starts12_kernel<<<1, 1024, 0, st>>>(seg_start, seg_end, n_seg, tile_start);
    else
        slab_tile_starts_kernel<<<1, 1024, 0, st>>>(seg_start, seg_end, n_seg, tile_start);
    return hipGetLastError();
}

hipError_t launch_bucket_dedupe(const uint32_t *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                uint32_t n_buckets, const uint32_t *weights, uint32_t *tmp_rec, uint32_t *tmp_count,
                                uint32_t *tmp_first, uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st,
                                HugeBuckets huge)
{
    const uint4 *part4 = reinterpret_cast<const uint4 *>(part);
    uint4 *tmp4 = reinterpret_cast<uint4 *>(tmp_rec);
    if (huge.slot && !bucket_end) {
        // exact bucket sizes (a context that has met overfull slabs): huge buckets in chunks first
        uint32_t *n_huge = huge.vunique + FQD_HUGE_MAX * FQD_HUGE_CHUNKS;
        huge_clear_kernel<<<(FQD_HUGE_MAX * FQD_HUGE_CHUNKS + 255) / 256, 256, 0, st>>>(huge, n_huge);
        huge_plan_kernel<<<(n_buckets + 255) / 256, 256, 0, st>>>(bucket_start, n_buckets, huge, n_huge);
        huge.mode = 1;
        bucket_dedupe_kernel<<<FQD_HUGE_MAX * FQD_HUGE_CHUNKS, DD_THREADS, 0, st>>>(part4, bucket_start, nullptr, weights, tmp4,
                                                                                   tmp_count, tmp_first, bucket_unique,
                                                                                   overflow, huge);
        huge.mode = 2;
    } else {
        huge = HugeBuckets();
    }
    bucket_dedupe_kernel<<<n_buckets, DD_THREADS, 0, st>>>(part4, bucket_start, bucket_end, weights, tmp4, tmp_count,
                                                           tmp_first, bucket_unique, overflow, huge);
    return hipGetLastError();
}

uint32_t part_tile_size12() { return fqd_partition::THREADS * CompactPolicy::EPT; }

hipError_t launch_part_scatter12(const uint32_t *in, uint32_t squeeze, SideSlabs side, const uint32_t *seg_start,
                                 const uint32_t *tile_start, uint32_t n_seg, uint32_t max_tiles, uint32_t shift,
                                 uint32_t n_bins, uint32_t *cursor, Rec12 *out, hipStream_t st, uint32_t slab_cap,
                                 uint32_t *slab_overflow, const uint32_t *seg_end, uint32_t seg_shift, uint32_t route_mask,
                                 uint32_t seg_mask, uint32_t stamp_div, uint32_t stamp_shift, const uint32_t *stamp_map)
{
    if (n_bins > fqd_partition::MAX_BINS || (squeeze != 1 && squeeze != 2))
        return hipErrorInvalidValue;
    if (squeeze == 1 && (!side.recs || !side.cursor || !side.overflow || !side.cap || !side.n_slabs ||
                         (side.n_slabs & (side.n_slabs - 1))))
        return hipErrorInvalidValue;
    const CompactPolicy::Source src{reinterpret_cast<const uint4 *>(in), squeeze, side, route_mask ? route_mask : 0xFFFFFFFFu,
                                    stamp_div, stamp_shift, stamp_map};
    if (side.spill_cursor) {
        if (squeeze != 1 || !slab_cap || !side.spill_cap || stamp_div)
            return hipErrorInvalidValue;
        const CompactPolicyT<true>::Source spill_src{reinterpret_cast<const uint4 *>(in), squeeze, side,
                                                     route_mask ? route_mask : 0xFFFFFFFFu, stamp_div, stamp_shift, stamp_map};
        if (n_bins <= 256)
            part_scatter12_kernel<256, 1, true><<<max_tiles, fqd_partition::THREADS, 0, st>>>(
                spill_src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out, slab_cap, slab_overflow, seg_end,
                seg_shift, seg_mask);
        else
            part_scatter12_kernel<fqd_partition::MAX_BINS, 1, true><<<max_tiles, fqd_partition::THREADS, 0, st>>>(
                spill_src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out, slab_cap, slab_overflow, seg_end,
                seg_shift, seg_mask);
        return hipGetLastError();
    }
    // (a workgroup taking TWO consecutive tiles of a slab, both tiles' loads requested up front -- scatter_body's NT --
    // was measured: 0.426 against 0.390 ms; 98 VGPRs, four workgroups per CU instead of five)
    if (n_bins <= 256)
        part_scatter12_kernel<256><<<max_tiles, fqd_partition::THREADS, 0, st>>>(
            src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out, slab_cap, slab_overflow, seg_end, seg_shift,
            seg_mask);
    else
        part_scatter12_kernel<fqd_partition::MAX_BINS><<<max_tiles, fqd_partition::THREADS, 0, st>>>(
            src, seg_start, tile_start, n_seg, shift, n_bins, cursor, out, slab_cap, slab_overflow, seg_end, seg_shift,
            seg_mask);
    return hipGetLastError();
}

hipError_t launch_bucket_dedupe12(const Rec12 *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                  uint32_t n_buckets, const uint32_t *weights, uint32_t *tmp_rec,
                                  uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st, uint32_t *group_total,
                                  bool big_table)
{
    if (big_table)
        bucket_dedupe12_kernel<false, 2048><<<n_buckets, DD_THREADS, 0, st>>>(part, bucket_start, bucket_end, weights,
                                                                              reinterpret_cast<uint4 *>(tmp_rec),
                                                                              bucket_unique, overflow, group_total,
                                                                              SideMerge{});
    else
        bucket_dedupe12_kernel<false><<<n_buckets, DD_THREADS, 0, st>>>(part, bucket_start, bucket_end, weights,
                                                                        reinterpret_cast<uint4 *>(tmp_rec), bucket_unique,
                                                                        overflow, group_total, SideMerge{});
    return hipGetLastError();
}

// ... in a context with a spill list: behind launch_side_begin, before launch_side_finish
hipError_t launch_bucket_dedupe12_merge(const Rec12 *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                        uint32_t n_buckets, const uint32_t *weights, uint32_t *tmp_rec,
                                        uint32_t *bucket_unique, uint32_t *overflow, hipStream_t st, uint32_t *group_total,
                                        const uint4 *side, uint32_t *table, uint32_t table_slots, const uint32_t *l1_over,
                                        uint32_t l1_shift)
{
    if (!side || !table || !table_slots || !l1_over)
        return hipErrorInvalidValue;
    bucket_dedupe12_kernel<true><<<n_buckets, DD_THREADS, 0, st>>>(part, bucket_start, bucket_end, weights,
                                                                   reinterpret_cast<uint4 *>(tmp_rec), bucket_unique,
                                                                   overflow, group_total,
                                                                   SideMerge{side, table, table_slots, l1_over, l1_shift});
    return hipGetLastError();
}

hipError_t launch_bucket_compact12(const uint32_t *bucket_start, const uint32_t *unique_incl, uint32_t n_buckets,
                                   const uint32_t *tmp_rec, uint32_t squeeze, const uint32_t *side_unique,
                                   uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst, hipStream_t st,
                                   SegHashOut seg_hashes, const uint32_t *bucket_unique, const uint32_t *group_total,
                                   Pass0 pass0, IdSource read_ids)
{
    // one wave per bucket + 64 waves for the segment hashes of the side path's keys; with search pass 0: as many waves
    // as the GPU holds at once (256 CUs x 5 workgroups of 4), each taking every waves-th bucket
    uint64_t threads = ((uint64_t)n_buckets + (side_unique && seg_hashes.nseg ? 64 : 0)) * 64;
    if (pass0.mask) {
        threads = std::min<uint64_t>(threads, (uint64_t)256 * 4 * 256);      // (107 VGPRs: four workgroups per CU)
        bucket_compact12_kernel<true><<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(
            bucket_start, unique_incl, n_buckets, reinterpret_cast<const uint4 *>(tmp_rec), squeeze, side_unique,
            reinterpret_cast<uint4 *>(urecs), ucounts, ufirst, seg_hashes, bucket_unique, group_total, pass0, read_ids);
        return hipGetLastError();
    }
    bucket_compact12_kernel<false><<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(
        bucket_start, unique_incl, n_buckets, reinterpret_cast<const uint4 *>(tmp_rec), squeeze, side_unique,
        reinterpret_cast<uint4 *>(urecs), ucounts, ufirst, seg_hashes, bucket_unique, group_total, pass0, read_ids);
    return hipGetLastError();
}

uint32_t pass0_max_rows() { return P0_ROWS; }

uint32_t side_table_words(uint32_t table_slots) { return 3 * table_slots + (table_slots + SIDE_BLOCK - 1) / SIDE_BLOCK; }

uint32_t collapse12_resident()
{
    static long resident = -1;          // (one kind of device per process: gfx950)
    if (resident < 0) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bucket_collapse12_kernel, DD_THREADS, 0) != hipSuccess)
            return 0;
        resident = (long)per_cu * prop.multiProcessorCount;
    }
    return (uint32_t)std::max<long>(resident, 0);
}

hipError_t launch_bucket_collapse12(const Rec12 *part, const uint32_t *bucket_start, const uint32_t *bucket_end,
                                    uint32_t n_buckets, const uint32_t *weights, uint32_t squeeze,
                                    const uint32_t *side_unique, uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst,
                                    hipStream_t st, SegHashOut seg_hashes, Pass0 pass0, IdSource read_ids,
                                    CollapseSync sync, uint32_t *overflow)
{
    const uint64_t grid = (uint64_t)sync.team_size * sync.teams_per_round;
    if (!grid || grid > collapse12_resident() || (uint64_t)sync.n_rounds * grid < n_buckets ||
        (seg_hashes.nseg && seg_hashes.nseg - seg_hashes.first > 1))
        return hipErrorInvalidValue;
    bucket_collapse12_kernel<<<(unsigned)grid, DD_THREADS, 0, st>>>(
        part, bucket_start, bucket_end, weights, n_buckets, squeeze, side_unique, reinterpret_cast<uint4 *>(urecs), ucounts,
        ufirst, seg_hashes, pass0, read_ids, sync, overflow);
    return hipGetLastError();
}


hipError_t launch_side_collapse(const uint4 *side, const uint32_t *cursor, uint32_t first_part, uint32_t subs, uint32_t cap,
                                const uint32_t *weights, uint32_t *table, uint32_t table_slots, uint32_t *block_counts,
                                uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst, uint32_t *side_unique,
                                uint32_t *overflow, hipStream_t st, Pass0 pass0, IdSource read_ids)
{
    if (!table_slots || (table_slots & (table_slots - 1)) || !subs || !cap)
        return hipErrorInvalidValue;
    const uint32_t blocks = (table_slots + SIDE_BLOCK - 1) / SIDE_BLOCK;
    side_clear_kernel<<<(table_slots + 255) / 256, 256, 0, st>>>(table, table_slots);
    side_insert_kernel<<<dim3((cap + 255) / 256, subs), 256, 0, st>>>(side, cursor, first_part, cap, weights, table,
                                                                     table_slots, overflow);
    side_count_kernel<<<blocks, SIDE_THREADS, 0, st>>>(table, table_slots, block_counts);
    side_emit_kernel<<<blocks, SIDE_THREADS, 0, st>>>(side, table, table_slots, block_counts,
                                                    reinterpret_cast<uint4 *>(urecs), ucounts, ufirst, side_unique, pass0,
                                                    read_ids);
    return hipGetLastError();
}

// The side path in two halves around the dedupe of a context with a spill list (SideSlabs::spill_cursor): the keys of
// the side slabs and the spill list into the table -- then the dedupe merges what it finds there (MERGE) -- then the
// table's live slots to the head of the unique table.
hipError_t launch_side_begin(SideSlabs side, const uint32_t *weights, uint32_t *table, uint32_t table_slots, hipStream_t st)
{
    if (!table_slots || (table_slots & (table_slots - 1)) || !side.n_slabs || !side.cap || !side.spill_cursor)
        return hipErrorInvalidValue;
    side_clear_kernel<<<(table_slots + 255) / 256, 256, 0, st>>>(table, table_slots);
    side_insert_kernel<<<dim3((side.cap + 255) / 256, side.n_slabs), 256, 0, st>>>(side.recs, side.cursor, 0, side.cap, weights,
                                                                                 table, table_slots, side.overflow);
    const uint32_t chunks = (side.spill_cap + SP_CHUNK - 1) / SP_CHUNK;
    side_insert_spill_kernel<<<std::max(1u, std::min(chunks, 4096u)), SP_THREADS, 0, st>>>(
        side.recs, side.spill_at, side.spill_cap, side.spill_cursor, weights, table, table_slots, side.overflow);
    return hipGetLastError();
}

hipError_t launch_side_finish(const uint4 *side, uint32_t *table, uint32_t table_slots, uint32_t *block_counts,
                              uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst, uint32_t *side_unique, hipStream_t st,
                              Pass0 pass0, IdSource read_ids)
{
    const uint32_t blocks = (table_slots + SIDE_BLOCK - 1) / SIDE_BLOCK;
    side_count_kernel<<<blocks, SIDE_THREADS, 0, st>>>(table, table_slots, block_counts);
    side_emit_kernel<<<blocks, SIDE_THREADS, 0, st>>>(side, table, table_slots, block_counts,
                                                    reinterpret_cast<uint4 *>(urecs), ucounts, ufirst, side_unique, pass0,
                                                    read_ids);
    return hipGetLastError();
}

hipError_t launch_bucket_compact(const uint32_t *bucket_start, const uint32_t *unique_incl, uint32_t n_buckets,
                                 const uint32_t *tmp_rec, const uint32_t *tmp_count, const uint32_t *tmp_first,
                                 IdSource read_ids, uint32_t *urecs, uint32_t *ucounts, uint64_t *ufirst,
                                 hipStream_t st, SegHashOut seg_hashes)
{
    const uint64_t threads = (uint64_t)n_buckets * 64;
    bucket_compact_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(
        bucket_start, unique_incl, n_buckets, reinterpret_cast<const uint4 *>(tmp_rec), tmp_count, tmp_first, read_ids,
        reinterpret_cast<uint4 *>(urecs), ucounts, ufirst, seg_hashes);
    return hipGetLastError();
}

}  // namespace fqd
