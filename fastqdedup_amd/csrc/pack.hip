// pack.hip -- stage 1: ASCII keys -> bit-plane records + 32-bit key hashes.
//
// Replaces the storage half of TrieNode_AddSequence (reference
// _triemodule.c:222-288): instead of one trie node per base, a key becomes K
// bit planes of ceil(len/32) words (fqd_internal.h).
//
// Kernel shape (HBM-bound: reads every input byte exactly once with coalesced
// 16-byte loads, writes each record once with 16-byte stores):
//   phase A  bytes -> K bit streams of the block's contiguous byte range in LDS.
//            Alphabets of <= 8 symbols (all DNA input) take the SWAR path: every
//            lane turns its own 2 x 16 bytes into 2 x 16 stream bits per plane with
//            no cross-lane traffic (v_perm_b32 as an 8-entry byte table, two
//            delta-swaps as the bit transpose). Larger alphabets take the table
//            path: 256-entry code table in LDS, one wave64 ballot per plane and row.
//   phase B  one thread per (key, 32-base word): funnel-shift the key's bits
//            out of the streams into the record tile in LDS.
//   phase C  one thread per key hashes its record from LDS; the tile is then
//            copied to HBM with uint4 stores.
#include "fqd_internal.h"

#ifndef FQD_PACK_NSUB
#define FQD_PACK_NSUB 2   // tiles per workgroup of the fused pack (config 3: 1: 0.56-0.58 ms, 2: 0.51-0.52, 3: 0.58 -- 107 VGPRs)
#endif
#ifndef FQD_PACK_PREFETCH
#define FQD_PACK_PREFETCH 2   // rows of key bytes requested ahead of the one being converted (1: 0.508, 2: 0.501 ms at config 3)
#endif
namespace {

constexpr int PACK_THREADS = 256;
constexpr int PACK_WAVES = PACK_THREADS / FQD_WAVE;

__global__ __launch_bounds__(256) void scan_bytes_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes,
                                                         uint32_t *__restrict__ present)
{
    __shared__ uint32_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t n16 = n_bytes / 16;
    const uint4 *v = reinterpret_cast<const uint4 *>(bytes);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 q = v[i];
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            seen[w[j] & 0xFF] = 1;
            seen[(w[j] >> 8) & 0xFF] = 1;
            seen[(w[j] >> 16) & 0xFF] = 1;
            seen[w[j] >> 24] = 1;
        }
    }
    if (blockIdx.x == 0)
        for (uint64_t i = n16 * 16 + threadIdx.x; i < n_bytes; i += blockDim.x)
            seen[bytes[i]] = 1;
    __syncthreads();
    if (seen[threadIdx.x])
        present[threadIdx.x] = 1;
}

__global__ void scan_lens_kernel(const uint64_t *__restrict__ offsets, uint64_t n, uint32_t *__restrict__ minmax)
{
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // (a wave's lanes stay in the loop TOGETHER -- the condition looks at the wave's first key: a lane past the last key
    // still hands its offset, offsets[n], to its left neighbour)
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 - (threadIdx.x & 63u) < n; i0 += 4 * stride) {
        uint64_t a[4], b[4];                      // four lengths per step, their loads in flight together
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint64_t i = i0 + t * stride;
            a[t] = i <= n ? offsets[i] : 0;       // (offsets has n + 1 entries)
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            // a key's end is its right neighbour's start: one load per key, the last lane of a wave fetches its own
            // (every offset was read twice: 0.21 ms for 50 M keys, on the way to the first read-back of the job)
            const uint64_t i = i0 + t * stride;
            const uint64_t nb = (uint64_t)__shfl_down((unsigned long long)a[t], 1);
            b[t] = (threadIdx.x & 63u) == 63u ? (i < n ? offsets[i + 1] : 0) : nb;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (i0 + t * stride >= n)
                continue;
            const uint64_t l = b[t] - a[t];
            const uint32_t l32 = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
            lo = min(lo, l32);
            hi = max(hi, l32);
        }
    }
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    // one pair of atomics per workgroup (one per wave: 16 K atomics on two words, ~36 ns each -- most of the
    // kernel's 1 ms at 50 M keys)
    __shared__ uint32_t s_lo[4], s_hi[4];
    if (fqd_lane() == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < blockDim.x / 64; w++) {
            lo = min(lo, s_lo[w]);
            hi = max(hi, s_hi[w]);
        }
        atomicMin(&minmax[0], lo);
        atomicMax(&minmax[1], hi);
    }
}

// Perfect hash of the alphabet into 8 slots, found on the host: slot(c) =
// ((c >> s1) ^ (c >> s2)) & 7 (s2 == PACK_NO_SHIFT: one shift only). code_tbl / char_tbl
// hold, per slot, the code of its symbol and the symbol itself (0x80 for an empty slot,
// which no ASCII byte equals).
constexpr uint32_t PACK_NO_SHIFT = 0xFFFFFFFFu;
constexpr uint32_t PACK_RARE_CAP = 32;      // keys with an N a workgroup parks in LDS (2048 keys hold ~7 at an N rate of 1e-4 per base)
struct PackHash {
    uint32_t s1, s2;
    uint64_t code_tbl, char_tbl;
    uint32_t fill;  // a valid symbol in all 4 bytes: stands in for bytes past the buffer end
};

// LDS carve (u32 words): lut[64] | scratch[PACK_WAVES][64] | planes[K][plane_words] | tile[kpb*stride]
// A "row" is 2048 bytes (64 lanes x 2 groups x 16 bytes) = 64 stream dwords per plane.
// FUSED (one-uint4 records, fixed length, kpb <= 1024): instead of writing the records in read order
// (phase C), the block partitions its tile by the top hash bits straight into level 1 of the LDS
// collapse (collapse_lds.hip) -- the records never make the round trip through HBM in read order.
// No histogram pass can have run (it would have to read the key bytes once more), so level 1 works
// in slab mode like level 2: part (bin, sub) owns part_out[(bin * subs + sub) * cap, +cap), and a
// tile reserves its share of a bin with ONE atomic on that part's cursor. `subs` parts per bin
// (tile t feeds sub t % subs) spread those atomics: with one cursor per bin the 48 K tiles of a
// 50 M-read job queue up on a few hundred addresses.
template <int K, bool SWAR, int FUSED = 0 /* 1: partitioned output; 2: ... with the spill list (PackScatter::spill) */>
__global__ __launch_bounds__(PACK_THREADS) void pack_kernel(
    const uint8_t *__restrict__ bytes, uint64_t n_bytes, const uint64_t *__restrict__ offsets, uint64_t n,
    uint32_t fixed_len, KeyShape sh, uint32_t kpb, uint32_t plane_words, const uint8_t *__restrict__ lut_g,
    PackHash ph, uint32_t *__restrict__ recs, uint32_t *__restrict__ lens, uint32_t *__restrict__ hashes,
    uint32_t *__restrict__ owners, fqd::OwnerRule rule, uint32_t *__restrict__ bad_flag, fqd::PackScatter fs)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *lut32 = smem;                                   // 256 bytes
    uint32_t *scratch = smem + 64;                            // PACK_WAVES * 64 words
    uint32_t *planes = scratch + PACK_WAVES * 64;             // K * plane_words
    uint32_t *tile = planes + K * plane_words;                // kpb * stride (16-byte aligned by host)

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // FUSED: a workgroup packs NSUB tiles of kpb keys one after the other, keeps their records in registers and
    // partitions them TOGETHER -- one cursor reservation and one run of 2 x 4 records per bin instead of two of
    // each (with one tile per workgroup the kernel is bound by its 12.5 M cursor atomics and 64-byte runs, not by
    // HBM: 128 bins instead of 256 made it 0.50 instead of 0.56 ms, but level 2 then pays more than that)
    constexpr uint32_t NSUB = FUSED ? FQD_PACK_NSUB : 1, R = 4;      // R: records per thread and tile (kpb <= 1024)
    uint32_t *s_hist = smem + fs.hist_at, *s_off = smem + fs.tables_at, *s_base = s_off + fs.n_bins;
    uint32_t *s_wave = s_base + fs.n_bins;
    uint16_t *s_bin16 = reinterpret_cast<uint16_t *>(s_wave + 4);
    uint4 *tile4 = reinterpret_cast<uint4 *>(tile);
    uint4 v[NSUB * R];
    uint32_t bin[NSUB * R], rank[NSUB * R];
    uint32_t n_block = 0;                                // keys of this workgroup
    // keys with an N (fs.side_recs): parked here, then appended to a side slab behind ONE cursor reservation
    uint32_t *s_rare_n = smem + fs.rare_at;              // [0] parked so far, [1] their place in the slab
    uint4 *s_rare = reinterpret_cast<uint4 *>(s_rare_n + 4);
    const bool park = FUSED && K == 3 && fs.side_recs != nullptr;
    if (FUSED) {
        if (park && tid == 0)
            s_rare_n[0] = 0;
        for (uint32_t b = tid; b < fs.n_bins; b += PACK_THREADS)
            s_hist[b] = 0;                               // (visible after the first barrier below)
#pragma unroll
        for (uint32_t e = 0; e < NSUB * R; e++)
            bin[e] = 0xFFFFFFFFu;
    }
#pragma unroll
  for (uint32_t sub = 0; sub < NSUB; sub++) {
    const uint64_t key0 = ((uint64_t)blockIdx.x * NSUB + sub) * kpb;
    if (key0 >= n)
        break;                                           // (the same for every thread of the workgroup)
    const uint32_t nk = (uint32_t)min((uint64_t)kpb, n - key0);
    n_block += nk;
    const uint64_t b0 = offsets ? offsets[key0] : key0 * fixed_len;
    const uint64_t b1 = offsets ? offsets[key0 + nk] : (key0 + nk) * fixed_len;
    const uint64_t a0 = b0 & ~15ull;
    const uint32_t span = (uint32_t)(b1 - a0);
    const uint32_t n_rows = (span + 2047u) / 2048u;
    uint32_t bad = 0;
    // ragged keys: the block's key offsets (relative to a0) once into LDS -- the byte scratch of the LUT path, free
    // under the SWAR conversion -- instead of two 8-byte global loads per (key, word) item of phase B: with 64 keys
    // of 300 nt per block those loads were a dependent round trip in each of its three sweeps
    uint32_t *s_koff = scratch;
    const bool koff_lds = SWAR && !FUSED && offsets != nullptr && nk + 1 <= PACK_WAVES * 64;
    if (koff_lds)
        for (uint32_t t = tid; t <= nk; t += PACK_THREADS)
            s_koff[t] = (uint32_t)(offsets[key0 + t] - a0);     // (visible behind the barrier that ends phase A)

    // ---- phase A: byte range -> K bit streams in LDS ------------------------
    // Bits of bytes outside [b0, b1) are never read by phase B; bytes past the END OF THE
    // BUFFER are replaced by a valid symbol so that they cannot raise the foreign-byte flag.
    if (SWAR) {
        const uint32_t code_lo = (uint32_t)ph.code_tbl, code_hi = (uint32_t)(ph.code_tbl >> 32);
        const uint32_t char_lo = (uint32_t)ph.char_tbl, char_hi = (uint32_t)(ph.char_tbl >> 32);
        auto load_group = [&](uint32_t row, uint32_t g) -> uint4 {
            const uint64_t at = a0 + (uint64_t)row * 2048u + g * 1024u + lane * 16u;
            uint4 v = make_uint4(ph.fill, ph.fill, ph.fill, ph.fill);
            if (row < n_rows) {
                if (at + 16 <= n_bytes) {
                    // (a non-temporal load here measured the same: 0.5635 vs 0.566 ms at 50 M x 32)
                    v = *reinterpret_cast<const uint4 *>(bytes + at);
                } else if (at < n_bytes) {
                    uint32_t w[4] = {ph.fill, ph.fill, ph.fill, ph.fill};
                    for (uint32_t j = 0; j < 16; j++)
                        if (at + j < n_bytes) {
                            w[j >> 2] &= ~(0xFFu << (8 * (j & 3)));
                            w[j >> 2] |= (uint32_t)bytes[at + j] << (8 * (j & 3));
                        }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            return v;
        };
        uint16_t *planes16 = reinterpret_cast<uint16_t *>(planes);
        auto convert_row = [&](uint32_t row, const uint4 &g0, const uint4 &g1) {
            const uint32_t x[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            uint32_t acc[K];
#pragma unroll
            for (int k = 0; k < K; k++)
                acc[k] = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int g = q >> 2, i = q & 3;
                // flag of (group g, dword i, byte b) goes to bit 8b + sigma, sigma = (i0, g, i1)
                const int sigma = ((i & 1) << 2) | (g << 1) | (i >> 1);
                uint32_t t = x[q] >> ph.s1;
                if (ph.s2 != PACK_NO_SHIFT)
                    t ^= x[q] >> ph.s2;
                const uint32_t sel = t & 0x07070707u;
                const uint32_t codes = __builtin_amdgcn_perm(code_hi, code_lo, sel);
                const uint32_t expect = __builtin_amdgcn_perm(char_hi, char_lo, sel);
                bad |= (expect ^ x[q]) | (x[q] & 0x80808080u);
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const uint32_t f = codes & (0x01010101u << k);   // flag at bit 8b + k
                    acc[k] |= sigma >= k ? (f << (sigma - k)) : (f >> (k - sigma));
                }
            }
            // bit index (b1 b0 i0 g i1) -> (g i1 i0 b1 b0): swap index bits 4<->1 and 3<->0
#pragma unroll
            for (int k = 0; k < K; k++) {
                uint32_t a = acc[k];
                uint32_t d = ((a >> 14) ^ a) & 0x0000CCCCu;
                a ^= d ^ (d << 14);
                d = ((a >> 7) ^ a) & 0x00AA00AAu;
                a ^= d ^ (d << 7);
                uint16_t *dst = planes16 + (size_t)k * plane_words * 2u + row * 128u;
                dst[lane] = (uint16_t)a;               // stream bits of bytes [16*lane, +16)
                dst[64u + lane] = (uint16_t)(a >> 16); // ... of bytes [1024 + 16*lane, +16)
            }
        };
#if FQD_PACK_PREFETCH == 2
        // two rows ahead: four 16-byte loads per lane in flight while a row is converted (one row ahead: two)
        uint4 c0 = load_group(wave, 0), c1 = load_group(wave, 1);
        uint4 d0 = load_group(wave + PACK_WAVES, 0), d1 = load_group(wave + PACK_WAVES, 1);
        for (uint32_t row = wave; row < n_rows; row += PACK_WAVES) {
            const uint4 n0 = load_group(row + 2 * PACK_WAVES, 0), n1 = load_group(row + 2 * PACK_WAVES, 1);
            convert_row(row, c0, c1);
            c0 = d0;
            c1 = d1;
            d0 = n0;
            d1 = n1;
        }
#else
        uint4 c0 = load_group(wave, 0), c1 = load_group(wave, 1);
        for (uint32_t row = wave; row < n_rows; row += PACK_WAVES) {
            const uint4 n0 = load_group(row + PACK_WAVES, 0), n1 = load_group(row + PACK_WAVES, 1);
            convert_row(row, c0, c1);
            c0 = n0;
            c1 = n1;
        }
#endif
    } else {
        const uint8_t *lut = reinterpret_cast<const uint8_t *>(lut32);
        uint32_t *my_scratch = scratch + wave * 64;
        const uint8_t *my_scratch8 = reinterpret_cast<const uint8_t *>(my_scratch);
        if (tid < 64)
            lut32[tid] = reinterpret_cast<const uint32_t *>(lut_g)[tid];
        __syncthreads();
        const uint32_t n_chunks = n_rows * 8u;  // 256-byte chunks
        for (uint32_t chunk = wave; chunk < n_chunks; chunk += PACK_WAVES) {
            const uint64_t at = a0 + (uint64_t)chunk * 256u + lane * 4u;
            uint32_t v = 0;
            if (at + 4 <= n_bytes) {
                v = *reinterpret_cast<const uint32_t *>(bytes + at);
            } else {
                for (uint32_t j = 0; j < 4; j++)
                    if (at + j < n_bytes)
                        v |= (uint32_t)bytes[at + j] << (8 * j);
            }
            my_scratch[lane] = v;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint64_t pos = a0 + (uint64_t)chunk * 256u + j * 64u + lane;
                const uint32_t code = lut[my_scratch8[j * 64u + lane]];
                bad |= (pos < n_bytes && code == 0xFFu) ? 1u : 0u;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const unsigned long long bits = __builtin_amdgcn_ballot_w64(((code >> k) & 1u) != 0u);
                    if (lane == 0) {
                        uint32_t *dst = planes + k * plane_words + (chunk * 8u + j * 2u);
                        dst[0] = (uint32_t)bits;
                        dst[1] = (uint32_t)(bits >> 32);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // two guard words behind each stream so the funnel shift may read one word ahead
    if (tid < K * 2)
        planes[(tid >> 1) * plane_words + n_rows * 64u + (tid & 1u)] = 0;
    if (bad)
        atomicOr(bad_flag, 1u);
    __syncthreads();

    // ---- phase B: bit streams -> record tile ---------------------------------
    const uint32_t W = sh.words, stride = sh.stride;
    for (uint32_t item = tid; item < nk * W; item += PACK_THREADS) {
        const uint32_t k = item / W, w = item - k * W;
        uint64_t kb, ke;
        if (koff_lds) {
            kb = a0 + s_koff[k];
            ke = a0 + s_koff[k + 1];
        } else if (offsets) {
            kb = offsets[key0 + k];
            ke = offsets[key0 + k + 1];
        } else {
            kb = (key0 + k) * fixed_len;
            ke = kb + fixed_len;
        }
        const uint32_t len = (uint32_t)(ke - kb);
        const uint32_t bitpos = (uint32_t)(kb - a0) + w * 32u;
        const uint32_t q = bitpos >> 5, sft = bitpos & 31u;
        const uint32_t done = w * 32u;
        const uint32_t rem = len > done ? len - done : 0u;
        const uint32_t mask = rem >= 32u ? 0xFFFFFFFFu : ((1u << rem) - 1u);
#pragma unroll
        for (int p = 0; p < K; p++) {
            uint32_t val = 0;
            if (rem) {
                const uint32_t lo = planes[p * plane_words + q], hi = planes[p * plane_words + q + 1];
                val = (uint32_t)((((uint64_t)hi << 32) | lo) >> sft) & mask;
            }
            tile[k * stride + w * K + p] = val;
        }
    }
    for (uint32_t item = tid; item < nk * (stride - W * K); item += PACK_THREADS) {
        const uint32_t pad = stride - W * K;
        const uint32_t k = item / pad, j = item - k * pad;
        uint32_t val = 0;
        if ((sh.ragged & 2u) && j == pad - 1u)
            // the key's LENGTH rides in the record's last padding word (api.hip recs_len_pad): two records are then equal
            // exactly when the keys are, and the long-key collapse needs no gathers out of lens[]
            val = koff_lds ? s_koff[k + 1] - s_koff[k]
                           : offsets ? (uint32_t)(offsets[key0 + k + 1] - offsets[key0 + k]) : fixed_len;
        tile[k * stride + W * K + j] = val;
    }
    __syncthreads();

    if (FUSED) {
        // ---- phase D, first half: this tile's records into registers, their bins and ranks inside the bins
#pragma unroll
        for (uint32_t e0 = 0; e0 < R; e0++) {
            const uint32_t e = sub * R + e0;
            const uint32_t k = e0 * PACK_THREADS + tid;
            if (k < nk) {
                v[e] = tile4[k];
                const uint32_t rec[3] = {v[e].x, v[e].y, v[e].z};
                v[e].w = (uint32_t)(key0 + k) + fs.id_base;   // the read index travels with the record
                if (park && (rec[0] & rec[1]) != 0u) {
                    // a key with an N (code 3): to the side slabs, not into a bin (bin[e] stays "none")
                    const uint32_t at = atomicAdd(&s_rare_n[0], 1u);
                    if (at < PACK_RARE_CAP) {
                        s_rare[at] = v[e];
                    } else {                              // (more of them than the parking space holds: one global atomic each,
                                                          //  spread over the slabs)
                        const uint32_t slab = (blockIdx.x + fs.sub_rot + at) & (fs.side_slabs - 1);
                        const uint32_t pos = atomicAdd(&fs.side_cursor[slab], 1u);
                        if (pos < (slab + 1) * fs.side_cap)
                            fs.side_recs[pos] = v[e];
                        else
                            atomicOr(fs.overflow, 16u);
                    }
                    continue;
                }
                // (route_mask: the bins follow segment 0 of the key alone, in the (a, b) form level 2 will use)
                const uint32_t h = fs.route_mask ? fqd::fqd_route_hash(K == 3 ? rec[0] | rec[2] : rec[0],
                                                                      K == 3 ? rec[1] | rec[2] : (K >= 2 ? rec[1] : 0u),
                                                                      fs.route_mask)
                                                 : fqd_hash_record(rec, W * K, fixed_len);
                if (fs.owner_parts) {
                    // multi-GPU: the bins are owner-major (owner = rank the read goes to, the pigeonhole
                    // rule of fqd_set_owner_rule), hash bins inside -- an owner's reads leave as ONE
                    // range of slabs that is already level 1 of the receiver's collapse
                    // (any function of the segment's bases will do -- every copy of a key and every pair of
                    // keys agreeing on the segment must meet on one rank -- so not fqd_segment_hash, whose
                    // ~45 operations per record showed as +0.08 ms on this bandwidth-bound kernel: one
                    // multiply-xor per plane word, the record being a single 32-base word per plane)
                    uint32_t slo, shi;
                    fqd_segment(fixed_len, rule.seg, rule.nseg, slo, shi);
                    const uint32_t sm = fqd_range_mask(0, slo, shi);
                    uint32_t oh = 0x9E3779B9u;
#pragma unroll
                    for (int kk = 0; kk < K; kk++)
                        oh = (oh ^ (rec[kk] & sm)) * 0x85EBCA6Bu;
                    const uint32_t owner = fqd_mix32(oh) % fs.owner_parts;
                    bin[e] = owner * fs.owner_hb + (fs.owner_hb > 1 ? h >> fs.shift : 0u);
                } else {
                    bin[e] = (h >> fs.shift) & (fs.n_bins - 1);
                }
                rank[e] = atomicAdd(&s_hist[bin[e]], 1u);
            }
        }
        continue;      // (the next tile's phase B waits behind the barrier that ends its phase A: every thread has its records by then)
    }
    // ---- phase C: hash per key, then stream the tile out ---------------------
    for (uint32_t k = tid; k < nk; k += PACK_THREADS) {
        const uint32_t len = koff_lds ? s_koff[k + 1] - s_koff[k]
                                      : offsets ? (uint32_t)(offsets[key0 + k + 1] - offsets[key0 + k]) : fixed_len;
        hashes[key0 + k] = fqd_hash_record(tile + k * stride, W * K, len);
        if (lens)
            lens[key0 + k] = len;
        if (owners)   // multi-GPU: the rank this read goes to
            owners[key0 + k] = fqd_segment_hash(tile + k * stride, K, W * K, len, rule.seg, rule.nseg) % rule.parts;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(recs + key0 * stride);
    const uint4 *src = reinterpret_cast<const uint4 *>(tile);
    for (uint32_t i = tid; i < nk * stride / 4u; i += PACK_THREADS)
        dst[i] = src[i];
  }
    if (FUSED) {
        // ---- phase D, second half: the workgroup's tiles leave partitioned by hash bin ------------
        // tables: hist[n_bins] | off[n_bins] | base[n_bins] | wave[4] | bin16[kpb], behind the record tile
        __syncthreads();
        // the parked keys with an N: one reservation in this workgroup's side slab (their answer is used behind the next
        // barrier); what is left of the block is what the bins hold
        // (two slabs per workgroup, half a table apart, the job-wide workgroup number deciding -- a job packed in
        // pieces starts every launch at block 0, and a job of fewer workgroups than slabs must not leave half of
        // them empty: 300 K reads with 3 % keys with an N overfilled the slabs of its 147 workgroups)
        const uint32_t n_rare = park ? s_rare_n[0] : 0u;
        const uint32_t n_parked = min(n_rare, PACK_RARE_CAP), n_first = (n_parked + 1u) / 2u;
        const uint32_t slab_a = (blockIdx.x + fs.sub_rot) & (fs.side_slabs - 1);
        const uint32_t slab_b = (slab_a + fs.side_slabs / 2u) & (fs.side_slabs - 1);
        if (n_rare && tid == 0)
            s_rare_n[1] = atomicAdd(&fs.side_cursor[slab_a], n_first);
        if (n_parked > n_first && tid == 1)
            s_rare_n[2] = atomicAdd(&fs.side_cursor[slab_b], n_parked - n_first);
        n_block -= n_rare;
        // exclusive scan of the bin counts: n_bins <= 256, one bin per thread
        const uint32_t mine = tid < fs.n_bins ? s_hist[tid] : 0u;
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += up;
        }
        if (lane == 63)
            s_wave[wave] = incl;
        __syncthreads();
        uint32_t excl = incl - mine;
        for (uint32_t wv = 0; wv < wave; wv++)
            excl += s_wave[wv];
        const uint32_t my_sub = (blockIdx.x + fs.sub_rot) & (fs.subs - 1);
        if (tid < fs.n_bins) {
            s_off[tid] = excl;
            uint32_t base = 0;
            if (mine) {
                const uint32_t part = tid * fs.subs + my_sub;
                const uint32_t g = atomicAdd(&fs.cursor[part], mine);
                const uint64_t end = ((uint64_t)part + 1) * fs.cap;
                if ((uint64_t)g + mine > end) {
                    if (FUSED == 2) {
                        // the part's slab is full: the records behind its end go to the spill list, position pos to
                        // spill[s_hist[bin] + pos] (the bin's count has been read: its word is free)
                        const uint32_t first = (uint32_t)max((uint64_t)g, end);
                        s_hist[tid] = atomicAdd(fs.spill_cursor, g + mine - first) - first;
                        fs.l1_over[tid] = 1u;
                    } else {
                        atomicOr(fs.overflow, 4u);    // the part's slab is full: the caller packs the plain way
                    }
                }
                base = g - excl;
            }
            s_base[tid] = base;
        }
        __syncthreads();
        if (tid < n_parked) {
            const uint32_t slab = tid < n_first ? slab_a : slab_b;
            const uint32_t pos = tid < n_first ? s_rare_n[1] + tid : s_rare_n[2] + (tid - n_first);
            if (pos < (slab + 1) * fs.side_cap)
                fs.side_recs[pos] = s_rare[tid];
            else
                atomicOr(fs.overflow, 16u);
        }
        // the records leave in rounds of kpb sorted positions through the one-tile staging area
        for (uint32_t round0 = 0; round0 < n_block; round0 += kpb) {
#pragma unroll
            for (uint32_t e = 0; e < NSUB * R; e++)
                if (bin[e] != 0xFFFFFFFFu) {
                    const uint32_t p = s_off[bin[e]] + rank[e] - round0;
                    if (p < kpb) {
                        tile4[p] = v[e];
                        s_bin16[p] = (uint16_t)bin[e];
                    }
                }
            __syncthreads();
#pragma unroll
            for (uint32_t e = 0; e < R; e++) {
                const uint32_t p = e * PACK_THREADS + tid;
                if (p < kpb && round0 + p < n_block) {
                    // (records that would land behind the slab's end are dropped: what is written stays
                    // gap-free, see partition.cuh)
                    const uint32_t bn = s_bin16[p];
                    const uint32_t pos = s_base[bn] + round0 + p;
                    // (non-temporal stores here: 0.89 ms instead of 0.56 -- the runs of consecutive tiles complete
                    // each other's partial lines in the XCD's L2, which a non-temporal store forgoes)
                    if (pos < (bn * fs.subs + my_sub + 1) * fs.cap) {
                        fs.out[pos] = tile4[p];
                    } else if (FUSED == 2) {
                        const uint32_t at = s_hist[bn] + pos;
                        if (at < fs.spill_cap)
                            fs.spill[at] = tile4[p];
                        else
                            atomicOr(fs.overflow, 4u);     // (the spill list is full as well)
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void hash_records_kernel(const uint32_t *__restrict__ recs,
                                                           const uint32_t *__restrict__ lens, uint64_t n,
                                                           KeyShape sh, uint32_t *__restrict__ hashes)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    hashes[i] = fqd_hash_record(recs + i * sh.stride, sh.words * sh.planes, fqd_key_len(sh, lens, i));
}

// The same for records of several uint4 (reads received from other ranks have no hashes: fqd_import_packed): a
// workgroup's HR_RECS records come in as one coalesced stream of uint4 into LDS rows one word longer than the record
// (lanes then walk THEIR record without bank conflicts); a thread per record walking its 128 bytes in global memory
// took 3.8 ms for 10 M records of 128 bytes -- 0.34 TB/s, the longest kernel of the one-rank plan on config 4's shape.
constexpr uint32_t HR_RECS = 128;
__global__ __launch_bounds__(256) void hash_records_staged_kernel(const uint32_t *__restrict__ recs,
                                                                  const uint32_t *__restrict__ lens, uint64_t n,
                                                                  KeyShape sh, uint32_t *__restrict__ hashes)
{
    extern __shared__ uint32_t s_rows[];                  // HR_RECS x (stride + 1) words
    const uint64_t r0 = (uint64_t)blockIdx.x * HR_RECS;
    const uint32_t nr = (uint32_t)min((uint64_t)HR_RECS, n - r0), q4 = sh.stride / 4u, row = sh.stride + 1u;
    const uint4 *src = reinterpret_cast<const uint4 *>(recs + r0 * sh.stride);
    const uint32_t total = nr * q4;
    for (uint32_t x0 = threadIdx.x; x0 < total; x0 += 4 * 256) {
        uint4 v[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; t++)                   // (clamped, unconditional: in flight together)
            v[t] = src[min(x0 + t * 256, total - 1)];
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) {
            const uint32_t x = x0 + t * 256;
            if (x < total) {
                uint32_t *dst = s_rows + (x / q4) * row + (x % q4) * 4u;
                dst[0] = v[t].x; dst[1] = v[t].y; dst[2] = v[t].z; dst[3] = v[t].w;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < nr)
        hashes[r0 + threadIdx.x] = fqd_hash_record(s_rows + threadIdx.x * row, sh.words * sh.planes,
                                                   fqd_key_len(sh, lens, r0 + threadIdx.x));
}

}  // namespace

namespace fqd {

hipError_t launch_scan_bytes(const uint8_t *bytes, uint64_t n_bytes, uint32_t *present256_dev, hipStream_t st)
{
    if (!n_bytes)
        return hipSuccess;
    uint64_t blocks = (n_bytes / 16 + 255) / 256;
    if (blocks > 2048)
        blocks = 2048;
    if (blocks < 1)
        blocks = 1;
    scan_bytes_kernel<<<(unsigned)blocks, 256, 0, st>>>(bytes, n_bytes, present256_dev);
    return hipGetLastError();
}

hipError_t launch_scan_lens(const uint64_t *offsets, uint64_t n, uint32_t *minmax_dev, hipStream_t st)
{
    if (!n)
        return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 1024)
        blocks = 1024;
    scan_lens_kernel<<<(unsigned)blocks, 256, 0, st>>>(offsets, n, minmax_dev);
    return hipGetLastError();
}

static uint32_t pack_lds_bytes(uint32_t kpb, const KeyShape &sh, uint32_t &plane_words)
{
    // worst-case byte span of kpb keys, plus 15 bytes of alignment lead, in 2048-byte rows
    uint64_t span = (uint64_t)kpb * sh.max_len + 15;
    uint64_t rows = (span + 2047) / 2048;
    plane_words = (uint32_t)(rows * 64 + 2);
    plane_words = (plane_words + 3u) & ~3u;  // keeps the tile 16-byte aligned
    uint64_t words = 64 + PACK_WAVES * 64 + (uint64_t)sh.planes * plane_words + (uint64_t)kpb * sh.stride;
    return words * 4 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(words * 4);
}

// slot(c) = ((c >> s1) ^ (c >> s2)) & 7 must be injective on the alphabet
static bool find_pack_hash(const uint8_t *lut_host, PackHash &ph)
{
    uint8_t syms[8];
    uint32_t nsym = 0;
    for (int b = 0; b < 128; b++)
        if (lut_host[b] != 0xFF) {
            if (nsym == 8)
                return false;
            syms[nsym++] = (uint8_t)b;
        }
    for (uint32_t s1 = 0; s1 <= 5; s1++)
        for (uint32_t s2 = s1; s2 <= 5; s2++) {
            const bool two = s2 != s1;
            uint32_t used = 0;
            bool ok = true;
            for (uint32_t i = 0; i < nsym && ok; i++) {
                const uint32_t slot = ((syms[i] >> s1) ^ (two ? (syms[i] >> s2) : 0)) & 7u;
                ok = !(used & (1u << slot));
                used |= 1u << slot;
            }
            if (!ok)
                continue;
            ph.s1 = s1;
            ph.s2 = two ? s2 : PACK_NO_SHIFT;
            ph.code_tbl = 0;
            ph.char_tbl = 0x8080808080808080ull;
            for (uint32_t i = 0; i < nsym; i++) {
                const uint32_t slot = ((syms[i] >> s1) ^ (two ? (syms[i] >> s2) : 0)) & 7u;
                ph.code_tbl |= (uint64_t)lut_host[syms[i]] << (8 * slot);
                ph.char_tbl &= ~(0xFFull << (8 * slot));
                ph.char_tbl |= (uint64_t)syms[i] << (8 * slot);
            }
            const uint32_t f = nsym ? syms[0] : 0x80u;
            ph.fill = f * 0x01010101u;
            return nsym > 0;
        }
    return false;
}

hipError_t launch_pack(const uint8_t *bytes, uint64_t n_bytes, const uint64_t *offsets, uint64_t n,
                       uint32_t fixed_len, KeyShape sh, const uint8_t *lut_dev, const uint8_t *lut_host,
                       uint32_t *recs, uint32_t *lens, uint32_t *hashes, uint32_t *owners, OwnerRule rule,
                       uint32_t *bad_flag, hipStream_t st, const PackScatter *fused)
{
    if (!rule.parts)
        owners = nullptr;
    if (!n)
        return hipSuccess;
    if ((uintptr_t)bytes & 15u)
        return hipErrorInvalidValue;  // 16-byte loads
    const uint32_t budget = 60 * 1024;
    // keys per block: about 32 KB of key bytes (measured optimum: 1024 keys at 32 nt, 512 at 50, 256
    // at 100, 64 at 300 -- fewer, fatter blocks amortise the three barriers; more LDS than that
    // costs occupancy), then whatever the LDS budget allows
    uint32_t kpb = 1024, plane_words = 0;
    while (kpb > 64 && (uint64_t)kpb * sh.max_len > 32768)
        kpb >>= 1;
    while (kpb > 1 && pack_lds_bytes(kpb, sh, plane_words) > budget)
        kpb >>= 1;
    uint32_t lds = pack_lds_bytes(kpb, sh, plane_words);
    if (lds > budget)
        return hipErrorInvalidValue;  // a single key does not fit the LDS tile
    PackScatter fs{};
    if (fused) {
        if (offsets || sh.ragged || sh.stride != 4 || sh.planes > 3 || kpb > 1024 || fused->n_bins > 256 ||
            (!fused->owner_parts && (fused->n_bins & (fused->n_bins - 1))) || (fused->subs & (fused->subs - 1)) || owners)
            return hipErrorInvalidValue;
        if (fused->owner_parts && (fused->owner_parts * fused->owner_hb != fused->n_bins ||
                                   (fused->owner_hb & (fused->owner_hb - 1)) || rule.parts != fused->owner_parts))
            return hipErrorInvalidValue;
        fs = *fused;
        // The bin counts are needed across the workgroup's tiles: they take the byte scratch of the LUT path (256
        // words, unused by the SWAR conversion) or sit behind the record tile. The other tables (offsets, bases,
        // wave totals, the bins of the staged records) are used when the last tile's bit streams are dead: over
        // those when they fit there, else behind the tile as well. (Everything behind the tile: one workgroup
        // fewer per CU, 0.57 instead of 0.51 ms.)
        const uint32_t tables = (2 * fs.n_bins + 4) * 4 + kpb * 2, streams = sh.planes * plane_words * 4;
        fs.tables_at = 64 + PACK_WAVES * 64;             // = where the streams start (words)
        if (tables > streams) {
            fs.tables_at = lds / 4;
            lds += tables;
        }
        fs.hist_at = 64;                                 // the scratch words
        PackHash ph_probe{};
        if (!(sh.planes <= 3 && find_pack_hash(lut_host, ph_probe)) || fs.n_bins > PACK_WAVES * 64) {
            fs.hist_at = lds / 4;
            lds += fs.n_bins * 4;
        }
        // (the parking space of the keys with an N: behind everything; 30.7 -> 31.3 KB at 32-nt keys, still five workgroups per CU)
        lds = (lds + 15u) & ~15u;
        fs.rare_at = lds / 4;
        if (fs.side_recs) {
            if (!fs.side_cursor || !fs.side_slabs || (fs.side_slabs & (fs.side_slabs - 1)) || fs.owner_parts || sh.planes != 3)
                return hipErrorInvalidValue;
            lds += 16 + PACK_RARE_CAP * 16;
        }
        if (lds > 64 * 1024)
            return hipErrorInvalidValue;
    }
    const uint64_t tiles_per_block = fused ? FQD_PACK_NSUB : 1;      // (pack_kernel NSUB)
    fs.sub_rot = (uint32_t)(fs.id_base / (kpb * tiles_per_block));
    const uint64_t blocks = (n + kpb * tiles_per_block - 1) / (kpb * tiles_per_block);
    if (blocks > 0x7FFFFFFFull)
        return hipErrorInvalidValue;
    PackHash ph{};
    const bool swar = sh.planes <= 3 && find_pack_hash(lut_host, ph);
#define FQD_PACK_CASE(KK, SW)                                                                              \
    pack_kernel<KK, SW><<<(unsigned)blocks, PACK_THREADS, lds, st>>>(bytes, n_bytes, offsets, n, fixed_len, sh, \
                                                                     kpb, plane_words, lut_dev, ph, recs, lens, \
                                                                     hashes, owners, rule, bad_flag, fs)
#define FQD_PACK_FUSED(KK, SW)                                                                             \
    pack_kernel<KK, SW, 1><<<(unsigned)blocks, PACK_THREADS, lds, st>>>(bytes, n_bytes, offsets, n, fixed_len, \
                                                                        sh, kpb, plane_words, lut_dev, ph, recs, \
                                                                        lens, hashes, owners, rule, bad_flag, fs)
    if (fused && fused->spill) {
        // (the spill list exists for the compact records of "ACGNT" keys alone: three planes)
        if (sh.planes != 3 || !fused->spill_cursor || !fused->l1_over || fused->owner_parts)
            return hipErrorInvalidValue;
        if (swar)
            pack_kernel<3, true, 2><<<(unsigned)blocks, PACK_THREADS, lds, st>>>(bytes, n_bytes, offsets, n, fixed_len, sh, kpb,
                                                                                 plane_words, lut_dev, ph, recs, lens, hashes,
                                                                                 owners, rule, bad_flag, fs);
        else
            pack_kernel<3, false, 2><<<(unsigned)blocks, PACK_THREADS, lds, st>>>(bytes, n_bytes, offsets, n, fixed_len, sh, kpb,
                                                                                  plane_words, lut_dev, ph, recs, lens, hashes,
                                                                                  owners, rule, bad_flag, fs);
        return hipGetLastError();
    }
    if (fused) {
        if (swar) {
            switch (sh.planes) {
            case 1: FQD_PACK_FUSED(1, true); break;
            case 2: FQD_PACK_FUSED(2, true); break;
            default: FQD_PACK_FUSED(3, true); break;
            }
        } else {
            switch (sh.planes) {
            case 1: FQD_PACK_FUSED(1, false); break;
            case 2: FQD_PACK_FUSED(2, false); break;
            default: FQD_PACK_FUSED(3, false); break;
            }
        }
        return hipGetLastError();
    }
    if (swar) {
        switch (sh.planes) {
        case 1: FQD_PACK_CASE(1, true); break;
        case 2: FQD_PACK_CASE(2, true); break;
        default: FQD_PACK_CASE(3, true); break;
        }
    } else {
        switch (sh.planes) {
        case 1: FQD_PACK_CASE(1, false); break;
        case 2: FQD_PACK_CASE(2, false); break;
        case 3: FQD_PACK_CASE(3, false); break;
        case 4: FQD_PACK_CASE(4, false); break;
        case 5: FQD_PACK_CASE(5, false); break;
        case 6: FQD_PACK_CASE(6, false); break;
        case 7: FQD_PACK_CASE(7, false); break;
        default: return hipErrorInvalidValue;
        }
    }
#undef FQD_PACK_CASE
#undef FQD_PACK_FUSED
    return hipGetLastError();
}

hipError_t launch_hash_records(const uint32_t *recs, const uint32_t *lens, uint64_t n, KeyShape sh,
                               uint32_t *hashes, hipStream_t st)
{
    if (!n)
        return hipSuccess;
    const size_t lds = (size_t)HR_RECS * (sh.stride + 1) * 4;
    if (sh.stride >= 8 && !(sh.stride & 3u) && lds <= 60 * 1024 && ((uintptr_t)recs & 15u) == 0)
        hash_records_staged_kernel<<<(unsigned)((n + HR_RECS - 1) / HR_RECS), 256, lds, st>>>(recs, lens, n, sh, hashes);
    else
        hash_records_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(recs, lens, n, sh, hashes);
    return hipGetLastError();
}

}  // namespace fqd
