// Micro-benchmarks of the access patterns the config-3 step is made of: what the MEMORY SYSTEM
// gives for each pattern, to hold the kernels' measured rates against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_patterns.hip && /tmp/mb
//  1 copy            device-to-device copy, 1 GiB (the pack kernel's and the dedupe's reads, every sequential write)
//  2 scatter_runs    16-byte records leave in RUNS of R records at random places of a 0.8 GB buffer (level 1 / level 2 of
//                    the partitions: a (tile, bin) run is 4 records out of the pack kernel, 8 out of level 2)
//  3 gather4         one random 4-byte word per item out of a 56 MB array, E items, L loads in flight per lane (the
//                    union-find's parent walks, the directional pass's state bytes, the kept list's verdicts)
//  4 gather_pair_cas per "edge": two random 4-byte reads and one compare-and-swap on a random word of the same 56 MB
//                    (one union of the union-find without its retries)
//  5 gather_rec      two random 16-byte records per candidate (the verification of the search)
//  6 lds_cas_insert  bucket_dedupe12_kernel without its output: 65 536 workgroups of 256 threads, each streams 768 items
//                    of 12 bytes and inserts them into a 1025-slot LDS table (one 64-bit compare-and-swap, an add and a
//                    min per item; ~210 distinct keys per workgroup), counts the live slots, writes one word
//  7 pass0_pairs     bucket_compact12_kernel<true> without its memory: 4096 persistent waves x 16 buckets of 210 rows
//                    made up in registers: the counting sort into 64 sub-bins in LDS and the pair compares, one word out
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; return x ^ (x >> 16); }

__global__ void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

// every group of R consecutive lanes writes one run of R records at a random run-aligned place
template <uint32_t R>
__global__ void scatter_runs_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint32_t n, uint32_t n_runs_mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t run = i / R, k = i % R;
    const uint32_t dst_run = (run * 2654435761u + 12345u) & n_runs_mask;      // (odd multiplier, power-of-two range: a permutation)
    out[(size_t)dst_run * R + k] = in[i];
}

template <uint32_t L>
__global__ void gather4_kernel(const uint32_t *__restrict__ table, uint32_t mask, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) * L;
    if (base >= n)
        return;
    uint32_t v[L], acc = 0;
#pragma unroll
    for (uint32_t l = 0; l < L; l++)
        v[l] = table[mix(base + l) & mask];
#pragma unroll
    for (uint32_t l = 0; l < L; l++)
        acc += v[l];
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <uint32_t L>
__global__ void pair_cas_kernel(uint32_t *table, uint32_t mask, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) * L;
    if (base >= n)
        return;
    uint32_t a[L], b[L], acc = 0;
#pragma unroll
    for (uint32_t l = 0; l < L; l++) {
        a[l] = table[mix(2 * (base + l)) & mask];
        b[l] = table[mix(2 * (base + l) + 1) & mask];
    }
#pragma unroll
    for (uint32_t l = 0; l < L; l++)
        acc += atomicCAS(&table[mix(3 * (base + l) + 7) & mask], a[l] ^ b[l] ^ 0xFFFFFFFFu, a[l]);   // (never equal: no store)
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <uint32_t L>
__global__ void gather_rec_kernel(const uint4 *__restrict__ recs, uint32_t mask, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) * L;
    if (base >= n)
        return;
    uint4 a[L], b[L];
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t l = 0; l < L; l++) {
        a[l] = recs[mix(2 * (base + l)) & mask];
        b[l] = recs[mix(2 * (base + l) + 1) & mask];
    }
#pragma unroll
    for (uint32_t l = 0; l < L; l++)
        acc += (a[l].x ^ b[l].x) + (a[l].y ^ b[l].y) + (a[l].z ^ b[l].z) + (a[l].w ^ b[l].w);   // (not a popcount: its range would prove the store below dead)
    if (acc == 0x12345678u)
        out[0] = acc;
}

// (6) the dedupe's LDS phase: items streamed from HBM, 64-bit CAS inserts, live-slot count
__global__ __launch_bounds__(256) void lds_cas_insert_kernel(const uint32_t *__restrict__ items, uint32_t per_wg, uint32_t distinct,
                                                             uint32_t *__restrict__ out)
{
    __shared__ unsigned long long s_key[1025];
    __shared__ uint32_t s_cnt[1025], s_min[1025];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    uint32_t a[4], b[4], id[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {                       // (clamped, unconditional: in flight together)
        const uint32_t i = min(k * 256u + tid, per_wg - 1u);
        const uint32_t *p = items + ((size_t)wg * per_wg + i) * 3u;
        a[k] = p[0]; b[k] = p[1]; id[k] = p[2];
    }
    for (uint32_t sl = tid; sl <= 1024; sl += 256) {
        s_key[sl] = ~0ull; s_cnt[sl] = 0; s_min[sl] = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t pend = 0, slot[4];
    unsigned long long key[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = k * 256u + tid;
        // (the key: one of `distinct` per workgroup, as if ~3.6 copies of each; the loaded words ride along so that the loads stay)
        const uint32_t which = mix(i * 2654435761u + wg) % distinct;
        const uint32_t ka = mix(wg * 4099u + which) | ((a[k] ^ b[k] ^ id[k]) == 0x9E3779B9u ? 1u : 0u), kb = mix(ka + 77u);
        key[k] = ((unsigned long long)kb << 32) | ka;
        slot[k] = (mix(ka ^ kb) * 0x9E3779B1u) >> 22;
        if (i < per_wg)
            pend |= 1u << k;
    }
    for (uint32_t probes = 0; pend && probes < 1024; probes++) {
        unsigned long long old[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (pend >> k & 1u)
                old[k] = atomicCAS(&s_key[slot[k]], ~0ull, key[k]);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (pend >> k & 1u) {
                if (old[k] == ~0ull || old[k] == key[k]) {
                    atomicAdd(&s_cnt[slot[k]], 1u);
                    atomicMin(&s_min[slot[k]], id[k]);
                    pend &= ~(1u << k);
                } else {
                    slot[k] = (slot[k] + 1) & 1023u;
                }
            }
    }
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t sl = tid; sl <= 1024; sl += 256)
        mine += s_cnt[sl] > 0 ? 1u : 0u;
    for (int o = 32; o; o >>= 1)
        mine += __shfl_xor(mine, o);
    if ((tid & 63u) == 0)
        atomicAdd(&s_cnt[1024], mine);
    __syncthreads();
    if (tid == 0)
        out[wg & 15u] = s_cnt[1024];
}

// (7) search pass 0 inside the compaction, the LDS part alone: rows made up in registers
__global__ __launch_bounds__(256) void pass0_pairs_kernel(uint32_t n_buckets, uint32_t rows, uint32_t mask, uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_pa[4][512], s_pb[4][512], s_poff[4][66];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t waves_total = (gridDim.x * blockDim.x) >> 6, wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t found = 0;
    for (uint32_t b = wave_global; b < n_buckets; b += waves_total) {
        uint32_t ra[8], rb[8], sr[8];
        s_poff[wave][lane] = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t j = lane + 64 * q;
            // (rows of a bucket: random keys, a share of them one substitution away from another row's)
            ra[q] = mix(b * 977u + (j >> 1));
            rb[q] = mix(ra[q] + 13u) ^ ((j & 1u) << ((j >> 1) & 15u));
            sr[q] = 0;
            if (j < rows) {
                const uint32_t sub = (mix((ra[q] & mask) * 0x9E3779B1u + (rb[q] & mask)) >> 10) & 63u;
                sr[q] = (sub << 16) | atomicAdd(&s_poff[wave][sub], 1u);
            }
        }
        {
            const uint32_t c = s_poff[wave][lane];
            uint32_t incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if ((int)lane >= o)
                    incl += up;
            }
            s_poff[wave][lane] = incl - c;
            if (lane == 63)
                s_poff[wave][64] = incl;
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
            if (lane + 64 * q < rows) {
                const uint32_t p = s_poff[wave][sr[q] >> 16] + (sr[q] & 0xFFFFu);
                s_pa[wave][p] = ra[q];
                s_pb[wave][p] = rb[q];
            }
        for (uint32_t i = lane; i < rows; i += 64) {
            const uint32_t ai = s_pa[wave][i], bi = s_pb[wave][i];
            const uint32_t end = s_poff[wave][((mix((ai & mask) * 0x9E3779B1u + (bi & mask)) >> 10) & 63u) + 1];
            for (uint32_t k = i + 1; k < end; k++) {
                const uint32_t x = (ai ^ s_pa[wave][k]) | (bi ^ s_pb[wave][k]);
                if (!(x & mask) && (uint32_t)__popc(x) <= 1u)
                    found++;
            }
        }
    }
    for (int o = 32; o; o >>= 1)
        found += __shfl_xor(found, o);
    if (lane == 0 && found)
        atomicAdd(out, found);
}


template <class F>
static float time_ms(F &&launch, int reps = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main()
{
    const size_t copy_bytes = 1ull << 30;
    const uint32_t n_rec = 50u << 20;                     // 52 M records of 16 bytes (0.84 GB)
    const uint32_t table_words = 1u << 24;                // 64 MB of 4-byte words (config 3: 14 M keys = 56 MB)
    const uint32_t E = 1u << 21;                          // 2 M "edges" (config 3: 1.67 M)
    uint4 *a, *b;
    uint32_t *table, *out;
    CK(hipMalloc(&a, copy_bytes)); CK(hipMalloc(&b, copy_bytes));
    CK(hipMalloc(&table, (size_t)table_words * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, copy_bytes)); CK(hipMemset(b, 0, copy_bytes));
    {   // (records that differ, so that nothing about the gathers can be folded)
        std::vector<uint32_t> h(1u << 20);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
        for (size_t off = 0; off < copy_bytes; off += h.size() * 4) CK(hipMemcpy((char *)a + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    } CK(hipMemset(table, 0x5A, (size_t)table_words * 4));
    printf("{\n");
    {
        const float ms = time_ms([&] { copy_kernel<<<256 * 64, 256>>>(a, b, copy_bytes / 16); });
        printf("  \"copy_1GiB\": {\"ms\": %.4f, \"GB_per_s_read_plus_write\": %.1f},\n", ms, 2.0 * copy_bytes / ms / 1e6);
    }
    {
        const uint32_t n = 1u << 25;                      // 32 M records of 16 bytes (0.5 GB in, 0.5 GB out), every run to a place of its own
        (void)n_rec;
        const float m4 = time_ms([&] { scatter_runs_kernel<4><<<(n + 255) / 256, 256>>>(a, b, n, n / 4 - 1); });
        const float m8 = time_ms([&] { scatter_runs_kernel<8><<<(n + 255) / 256, 256>>>(a, b, n, n / 8 - 1); });
        const float m16 = time_ms([&] { scatter_runs_kernel<16><<<(n + 255) / 256, 256>>>(a, b, n, n / 16 - 1); });
        printf("  \"scatter_runs_32M_records_of_16B\": {\"runs_of_4\": {\"ms\": %.4f, \"GB_per_s_read_plus_write\": %.1f}, "
               "\"runs_of_8\": {\"ms\": %.4f, \"GB_per_s_read_plus_write\": %.1f}, \"runs_of_16\": {\"ms\": %.4f, \"GB_per_s_read_plus_write\": %.1f}},\n",
               m4, 32.0 * n / m4 / 1e6, m8, 32.0 * n / m8 / 1e6, m16, 32.0 * n / m16 / 1e6);
    }
    {
        const uint32_t n = 4 * E;                         // 8 M random words
        const float l1 = time_ms([&] { gather4_kernel<1><<<(n + 255) / 256, 256>>>(table, table_words - 1, n, out); });
        const float l4 = time_ms([&] { gather4_kernel<4><<<(n / 4 + 255) / 256, 256>>>(table, table_words - 1, n, out); });
        const float l16 = time_ms([&] { gather4_kernel<16><<<(n / 16 + 255) / 256, 256>>>(table, table_words - 1, n, out); });
        printf("  \"gather4_8M_words_of_64MB\": {\"1_in_flight\": {\"ms\": %.4f, \"G_words_per_s\": %.1f}, \"4_in_flight\": {\"ms\": %.4f, "
               "\"G_words_per_s\": %.1f}, \"16_in_flight\": {\"ms\": %.4f, \"G_words_per_s\": %.1f}},\n",
               l1, n / l1 / 1e6, l4, n / l4 / 1e6, l16, n / l16 / 1e6);
    }
    {
        const uint32_t n = E;
        const float l1 = time_ms([&] { pair_cas_kernel<1><<<(n + 255) / 256, 256>>>(table, table_words - 1, n, out); });
        const float l4 = time_ms([&] { pair_cas_kernel<4><<<(n / 4 + 255) / 256, 256>>>(table, table_words - 1, n, out); });
        printf("  \"two_reads_and_a_cas_2M_edges_of_64MB\": {\"1_edge_per_lane\": {\"ms\": %.4f, \"M_edges_per_ms\": %.2f}, "
               "\"4_edges_per_lane\": {\"ms\": %.4f, \"M_edges_per_ms\": %.2f}},\n", l1, n / l1 / 1e6, l4, n / l4 / 1e6);
    }
    {
        const uint32_t n = 14u << 20;                     // 14 M candidates (config 3: one search pass)
        const uint32_t mask = (1u << 26) - 1;             // 64 M records of 16 bytes = 1 GiB (beyond the 256 MB Infinity Cache; config 3: 224 MB)
        const float l1 = time_ms([&] { gather_rec_kernel<1><<<(n + 255) / 256, 256>>>(a, mask, n, out); });
        const float l4 = time_ms([&] { gather_rec_kernel<4><<<(n / 4 + 255) / 256, 256>>>(a, mask, n, out); });
        printf("  \"two_random_16B_records_per_candidate_14M\": {\"1_per_lane\": {\"ms\": %.4f, \"G_candidates_per_s\": %.2f}, "
               "\"4_per_lane\": {\"ms\": %.4f, \"G_candidates_per_s\": %.2f}},\n", l1, n / l1 / 1e6, l4, n / l4 / 1e6);
    }
    {
        // config 3: 65 536 buckets of ~763 items (50 M x 12 bytes = 0.6 GB streamed), ~210 distinct keys each
        const uint32_t wgs = 65536, per = 763;
        const float ms = time_ms([&] { lds_cas_insert_kernel<<<wgs, 256>>>((const uint32_t *)a, per, 210, out); });
        printf("  \"lds_cas_insert_65536_workgroups_of_763_items\": {\"ms\": %.4f, \"G_items_per_s\": %.1f, \"GB_per_s_streamed\": %.1f},\n",
               ms, (double)wgs * per / ms / 1e6, (double)wgs * per * 12 / ms / 1e6);
    }
    {
        const float ms = time_ms([&] { pass0_pairs_kernel<<<1024, 256>>>(65536, 210, 0xFFFFu, out); });
        printf("  \"pass0_pairs_65536_buckets_of_210_rows\": {\"ms\": %.4f, \"M_buckets_per_ms\": %.2f}\n", ms, 65536 / ms / 1e3);
    }
    printf("}\n");
    return 0;
}
