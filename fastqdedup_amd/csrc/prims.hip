// prims.hip -- device-wide sort / scan from rocPRIM (AMD's own primitives
// library, tuned per-arch incl. gfx950) for the fall-back paths and long arrays, and a
// one-launch scan of our own for the short arrays of the hot path. Kept in one translation unit so
// the heavy headers are compiled once. Everything domain-specific (packing,
// bucket pair search, components, dissection) is hand-written elsewhere.
#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fqd_internal.h"

namespace fqd {

size_t sort_pairs_u32_u32_temp(uint64_t n, int begin_bit, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, n, begin_bit, end_bit);
    return bytes;
}

hipError_t sort_pairs_u32_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, uint64_t n, int begin_bit, int end_bit,
                              hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
}

size_t sort_keys_u64_temp(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, n);
    return bytes;
}

hipError_t sort_keys_u64(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, uint64_t n,
                         int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_keys(tmp, tmp_bytes, kin, kout, n, 0, end_bit, st);
}

size_t sort_pairs_u64_u32_temp(uint64_t n, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, end_bit);
    return bytes;
}

hipError_t sort_pairs_u64_u32(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                              uint32_t *vout, uint64_t n, int end_bit, hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, st);
}

size_t scan_max_u32_temp(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, n,
                                  rocprim::maximum<uint32_t>());
    return bytes;
}

hipError_t inclusive_scan_max_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n,
                                  hipStream_t st)
{
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::maximum<uint32_t>(), st);
}

constexpr uint32_t SL_BLOCK = 1024 * 16;      // words a workgroup of the long scan takes (scan_small_kernel's step)

size_t scan_u32_temp(uint64_t n)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, n,
                                  rocprim::plus<uint32_t>());
    // (the long scan below: a sum per block of SL_BLOCK words, and their scan, 16-byte aligned)
    const size_t blocks = (size_t)((n + SL_BLOCK - 1) / SL_BLOCK);
    return std::max(bytes, ((blocks + 3) & ~(size_t)3) * 8 + 64);
}

// Short scans -- the per-bucket unique counts of the collapse, 2^16 words at config 3 -- by ONE workgroup in ONE
// launch (rocPRIM's look-back scan is two: 4 + 6 us and a hand-over, between the dedupe and the compaction):
// 16 consecutive words per thread and step (four 16-byte loads in flight), scanned in registers, the thread totals
// through a wave scan and LDS, a running carry across the steps of 16 384 words.
constexpr uint32_t SS_THREADS = 1024, SS_PER = 16, SS_MAX = 1u << 18;

__global__ __launch_bounds__(SS_THREADS) void scan_small_kernel(const uint32_t *__restrict__ in,
                                                                uint32_t *__restrict__ out, uint32_t n)
{
    __shared__ uint32_t s_wave[SS_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += SS_THREADS * SS_PER) {
        const uint32_t i0 = base + tid * SS_PER;
        uint32_t v[SS_PER];
        if (i0 + SS_PER <= n) {
#pragma unroll
            for (uint32_t k = 0; k < SS_PER; k += 4) {
                const uint4 q = *reinterpret_cast<const uint4 *>(in + i0 + k);   // (i0 is a multiple of 16 words)
                v[k] = q.x;
                v[k + 1] = q.y;
                v[k + 2] = q.z;
                v[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < SS_PER; k++)
                v[k] = i0 + k < n ? in[i0 + k] : 0u;
        }
#pragma unroll
        for (uint32_t k = 1; k < SS_PER; k++)
            v[k] += v[k - 1];
        uint32_t incl = v[SS_PER - 1];
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o)
                incl += up;
        }
        __syncthreads();                                   // (s_wave of the step before has been read)
        if (lane == 63)
            s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = carry + incl - v[SS_PER - 1];
        uint32_t step_total = 0;
        for (uint32_t w = 0; w < SS_THREADS / 64; w++) {
            before += w < wave ? s_wave[w] : 0u;
            step_total += s_wave[w];
        }
        if (i0 + SS_PER <= n) {
#pragma unroll
            for (uint32_t k = 0; k < SS_PER; k += 4)
                *reinterpret_cast<uint4 *>(out + i0 + k) =
                    make_uint4(v[k] + before, v[k + 1] + before, v[k + 2] + before, v[k + 3] + before);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < SS_PER; k++)
                if (i0 + k < n)
                    out[i0 + k] = v[k] + before;
        }
        carry += step_total;
    }
}

// Long scans -- the (bin x tile) count matrices of a partition with exact bucket sizes: a few million words on the
// skewed workloads -- in three launches of our own: a sum per block of 16 384 words, the scan of the sums (the kernel
// above), the blocks scanned with their carry. (rocPRIM's look-back scan did this until round 4; the skewed step's
// rocprofv3 summary carried its trampoline_kernel / init_lookback_scan 17 times.)
__global__ __launch_bounds__(SS_THREADS) void scan_block_sums_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                                     uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_wave[SS_THREADS / 64];
    const uint64_t i0 = (uint64_t)blockIdx.x * SL_BLOCK + (uint64_t)threadIdx.x * SS_PER;
    uint32_t acc = 0;
    if (i0 + SS_PER <= n) {
#pragma unroll
        for (uint32_t k = 0; k < SS_PER; k += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(in + i0 + k);
            acc += q.x + q.y + q.z + q.w;
        }
    } else {
        for (uint32_t k = 0; k < SS_PER; k++)
            acc += i0 + k < n ? in[i0 + k] : 0u;
    }
    for (int o = 32; o; o >>= 1)
        acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63u) == 0)
        s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < SS_THREADS / 64; w++)
            t += s_wave[w];
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(SS_THREADS) void scan_blocks_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                                 uint64_t n, const uint32_t *__restrict__ sums_incl)
{
    __shared__ uint32_t s_wave[SS_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t carry = blockIdx.x ? sums_incl[blockIdx.x - 1] : 0u;
    const uint64_t i0 = (uint64_t)blockIdx.x * SL_BLOCK + (uint64_t)tid * SS_PER;
    uint32_t v[SS_PER];
    if (i0 + SS_PER <= n) {
#pragma unroll
        for (uint32_t k = 0; k < SS_PER; k += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(in + i0 + k);
            v[k] = q.x;
            v[k + 1] = q.y;
            v[k + 2] = q.z;
            v[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SS_PER; k++)
            v[k] = i0 + k < n ? in[i0 + k] : 0u;
    }
#pragma unroll
    for (uint32_t k = 1; k < SS_PER; k++)
        v[k] += v[k - 1];
    uint32_t incl = v[SS_PER - 1];
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = carry + incl - v[SS_PER - 1];
    for (uint32_t w = 0; w < wave; w++)
        before += s_wave[w];
    if (i0 + SS_PER <= n) {
#pragma unroll
        for (uint32_t k = 0; k < SS_PER; k += 4)
            *reinterpret_cast<uint4 *>(out + i0 + k) =
                make_uint4(v[k] + before, v[k + 1] + before, v[k + 2] + before, v[k + 3] + before);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SS_PER; k++)
            if (i0 + k < n)
                out[i0 + k] = v[k] + before;
    }
}

hipError_t inclusive_scan_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n,
                              hipStream_t st)
{
    const bool aligned = !((uintptr_t)in & 15u) && !((uintptr_t)out & 15u);
    if (n && n <= SS_MAX && aligned) {
        scan_small_kernel<<<1, SS_THREADS, 0, st>>>(in, out, (uint32_t)n);
        return hipGetLastError();
    }
    const uint64_t blocks = (n + SL_BLOCK - 1) / SL_BLOCK;
    const size_t sums_words = (size_t)((blocks + 3) & ~3ull);
    uint32_t *sums = reinterpret_cast<uint32_t *>(((uintptr_t)tmp + 15u) & ~(uintptr_t)15u);
    if (n && aligned && blocks <= SS_MAX && tmp && tmp_bytes >= sums_words * 8 + 16) {
        uint32_t *sums_incl = sums + sums_words;
        scan_block_sums_kernel<<<(unsigned)blocks, SS_THREADS, 0, st>>>(in, n, sums);
        scan_small_kernel<<<1, SS_THREADS, 0, st>>>(sums, sums_incl, (uint32_t)blocks);
        scan_blocks_kernel<<<(unsigned)blocks, SS_THREADS, 0, st>>>(in, out, n, sums_incl);
        return hipGetLastError();
    }
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::plus<uint32_t>(), st);
}

}  // namespace fqd
