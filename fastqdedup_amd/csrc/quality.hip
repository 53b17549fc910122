// quality.hip -- the per-read quality gate that runs in front of the hot path.
// Replaces _fastq.average_error_rate (reference _fastqmodule.c:38-76) as it is used by
// deduplicate_cluster (__init__.py:243-250): mean over the bytes of a read's (sliced,
// concatenated) phred string of 10**-(q/10), read discarded when mean > threshold.
//
// Bit-exactness: the reference adds the table values one by one in string order into a
// double and divides by the length once. One thread per read does exactly that (no
// tree reduction, no FMA), so the mean -- and therefore the comparison -- is the same
// double. An empty string gives 0.0/0 = NaN, and NaN > t is false: kept, as measured on
// the reference.
//
// A block stages its reads' bytes in LDS with coalesced 16-byte loads; threads then walk
// their own read from LDS. The 128-entry table (8-byte doubles) also sits in LDS.
#include "fqd_internal.h"

namespace {

constexpr uint32_t Q_THREADS = 256;
constexpr uint32_t Q_TILE_BYTES = 48 * 1024;

__global__ __launch_bounds__(Q_THREADS) void quality_kernel(const uint8_t *__restrict__ bytes, uint64_t n_bytes,
                                                            const uint64_t *__restrict__ offsets, uint64_t n,
                                                            uint32_t fixed_len, uint32_t reads_per_block,
                                                            const double *__restrict__ table, uint32_t phred_offset,
                                                            uint32_t max_score, double threshold,
                                                            uint32_t *__restrict__ pass, double *__restrict__ means,
                                                            uint32_t *__restrict__ bad_flag)
{
    __shared__ double s_table[128];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_bytes[];
    const uint32_t tid = threadIdx.x;
    if (tid < 128)
        s_table[tid] = table[tid];
    const uint64_t r0 = (uint64_t)blockIdx.x * reads_per_block;
    const uint32_t nr = (uint32_t)min((uint64_t)reads_per_block, n - r0);
    const uint64_t b0 = offsets ? offsets[r0] : r0 * fixed_len;
    const uint64_t b1 = offsets ? offsets[r0 + nr] : (r0 + nr) * fixed_len;
    const uint64_t a0 = b0 & ~15ull;
    const uint32_t span = (uint32_t)(b1 - a0);
    const bool staged = span <= Q_TILE_BYTES;
    if (staged) {
        for (uint32_t o = tid * 16; o < span; o += Q_THREADS * 16) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a0 + o + 16 <= n_bytes) {
                v = *reinterpret_cast<const uint4 *>(bytes + a0 + o);
            } else {
                uint8_t *p = reinterpret_cast<uint8_t *>(&v);
                for (uint32_t j = 0; j < 16; j++)
                    if (a0 + o + j < n_bytes)
                        p[j] = bytes[a0 + o + j];
            }
            *reinterpret_cast<uint4 *>(s_bytes + o) = v;
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < nr; k += Q_THREADS) {
        const uint64_t kb = offsets ? offsets[r0 + k] : (r0 + k) * fixed_len;
        const uint64_t ke = offsets ? offsets[r0 + k + 1] : kb + fixed_len;
        const uint64_t len = ke - kb;
        const uint8_t *src = staged ? s_bytes + (kb - a0) : bytes + kb;
        double total = 0.0;
        bool bad = false;
        for (uint64_t i = 0; i < len; i++) {
            const uint32_t score = (uint32_t)src[i] - phred_offset;   // wraps below the offset
            if (score > max_score || src[i] > 127) {
                bad = true;
                break;
            }
            total = __dadd_rn(total, s_table[score]);
        }
        const double mean = __ddiv_rn(total, (double)len);
        if (means)
            means[r0 + k] = mean;
        pass[r0 + k] = (mean > threshold) ? 0u : 1u;
        if (bad)
            atomicOr(bad_flag, 1u);
    }
}

}  // namespace

namespace fqd {

hipError_t launch_quality(const uint8_t *bytes, uint64_t n_bytes, const uint64_t *offsets, uint64_t n,
                          uint32_t fixed_len, uint32_t max_len, const double *table_dev, uint32_t phred_offset,
                          uint32_t max_score, double threshold, uint32_t *pass, double *means, uint32_t *bad_flag,
                          hipStream_t st)
{
    if (!n)
        return hipSuccess;
    // reads per block so that the block's bytes fit the LDS tile (else the kernel reads HBM directly)
    uint32_t rpb = 256;
    while (rpb > 1 && (uint64_t)rpb * max_len + 16 > Q_TILE_BYTES)
        rpb >>= 1;
    const uint64_t blocks = (n + rpb - 1) / rpb;
    if (blocks > 0x7FFFFFull * 256)
        return hipErrorInvalidValue;
    quality_kernel<<<(unsigned)blocks, Q_THREADS, Q_TILE_BYTES, st>>>(bytes, n_bytes, offsets, n, fixed_len, rpb,
                                                                       table_dev, phred_offset, max_score, threshold,
                                                                       pass, means, bad_flag);
    return hipGetLastError();
}

}  // namespace fqd
