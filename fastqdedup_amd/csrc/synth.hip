// synth.hip -- device twin of fastqdedup_amd/synth.py (byte-identical output).
// Bench/test utility: lets bench.py create its workload directly in HBM.
#include "fqd_internal.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t stream_base(uint64_t seed, uint64_t stream)
{
    return splitmix64(seed + stream * 0xD1B54A32D192ED03ull);
}

// one thread per (read, base); consecutive threads write consecutive bytes
__global__ __launch_bounds__(256) void synth_kernel(uint8_t *__restrict__ out, uint64_t n_total, uint64_t start,
                                                    uint64_t count, uint32_t length, uint32_t umi, uint64_t seed,
                                                    uint32_t copies, uint64_t thr_n, uint64_t thr_sub)
{
    uint64_t M = n_total / copies;
    if (M < 1)
        M = 1;
    uint64_t F = M / 4;
    if (F < 1)
        F = 1;
    const uint64_t total = count * length;
    // grid-stride: HIP caps a launch at 2^32 threads and config 5 has 1.5e10 bases
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t local = t / length;
        const uint32_t b = (uint32_t)(t - local * length);
        const uint64_t r = start + local;
        const uint64_t mol = splitmix64(stream_base(seed, 0) ^ r) % M;
        uint64_t truth;
        if (b < umi) {
            truth = splitmix64(stream_base(seed, 2) ^ (mol * umi + b)) & 3ull;
        } else {
            const uint64_t ins = splitmix64(stream_base(seed, 1) ^ mol) % F;
            truth = splitmix64(stream_base(seed, 3) ^ (ins * (uint64_t)(length - umi) + (b - umi))) & 3ull;
        }
        const uint64_t e = splitmix64(stream_base(seed, 4) ^ (r * length + b));
        const uint64_t u = e >> 11;
        const uint64_t sub = (truth + 1ull + (e & 0x7FFull) % 3ull) & 3ull;
        const uint64_t code = u < thr_n + thr_sub ? sub : truth;
        const char bases[4] = {'A', 'C', 'G', 'T'};
        out[t] = u < thr_n ? (uint8_t)'N' : (uint8_t)bases[code];
    }
}

}  // namespace

namespace fqd {

hipError_t launch_synth(uint8_t *out, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length,
                        uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub,
                        hipStream_t st)
{
    const uint64_t total = count * length;
    if (!total)
        return hipSuccess;
    if (umi > length)
        umi = length;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > (1u << 20))
        blocks = 1u << 20;
    synth_kernel<<<(unsigned)blocks, 256, 0, st>>>(out, n_total, start, count, length, umi, seed, copies, thr_n,
                                                   thr_sub);
    return hipGetLastError();
}

}  // namespace fqd
