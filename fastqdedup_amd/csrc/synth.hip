// synth.hip -- device twin of fastqdedup_amd/synth.py (byte-identical output).
// Bench/test utility: lets bench.py create its workload directly in HBM.
#include <cstdlib>
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

struct SynthParams {
    uint64_t n_total, seed, thr_n, thr_sub;
    uint32_t length, umi, copies;
    // the skewed model of synth.py (all zero: the uniform one): thresholds of the hot molecule and of the ladder,
    // every lowc_every-th molecule poly-A in its first half
    uint64_t thr_hot = 0, thr_ladder = 0;
    uint32_t lowc_every = 0, skew = 0;
};

// base b of read r of the fixed-length job
__device__ __forceinline__ uint8_t synth_base(const SynthParams &p, uint64_t r, uint32_t b)
{
    uint64_t M = p.n_total / p.copies;
    if (M < 1)
        M = 1;
    uint64_t F = M / 4;
    if (F < 1)
        F = 1;
    uint64_t mol = splitmix64(stream_base(p.seed, 0) ^ r);
    bool ladder = false;
    if (p.skew) {
        // (the same two double multiplications, in the same order, as numpy's: no fused multiply-add can form)
        const double x = (double)(mol >> 11) * (1.0 / 9007199254740992.0);
        const double scaled = (double)M * x;
        mol = (uint64_t)(scaled * x);
        if (mol > M - 1)
            mol = M - 1;
        const uint64_t pick = splitmix64(stream_base(p.seed, 8) ^ r) >> 11;
        if (pick < p.thr_hot)
            mol = 0;
        ladder = pick >= p.thr_hot && pick < p.thr_hot + p.thr_ladder;
    } else {
        mol %= M;
    }
    uint64_t truth;
    if (ladder) {
        if (b + 8 >= p.length)
            truth = ((splitmix64(stream_base(p.seed, 9) ^ r) & 0xFFFFull) >> (2 * (b + 8 - p.length))) & 3ull;
        else
            truth = splitmix64(stream_base(p.seed, 7) ^ (uint64_t)b) & 3ull;
    } else if (p.lowc_every && b < p.length / 2 && mol % p.lowc_every == p.lowc_every / 2) {
        truth = 0;
    } else if (b < p.umi) {
        truth = splitmix64(stream_base(p.seed, 2) ^ (mol * p.umi + b)) & 3ull;
    } else {
        const uint64_t ins = splitmix64(stream_base(p.seed, 1) ^ mol) % F;
        truth = splitmix64(stream_base(p.seed, 3) ^ (ins * (uint64_t)(p.length - p.umi) + (b - p.umi))) & 3ull;
    }
    const uint64_t e = splitmix64(stream_base(p.seed, 4) ^ (r * p.length + b));
    const uint64_t u = e >> 11;
    const uint64_t sub = (truth + 1ull + (e & 0x7FFull) % 3ull) & 3ull;
    const uint64_t code = u < p.thr_n + p.thr_sub ? sub : truth;
    const char bases[4] = {'A', 'C', 'G', 'T'};
    return u < p.thr_n ? (uint8_t)'N' : (uint8_t)bases[code];
}

// one thread per (read, base); consecutive threads write consecutive bytes
__global__ __launch_bounds__(256) void synth_kernel(uint8_t *__restrict__ out, SynthParams p, uint64_t start,
                                                    uint64_t count)
{
    const uint64_t total = count * p.length;
    // grid-stride: HIP caps a launch at 2^32 threads and config 5 has 1.5e10 bases
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t local = t / p.length;
        out[t] = synth_base(p, start + local, (uint32_t)(t - local * p.length));
    }
}

// ---- the indel tail (synth.py indel_variant): a share thr_indel / 2^53 of the reads loses one base
// or gains one, so the job has three key lengths (the 149/151-nt sub-population)
struct IndelOf {
    int delta;          // -1, 0, +1
    uint32_t pos;       // deleted base / position of the inserted base
    uint8_t base;       // the inserted base
};

__device__ __forceinline__ IndelOf synth_indel(const SynthParams &p, uint64_t thr_indel, uint64_t r)
{
    IndelOf o{0, 0, 0};
    const uint64_t e = splitmix64(stream_base(p.seed, 5) ^ r);
    if ((e >> 11) >= thr_indel)
        return o;
    const uint64_t h = splitmix64(stream_base(p.seed, 6) ^ r);
    const char bases[4] = {'A', 'C', 'G', 'T'};
    if (e & 1ull) {
        o.delta = 1;
        o.pos = (uint32_t)((h & 0xFFFFFFFFull) % (p.length + 1));
        o.base = (uint8_t)bases[(h >> 32) & 3ull];
    } else {
        o.delta = -1;
        o.pos = (uint32_t)((h & 0xFFFFFFFFull) % p.length);
    }
    return o;
}

__global__ void synth_indel_lens_kernel(SynthParams p, uint64_t thr_indel, uint64_t start, uint64_t count,
                                        unsigned long long *__restrict__ lens)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
        lens[i] = (unsigned long long)((int)p.length + synth_indel(p, thr_indel, start + i).delta);
}

// one thread per (read, output position < length + 1)
__global__ __launch_bounds__(256) void synth_indel_bytes_kernel(SynthParams p, uint64_t thr_indel, uint64_t start,
                                                                uint64_t count,
                                                                const unsigned long long *__restrict__ offsets,
                                                                uint8_t *__restrict__ out)
{
    const uint32_t span = p.length + 1;
    const uint64_t total = count * span;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t local = t / span;
        const uint32_t q = (uint32_t)(t - local * span);
        const uint64_t r = start + local;
        const IndelOf in = synth_indel(p, thr_indel, r);
        const uint32_t len = (uint32_t)((int)p.length + in.delta);
        if (q >= len)
            continue;
        uint8_t ch;
        if (in.delta < 0)
            ch = synth_base(p, r, q < in.pos ? q : q + 1);
        else if (in.delta > 0)
            ch = q == in.pos ? in.base : synth_base(p, r, q < in.pos ? q : q - 1);
        else
            ch = synth_base(p, r, q);
        out[offsets[local] + q] = ch;
    }
}

// Device-to-device copy, 16 bytes per lane and step, four loads in flight per lane: the "achievable
// HBM bandwidth" reference of bench.py (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4_t *__restrict__ src, u32x4_t *__restrict__ dst, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto ld = [&](uint64_t k) { return MODE & 1 ? __builtin_nontemporal_load(src + k) : src[k]; };
    auto st = [&](uint64_t k, u32x4_t v) {
        if (MODE & 2)
            __builtin_nontemporal_store(v, dst + k);
        else
            dst[k] = v;
    };
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4_t a = ld(i), b = ld(i + stride), c = ld(i + 2 * stride), d = ld(i + 3 * stride);
        st(i, a);
        st(i + stride, b);
        st(i + 2 * stride, c);
        st(i + 3 * stride, d);
    }
    for (; i < n16; i += stride)
        st(i, ld(i));
}

}  // namespace

namespace fqd {

hipError_t launch_synth(uint8_t *out, uint64_t n_total, uint64_t start, uint64_t count, uint32_t length,
                        uint32_t umi, uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub,
                        hipStream_t st, uint64_t thr_hot, uint64_t thr_ladder, uint32_t lowc_every, uint32_t skew)
{
    const uint64_t total = count * length;
    if (!total)
        return hipSuccess;
    if (umi > length)
        umi = length;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > (1u << 20))
        blocks = 1u << 20;
    SynthParams p{n_total, seed, thr_n, thr_sub, length, umi, copies};
    p.thr_hot = thr_hot;
    p.thr_ladder = thr_ladder;
    p.lowc_every = lowc_every;
    p.skew = skew;
    synth_kernel<<<(unsigned)blocks, 256, 0, st>>>(out, p, start, count);
    return hipGetLastError();
}

hipError_t launch_copy16(const void *src, void *dst, uint64_t bytes, hipStream_t st)
{
    const uint64_t n16 = bytes / 16;
    if (!n16)
        return hipSuccess;
    const u32x4_t *s4 = reinterpret_cast<const u32x4_t *>(src);
    u32x4_t *d4 = reinterpret_cast<u32x4_t *>(dst);
    // measured on MI355X boxes (2 GiB, GB/s read + write): plain loads and stores 4.6-4.8 (5.1 with 65 536
    // blocks), non-temporal stores 4.8-5.1, non-temporal loads AND stores 5.2-5.6 at 8 192-16 384 blocks;
    // torch's copy 5.4-5.5
    copy16_kernel<3><<<16384, 256, 0, st>>>(s4, d4, n16);
    return hipGetLastError();
}

// lens != NULL: the key lengths (u64, for a scan into offsets); else offsets + out: the key bytes
hipError_t launch_synth_indels(uint64_t n_total, uint64_t start, uint64_t count, uint32_t length, uint32_t umi,
                               uint64_t seed, uint32_t copies, uint64_t thr_n, uint64_t thr_sub, uint64_t thr_indel,
                               unsigned long long *lens, const unsigned long long *offsets, uint8_t *out, hipStream_t st)
{
    if (!count || !length)
        return hipSuccess;
    if (umi > length)
        umi = length;
    const SynthParams p{n_total, seed, thr_n, thr_sub, length, umi, copies};
    if (lens) {
        synth_indel_lens_kernel<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(p, thr_indel, start, count, lens);
    } else {
        uint64_t blocks = (count * (length + 1) + 255) / 256;
        if (blocks > (1u << 20))
            blocks = 1u << 20;
        synth_indel_bytes_kernel<<<(unsigned)blocks, 256, 0, st>>>(p, thr_indel, start, count, offsets, out);
    }
    return hipGetLastError();
}

}  // namespace fqd
