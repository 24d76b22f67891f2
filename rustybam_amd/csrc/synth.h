// synth.h -- counter-based synthetic CIGAR generator, integer-only so that the
// host (tests) and device (bench) versions produce identical bytes.  Not a reference function.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define RB_HD __host__ __device__ inline
#else
#define RB_HD static inline
#endif

RB_HD uint64_t rb_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ops per record: uniform in [lo, hi], forced odd so a record starts and ends on '='
RB_HD uint32_t rb_synth_n_ops_impl(uint64_t seed, uint64_t record, uint32_t lo, uint32_t hi) {
    const uint64_t h = rb_splitmix64(seed ^ rb_splitmix64(record ^ 0xA5A5A5A5A5A5A5A5ull));
    uint32_t n = lo + (uint32_t)(h % (uint64_t)(hi - lo + 1));
    if ((n & 1u) == 0) n = (n + 1 <= hi) ? n + 1 : n - 1;
    return n;
}

// op j of a record: even j is an '=' run (about exponential, mean ~360, integer log2 approximation),
// odd j is an event: X .88 / I .062 / D .058; X len 1 (.98) else 2; I/D len 1 (.5), 2..10 (.4),
// 11..100 (.09), 101..5000 (.01)
RB_HD uint32_t rb_synth_op(uint64_t seed, uint64_t record, uint64_t j) {
    const uint64_t h = rb_splitmix64(rb_splitmix64(seed ^ (record * 0xD1342543DE82EF95ull)) + j * 0x9E3779B97F4A7C15ull);
    if ((j & 1ull) == 0) {
        const uint64_t x = h | 1ull;
        const int lz = __builtin_clzll(x);
        const uint32_t e = 63u - (uint32_t)lz;
        const uint64_t f16 = lz == 63 ? 0 : (((x << lz) << 1) >> 48);
        const uint64_t L = ((uint64_t)(64u - e) << 16) - f16; // -log2(x / 2^64) in Q16
        const uint32_t len = 1u + (uint32_t)((L * 360ull * 45426ull) >> 32);
        return (len << 4) | 7u;
    }
    const uint32_t t = (uint32_t)(h >> 8) & 0xFFFFu;
    const uint64_t h2 = rb_splitmix64(h);
    if (t < 57672u) {
        const uint32_t len = ((uint32_t)h2 & 0xFFu) < 251u ? 1u : 2u;
        return (len << 4) | 8u;
    }
    const uint32_t u = (uint32_t)h2 & 0xFFFFu;
    const uint32_t v = (uint32_t)(h2 >> 16);
    uint32_t len;
    if (u < 32768u) len = 1u;
    else if (u < 58982u) len = 2u + v % 9u;
    else if (u < 64880u) len = 11u + v % 90u;
    else len = 101u + v % 4900u;
    return (len << 4) | (t < 61735u ? 1u : 2u);
}
