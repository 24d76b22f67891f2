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

// ops per record, the secondary ("imbalance") shape of SURVEY.md 8(d): log-normal (mu = ln 2000, sigma = 1.35) clipped to [31, 80000]
// -- the fixture's own range (31 .. 75,176 ops) --, forced odd.  Integer-only: 257 quantiles at k / 256 (tools/gen_lognormal_table.py),
// linear in between, so that the numpy twin (rustybam_amd/workload.py n_ops_lognormal) gives the same counts.
RB_HD uint32_t rb_synth_n_ops_lognormal_impl(uint64_t seed, uint64_t record) {
    static const uint32_t Q[257] = {
    31, 55, 76, 94, 109, 123, 137, 149, 162, 174, 185, 197,
    208, 219, 230, 241, 252, 263, 274, 284, 295, 306, 316, 327,
    338, 348, 359, 369, 380, 391, 402, 412, 423, 434, 445, 456,
    467, 478, 489, 500, 512, 523, 534, 546, 557, 569, 580, 592,
    604, 616, 628, 640, 652, 664, 676, 689, 701, 714, 726, 739,
    752, 765, 778, 791, 805, 818, 832, 845, 859, 873, 887, 901,
    915, 930, 944, 959, 973, 988, 1003, 1019, 1034, 1049, 1065, 1081,
    1097, 1113, 1129, 1145, 1162, 1179, 1196, 1213, 1230, 1247, 1265, 1283,
    1301, 1319, 1337, 1356, 1375, 1394, 1413, 1432, 1452, 1472, 1492, 1512,
    1533, 1554, 1575, 1596, 1617, 1639, 1661, 1683, 1706, 1729, 1752, 1775,
    1799, 1823, 1847, 1872, 1897, 1922, 1948, 1974, 2000, 2027, 2054, 2081,
    2109, 2137, 2165, 2194, 2223, 2253, 2283, 2314, 2345, 2376, 2408, 2440,
    2473, 2507, 2540, 2575, 2610, 2645, 2681, 2718, 2755, 2793, 2831, 2870,
    2910, 2950, 2991, 3033, 3075, 3118, 3162, 3207, 3252, 3298, 3346, 3394,
    3442, 3492, 3543, 3595, 3647, 3701, 3756, 3812, 3869, 3927, 3987, 4047,
    4109, 4173, 4237, 4303, 4371, 4440, 4511, 4583, 4657, 4733, 4810, 4890,
    4971, 5055, 5141, 5229, 5319, 5411, 5507, 5604, 5705, 5808, 5915, 6024,
    6137, 6253, 6373, 6497, 6625, 6757, 6893, 7034, 7180, 7331, 7488, 7651,
    7820, 7995, 8178, 8368, 8566, 8773, 8989, 9214, 9451, 9699, 9959, 10233,
    10522, 10826, 11148, 11489, 11852, 12237, 12648, 13088, 13561, 14069, 14619, 15216,
    15866, 16580, 17366, 18239, 19215, 20316, 21572, 23023, 24726, 26762, 29259, 32424,
    36630, 42630, 52292, 72547, 80000,
    };
    const uint64_t h = rb_splitmix64(seed ^ rb_splitmix64(record ^ 0x5A5A5A5A5A5A5A5Aull));
    const uint32_t k = (uint32_t)(h >> 56), f = (uint32_t)(h >> 40) & 0xFFFFu;
    uint32_t n = Q[k] + (uint32_t)(((uint64_t)(Q[k + 1] - Q[k]) * f) >> 16);
    if ((n & 1u) == 0) n = (n + 1 <= 80000u) ? n + 1 : n - 1;
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
