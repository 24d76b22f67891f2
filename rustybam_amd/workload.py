"""Synthetic liftover workloads of SURVEY.md 8(d) / BASELINE.md section 3 (bench + tests plumbing).

Counter-based: record r of a workload depends only on (seed, r), so every rank of a multi-GPU run
generates its own shard.  CIGAR ops come from the library's generator (rb_synth_* in
include/rustybam_amd.h, identical on host and device); the record headers are derived here from the
per-record reference/query spans that the record-scan kernel returns.
"""
import numpy as np

CHR1_LEN = 248_387_497
SEED_CONFIG2 = 0x5EED0002
SEED_CONFIG3 = 0x5EED0003
SEED_CONFIG4 = 0x5EED0004
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def n_ops(seed, first_record, n_rec, lo=1000, hi=9000):
    """numpy twin of rb_synth_n_ops (csrc/synth.h): uniform in [lo, hi], forced odd."""
    rec = np.arange(first_record, first_record + n_rec, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ splitmix64(rec ^ np.uint64(0xA5A5A5A5A5A5A5A5)))
    n = np.uint64(lo) + h % np.uint64(hi - lo + 1)
    even = (n & np.uint64(1)) == 0
    n = np.where(even, np.where(n + np.uint64(1) <= np.uint64(hi), n + np.uint64(1), n - np.uint64(1)), n)
    return n.astype(np.uint64)


LOGNORMAL_Q = np.array([  # tools/gen_lognormal_table.py: quantiles k / 256 of lognormal(ln 2000, 1.35) clipped to [31, 80000]
    31, 55, 76, 94, 109, 123, 137, 149, 162, 174, 185, 197, 208, 219, 230, 241,
    252, 263, 274, 284, 295, 306, 316, 327, 338, 348, 359, 369, 380, 391, 402, 412,
    423, 434, 445, 456, 467, 478, 489, 500, 512, 523, 534, 546, 557, 569, 580, 592,
    604, 616, 628, 640, 652, 664, 676, 689, 701, 714, 726, 739, 752, 765, 778, 791,
    805, 818, 832, 845, 859, 873, 887, 901, 915, 930, 944, 959, 973, 988, 1003, 1019,
    1034, 1049, 1065, 1081, 1097, 1113, 1129, 1145, 1162, 1179, 1196, 1213, 1230, 1247, 1265, 1283,
    1301, 1319, 1337, 1356, 1375, 1394, 1413, 1432, 1452, 1472, 1492, 1512, 1533, 1554, 1575, 1596,
    1617, 1639, 1661, 1683, 1706, 1729, 1752, 1775, 1799, 1823, 1847, 1872, 1897, 1922, 1948, 1974,
    2000, 2027, 2054, 2081, 2109, 2137, 2165, 2194, 2223, 2253, 2283, 2314, 2345, 2376, 2408, 2440,
    2473, 2507, 2540, 2575, 2610, 2645, 2681, 2718, 2755, 2793, 2831, 2870, 2910, 2950, 2991, 3033,
    3075, 3118, 3162, 3207, 3252, 3298, 3346, 3394, 3442, 3492, 3543, 3595, 3647, 3701, 3756, 3812,
    3869, 3927, 3987, 4047, 4109, 4173, 4237, 4303, 4371, 4440, 4511, 4583, 4657, 4733, 4810, 4890,
    4971, 5055, 5141, 5229, 5319, 5411, 5507, 5604, 5705, 5808, 5915, 6024, 6137, 6253, 6373, 6497,
    6625, 6757, 6893, 7034, 7180, 7331, 7488, 7651, 7820, 7995, 8178, 8368, 8566, 8773, 8989, 9214,
    9451, 9699, 9959, 10233, 10522, 10826, 11148, 11489, 11852, 12237, 12648, 13088, 13561, 14069, 14619, 15216,
    15866, 16580, 17366, 18239, 19215, 20316, 21572, 23023, 24726, 26762, 29259, 32424, 36630, 42630, 52292, 72547,
    80000,
], dtype=np.uint64)


def n_ops_lognormal(seed, first_record, n_rec):
    """numpy twin of rb_synth_n_ops_lognormal (csrc/synth.h): SURVEY 8(d)'s imbalance shape, 31 .. 80000 ops, forced odd."""
    rec = np.arange(first_record, first_record + n_rec, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ splitmix64(rec ^ np.uint64(0x5A5A5A5A5A5A5A5A)))
    k = (h >> np.uint64(56)).astype(np.int64)
    f = (h >> np.uint64(40)) & np.uint64(0xFFFF)
    n = LOGNORMAL_Q[k] + (((LOGNORMAL_Q[k + 1] - LOGNORMAL_Q[k]) * f) >> np.uint64(16))
    even = (n & np.uint64(1)) == 0
    n = np.where(even, np.where(n + np.uint64(1) <= np.uint64(80000), n + np.uint64(1), n - np.uint64(1)), n)
    return n.astype(np.uint64)


def op_offsets(n):
    off = np.zeros(len(n) + 1, np.uint64)
    off[1:] = np.cumsum(n, dtype=np.uint64)
    return off


def headers(seed, first_record, t_bases, q_bases, placement="uniform", window=(12_000_000, 13_000_000),
            t_len=CHR1_LEN):
    """(t_st, t_en, q_st, q_en, strand) for records first_record.. given their CIGAR spans.
    placement 'uniform': anywhere on the target (config 3); 'overlap': every record overlaps `window`
    (config 2)."""
    t_bases = np.asarray(t_bases, np.uint64)
    q_bases = np.asarray(q_bases, np.uint64)
    n = len(t_bases)
    rec = np.arange(first_record, first_record + n, dtype=np.uint64)
    h1 = splitmix64(np.uint64(seed) ^ splitmix64(rec ^ np.uint64(0x1111111111111111)))
    h2 = splitmix64(h1)
    h3 = splitmix64(h2)
    if placement == "uniform":
        room = np.uint64(t_len) - np.minimum(t_bases, np.uint64(t_len)) + np.uint64(1)
        t_st = h1 % room
    elif placement == "overlap":
        w0, w1 = np.uint64(window[0]), np.uint64(window[1])
        lo = np.where(t_bases > w0, np.uint64(0), w0 - t_bases + np.uint64(1))  # t_en > w0
        hi = w1 - np.uint64(1)                                                   # t_st < w1
        t_st = lo + h1 % (hi - lo + np.uint64(1))
    else:
        raise ValueError(placement)
    strand = np.where((h2 & np.uint64(1)) == 0, ord("+"), ord("-")).astype(np.uint8)
    q_st = h3 % np.uint64(100_001)
    return t_st, t_st + t_bases, q_st, q_st + q_bases, strand


def sliding_windows(n_win=3000, step=82_796, width=100_000, t_len=CHR1_LEN):
    """config 3: st = i * 82,796, en = min(st + 100 kb, chr1 length), one contig."""
    st = np.arange(n_win, dtype=np.uint64) * np.uint64(step)
    en = np.minimum(st + np.uint64(width), np.uint64(t_len))
    return np.zeros(n_win, np.uint32), st, en


def algorithmic_bytes(n_ops_total, n_rec, n_hits, n_out_ops):
    """4 B per input op + 48 B per record + 88 B per hit + 4 B per emitted op."""
    return 4 * int(n_ops_total) + 48 * int(n_rec) + 88 * int(n_hits) + 4 * int(n_out_ops)


def kernel_source_sha():
    """sha256 (first 16 hex digits) of the streaming clip kernel's sources: the PMC traffic figure bench.py reports (measured in a
    separate rocprofv3 --pmc run, profiles/traffic_*.json) is valid for exactly these"""
    import hashlib
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in ("k_liftover.hip", "k_tile.hip", "rb_lift.h", "rb_device.h"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
