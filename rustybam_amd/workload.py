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
    for f in ("k_liftover.hip", "rb_lift.h", "rb_device.h"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
