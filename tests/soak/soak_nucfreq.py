"""Randomised soak of rb_dev_nucfreq against the oracle's pileup: many seeds, read shapes and regions; then a slice of the
config-5 bench workload (tools/bench_nucfreq.py) region by region."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
import rustybam_amd
from oracle import pyoracle as oracle
from nf_util import Reads, random_reads, check_regions

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
oracle.build()
eng = rustybam_amd.Engine(0)
pos_total = 0
for case in range(n_cases):
    rng = np.random.default_rng(0xABC000 + case)
    shape = case % 4
    if shape == 0:
        rd = random_reads(rng, int(rng.integers(1, 400)), n_contig=3, span=40000, long_frac=0.15)
    elif shape == 1:
        rd = random_reads(rng, int(rng.integers(500, 3000)), n_contig=1, span=6000, long_frac=0.0, max_ops=6)   # deep
    elif shape == 2:
        rd = random_reads(rng, int(rng.integers(1, 30)), n_contig=2, span=200000, long_frac=0.9, max_ops=200)   # long cigars
    else:
        rd = random_reads(rng, int(rng.integers(1, 200)), n_contig=4, span=10000, long_frac=0.05, odd_flags=False, max_ops=3)
    regions = []
    for _ in range(int(rng.integers(1, 8))):
        t = int(rng.integers(0, 4))
        st = int(rng.integers(0, 50000))
        regions.append((t, st, st + int(rng.integers(1, 30000))))
    check_regions(eng, oracle, rd, regions)
    pos_total += sum(e - s for _t, s, e in regions)

# piles that cross htslib's cap of 8000 buffered reads: random starts, lengths, flags and regions (every region at most one
# 10 kb fetch of the reference, so that the oracle's per-fetch admission is the device's per-region admission)
n_deep = max(2, n_cases // 10)
for case in range(n_deep):
    rng = np.random.default_rng(0xDEE9000 + case)
    n = int(rng.integers(8500, 20000))
    span = int(rng.choice([1, 5, 60, 300]))
    pos = np.sort(rng.integers(0, span, n)).tolist()
    cigs, seqs, flags = [], [], []
    for k in range(n):
        l = int(rng.integers(5, 120)) if rng.random() < 0.98 else int(rng.integers(2000, 9000))
        if rng.random() < 0.15 and l > 8:
            d = int(rng.integers(1, 6))
            cigs.append([((l // 2) << 4), (d << 4) | int(rng.choice([1, 2])), ((l - l // 2) << 4)])
            q = l + (d if cigs[-1][1] & 15 == 1 else 0)
        else:
            cigs.append([(l << 4)])
            q = l
        seqs.append(rng.choice([1, 2, 4, 8, 15], size=q).tolist())
        flags.append(int(rng.choice([0, 0, 0, 0, 0, 0, 16, 1024, 256])))
    rd = Reads([0] * n, pos, flags, cigs, seqs)
    regions = [(0, 0, 9000)]
    for _ in range(int(rng.integers(1, 4))):
        st = int(rng.integers(0, span + 100))
        regions.append((0, st, st + int(rng.integers(1, 9000))))
    counts, status, ctr = check_regions(eng, oracle, rd, regions)
    assert ctr["cap_overflow"] == 0
    pos_total += sum(e - s for _t, s, e in regions)
    print(f"deep case {case}: {n} reads over {span} starts, max depth {ctr['max_depth']}, dropped {ctr['n_dropped']}", flush=True)

# the bench workload, scaled down: 30x of 3 Mbp by 15 kb reads
import bench_nucfreq as B
pos, ops, op_off, n = B.make_reads(3_000_000, 30, 15000)
rng = np.random.default_rng(5)
bpr = 7500
seq = np.random.default_rng(6).choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], np.uint8), n * bpr + 16)
rd = Reads.__new__(Reads)
rd.tid, rd.pos, rd.flag = np.zeros(n, np.int32), pos, np.zeros(n, np.uint32)
rd.op_off, rd.ops, rd.l_seq = op_off, ops, np.full(n, 15000, np.uint32)
rd.seq_off, rd.seq, rd.n = np.arange(n, dtype=np.uint64) * np.uint64(bpr), seq, n
regs = [(0, int(s), int(s) + 20000) for s in rng.integers(0, 2_950_000, 6)] + [(0, 0, 30000), (0, 2_960_000, 3_000_000)]
check_regions(eng, oracle, rd, regs)
pos_total += sum(e - s for _t, s, e in regs)
# the oracle alone on the same regions (one core, per 10 kb fetch + pileup like the reference): the CPU figure quoted beside the
# device number in profiles/r01_nf_summary.md
import time
from nf_util import oracle_region
t0 = time.perf_counter()
based = 0
for t, s0, e0 in regs:
    cov, cnt = oracle_region(oracle, rd, t, s0, e0)
    based += int(cnt.sum())
dt = time.perf_counter() - t0
print(f"oracle pileup, 1 core: {based} bases over {sum(e - s for _t, s, e in regs)} positions in {dt:.2f} s = {based / dt:.3g} bases/s")
print(f"nucfreq soak ok: {n_cases} random cases + 8 regions of the bench workload, {pos_total} positions compared")
