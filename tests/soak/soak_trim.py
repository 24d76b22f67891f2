"""Randomised soak of the trim-paf pair kernel against the oracle, biased towards coordinates that start at 0."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rustybam_amd
from oracle import pyoracle as oracle
from rbtest_util import random_cigar, sums
from test_gpu_trim import _compare

def pairs_batch(rng, n_pairs, mode):
    cig, t_st, t_en, q_st, q_en, strand, left, right = [], [], [], [], [], [], [], []
    for _ in range(n_pairs):
        ca = random_cigar(rng, int(rng.integers(3, 60)), mode)
        cb = random_cigar(rng, int(rng.integers(3, 60)), mode)
        (ra, qa), (rb, qb) = sums(ca), sums(cb)
        if min(qa, qb) < 2:
            continue
        a0 = 0 if rng.random() < 0.5 else int(rng.integers(0, 1000))
        o = int(rng.integers(1, min(qa, qb)))
        b0 = a0 + qa - o
        for c, r, q, s0 in ((ca, ra, qa, a0), (cb, rb, qb, b0)):
            ts = 0 if rng.random() < 0.5 else int(rng.integers(0, 5000))
            cig.append(c); t_st.append(ts); t_en.append(ts + r); q_st.append(s0); q_en.append(s0 + q)
            strand.append(ord("+") if rng.random() < .5 else ord("-"))
        left.append(len(cig) - 2); right.append(len(cig) - 1)
    off = np.zeros(len(cig) + 1, np.uint64)
    off[1:] = np.cumsum([len(c) for c in cig])
    return dict(ops=np.concatenate(cig), op_off=off, t_st=np.array(t_st, np.uint64), t_en=np.array(t_en, np.uint64),
                q_st=np.array(q_st, np.uint64), q_en=np.array(q_en, np.uint64), strand=np.array(strand, np.uint8)), \
        np.array(left, np.uint32), np.array(right, np.uint32)

eng = rustybam_amd.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tot = 0
for seed in range(n):
    rng = np.random.default_rng(7000 + seed)
    mode = ["regular", "indel_ends", "wild"][seed % 3]
    b, left, right = pairs_batch(rng, 200, mode)
    for pol in (0, 1):
        for scores in ((1, 1, 1), (2, 3, 5)):
            rows, out = eng.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right, scores, pol)
            ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], np.zeros(len(b["t_st"]), np.uint32))
            orows, oout = oracle.overlap_split(ob, left, right, scores, pol)
            _compare(rows, out, orows, oout, f"seed {seed} {mode} pol {pol} {scores}")
            tot += len(rows)
# the shape of SURVEY 8d config 4 (tools/bench_config4.py): synthetic records of 300-700 ops, 4 per query, consecutive query spans
# overlapping by U[100, 10000] bases; device against oracle, and the oracle alone timed (one core) for profiles/r01_c4_summary.md
import time
from rustybam_amd import workload as wl, capi
SEED4, n4 = 0x5EED0004, 2400
nops = wl.n_ops(SEED4, 0, n4, 300, 700)
off4 = wl.op_offsets(nops)
ops4 = capi.synth_fill_ops_host(SEED4, 0, off4)
strand4 = np.where(np.arange(n4) % 3 == 0, ord("-"), ord("+")).astype(np.uint8)
z = np.zeros(n4, np.uint64)
red, _ = eng.scan_records(ops4, off4, z, z, z, z, strand4)
tb, qb = red["t_bases"].astype(np.uint64), red["q_bases"].astype(np.uint64)
rng = np.random.default_rng(SEED4)
q_st = np.zeros(n4, np.uint64)
ov = rng.integers(100, 10001, n4).astype(np.uint64)
for j in range(1, 4):
    prev_en = q_st[j - 1::4] + qb[j - 1::4]
    q_st[j::4] = prev_en - np.minimum(ov[j::4], np.minimum(qb[j - 1::4], qb[j::4]) // np.uint64(2))
q_en = q_st + qb
t_st = rng.integers(0, 200_000_000, n4).astype(np.uint64)
t_en = t_st + tb
left4 = np.arange(n4, dtype=np.uint32)[np.arange(n4) % 4 != 3]
right4 = left4 + 1
rows, out = eng.overlap_split(ops4, off4, t_st, t_en, q_st, q_en, strand4, left4, right4)
ob = oracle.Batch(ops4, off4, t_st, t_en, q_st, q_en, strand4, np.zeros(n4, np.uint32))
t0 = time.perf_counter()
orows, oout = oracle.overlap_split(ob, left4, right4, (1, 1, 1), 0)
dt = time.perf_counter() - t0
for k in ("split_idx", "split_score", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
    assert np.array_equal(rows[k], orows[k]), k
assert np.array_equal(out, oout)
print(f"config-4 shape: {len(left4)} pairs identical; oracle alone, 1 core: {dt:.2f} s = {len(left4) / dt:.3g} pairs/s")
# ... and the op-space CPU port of the pair step (oracle/rb_opspace.c, the checker of the full-size test) on the same pairs, timed: what a CPU
# implementation that does not expand CIGARs achieves (the pairs tiled to a few hundred thousand so that the threads have something to do)
z4 = np.zeros(n4, np.uint32)
reps = 200
L_, R_ = np.tile(left4, reps), np.tile(right4, reps)
for nt in (1, min(64, os.cpu_count() or 1)):
    t0 = time.perf_counter()
    prow, bad = oracle.overlap_split_opspace(ops4, off4[:-1], np.diff(off4).astype(np.uint32), z4, z4, t_st, t_en, q_st, q_en, strand4, L_, R_, (1, 1, 1), n_threads=nt)
    dt = time.perf_counter() - t0
    assert bad == 0
    for k in ("split_idx", "split_score", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
        assert np.array_equal(prow[k][:len(left4)], orows[k]), k
    print(f"op-space CPU port, {nt} thread(s): {len(L_)} pairs in {dt:.2f} s = {len(L_) / dt:.3g} pairs/s")
print(f"trim soak ok: {n} cases, {tot} pairs compared")
