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
print(f"trim soak ok: {n} cases, {tot} pairs compared")
