"""Soak with long records (several 512-op steps and several checkpoint segments) and many windows per record."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rustybam_amd
from oracle import pyoracle as oracle
from rbtest_util import random_cigar, sums, batch_args, compare_hits

eng = rustybam_amd.Engine(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = 0
for seed in range(n_cases):
    rng = np.random.default_rng(5000 + seed)
    mode = ["regular", "indel_ends", "spliced", "mixed"][seed % 4]
    cig, t_st, t_en, q_st, q_en, strand = [], [], [], [], [], []
    for _ in range(int(rng.integers(2, 14))):
        n_ops = int(rng.choice([511, 512, 513, 1023, 1025, 5119, 5120, 5121, 5125, int(rng.integers(3000, 16000))]))
        m = mode if mode != "mixed" else str(rng.choice(["regular", "indel_ends", "wild"]))
        c = random_cigar(rng, n_ops, m)
        R, Q = sums(c)
        ts, qs = int(rng.integers(0, 2000)), int(rng.integers(0, 2000))
        cig.append(c); t_st.append(ts); t_en.append(ts + R); q_st.append(qs); q_en.append(qs + Q)
        strand.append(ord("+") if rng.random() < .5 else ord("-"))
    off = np.zeros(len(cig) + 1, np.uint64); off[1:] = np.cumsum([len(c) for c in cig])
    b = dict(ops=np.concatenate(cig), op_off=off, t_st=np.array(t_st, np.uint64), t_en=np.array(t_en, np.uint64),
             q_st=np.array(q_st, np.uint64), q_en=np.array(q_en, np.uint64), strand=np.array(strand, np.uint8),
             contig=np.zeros(len(cig), np.uint32))
    hi = int(b["t_en"].max()) + 10
    nw = int(rng.integers(20, 1500))
    st = np.sort(rng.integers(0, hi, nw)).astype(np.uint64)
    ln = rng.choice([1, 3, 40, 700, 9000], nw).astype(np.uint64)
    en = np.maximum.accumulate(st + ln) if seed % 3 else st + ln  # monotone or not
    w = (np.zeros(nw, np.uint32), st, en.astype(np.uint64))
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"])
    for pol in (rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY):
        orows, oops = oracle.liftover(ob, *w, policy=pol)
        for extra in (0, rustybam_amd.LIFT_FUSED_SCAN, rustybam_amd.LIFT_FUSED_SCAN | rustybam_amd.LIFT_EARLY_EXIT):
            rows, ops, norm, cnt = eng.liftover(*batch_args(b), b["contig"], *w, policy=pol | extra)
            keep = (norm["status"] == 0)[rows["rec"]] if len(rows) else np.zeros(0, bool)
            compare_hits(rows[keep], ops, orows, oops, f"seed {seed} {mode} pol {pol} extra {extra}")
        tot += len(orows)
    orows, oops = oracle.break_paf(ob, 8)
    rows, ops, norm, cnt = eng.break_paf(*batch_args(b), 8, policy=rustybam_amd.LIFT_FUSED_SCAN)
    keep = (norm["status"] == 0)[rows["rec"]] if len(rows) else np.zeros(0, bool)
    compare_hits(rows[keep], ops, orows, oops, f"seed {seed} break")
    tot += len(orows)
print(f"long soak ok: {n_cases} cases, {tot} rows compared")
