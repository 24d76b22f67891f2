"""Soak of break-paf in one walk (RB_BREAK_ONE_WALK, k_liftover.hip BRK build) against the two-walk path and the oracle: random
regular / indel-ended / spliced batches with long records (several segments of the stream, many passes of 32 pieces), random
--max-size.  usage: python tests/soak/soak_break.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import rustybam_amd
from oracle import pyoracle as oracle
from devutil import DevBatch
from rbtest_util import batch_args, digest_rows, random_cigar, sums

oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
BASE = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
rows_total = taken = 0
for case in range(n_cases):
    rng = np.random.default_rng(0xB4EA000 + case)
    mode = ["regular", "indel_ends", "spliced"][case % 3]
    cig, t_st, t_en, q_st, q_en, strand = [], [], [], [], [], []
    for _ in range(int(rng.integers(1, 40))):
        n_ops = int(rng.choice([1, 3, 40, 511, 513, 5119, 5121, int(rng.integers(100, 12000))]))
        c = random_cigar(rng, n_ops, mode)
        R, Q = sums(c)
        ts, qs = int(rng.integers(0, 3)) * int(rng.integers(0, 5000)), int(rng.integers(0, 3)) * int(rng.integers(0, 5000))
        cig.append(c); t_st.append(ts); t_en.append(ts + R); q_st.append(qs); q_en.append(qs + Q)
        strand.append(ord("+") if rng.random() < .5 else ord("-"))
    off = np.zeros(len(cig) + 1, np.uint64); off[1:] = np.cumsum([len(c) for c in cig])
    b = dict(ops=np.concatenate(cig), op_off=off, t_st=np.array(t_st, np.uint64), t_en=np.array(t_en, np.uint64),
             q_st=np.array(q_st, np.uint64), q_en=np.array(q_en, np.uint64), strand=np.array(strand, np.uint8),
             contig=np.zeros(len(cig), np.uint32))
    max_size = int(rng.choice([0, 1, 3, 10, 29, 100]))
    D = DevBatch(torch, eng, dev, b)
    cap = int(off[-1]) + 64
    rows2, out2, cnt2 = D.run(max_size=max_size, policy=BASE, rows_cap=cap)
    want = D.digest(rows2, out2)
    rows1, out1, cnt1 = D.run(max_size=max_size, policy=BASE | rustybam_amd.BREAK_ONE_WALK, rows_cap=cap)
    if cnt1["redo_two_walk"]:
        assert cnt2["n_generic"] > 0 or mode != "regular", (case, mode, max_size)   # only what the fast path cannot resolve may be declined
    else:
        taken += 1
        if mode == "regular":
            assert rows1.shape[0] == rows2.shape[0] and D.digest(rows1, out1) == want, (case, mode, max_size)
        else:
            # records the reference panics on (norm status != 0) have rows in neither path that anybody reads -- the two-walk path
            # leaves rows that carry the status, the one-walk path declines the record and leaves none: the rows of the others must agree
            norm = D.d_norm.cpu().numpy().view(rustybam_amd.NORM_DT)
            h1, o1 = D.host_rows(rows1, out1)
            h2, o2 = D.host_rows(rows2, out2)
            ok = norm["status"] == 0
            h1, h2 = h1[ok[h1["rec"]]], h2[ok[h2["rec"]]]
            assert len(h1) == len(h2), (case, mode, max_size, len(h1), len(h2))
            for f in ("rec", "win", "status"):
                assert np.array_equal(h1[f], h2[f]), (case, mode, max_size, f)
            good = h1["status"] == 0
            for f in ("t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
                assert np.array_equal(h1[f][good], h2[f][good]), (case, mode, max_size, f)
            for a, c in zip(h1[good], h2[good]):
                assert np.array_equal(o1[int(a["out_off"]):int(a["out_off"]) + int(a["out_n"])], o2[int(c["out_off"]):int(c["out_off"]) + int(c["out_n"])]), (case, mode, max_size, int(a["rec"]))
    if mode == "regular":   # (the other modes hold records the reference panics on: compared between the device paths only)
        orows, oops = oracle.break_paf(oracle.Batch(*batch_args(b), b["contig"]), max_size)
        assert rows2.shape[0] == len(orows) and want == digest_rows(orows, oops), (case, max_size)
    rows_total += rows2.shape[0]
print(f"break soak ok: {n_cases} cases, {rows_total} rows, one walk taken in {taken} cases")
