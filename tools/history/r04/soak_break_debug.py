"""diagnostics: one case of tests/soak/soak_break.py, one-walk against two-walk row by row.  usage: python tools/soak_break_debug.py <case>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rustybam_amd
from devutil import DevBatch
from rbtest_util import random_cigar, sums, unpack

case = int(sys.argv[1])
dev = torch.device("cuda", 0)
eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
BASE = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
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
h2, o2 = D.host_rows(rows2, out2)
norm2 = D.d_norm.cpu().numpy().view(rustybam_amd.NORM_DT).copy()
rows1, out1, cnt1 = D.run(max_size=max_size, policy=BASE | rustybam_amd.BREAK_ONE_WALK, rows_cap=cap)
h1, o1 = D.host_rows(rows1, out1)
norm1 = D.d_norm.cpu().numpy().view(rustybam_amd.NORM_DT).copy()
print("case", case, mode, "max_size", max_size, "records", len(cig), "rows", len(h1), len(h2), "redo", cnt1["redo_two_walk"], "generic", cnt1["n_generic"], cnt2["n_generic"])
print("norm status one-walk", norm1["status"].tolist())
print("norm status two-walk", norm2["status"].tolist())
n = min(len(h1), len(h2))
for k in range(n):
    a, c = h1[k], h2[k]
    same = all(a[f] == c[f] for f in ("rec", "win", "status")) and (a["status"] != 0 or (all(a[f] == c[f] for f in ("t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n")) and
            np.array_equal(o1[int(a["out_off"]):int(a["out_off"]) + int(a["out_n"])], o2[int(c["out_off"]):int(c["out_off"]) + int(c["out_n"])])))
    if not same:
        print("first difference at row", k)
        print(" one-walk", {f: int(a[f]) for f in a.dtype.names if f != "_pad"}, unpack(o1[int(a["out_off"]):int(a["out_off"]) + min(int(a["out_n"]), 12)]))
        print(" two-walk", {f: int(c[f]) for f in c.dtype.names if f != "_pad"}, unpack(o2[int(c["out_off"]):int(c["out_off"]) + min(int(c["out_n"]), 12)]))
        r = int(a["rec"])
        print(" record", r, "n_ops", len(cig[r]), "strand", chr(strand[r]), "t", t_st[r], t_en[r], "q", q_st[r], q_en[r], "norm", {f: int(norm2[f][r]) for f in norm2.dtype.names if f != "_pad"})
        print(" cigar head", unpack(cig[r][:16]), "... tail", unpack(cig[r][-16:]))
        break
else:
    print("rows identical up to", n)
for r in range(len(cig)):
    c1, c2 = int((h1["rec"] == r).sum()), int((h2["rec"] == r).sum())
    if c1 != c2:
        print("record", r, "rows one-walk", c1, "two-walk", c2, "norm status", int(norm1["status"][r]), int(norm2["status"][r]))
