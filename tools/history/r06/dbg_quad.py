import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))
import rustybam_amd
from oracle import pyoracle as oracle
from test_gpu_trim import _pairs_batch
eng = rustybam_amd.Engine(0)
rng = np.random.default_rng(60)
b, left, right = _pairs_batch(rng, 40, "regular", ops_range=(60, 200))
rows, out = eng.overlap_split(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right, (1, 1, 1))
ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
orows, oout = oracle.overlap_split(ob, left, right, (1, 1, 1))
nops = np.diff(b["op_off"]).astype(int)
for i in range(len(rows)):
    diffs = [k for k in ("status", "split_idx", "split_score", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n") if not np.array_equal(rows[k][i], orows[k][i])]
    l, r = int(left[i]), int(right[i])
    print(i, "pad", int(rows["_pad"][i]), "n", nops[l], nops[r], "strand", chr(b["strand"][l]), chr(b["strand"][r]), "ovl", int(b["q_en"][l]) - int(b["q_st"][r]), "diffs", diffs,
          [(k, rows[k][i].tolist(), orows[k][i].tolist()) for k in diffs][:3])
print("debug build fields:")
for i in (8, 13, 14, 15, 35, 36):
    l = int(left[i]); o0, o1 = int(b["op_off"][l]), int(b["op_off"][l + 1])
    ops = b["ops"][o0:o1]; ln = (ops >> 4).astype(np.int64); oc = ops & 15
    n = len(ops); m = min(n, 128); i0 = n - m
    refm = np.isin(oc, (0, 2, 3, 7, 8)); qm = np.isin(oc, (0, 1, 4, 7, 8))
    f = lambda k: (int(rows[k][i][0]) >> 32, int(rows[k][i][0]) & 0xffffffff)
    print(i, "gpu bU,bR", f("t_st"), "tu,tr", f("t_en"), "tq,N", f("q_st"), "i0,m", f("q_en"),
          "| want bU,bR", int(ln[:i0].sum()), int(ln[:i0][refm[:i0]].sum()), "tu,tr,tq", int(ln[i0:].sum()), int(ln[i0:][refm[i0:]].sum()), int(ln[i0:][qm[i0:]].sum()), "N", int(ln.sum()), "i0,m", i0, m)
