import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import rustybam_amd
from oracle import pyoracle
from rbtest_util import read_paf, batch_args
g = "/root/repo/tests/golden"
r = read_paf(os.path.join(g, "asm_small.paf"))
b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
wc, ws, we = [], [], []
tlen = {}
for n, L in zip(r.t_name, r.t_len):
    tlen.setdefault(n, L)
for name, L in tlen.items():
    for s in range(0, L, 100000):
        wc.append(r.contig_names[name]); ws.append(s); we.append(min(s + 100000, L))
w = (np.array(wc, np.uint32), np.array(ws, np.uint64), np.array(we, np.uint64))
eng = rustybam_amd.Engine(0)
rows, ops, norm, cnt = eng.liftover(*batch_args(b), b["contig"], *w, policy=0)
orows, oops = pyoracle.liftover(pyoracle.Batch(*batch_args(b), b["contig"]), *w, policy=0)
print("rows", len(rows), len(orows), "generic", cnt["n_generic"])
first = {}
for i in range(len(rows)):
    rec = int(rows["rec"][i]); first.setdefault(rec, i)
for k in ("q_st", "q_en", "nmatch", "aln_len", "out_n"):
    bad = np.nonzero((orows["status"] == 0) & (rows[k].astype(np.int64) != orows[k].astype(np.int64)))[0]
    print(k, len(bad))
    for i in bad[:40]:
        rec = int(rows["rec"][i]); j = i - first[rec]
        print("   row", i, "rec", rec, "hit", j, "pass", j // 32, "lane", j % 32, "flags", int(rows["flags"][i]), "strand", chr(b["strand"][rec]), "diff", int(rows[k][i]) - int(orows[k][i]),
              "n_ops", int(r.op_off[rec + 1] - r.op_off[rec]), "out_n", int(rows["out_n"][i]))
