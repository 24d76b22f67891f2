"""Text form of the scaled SURVEY 8d config-4 workload (tools/bench_config4.py): `python tools/gen_config4_paf.py <records> > w.paf`.
4 records per query q<k>, consecutive query spans overlapping by U[100, 10000] bases, 25 targets; needs no GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rustybam_amd import workload as wl, capi

SEED = 0x5EED0004
n = int(sys.argv[1]) // 4 * 4
nops = wl.n_ops(SEED, 0, n, 300, 700)
off = wl.op_offsets(nops)
ops = capi.synth_fill_ops_host(SEED, 0, off)
ln, oc = (ops >> 4).astype(np.int64), (ops & 15)
ref = np.add.reduceat(np.where(np.isin(oc, [0, 2, 3, 7, 8]), ln, 0), off[:-1].astype(np.int64))
qry = np.add.reduceat(np.where(np.isin(oc, [0, 1, 4, 7, 8]), ln, 0), off[:-1].astype(np.int64))
rng = np.random.default_rng(SEED)
q_st = np.zeros(n, np.int64)
ov = rng.integers(100, 10001, n)
for j in range(1, 4):
    q_st[j::4] = q_st[j - 1::4] + qry[j - 1::4] - np.minimum(ov[j::4], np.minimum(qry[j - 1::4], qry[j::4]) // 2)
q_en = q_st + qry
t_st = rng.integers(0, 40_000_000, n)
tname = rng.integers(1, 26, n)
out = sys.stdout
sym = "MIDNSHP=X"
for r in range(n):
    a, b = int(off[r]), int(off[r + 1])
    cg = "".join(f"{int(l)}{sym[int(o)]}" for l, o in zip(ln[a:b], oc[a:b]))
    qlen = int(q_en[r // 4 * 4 + 3]) + 1000
    out.write(f"q{r // 4}\t{qlen}\t{q_st[r]}\t{q_en[r]}\t+\tchr{tname[r]}\t250000000\t{t_st[r]}\t{t_st[r] + ref[r]}\t0\t0\t60\tcg:Z:{cg}\n")
