"""Randomised soak: many seeds x batch modes, liftover (fused and two-pass) and break-paf against the oracle."""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rustybam_amd
from oracle import pyoracle as oracle
from rbtest_util import random_batch, random_windows, batch_args, compare_hits

eng = rustybam_amd.Engine(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tot = 0
for seed in range(n_cases):
    rng = np.random.default_rng(1000 + seed)
    mode = ["regular", "indel_ends", "wild", "mixed", "spliced"][seed % 5]
    b = random_batch(rng, int(rng.integers(1, 400)), mode, n_contig=int(rng.integers(1, 4)), long_frac=float(rng.random() * 0.3))
    if seed % 7 == 0:  # coordinates that start at 0
        z = rng.random(len(b["t_st"])) < 0.5
        b["t_en"] = np.where(z, b["t_en"] - b["t_st"], b["t_en"]); b["t_st"] = np.where(z, 0, b["t_st"]).astype(np.uint64)
        z = rng.random(len(b["q_st"])) < 0.5
        b["q_en"] = np.where(z, b["q_en"] - b["q_st"], b["q_en"]); b["q_st"] = np.where(z, 0, b["q_st"]).astype(np.uint64)
        b["t_en"], b["q_en"] = b["t_en"].astype(np.uint64), b["q_en"].astype(np.uint64)
    w = random_windows(rng, b, int(rng.integers(1, 600 if seed % 7 == 0 else 200)), bool(seed % 3))
    # scan rows as well
    red, norm0 = eng.scan_records(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
    ored, onorm = oracle.reduce(oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"])), \
        oracle.normalize(oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"]))
    assert np.array_equal(red["status"], ored["status"]) and np.array_equal(norm0["status"], onorm["status"]), (seed, "scan status")
    okn = onorm["status"] == 0
    for kk in ("t_st", "t_en", "q_st", "q_en", "first_op", "n_ops", "nmatch", "aln_len"):
        assert np.array_equal(norm0[kk][okn], onorm[kk][okn]), (seed, "norm", kk)
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"])
    for pol in (rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY):
        orows, oops = oracle.liftover(ob, *w, policy=pol)
        rows, ops, norm, cnt = eng.liftover(*batch_args(b), b["contig"], *w, policy=pol)
        compare_hits(rows, ops, orows, oops, f"seed {seed} {mode} two-pass")
        frows, fops, fnorm, _ = eng.liftover(*batch_args(b), b["contig"], *w, policy=pol | rustybam_amd.LIFT_FUSED_SCAN)
        assert np.array_equal(fnorm["status"], norm["status"]), (seed, "norm status")
        keep = (norm["status"] == 0)[frows["rec"]] if len(frows) else np.zeros(0, bool)
        compare_hits(frows[keep], fops, orows, oops, f"seed {seed} {mode} fused")
        tot += len(orows)
    for ms in (0, 50):
        orows, oops = oracle.break_paf(ob, ms)
        frows, fops, fnorm, _ = eng.break_paf(*batch_args(b), ms, policy=rustybam_amd.LIFT_FUSED_SCAN)
        keep = (fnorm["status"] == 0)[frows["rec"]] if len(frows) else np.zeros(0, bool)
        compare_hits(frows[keep], fops, orows, oops, f"seed {seed} {mode} break {ms}")
        tot += len(orows)
print(f"soak ok: {n_cases} cases, {tot} rows compared")
