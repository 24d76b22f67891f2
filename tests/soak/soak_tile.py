"""Randomised soak of the tile kernel (k_tile.hip): batches of SHORT records of random shapes under random window lists, liftover and
break-paf through the tiles, through the per-record kernel (RB_TILE=0) and through the per-base oracle.  What the seeds vary is what a
tile is made of: record lengths from one op to just past the short-record limit (2048), a handful to a few thousand records, one or three
contigs, both strands, windows from one base to longer than a record, dense enough that tiles are cut by their hits, sorted or not,
three policies, break-paf sizes from 0 to 1000, and every fifth batch mixed with irregular records (handed back one by one).

    python3 tests/soak/soak_tile.py [cases] [first seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rustybam_amd
from oracle import pyoracle as oracle
from rbtest_util import random_batch, random_windows
from test_gpu_tile import synth_batch, lift_both, break_both, FUSED

eng = rustybam_amd.Engine(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tiles = back = rows_seen = 0
for case in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(77_000 + case)
    lo = int(rng.choice([1, 2, 4, 7, 8, 9, 20, 63, 64, 65, 100, 300, 511, 1000, 1500, 2040]))
    hi = lo + int(rng.choice([0, 1, 7, 60, 400, 1100]))
    budget = int(rng.choice([20_000, 200_000, 1_200_000]))  # ops of the batch
    n_rec = max(1, min(3000, budget // max(1, (lo + hi) // 2)))
    n_contig = int(rng.choice([1, 1, 3]))
    mean_bases = 190 * (lo + hi) // 2 + 1
    span = int(rng.choice([mean_bases // 2 + 1, 4 * mean_bases, 40 * mean_bases]))  # records piled up ... spread out
    if case % 5 == 4:
        b = random_batch(rng, min(n_rec, 400), "mixed", n_contig=n_contig, long_frac=0.0)
        what = f"case {case}: mixed batch of {len(b['t_st'])} records"
        w = random_windows(rng, b, int(rng.integers(1, 300)), bool(case % 2))
    else:
        b = synth_batch(eng, 0xA000 + case, n_rec, lo, hi, span=span, n_contig=n_contig)
        width = int(rng.choice([1, 50, 1000, 20_000, 100_000, 2 * mean_bases]))
        step = max(1, int(width * rng.choice([0.25, 0.83, 1.0, 3.0])))
        top = span + mean_bases * 2
        n_w = min(4000, top // step + 1)
        st = (np.arange(n_w, dtype=np.uint64) * np.uint64(step * max(1, (top // step + 1) // n_w)))
        st = np.tile(st, n_contig)
        wc = np.repeat(np.arange(n_contig, dtype=np.uint32), n_w)
        if case % 3 == 2:  # an unsorted list (the plan sorts it or the records go back)
            perm = rng.permutation(len(st))
            st, wc = st[perm], wc[perm]
        w = (wc, st, st + np.uint64(width))
        # (bounded: about 3e5 rows and 2e7 clipped ops a case)
        eff = step * max(1, (top // step + 1) // n_w)
        per_rec = min(n_w, (mean_bases + width) // eff + 1)
        per_row = min((lo + hi) // 2, width // 150 + 2)
        cap = max(1, min(300_000 // per_rec, 20_000_000 // (per_rec * per_row)))
        if n_rec > cap:
            n_rec = cap
            b = synth_batch(eng, 0xA000 + case, n_rec, lo, hi, span=span, n_contig=n_contig)
        what = f"case {case}: {n_rec} records of {lo}-{hi} ops, span {span}, windows of {width} every {step}"
    policy = [FUSED, 0, FUSED | rustybam_amd.BSEARCH_LEGACY][case % 3]
    c = lift_both(eng, oracle, b, w, policy, what)
    tiles += int(c["phase"][3]); back += int(c["phase"][4]); rows_seen += int(c["n_hits"])
    ms = int(rng.choice([0, 5, 30, 100, 1000]))
    c = break_both(eng, oracle, b, ms, FUSED | rustybam_amd.BREAK_ONE_WALK, what + f", break {ms}")
    tiles += int(c["phase"][3]); back += int(c["phase"][4]); rows_seen += int(c["n_hits"])
    if case % 4 == 0:
        break_both(eng, oracle, b, ms, FUSED, what + f", break {ms}, two walks")
print(f"soak_tile ok: {n_cases} cases from {seed0}, {tiles} tiles, {back} records handed back, {rows_seen} rows")
