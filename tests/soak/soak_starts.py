"""Randomised soak of RB_LIFT_OP_STARTS (break-paf / liftover straight off a batch that trim-paf cut in place) against the route through the dense
copy: random batch sizes and record-length ranges around the tile kernel's geometry (records of a few dozen ops: gaps as long as records, tiles of 32
records; records around 2048 ops: tiles of one or two records next to records the per-record kernel takes).  usage: soak_starts.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: F401  (first: conftest.py says why)
from test_gpu_trim import test_break_paf_straight_off_the_trimmed_batch as brk, test_liftover_straight_off_the_trimmed_batch as lift

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(20260606)
for c in range(cases):
    lo = int(rng.choice([24, 40, 100, 300, 700, 1500, 1900]))
    hi = lo + int(rng.choice([8, 40, 200, 700]))
    n = int(rng.integers(2_000, 30_000)) // 4 * 4
    (brk if c % 3 else lift)(n, lo, hi)
    print(f"case {c}: {'break' if c % 3 else 'liftover'} n {n} ops {lo}-{hi} ok", flush=True)
print(f"starts soak ok: {cases} cases")
