"""N > 1 path on CPU: two gloo ranks each own a contiguous, op-balanced record range (rustybam_amd.shard),
run liftover on their shard and rank 0 gathers host-side.  The gathered rows must be identical to the
single-process result.  There is no GPU here, so each rank's compute stand-in is the oracle; what is under
test is the product's host logic: shard_bounds / shard_slice / gather_rows / canonical_sort."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from rustybam_amd import shard
from oracle import pyoracle
from rbtest_util import random_batch, random_windows
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
rng = np.random.default_rng(1234)           # every rank builds the same global input
b = random_batch(rng, 400, "mixed", n_contig=3)
w = random_windows(rng, b, 60, True)
bounds = shard.shard_bounds(b["op_off"], 2)
lo, hi = int(bounds[rank]), int(bounds[rank + 1])
s = shard.shard_slice(b, b["op_off"], lo, hi)
rows, ops = pyoracle.liftover(pyoracle.Batch(s["ops"], s["op_off"], s["t_st"], s["t_en"], s["q_st"], s["q_en"],
                                             s["strand"], s["contig"]), *w)
parts = [None, None]
dist.all_gather_object(parts, (rows, ops))   # host-side gather; no data-path collective on a GPU run
dist.barrier()
if rank == 0:
    grows, gops = shard.gather_rows(parts, bounds)
    grows = shard.canonical_sort(grows, b["contig"])
    pickle.dump((grows, gops, bounds), open({out!r}, "wb"))
dist.destroy_process_group()
'''


def test_two_rank_shard_and_gather_matches_single_process(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rbtest_util import random_batch, random_windows
    from rustybam_amd import shard
    import pickle
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gathered.pkl")
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER.format(root=ROOT, port=port, out=out))
        procs = [subprocess.Popen([sys.executable, script, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(2)]
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        grows, gops, bounds = pickle.load(open(out, "rb"))
    rng = np.random.default_rng(1234)
    b = random_batch(rng, 400, "mixed", n_contig=3)
    w = random_windows(rng, b, 60, True)
    rows, ops = oracle.liftover(oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"],
                                             b["strand"], b["contig"]), *w)
    assert 0 < bounds[1] < 400 and len(rows) > 50
    # op-balanced split
    tot = int(b["op_off"][-1])
    assert abs(int(b["op_off"][bounds[1]]) - tot // 2) <= int(np.diff(b["op_off"].astype(np.int64)).max())
    assert len(grows) == len(rows)
    for k in ("rec", "win", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
        assert np.array_equal(grows[k], rows[k]), k
    for g, o in zip(grows, rows):
        assert np.array_equal(gops[int(g["out_off"]):int(g["out_off"]) + int(g["out_n"])],
                              ops[int(o["out_off"]):int(o["out_off"]) + int(o["out_n"])])


def test_shard_bounds_properties():
    from rustybam_amd import shard
    rng = np.random.default_rng(5)
    n = rng.integers(1, 9000, 1000)
    off = np.zeros(1001, np.uint64)
    off[1:] = np.cumsum(n)
    for k in (1, 2, 4, 8):
        b = shard.shard_bounds(off, k)
        assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) >= 0).all()
        per = np.array([int(off[b[i + 1]]) - int(off[b[i]]) for i in range(k)])
        assert per.max() - per.min() <= 2 * n.max()
    assert list(shard.shard_bounds(np.zeros(1, np.uint64), 4)) == [0, 0, 0, 0, 0]
