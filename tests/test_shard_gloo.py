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


NF_WORKER = r'''
import os, sys, pickle
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from rustybam_amd import shard
from oracle import pyoracle
from nf_util import random_reads
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
rng = np.random.default_rng(4321)           # every rank builds the same global input
rd = random_reads(rng, 400, n_contig=2, span=30000, long_frac=0.1)
rg = [(0, 100, 21000), (1, 5000, 5001), (1, 0, 30000), (0, 9000, 9100)]
pieces = shard.shard_regions([r[0] for r in rg], [r[1] for r in rg], [r[2] for r in rg], 2)[rank]
# reference end of every read (what rb_k_nf_read_spans computes on a GPU run)
ref = np.array([sum(int(w) >> 4 for w in rd.ops[int(rd.op_off[i]):int(rd.op_off[i + 1])] if (0x18D >> (int(w) & 15)) & 1) for i in range(rd.n)])
ok = (rd.tid >= 0) & ((rd.flag & 0x704) == 0)
end = np.where(ok, rd.pos + np.maximum(ref, 1), 0)
lo, hi = shard.shard_reads(rd.tid, rd.pos, end, [(rg[r][0], st, en) for r, st, en in pieces])
sub = rd.slice(lo, hi)
chunks = []
for r, st, en in pieces:                    # the rank's compute stand-in: the oracle on its slice of reads
    cnt = np.zeros((en - st, 4), np.uint32)
    rc, p, c = pyoracle.nucfreq(*sub.args(), rg[r][0], st, en)
    assert rc == 0
    cnt[p.astype(np.int64) - st] = c
    cnt[p.astype(np.int64) - st, 0] |= 0x80000000
    chunks.append(cnt)
mine = np.concatenate(chunks) if chunks else np.zeros((0, 4), np.uint32)
parts = [None, None]
dist.all_gather_object(parts, mine)
dist.barrier()
if rank == 0:
    allp = shard.shard_regions([r[0] for r in rg], [r[1] for r in rg], [r[2] for r in rg], 2)
    pickle.dump((shard.gather_counts(parts, allp, [r[1] for r in rg], [r[2] for r in rg]), allp), open({out!r}, "wb"))
dist.destroy_process_group()
'''


def test_two_rank_nucfreq_position_shards_match_single_process(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from nf_util import random_reads
    import pickle
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gathered.pkl")
        script = os.path.join(d, "worker.py")
        open(script, "w").write(NF_WORKER.format(root=ROOT, port=port, out=out))
        procs = [subprocess.Popen([sys.executable, script, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        got, pieces = pickle.load(open(out, "rb"))
    assert len(pieces[0]) >= 1 and len(pieces[1]) >= 1                 # both ranks had work
    rg_st = [100, 5000, 0, 9000]
    assert all((st - rg_st[r]) % 4096 == 0 for r, st, _e in pieces[0] + pieces[1])   # cuts fall on tile edges of the region
    rng = np.random.default_rng(4321)
    rd = random_reads(rng, 400, n_contig=2, span=30000, long_frac=0.1)
    rg = [(0, 100, 21000), (1, 5000, 5001), (1, 0, 30000), (0, 9000, 9100)]
    want = []
    for t, st, en in rg:
        cnt = np.zeros((en - st, 4), np.uint32)
        rc, p, c = oracle.nucfreq(*rd.args(), t, st, en)
        assert rc == 0
        cnt[p.astype(np.int64) - st] = c
        cnt[p.astype(np.int64) - st, 0] |= 0x80000000
        want.append(cnt)
    assert np.array_equal(got, np.concatenate(want))


def test_trim_paf_shards_by_query_group_concatenate_to_the_whole(oracle, golden, tmp_path):
    """trim-paf partitions by query name: whole groups per rank, outputs concatenated in shard order.  The per-rank
    compute stand-in is the oracle CLI; under test is shard_query_groups."""
    from rustybam_amd import shard
    lines = [l for l in open(os.path.join(golden, "asm_small.paf")).read().split("\n") if l]
    q = [l.split("\t")[0] for l in lines]
    w = [l.count("=") + l.count("X") + 1 for l in lines]           # ~ ops per record
    rc, whole = oracle.cli("trim-paf", os.path.join(golden, "asm_small.paf"))
    assert rc == 0 and whole.count(b"\n") > 100
    for n_shards in (2, 3):
        parts = shard.shard_query_groups(q, w, n_shards)
        assert sorted(np.concatenate(parts).tolist()) == list(range(len(lines)))      # a partition of the records
        assert len({q[i] for i in parts[0]} & {q[i] for i in parts[1]}) == 0           # no group is split
        got = b""
        for s, idx in enumerate(parts):
            f = tmp_path / f"shard{n_shards}_{s}.paf"
            f.write_text("\n".join(lines[i] for i in idx) + "\n")
            rc, out = oracle.cli("trim-paf", str(f))
            assert rc == 0
            got += out
        assert got == whole
