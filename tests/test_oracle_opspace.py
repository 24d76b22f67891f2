"""The op-space CPU baseline (oracle/rb_opspace.c, SURVEY.md 8d "also time the op-space CPU path") against the per-base oracle: same
rows, same clipped CIGARs, same order, on random regular batches (window edges on every kind of op boundary), on the bench
workload's synthetic records and on the reference fixture; batches it does not take are refused, not mangled."""
import os

import numpy as np

from rbtest_util import random_batch, random_windows, read_paf


def _same(oracle, b, w, what):
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"])
    want_rows, want_ops = oracle.liftover(ob, *w)
    got = oracle.liftover_opspace(ob, *w, n_threads=4)
    assert got is not None, what
    rows, ops = got
    assert len(rows) == len(want_rows), what
    for k in ("rec", "win", "status", "flags", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
        sel = want_rows["status"] == 0 if k not in ("rec", "win", "status") else slice(None)
        assert np.array_equal(rows[k][sel], want_rows[k][sel]), (what, k)
    for g, o in zip(rows, want_rows):
        if o["status"] == 0:
            assert np.array_equal(ops[int(g["out_off"]):int(g["out_off"]) + int(g["out_n"])],
                                  want_ops[int(o["out_off"]):int(o["out_off"]) + int(o["out_n"])]), what
    return len(rows)


def test_random_regular_batches(oracle):
    total = 0
    for seed in range(12):
        rng = np.random.default_rng(7000 + seed)
        b = random_batch(rng, 150, "regular", n_contig=3, long_frac=0.2)
        if seed % 3 == 0:  # coordinates that start at 0
            b["t_en"] = (b["t_en"] - b["t_st"]).astype(np.uint64)
            b["t_st"] = np.zeros_like(b["t_st"])
        w = random_windows(rng, b, 300, monotone=bool(seed % 2))
        total += _same(oracle, b, w, f"seed {seed}")
    assert total > 5000


def test_bench_workload_sample_and_fixture(oracle, golden):
    from rustybam_amd import capi, workload as wl
    seed, n = wl.SEED_CONFIG3, 48
    nops = wl.n_ops(seed, 0, n)
    off = wl.op_offsets(nops)
    ops = capi.synth_fill_ops_host(seed, 0, off)
    b0 = oracle.Batch(ops, off, np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64),
                      np.full(n, ord("+"), np.uint8), np.zeros(n, np.uint32))
    red = oracle.reduce(b0)
    t_st, t_en, q_st, q_en, strand = wl.headers(seed, 0, red["t_bases"], red["q_bases"], "uniform")
    b = dict(ops=ops, op_off=off, t_st=t_st, t_en=t_en, q_st=q_st, q_en=q_en, strand=strand, contig=np.zeros(n, np.uint32))
    assert _same(oracle, b, wl.sliding_windows(3000), "config 3 sample") > 300
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    fb = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
    rng = np.random.default_rng(3)
    assert _same(oracle, fb, random_windows(rng, fb, 400), "fixture") > 100


def test_refuses_what_it_does_not_take(oracle):
    rng = np.random.default_rng(5)
    b = random_batch(rng, 40, "wild", n_contig=1)
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"])
    assert oracle.liftover_opspace(ob, *random_windows(rng, b, 10)) is None


def _same_break(oracle, b, max_size, what):
    ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"])
    want_rows, want_ops = oracle.break_paf(ob, max_size)
    got = oracle.break_opspace(ob, max_size, n_threads=4)
    assert got is not None, what
    rows, ops = got
    assert len(rows) == len(want_rows), (what, len(rows), len(want_rows))
    for k in ("rec", "win", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
        sel = want_rows["status"] == 0 if k not in ("rec", "win", "status") else slice(None)
        assert np.array_equal(rows[k][sel], want_rows[k][sel]), (what, k)
    for g, o in zip(rows, want_rows):
        if o["status"] == 0:
            assert np.array_equal(ops[int(g["out_off"]):int(g["out_off"]) + int(g["out_n"])],
                                  want_ops[int(o["out_off"]):int(o["out_off"]) + int(o["out_n"])]), what
    return len(rows)


def test_break_paf_opspace_equals_the_per_base_oracle(oracle, golden):
    """liftover.rs:182-226 in op space (rbo_break_opspace_arrays) against the per-base restatement: random regular batches at several
    --max-size values (0: every indel cuts), the bench workload's records, the reference fixture at the README's --max-size 100."""
    from rustybam_amd import capi, workload as wl
    total = 0
    for seed in range(8):
        rng = np.random.default_rng(9100 + seed)
        b = random_batch(rng, 120, "regular", n_contig=2, long_frac=0.2)
        if seed % 3 == 0:
            b["t_en"] = (b["t_en"] - b["t_st"]).astype(np.uint64)
            b["t_st"] = np.zeros_like(b["t_st"])
        total += _same_break(oracle, b, (0, 2, 10, 100)[seed % 4], f"seed {seed}")
    assert total > 2000
    seed, n = wl.SEED_CONFIG3, 40
    nops = wl.n_ops(seed, 0, n)
    off = wl.op_offsets(nops)
    ops = capi.synth_fill_ops_host(seed, 0, off)
    b0 = oracle.Batch(ops, off, np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64),
                      np.full(n, ord("+"), np.uint8), np.zeros(n, np.uint32))
    red = oracle.reduce(b0)
    t_st, t_en, q_st, q_en, strand = wl.headers(seed, 0, red["t_bases"], red["q_bases"], "uniform")
    b = dict(ops=ops, op_off=off, t_st=t_st, t_en=t_en, q_st=q_st, q_en=q_en, strand=strand, contig=np.zeros(n, np.uint32))
    assert _same_break(oracle, b, 100, "config 3 sample") > 80
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    fb = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
    assert _same_break(oracle, fb, 100, "fixture") == 2447  # 2,447 pieces under the modern policy
