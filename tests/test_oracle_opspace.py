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


def _materialise(ops, off, n, fl, ll):
    c = ops[int(off):int(off) + int(n)].copy()
    if fl:
        c[0] = (int(fl) << 4) | (int(c[0]) & 15)
    if ll:
        c[-1] = (int(ll) << 4) | (int(c[-1]) & 15)
    return c


def test_pair_step_opspace_equals_the_per_base_oracle(oracle):
    """trim_overlapping_pafs + truncate_record_by_query in op space (rbo_overlap_split_opspace_arrays, the checker of the full-size trim-paf
    test) against the per-base restatement: split index and score, both cuts' coordinates, nmatch, aln_len and every kept op, on random
    regular pairs (both strands, records of 3 to 900 ops, overlaps from one base to nearly the whole record, two score sets); then a
    second cut of the already cut records THROUGH THE VIEWS (first op / count / end lengths: the batch is never rewritten), against the
    per-base oracle on the materialised records."""
    from test_gpu_trim import _pairs_batch
    total = 0
    for seed, ops_range, scores in ((1, (3, 60), (1, 1, 1)), (2, (60, 200), (2, 3, 5)), (3, (600, 900), (1, 1, 1)), (4, (3, 30), (3, 1, 7)), (5, (100, 400), (1, 1, 1))):
        rng = np.random.default_rng(9100 + seed)
        b, left, right = _pairs_batch(rng, 120, "regular", zero_bias=bool(seed % 2), ops_range=ops_range)
        ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
        want, wout = oracle.overlap_split(ob, left, right, scores)
        n = np.diff(b["op_off"]).astype(np.uint32)
        z = np.zeros(len(n), np.uint32)
        got, bad = oracle.overlap_split_opspace(b["ops"], b["op_off"][:-1], n, z, z, b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], left, right,
                                                scores, n_threads=4)
        assert bad == int((got["status"] != 0).sum())
        ok = want["status"] == 0
        assert ok.sum() > 100 and ((got["status"] == 0) == ok).all(), f"seed {seed}: the port takes exactly the pairs the reference cuts"
        for k in ("split_idx", "split_score", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
            assert np.array_equal(got[k][ok], want[k][ok]), (seed, k)
        assert np.array_equal(got["count"][ok], want["out_n"][ok])
        # the cuts, op by op; and the views of the cut records for the second round
        off2, n2, fl2, ll2 = b["op_off"][:-1].copy(), n.copy(), z.copy(), z.copy()
        c2 = {k: b[k].copy() for k in ("t_st", "t_en", "q_st", "q_en")}
        for i in np.nonzero(ok)[0]:
            for s, rec in ((0, int(left[i])), (1, int(right[i]))):
                g = got[i]
                c = _materialise(b["ops"], int(b["op_off"][rec]) + int(g["first"][s]), g["count"][s], g["first_len"][s], g["last_len"][s])
                o = wout[int(want["out_off"][i][s]):int(want["out_off"][i][s]) + int(want["out_n"][i][s])]
                assert np.array_equal(c, o), (seed, i, s)
                off2[rec] += g["first"][s]
                n2[rec], fl2[rec], ll2[rec] = g["count"][s], g["first_len"][s], g["last_len"][s]
                for k in c2:
                    c2[k][rec] = g[k][s]
            total += 1
        # second round: every cut pair once more with a query overlap made by pulling the right record's start back into the left one --
        # the records now exist only as views; the per-base oracle gets them materialised
        keep = [i for i in np.nonzero(ok)[0] if n2[left[i]] >= 3 and n2[right[i]] >= 3]
        if not keep:
            continue
        cig, cl, cr = [], [], []
        m = {k: [] for k in ("t_st", "t_en", "q_st", "q_en", "strand")}
        v_off, v_n, v_fl, v_ll = [], [], [], []
        for i in keep:
            l, r = int(left[i]), int(right[i])
            ql, qr = int(c2["q_en"][l] - c2["q_st"][l]), int(c2["q_en"][r] - c2["q_st"][r])
            if min(ql, qr) < 3:
                continue
            ov = int(rng.integers(1, min(ql, qr)))
            shift = int(c2["q_en"][l]) - ov - int(c2["q_st"][r])  # move the right record so that it starts ov bases before the left one ends
            for rec, sh in ((l, 0), (r, shift)):
                cig.append(_materialise(b["ops"], off2[rec], n2[rec], fl2[rec], ll2[rec]))
                v_off.append(off2[rec]); v_n.append(n2[rec]); v_fl.append(fl2[rec]); v_ll.append(ll2[rec])
                m["t_st"].append(c2["t_st"][rec]); m["t_en"].append(c2["t_en"][rec])
                m["q_st"].append(int(c2["q_st"][rec]) + sh); m["q_en"].append(int(c2["q_en"][rec]) + sh); m["strand"].append(b["strand"][rec])
            cl.append(len(cig) - 2); cr.append(len(cig) - 1)
        off = np.zeros(len(cig) + 1, np.uint64)
        off[1:] = np.cumsum([len(c) for c in cig])
        arr = lambda k, dt=np.uint64: np.array(m[k], dt)  # noqa: E731
        ob2 = oracle.Batch(np.concatenate(cig), off, arr("t_st"), arr("t_en"), arr("q_st"), arr("q_en"), arr("strand", np.uint8))
        want2, wout2 = oracle.overlap_split(ob2, cl, cr, scores)
        got2, _ = oracle.overlap_split_opspace(b["ops"], v_off, v_n, v_fl, v_ll, arr("t_st"), arr("t_en"), arr("q_st"), arr("q_en"), arr("strand", np.uint8), cl, cr,
                                               scores, n_threads=2)
        ok2 = want2["status"] == 0
        assert ((got2["status"] == 0) == ok2).all()
        for k in ("split_idx", "split_score", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
            assert np.array_equal(got2[k][ok2], want2[k][ok2]), (seed, "second round", k)
        for i in np.nonzero(ok2)[0]:
            for s, rec in ((0, cl[i]), (1, cr[i])):
                g = got2[i]
                first_len = g["first_len"][s]
                c = _materialise(cig[rec], int(g["first"][s]), g["count"][s], first_len, g["last_len"][s])
                o = wout2[int(want2["out_off"][i][s]):int(want2["out_off"][i][s]) + int(want2["out_n"][i][s])]
                assert np.array_equal(c, o), (seed, "second round", i, s)
        total += int(ok2.sum())
    assert total > 800
