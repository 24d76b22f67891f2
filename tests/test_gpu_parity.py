"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Bit-exact: integer / index work."""
import json
import os

import numpy as np
import pytest

import rustybam_amd
from rbtest_util import (batch_args, compare_hits, random_batch, random_windows, read_bed, read_paf,
                         recs_from_lines, unpack)

pytestmark = pytest.mark.gpu


def _obatch(oracle, b):
    return oracle.Batch(*batch_args(b), b["contig"])


def _check_scan(engine, oracle, b, what):
    red, norm = engine.scan_records(*batch_args(b))
    ob = _obatch(oracle, b)
    ored, onorm = oracle.reduce(ob), oracle.normalize(ob)
    for k in ("t_bases", "q_bases", "nmatch", "aln_len", "equal", "diff", "ins", "del", "matches", "ins_events",
              "del_events", "status"):
        bad = np.nonzero(red[k] != ored[k])[0]
        assert len(bad) == 0, f"{what}: reduce.{k} differs at {bad[:5]}: {red[k][bad[:5]]} vs {ored[k][bad[:5]]}"
    for k in ("id_by_all", "id_by_events", "id_by_matches"):  # f32 computed with the same three IEEE ops: bit-exact
        a, o = red[k].view(np.uint32), ored[k].view(np.uint32)
        nan = np.isnan(red[k]) & np.isnan(ored[k])
        bad = np.nonzero((a != o) & ~nan)[0]
        assert len(bad) == 0, f"{what}: reduce.{k} differs at {bad[:5]}"
    bad = np.nonzero(norm["status"] != onorm["status"])[0]
    assert len(bad) == 0, f"{what}: norm.status differs at {bad[:5]}: {norm['status'][bad[:5]]} vs {onorm['status'][bad[:5]]}"
    ok = onorm["status"] == 0
    for k in ("t_st", "t_en", "q_st", "q_en", "first_op", "n_ops", "nmatch", "aln_len"):
        bad = np.nonzero(ok & (norm[k] != onorm[k]))[0]
        assert len(bad) == 0, f"{what}: norm.{k} differs at {bad[:5]}: {norm[k][bad[:5]]} vs {onorm[k][bad[:5]]}"
    for k in ("lead_ops", "trail_ops"):
        bad = np.nonzero((onorm["status"] != oracle.PANIC_EMPTY_CIGAR) & (norm[k] != onorm[k]))[0]
        assert len(bad) == 0, f"{what}: norm.{k} differs at {bad[:5]}"
    return red, norm


def _check_liftover(engine, oracle, b, w, policy, what):
    rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy)
    orows, oops = oracle.liftover(_obatch(oracle, b), *w, policy=policy)
    compare_hits(rows, ops, orows, oops, what)
    # the same call with the record scan fused into the clip kernel (RB_LIFT_FUSED_SCAN): identical normalised rows, and
    # identical hit rows for every record the reference would not panic on (rows of the others only carry the status)
    frows, fops, fnorm, fcnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy | rustybam_amd.LIFT_FUSED_SCAN)
    assert np.array_equal(fnorm["status"], norm["status"]), f"{what}: fused norm.status"
    ok = norm["status"] == 0
    for k in ("t_st", "t_en", "q_st", "q_en", "first_op", "n_ops", "nmatch", "aln_len", "lead_ops", "trail_ops"):
        bad = np.nonzero(ok & (fnorm[k] != norm[k]))[0]
        assert len(bad) == 0, f"{what}: fused norm.{k} differs at {bad[:5]}: {fnorm[k][bad[:5]]} vs {norm[k][bad[:5]]}"
    # flags: the fused scan judges regularity on the KEPT ops only, the stand-alone scan on the whole record (stripped end
    # indels included), so "regular" may be granted more often; the other bits agree, except HAS_M which the fused scan
    # also derives from the kept ops
    fl, nl = fnorm["flags"].astype(np.int64), norm["flags"].astype(np.int64)
    assert ((fl & 8) == 0).all(), f"{what}: a provisional row leaked"
    assert ((fl & 2) == (nl & 2))[ok].all(), f"{what}: fused norm.flags STRIPPED"
    assert ((nl & 1) <= (fl & 1))[ok].all(), f"{what}: fused norm.flags REGULAR"
    keep = ok[frows["rec"]] if len(frows) else np.zeros(0, bool)
    assert (frows["status"][~keep] != 0).all(), f"{what}: fused rows of panicking records must carry a status"
    compare_hits(frows[keep], fops, orows, oops, what + " (fused)")
    return rows, cnt


def test_scan_records_fixture(engine, oracle, golden):
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand,
             contig=r.contig)
    red, norm = _check_scan(engine, oracle, b, "fixture")
    assert (red["status"] == 0).all() and (norm["flags"] & 1).all()  # minimap2 cigars are all "regular"
    assert int(red["aln_len"].astype(np.uint64).sum()) == 142350580  # aligned units of the fixture: the sum of every CIGAR length in asm_small.paf
    # first data line of `rb stats --paf`
    assert (int(red["equal"][0]), int(red["diff"][0]), int(red["del_events"][0]), int(red["ins_events"][0]),
            int(red["del"][0]), int(red["ins"][0])) == (10692453, 11023, 1441, 1300, 41072, 40500)
    assert [oracle.f32_display(float(red[k][0])) for k in ("id_by_matches", "id_by_events", "id_by_all")] == \
        ["99.89702", "99.87144", "99.14145"]


@pytest.mark.parametrize("mode", ["regular", "indel_ends", "wild", "mixed"])
def test_scan_records_random(engine, oracle, mode):
    import zlib
    rng = np.random.default_rng(zlib.crc32(mode.encode()))
    b = random_batch(rng, 600, mode, break_frac=0.1)
    # empty cigars and single-op records
    _check_scan(engine, oracle, b, mode)


def test_scan_records_edge_cases(engine, oracle):
    lines = ["Q 10 0 0 + T 10 0 0 0 0 60",  # empty cigar
             "Q 10 0 3 + T 10 0 0 0 0 60 cg:Z:3I", "Q 10 0 0 + T 10 0 3 0 0 60 cg:Z:3D",  # all indel
             "Q 10 0 5 + T 10 0 5 0 0 60 cg:Z:5=", "Q 10 0 7 - T 10 0 5 0 0 60 cg:Z:2I5=",
             "Q 10 0 5 - T 10 0 8 0 0 60 cg:Z:5=3D", "Q 10 0 6 + T 10 0 8 0 0 60 cg:Z:2D1I5=",
             "Q 10 0 6 + T 10 0 8 0 0 60 cg:Z:1I2D5=", "Q 10 0 5 + T 10 0 9 0 0 60 cg:Z:5=2D1I2D"]
    r = recs_from_lines(lines)
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand,
             contig=r.contig)
    _check_scan(engine, oracle, b, "edge")


def test_liftover_known_answer_ka1(engine, oracle, golden):
    k = json.load(open(os.path.join(golden, "known_answers.json")))["KA1_liftover"]
    r = recs_from_lines(k["records"])
    wc = np.zeros(len(k["windows"]), np.uint32)
    ws = np.array([w[1] for w in k["windows"]], np.uint64)
    we = np.array([w[2] for w in k["windows"]], np.uint64)
    for pol in (rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY):
        rows, ops, norm, cnt = engine.liftover(*r.arrays(), r.contig, wc, ws, we, policy=pol)
        assert len(rows) == 12 and (rows["status"] == 0).all()
        for row in rows:
            rec, win = int(row["rec"]), int(row["win"])
            assert (int(row["q_st"]), int(row["q_en"])) == (k["q_st"][2 * win + rec], k["q_en"][2 * win + rec])
        assert rows["flags"][5] & 1  # window strictly containing the record: own id
        # windows here are not monotone (st 14,14,12,12,5,5): still the streaming kernel (records are regular)
        assert cnt["n_generic"] == 0


@pytest.mark.parametrize("policy", [rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY])
def test_liftover_fixture_bed(engine, oracle, golden, policy):
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand,
             contig=r.contig)
    wc, ws, we, ids = read_bed(os.path.join(golden, "asm_small.bed"), r.contig_names)
    rows, cnt = _check_liftover(engine, oracle, b, (wc, ws, we), policy, "fixture bed")
    assert int((rows["status"] == 0).sum()) == 12  # 12 output records


def test_liftover_fixture_tiled_100kb(engine, oracle, golden):
    """the headline shape: sliding 100 kb windows over every target (1,636 windows -> 1,657 records)"""
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand,
             contig=r.contig)
    wc, ws, we = [], [], []
    tlen = {}
    for n, L in zip(r.t_name, r.t_len):
        tlen.setdefault(n, L)
    for name, L in tlen.items():
        for s in range(0, L, 100000):
            wc.append(r.contig_names[name]); ws.append(s); we.append(min(s + 100000, L))
    w = (np.array(wc, np.uint32), np.array(ws, np.uint64), np.array(we, np.uint64))
    rows, cnt = _check_liftover(engine, oracle, b, w, rustybam_amd.BSEARCH_MODERN, "tiled")
    assert int((rows["status"] == 0).sum()) == 1657
    assert cnt["n_generic"] < len(rows) // 50  # the streaming kernel did the work


@pytest.mark.parametrize("mode,monotone", [("regular", True), ("regular", False), ("indel_ends", True), ("spliced", True),
                                           ("spliced", False), ("wild", True), ("mixed", False), ("mixed", True)])
@pytest.mark.parametrize("policy", [rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY])
def test_liftover_random(engine, oracle, mode, monotone, policy):
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{mode}{monotone}{policy}".encode()))
    for rep in range(3):
        b = random_batch(rng, 300, mode, n_contig=3)
        w = random_windows(rng, b, 120, monotone)
        rows, cnt = _check_liftover(engine, oracle, b, w, policy, f"{mode} mono={monotone} rep={rep}")
        if mode == "regular" and policy == rustybam_amd.BSEARCH_MODERN:
            assert cnt["n_generic"] == 0  # sorted or not, regular records never leave the streaming kernel
        if mode == "spliced":
            assert cnt["n_generic"] < len(rows) // 4  # N ops stay on the fast path unless one sits at an end of the record


def test_liftover_many_windows_per_record(engine, oracle):
    """> 64 windows per record forces several streaming passes over one record"""
    rng = np.random.default_rng(11)
    b = random_batch(rng, 40, "regular", n_contig=1, long_frac=1.0)
    hi = int(b["t_en"].max())
    st = np.arange(0, hi, 37, dtype=np.uint64)
    w = (np.zeros(len(st), np.uint32), st, st + 50)
    rows, cnt = _check_liftover(engine, oracle, b, w, rustybam_amd.BSEARCH_MODERN, "dense windows")
    assert len(rows) > 64 * 40


def test_liftover_empty_inputs(engine, oracle):
    rng = np.random.default_rng(3)
    b = random_batch(rng, 50, "regular")
    e32, e64 = np.zeros(0, np.uint32), np.zeros(0, np.uint64)
    rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], e32, e64, e64)
    assert len(rows) == 0
    z = random_batch(rng, 0, "regular")
    rows, ops, norm, cnt = engine.liftover(*batch_args(z), z["contig"], np.zeros(1, np.uint32), np.zeros(1, np.uint64),
                                           np.ones(1, np.uint64))
    assert len(rows) == 0


@pytest.mark.parametrize("policy", [rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY])
@pytest.mark.parametrize("max_size", [0, 2, 100])
def test_break_paf_fixture_and_random(engine, oracle, golden, policy, max_size):
    r = read_paf(os.path.join(golden, "asm_small.paf"))
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand,
             contig=r.contig)
    rng = np.random.default_rng(5 + max_size)
    for what, bb in (("fixture", b), ("mixed", random_batch(rng, 300, "mixed")), ("regular", random_batch(rng, 300, "regular"))):
        rows, ops, norm, cnt = engine.break_paf(*batch_args(bb), max_size, policy=policy)
        orows, oops = oracle.break_paf(_obatch(oracle, bb), max_size, policy=policy)
        compare_hits(rows, ops, orows, oops, f"break {what} max={max_size}")
        # the same with the record scan fused into the clip kernel
        frows, fops, fnorm, _ = engine.break_paf(*batch_args(bb), max_size, policy=policy | rustybam_amd.LIFT_FUSED_SCAN)
        assert np.array_equal(fnorm["status"], norm["status"]) and not (fnorm["flags"] & 8).any()
        keep = (norm["status"] == 0)[frows["rec"]] if len(frows) else np.zeros(0, bool)
        compare_hits(frows[keep], fops, orows, oops, f"break {what} max={max_size} (fused)")
        if what == "fixture" and max_size == 100 and policy == rustybam_amd.BSEARCH_MODERN:
            assert int((rows["status"] == 0).sum()) == 2447


def test_swap(engine, oracle):
    rng = np.random.default_rng(9)
    b = random_batch(rng, 400, "mixed")
    got = engine.swap(b["ops"], b["op_off"], b["strand"])
    want = oracle.swap(_obatch(oracle, b))
    assert np.array_equal(got, want)


def test_synth_device_matches_host(engine):
    """the device generator used by bench.py produces the bytes of the host generator"""
    import ctypes as C
    from rustybam_amd import capi
    n = capi.synth_n_ops(0x5EED0003, 17, 64, 1000, 9000)
    off = np.zeros(len(n) + 1, np.uint64)
    off[1:] = np.cumsum(n)
    host = capi.synth_fill_ops_host(0x5EED0003, 17, off)
    L = engine.L
    d_off, d_ops = C.c_void_p(), C.c_void_p()
    L.rb_dev_alloc(engine.ctx, C.c_size_t(off.nbytes), C.byref(d_off))
    L.rb_dev_alloc(engine.ctx, C.c_size_t(host.nbytes), C.byref(d_ops))
    L.rb_dev_upload(engine.ctx, d_off, C.c_void_p(off.ctypes.data), C.c_size_t(off.nbytes))
    assert L.rb_dev_synth_fill_ops(engine.ctx, C.c_uint64(0x5EED0003), C.c_uint64(17), C.c_uint64(len(n)), d_off, d_ops) == 0
    dev = np.zeros_like(host)
    assert L.rb_dev_download(engine.ctx, C.c_void_p(dev.ctypes.data), d_ops, C.c_size_t(host.nbytes)) == 0
    L.rb_dev_free(engine.ctx, d_off); L.rb_dev_free(engine.ctx, d_ops)
    assert np.array_equal(dev, host)


def _rebuild_from_descriptor(b, row, desc):
    """clip descriptor -> packed ops, the way a host that still holds the record's cigar would"""
    first, n, flen, llen = (int(x) for x in desc)
    o0 = int(b["op_off"][int(row["rec"])])
    ops = b["ops"][o0 + first:o0 + first + n].copy()
    if not (int(row["flags"]) & rustybam_amd.HIT_INSIDE):
        if n == 1:
            ops[0] = (int(row["aln_len"]) << 4) | (int(ops[0]) & 15)
        else:
            ops[0] = (flen << 4) | (int(ops[0]) & 15)
            ops[-1] = (llen << 4) | (int(ops[-1]) & 15)
    return ops


@pytest.mark.parametrize("mode", ["regular", "indel_ends", "mixed"])
def test_liftover_descriptor_mode_and_early_exit(engine, oracle, mode):
    """RB_LIFT_DESCRIPTORS returns which ops each clip keeps instead of copying them; RB_LIFT_EARLY_EXIT stops
    walking a record after its last window.  Both must describe exactly the clips of the default mode."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(("desc" + mode).encode()))
    b = random_batch(rng, 400, mode, n_contig=2, long_frac=0.3)
    w = random_windows(rng, b, 150, True)
    base_rows, base_ops, _, _ = engine.liftover(*batch_args(b), b["contig"], *w)
    for pol in (rustybam_amd.LIFT_EARLY_EXIT, rustybam_amd.LIFT_DESCRIPTORS,
                rustybam_amd.LIFT_DESCRIPTORS | rustybam_amd.LIFT_EARLY_EXIT):
        rows, ops, _, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=pol)
        assert len(rows) == len(base_rows)
        for k in ("rec", "win", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
            assert np.array_equal(rows[k], base_rows[k]), (pol, k)
        n_desc = 0
        for g, o in zip(rows, base_rows):
            if int(o["status"]) != 0:
                continue
            want = base_ops[int(o["out_off"]):int(o["out_off"]) + int(o["out_n"])]
            if int(g["flags"]) & rustybam_amd.HIT_DESCRIPTOR:
                got = _rebuild_from_descriptor(b, g, ops[int(g["out_off"]):int(g["out_off"]) + 4])
                n_desc += 1
            else:
                got = ops[int(g["out_off"]):int(g["out_off"]) + int(g["out_n"])]
            assert np.array_equal(got, want), (pol, int(g["rec"]), int(g["win"]))
        if pol & rustybam_amd.LIFT_DESCRIPTORS:
            assert n_desc > 0
            if mode == "regular":
                assert n_desc == int((base_rows["status"] == 0).sum())


def test_liftover_long_records_many_segments(engine, oracle):
    """records far longer than one LDS checkpoint segment (20 steps x 256 ops), windows of every size"""
    rng = np.random.default_rng(77)
    b = random_batch(rng, 6, "regular", n_contig=1, max_ops=40, long_frac=0.0)
    from rbtest_util import random_cigar, sums
    cig = [random_cigar(rng, n, "regular") for n in (6000, 12000, 23000)]
    for c in cig:
        R, Q = sums(c)
        ts, qs = int(rng.integers(0, 500)), int(rng.integers(0, 500))
        b["ops"] = np.concatenate([b["ops"], c])
        b["op_off"] = np.append(b["op_off"], b["op_off"][-1] + np.uint64(len(c)))
        for k, v in (("t_st", ts), ("t_en", ts + R), ("q_st", qs), ("q_en", qs + Q), ("strand", ord("-") if len(c) % 2 else ord("+")),
                     ("contig", 0)):
            b[k] = np.append(b[k], np.array([v], b[k].dtype))
    hi = int(b["t_en"].max())
    st = np.sort(rng.integers(0, hi, 200)).astype(np.uint64)
    ln = rng.choice([1, 50, 3000, 200000], 200).astype(np.uint64)
    en = np.maximum.accumulate(st + ln)
    w = (np.zeros(200, np.uint32), st, en)
    rows, cnt = _check_liftover(engine, oracle, b, w, rustybam_amd.BSEARCH_MODERN, "long records")
    assert cnt["n_generic"] == 0


def test_liftover_unsorted_tpos_array(engine, oracle):
    """t_st == 0 and a leading op that consumes no reference: the reference's tpos_aln starts with units at t_pos = -1
    (u64::MAX), is not sorted, and binary_search returns whatever its probe sequence leads to (found by tests/soak/soak.py).
    The generic kernel replays the probe sequence; both generations of the standard library."""
    lines = ["Q 2000 1666 1766 + T 100 0 23 0 0 60 cg:Z:40S2=2N2I1=1D1N3I2=40I3=1=2=1I3D2=1=2N",
             "Q 2000 0 12 + T 100 0 9 0 0 60 cg:Z:5H3=1X2N3=3S", "Q 2000 0 12 - T 100 0 9 0 0 60 cg:Z:3S3=1X2N3=5H",
             "Q 2000 0 65 + T 100 0 23 0 0 60 cg:Z:0=2S40I2D7X1I1D2I3=3=3X3I3D1X"]  # (a zero-length op adds no unit)
    r = recs_from_lines(lines)
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
    ws = np.array([2, 0, 1, 5, 0, 8], np.uint64)
    we = np.array([88, 4, 9, 6, 23, 9], np.uint64)
    w = (np.zeros(len(ws), np.uint32), ws, we)
    for pol in (rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_LEGACY):
        _check_liftover(engine, oracle, b, w, pol, f"unsorted tpos policy {pol}")


# ---- positional output slots of the streaming clip kernel: clips that lose their place must still come out right ----
def _run_slots(oracle, b, w, env_slots, what):
    """rb_host_liftover in a fresh process with RB_DEBUG_SLOTS (the library reads it per call, but a fresh engine keeps this honest)"""
    import subprocess
    import sys
    import tempfile
    import pickle
    code = r'''
import pickle, sys
sys.path.insert(0, %r)
import torch  # noqa: F401 (HIP runtime load order, see conftest)
import rustybam_amd
b, w = pickle.load(open(sys.argv[1], "rb"))
eng = rustybam_amd.Engine(0)
out = {}
for pol in (rustybam_amd.BSEARCH_MODERN, rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN):
    rows, ops, norm, cnt = eng.liftover(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"], b["contig"], *w, policy=pol)
    out[pol] = (rows, ops)
pickle.dump(out, open(sys.argv[2], "wb"))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        pickle.dump((b, w), open(os.path.join(d, "in.pkl"), "wb"))
        env = dict(os.environ)
        if env_slots is not None:
            env["RB_DEBUG_SLOTS"] = str(env_slots)
        r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "in.pkl"), os.path.join(d, "out.pkl")], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out = pickle.load(open(os.path.join(d, "out.pkl"), "rb"))
    orows, oops = oracle.liftover(_obatch(oracle, b), *w)
    for pol, (rows, ops) in out.items():
        compare_hits(rows, ops, orows, oops, f"{what} slots={env_slots} policy={pol}")
    return len(orows)


@pytest.mark.parametrize("slots", [None, 0, 1])
def test_liftover_clips_without_a_slot_are_copied(oracle, slots):
    """windows overlapping 5 deep (more than the kernel has slots), and the slots switched off / cut to one: every clip that loses
    its place goes through rb_k_copy_clips and must be the same bytes"""
    rng = np.random.default_rng(4242)
    b = random_batch(rng, 120, "regular", n_contig=1, long_frac=0.6)
    hi = int(b["t_en"].max())
    st = np.arange(0, hi, 40, dtype=np.uint64)
    w = (np.zeros(len(st), np.uint32), st, st + 200)          # five windows over every base
    n = _run_slots(oracle, b, w, slots, "deep windows")
    assert n > 2000


def test_liftover_many_windows_inside_one_op(oracle):
    """disjoint windows that all cut the same long '=' op: their clips are one op each, at the same position of the record -- in a
    positional slot they would sit on top of each other, so all but the first of a class must be copied; also clips that share a
    16-byte group with the previous one of their class"""
    lines = ["Q 200000 0 100000 + T 200000 0 100000 100000 100000 60 cg:Z:100000=",
             "Q 200000 100 50103 - T 200000 1000 51003 0 0 60 cg:Z:3=1X20000=1I2=1D29996="]
    r = recs_from_lines(lines)
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
    st = np.arange(0, 100000, 150, dtype=np.uint64)
    w = (np.zeros(len(st), np.uint32), st, st + 100)           # gaps between the windows: depth 1
    n = _run_slots(oracle, b, w, None, "one op")
    assert n > 900
    st = np.arange(0, 100000, 70, dtype=np.uint64)
    w = (np.zeros(len(st), np.uint32), st, st + 100)           # depth 2, still inside single ops
    _run_slots(oracle, b, w, None, "one op, overlapping")
