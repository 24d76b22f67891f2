"""CIGAR lengths of 2^28 and more (rust-htslib's Cigar holds a u32; SURVEY.md 8c "maximum sizes").

The packed word has 28 bits of length; a longer op takes a second, CONTINUATION word (include/rustybam_amd.h).  The oracle keeps
Cigar(op, u32 len) whole inside (oracle/rb_oracle.h, rbo_cig) and speaks words at its array boundary, so these tests compare the
general kernels (the only ones that ever see such a record) with the per-base restatement on records of ~3e8 units: 17 bytes a
unit in the oracle, 5 GB and a few seconds a record.  CPU part: the oracle's word form.  GPU part: parity through the C ABI and
the `rb` front end.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from rbtest_util import CONT, batch_args, compare_hits, pack, recs_from_lines, unpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RB = os.path.join(ROOT, "rustybam_amd", "rb")
REF, QRY, MAT = "MDN=X", "MIS=X", "M=X"


def line(name, cg, t_st=100, q_st=50, strand="+", contig="chrBig", t_len=700_000_000):
    ops = [(int(a), b) for a, b in re.findall(r"(\d+)([MIDNSHP=X])", cg)]
    R = sum(n for n, o in ops if o in REF)
    Q = sum(n for n, o in ops if o in QRY)
    M = sum(n for n, o in ops if o in MAT)
    return f"{name}\t{q_st + Q + 1000}\t{q_st}\t{q_st + Q}\t{strand}\t{contig}\t{t_len}\t{t_st}\t{t_st + R}\t{M}\t{R + Q}\t60\tcg:Z:{cg}"


# every way a long op can meet the walk: in the middle, as the exact power of two (low word zero), one below it (a single word: the
# record stays regular and takes the streaming kernel), at the record's ends (the end-indel strip and its quirks count OPS), on
# either strand, as a long indel for break-paf, and as the MERGE of two shorter neighbours (no continuation word in the input)
CASES = [
    line("qA", "1000=5X300000000=3I2000=1X500="),
    line("qB", "20=10I268435456D30=5X40=", strand="-"),
    line("qC", "268435460I100=2D50=7I", t_st=0, q_st=0),
    line("qD", "5I100=2D50=268435999I", strand="-"),
    line("qE", "200000000=100000000=5X10="),
    line("qF", "150=300000000D20=3X7="),
    line("qG", "100=2X50="),
    line("qH", "10X268435455=5X"),
    line("qI", "40=268435457I30=5D9=", t_st=5),
    line("qJ", "268435460D3I100=2D50=7I", t_st=0, q_st=0),          # (the reference panics on this one: the quirks of paf.rs:673, :690-701)
]
W = [(500, 1200), (1000, 200_000_000), (150_000_000, 300_002_000), (268_435_500, 268_436_700), (0, 700_000_000),
     (100, 300), (300_000_900, 300_004_000), (268_435_456 + 90, 268_435_456 + 400), (2, 268_435_470), (268_435_455, 268_435_465)]


def _batch():
    r = recs_from_lines(CASES)
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
    w = (np.zeros(len(W), np.uint32), np.array([a for a, _ in W], np.uint64), np.array([e for _, e in W], np.uint64))
    return r, b, w


# ------------------------------------------------------------------------------------------------ CPU: the oracle's word form
def test_words_round_trip_through_the_oracle(oracle):
    L = oracle.lib()
    L.rbo_cigar_to_string.restype = C.c_size_t
    for cg in ("5=", "268435455=", "268435456=", "268435457I3=", "4294967295D", "1X4026531840=7I", "300000000=300000000="):
        b = cg.encode()
        ops, n = C.POINTER(C.c_uint32)(), C.c_size_t()
        assert L.rbo_parse_cigar(b, C.c_size_t(len(b)), C.byref(ops), C.byref(n)) == 0
        words = np.ctypeslib.as_array(ops, shape=(n.value,)).copy()
        assert np.array_equal(words, pack(cg)), cg                       # the test helper and the oracle agree on the words
        nums = [int(x) for x in re.findall(r"\d+", cg)]
        assert n.value == len(nums) + sum(1 for x in nums if x >> 28)
        for k in np.nonzero((words & 15) == CONT)[0]:
            assert k > 0 and (words[k - 1] & 15) <= 8 and 1 <= (int(words[k]) >> 4) <= 15
        out = C.c_char_p()
        k = L.rbo_cigar_to_string(words.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(len(words)), C.byref(out))
        assert out.value[:k].decode() == cg == unpack(words)
    for bad in (b"4294967296=", b"99999999999M"):                          # u32::from_str fails: "Unable to parse cigar string."
        ops, n = C.POINTER(C.c_uint32)(), C.c_size_t()
        assert L.rbo_parse_cigar(bad, C.c_size_t(len(bad)), C.byref(ops), C.byref(n)) != 0


def test_oracle_rows_count_words(oracle):
    """first_op / n_ops / lead_ops / trail_ops / out_n of the oracle's array interface count WORDS, like the product's"""
    r, b, w = _batch()
    ob = oracle.Batch(*batch_args(b), b["contig"])
    norm = oracle.normalize(ob)
    names = [ln.split()[0] for ln in CASES]
    assert (norm["status"][:-1] == 0).all() and norm["status"][names.index("qJ")] == oracle.PANIC_INTEGRITY_T
    i = names.index("qC")
    assert (norm["lead_ops"][i], norm["trail_ops"][i], norm["first_op"][i]) == (2, 1, 2)      # 268435460I = two words
    assert int(norm["n_ops"][i]) == 3
    j = names.index("qD")
    assert (norm["lead_ops"][j], norm["trail_ops"][j]) == (1, 2)
    k = names.index("qJ")
    assert (norm["lead_ops"][k], norm["trail_ops"][k]) == (3, 1)                               # 268435460D two words, 3I one
    red = oracle.reduce(ob)
    assert int(red["del_events"][k]) == 2 and int(red["del"][k]) == 268435462 and int(red["ins_events"][j]) == 2


def test_oracle_cli_prints_long_lengths(oracle, tmp_path):
    paf = tmp_path / "long.paf"
    paf.write_text("\n".join(CASES[:1] + CASES[4:5]) + "\n")
    bed = tmp_path / "w.bed"
    bed.write_text("chrBig\t1000\t300002000\n")
    rc, out = oracle.cli("liftover", "--bed", bed, paf)
    assert rc == 0
    cgs = [ln.split("cg:Z:")[1] for ln in out.decode().splitlines()]
    assert cgs == ["100=5X300000000=3I895=", "299999100=5X10="]         # (the second: two neighbours merged, paf.rs:602-620)


# ------------------------------------------------------------------------------------------------ GPU: parity
gpu = pytest.mark.gpu


@gpu
def test_scan_records_with_long_ops(engine, oracle):
    from test_gpu_parity import _check_scan
    _, b, _ = _batch()
    red, norm = _check_scan(engine, oracle, b, "long ops")
    names = [ln.split()[0] for ln in CASES]
    regular = (norm["flags"] & 1) != 0                                   # RB_F_REGULAR
    assert regular[names.index("qH")] and regular[names.index("qG")] and not regular[names.index("qA")] and not regular[names.index("qE")]


@gpu
@pytest.mark.parametrize("policy", [0, 1])
def test_liftover_with_long_ops(engine, oracle, policy):
    from test_gpu_parity import _check_liftover
    _, b, w = _batch()
    _check_liftover(engine, oracle, b, w, policy, f"long ops policy={policy}")
    rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy)
    ok = rows[rows["status"] == 0]
    assert len(ok) >= 30
    # some clips keep a long op whole, some cut it below 2^28, some land inside the bases of the continuation word, and one is
    # the merge of two shorter neighbours
    texts = [unpack(ops[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])]) for h in ok]
    assert any("300000000=" in t for t in texts) and any(t == "268435370=" for t in texts) and any("268435456D" in t for t in texts)
    assert any((ops[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])] & 15 == CONT).any() for h in ok)


@gpu
@pytest.mark.parametrize("max_size", [100, 268435456, 300000000])
def test_break_paf_with_long_ops(engine, oracle, max_size):
    import rustybam_amd
    _, b, _ = _batch()
    for policy in (0, 1):
        for flags in (0, rustybam_amd.LIFT_FUSED_SCAN, rustybam_amd.LIFT_FUSED_SCAN | getattr(rustybam_amd, "BREAK_ONE_WALK", 0)):
            rows, ops, norm, cnt = engine.break_paf(*batch_args(b), max_size, policy=policy | flags)
            orows, oops = oracle.break_paf(oracle.Batch(*batch_args(b), b["contig"]), max_size, policy=policy)
            compare_hits(rows, ops, orows, oops, f"break long ops max={max_size} policy={policy} flags={flags}")


@gpu
def test_swap_with_long_ops(engine, oracle):
    _, b, _ = _batch()
    got = engine.swap(b["ops"], b["op_off"], b["strand"])
    want = oracle.swap(oracle.Batch(*batch_args(b), b["contig"]))
    assert np.array_equal(got, want)
    names = [ln.split()[0] for ln in CASES]
    o0, o1 = int(b["op_off"][names.index("qB")]), int(b["op_off"][names.index("qB") + 1])
    assert unpack(got[o0:o1]) == "40=5X30=268435456I10D20="            # reversed ('-'), I <-> D, the long op still one op


@gpu
@pytest.mark.parametrize("policy", [0, 1])
def test_trim_pairs_with_long_ops(engine, oracle, policy):
    from test_gpu_trim import _compare
    lines = [line("q1", "300000000=5X1000=", t_st=1000, q_st=0),
             line("q1", "500=3X2000=2I40=", t_st=400_000_000, q_st=300_000_400),
             line("q2", "30=268435456I20=1X900=", t_st=10, q_st=0, strand="-"),
             line("q2", "700=1X268435460D60=", t_st=500, q_st=268_436_000)]
    r = recs_from_lines(lines)
    left, right = np.array([0, 2], np.uint32), np.array([1, 3], np.uint32)
    for scores in ((1, 1, 1), (2, 3, 5)):
        rows, out = engine.overlap_split(r.ops, r.op_off, r.t_st, r.t_en, r.q_st, r.q_en, r.strand, left, right, scores, policy)
        ob = oracle.Batch(r.ops, r.op_off, r.t_st, r.t_en, r.q_st, r.q_en, r.strand)
        orows, oout = oracle.overlap_split(ob, left, right, scores, policy)
        _compare(rows, out, orows, oout, f"long ops policy={policy} {scores}")
        assert (rows["status"] == 0).all()


def _rb(*args):
    r = subprocess.run([RB, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout


@gpu
def test_rb_front_end_with_long_ops(oracle, tmp_path):
    """text in, text out: the device's CIGAR parser reports the long lengths, the file takes the line-by-line route, and every
    subcommand on the path prints what the oracle prints"""
    paf = tmp_path / "long.paf"
    paf.write_text("\n".join(CASES[:-1]) + "\n")                       # (without qJ: the reference panics on it, below)
    bed = tmp_path / "w.bed"
    bed.write_text("".join(f"chrBig\t{a}\t{e}\n" for a, e in W))
    trim = tmp_path / "trim.paf"
    trim.write_text("\n".join([line("q1", "300000000=5X1000=", t_st=1000, q_st=0), line("q1", "500=3X2000=2I40=", t_st=400_000_000, q_st=300_000_400)]) + "\n")
    for args in (["liftover", "--bed", bed, paf], ["--bsearch", "legacy", "liftover", "--bed", bed, paf], ["liftover", "--largest", "--bed", bed, paf],
                 ["break-paf", "--max-size", "100", paf], ["break-paf", "--max-size", "270000000", paf], ["invert", paf],
                 ["stats", "--paf", paf], ["trim-paf", trim]):
        rc, out = _rb(*args)
        orc, oout = oracle.cli(*args)
        assert (rc, orc) == (0, 0), args
        assert out == oout, args
    # two worker processes (both on the one GPU here), their outputs put together by the parent
    r = subprocess.run([RB, "--gpus", "2", "liftover", "--bed", str(bed), str(paf)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "RB_GPUS_SAME_DEVICE": "1"})
    assert r.returncode == 0 and r.stdout == oracle.cli("liftover", "--bed", bed, paf)[1]
    panics = tmp_path / "panics.paf"
    panics.write_text("\n".join(CASES) + "\n")
    assert _rb("liftover", "--bed", bed, panics)[0] == 101 and oracle.cli("liftover", "--bed", bed, panics)[0] == 101
    big = tmp_path / "toolong.paf"
    big.write_text("q\t10\t0\t5\t+\tt\t10\t0\t5\t5\t5\t60\tcg:Z:4294967296=\n")    # past u32: "Unable to parse cigar string."
    assert _rb("stats", "--paf", big)[0] == 101 and oracle.cli("stats", "--paf", big)[0] != 0


@gpu
def test_format_kernel_prints_continuation_words(engine):
    """rb_dev_format_cigars (impl Display for CigarString): an op and its continuation word print as ONE op with the whole length,
    up to ten digits; items that begin or end at such an op, a pair split across two steps of 256 words and across two lanes"""
    rng = np.random.default_rng(28)
    cigs = ["4294967295=", "5X268435456=3I", "1000000000D", "7=999999999I268435455X268435457N", "3S4026531840H2P"]
    filler = "".join(f"{int(v)}{'=XID'[int(k)]}" for v, k in zip(rng.integers(1, 3000, 700), rng.integers(0, 4, 700)))
    for cut in (251, 252, 253, 254, 255, 256, 257, 511, 512):              # the long op's two words at every phase of a step / lane
        head = "".join(f"{int(v)}{'=XID'[int(k)]}" for v, k in zip(rng.integers(1, 99, cut), rng.integers(0, 4, cut)))
        cigs.append(head + "3000000000=" + filler)
    words = [pack(c) for c in cigs]
    ops = np.concatenate(words)
    count = np.array([len(w) for w in words], np.uint32)
    first = np.zeros(len(words), np.uint64)
    first[1:] = np.cumsum(count)[:-1]
    toff, text = engine.format_cigars(ops, first, count)
    for i, c in enumerate(cigs):
        assert bytes(text[int(toff[i]):int(toff[i + 1])]).decode() == c, (i, c[:40])


@gpu
def test_text_route_prints_runs_that_merge_past_2_28(oracle, tmp_path):
    """no length in the FILE reaches 2^28, so the device parses it and the text route stays on; two neighbours of one type merge into a
    run that does (paf.rs:602-620), the generic kernel writes it as two words and rb_k_format_cigars prints them as one op"""
    paf = tmp_path / "merge.paf"
    paf.write_text("\n".join([line("qE", "200000000=100000000=5X10="), line("qG", "100=2X50="), line("qK", "7=3I268435455=268435455=9X4=", strand="-")]) + "\n")
    bed = tmp_path / "w.bed"
    bed.write_text("chrBig\t2\t268435470\nchrBig\t150\t600000000\nchrBig\t0\t700000000\nchrBig\t120\t200\n")
    for args in (["liftover", "--bed", bed, paf], ["break-paf", "--max-size", "2", paf]):
        rc, out = _rb(*args)
        orc, oout = oracle.cli(*args)
        assert (rc, orc) == (0, 0), args
        assert out == oout, args
    assert b"cg:Z:268435370=" in _rb("liftover", "--bed", bed, paf)[1]


# Records of nearly 2^32 and of more than 2^32 units.  The reference sums a record's lengths in u32 (infer_n_bases, paf.rs:632-647): a
# record of 2^32 units and more is PANIC_OVERFLOW at the scan and never reaches a walk -- which is what keeps the 32-bit checkpoints of
# the general kernel (units / reference / query / match bases in front of every 256th op) from wrapping; rb_k_generic_checkpoints sums
# in 64 bits and leaves such a record without checkpoints all the same.  A record just under 2^32 units has checkpoint
# values up to 4.29e9.  Neither fits the per-base oracle (73 GB): the expectation is worked out here -- a record of = and X only maps
# reference to query base by base, a clip is the run of ops under the window with its first and last op cut.
def _big_record(head_len):
    head = f"{head_len}=1X" * 3
    tail = "5=1X" * 400
    return line("qBig", head + tail, t_st=1000, q_st=7, t_len=9_000_000_000), 3 * (head_len + 1)


def _clip_of_match_only(cg, t0, ws, we):
    out, pos = [], t0
    for n, o in ((int(a), b) for a, b in re.findall(r"(\d+)([=X])", cg)):
        lo, hi = max(pos, ws), min(pos + n, we)
        if lo < hi:
            out.append(f"{hi - lo}{o}")
        pos += n
    return "".join(out)


def _big_batch(head_len):
    ln, big = _big_record(head_len)
    r = recs_from_lines([ln])
    b = dict(ops=r.ops, op_off=r.op_off, t_st=r.t_st, t_en=r.t_en, q_st=r.q_st, q_en=r.q_en, strand=r.strand, contig=r.contig)
    t0 = 1000
    wins = [(t0 + big + 100, t0 + big + 163), (t0 + big + 1200, t0 + big + 1207), (t0 + head_len - 5, t0 + head_len + 9),
            (t0 + big - 3, t0 + big + 14), (t0 + big + 2395, t0 + big + 2400)]
    w = (np.zeros(len(wins), np.uint32), np.array([a for a, _ in wins], np.uint64), np.array([e for _, e in wins], np.uint64))
    return ln, b, wins, w


def test_oracle_scan_of_records_around_2_32_units(oracle):
    for head_len, status in ((1431654000, 0), (2147483648, oracle.PANIC_OVERFLOW)):
        ln, b, _, _ = _big_batch(head_len)
        ob = oracle.Batch(b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])
        assert int(oracle.normalize(ob)["status"][0]) == status and int(oracle.reduce(ob)["status"][0]) == status


@gpu
@pytest.mark.parametrize("policy", [0, 1])
def test_records_around_2_32_units_through_the_checkpointed_walk(engine, policy):
    # just under 2^32 units: checkpoint values up to 4.29e9
    ln, b, wins, w = _big_batch(1431654000)
    rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy)
    assert int(norm["status"][0]) == 0 and len(rows) == len(wins) and (rows["status"] == 0).all()
    cg, t0 = ln.split("cg:Z:")[1], 1000
    for h in rows:
        ws, we = wins[int(h["win"])]
        text = unpack(ops[int(h["out_off"]):int(h["out_off"]) + int(h["out_n"])])
        assert text == _clip_of_match_only(cg, t0, ws, we), (ws, we, text)
        assert (int(h["t_st"]), int(h["t_en"])) == (ws, we)
        assert (int(h["q_st"]), int(h["q_en"])) == (7 + ws - t0, 7 + we - t0)
    # 2^32 units and more: the scan's verdict, no rows
    ln, b, wins, w = _big_batch(2147483648)
    rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy)
    assert int(norm["status"][0]) == 22 and len(rows) == 0
