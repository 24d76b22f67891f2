"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/known_answers.json)."""
import json
import os

import numpy as np
import pytest

from rbtest_util import pack, recs_from_lines, unpack


@pytest.fixture(scope="module")
def ka(golden):
    return json.load(open(os.path.join(golden, "known_answers.json")))


def _paf(tmp_path, lines, name="in.paf"):
    p = tmp_path / name
    p.write_text("".join(ln + "\n" for ln in lines))
    return str(p)


def _parse_out(out):
    rows = []
    for ln in out.decode().splitlines():
        t = ln.split("\t")
        rows.append(dict(q_name=t[0], t_name=t[5], q_st=int(t[2]), q_en=int(t[3]), strand=t[4], t_st=int(t[7]), t_en=int(t[8]),
                         nmatch=int(t[9]), aln_len=int(t[10]), id=t[12][5:], cigar=t[13][5:]))
    return rows


@pytest.mark.parametrize("policy", ["modern", "legacy"])
def test_ka1_liftover_cli(oracle, ka, tmp_path, policy):
    k = ka["KA1_liftover"]
    paf = _paf(tmp_path, k["records"])
    bed = tmp_path / "w.bed"
    bed.write_text("".join(f"{c}\t{s}\t{e}\tw{i}\n" for i, (c, s, e) in enumerate(k["windows"])))
    rc, out = oracle.cli("--bsearch", policy, "liftover", "--bed", bed, paf)
    assert rc == 0
    rows = _parse_out(out)
    nw = len(k["windows"])
    assert len(rows) == 2 * nw  # record-major: forward record x windows, then reverse record x windows
    for w in range(nw):
        for s in range(2):
            got = rows[s * nw + w]
            assert (got["q_st"], got["q_en"]) == (k["q_st"][2 * w + s], k["q_en"][2 * w + s]), (w, s, got)
    # quirk 9.3.3: the window strictly containing the record returns it with its OWN (empty) id
    assert rows[5]["id"] == "" and rows[2]["id"] == "w2"


def test_ka1_liftover_arrays(oracle, ka):
    k = ka["KA1_liftover"]
    r = recs_from_lines(k["records"])
    b = oracle.Batch(*r.arrays(), r.contig)
    wc = np.zeros(len(k["windows"]), np.uint32)
    ws = np.array([w[1] for w in k["windows"]], np.uint64)
    we = np.array([w[2] for w in k["windows"]], np.uint64)
    for pol in (oracle.MODERN, oracle.LEGACY):
        rows, ops = oracle.liftover(b, wc, ws, we, pol)
        assert len(rows) == 12 and (rows["status"] == 0).all()
        for i, row in enumerate(rows):
            rec, win = int(row["rec"]), int(row["win"])
            assert (int(row["q_st"]), int(row["q_en"])) == (k["q_st"][2 * win + rec], k["q_en"][2 * win + rec])
        assert rows["flags"][5] & 1 and not rows["flags"][2] & 1


@pytest.mark.parametrize("policy", ["modern", "legacy"])
def test_ka2_break(oracle, ka, tmp_path, policy):
    k = ka["KA2_break"]
    rc, out = oracle.cli("--bsearch", policy, "break-paf", "--max-size", k["break_length"], _paf(tmp_path, [k["record"]]))
    assert rc == 0
    rows = _parse_out(out)
    assert len(rows) == k["expect_n_pieces"]
    for r in rows:
        assert r["t_en"] - r["t_st"] == k["expect_piece_t_span"]


def test_ka3_trim_pair(oracle, ka):
    k = ka["KA3_trim_pair"]
    r = recs_from_lines([k["left"], k["right"]])
    b = oracle.Batch(*r.arrays(), r.contig)
    rows, ops = oracle.overlap_split(b, [0], [1], tuple(k["scores"]))
    assert rows["status"][0] == 0
    got = [unpack(ops[int(rows["out_off"][0][s]):int(rows["out_off"][0][s]) + int(rows["out_n"][0][s])]) for s in (0, 1)]
    assert got == [k["left_cigar"], k["right_cigar"]]
    assert int(rows["split_idx"][0]) == 2 and int(rows["split_score"][0]) == 5


@pytest.mark.parametrize("policy", ["modern", "legacy"])
def test_ka4_trim_paf(oracle, ka, tmp_path, policy):
    k = ka["KA4_trim_paf"]
    rc, out = oracle.cli("--bsearch", policy, "trim-paf", _paf(tmp_path, k["records"]))
    assert rc == 0
    assert [r["cigar"] for r in _parse_out(out)] == k["cigars"]


def test_ka5_stats(oracle, ka):
    k = ka["KA5_stats"]
    ops = pack(k["cigar"])
    b = oracle.Batch(ops, [0, len(ops)], [0], [20], [0], [20], [ord("+")])
    row = oracle.reduce(b)[0]
    assert abs(float(row["id_by_all"]) - k["id_by_all"]) < 1e-10
    assert row["status"] == 0


def test_ka7_fixture_loads(oracle, ka, golden):
    rc, out = oracle.cli("stats", "--paf", os.path.join(golden, ka["KA7_fixture"]["file"]))
    assert rc == 0
    assert out.count(b"\n") == ka["KA7_fixture"]["n_records"] + 1  # 249 records pass check_integrity + header


def test_ka8_ka10_cigar_text(oracle, ka, tmp_path):
    rc, out = oracle.cli("invert", _paf(tmp_path, [ka["KA8_fake_rec"]["line"]]))
    assert rc == 0  # 4M1I1D3= on '-': I<->D then reversed
    assert _parse_out(out)[0]["cigar"] == "3=1I1D4M"
    for cg in ka["KA10_cigar_parse"]["cigars"] + [ka["KA8_fake_rec"]["cigar"]]:
        assert unpack(pack(cg)) == cg
        r, q = 0, 0
        line = f"Q 1000000 0 {sum(int(v) >> 4 for v in pack(cg) if (int(v) & 15) in (0, 1, 4, 7, 8))} + T 1000000 0 " \
               f"{sum(int(v) >> 4 for v in pack(cg) if (int(v) & 15) in (0, 2, 3, 7, 8))} 0 0 60 cg:Z:{cg}"
        rc, out = oracle.cli("trim-paf", _paf(tmp_path, [line], "one.paf"))
        assert rc == 0 and _parse_out(out)[0]["cigar"] == cg


def test_ka9_predicates(oracle):
    L = oracle.lib()
    X, M, EQ, I, D = 8, 0, 7, 1, 2
    assert L.rbo_consumes_query(X) and L.rbo_is_match(M) and L.rbo_is_match(X) and L.rbo_is_match(EQ)
    assert not L.rbo_is_match(I) and not L.rbo_consumes_reference(I) and not L.rbo_consumes_query(D)


def test_ka11_bed(oracle, ka, golden, tmp_path):
    # 10 regions from the 11-line fixture: `liftover` with every record must see exactly 10 windows;
    # checked through the default id rule as well
    paf = os.path.join(golden, "asm_small.paf")
    rc, out = oracle.cli("liftover", "--bed", os.path.join(golden, ka["KA11_bed"]["file"]), paf)
    assert rc == 0 and out.count(b"\n") == 12
    bed = tmp_path / "d.bed"
    bed.write_text("T\t13\t18\n")
    rc, out = oracle.cli("liftover", "--bed", bed, _paf(tmp_path, ["Q 10 2 10 + T 40 12 20 3 9 60 cg:Z:4M1I1=1D2="]))
    assert _parse_out(out)[0]["id"] == "T:14-18"  # bed.rs:150-153 default id = name:st+1-en


def test_ka12_gz_and_bgz_same_as_plain(oracle, golden):
    """src/myio.rs:38-46: reader() on the plain file, its .bgz and its .gz yields the same lines (all three legs of the doc test)"""
    a = oracle.cli("stats", "--paf", os.path.join(golden, "asm_small.paf"))
    b = oracle.cli("stats", "--paf", os.path.join(golden, "asm_small.paf.gz"))
    c = oracle.cli("stats", "--paf", os.path.join(golden, "asm_small.paf.bgz"))
    assert a == b and a == c and a[0] == 0


def test_f32_display(oracle):
    # Rust `{}` on f32: shortest round trip, positional, "NaN"; checked here against numpy's shortest repr
    rng = np.random.default_rng(7)
    assert oracle.f32_display(100.0) == "100" and oracle.f32_display(float("nan")) == "NaN"
    assert oracle.f32_display(99.89702) == "99.89702" and oracle.f32_display(0.0) == "0"
    e = rng.integers(0, 2 ** 31 - 1, 3000, dtype=np.int64)
    s = e + rng.integers(0, 2 ** 20, 3000)
    vals = (np.float32(100.0) * e.astype(np.float32)) / s.astype(np.float32)
    for v in vals:
        want = np.format_float_positional(np.float32(v), unique=True, trim="-")
        assert oracle.f32_display(float(v)) == want, (v, want)


def test_remove_trailing_indels_quirks(oracle):
    """leading I shifts the query (end on '-'), leading D makes the reference panic."""
    rows = oracle.normalize(oracle.Batch(pack("2I5=3D"), [0, 3], [100], [108], [10], [17], [ord("+")]))
    r = rows[0]
    assert r["status"] == 0 and (r["t_st"], r["t_en"], r["q_st"], r["q_en"]) == (100, 105, 12, 17)
    assert (r["first_op"], r["n_ops"], r["lead_ops"], r["trail_ops"]) == (1, 1, 1, 1)
    r = oracle.normalize(oracle.Batch(pack("2I5=3D"), [0, 3], [100], [108], [10], [17], [ord("-")]))[0]
    assert r["status"] == 0 and (r["q_st"], r["q_en"]) == (10, 15)
    r = oracle.normalize(oracle.Batch(pack("2D5="), [0, 2], [100], [107], [10], [15], [ord("+")]))[0]
    assert r["status"] == oracle.PANIC_INTEGRITY_Q
    r = oracle.normalize(oracle.Batch(pack("2D3I"), [0, 2], [100], [102], [10], [13], [ord("+")]))[0]
    assert r["status"] == oracle.PANIC_ALL_INDEL
    r = oracle.normalize(oracle.Batch(np.zeros(0, np.uint32), [0, 0], [1], [1], [1], [1], [ord("+")]))[0]
    assert r["status"] == oracle.PANIC_EMPTY_CIGAR


def test_ka6_md_parse(oracle, ka):
    import ctypes as C
    m4 = (C.c_uint32 * 4)()
    oracle.lib().rbo_parse_md_for_stats(ka["KA6_md"]["md"].encode(), m4)
    assert list(m4) == ka["KA6_md"]["expect"]


def test_bam_stats_oracle(oracle, golden):
    """`rb stats <bam>` (bamstats.rs:156-222): the 70 alignments of asm_small.bam are also in asm_small.paf, so every
    BAM stats line must appear verbatim among the PAF stats lines -- pins the q-coordinate logic (hard clips,
    reverse-strand flip, end_pos / read_pos) independently of this restatement of rust-htslib."""
    import hashlib
    import json
    rc, b = oracle.cli("stats", f"{golden}/asm_small.bam")
    rc2, p = oracle.cli("stats", "--paf", f"{golden}/asm_small.paf")
    assert (rc, rc2) == (0, 0)
    blines, plines = b.splitlines()[1:], set(p.splitlines()[1:])
    assert len(blines) == 70 and all(x in plines for x in blines)
    dig = json.load(open(f"{golden}/digests.json"))
    for key, f in (("stats_bam_asm_small", "asm_small.bam"), ("stats_bam_test", "test.bam"), ("stats_bam_stats", "stats.bam")):
        rc, out = oracle.cli("stats", f"{golden}/{f}")
        assert rc == 0 and hashlib.md5(out).hexdigest() == dig[key]["md5"] and out.count(b"\n") == dig[key]["lines"]


def test_ka13_orient_scaffold_filter(oracle, tmp_path):
    """Hand-derived from paf.rs:91-207.  (T,A): orient = +10 - 30 < 0 -> "A-", query coordinates flipped against q_len 100,
    strands swapped; order = (10*10/2 + 30*230/2) / 40 = 87 for both A records, (T,B): 502.  Scaffold: sort by (order, q_st)
    puts the flipped 50..80 record first; A- spans 50..100 -> 0..30 and 40..50, spacer 7, B+ 0..5 -> 57..62, length 62.
    filter --paired-len 5: (T,A) sums 40 > 5 kept, (T,B) sums 5, 5 < 5 false -> dropped."""
    p = tmp_path / "o.paf"
    p.write_text("A\t100\t0\t10\t+\tT\t1000\t0\t10\t10\t10\t60\tcg:Z:10=\n"
                 "A\t100\t20\t50\t-\tT\t1000\t100\t130\t30\t30\t60\tcg:Z:30=\n"
                 "B\t100\t0\t5\t+\tT\t1000\t500\t505\t5\t5\t60\tcg:Z:5=\n")

    def cols(out):
        return [ln.split("\t")[:9] for ln in out.decode().splitlines()]
    rc, out = oracle.cli("orient", p)
    assert rc == 0 and cols(out) == [
        ["A-", "100", "90", "100", "-", "T", "1000", "0", "10"],
        ["A-", "100", "50", "80", "+", "T", "1000", "100", "130"],
        ["B+", "100", "0", "5", "+", "T", "1000", "500", "505"]]
    rc, out = oracle.cli("orient", "--scaffold", "--insert", "7", p)
    assert rc == 0 and cols(out) == [
        ["A-::B+", "62", "0", "30", "+", "T", "1000", "100", "130"],
        ["A-::B+", "62", "40", "50", "-", "T", "1000", "0", "10"],
        ["A-::B+", "62", "57", "62", "+", "T", "1000", "500", "505"]]
    rc, out = oracle.cli("filter", "--paired-len", "5", p)
    assert rc == 0 and [c[0] for c in cols(out)] == ["A", "A"]
    rc, out = oracle.cli("filter", "--aln", "9", "--query", "99", p)
    assert rc == 0 and [c[2] for c in cols(out)] == ["0", "20"]


def test_ka14_ka15_region_helpers(oracle, ka):
    """bed.rs:51-64 has_overlap and bed.rs:199-214 split_region: the reference's doctest vectors on the oracle's restatements, and
    the first through the path that uses it (paf_overlaps_rgn, paf.rs:622-627: which windows a record is clipped to)"""
    import ctypes as C
    L = oracle.lib()
    L.rbo_get_overlap.restype = C.c_uint64
    k = ka["KA14_has_overlap"]
    n1, s1, e1 = k["rgn1"]
    for (n2, s2, e2), want in k["cases"]:
        assert bool(L.rbo_has_overlap(n1.encode(), C.c_uint64(s1), C.c_uint64(e1), n2.encode(), C.c_uint64(s2), C.c_uint64(e2))) == want
        ov = int(L.rbo_get_overlap(n1.encode(), C.c_uint64(s1), C.c_uint64(e1), n2.encode(), C.c_uint64(s2), C.c_uint64(e2)))
        assert (ov > 0) == want and ov == max(0, min(e1, e2) - max(s1, s2))
    assert int(L.rbo_get_overlap(b"chr1", C.c_uint64(0), C.c_uint64(9), b"chr2", C.c_uint64(0), C.c_uint64(9))) == 0
    # ... and as the window filter of liftover: a record on chr1:[10,15) against the six regions of the doctest
    ops = pack("5=")
    b = oracle.Batch(ops, np.array([0, 1], np.uint64), np.array([s1], np.uint64), np.array([e1], np.uint64), np.array([0], np.uint64),
                     np.array([5], np.uint64), np.array([ord("+")], np.uint8), np.zeros(1, np.uint32))
    w_st = np.array([c[0][1] for c in k["cases"]], np.uint64)
    w_en = np.array([c[0][2] for c in k["cases"]], np.uint64)
    rows, _ = oracle.liftover(b, np.zeros(len(w_st), np.uint32), w_st, w_en)
    assert sorted(rows["win"].tolist()) == [i for i, c in enumerate(k["cases"]) if c[1]]
    s = ka["KA15_split_region"]
    st, en = s["region"][1], s["region"][2]

    def pieces(window):
        out, a, e = [], C.c_uint64(), C.c_uint64()
        while L.rbo_split_region(C.c_uint64(st), C.c_uint64(en), C.c_uint64(window), C.c_uint64(len(out)), C.byref(a), C.byref(e)):
            out.append([a.value, e.value])
        return out
    p10, p100 = pieces(10), pieces(100)
    assert len(p10) == s["window10"]["n"] and p10[0] == s["window10"]["first"] and p10[-1] == s["window10"]["last"]
    assert len(p100) == s["window100"]["n"] and p100[0] == s["window100"]["first"]


def test_ka16_ka17_target_region_and_parse_region(oracle, ka, tmp_path):
    """paf.rs:468-478 get_target_as_region (the target columns of a parsed line) and bed.rs:88-96 parse_region"""
    import ctypes as C
    k = ka["KA16_target_as_region"]

    class Rec(C.Structure):  # rbo_rec (oracle/rb_oracle.h); PafRecord::new alone -- the doctest's line would not pass check_integrity
        _fields_ = [("q_name", C.c_char_p), ("q_len", C.c_uint64), ("q_st", C.c_uint64), ("q_en", C.c_uint64), ("strand", C.c_char),
                    ("t_name", C.c_char_p), ("t_len", C.c_uint64), ("t_st", C.c_uint64), ("t_en", C.c_uint64), ("nmatch", C.c_uint64),
                    ("aln_len", C.c_uint64), ("mapq", C.c_uint64), ("cigar", C.c_void_p), ("n_cigar", C.c_size_t), ("id", C.c_char_p),
                    ("tpos_aln", C.c_void_p), ("qpos_aln", C.c_void_p), ("long_cigar", C.c_void_p), ("n_aln", C.c_size_t), ("contained", C.c_int)]
    rec = Rec()
    oracle.lib().rbo_rec_init(C.byref(rec))
    assert oracle.lib().rbo_rec_from_line(k["line"].encode(), C.byref(rec)) == 0
    assert [rec.t_name.decode(), rec.t_st, rec.t_en] == k["region"] and rec.n_cigar == 3
    oracle.lib().rbo_rec_free(C.byref(rec))

    class Rg(C.Structure):
        _fields_ = [("name", C.c_char_p), ("st", C.c_uint64), ("en", C.c_uint64), ("id", C.c_char_p)]
    for text, (name, st, en) in ka["KA17_parse_region"]["cases"]:
        r = Rg()
        assert oracle.lib().rbo_parse_region(text.encode(), C.byref(r)) == 0
        assert (r.name.decode(), r.st, r.en) == (name, st, en)


def test_ka18_ka19_doctest_inputs_without_an_assert_of_their_own(oracle, ka):
    """The two doctest inputs of src/paf.rs that no KA held so far (tests/golden/README.md, "doctest blocks"): PafRecord::new on a line of
    the twelve mandatory columns without a tag must succeed (paf.rs:374: `.unwrap()`), and update_cigar_opt_len keeps the op and replaces
    the length (paf.rs:981-982)."""
    import ctypes as C
    k = ka["KA18_new_without_tags"]

    class Rec(C.Structure):  # rbo_rec (oracle/rb_oracle.h)
        _fields_ = [("q_name", C.c_char_p), ("q_len", C.c_uint64), ("q_st", C.c_uint64), ("q_en", C.c_uint64), ("strand", C.c_char),
                    ("t_name", C.c_char_p), ("t_len", C.c_uint64), ("t_st", C.c_uint64), ("t_en", C.c_uint64), ("nmatch", C.c_uint64),
                    ("aln_len", C.c_uint64), ("mapq", C.c_uint64), ("cigar", C.c_void_p), ("n_cigar", C.c_size_t), ("id", C.c_char_p),
                    ("tpos_aln", C.c_void_p), ("qpos_aln", C.c_void_p), ("long_cigar", C.c_void_p), ("n_aln", C.c_size_t), ("contained", C.c_int)]
    L = oracle.lib()
    rec = Rec()
    L.rbo_rec_init(C.byref(rec))
    assert L.rbo_rec_from_line(k["line"].encode(), C.byref(rec)) == 0
    got = {"q_name": rec.q_name.decode(), "q_len": rec.q_len, "q_st": rec.q_st, "q_en": rec.q_en, "strand": rec.strand.decode(),
           "t_name": rec.t_name.decode(), "t_len": rec.t_len, "t_st": rec.t_st, "t_en": rec.t_en, "nmatch": rec.nmatch, "aln_len": rec.aln_len,
           "mapq": rec.mapq}
    assert got == k["fields"] and rec.n_cigar == k["n_cigar"]
    L.rbo_rec_free(C.byref(rec))
    L.rbo_update_cigar_opt_len.restype = C.c_uint64
    L.rbo_update_cigar_opt_len.argtypes = [C.c_uint64, C.c_uint32]
    for before, new_len, after in ka["KA19_update_cigar_opt_len"]["cases"]:
        assert int(L.rbo_update_cigar_opt_len(int(pack(before)[0]), new_len)) == int(pack(after)[0])
