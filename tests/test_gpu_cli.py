"""File-level drop-in check: the C++ `rb` front end (rustybam_amd/rb, host mirror over the C ABI) must print
byte-for-byte what the oracle CLI prints for the hot-path subcommands on the reference fixture."""
import hashlib
import json
import os
import subprocess

import pytest

from golden.make_digests import tile_bed

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RB = os.path.join(ROOT, "rustybam_amd", "rb")


def rb(*args, env=None):
    assert os.path.exists(RB), "rustybam_amd/rb missing: run __graft_entry__.build()"
    r = subprocess.run([RB, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=None if env is None else {**os.environ, **env})
    return r.returncode, r.stdout


@pytest.fixture(scope="module")
def dig(golden):
    return json.load(open(os.path.join(golden, "digests.json")))


@pytest.mark.parametrize("key,args", [
    ("stats_paf", ["stats", "--paf", "{paf}"]),
    ("liftover_asm_small_bed", ["liftover", "--bed", "{bed}", "{paf}"]),
    ("liftover_asm_small_bed", ["--bsearch", "legacy", "liftover", "--bed", "{bed}", "{paf}"]),
    ("break_paf_100_modern", ["break-paf", "--max-size", "100", "{paf}"]),
    ("break_paf_100_legacy", ["--bsearch", "legacy", "break-paf", "--max-size", "100", "{paf}"]),
    ("trim_paf_modern", ["trim-paf", "{paf}"]),
    ("trim_paf_legacy", ["--bsearch", "legacy", "trim-paf", "{paf}"]),
    ("invert", ["invert", "{paf}"]),
])
def test_fixture_digests(dig, golden, key, args):
    a = [x.format(paf=f"{golden}/asm_small.paf", bed=f"{golden}/asm_small.bed") for x in args]
    rc, out = rb(*a)
    assert rc == 0
    assert hashlib.md5(out).hexdigest() == dig[key]["md5"], key


def test_tiled_windows_and_gz_input(dig, golden, tmp_path):
    bed = str(tmp_path / "tile.bed")
    tile_bed(bed)
    for ext in (".gz", ".bgz"):  # src/myio.rs:38-46: both compressed forms read like the plain file
        rc, out = rb("liftover", "--bed", bed, f"{golden}/asm_small.paf{ext}")
        assert rc == 0 and hashlib.md5(out).hexdigest() == dig["liftover_tile_100kb"]["md5"] and out.count(b"\n") == 1657, ext


@pytest.mark.parametrize("args", [
    ["liftover", "--qbed", "--bed", "{trimbed}", "{paf}"],
    ["liftover", "--largest", "--bed", "{bed}", "{paf}"],
    ["trim-paf", "-r", "{paf}"],
    ["trim-paf", "--match-score", "2", "--diff-score", "3", "--indel-score", "5", "{paf}"],
    ["break-paf", "--max-size", "0", "{paf}"],
    ["stats", "--qbed", "--paf", "{paf}"],
    ["orient", "{paf}"],
    ["orient", "--scaffold", "{paf}"],
    ["orient", "-s", "-i", "500", "{paf}"],
    ["filter", "--paired-len", "100000", "{paf}"],
    ["filter", "-a", "20000", "-q", "60000000", "-p", "50000", "{paf}"],
])
def test_matches_oracle_cli(oracle, golden, args):
    a = [x.format(paf=f"{golden}/asm_small.paf", bed=f"{golden}/asm_small.bed", trimbed=f"{golden}/trim_asm_small.bed") for x in args]
    rc, out = rb(*a)
    orc, oout = oracle.cli(*a)
    assert (rc, orc) == (0, 0)
    assert out == oout


def test_panics_like_the_reference(tmp_path):
    bad = tmp_path / "bad.paf"
    bad.write_text("Q 10 0 5 + T 10 0 6 0 0 60 cg:Z:5=\n")  # target span 6 != 5 reference bases
    rc, out = rb("stats", "--paf", bad)
    assert rc == 101  # check_integrity().unwrap() at paf.rs:70
    short = tmp_path / "short.paf"
    short.write_text("Q 10 0 5 + T 10\n")
    assert rb("invert", short)[0] == 101  # assert!(t.len() >= 12) at paf.rs:381
    skip = tmp_path / "skip.paf"
    skip.write_text("Q x 0 5 + T 10 0 5 0 0 60 cg:Z:5=\nQ 10 0 5 + T 10 0 5 0 0 60 cg:Z:5=\n")
    rc, out = rb("invert", skip)
    assert rc == 0 and out.count(b"\n") == 1  # unparsable numeric column: the line is skipped (paf.rs:73)


# ---- BAM input of `rb stats`: BGZF/BAM decode on the host, CIGAR counters on the device ----
def _write_bam(path, refs, recs):
    """minimal BAM writer (one gzip member per BGZF-less stream is fine for zlib readers)"""
    import gzip
    import struct
    out = bytearray(b"BAM\x01")
    text = b"@HD\tVN:1.6\n"
    out += struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for nm, ln in refs:
        out += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    for r in recs:
        name = r["name"].encode() + b"\0"
        cig = b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(c)) for l, c in r["cigar"])
        l_seq = r["l_seq"]
        body = struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), 60, 0, len(r["cigar"]), r["flag"], l_seq, -1, -1, 0)
        body += name + cig + b"\x11" * ((l_seq + 1) // 2) + b"\xff" * l_seq + r.get("aux", b"")
        out += struct.pack("<i", len(body)) + body
    with gzip.open(path, "wb") as f:
        f.write(bytes(out))


@pytest.mark.parametrize("bam", ["asm_small.bam", "test.bam", "stats.bam"])
@pytest.mark.parametrize("qbed", [False, True])
def test_stats_bam_fixtures(oracle, golden, bam, qbed):
    a = ["stats"] + (["--qbed"] if qbed else []) + [f"{golden}/{bam}"]
    rc, out = rb(*a)
    orc, oout = oracle.cli(*a)
    assert (rc, orc) == (0, 0)
    assert out == oout and out.count(b"\n") > 5


def test_stats_bam_equals_stats_paf_of_the_same_alignments(golden):
    """asm_small.bam and asm_small.paf hold the same alignments: every BAM stats line appears among the PAF stats lines
    (an independent check of the hard-clip / reverse-strand query coordinates of bamstats.rs:190-207)"""
    rc1, b = rb("stats", f"{golden}/asm_small.bam")
    rc2, p = rb("stats", "--paf", f"{golden}/asm_small.paf")
    assert (rc1, rc2) == (0, 0)
    plines = set(p.splitlines()[1:])
    blines = b.splitlines()[1:]
    assert len(blines) == 70 and all(x in plines for x in blines)


def test_stats_bam_md_tag_cg_tag_and_panics(oracle, tmp_path):
    import struct
    refs = [("chrA", 100000)]
    md = b"MDZ" + b"10A3T0T10^ACGT5" + b"\0"  # KA6 string + 5 more matches; 23+5 matches, 3 mismatches
    recs = [
        dict(name="md_refines_M", ref=0, pos=100, flag=0, l_seq=31, cigar=[(14, "M"), (4, "D"), (17, "M")], aux=md),   # 31 M bases = 28 + 3
        dict(name="no_md", ref=0, pos=200, flag=16, l_seq=40, cigar=[(5, "S"), (20, "M"), (3, "I"), (12, "M")], aux=b"NMC\x03"),
        dict(name="hard_soft", ref=0, pos=300, flag=16, l_seq=25, cigar=[(7, "H"), (5, "S"), (20, "="), (9, "H")]),
        dict(name="unmapped", ref=-1, pos=-1, flag=4, l_seq=10, cigar=[]),
    ]
    real = [(30, "="), (2, "X"), (8, "=")]
    cg = b"CGBI" + struct.pack("<i", len(real)) + b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(c)) for l, c in real)
    recs.append(dict(name="long_cigar_in_CG", ref=0, pos=500, flag=0, l_seq=40, cigar=[(40, "S"), (40, "N")], aux=cg))
    bam = tmp_path / "synth.bam"
    _write_bam(bam, refs, recs)
    rc, out = rb("stats", bam)
    orc, oout = oracle.cli("stats", bam)
    assert (rc, orc) == (0, 0) and out == oout
    lines = [ln.split(b"\t") for ln in out.splitlines()[1:]]
    assert len(lines) == 4  # the unmapped record is skipped (main.rs:73)
    assert lines[0][5] == b"md_refines_M" and (lines[0][12], lines[0][13]) == (b"28", b"3")  # equal / diff from the MD tag
    assert lines[2][5:9] == [b"hard_soft", b"9", b"29", b"41"]  # reverse strand: q_len 41, q_st = 41 - 32, q_en = 41 - 12
    assert lines[3][5] == b"long_cigar_in_CG" and lines[3][12] == b"38"
    # KA6 (bamstats.rs:41-46)
    import ctypes as C
    m4 = (C.c_uint32 * 4)()
    oracle.lib().rbo_parse_md_for_stats(b"10A3T0T10^ACGT", m4)
    assert list(m4) == [23, 3, 1, 4]
    # alignment whose reference span ends in a deletion: read_pos(...).unwrap().unwrap() panics
    bad = tmp_path / "bad.bam"
    _write_bam(bad, refs, [dict(name="ends_in_D", ref=0, pos=10, flag=0, l_seq=10, cigar=[(10, "="), (5, "D")])])
    assert rb("stats", bad)[0] == 101 and oracle.cli("stats", bad)[0] == 101
    bad2 = tmp_path / "bad2.bam"
    _write_bam(bad2, refs, [dict(name="starts_with_D", ref=0, pos=10, flag=0, l_seq=10, cigar=[(5, "D"), (10, "=")])])
    assert rb("stats", bad2)[0] == 101 and oracle.cli("stats", bad2)[0] == 101


def test_stats_bam_prints_what_came_before_a_panic(oracle, golden, tmp_path):
    """The reference prints record by record (main.rs:71-76), so the header and the lines of the records before the one that panics
    are on stdout when it dies; `rb` must leave the same bytes behind (the stdout buffer used to die with the try block)."""
    refs = [("chrA", 100000)]
    good = [dict(name=f"ok{i}", ref=0, pos=100 * i, flag=0, l_seq=30, cigar=[(10, "="), (2, "X"), (18, "=")]) for i in range(5)]
    bad = dict(name="ends_in_D", ref=0, pos=900, flag=0, l_seq=10, cigar=[(10, "="), (5, "D")])
    bam = tmp_path / "mid_panic.bam"
    _write_bam(bam, refs, good[:3] + [bad] + good[3:])
    rc, out = rb("stats", bam)
    orc, oout = oracle.cli("stats", bam)
    assert (rc, orc) == (101, 101)
    assert out == oout and out.count(b"\n") == 1 + 3        # header + the three records before the panic
    # nucfreq: the regions before an unknown contig are printed (main.rs:98-120 goes region by region)
    bed = tmp_path / "r.bed"
    bed.write_text("CHROMOSOME_I\t0\t50\nnope\t0\t10\n")
    a = ["nucfreq", "-b", bed, f"{golden}/test_nucfreq.bam"]
    rc, out = rb(*a)
    orc, oout = oracle.cli(*a)
    assert (rc, orc) == (101, 101) and out == oout and len(out) > 50


def test_stats_truncated_and_malformed_bam(oracle, golden, tmp_path):
    """htslib's bam_read1 refuses a file that ends inside a record, a block_size below 32 and a record whose name + cigar + sequence
    do not fit its block; the reference's rec.unwrap() (main.rs:72) then panics after the records it has printed.  The host
    decoder used to trust every length field."""
    import gzip
    import struct
    raw = gzip.decompress(open(f"{golden}/stats.bam", "rb").read())
    # walk the header to the first record
    p = 12 + struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, p - 4)[0]
    for _ in range(n_ref):
        p += 4 + struct.unpack_from("<i", raw, p)[0] + 4
    recs = []
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]
        recs.append((p, bs))
        p += 4 + bs
    assert len(recs) > 20
    k = 12
    cut_at = recs[k][0] + 4 + recs[k][1] // 2            # inside record k
    cases = {"cut_in_record": raw[:cut_at], "cut_in_length": raw[:recs[k][0] + 2], "cut_in_header": raw[:20]}
    tiny = bytearray(raw)                                 # block_size 16 for record k
    struct.pack_into("<i", tiny, recs[k][0], 16)
    cases["block_size_16"] = bytes(tiny[:recs[k][0] + 4 + 16])
    big = bytearray(raw)                                  # n_cigar_op = 65535 in record k: the cigar runs past the block
    struct.pack_into("<H", big, recs[k][0] + 4 + 12, 65535)
    cases["cigar_past_block"] = bytes(big)
    lseq = bytearray(raw)                                 # l_seq = 2^31 - 1
    struct.pack_into("<i", lseq, recs[k][0] + 4 + 16, 0x7FFFFFFF)
    cases["l_seq_past_block"] = bytes(lseq)
    noname = bytearray(raw)                               # read name without its NUL
    l_rn = raw[recs[k][0] + 4 + 8]
    noname[recs[k][0] + 4 + 32 + l_rn - 1] = ord("x")
    cases["name_not_terminated"] = bytes(noname)
    full_rc, full_out = rb("stats", f"{golden}/stats.bam")
    assert full_rc == 0
    for name, data in cases.items():
        f = tmp_path / f"{name}.bam"
        with gzip.open(f, "wb") as g:
            g.write(data)
        rc, out = rb("stats", f)
        assert rc == 101, name
        if name == "cut_in_header":
            continue
        orc, oout = oracle.cli("stats", f)
        assert orc == 101 and out == oout, name
        assert full_out.startswith(out) and out.count(b"\n") >= 2, name    # header + the mapped records before record k
        # nucfreq reads the same records: it must refuse the file too, not read past the block
        assert rb("nucfreq", "-r", "chr1:1-100", f)[0] in (101, 1), name


def test_readme_pipeline(oracle, golden, tmp_path):
    """The reference README's showcase chain (trim-paf | break-paf | orient | liftover | filter | stats), every stage through
    `rb`, against the same chain through the oracle CLI; each intermediate file must be byte-identical too."""
    bed = tmp_path / "rgn.bed"
    bed.write_text("chr22\t12000000\t13000000\n")
    stages = [
        ["trim-paf", "{inp}"],
        ["break-paf", "--max-size", "100", "{inp}"],
        ["orient", "{inp}"],
        ["liftover", "--bed", str(bed), "{inp}"],
        ["filter", "--paired-len", "10000", "{inp}"],
        ["stats", "--paf", "{inp}"],
    ]
    inp = f"{golden}/asm_small.paf"
    for i, st in enumerate(stages):
        a = [x.format(inp=inp) for x in st]
        rc, out = rb(*a)
        orc, oout = oracle.cli(*a)
        assert (rc, orc) == (0, 0), st
        assert out == oout, st
        assert out.count(b"\n") > 0, st
        nxt = tmp_path / f"stage{i}.paf"
        nxt.write_bytes(out)
        inp = str(nxt)
    assert hashlib.md5(out).hexdigest() == "c4e325dd79f7e580f63a388e1636ca97"


def test_liftover_general_path_still_agrees(dig, golden, tmp_path):
    """`rb liftover` normally goes text -> text with the CIGAR text parsed / printed on the device; RB_GENERAL_PATH=1 keeps
    the record-based path (the one --qbed / --largest use) under test on the same inputs."""
    bed = str(tmp_path / "tile.bed")
    tile_bed(bed)
    for env in (None, {"RB_GENERAL_PATH": "1"}):
        rc, out = rb("liftover", "--bed", bed, f"{golden}/asm_small.paf", env=env)
        assert rc == 0 and hashlib.md5(out).hexdigest() == dig["liftover_tile_100kb"]["md5"], env


@pytest.mark.parametrize("args", [["break-paf", "--max-size", "100", "{paf}"], ["break-paf", "--max-size", "7", "{paf}"],
                                  ["stats", "--paf", "{paf}"], ["stats", "--qbed", "--paf", "{paf}"]])
def test_text_and_general_paths_agree(oracle, golden, args):
    a = [x.format(paf=f"{golden}/asm_small.paf") for x in args]
    rc, out = rb(*a)
    rc2, out2 = rb(*a, env={"RB_GENERAL_PATH": "1"})
    orc, oout = oracle.cli(*a)
    assert (rc, rc2, orc) == (0, 0, 0)
    assert out == oout and out2 == oout


def test_liftover_text_path_panics_and_skips(tmp_path):
    bed = tmp_path / "r.bed"
    bed.write_text("T\t0\t100\n")
    bad = tmp_path / "bad.paf"
    bad.write_text("Q 10 0 5 + T 10 0 5 0 0 60 cg:Z:5=\nQ 10 0 5 + T 10 0 5 0 0 60 cg:Z:5Q\n")
    assert rb("liftover", "--bed", bed, bad)[0] == 101   # "Unable to parse cigar string." (paf.rs:399)
    bad.write_text("Q 10 0 5 + T 10 0 6 0 0 60 cg:Z:5=\n")
    assert rb("liftover", "--bed", bed, bad)[0] == 101   # check_integrity().unwrap() (paf.rs:70)
    bad.write_text("Q 10 0 5 + T 10\n")
    assert rb("liftover", "--bed", bed, bad)[0] == 101   # assert!(t.len() >= 12)
    # a line with a bad numeric column AND a malformed cigar: PafRecord::new parses the tags first (paf.rs:387-399), so the
    # reference panics instead of skipping the line; the text path and the general path must agree on that
    bad.write_text("Q 10 0 5 + T 10 0 5 0 0 60 cg:Z:5=\nQ x 0 5 + T 10 0 5 0 0 60 cg:Z:5Q\n")
    assert rb("liftover", "--bed", bed, bad)[0] == 101
    assert rb("liftover", "--bed", bed, bad, env={"RB_GENERAL_PATH": "1"})[0] == 101
    ok = tmp_path / "ok.paf"
    ok.write_text("Q x 0 5 + T 10 0 5 0 0 60 cg:Z:5=\nQ 10 0 5 + T 10 0 5 0 0 60 cg:Z:5=\n")
    rc, out = rb("liftover", "--bed", bed, ok)
    assert rc == 0 and out.count(b"\n") == 1            # the unparsable line is skipped (paf.rs:73)
    rc2, out2 = rb("liftover", "--bed", bed, ok, env={"RB_GENERAL_PATH": "1"})
    assert (rc2, out2) == (rc, out)


def test_empty_and_tagless_inputs(oracle, tmp_path):
    bed = tmp_path / "r.bed"
    bed.write_text("T\t0\t100\n")
    empty = tmp_path / "empty.paf"
    empty.write_text("")
    for args in (["liftover", "--bed", bed, empty], ["break-paf", empty], ["stats", "--paf", empty]):
        rc, out = rb(*args)
        orc, oout = oracle.cli(*args)
        assert (rc, out) == (orc, oout), args
    nocg = tmp_path / "nocg.paf"  # a record without a cg tag has an empty CIGAR: whatever the reference path does with it,
    nocg.write_text("Q 10 0 5 + T 10 0 5 0 0 60 tp:A:P\n")  # the text path and the record path must agree
    for args in (["liftover", "--bed", bed, nocg], ["stats", "--paf", nocg]):
        a = rb(*args)
        b = rb(*args, env={"RB_GENERAL_PATH": "1"})
        assert a == b, args


def test_zero_padded_cigar_numbers(oracle, tmp_path):
    """u32::from_str accepts leading zeros; a number longer than a 16-byte lane makes the text path step aside"""
    bed = tmp_path / "r.bed"
    bed.write_text("T\t1\t4\n")
    paf = tmp_path / "z.paf"
    paf.write_text("Q 10 0 5 + T 10 0 5 0 0 60 cg:Z:00000000000000000005=\nQ 10 0 5 + T 10 0 5 0 0 60 cg:Z:003=02X\n")
    for args in (["liftover", "--bed", bed, paf], ["stats", "--paf", paf], ["break-paf", paf]):
        rc, out = rb(*args)
        orc, oout = oracle.cli(*args)
        assert (rc, out) == (orc, oout) and rc == 0 and out, args


# ---- `rb nucfreq`: BAM decode on the host, the pileup on the device ----
@pytest.mark.parametrize("small", [False, True])
def test_nucfreq_ka13_fixture(oracle, golden, small):
    a = ["nucfreq", "-r", "CHROMOSOME_I:2-102"] + (["-s"] if small else []) + [f"{golden}/test_nucfreq.bam"]
    rc, out = rb(*a)
    orc, oout = oracle.cli(*a)
    assert (rc, orc) == (0, 0)
    assert out == oout and out.count(b"\n") == 102


def test_nucfreq_bed_regions_and_pieces(oracle, golden, tmp_path):
    """regions from --region and a BED file (4th column = id), one of them longer than the 1 Mbp print piece"""
    import subprocess
    rc, st = rb("stats", f"{golden}/asm_small.bam")
    f = st.splitlines()[1].split(b"\t")
    name, r_st = f[0].decode(), int(f[1])
    bed = tmp_path / "r.bed"
    bed.write_text(f"{name}\t{r_st + 5000}\t{r_st + 5200}\tmy_id\n{name}\t{max(r_st - 300, 0)}\t{r_st + 300}\n{name}\t{r_st + 999000}\t{r_st + 2001500}\tlong\n")
    for extra in ([], ["--small"]):
        a = ["nucfreq", "--region", f"{name}:{r_st + 10001}-{r_st + 10050}", "--bed", str(bed)] + extra + [f"{golden}/asm_small.bam"]
        rc, out = rb(*a)
        orc, oout = oracle.cli(*a)
        assert (rc, orc) == (0, 0)
        assert out == oout
        if not extra:
            assert out.count(b"#chr\tstart") == 1 + 1 + 1 + 2 and out.count(b"\tmy_id\n") == 200


def test_nucfreq_unknown_contig_panics(oracle, golden):
    a = ["nucfreq", "-r", "nope:1-100", f"{golden}/test_nucfreq.bam"]
    rc, out = rb(*a)
    orc, oout = oracle.cli(*a)
    assert rc == 101 and orc == 101 and out == oout == b""


def test_nucfreq_synthetic_bam_flags_and_clips(oracle, tmp_path):
    refs = [("chrA", 50000), ("chrB", 30000)]
    recs = [
        dict(name="r1", ref=0, pos=100, flag=0, l_seq=50, cigar=[(5, "S"), (20, "M"), (3, "I"), (4, "D"), (22, "M")]),
        dict(name="r2", ref=0, pos=110, flag=16, l_seq=30, cigar=[(2, "H"), (10, "="), (1, "X"), (100, "N"), (19, "M"), (3, "H")]),
        dict(name="dup", ref=0, pos=115, flag=1024, l_seq=10, cigar=[(10, "M")]),
        dict(name="sec", ref=0, pos=118, flag=256, l_seq=10, cigar=[(10, "M")]),
        dict(name="r3", ref=0, pos=4090, flag=0, l_seq=20, cigar=[(20, "M")]),
        dict(name="r4", ref=1, pos=0, flag=0, l_seq=12, cigar=[(12, "M")]),
        dict(name="un", ref=-1, pos=-1, flag=4, l_seq=5, cigar=[]),
    ]
    import struct
    real = [(10, "M"), (2, "I"), (6, "D"), (28, "M")]   # htslib's long-cigar convention: <l_seq>S<ref_len>N in the record, the real cigar in CG:B,I
    cg = b"CGBI" + struct.pack("<i", len(real)) + b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(c)) for l, c in real)
    recs.insert(6, dict(name="long_cigar_in_CG", ref=1, pos=100, flag=0, l_seq=40, cigar=[(40, "S"), (44, "N")], aux=cg))
    p = tmp_path / "s.bam"
    _write_bam(str(p), refs, recs)
    for a in (["-r", "chrA:1-50000"], ["-r", "chrA:105-125", "-s"], ["-r", "chrB:1-4294967295"], ["-r", "chrA:4097-4100"], ["-r", "chrB:90-160"]):
        rc, out = rb("nucfreq", *a, str(p))
        orc, oout = oracle.cli("nucfreq", *a, str(p))
        assert (rc, orc) == (0, 0), a
        assert out == oout, a
    rc, out = rb("nucfreq", "-r", "chrA:1-50000", str(p))
    assert out.count(b"\n") == 1 + 140 + 20   # r1 [100,146) and r2 [110,240) merge; r3 [4090,4110); dup / secondary reads add nothing


def test_nucfreq_many_bed_regions_share_device_calls(oracle, golden, tmp_path):
    """hundreds of small regions, contigs interleaved, some overlapping, some empty: the host groups consecutive regions of a
    contig into one device call; the text must still be the reference's, region by region"""
    import random
    rc, st = rb("stats", f"{golden}/asm_small.bam")
    recs = [l.split(b"\t") for l in st.splitlines()[1:]]
    rnd = random.Random(3)
    lines = []
    for k in range(240):
        f = recs[rnd.randrange(len(recs))] if k % 7 else recs[k % 3]
        name, r_st, r_en = f[0].decode(), int(f[1]), int(f[2])
        a = rnd.randrange(max(r_st - 2000, 0), r_en + 2000)
        b = a + (0 if k % 41 == 0 else rnd.randrange(1, 3000))
        lines.append(f"{name}\t{a}\t{b}" + (f"\tid{k}" if k % 3 == 0 else ""))
    bed = tmp_path / "many.bed"
    bed.write_text("\n".join(lines) + "\n")
    for extra in ([], ["-s"]):
        a = ["nucfreq", "--bed", str(bed)] + extra + [f"{golden}/asm_small.bam"]
        rc, out = rb(*a)
        orc, oout = oracle.cli(*a)
        assert (rc, orc) == (0, 0)
        assert out == oout and out.count(b"\n") > 1000


def test_nucfreq_bai_index_and_full_scan_agree(oracle, golden, tmp_path):
    """with <bam>.bai next to the file only the BGZF members the regions need are inflated; without it (RB_NO_BAI) the whole file is.
    Same text either way, and the oracle's (which always scans)"""
    import subprocess
    rc, st = rb("stats", f"{golden}/asm_small.bam")
    recs = [l.split(b"\t") for l in st.splitlines()[1:]]
    bed = tmp_path / "r.bed"
    bed.write_text("".join(f"{f[0].decode()}\t{max(int(f[1]) - 50, 0)}\t{int(f[1]) + 400}\n" for f in recs[::9]) +
                   "".join(f"{f[0].decode()}\t{int(f[2]) - 300}\t{int(f[2]) + 50}\n" for f in recs[4::11]))
    for args in (["--bed", str(bed)], ["-r", f"{recs[0][0].decode()}:{int(recs[0][1]) + 1}-{int(recs[0][1]) + 70000}", "-s"],
                 ["-r", "chr1:1-1000"]):
        a = ["nucfreq", *args, f"{golden}/asm_small.bam"]
        rc, out = rb(*a)
        env = dict(os.environ, RB_NO_BAI="1")
        full = subprocess.run([RB, *a], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        orc, oout = oracle.cli(*a)
        assert rc == 0 and full.returncode == 0 and orc == 0
        assert out == full.stdout == oout
    assert os.path.exists(f"{golden}/asm_small.bam.bai")


def _shuffled(golden, tmp_path, seed):
    """the fixture's records in random order: target contigs and query names interleave (the fixture itself is sorted by target)"""
    import random
    lines = [l for l in open(f"{golden}/asm_small.paf", "rb").read().split(b"\n") if l]
    random.Random(seed).shuffle(lines)
    p = tmp_path / f"shuf{seed}.paf"
    p.write_bytes(b"\n".join(lines) + b"\n")
    return str(p)


def _qbed(golden, tmp_path):
    """windows in QUERY coordinates that do hit: a stretch of every fifth record's query span (the fixture's own trim_asm_small.bed
    hits nothing under --qbed, which made the old check vacuous)"""
    p = tmp_path / "q.bed"
    recs = [l.split("\t") for l in open(f"{golden}/asm_small.paf")]
    p.write_text("".join(f"{f[0]}\t{int(f[2]) + 100}\t{min(int(f[3]), int(f[2]) + 30000)}\n" for f in recs[::5]))
    return str(p)


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("args", [
    ["liftover", "--bed", "{bed}", "{paf}"],
    ["liftover", "--qbed", "--bed", "{qbed}", "{paf}"],
    ["liftover", "--qbed", "--largest", "--bed", "{qbed}", "{paf}"],
    ["liftover", "--largest", "--bed", "{bed}", "{paf}"],
    ["liftover", "--bed", "{bed}", "{paf}.gz"],
    ["break-paf", "--max-size", "100", "{paf}"],
    ["stats", "--paf", "{paf}"],
    ["invert", "{paf}"],
    ["trim-paf", "{paf}"],
    ["trim-paf", "-r", "{paf}"],
])
def test_gpus_flag_gathers_the_single_gpu_bytes(golden, args, n, tmp_path):
    """`rb --gpus N`: N worker processes forked before the GPU is touched, each on its own share of the lines; the outputs are put
    together in the reference's order (contig-major for liftover, liftover.rs:151-164).  One GPU here: RB_GPUS_SAME_DEVICE=1 puts
    every worker on device 0 (the fork, the cuts and the gather are what is checked; the per-device part is the ordinary
    single-GPU run).  Checked on the fixture (sorted by target) AND on shuffled copies where contigs and query names interleave,
    into a pipe and into a regular file (workers pwrite at offsets)."""
    srcs = [f"{golden}/asm_small.paf"] + ([] if ".gz" in args[-1] else [_shuffled(golden, tmp_path, 7), _shuffled(golden, tmp_path, 8)])
    for src in srcs:
        a = [x.format(paf=src, bed=f"{golden}/asm_small.bed", qbed=_qbed(golden, tmp_path)) for x in args]
        rc1, out1 = rb(*a)
        rcn, outn = rb("--gpus", n, *a, env={"RB_GPUS_SAME_DEVICE": "1"})
        assert (rc1, rcn) == (0, 0)
        assert len(out1) > 1000, a
        assert outn == out1, (a, n)
        with open(tmp_path / "out.paf", "wb") as f:
            r = subprocess.run([RB, "--gpus", str(n), *a], stdout=f, stderr=subprocess.PIPE, env={**os.environ, "RB_GPUS_SAME_DEVICE": "1"})
        assert r.returncode == 0 and open(tmp_path / "out.paf", "rb").read() == out1, (a, n, "file")


def test_gpus_flag_interleaved_contigs_vs_oracle(golden, oracle, tmp_path):
    """the order itself (not only `--gpus N` == `--gpus 1`) against the oracle CLI on a shuffled input"""
    src = _shuffled(golden, tmp_path, 11)
    for a in (["liftover", "--bed", f"{golden}/asm_small.bed", src], ["liftover", "--qbed", "--bed", _qbed(golden, tmp_path), src],
              ["liftover", "--largest", "--bed", f"{golden}/asm_small.bed", src], ["trim-paf", src]):
        orc, oout = oracle.cli(*a)
        rc, out = rb("--gpus", 3, *a, env={"RB_GPUS_SAME_DEVICE": "1"})
        assert (orc, rc) == (0, 0) and out == oout and len(out) > 1000, a


def test_gpus_flag_stdin_general_path_and_refusals(golden):
    paf = open(f"{golden}/asm_small.paf", "rb").read()
    assert os.path.exists(RB)
    env = {**os.environ, "RB_GPUS_SAME_DEVICE": "1"}
    one = subprocess.run([RB, "break-paf"], input=paf, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    two = subprocess.run([RB, "--gpus", "2", "break-paf"], input=paf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert (one.returncode, two.returncode) == (0, 0) and one.stdout == two.stdout
    gen = subprocess.run([RB, "--gpus", "2", "break-paf", f"{golden}/asm_small.paf"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env={**env, "RB_GENERAL_PATH": "1"})
    assert gen.returncode == 0 and gen.stdout == one.stdout
    tone = subprocess.run([RB, "trim-paf"], input=paf, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    ttwo = subprocess.run([RB, "--gpus", "2", "trim-paf"], input=paf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert (tone.returncode, ttwo.returncode) == (0, 0) and tone.stdout == ttwo.stdout and len(tone.stdout) > 1000
    # commands that are not a map over PAF records are refused, not silently run on one GPU
    for a in (["orient", f"{golden}/asm_small.paf"], ["stats", f"{golden}/asm_small.bam"]):
        r = subprocess.run([RB, "--gpus", "2", *a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 2 and not r.stdout


def test_stdin_takes_the_text_route_and_prints_what_the_file_prints(golden):
    """`rb trim-paf x | rb break-paf -` (README.md:22-23): a command that reads stdin keeps the text and takes the device text route like
    a file (round 3 sent stdin down the line-by-line route); same bytes as the file run for every hot-path command."""
    paf = open(f"{golden}/asm_small.paf", "rb").read()
    for a in (["break-paf", "--max-size", "100"], ["trim-paf"], ["stats", "--paf"], ["invert"], ["liftover", "--bed", f"{golden}/asm_small.bed"]):
        rc_f, out_f = rb(*a, f"{golden}/asm_small.paf")
        r = subprocess.run([RB, *a, "-"], input=paf, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert rc_f == 0 and r.returncode == 0 and r.stdout == out_f and len(out_f) > 500, a
    # the README pipeline, through a real pipe
    p1 = subprocess.Popen([RB, "trim-paf", f"{golden}/asm_small.paf"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p2 = subprocess.run([RB, "break-paf", "--max-size", "100", "-"], stdin=p1.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p1.stdout.close()
    assert p1.wait() == 0 and p2.returncode == 0
    rc_t, trimmed = rb("trim-paf", f"{golden}/asm_small.paf")
    want = subprocess.run([RB, "break-paf", "--max-size", "100", "-"], input=trimmed, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert p2.stdout == want.stdout and p2.stdout.count(b"\n") > 2000


def test_gpus_flag_panic_in_one_shard(golden, tmp_path):
    """a line the reference panics on, in the second shard: what the single run printed before the panic (the stats header, nothing
    for the other commands), exit code 101"""
    lines = open(f"{golden}/asm_small.paf", "rb").read().split(b"\n")
    lines = [l for l in lines if l]
    bad = lines[-2].replace(b"cg:Z:", b"cg:Z:12Q")
    src = tmp_path / "bad.paf"
    src.write_bytes(b"\n".join(lines[:-2] + [bad, lines[-1]]) + b"\n")
    rc1, out1 = rb("stats", "--paf", src)
    rc2, out2 = rb("--gpus", 2, "stats", "--paf", src, env={"RB_GPUS_SAME_DEVICE": "1"})
    assert rc1 == 101 and rc2 == 101
    assert out2 == out1 and out1.startswith(b"#")
    rc3, out3 = rb("--gpus", 2, "break-paf", src, env={"RB_GPUS_SAME_DEVICE": "1"})
    assert rc3 == 101 and out3 == b""


@pytest.mark.parametrize("args", [
    ["liftover", "--bed", "{bed}", "{paf}"],
    ["break-paf", "--max-size", "100", "{paf}"],
    ["--bsearch", "legacy", "break-paf", "--max-size", "10", "{paf}"],
])
def test_pipelined_text_route_equals_the_whole_file_route(golden, args, tmp_path):
    """big plain files go through a pipeline over chunks (rb_host lift_file_text_pipelined; RB_CHUNK_KB makes the 2 MB fixture 'big'):
    same bytes as the whole-file route -- streamed into a regular file when the input is sorted by target, after a rewind when it
    is not (shuffled copy: contig ranks decrease along the chunks), kept and ordered at the end into a pipe and under --gpus"""
    for src in (f"{golden}/asm_small.paf", _shuffled(golden, tmp_path, 21)):
        a = [x.format(paf=src, bed=f"{golden}/asm_small.bed") for x in args]
        rc0, want = rb(*a, env={"RB_NO_PIPELINE": "1"})
        assert rc0 == 0 and len(want) > 1000
        for kb in ("64", "300", "900"):
            env = {"RB_CHUNK_KB": kb, "RB_GPUS_SAME_DEVICE": "1"}
            if kb == "300":
                env["RB_MMAP_WRITE_MIN"] = "1"  # (the route big outputs take into a file: a shared mapping instead of pwrite)
            rc1, piped = rb(*a, env=env)
            assert rc1 == 0 and piped == want, (a, kb, "pipe")
            with open(tmp_path / "o.paf", "wb") as f:
                r = subprocess.run([RB, *a], stdout=f, stderr=subprocess.PIPE, env={**os.environ, **env})
            assert r.returncode == 0 and open(tmp_path / "o.paf", "rb").read() == want, (a, kb, "file")
            rc2, sharded = rb("--gpus", 2, *a, env=env)
            assert rc2 == 0 and sharded == want, (a, kb, "--gpus 2")
    # a line the reference panics on, in a late chunk: exit code 101, nothing but what the single run prints
    lines = [l for l in open(f"{golden}/asm_small.paf", "rb").read().split(b"\n") if l]
    bad = tmp_path / "bad.paf"
    bad.write_bytes(b"\n".join(lines[:-3] + [lines[-3].replace(b"cg:Z:", b"cg:Z:7Q")] + lines[-2:]) + b"\n")
    a = [x.format(paf=str(bad), bed=f"{golden}/asm_small.bed") for x in args]
    rc0, out0 = rb(*a, env={"RB_NO_PIPELINE": "1"})
    rc1, out1 = rb(*a, env={"RB_CHUNK_KB": "200"})
    assert rc0 == 101 and rc1 == 101 and out0 == b""


def test_pipelined_route_reports_the_panic_the_reference_reports(golden, tmp_path):
    """A record that loads but panics in the liftover stage (a leading deletion at t_st = 0: remove_trailing_indels, paf.rs:656-783,
    then unwrap) in an EARLY chunk, and a line Paf::from_file panics on (paf.rs:381: fewer than 12 columns) in a LATE one.  The
    reference reads the whole file before it lifts anything, so the parse panic is the one it dies of; the pipelined route must say
    what the whole-file route says (it reported the first panic in chunk order)."""
    lines = [l for l in open(f"{golden}/asm_small.paf", "rb").read().split(b"\n") if l]
    late = b"q\t200\t0\t100\t+\t" + lines[0].split(b"\t")[5] + b"\t100000000\t0\t105\t100\t205\t60\tcg:Z:5D100="
    bad = tmp_path / "two_panics.paf"
    bad.write_bytes(b"\n".join(lines[:40] + [late] + lines[40:-2] + [b"short\tline"] + lines[-2:]) + b"\n")
    for a in (["liftover", "--bed", f"{golden}/asm_small.bed", str(bad)], ["break-paf", "--max-size", "100", str(bad)]):
        whole = subprocess.run([RB, *a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "RB_NO_PIPELINE": "1"})
        piped = subprocess.run([RB, *a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "RB_CHUNK_KB": "200"})
        assert whole.returncode == 101 and piped.returncode == 101, a
        assert piped.stderr == whole.stderr and piped.stdout == whole.stdout == b"", (a, piped.stderr, whole.stderr)
    # the early record alone: the liftover-stage panic is what both routes report
    only = tmp_path / "one_panic.paf"
    only.write_bytes(b"\n".join(lines[:40] + [late] + lines[40:]) + b"\n")
    a = ["liftover", "--bed", f"{golden}/asm_small.bed", str(only)]
    whole = subprocess.run([RB, *a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "RB_NO_PIPELINE": "1"})
    piped = subprocess.run([RB, *a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "RB_CHUNK_KB": "200"})
    assert whole.returncode == 101 and piped.returncode == 101 and piped.stderr == whole.stderr and piped.stdout == b""


def test_trim_paf_in_place_and_copied_clips_print_the_same(oracle, tmp_path):
    """rb trim-paf cuts regular records where they are (RB_TRIM_IN_PLACE: two words a record) and copies the clips of the others behind
    the ops in use; RB_TRIM_COPY=1 copies all of them as rounds 1 and 2 did.  Same bytes, and the oracle's on the first queries."""
    paf = tmp_path / "c4.paf"
    with open(paf, "wb") as f:
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_config4_paf.py"), "4000"], stdout=f)
    rc, out = rb("trim-paf", paf)
    rc2, out2 = rb("trim-paf", paf, env={"RB_TRIM_COPY": "1"})
    assert (rc, rc2) == (0, 0) and out == out2 and out.count(b"\n") == 4000
    head = tmp_path / "head.paf"
    head.write_bytes(b"".join(open(paf, "rb").readlines()[:400]))
    rc3, out3 = rb("trim-paf", head)
    orc, oout = oracle.cli("trim-paf", head)
    assert (rc3, orc) == (0, 0) and out3 == oout
