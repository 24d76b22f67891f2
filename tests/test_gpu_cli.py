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


def rb(*args):
    assert os.path.exists(RB), "rustybam_amd/rb missing: run __graft_entry__.build()"
    r = subprocess.run([RB, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
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
    rc, out = rb("liftover", "--bed", bed, f"{golden}/asm_small.paf.gz")
    assert rc == 0 and hashlib.md5(out).hexdigest() == dig["liftover_tile_100kb"]["md5"] and out.count(b"\n") == 1657


@pytest.mark.parametrize("args", [
    ["liftover", "--qbed", "--bed", "{trimbed}", "{paf}"],
    ["liftover", "--largest", "--bed", "{bed}", "{paf}"],
    ["trim-paf", "-r", "{paf}"],
    ["trim-paf", "--match-score", "2", "--diff-score", "3", "--indel-score", "5", "{paf}"],
    ["break-paf", "--max-size", "0", "{paf}"],
    ["stats", "--qbed", "--paf", "{paf}"],
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
