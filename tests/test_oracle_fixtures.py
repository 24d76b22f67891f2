"""Whole-file digests of the oracle CLI on the reference fixture (tests/golden/digests.json)."""
import hashlib
import json
import os

import pytest

from golden.make_digests import tile_bed


@pytest.fixture(scope="module")
def dig(golden):
    return json.load(open(os.path.join(golden, "digests.json")))


def _md5(oracle, *a):
    rc, out = oracle.cli(*a)
    assert rc == 0
    return hashlib.md5(out).hexdigest(), out.count(b"\n")


def test_stats(oracle, dig, golden):
    assert _md5(oracle, "stats", "--paf", f"{golden}/asm_small.paf") == (dig["stats_paf"]["md5"], 250)


def test_liftover_fixture_bed(oracle, dig, golden):
    for pol in ("modern", "legacy"):
        assert _md5(oracle, "--bsearch", pol, "liftover", "--bed", f"{golden}/asm_small.bed",
                    f"{golden}/asm_small.paf") == (dig["liftover_asm_small_bed"]["md5"], 12)


def test_liftover_tiled(oracle, dig, golden, tmp_path):
    bed = str(tmp_path / "tile.bed")
    tile_bed(bed)
    assert _md5(oracle, "liftover", "--bed", bed, f"{golden}/asm_small.paf") == (dig["liftover_tile_100kb"]["md5"], 1657)


def test_break_paf(oracle, dig, golden):
    assert _md5(oracle, "break-paf", "--max-size", 100, f"{golden}/asm_small.paf") == (dig["break_paf_100_modern"]["md5"], 2447)
    assert _md5(oracle, "--bsearch", "legacy", "break-paf", "--max-size", 100, f"{golden}/asm_small.paf")[0] == \
        dig["break_paf_100_legacy"]["md5"]


def test_trim_paf_and_invert(oracle, dig, golden):
    assert _md5(oracle, "trim-paf", f"{golden}/asm_small.paf") == (dig["trim_paf_modern"]["md5"], 249)
    assert _md5(oracle, "--bsearch", "legacy", "trim-paf", f"{golden}/asm_small.paf")[0] == dig["trim_paf_legacy"]["md5"]
    assert _md5(oracle, "invert", f"{golden}/asm_small.paf")[0] == dig["invert"]["md5"]
