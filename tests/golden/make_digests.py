"""Regenerates the digests in digests.json from oracle/rb_oracle (run from the repo root)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle  # noqa: E402

G = os.path.dirname(os.path.abspath(__file__))


def tile_bed(path):
    seen = {}
    for line in open(os.path.join(G, "asm_small.paf")):
        t = line.split("\t")
        seen.setdefault(t[5], int(t[6]))
    with open(path, "w") as f:
        for name, L in seen.items():
            for s in range(0, L, 100000):
                f.write(f"{name}\t{s}\t{min(s + 100000, L)}\n")


if __name__ == "__main__":
    paf, bed = os.path.join(G, "asm_small.paf"), os.path.join(G, "asm_small.bed")
    tile_bed("/tmp/rb_tile.bed")
    jobs = {
        "stats_paf": ["stats", "--paf", paf],
        "liftover_asm_small_bed": ["liftover", "--bed", bed, paf],
        "liftover_tile_100kb": ["liftover", "--bed", "/tmp/rb_tile.bed", paf],
        "break_paf_100_modern": ["break-paf", "--max-size", "100", paf],
        "break_paf_100_legacy": ["--bsearch", "legacy", "break-paf", "--max-size", "100", paf],
        "trim_paf_modern": ["trim-paf", paf],
        "trim_paf_legacy": ["--bsearch", "legacy", "trim-paf", paf],
        "invert": ["invert", paf],
    }
    for k, a in jobs.items():
        rc, out = pyoracle.cli(*a)
        print(k, rc, hashlib.md5(out).hexdigest(), out.count(b"\n"))
