"""`rb --gpus N` puts the workers' outputs together on the host (no collective).  The gather -- line cuts, query-name
ranges, the piece index, contig-major order over shards (liftover.rs:151-164), the --largest reduction (main.rs:200-208), pipes and
pwrite-at-offsets -- needs no device, so it is checked here on the CPU through `rb regroup`, a diagnostic arm that prints the lines
of a PAF in the order the hot-path commands emit.  tests/test_gpu_cli.py repeats the check with the real commands on a GPU."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RB = os.path.join(ROOT, "rustybam_amd", "rb")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def shuffled_paf(tmp_path, seed, header_only=True):
    """the fixture's records in random order: target contigs (and query names) interleave"""
    lines = [l for l in open(os.path.join(GOLDEN, "asm_small.paf"), "rb").read().split(b"\n") if l]
    if header_only:  # (the CIGARs do not matter to the gather; short lines make many shards per kilobyte)
        lines = [b"\t".join(l.split(b"\t")[:12]) + b"\tcg:Z:" + l.split(b"cg:Z:")[1][:int(40 + 300 * random.Random(i).random())]
                 for i, l in enumerate(lines)]
    random.Random(seed).shuffle(lines)
    p = tmp_path / f"shuf{seed}.paf"
    p.write_bytes(b"\n".join(lines) + b"\n")
    return str(p)


def run(args, to_file=None, stdin=None):
    if to_file is not None:
        with open(to_file, "wb") as f:
            r = subprocess.run([RB, *map(str, args)], stdout=f, stderr=subprocess.PIPE, stdin=stdin)
        return r.returncode, open(to_file, "rb").read()
    r = subprocess.run([RB, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=stdin)
    return r.returncode, r.stdout


@pytest.mark.skipif(not os.path.exists(RB), reason="rustybam_amd/rb not built")
@pytest.mark.parametrize("flags", [[], ["-q"], ["-l"]])
@pytest.mark.parametrize("n", [2, 3, 8])
def test_gather_equals_the_single_run(tmp_path, flags, n):
    for seed in (1, 2):
        paf = shuffled_paf(tmp_path, seed)
        rc1, one = run(["regroup", *flags, paf])
        assert rc1 == 0 and one.count(b"\n") > (3 if flags == ["-l"] else 200)
        rcp, piped = run(["--gpus", n, "regroup", *flags, paf])
        rcf, filed = run(["--gpus", n, "regroup", *flags, paf], to_file=tmp_path / "out.paf")
        assert (rcp, rcf) == (0, 0)
        assert piped == one, f"pipe gather, {flags} n={n} seed={seed}"
        assert filed == one, f"file gather, {flags} n={n} seed={seed}"


@pytest.mark.skipif(not os.path.exists(RB), reason="rustybam_amd/rb not built")
def test_interleaved_contigs_are_contig_major(tmp_path):
    """`A B | A B` must come out `A A B B` (the old gather printed `A B A B`)"""
    def rec(q, t, st):
        return f"{q}\t100\t0\t10\t+\t{t}\t1000\t{st}\t{st + 10}\t10\t10\t60\tcg:Z:10=\n"
    p = tmp_path / "abab.paf"
    p.write_text(rec("q1", "A", 0) + rec("q2", "B", 0) + rec("q3", "A", 50) + rec("q4", "B", 50))
    rc, out = run(["--gpus", 2, "regroup", str(p)])
    assert rc == 0 and [l.split("\t")[0] for l in out.decode().splitlines()] == ["q1", "q3", "q2", "q4"]
    # a contig whose first appearance is in shard 0 but which only ... keeps its rank: B first, then A
    p.write_text(rec("q1", "B", 0) + rec("q2", "B", 20) + rec("q3", "A", 50) + rec("q4", "B", 50))
    rc, out = run(["--gpus", 2, "regroup", str(p)])
    assert rc == 0 and [l.split("\t")[0] for l in out.decode().splitlines()] == ["q1", "q2", "q4", "q3"]


@pytest.mark.skipif(not os.path.exists(RB), reason="rustybam_amd/rb not built")
def test_gather_from_stdin_and_gzip(tmp_path):
    paf = shuffled_paf(tmp_path, 5)
    rc1, one = run(["regroup", paf])
    with open(paf, "rb") as f:
        rc2, two = run(["--gpus", 3, "regroup"], stdin=f)
    subprocess.check_call(["gzip", "-k", paf])
    rc3, three = run(["--gpus", 2, "regroup", paf + ".gz"])
    assert (rc1, rc2, rc3) == (0, 0, 0) and one == two == three


@pytest.mark.skipif(not os.path.exists(RB), reason="rustybam_amd/rb not built")
def test_more_shards_than_query_names(tmp_path):
    p = tmp_path / "few.paf"
    line = "{q}\t100\t0\t10\t+\tT\t1000\t0\t10\t10\t10\t60\tcg:Z:10=\n"
    p.write_text("".join(line.format(q=q) for q in ["b", "a", "b", "a", "a"]))
    rc, out = run(["--gpus", 8, "regroup", "-q", str(p)])
    assert rc == 0 and [l.split("\t")[0] for l in out.decode().splitlines()] == ["a", "a", "a", "b", "b"]
    (tmp_path / "empty.paf").write_text("")
    for flags in ([], ["-q"], ["-l"]):
        rc, out = run(["--gpus", 4, "regroup", *flags, str(tmp_path / "empty.paf")])
        assert rc == 0 and out == b""


@pytest.mark.skipif(not os.path.exists(RB), reason="rustybam_amd/rb not built")
def test_gather_through_a_shared_mapping(tmp_path):
    """gigabyte outputs are copied into a shared mapping of the output file (parallel page faults) instead of pwrite (one inode
    lock); RB_MMAP_WRITE_MIN=1 takes that route for the small test file: same bytes, with workers and without, appended after
    bytes the file already holds"""
    paf = shuffled_paf(tmp_path, 9)
    rc1, one = run(["regroup", paf])
    assert rc1 == 0
    for pre in ([], ["--gpus", "3"]):
        out = tmp_path / "m.paf"
        with open(out, "wb") as f:
            f.write(b"# header the output follows\n")
            f.flush()
            r = subprocess.run([RB, *pre, "regroup", paf], stdout=f, stderr=subprocess.PIPE, env={**os.environ, "RB_MMAP_WRITE_MIN": "1"})
        assert r.returncode == 0 and open(out, "rb").read() == b"# header the output follows\n" + one, pre
