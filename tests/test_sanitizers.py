"""CPU-side sanitizer coverage.  GPU AddressSanitizer is not available on the pool, so what runs under sanitizers is
the code that needs no device:
  * the oracle (`make -C oracle asan`: AddressSanitizer + UBSan build of its CLI) over the reference fixture's commands -- output must
    still equal the committed digests, stderr must carry no sanitizer report;
  * the host front end `rb` built with -fsanitize=address,undefined and with -fsanitize=thread (`make -C rustybam_amd/csrc sanitizers`):
    the threaded line splitter and the fork / socket / pwrite gather of `rb --gpus N`, through the CPU-only `rb regroup` arm
    (tests/test_rb_gather_cpu.py runs the same cases on the plain binary)."""
import hashlib
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CSRC = os.path.join(ROOT, "rustybam_amd", "csrc")
REPORT = (b"ERROR: AddressSanitizer", b"ERROR: LeakSanitizer", b"runtime error:", b"WARNING: ThreadSanitizer", b"ERROR: ThreadSanitizer")


def _clean(stderr, what):
    for tag in REPORT:
        assert tag not in stderr, f"{what}: {stderr[-2000:].decode(errors='replace')}"


def _toolchain_has(flags, compiler):
    """does a trivial program build and link under `flags` here?  Only a box WITHOUT the sanitizer runtimes may skip these tests: a
    sanitizer build of the product's sources that fails on a box that has them is a failure, not a skip."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c" if compiler == "gcc" else "t.cpp")
        with open(src, "w") as f:
            f.write("int main(void) { return 0; }\n")
        r = subprocess.run([compiler, *flags, src, "-o", os.path.join(d, "t")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        return r.returncode == 0


@pytest.fixture(scope="module")
def oracle_asan():
    if not _toolchain_has(["-fsanitize=address,undefined"], "gcc"):
        pytest.skip("this toolchain has no libasan / libubsan: nothing to run the oracle under")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])  # (a build error here is a regression, not a skip)
    return os.path.join(ROOT, "oracle", "rb_oracle_asan")


@pytest.mark.parametrize("key,args,lines", [
    ("stats_paf", ["stats", "--paf", "{g}/asm_small.paf"], 250),
    ("liftover_asm_small_bed", ["liftover", "--bed", "{g}/asm_small.bed", "{g}/asm_small.paf"], 12),
    ("break_paf_100_modern", ["break-paf", "--max-size", "100", "{g}/asm_small.paf"], 2447),
    ("break_paf_100_legacy", ["--bsearch", "legacy", "break-paf", "--max-size", "100", "{g}/asm_small.paf"], None),
    ("trim_paf_modern", ["trim-paf", "{g}/asm_small.paf"], 249),
    ("invert", ["invert", "{g}/asm_small.paf.gz"], None),
])
def test_oracle_under_asan_and_ubsan(oracle_asan, key, args, lines):
    dig = json.load(open(os.path.join(GOLDEN, "digests.json")))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([oracle_asan] + [a.format(g=GOLDEN) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    _clean(r.stderr, key)
    assert r.returncode == 0, r.stderr[-500:]
    assert hashlib.md5(r.stdout).hexdigest() == dig[key]["md5"], key
    if lines is not None:
        assert r.stdout.count(b"\n") == lines


@pytest.fixture(scope="module")
def rb_san():
    if not os.path.exists(os.path.join(ROOT, "rustybam_amd", "librustybam_amd.so")):
        pytest.skip("librustybam_amd.so not built")
    if not (_toolchain_has(["-fsanitize=address,undefined"], "g++") and _toolchain_has(["-fsanitize=thread"], "g++")):
        pytest.skip("this toolchain has no libasan / libubsan / libtsan: nothing to run rb under")
    subprocess.check_call(["make", "-s", "-j2", "-C", CSRC, "sanitizers"])  # (a build error here is a regression, not a skip)
    return {"asan": os.path.join(ROOT, "rustybam_amd", "rb_asan"), "tsan": os.path.join(ROOT, "rustybam_amd", "rb_tsan"),
            "plain": os.path.join(ROOT, "rustybam_amd", "rb")}


def _shuffled(tmp_path, seed):
    lines = [l for l in open(os.path.join(GOLDEN, "asm_small.paf"), "rb").read().split(b"\n") if l]
    lines = [b"\t".join(l.split(b"\t")[:12]) + b"\tcg:Z:" + l.split(b"cg:Z:")[1][:int(40 + 300 * random.Random(i).random())] for i, l in enumerate(lines)]
    random.Random(seed).shuffle(lines)
    p = tmp_path / f"shuf{seed}.paf"
    p.write_bytes(b"\n".join(lines) + b"\n")
    return str(p)


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_rb_gather_and_line_splitter_under_sanitizers(rb_san, tmp_path, san):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="report_signal_unsafe=0")
    paf = _shuffled(tmp_path, 5)
    want = subprocess.run([rb_san["plain"], "regroup", paf], stdout=subprocess.PIPE).stdout
    assert want.count(b"\n") > 200
    for flags in ([], ["-q"], ["-l"]):
        one = subprocess.run([rb_san["plain"], "regroup", *flags, paf], stdout=subprocess.PIPE).stdout
        for n in (1, 3):
            pre = ["--gpus", str(n)] if n > 1 else []
            r = subprocess.run([rb_san[san], *pre, "regroup", *flags, paf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            _clean(r.stderr, f"{san} regroup {flags} n={n}")
            assert r.returncode == 0 and r.stdout == one, (san, flags, n, r.stderr[-400:])
            out = tmp_path / f"o_{san}_{n}.paf"
            with open(out, "wb") as f:  # (a regular file: workers pwrite at assigned offsets)
                r = subprocess.run([rb_san[san], *pre, "regroup", *flags, paf], stdout=f, stderr=subprocess.PIPE, env=env)
            _clean(r.stderr, f"{san} regroup to a file {flags} n={n}")
            assert r.returncode == 0 and open(out, "rb").read() == one
    # the whole fixture (2 MB of CIGAR text) through the threaded line splitter
    full = os.path.join(GOLDEN, "asm_small.paf")
    r = subprocess.run([rb_san[san], "--gpus", "2", "regroup", full], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    _clean(r.stderr, f"{san} regroup of the fixture")
    assert r.returncode == 0 and r.stdout == subprocess.run([rb_san["plain"], "regroup", full], stdout=subprocess.PIPE).stdout
