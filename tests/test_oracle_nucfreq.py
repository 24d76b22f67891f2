"""CPU: the oracle's pileup restatement (htslib bam_plp / resolve_cigar2 behind nucfreq.rs:61-95) against the
reference's own known answer (KA13, nucfreq.rs:41-60), hand-worked cases, and an independent read-major model."""
import os

import numpy as np
import pytest

from nf_util import Reads, read_bam, random_reads, REF_OPS, QRY_OPS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ka13_test_nucfreq_bam(oracle):
    """nucfreq.rs:41-60: CHROMOSOME_I [1, 102): the largest count at every position is 0 or 2"""
    names, lens, rd = read_bam(f"{GOLD}/test_nucfreq.bam")
    assert names[0] == "CHROMOSOME_I"
    rc, pos, cnt = oracle.nucfreq(*rd.args(), 0, 1, 102)
    assert rc == 0 and len(pos) > 50
    mx = cnt.max(axis=1)
    assert set(mx.tolist()) <= {0, 2} and (mx == 2).sum() >= 50


def test_ka13_cli_text(oracle):
    rc, out = oracle.cli("nucfreq", "-r", "CHROMOSOME_I:2-102", f"{GOLD}/test_nucfreq.bam")
    assert rc == 0
    lines = out.decode().splitlines()
    assert lines[0] == "#chr\tstart\tend\tA\tC\tG\tT\tregion_id"
    f = lines[1].split("\t")
    assert f[0] == "CHROMOSOME_I" and int(f[2]) == int(f[1]) + 1 and f[7] == "CHROMOSOME_I:2-102"
    rc, small = oracle.cli("nucfreq", "-s", "-r", "CHROMOSOME_I:2-102", f"{GOLD}/test_nucfreq.bam")
    s = small.decode().splitlines()
    assert s[0] == "#CHROMOSOME_I\t%s\tCHROMOSOME_I:2-102" % f[1] and len(s) == len(lines)
    assert all(x.split("\t")[0] in ("0", "2") for x in s[1:])


def _one(cigar, seq, pos=10, flag=0):
    return Reads([0], [pos], [flag], [[(l << 4) | "MIDNSHP=X".index(c) for l, c in cigar]], [seq])


def test_hand_worked_deletion_insertion_softclip(oracle):
    # 2S 3M 1I 2M 2D 2M at pos 10: bases A C | G T A | (C) | G T | -- | A C
    A, Cc, G, T = 1, 2, 4, 8
    rd = _one([(2, "S"), (3, "M"), (1, "I"), (2, "M"), (2, "D"), (2, "M")], [A, Cc, G, T, A, Cc, G, T, A, Cc])
    rc, pos, cnt = oracle.nucfreq(*rd.args(), 0, 0, 100)
    assert rc == 0
    assert pos.tolist() == list(range(10, 19))  # 3 + 2 + 2 (deleted, still reported) + 2
    col = {1: 0, 2: 1, 4: 2, 8: 3}
    exp = [G, T, A, G, T, None, None, A, Cc]
    for k, e in enumerate(exp):
        want = [0, 0, 0, 0]
        if e is not None:
            want[col[e]] = 1
        assert cnt[k].tolist() == want, k


def test_filtered_flags_and_refskip(oracle):
    A = 1
    for flag, kept in [(0, True), (16, True), (2048, True), (4, False), (256, False), (512, False), (1024, False)]:
        rd = _one([(3, "M"), (5, "N"), (2, "M")], [A] * 5, flag=flag)
        rc, pos, cnt = oracle.nucfreq(*rd.args(), 0, 0, 100)
        assert rc == 0
        if kept:
            assert pos.tolist() == list(range(10, 20)) and cnt[:, 0].tolist() == [1, 1, 1, 0, 0, 0, 0, 0, 1, 1]
        else:
            assert len(pos) == 0


def test_region_clips_and_fetch_window(oracle):
    rd = _one([(10, "M")], [2] * 10)
    rc, pos, cnt = oracle.nucfreq(*rd.args(), 0, 12, 15)
    assert pos.tolist() == [12, 13, 14] and cnt[:, 1].tolist() == [1, 1, 1]
    rc, pos, cnt = oracle.nucfreq(*rd.args(), 0, 20, 30)  # endpos 20 is not > 20: not fetched
    assert rc == 0 and len(pos) == 0
    rc, pos, cnt = oracle.nucfreq(*rd.args(), 1, 0, 30)   # other contig
    assert rc == 0 and len(pos) == 0


def test_sequence_shorter_than_cigar_is_a_panic(oracle):
    rd = Reads([0], [10], [0], [[(10 << 4) | 0]], [[1] * 6])
    rc, _, _ = oracle.nucfreq(*rd.args(), 0, 0, 14)
    assert rc == 0  # the missing bases are not reached inside [0, 14)
    rc, _, _ = oracle.nucfreq(*rd.args(), 0, 0, 30)
    assert rc == -4


def test_unsorted_reads_error(oracle):
    rd = Reads([0, 0], [50, 10], [0, 0], [[(10 << 4)], [(10 << 4)]], [[1] * 10, [1] * 10])
    rc, _, _ = oracle.nucfreq(*rd.args(), 0, 0, 100)
    assert rc == -1


def test_parse_region(oracle):
    import ctypes as C
    L = oracle.lib()

    class Rg(C.Structure):
        _fields_ = [("name", C.c_char_p), ("st", C.c_uint64), ("en", C.c_uint64), ("id", C.c_char_p)]
    for s, want in [("chr1:1-1000", ("chr1", 0, 1000, "chr1:1-1000")), ("chr1:2-2000:1-1000", ("chr1:2-2000", 0, 1000, "chr1:2-2000:1-1000")),
                    ("c:5-99999999999999999999999", ("c", 4, 4294967295, "c:5-4294967295"))]:
        r = Rg()
        assert L.rbo_parse_region(s.encode(), C.byref(r)) == 0
        assert (r.name.decode(), r.st, r.en, r.id.decode()) == want   # bed.rs:88-97
    r = Rg()
    assert L.rbo_parse_region(b"chr1", C.byref(r)) != 0 and L.rbo_parse_region(b"chr1:0-5", C.byref(r)) != 0


def _model(rd, tid, st, en):
    """read-major model: every read scatters its match-type bases; coverage = union of [pos, end)"""
    cov = np.zeros(en - st, bool)
    cnt = np.zeros((en - st, 4), np.uint64)
    col = {1: 0, 2: 1, 4: 2, 8: 3}
    for i in range(rd.n):
        if rd.tid[i] != tid or (int(rd.flag[i]) & 0x704):
            continue
        ops = rd.ops[int(rd.op_off[i]):int(rd.op_off[i + 1])]
        r, q = int(rd.pos[i]), 0
        base = int(rd.seq_off[i])
        for w in ops.tolist():
            o, l = w & 15, w >> 4
            if o in (0, 7, 8):
                for k in range(l):
                    if st <= r + k < en:
                        b = int(rd.seq[base + ((q + k) >> 1)])
                        nib = (b >> 4) if ((q + k) & 1) == 0 else (b & 15)
                        if nib in col:
                            cnt[r + k - st, col[nib]] += 1
            if o in REF_OPS:
                a, b_ = max(r, st), min(r + l, en)
                if b_ > a:
                    cov[a - st:b_ - st] = True
                r += l
            if o in QRY_OPS:
                q += l
    return cov, cnt


@pytest.mark.parametrize("seed", range(6))
def test_pileup_restatement_equals_read_major_model(oracle, seed):
    from nf_util import oracle_region
    rng = np.random.default_rng(1000 + seed)
    rd = random_reads(rng, 60, n_contig=2, span=3000, long_frac=0.05, max_ops=12)
    for tid in (0, 1):
        st = int(rng.integers(0, 2000))
        en = st + int(rng.integers(1, 6000))
        cov, cnt = oracle_region(oracle, rd, tid, st, en, piece=1000)
        mcov, mcnt = _model(rd, tid, st, en)
        assert np.array_equal(cov, mcov)
        assert np.array_equal(cnt, mcnt)
