"""CIGAR text <-> packed ops on the device (k_text.hip) against the oracle's restatement of rust-htslib's
CigarString::try_from / Display (oracle/rb_oracle.c: rbo_parse_cigar, rbo_cigar_to_string) and against the fixture."""
import ctypes as C
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
OPS = "MIDNSHP=X"


def oracle_parse(oracle, s):
    L = oracle.lib()
    ops, n = C.POINTER(C.c_uint32)(), C.c_size_t()
    b = s if isinstance(s, bytes) else s.encode()
    rc = L.rbo_parse_cigar(b, C.c_size_t(len(b)), C.byref(ops), C.byref(n))
    if rc != 0:
        return None
    out = np.ctypeslib.as_array(ops, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
    L.rbo_free(ops)
    return out


def oracle_format(oracle, ops):
    L = oracle.lib()
    out = C.c_char_p()
    a = np.ascontiguousarray(ops, np.uint32)
    L.rbo_cigar_to_string.restype = C.c_size_t
    k = L.rbo_cigar_to_string(a.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(len(a)), C.byref(out))
    s = C.string_at(out, k)
    L.rbo_free(out)
    return s


def rand_cigar(rng, n_ops, big=False):
    parts = []
    for _ in range(n_ops):
        r = rng.random()
        if big and r < 0.02:
            ln = int(rng.integers(1 << 20, (1 << 28) - 1))
        elif r < 0.5:
            ln = int(rng.integers(1, 10))
        elif r < 0.9:
            ln = int(rng.integers(10, 5000))
        else:
            ln = int(rng.integers(5000, 3_000_000))
        parts.append(f"{ln}{OPS[int(rng.integers(0, 9))]}")
    return "".join(parts)


def test_parse_matches_oracle_random(engine, oracle):
    rng = np.random.default_rng(zlib.crc32(b"text-parse"))
    cigs = [rand_cigar(rng, int(n), big=True) for n in list(rng.integers(0, 40, 200)) + [1000, 5000, 257, 256, 255, 64, 63, 65]]
    cigs += ["", "1M", "0M", "268435455=", "12=1X3I", "4294967295M"[:0] + "5M"]
    op_off, ops, status = engine.parse_cigars(cigs)
    assert status.tolist() == [0] * len(cigs)
    for i, c in enumerate(cigs):
        want = oracle_parse(oracle, c)
        assert want is not None
        assert np.array_equal(ops[int(op_off[i]):int(op_off[i + 1])], want), (i, c[:60])


@pytest.mark.parametrize("shift", range(0, 17))
def test_parse_every_alignment(engine, oracle, shift):
    """strings that start at every byte phase of a 16-byte lane chunk, with numbers straddling lanes and 1 KiB steps"""
    rng = np.random.default_rng(1000 + shift)
    cigs = ["7M" * shift] + [rand_cigar(rng, int(n)) for n in (1, 3, 9, 130, 700, 2100)] + ["123456789=" * 120, "1M" * 1500]
    op_off, ops, status = engine.parse_cigars(cigs)
    assert not status.any()
    for i, c in enumerate(cigs):
        assert np.array_equal(ops[int(op_off[i]):int(op_off[i + 1])], oracle_parse(oracle, c)), (shift, i)


@pytest.mark.parametrize("bad", [
    "M", "1", "12", "1M2", "1MM", "MM", "1Q", "1m", "1M 2M", "-1M", "+1M", "1.5M", "4294967296M", "99999999999M", "1M\n",
    "10=" * 40 + "X", "10=" * 700 + "5", "12345678901234567=", "1=" * 8 + "=",
])
def test_parse_errors_like_the_reference(engine, oracle, bad):
    """everything CigarString::try_from rejects is the `.expect()` panic at paf.rs:399"""
    assert oracle_parse(oracle, bad) is None
    good = "5=2X"
    _, _, status = engine.parse_cigars([good, bad, good])
    # 1 = rejected on the device; 3 = "a number longer than a lane holds, ask the host's parser" (which rejects it)
    assert status[0] == 0 and status[2] == 0 and status[1] in (1, 3)


def test_parse_too_long_for_the_packed_form(engine):
    _, _, status = engine.parse_cigars(["268435456M", "3=268435456D1=", "268435455M"])
    assert status.tolist() == [2, 2, 0]


def test_parse_fixture_file(engine, oracle, golden):
    cigs = []
    for line in open(f"{golden}/asm_small.paf"):
        for tok in line.rstrip("\n").split("\t")[12:]:
            if tok.startswith("cg:Z:"):
                cigs.append(tok[5:])
    assert len(cigs) == 249
    op_off, ops, status = engine.parse_cigars(cigs)
    assert not status.any()
    for i in range(0, len(cigs), 7):
        assert np.array_equal(ops[int(op_off[i]):int(op_off[i + 1])], oracle_parse(oracle, cigs[i]))
    # and back: the device prints what it parsed
    first, count = op_off[:-1].copy(), np.diff(op_off).astype(np.uint32)
    toff, text = engine.format_cigars(ops, first, count)
    for i, c in enumerate(cigs):
        assert text[int(toff[i]):int(toff[i + 1])] == c.encode(), i


def test_format_matches_oracle_and_clips(engine, oracle):
    rng = np.random.default_rng(zlib.crc32(b"text-format"))
    lens = np.where(rng.random(6000) < 0.7, rng.integers(1, 300, 6000), rng.integers(1, 1 << 27, 6000)).astype(np.uint32)
    ops = ((lens << 4) | rng.integers(0, 9, 6000).astype(np.uint32)).astype(np.uint32)
    ops[10] = (9 << 4) | 7  # the one-op item below keeps 5 + 7 - 9 = 3 bases of it
    first = np.array([0, 0, 10, 10, 5000, 17, 300, 301, 5999, 42], np.uint64)
    count = np.array([6000, 1, 1, 700, 1000, 0, 257, 256, 1, 2], np.uint32)
    fl = np.array([0, 0, 5, 3, 0, 0, 9, 0, 0, 4], np.uint32)
    ll = np.array([0, 0, 7, 0, 8, 0, 2, 11, 0, 6], np.uint32)
    toff, text = engine.format_cigars(ops, first, count, fl, ll)
    for i in range(len(first)):
        seg = ops[int(first[i]):int(first[i]) + int(count[i])].copy()
        if len(seg) == 1 and fl[i] and ll[i]:
            seg[0] = ((int(fl[i]) + int(ll[i]) - (int(seg[0]) >> 4)) << 4) | (int(seg[0]) & 15)
        else:
            if len(seg) and fl[i]:
                seg[0] = (int(fl[i]) << 4) | (int(seg[0]) & 15)
            if len(seg) and ll[i]:
                seg[-1] = (int(ll[i]) << 4) | (int(seg[-1]) & 15)
        assert text[int(toff[i]):int(toff[i + 1])] == oracle_format(oracle, seg), i


def test_parse_zero_padded_numbers(engine, oracle):
    """u32::from_str accepts any number of leading zeros: paddings up to nine digits in all are parsed on the device, a number written
    with ten or more digits is handed to the host (status 3), never rejected"""
    cigs = ["000000005M", "0000000000000012=3X", "7=" + "0" * 9 + "4294967295D"[:0] + "000000000268435455D", "0" * 40 + "9M", "5=" + "0" * 20 + "1X"]
    op_off, ops, status = engine.parse_cigars(cigs)
    for i, c in enumerate(cigs):
        want = oracle_parse(oracle, c)
        assert want is not None
        assert status[i] in (0, 3), (c, int(status[i]))
        if status[i] == 0:
            assert np.array_equal(ops[int(op_off[i]):int(op_off[i + 1])], want), c
    assert status[0] == 0 and status[3] == 3


def test_format_digit_boundaries(engine, oracle):
    """every digit count and the values around each power of ten (the format kernel takes lengths below 10000 and the others on
    different routes), at every position of a lane's four ops and across 256-op steps"""
    edges = [1, 9, 10, 11, 99, 100, 101, 999, 1000, 1001, 9999, 10000, 10001, 43698, 43699, 99999, 100000, 999999, 1000000, 9999999,
             10000000, 99999999, 100000000, 100000001, 199999999, 200000000, 268435455]
    rng = np.random.default_rng(5)
    lens = np.array([edges[int(i)] for i in rng.integers(0, len(edges), 4000)] + edges * 3, np.uint32)
    ops = ((lens << 4) | rng.integers(0, 9, len(lens)).astype(np.uint32)).astype(np.uint32)
    first = np.array([0, 1, 2, 3, 255, 256, 257, len(ops) - len(edges)], np.uint64)
    count = np.array([len(ops), 4, 7, 300, 2, 1, 600, len(edges)], np.uint32)
    z = np.zeros(len(first), np.uint32)
    toff, text = engine.format_cigars(ops, first, count, z, z)
    for i in range(len(first)):
        assert text[int(toff[i]):int(toff[i + 1])] == oracle_format(oracle, ops[int(first[i]):int(first[i]) + int(count[i])]), i


def test_staged_transfers_keep_every_byte(engine):
    """rb_dev_upload / rb_dev_download go through a page-locked ring in 32 MB chunks copied by four host threads: sizes whose last
    chunk is 4 k + 1..3 bytes long (a quarter that is a multiple of 64 lost the remainder once) must arrive complete"""
    import ctypes as C
    L, ctx = engine.L, engine.ctx
    for n in (32 * (1 << 20) + 4 * 64 * 70000 + 3, 8 * (1 << 20) + 4 * 64 * 1000 + 1, 64 * (1 << 20) + 21063171):
        src = (np.arange(n, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
        d = C.c_void_p()
        assert L.rb_dev_alloc(ctx, C.c_size_t(n + 64), C.byref(d)) == 0
        assert L.rb_dev_upload(ctx, d, src.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == 0
        back = np.zeros(n, np.uint8)
        assert L.rb_dev_download(ctx, back.ctypes.data_as(C.c_void_p), d, C.c_size_t(n)) == 0
        assert np.array_equal(back[-4096:], src[-4096:]) and np.array_equal(back, src), n
        L.rb_dev_free(ctx, d)
