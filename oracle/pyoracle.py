"""ctypes front end of the CPU oracle (oracle/rb_oracle.c).  TEST INFRASTRUCTURE ONLY.

The oracle is a plain-C per-base restatement of the reference (see rb_oracle.h for what it
is pinned against and what is "parity unpinned").  Nothing here is imported by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "librb_oracle.so")
CLI = os.path.join(HERE, "rb_oracle")

MODERN, LEGACY = 0, 1

# statuses (rb_oracle.h)
OK = 0
NONE_INDEL, NONE_NOMATCH, NONE_EMPTY, NONE_INVERTED, NONE_INTEGRITY = 1, 2, 3, 4, 5
PANIC_NOTFOUND, PANIC_EMPTY_CIGAR, PANIC_INTEGRITY_T, PANIC_INTEGRITY_Q = 16, 17, 18, 19
PANIC_ALL_INDEL, PANIC_ASSERT, PANIC_OVERFLOW = 20, 21, 22

REDUCE_DT = np.dtype(
    [("t_bases", "<u8"), ("q_bases", "<u8"), ("nmatch", "<u4"), ("aln_len", "<u4"), ("equal", "<u4"),
     ("diff", "<u4"), ("ins", "<u4"), ("del", "<u4"), ("matches", "<u4"), ("ins_events", "<u4"),
     ("del_events", "<u4"), ("id_by_all", "<f4"), ("id_by_events", "<f4"), ("id_by_matches", "<f4"),
     ("status", "<u4"), ("_pad", "<u4")])
NORM_DT = np.dtype(
    [("t_st", "<u8"), ("t_en", "<u8"), ("q_st", "<u8"), ("q_en", "<u8"), ("first_op", "<u4"), ("n_ops", "<u4"),
     ("lead_ops", "<u4"), ("trail_ops", "<u4"), ("nmatch", "<u4"), ("aln_len", "<u4"), ("status", "<u4"),
     ("_pad", "<u4")])
HIT_DT = np.dtype(
    [("rec", "<u4"), ("win", "<u4"), ("status", "<u4"), ("flags", "<u4"), ("t_st", "<u8"), ("t_en", "<u8"),
     ("q_st", "<u8"), ("q_en", "<u8"), ("nmatch", "<u4"), ("aln_len", "<u4"), ("out_off", "<u8"),
     ("out_n", "<u4"), ("_pad", "<u4")])
PAIR_DT = np.dtype(
    [("split_idx", "<u8"), ("split_score", "<i4"), ("status", "<u4"), ("t_st", "<u8", 2), ("t_en", "<u8", 2),
     ("q_st", "<u8", 2), ("q_en", "<u8", 2), ("nmatch", "<u4", 2), ("aln_len", "<u4", 2), ("out_off", "<u8", 2),
     ("out_n", "<u4", 2)])
assert REDUCE_DT.itemsize == 72 and NORM_DT.itemsize == 64 and HIT_DT.itemsize == 72


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not (os.path.exists(LIB) and os.path.exists(CLI)) or \
            os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "rb_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.rbo_f32_display.restype = C.c_size_t
        _lib.rbo_f32_display.argtypes = [C.c_float, C.c_char_p, C.c_size_t]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _arr(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


class Batch:
    """Packed record batch (same data model as include/rustybam_amd.h)."""

    def __init__(self, ops, op_off, t_st, t_en, q_st, q_en, strand, contig=None):
        self.ops = _arr(ops, np.uint32)
        self.op_off = _arr(op_off, np.uint64)
        self.t_st, self.t_en = _arr(t_st, np.uint64), _arr(t_en, np.uint64)
        self.q_st, self.q_en = _arr(q_st, np.uint64), _arr(q_en, np.uint64)
        self.strand = _arr(strand, np.uint8)
        self.n = len(self.t_st)
        self.contig = _arr(contig if contig is not None else np.zeros(self.n), np.uint32)
        assert len(self.op_off) == self.n + 1

    def args(self, with_strand=True):
        a = [C.c_uint64(self.n), _p(self.ops), _p(self.op_off), _p(self.t_st), _p(self.t_en), _p(self.q_st),
             _p(self.q_en)]
        if with_strand:
            a.append(_p(self.strand))
        return a


def f32_display(v):
    buf = C.create_string_buffer(64)
    lib().rbo_f32_display(C.c_float(v), buf, 64)
    return buf.value.decode()


def reduce(b):
    out = np.zeros(b.n, dtype=REDUCE_DT)
    lib().rbo_reduce_arrays(*b.args(False), _p(out))
    return out


def normalize(b):
    out = np.zeros(b.n, dtype=NORM_DT)
    lib().rbo_normalize_arrays(*b.args(True), _p(out))
    return out


def _take(ptr, n, dt):
    if n == 0:
        out = np.zeros(0, dtype=dt)
    else:
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=dt).copy()
    lib().rbo_free(ptr)
    return out


def liftover(b, w_contig, w_st, w_en, policy=MODERN, n_threads=1):
    w_contig, w_st, w_en = _arr(w_contig, np.uint32), _arr(w_st, np.uint64), _arr(w_en, np.uint64)
    hits, ops = C.c_void_p(), C.c_void_p()
    nh, no = C.c_uint64(), C.c_uint64()
    lib().rbo_liftover_arrays(*b.args(True), _p(b.contig), C.c_uint64(len(w_st)), _p(w_contig), _p(w_st), _p(w_en),
                              C.c_int(policy), C.c_int(n_threads), C.byref(hits), C.byref(nh), C.byref(ops),
                              C.byref(no))
    return _take(hits, nh.value, HIT_DT), _take(ops, no.value, np.uint32)


def liftover_opspace(b, w_contig, w_st, w_en, n_threads=1):
    """the op-space CPU baseline (rb_opspace.c): same rows as liftover() for regular records, modern policy; None if unsupported"""
    w_contig, w_st, w_en = _arr(w_contig, np.uint32), _arr(w_st, np.uint64), _arr(w_en, np.uint64)
    hits, ops = C.c_void_p(), C.c_void_p()
    nh, no = C.c_uint64(), C.c_uint64()
    f = lib().rbo_liftover_opspace_arrays
    f.restype = C.c_int
    rc = f(*b.args(True), _p(b.contig), C.c_uint64(len(w_st)), _p(w_contig), _p(w_st), _p(w_en), C.c_int(n_threads), C.byref(hits),
           C.byref(nh), C.byref(ops), C.byref(no))
    if rc != 0:
        return None
    return _take(hits, nh.value, HIT_DT), _take(ops, no.value, np.uint32)


def break_opspace(b, max_size, n_threads=1):
    """break-paf in op space on the CPU (rb_opspace.c): same rows as break_paf() for regular records, modern policy; None if unsupported"""
    hits, ops = C.c_void_p(), C.c_void_p()
    nh, no = C.c_uint64(), C.c_uint64()
    f = lib().rbo_break_opspace_arrays
    f.restype = C.c_int
    rc = f(*b.args(True), C.c_uint32(max_size), C.c_int(n_threads), C.byref(hits), C.byref(nh), C.byref(ops), C.byref(no))
    if rc != 0:
        return None
    return _take(hits, nh.value, HIT_DT), _take(ops, no.value, np.uint32)


def break_paf(b, max_size, policy=MODERN, n_threads=1):
    hits, ops = C.c_void_p(), C.c_void_p()
    nh, no = C.c_uint64(), C.c_uint64()
    lib().rbo_break_arrays(*b.args(True), C.c_uint32(max_size), C.c_int(policy), C.c_int(n_threads),
                           C.byref(hits), C.byref(nh), C.byref(ops), C.byref(no))
    return _take(hits, nh.value, HIT_DT), _take(ops, no.value, np.uint32)


def overlap_split(b, left, right, scores=(1, 1, 1), policy=MODERN):
    left, right = _arr(left, np.uint32), _arr(right, np.uint32)
    rows = np.zeros(len(left), dtype=PAIR_DT)
    ops, no = C.c_void_p(), C.c_uint64()
    lib().rbo_overlap_split_arrays(*b.args(True), C.c_uint64(len(left)), _p(left), _p(right), C.c_int(scores[0]),
                                   C.c_int(scores[1]), C.c_int(scores[2]), C.c_int(policy), _p(rows),
                                   C.byref(ops), C.byref(no))
    return rows, _take(ops, no.value, np.uint32)


PAIR_CLIP_DT = np.dtype([("split_idx", "<u8"), ("split_score", "<i4"), ("status", "<u4"), ("t_st", "<u8", (2,)), ("t_en", "<u8", (2,)),
                         ("q_st", "<u8", (2,)), ("q_en", "<u8", (2,)), ("nmatch", "<u4", (2,)), ("aln_len", "<u4", (2,)), ("first", "<u4", (2,)),
                         ("count", "<u4", (2,)), ("first_len", "<u4", (2,)), ("last_len", "<u4", (2,))])
PAIR_UNSUPPORTED = 0xFFFFFFFF


def overlap_split_opspace(ops, rec_off, rec_n, first_len, last_len, t_st, t_en, q_st, q_en, strand, left, right, scores=(1, 1, 1), n_threads=1):
    """trim-paf's pair step in op space on record VIEWS (rb_opspace.c): the cut of every pair as a view (first kept op, count, new end
    lengths) with its coordinates, nmatch, aln_len.  Regular records, modern policy: (rows, number of pairs outside that scope)."""
    a32, a64 = (lambda x: _arr(x, np.uint32)), (lambda x: _arr(x, np.uint64))
    ops, rec_off, rec_n, first_len, last_len = a32(ops), a64(rec_off), a32(rec_n), a32(first_len), a32(last_len)
    t_st, t_en, q_st, q_en, strand, left, right = a64(t_st), a64(t_en), a64(q_st), a64(q_en), _arr(strand, np.uint8), a32(left), a32(right)
    rows = np.zeros(len(left), dtype=PAIR_CLIP_DT)
    assert PAIR_CLIP_DT.itemsize == 128  # (sizeof(rbo_pair_clip_row))
    f = lib().rbo_overlap_split_opspace_arrays
    f.restype = C.c_int64
    bad = f(C.c_uint64(len(rec_off)), _p(ops), _p(rec_off), _p(rec_n), _p(first_len), _p(last_len), _p(t_st), _p(t_en), _p(q_st), _p(q_en), _p(strand),
            C.c_uint64(len(left)), _p(left), _p(right), C.c_int(scores[0]), C.c_int(scores[1]), C.c_int(scores[2]), C.c_int(n_threads), _p(rows))
    return rows, int(bad)


def swap(b):
    out = np.zeros_like(b.ops)
    lib().rbo_swap_arrays(C.c_uint64(b.n), _p(b.ops), _p(b.op_off), _p(b.strand), _p(out))
    return out


def nucfreq(tid, pos, flag, op_off, ops, l_seq, seq_off, seq, rtid, st, en):
    """One fetch + pileup (nucfreq.rs:111-125, :61-95): (positions, counts [n, 4]) of the covered positions of [st, en)."""
    tid, pos, flag = _arr(tid, np.int32), _arr(pos, np.int64), _arr(flag, np.uint32)
    op_off, ops = _arr(op_off, np.uint64), _arr(ops, np.uint32)
    l_seq, seq_off, seq = _arr(l_seq, np.uint32), _arr(seq_off, np.uint64), _arr(seq, np.uint8)
    cap = max(int(en) - int(st), 1)
    out_pos = np.zeros(cap, np.uint32)
    out_cnt = np.zeros((cap, 4), np.uint64)
    f = lib().rbo_nucfreq_arrays
    f.restype = C.c_int64
    n = f(C.c_uint64(len(tid)), _p(tid), _p(pos), _p(flag), _p(op_off), _p(ops), _p(l_seq), _p(seq_off), _p(seq), C.c_int32(rtid),
          C.c_uint64(st), C.c_uint64(en), _p(out_pos), _p(out_cnt), C.c_uint64(cap))
    if n < 0:
        return int(n), None, None
    return 0, out_pos[:n].copy(), out_cnt[:n].copy()


def cli(*args, stdin=None):
    """Run the oracle CLI; returns (returncode, stdout bytes)."""
    build()
    r = subprocess.run([CLI, *map(str, args)], input=stdin, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    return r.returncode, r.stdout
