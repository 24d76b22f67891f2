"""ctypes binding of include/rustybam_amd.h (plumbing for tests and bench.py)."""
import ctypes as C
import os
import sys
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "rustybam_amd.h")

BSEARCH_MODERN, BSEARCH_LEGACY, LIFT_EARLY_EXIT, LIFT_DESCRIPTORS, LIFT_FUSED_SCAN = 0, 1, 16, 32, 64
TRIM_IN_PLACE = 256  # rb_dev_overlap_split on a resident batch (out_ops = the batch's ops): regular records are cut where they are, not copied
LIFT_OP_STARTS = 1 << 20  # rb_dev_liftover / rb_dev_break on a batch trim-paf has cut in place: op_off is a table of starts, extents come from the norm rows
BREAK_ONE_WALK = 128  # rb_dev_break: the clip kernel finds the long indels itself; look at counters redo_two_walk afterwards
HIT_INSIDE, HIT_GENERIC, HIT_DESCRIPTOR = 1, 2, 4
NF_COVERED = 0x80000000
RD_OK, RD_FILTERED, RD_BAD_CIGAR, RD_SEQ_SHORT = 0, 1, 2, 3

REDUCE_DT = np.dtype(
    [("t_bases", "<u8"), ("q_bases", "<u8"), ("nmatch", "<u4"), ("aln_len", "<u4"), ("equal", "<u4"),
     ("diff", "<u4"), ("ins", "<u4"), ("del", "<u4"), ("matches", "<u4"), ("ins_events", "<u4"),
     ("del_events", "<u4"), ("id_by_all", "<f4"), ("id_by_events", "<f4"), ("id_by_matches", "<f4"),
     ("status", "<u4"), ("flags", "<u4")])
NORM_DT = np.dtype(
    [("t_st", "<u8"), ("t_en", "<u8"), ("q_st", "<u8"), ("q_en", "<u8"), ("first_op", "<u4"), ("n_ops", "<u4"),
     ("lead_ops", "<u4"), ("trail_ops", "<u4"), ("nmatch", "<u4"), ("aln_len", "<u4"), ("status", "<u4"),
     ("flags", "<u4")])
HIT_DT = np.dtype(
    [("rec", "<u4"), ("win", "<u4"), ("status", "<u2"), ("flags", "<u2"), ("out_n", "<u4"), ("t_st", "<u8"),
     ("t_en", "<u8"), ("q_st", "<u8"), ("q_en", "<u8"), ("nmatch", "<u4"), ("aln_len", "<u4"), ("out_off", "<u8")])
PAIR_DT = np.dtype(
    [("split_idx", "<u8"), ("split_score", "<i4"), ("status", "<u4"), ("t_st", "<u8", 2), ("t_en", "<u8", 2),
     ("q_st", "<u8", 2), ("q_en", "<u8", 2), ("nmatch", "<u4", 2), ("aln_len", "<u4", 2), ("out_off", "<u8", 2),
     ("out_n", "<u4", 2), ("_pad", "<u8")])
TRIM_PASS_DT = np.dtype([("n_pairs", "<u8"), ("n_deferred", "<u8"), ("ops_end", "<u8"), ("bad_status", "<u4"), ("_pad", "<u4"), ("_reserved", "<u8", 4)])
COUNTERS_DT = np.dtype(
    [("n_hits", "<u8"), ("out_ops_needed", "<u8"), ("out_ops_used", "<u8"), ("n_generic", "<u8"),
     ("overflow", "<u4"), ("phase", "<u4", 5), ("brk_scratch_short", "<u4"), ("redo_two_walk", "<u4")])
assert REDUCE_DT.itemsize == 72 and NORM_DT.itemsize == 64 and HIT_DT.itemsize == 64 and COUNTERS_DT.itemsize == 64
assert PAIR_DT.itemsize == 128


def hit_rows_from(rows):
    """Rows of another layout with the same field names (the oracle's 72-byte rows) as rb_hit_row records (HIT_DT, 64 bytes): what
    rb_dev_digest_rows reads.  flags keeps bit 0 only (HIT_INSIDE)."""
    out = np.zeros(len(rows), dtype=HIT_DT)
    for k in ("rec", "win", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_off"):
        out[k] = rows[k]
    out["status"] = rows["status"].astype(np.uint16)
    out["flags"] = (rows["flags"] & 1).astype(np.uint16)
    return out


class RbError(RuntimeError):
    pass


def lib_path():
    """the product library; RB_VARIANT=<name> (diagnostics only: tools/mkvariant.sh builds rustybam_amd/variants/<name>.so, some of them
    timing builds with wrong results) loads that one instead and says so -- the A/B scripts select a variant this way and never copy
    one over the product library"""
    v = os.environ.get("RB_VARIANT")
    if v:
        p = os.path.join(HERE, "variants", v + ".so")
        print(f"[rustybam_amd] RB_VARIANT: loading {p} instead of the product library", file=sys.stderr)
        return p
    return os.path.join(HERE, "librustybam_amd.so")


_lib = None


def lib():
    """Load the product library.  Raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RbError(f"{p} is missing: run `make -C rustybam_amd/csrc` (or __graft_entry__.build())")
        _lib = C.CDLL(p)
        _lib.rb_ctx_last_error.restype = C.c_char_p
        _lib.rb_ctx_stream.restype = C.c_void_p
        _lib.rb_plan_workspace_bytes.restype = C.c_size_t
        _lib.rb_plan_diag_stamps_offset.restype = C.c_size_t
        _lib.rb_plan_out_capacity.restype = C.c_uint64
        _lib.rb_synth_n_ops.restype = C.c_uint32
        _lib.rb_synth_n_ops.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    return _lib


def declared_symbols():
    """Every function name declared in include/rustybam_amd.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rb_[a-z0-9_]+)\s*\(", txt)))


def exported_symbols():
    out = []
    L = lib()
    for s in declared_symbols():
        try:
            getattr(L, s)
            out.append(s)
        except AttributeError:
            pass
    return out


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _arr(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


class BatchView(C.Structure):
    _fields_ = [("n_rec", C.c_uint64), ("n_ops", C.c_uint64), ("ops", C.c_void_p), ("op_off", C.c_void_p),
                ("t_st", C.c_void_p), ("t_en", C.c_void_p), ("q_st", C.c_void_p), ("q_en", C.c_void_p),
                ("strand", C.c_void_p), ("contig", C.c_void_p)]


class Engine:
    """One rb_ctx.  `stream` is a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or None."""

    def __init__(self, device=0, stream=None):
        self.L = lib()
        self.ctx = C.c_void_p()
        rc = self.L.rb_ctx_create(C.c_int(device), C.c_void_p(stream or 0), C.byref(self.ctx))
        if rc != 0:
            raise RbError(f"rb_ctx_create(device={device}) failed with {rc}: no usable gfx950 device "
                          f"(there is no CPU fallback)")

    def close(self):
        if self.ctx:
            self.L.rb_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RbError(f"{what} failed with {rc}: {self.L.rb_ctx_last_error(self.ctx).decode()}")

    def sync(self):
        self._chk(self.L.rb_ctx_sync(self.ctx), "rb_ctx_sync")

    # ---- device-pointer level (bench.py: buffers are torch tensors already resident in HBM) ----
    def trim_reserve(self, n_pairs):
        self._chk(self.L.rb_dev_trim_reserve(self.ctx, C.c_uint64(int(n_pairs))), "rb_dev_trim_reserve")

    def set_timing(self, on=True):
        self._chk(self.L.rb_ctx_set_timing(self.ctx, C.c_int(1 if on else 0)), "rb_ctx_set_timing")

    def get_timing(self):
        buf = (C.c_double * 256)()
        n = C.c_int()
        self._chk(self.L.rb_ctx_get_timing(self.ctx, buf, C.c_int(256), C.byref(n)), "rb_ctx_get_timing")
        return [buf[i] for i in range(n.value)]

    @staticmethod
    def batch_view(n_rec, n_ops, ops, op_off, t_st, t_en, q_st, q_en, strand, contig):
        """All arguments after n_ops are raw device addresses (ints)."""
        return BatchView(n_rec, n_ops, ops, op_off, t_st, t_en, q_st, q_en, strand, contig)

    def dev_synth_fill_ops(self, seed, first_record, n_rec, op_off_ptr, ops_ptr):
        self._chk(self.L.rb_dev_synth_fill_ops(self.ctx, C.c_uint64(seed), C.c_uint64(first_record), C.c_uint64(n_rec),
                                               C.c_void_p(op_off_ptr), C.c_void_p(ops_ptr)), "rb_dev_synth_fill_ops")

    def dev_scan_records(self, view, reduce_ptr, norm_ptr):
        self._chk(self.L.rb_dev_scan_records(self.ctx, C.byref(view), C.c_void_p(reduce_ptr or 0),
                                             C.c_void_p(norm_ptr or 0)), "rb_dev_scan_records")

    def dev_digest_rows(self, view, rows_ptr, n_rows, out_ptr, row_base, rec_base, digest_ptr):
        self._chk(self.L.rb_dev_digest_rows(self.ctx, C.byref(view), C.c_void_p(rows_ptr), C.c_uint64(n_rows), C.c_void_p(out_ptr),
                                            C.c_uint64(row_base), C.c_uint64(rec_base), C.c_void_p(digest_ptr)), "rb_dev_digest_rows")

    def dev_overlap_split(self, view, norm_ptr, n_pairs, left_ptr, right_ptr, pair_out_off_ptr, scores, policy, rows_ptr, out_ptr):
        self._chk(self.L.rb_dev_overlap_split(self.ctx, C.byref(view), C.c_void_p(norm_ptr), C.c_uint64(n_pairs), C.c_void_p(left_ptr),
                                              C.c_void_p(right_ptr), C.c_void_p(pair_out_off_ptr), C.c_int(scores[0]), C.c_int(scores[1]),
                                              C.c_int(scores[2]), C.c_int(policy), C.c_void_p(rows_ptr), C.c_void_p(out_ptr)), "rb_dev_overlap_split")

    def dev_apply_pairs(self, n_pairs, left_ptr, right_ptr, rows_ptr, op_off_ptr, norm_ptr):
        self._chk(self.L.rb_dev_apply_pairs(self.ctx, C.c_uint64(n_pairs), C.c_void_p(left_ptr), C.c_void_p(right_ptr), C.c_void_p(rows_ptr),
                                            C.c_void_p(op_off_ptr), C.c_void_p(norm_ptr)), "rb_dev_apply_pairs")

    def trim_select_scratch_bytes(self, n_groups):
        f = self.L.rb_trim_select_scratch_bytes
        f.restype = C.c_size_t
        return int(f(C.c_uint64(n_groups)))

    def dev_trim_select(self, n_rec, n_groups, order_ptr, grp_off_ptr, norm_ptr, out_base, contained_ptr, left_ptr, right_ptr, pair_out_off_ptr,
                        pass_ptr, scratch_ptr):
        self._chk(self.L.rb_dev_trim_select(self.ctx, C.c_uint64(n_rec), C.c_uint64(n_groups), C.c_void_p(order_ptr), C.c_void_p(grp_off_ptr),
                                            C.c_void_p(norm_ptr), C.c_uint64(out_base), C.c_void_p(contained_ptr), C.c_void_p(left_ptr),
                                            C.c_void_p(right_ptr), C.c_void_p(pair_out_off_ptr), C.c_void_p(pass_ptr), C.c_void_p(scratch_ptr)),
                  "rb_dev_trim_select")

    def dev_trim_check(self, n_pairs, rows_ptr, pass_ptr):
        self._chk(self.L.rb_dev_trim_check(self.ctx, C.c_uint64(n_pairs), C.c_void_p(rows_ptr), C.c_void_p(pass_ptr)), "rb_dev_trim_check")

    def dev_gather_records(self, n_rec, ops_ptr, op_off_ptr, norm_ptr, new_off_ptr, new_ops_ptr, scratch_ptr):
        self._chk(self.L.rb_dev_gather_records(self.ctx, C.c_uint64(n_rec), C.c_void_p(ops_ptr), C.c_void_p(op_off_ptr), C.c_void_p(norm_ptr),
                                               C.c_void_p(new_off_ptr), C.c_void_p(new_ops_ptr or 0), C.c_void_p(scratch_ptr)), "rb_dev_gather_records")

    def dev_alloc(self, n_bytes):
        """device memory from the library's allocator (requests of 256 MB and more: 2 MB physical chunks mapped side by side); -> address"""
        d = C.c_void_p()
        self._chk(self.L.rb_dev_alloc(self.ctx, C.c_size_t(n_bytes), C.byref(d)), "rb_dev_alloc")
        return int(d.value)

    SCORE_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p)

    def dev_alloc_placed(self, n_bytes, tries, score=None):
        """a buffer that will be written at streaming rate, placed by measurement (rb_dev_alloc_placed: up to `tries` candidates, a store
        sweep over each -- or score(address) -> time, the caller's own launch (rb_dev_alloc_placed_by) --, the fastest kept);
        -> (address, [score per candidate], index kept)"""
        d, kept = C.c_void_p(), C.c_int(-1)
        ms = (C.c_double * max(1, tries))()
        if score is None:
            self._chk(self.L.rb_dev_alloc_placed(self.ctx, C.c_uint64(n_bytes), C.c_int(tries), C.byref(d), ms, C.byref(kept)), "rb_dev_alloc_placed")
        else:
            err = []

            def cb(ptr, _user):
                try:
                    return float(score(int(ptr)))
                except Exception as e:  # (no exception may cross the C frames: the search ends, the caller hears about it below)
                    err.append(e)
                    return -1.0
            fn = self.SCORE_CB(cb)
            rc = self.L.rb_dev_alloc_placed_by(self.ctx, C.c_uint64(n_bytes), C.c_int(tries), fn, None, C.byref(d), ms, C.byref(kept))
            if err:
                raise err[0]
            self._chk(rc, "rb_dev_alloc_placed_by")
        return int(d.value), [round(ms[i], 4) for i in range(tries) if ms[i] >= 0], int(kept.value)

    def dev_free(self, ptr):
        self._chk(self.L.rb_dev_free(self.ctx, C.c_void_p(ptr)), "rb_dev_free")

    def dev_release(self, ptr):
        """back to the context's cache: still mapped, handed out again by the next dev_alloc / dev_alloc_placed of the same size"""
        self._chk(self.L.rb_dev_release(self.ctx, C.c_void_p(ptr)), "rb_dev_release")

    def dev_cache_trim(self, keep_bytes=0):
        self._chk(self.L.rb_dev_cache_trim(self.ctx, C.c_uint64(keep_bytes)), "rb_dev_cache_trim")

    def dev_alloc_stats(self):
        """-> dict(live, cached, retired_va, va_cap, fallbacks): rb_dev_alloc_stats"""
        o = (C.c_uint64 * 5)()
        self._chk(self.L.rb_dev_alloc_stats(self.ctx, o), "rb_dev_alloc_stats")
        return dict(live=int(o[0]), cached=int(o[1]), retired_va=int(o[2]), va_cap=int(o[3]), fallbacks=int(o[4]))

    def text_scratch_bytes(self, n):
        f = self.L.rb_text_scratch_bytes
        f.restype = C.c_size_t
        return int(f(C.c_uint64(n)))

    def plan_create(self, op_off, contig, w_contig=None, w_st=None, w_en=None):
        op_off, contig = _arr(op_off, np.uint64), _arr(contig, np.uint32)
        nw = 0 if w_st is None else len(w_st)
        wc = _arr(w_contig if nw else np.zeros(0), np.uint32)
        ws = _arr(w_st if nw else np.zeros(0), np.uint64)
        we = _arr(w_en if nw else np.zeros(0), np.uint64)
        plan = C.c_void_p()
        self._chk(self.L.rb_plan_create(self.ctx, C.c_uint64(len(contig)), _p(op_off), _p(contig), C.c_uint64(nw),
                                        _p(wc), _p(ws), _p(we), C.byref(plan)), "rb_plan_create")
        return plan

    def plan_destroy(self, plan):
        self.L.rb_plan_destroy(plan)

    def plan_out_capacity(self, plan, for_break=False):
        return int(self.L.rb_plan_out_capacity(plan, C.c_int(1 if for_break else 0)))

    def plan_workspace_bytes(self, plan, rows_cap):
        return int(self.L.rb_plan_workspace_bytes(plan, C.c_uint64(rows_cap)))

    def dev_box_probe(self, src_ptr, src_bytes, dst0_ptr, dst1_ptr, reps=10, scatter=False):
        """-> (ms per launch, shader MHz held): the clip kernel's memory mix without its instructions (diagnostics)."""
        ms, mhz = C.c_double(0), C.c_double(0)
        self._chk(self.L.rb_dev_box_probe(self.ctx, C.c_void_p(src_ptr), C.c_uint64(src_bytes), C.c_void_p(dst0_ptr), C.c_void_p(dst1_ptr),
                                          C.c_int(reps), C.c_int(int(scatter)), C.byref(ms), C.byref(mhz)), "rb_dev_box_probe")
        return ms.value, mhz.value

    def plan_diag_stamps_offset(self, plan, rows_cap):
        return int(self.L.rb_plan_diag_stamps_offset(plan, C.c_uint64(rows_cap)))

    def dev_liftover(self, plan, view, norm_ptr, policy, ws_ptr, rows_ptr, rows_cap, out_ptr, out_cap, counters_ptr):
        self._chk(self.L.rb_dev_liftover(self.ctx, plan, C.byref(view), C.c_void_p(norm_ptr), C.c_int(policy),
                                         C.c_void_p(ws_ptr), C.c_void_p(rows_ptr), C.c_uint64(rows_cap),
                                         C.c_void_p(out_ptr), C.c_uint64(out_cap), C.c_void_p(counters_ptr)),
                  "rb_dev_liftover")

    def dev_break(self, plan, view, norm_ptr, max_size, policy, ws_ptr, rows_ptr, rows_cap, out_ptr, out_cap,
                  counters_ptr):
        self._chk(self.L.rb_dev_break(self.ctx, plan, C.byref(view), C.c_void_p(norm_ptr), C.c_uint32(max_size),
                                      C.c_int(policy), C.c_void_p(ws_ptr), C.c_void_p(rows_ptr), C.c_uint64(rows_cap),
                                      C.c_void_p(out_ptr), C.c_uint64(out_cap), C.c_void_p(counters_ptr)),
                  "rb_dev_break")

    # ---- host-buffer wrappers ----
    def scan_records(self, ops, op_off, t_st, t_en, q_st, q_en, strand):
        ops, op_off = _arr(ops, np.uint32), _arr(op_off, np.uint64)
        n = len(op_off) - 1
        ops = np.concatenate([ops, np.zeros(4, np.uint32)])
        red, norm = np.zeros(n, REDUCE_DT), np.zeros(n, NORM_DT)
        a = [_arr(x, np.uint64) for x in (t_st, t_en, q_st, q_en)]
        s = _arr(strand, np.uint8)
        self._chk(self.L.rb_host_scan_records(self.ctx, C.c_uint64(n), _p(ops), _p(op_off), *map(_p, a), _p(s),
                                              _p(red), _p(norm)), "rb_host_scan_records")
        return red, norm

    def _lift(self, fn, what, n, args):
        rows, out = C.c_void_p(), C.c_void_p()
        nr, no = C.c_uint64(), C.c_uint64()
        norm = np.zeros(n, NORM_DT)
        cnt = np.zeros(1, COUNTERS_DT)
        self._chk(fn(self.ctx, *args, _p(norm), C.byref(rows), C.byref(nr), C.byref(out), C.byref(no), _p(cnt)), what)

        def take(ptr, k, dt):
            if k == 0:
                r = np.zeros(0, dt)
            else:
                buf = (C.c_char * (k * np.dtype(dt).itemsize)).from_address(ptr.value)
                r = np.frombuffer(buf, dtype=dt).copy()
            self.L.rb_host_free(ptr)
            return r
        return take(rows, nr.value, HIT_DT), take(out, no.value, np.uint32), norm, cnt[0]

    def liftover(self, ops, op_off, t_st, t_en, q_st, q_en, strand, contig, w_contig, w_st, w_en,
                 policy=BSEARCH_MODERN):
        ops, op_off = _arr(ops, np.uint32), _arr(op_off, np.uint64)
        n = len(op_off) - 1
        ops = np.concatenate([ops, np.zeros(4, np.uint32)])
        a = [_arr(x, np.uint64) for x in (t_st, t_en, q_st, q_en)]
        s, c = _arr(strand, np.uint8), _arr(contig, np.uint32)
        wc, ws, we = _arr(w_contig, np.uint32), _arr(w_st, np.uint64), _arr(w_en, np.uint64)
        args = [C.c_uint64(n), _p(ops), _p(op_off), *map(_p, a), _p(s), _p(c), C.c_uint64(len(ws)), _p(wc), _p(ws),
                _p(we), C.c_int(policy)]
        return self._lift(self.L.rb_host_liftover, "rb_host_liftover", n, args)

    def break_paf(self, ops, op_off, t_st, t_en, q_st, q_en, strand, max_size, policy=BSEARCH_MODERN):
        ops, op_off = _arr(ops, np.uint32), _arr(op_off, np.uint64)
        n = len(op_off) - 1
        ops = np.concatenate([ops, np.zeros(4, np.uint32)])
        a = [_arr(x, np.uint64) for x in (t_st, t_en, q_st, q_en)]
        s = _arr(strand, np.uint8)
        args = [C.c_uint64(n), _p(ops), _p(op_off), *map(_p, a), _p(s), C.c_uint32(max_size), C.c_int(policy)]
        return self._lift(self.L.rb_host_break, "rb_host_break", n, args)

    def overlap_split(self, ops, op_off, t_st, t_en, q_st, q_en, strand, left, right, scores=(1, 1, 1),
                      policy=BSEARCH_MODERN):
        ops, op_off = _arr(ops, np.uint32), _arr(op_off, np.uint64)
        n = len(op_off) - 1
        ops = np.concatenate([ops, np.zeros(4, np.uint32)])
        a = [_arr(x, np.uint64) for x in (t_st, t_en, q_st, q_en)]
        s = _arr(strand, np.uint8)
        left, right = _arr(left, np.uint32), _arr(right, np.uint32)
        rows = np.zeros(len(left), PAIR_DT)
        out, no = C.c_void_p(), C.c_uint64()
        self._chk(self.L.rb_host_overlap_split(self.ctx, C.c_uint64(n), _p(ops), _p(op_off), *map(_p, a), _p(s),
                                               C.c_uint64(len(left)), _p(left), _p(right), C.c_int(scores[0]),
                                               C.c_int(scores[1]), C.c_int(scores[2]), C.c_int(policy), _p(rows),
                                               C.byref(out), C.byref(no)), "rb_host_overlap_split")
        if no.value:
            buf = (C.c_char * (no.value * 4)).from_address(out.value)
            o = np.frombuffer(buf, dtype=np.uint32).copy()
        else:
            o = np.zeros(0, np.uint32)
        self.L.rb_host_free(out)
        return rows, o

    def swap(self, ops, op_off, strand):
        ops, op_off, s = _arr(ops, np.uint32), _arr(op_off, np.uint64), _arr(strand, np.uint8)
        n = len(op_off) - 1
        out = np.zeros(len(ops) + 4, np.uint32)
        src = np.concatenate([ops, np.zeros(4, np.uint32)])
        self._chk(self.L.rb_host_swap(self.ctx, C.c_uint64(n), _p(src), _p(op_off), _p(s), _p(out)), "rb_host_swap")
        return out[:len(ops)]

    # ---- CIGAR text <-> packed ops (k_text.hip) ----
    def parse_cigars(self, cigars):
        """cigars: list of bytes / str (the cg:Z: values).  Returns (op_off, ops, status)."""
        bs = [c.encode() if isinstance(c, str) else bytes(c) for c in cigars]
        off = np.zeros(len(bs) + 1, np.uint64)
        if bs:
            off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        text = np.frombuffer(b"".join(bs) + b"\0" * 32, dtype=np.uint8).copy()
        return self.parse_cigar_text(text, off, None)

    def parse_cigar_text(self, text, text_off, text_end=None):
        text, text_off = _arr(text, np.uint8), _arr(text_off, np.uint64)
        n = len(text_off) - 1
        end = None if text_end is None else _arr(text_end, np.uint64)
        op_off = np.zeros(n + 1, np.uint64)
        status = np.zeros(max(n, 1), np.uint8)
        out = C.c_void_p()
        self._chk(self.L.rb_host_parse_cigars(self.ctx, _p(text), _p(text_off), _p(end) if end is not None else None,
                                              C.c_uint64(n), _p(op_off), C.byref(out), _p(status)), "rb_host_parse_cigars")
        n_ops = int(op_off[n])
        if n_ops:
            ops = np.frombuffer((C.c_char * (n_ops * 4)).from_address(out.value), dtype=np.uint32).copy()
        else:
            ops = np.zeros(0, np.uint32)
        self.L.rb_host_free(out)
        return op_off, ops, status[:n]

    def format_cigars(self, ops, first, count, first_len=None, last_len=None):
        """Items = runs ops[first[i] : first[i] + count[i]] with optional clipped first / last lengths.  Returns (text_off, text)."""
        ops, first, count = _arr(ops, np.uint32), _arr(first, np.uint64), _arr(count, np.uint32)
        n = len(first)
        fl = None if first_len is None else _arr(first_len, np.uint32)
        ll = None if last_len is None else _arr(last_len, np.uint32)
        toff = np.zeros(n + 1, np.uint64)
        out = C.c_void_p()
        src = np.concatenate([ops, np.zeros(4, np.uint32)])
        self._chk(self.L.rb_host_format_cigars(self.ctx, _p(src), C.c_uint64(len(ops)), C.c_uint64(n), _p(first), _p(count),
                                               _p(fl) if fl is not None else None, _p(ll) if ll is not None else None, _p(toff),
                                               C.byref(out)), "rb_host_format_cigars")
        nb = int(toff[n])
        text = bytes((C.c_char * nb).from_address(out.value)) if nb else b""
        self.L.rb_host_free(out)
        return toff, text


    # ---- nucfreq (k_nucfreq.hip) ----
    def nucfreq_workspace_bytes(self, n_reads, n_regions, n_pos):
        f = self.L.rb_nucfreq_workspace_bytes
        f.restype = C.c_size_t
        return int(f(C.c_uint64(n_reads), C.c_uint64(n_regions), C.c_uint64(n_pos)))

    def dev_nucfreq(self, n_reads, ops, op_off, seq, seq_off, l_seq, tid, pos, flag, n_regions, rg_tid, rg_st, rg_en, out_off, n_pos,
                    counts, status, counters, ws, ws_bytes):
        """every pointer argument is a device address (int); enqueues on the context's stream"""
        class View(C.Structure):
            _fields_ = [("n_reads", C.c_uint64), ("ops", C.c_void_p), ("op_off", C.c_void_p), ("seq", C.c_void_p), ("seq_off", C.c_void_p),
                        ("l_seq", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p)]
        v = View(n_reads, ops, op_off, seq, seq_off, l_seq, tid, pos, flag)
        self._chk(self.L.rb_dev_nucfreq(self.ctx, C.byref(v), C.c_uint64(n_regions), C.c_void_p(rg_tid), C.c_void_p(rg_st), C.c_void_p(rg_en),
                                        C.c_void_p(out_off), C.c_uint64(n_pos), C.c_void_p(counts), C.c_void_p(status), C.c_void_p(counters),
                                        C.c_void_p(ws), C.c_size_t(ws_bytes)), "rb_dev_nucfreq")

    def nucfreq(self, tid, pos, flag, op_off, ops, l_seq, seq_off, seq, rg_tid, rg_st, rg_en):
        """Reads (BAM file order) x regions -> (counts [n_positions, 4] u32 with NF_COVERED in bit 31 of column 0,
        read_status [n_reads], counters dict)."""
        tid, pos, flag = _arr(tid, np.int32), _arr(pos, np.int64), _arr(flag, np.uint32)
        op_off, ops = _arr(op_off, np.uint64), np.concatenate([_arr(ops, np.uint32), np.zeros(4, np.uint32)])
        l_seq, seq_off = _arr(l_seq, np.uint32), _arr(seq_off, np.uint64)
        seq = np.concatenate([_arr(seq, np.uint8), np.zeros(32, np.uint8)])
        rg_tid, rg_st, rg_en = _arr(rg_tid, np.int32), _arr(rg_st, np.uint64), _arr(rg_en, np.uint64)
        n, nr = len(tid), len(rg_tid)
        n_pos = int((rg_en.astype(np.int64) - rg_st.astype(np.int64)).sum()) if nr else 0

        class View(C.Structure):
            _fields_ = [("n_reads", C.c_uint64), ("ops", C.c_void_p), ("op_off", C.c_void_p), ("seq", C.c_void_p), ("seq_off", C.c_void_p),
                        ("l_seq", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p)]
        v = View(n, ops.ctypes.data, op_off.ctypes.data, seq.ctypes.data, seq_off.ctypes.data, l_seq.ctypes.data, tid.ctypes.data,
                 pos.ctypes.data, flag.ctypes.data)
        counts = np.zeros((max(n_pos, 1), 4), np.uint32)
        status = np.zeros(max(n, 1), np.uint32)
        ctr = np.zeros(6, np.uint64)
        self._chk(self.L.rb_host_nucfreq(self.ctx, C.byref(v), C.c_uint64(nr), _p(rg_tid), _p(rg_st), _p(rg_en), _p(counts), _p(status),
                                         _p(ctr)), "rb_host_nucfreq")
        return counts[:n_pos], status[:n], dict(max_depth=int(ctr[0]), n_covered=int(ctr[1]), n_bad=int(ctr[2]), unsorted=int(ctr[3]),
                                                    n_dropped=int(ctr[4]), cap_overflow=int(ctr[5]))


def synth_n_ops(seed, first_record, n_rec, lo, hi):
    L = lib()
    return np.array([L.rb_synth_n_ops(seed, first_record + i, lo, hi) for i in range(n_rec)], dtype=np.uint64)


def synth_fill_ops_host(seed, first_record, op_off):
    L = lib()
    op_off = _arr(op_off, np.uint64)
    ops = np.zeros(int(op_off[-1]), np.uint32)
    L.rb_synth_fill_ops_host(C.c_uint64(seed), C.c_uint64(first_record), C.c_uint64(len(op_off) - 1), _p(op_off),
                             _p(ops))
    return ops


class DevBuf:
    """Device memory from the LIBRARY's allocator (rb_dev_alloc: what a host of the C ABI is told to use for a resident batch --
    requests of 256 MB and more are pieced together from 2 MB physical chunks, which decides 10-15 % of the streaming kernels' time,
    DESIGN.md section 3), seen by torch through the CUDA array interface without a copy: `.t` is the tensor.  free() gives the
    memory back (before the engine is closed)."""

    def __init__(self, eng, torch, n, dtype, device=None, placed_tries=1, score=None):
        self.eng, self.n, self.dtype = eng, int(n), dtype
        self.item = torch.empty(0, dtype=dtype).element_size()
        self.placement = None                       # (placed_tries > 1: {"sweep_ms": [...], "kept": i} of rb_dev_alloc_placed)

        def take():
            if placed_tries > 1:
                ptr, ms, kept = eng.dev_alloc_placed(max(self.n * self.item, 256), placed_tries, score)
                self.placement = {("launch_ms" if score else "sweep_ms"): ms, "kept": kept}
                return ptr
            return eng.dev_alloc(max(self.n * self.item, 256))
        try:
            self.ptr = take()
        except Exception:
            torch.cuda.empty_cache()                # (memory torch's caching allocator holds but does not use is not free to the driver)
            self.ptr = take()
        self.chunked = eng.L.rb_dev_alloc_mode(eng.ctx, C.c_void_p(self.ptr)) == 1  # (False: plain hipMalloc memory -- small, or the fallback)
        self.__cuda_array_interface__ = {"shape": (self.n * self.item,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}
        try:
            self.t = torch.as_tensor(self, device=device if device is not None else "cuda").view(dtype)
            if device is not None and self.t.device != torch.device(device):
                raise RuntimeError(f"the buffer was taken for {self.t.device}, not {device}")
        except Exception:
            self.t = None
            eng.dev_free(self.ptr)
            self.ptr = 0
            raise

    def free(self):
        if self.ptr:
            self.t = None
            self.eng.dev_free(self.ptr)
            self.ptr = 0

    def release(self):
        """give the buffer back to the context (rb_dev_release): it stays mapped and keeps its pages for the next DevBuf of this size"""
        if self.ptr:
            self.t = None
            self.eng.dev_release(self.ptr)
            self.ptr = 0
