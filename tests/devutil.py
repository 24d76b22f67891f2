"""Device-pointer level plumbing for the -m gpu tests: a batch resident in HBM (torch tensors), rb_dev_liftover / rb_dev_break
with the output-sizing loop, rb_dev_digest_rows.  Torch is memory and streams only; every compute call goes through the C ABI."""
import numpy as np

import rustybam_amd


def _i64(torch, dev, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)


class DevBatch:
    def __init__(self, torch, eng, dev, b):
        """b: dict of host arrays ops, op_off, t_st, t_en, q_st, q_en, strand, contig"""
        self.torch, self.eng, self.dev = torch, eng, dev
        self.n_rec = len(b["op_off"]) - 1
        self.n_ops = int(b["op_off"][-1])
        self.op_off_host = np.ascontiguousarray(b["op_off"], dtype=np.uint64)
        self.contig_host = np.ascontiguousarray(b["contig"], dtype=np.uint32)
        ops = np.concatenate([np.ascontiguousarray(b["ops"], dtype=np.uint32), np.zeros(64, np.uint32)])
        self.d_ops = torch.from_numpy(ops.view(np.int32)).to(dev)
        self.d_off = _i64(torch, dev, self.op_off_host)
        self.d_c = [_i64(torch, dev, np.ascontiguousarray(b[k], dtype=np.uint64)) for k in ("t_st", "t_en", "q_st", "q_en")]
        self.d_strand = torch.from_numpy(np.ascontiguousarray(b["strand"], dtype=np.uint8)).to(dev)
        self.d_contig = torch.from_numpy(self.contig_host.view(np.int32)).to(dev)
        self.d_norm = torch.zeros(max(self.n_rec, 1) * 64, dtype=torch.uint8, device=dev)
        self.view = eng.batch_view(self.n_rec, self.n_ops, self.d_ops.data_ptr(), self.d_off.data_ptr(), *[x.data_ptr() for x in self.d_c],
                                   self.d_strand.data_ptr(), self.d_contig.data_ptr())
        self.d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)

    @classmethod
    def from_device(cls, torch, eng, dev, d_ops, n_ops, op_off_host, d_coords, d_strand):
        """a batch whose arrays are already in HBM (d_ops int32 with readable slack behind n_ops, d_coords = [t_st, t_en, q_st, q_en] int64)"""
        self = cls.__new__(cls)
        self.torch, self.eng, self.dev = torch, eng, dev
        self.n_rec, self.n_ops = len(op_off_host) - 1, int(n_ops)
        self.op_off_host = np.ascontiguousarray(op_off_host, dtype=np.uint64)
        self.contig_host = np.zeros(self.n_rec, np.uint32)
        self.d_ops = d_ops
        self.d_off = _i64(torch, dev, self.op_off_host)
        self.d_c = d_coords
        self.d_strand = d_strand
        self.d_contig = torch.zeros(self.n_rec, dtype=torch.int32, device=dev)
        self.d_norm = torch.zeros(max(self.n_rec, 1) * 64, dtype=torch.uint8, device=dev)
        self.view = eng.batch_view(self.n_rec, self.n_ops, self.d_ops.data_ptr(), self.d_off.data_ptr(), *[x.data_ptr() for x in self.d_c],
                                   self.d_strand.data_ptr(), self.d_contig.data_ptr())
        self.d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)
        return self

    @classmethod
    def from_trimmed(cls, torch, eng, dev, T):
        """the batch of a trim_driver.ResidentTrim as its passes left it -- cut in place, op_off a table of starts, the extents in the norm
        rows -- for rb_dev_liftover / rb_dev_break with RB_LIFT_OP_STARTS: no rb_dev_gather_records in between.  The plan comes from the
        op offsets the batch had before the passes.  Only valid while no pass has moved a record (T.pairs_by_wave == T.pairs_done)."""
        self = cls.__new__(cls)
        self.torch, self.eng, self.dev = torch, eng, dev
        self.n_rec, self.n_ops = T.n, T.n_ops0
        self.op_off_host = np.ascontiguousarray(T.op_off_host, dtype=np.uint64)
        self.contig_host = np.zeros(self.n_rec, np.uint32)
        self.d_ops, self.d_off, self.d_c, self.d_strand, self.d_contig, self.d_norm = T.d_ops, T.d_off, T.d_c, T.d_strand, T.d_contig, T.d_norm
        self.view = eng.batch_view(self.n_rec, self.n_ops, self.d_ops.data_ptr(), self.d_off.data_ptr(), *[x.data_ptr() for x in self.d_c],
                                   self.d_strand.data_ptr(), self.d_contig.data_ptr())
        self.d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)
        self.norm_ready = True
        return self

    def run(self, windows=None, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN, max_size=None, rows_cap=None, out_cap=None):
        """liftover over `windows` = (w_contig, w_st, w_en), or break-paf when max_size is given.  Returns (rows tensor [n, 16] int32,
        out tensor, counters) with the buffers left on the device."""
        torch, eng, dev = self.torch, self.eng, self.dev
        if not (policy & rustybam_amd.LIFT_FUSED_SCAN) and not getattr(self, "norm_ready", False):
            torch.cuda.synchronize()
            eng.dev_scan_records(self.view, 0, self.d_norm.data_ptr())
        plan = eng.plan_create(self.op_off_host, self.contig_host, *(windows if windows is not None else (None, None, None)))
        rows_cap = rows_cap or max(1024, 4 * self.n_rec)
        out_cap = out_cap or max(4096, eng.plan_out_capacity(plan, max_size is not None))
        if policy & rustybam_amd.LIFT_DESCRIPTORS:
            out_cap = max(out_cap, 4 * rows_cap + 65536)
        try:
            for _ in range(8):
                ws = torch.empty(eng.plan_workspace_bytes(plan, rows_cap), dtype=torch.uint8, device=dev)
                rows = torch.full(((rows_cap + 1) * 64,), 0xEE, dtype=torch.uint8, device=dev)
                out = torch.empty(out_cap + 64, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                if max_size is not None:
                    eng.dev_break(plan, self.view, self.d_norm.data_ptr(), max_size, policy, ws.data_ptr(), rows.data_ptr(), rows_cap, out.data_ptr(),
                                  out_cap, self.d_cnt.data_ptr())
                else:
                    eng.dev_liftover(plan, self.view, self.d_norm.data_ptr(), policy, ws.data_ptr(), rows.data_ptr(), rows_cap, out.data_ptr(), out_cap,
                                     self.d_cnt.data_ptr())
                torch.cuda.synchronize()
                cnt = self.d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0].copy()
                if cnt["redo_two_walk"]:  # RB_BREAK_ONE_WALK declined the batch: the caller decides what to do about it
                    self.last = (rows, out, ws)
                    return rows[:0].view(torch.int32).view(0, 16), out, cnt
                if not cnt["overflow"]:
                    n = int(cnt["n_hits"])
                    self.last = (rows, out, ws)  # keep the buffers alive for the caller
                    return rows[:n * 64].view(torch.int32).view(n, 16), out, cnt
                rows_cap = max(rows_cap, int(cnt["n_hits"]) + 64)
                out_cap = max(out_cap * 2, int(int(cnt["out_ops_needed"]) * 1.25) + 4096)
                if policy & rustybam_amd.LIFT_DESCRIPTORS:
                    out_cap = max(out_cap, 4 * rows_cap + 65536)
                del ws, rows, out
            raise AssertionError("could not size the outputs")
        finally:
            eng.plan_destroy(plan)

    def digest(self, rows, out, row_base=0, rec_base=0):
        torch = self.torch
        d = torch.zeros(1, dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize()
        self.eng.dev_digest_rows(self.view, rows.data_ptr(), rows.shape[0], out.data_ptr(), row_base, rec_base, d.data_ptr())
        torch.cuda.synchronize()
        return int(d.item()) & ((1 << 64) - 1)

    def host_rows(self, rows, out):
        """(HIT_DT rows, out ops as numpy u32 up to the highest op a row points at)"""
        r = rows.contiguous().cpu().numpy().view(np.uint8).reshape(-1).view(rustybam_amd.HIT_DT)
        ok = r["status"] == 0
        words = np.where((r["flags"] & rustybam_amd.HIT_DESCRIPTOR) != 0, 4, r["out_n"]).astype(np.uint64)
        hi = int((r["out_off"][ok] + words[ok]).max()) if ok.any() else 0
        return r, out[:hi].cpu().numpy().view(np.uint32)


def config4_resident(torch, eng, dev, n, seed=0x5EED0004, lo=300, hi=700, room_factor=1.6, strands="random"):
    """SURVEY 8d config 4's shape, resident in HBM: n records (a multiple of 4) of lo..hi synthetic ops, 4 records per query whose
    consecutive query spans overlap by U[100, 10000] bases (nothing contained).  Returns (ResidentTrim before its passes, host dict of
    the coordinates / strands / op counts / op offsets)."""
    from rustybam_amd import workload as wl, trim_driver
    n = n // 4 * 4
    nops = wl.n_ops(seed, 0, n, lo, hi)
    op_off = wl.op_offsets(nops)
    total_ops = int(op_off[-1])
    d_off = _i64(torch, dev, op_off)
    d_ops = torch.empty(total_ops + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(seed, 0, n, d_off.data_ptr(), d_ops.data_ptr())
    zeros = torch.zeros(n, dtype=torch.int64, device=dev)
    d_red = torch.empty(n * 72, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n, total_ops, d_ops.data_ptr(), d_off.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(),
                        torch.full((n,), ord("+"), dtype=torch.uint8, device=dev).data_ptr(), torch.zeros(n, dtype=torch.int32, device=dev).data_ptr())
    torch.cuda.synchronize()
    eng.dev_scan_records(v0, d_red.data_ptr(), 0)
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    tb, qb = red["t_bases"].astype(np.uint64), red["q_bases"].astype(np.uint64)
    rng = np.random.default_rng(seed)
    q_st = np.zeros(n, np.uint64)
    ov = rng.integers(100, 10001, n).astype(np.uint64)
    for j in range(1, 4):
        prev_en = q_st[j - 1::4] + qb[j - 1::4]
        q_st[j::4] = prev_en - np.minimum(ov[j::4], np.minimum(qb[j - 1::4], qb[j::4]) // np.uint64(2))
    q_en = q_st + qb
    t_st = rng.integers(0, 200_000_000, n).astype(np.uint64)
    t_en = t_st + tb
    strand = (np.where(rng.integers(0, 2, n) == 0, ord("+"), ord("-")) if strands == "random" else np.full(n, ord(strands))).astype(np.uint8)
    T = trim_driver.ResidentTrim(eng, torch, dev, d_ops, op_off, t_st, t_en, q_st, q_en, strand, np.arange(n) // 4, room_factor=room_factor)
    return T, dict(n=n, nops=nops, op_off=op_off, total_ops=total_ops, t_st=t_st, t_en=t_en, q_st=q_st, q_en=q_en, strand=strand)


class PairPortCheck:
    """Every pair row of every pass of a ResidentTrim against the op-space CPU port of the pair step (oracle/rb_opspace.c, held to the
    per-base oracle by tests/test_oracle_opspace.py): the port carries the passes over the ORIGINAL ops as views (start, count, end
    lengths) -- so the check is independent of what the device wrote into its batch -- and, pass by pass, compares split index and score,
    both cuts' coordinates, nmatch, aln_len, the kept range (out_off, out_n) and the two end words the device rewrote in place."""

    def __init__(self, oracle, torch, T, ops_host, h, n_threads=32):
        self.oracle, self.torch, self.T, self.ops, self.n_threads = oracle, torch, T, ops_host, n_threads
        self.off = h["op_off"][:-1].astype(np.uint64).copy()
        self.n = np.diff(h["op_off"]).astype(np.uint32)
        self.fl, self.ll = np.zeros(len(self.n), np.uint32), np.zeros(len(self.n), np.uint32)
        self.c = {k: h[k].astype(np.uint64).copy() for k in ("t_st", "t_en", "q_st", "q_en")}
        self.strand = h["strand"]
        self.pairs = 0

    def __call__(self, i, k, d_l, d_r, d_rows):
        torch = self.torch
        left, right = d_l[:k].cpu().numpy().view(np.uint32), d_r[:k].cpu().numpy().view(np.uint32)
        rows = d_rows[: k * 128].cpu().numpy().view(rustybam_amd.capi.PAIR_DT)
        got, bad = self.oracle.overlap_split_opspace(self.ops, self.off, self.n, self.fl, self.ll, self.c["t_st"], self.c["t_en"], self.c["q_st"], self.c["q_en"],
                                                     self.strand, left, right, (1, 1, 1), n_threads=self.n_threads)
        assert bad == 0, f"pass {i}: {bad} pairs outside the port's scope"
        assert (rows["status"] == 0).all()
        for f in ("split_idx", "split_score", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
            assert np.array_equal(rows[f], got[f]), f"pass {i}: {f} differs from the op-space port"
        assert np.array_equal(rows["out_n"], got["count"]), f"pass {i}: out_n"
        for s, rec in ((0, left), (1, right)):
            first = self.off[rec] + got["first"][:, s]
            assert np.array_equal(rows["out_off"][:, s], first), f"pass {i}: out_off (side {s})"
            last = first + got["count"][:, s].astype(np.uint64) - np.uint64(1)
            want_first = (got["first_len"][:, s].astype(np.uint32) << np.uint32(4)) | (self.ops[first.astype(np.int64)] & np.uint32(15))
            want_last = (got["last_len"][:, s].astype(np.uint32) << np.uint32(4)) | (self.ops[last.astype(np.int64)] & np.uint32(15))
            dev = self.T.d_ops.device
            g_first = self.T.d_ops[torch.from_numpy(first.astype(np.int64)).to(dev)].cpu().numpy().view(np.uint32)
            g_last = self.T.d_ops[torch.from_numpy(last.astype(np.int64)).to(dev)].cpu().numpy().view(np.uint32)
            assert np.array_equal(g_first, want_first) and np.array_equal(g_last, want_last), f"pass {i}: the end words of the clips (side {s})"
            # the port's state: the records as views
            self.off[rec] = first
            self.n[rec], self.fl[rec], self.ll[rec] = got["count"][:, s], got["first_len"][:, s], got["last_len"][:, s]
            for f in self.c:
                self.c[f][rec] = got[f][:, s]
        self.pairs += k
