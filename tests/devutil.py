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

    def run(self, windows=None, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN, max_size=None, rows_cap=None, out_cap=None):
        """liftover over `windows` = (w_contig, w_st, w_en), or break-paf when max_size is given.  Returns (rows tensor [n, 16] int32,
        out tensor, counters) with the buffers left on the device."""
        torch, eng, dev = self.torch, self.eng, self.dev
        if not (policy & rustybam_amd.LIFT_FUSED_SCAN):
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
