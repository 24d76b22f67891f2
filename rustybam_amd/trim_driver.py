"""Host-side pass driver of trim-paf (Paf::overlapping_paf_recs, paf.rs:210-305) over the device pair kernel.

Plumbing for tests: record bookkeeping (stable sort by query name, pair scan, one pair per query per pass,
recursion as a loop, contained flags) happens here exactly as in the reference; every pass's independent
pairs go to the GPU in one rb_host_overlap_split call."""
import os

import numpy as np

from . import capi


def overlapping_paf_recs(eng, recs, scores=(1, 1, 1), remove_contained=False, policy=capi.BSEARCH_MODERN, max_passes=10000):
    """recs: list of dicts with q_name, q_st, q_en, t_st, t_en, strand (int), cigar (uint32 array), plus
    any passthrough fields.  Returns the new list (re-ordered by q_name like the reference)."""
    recs = [dict(r) for r in recs]
    for _ in range(max_passes):
        # remove_trailing_indels on every record (paf.rs:218-220): take the normalised view from the device
        off = np.zeros(len(recs) + 1, np.uint64)
        off[1:] = np.cumsum([len(r["cigar"]) for r in recs])
        ops = np.concatenate([r["cigar"] for r in recs]) if recs else np.zeros(0, np.uint32)
        arr = lambda k: np.array([r[k] for r in recs], np.uint64)  # noqa: E731
        strand = np.array([r["strand"] for r in recs], np.uint8)
        _, norm = eng.scan_records(ops, off, arr("t_st"), arr("t_en"), arr("q_st"), arr("q_en"), strand)
        for r, nr in zip(recs, norm):
            if nr["status"] != 0:
                raise RuntimeError(f"remove_trailing_indels: status {nr['status']} (the reference panics)")
            if nr["lead_ops"] or nr["trail_ops"]:
                lead, trail = r["cigar"][:nr["lead_ops"]], r["cigar"][len(r["cigar"]) - nr["trail_ops"]:][::-1]
                cs = lambda o: "".join(f"{int(v) >> 4}{'MIDNSHP=X'[int(v) & 15]}" for v in o)  # noqa: E731
                r["id"] = r.get("id", "") + f"_TO.{cs(lead)}.{cs(trail)}"
                r["cigar"] = r["cigar"][nr["first_op"]:nr["first_op"] + nr["n_ops"]]
            for k in ("t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
                r[k] = int(nr[k])
        order = sorted(range(len(recs)), key=lambda i: recs[i]["q_name"])  # stable, paf.rs:223
        recs = [recs[i] for i in order]
        n = len(recs)
        contained = [False] * n
        if n < 2:
            return recs
        pairs = []
        for i in range(n - 1):
            j = i + 1
            while j < n and recs[i]["q_name"] == recs[j]["q_name"]:
                r1, r2 = recs[i], recs[j]
                mn, mx = min(r1["q_en"], r2["q_en"]), max(r1["q_st"], r2["q_st"])
                ov = mn - mx if mn >= mx else 0
                if ov >= 1:
                    if ov == r2["q_en"] - r2["q_st"]:
                        contained[j] = True
                    elif ov == r1["q_en"] - r1["q_st"]:
                        contained[i] = True
                    else:
                        pairs.append((ov, i, j) if r1["q_st"] <= r2["q_st"] else (ov, j, i))
                j += 1
        pairs.sort(key=lambda t: -t[0])  # stable descending by overlap, paf.rs:262
        seen, todo, unseen = set(), [], 0
        for _ov, i, j in pairs:
            q = recs[i]["q_name"]
            if q in seen:
                unseen += 1
            else:
                seen.add(q)
                todo.append((i, j))
        if todo:
            off = np.zeros(n + 1, np.uint64)
            off[1:] = np.cumsum([len(r["cigar"]) for r in recs])
            ops = np.concatenate([r["cigar"] for r in recs])
            arr = lambda k: np.array([r[k] for r in recs], np.uint64)  # noqa: E731
            strand = np.array([r["strand"] for r in recs], np.uint8)
            rows, out = eng.overlap_split(ops, off, arr("t_st"), arr("t_en"), arr("q_st"), arr("q_en"), strand,
                                          [p[0] for p in todo], [p[1] for p in todo], scores, policy)
            for (i, j), row in zip(todo, rows):
                if row["status"] != 0:
                    raise RuntimeError(f"trim pair ({i},{j}): status {row['status']} (the reference panics)")
                for s, idx in ((0, i), (1, j)):
                    r = recs[idx]
                    r["t_st"], r["t_en"] = int(row["t_st"][s]), int(row["t_en"][s])
                    r["q_st"], r["q_en"] = int(row["q_st"][s]), int(row["q_en"][s])
                    r["nmatch"], r["aln_len"] = int(row["nmatch"][s]), int(row["aln_len"][s])
                    r["cigar"] = out[int(row["out_off"][s]):int(row["out_off"][s]) + int(row["out_n"][s])].copy()
        if unseen > 0:
            continue
        if remove_contained:
            recs = [r for r, c in zip(recs, contained) if not c]
        return recs
    raise RuntimeError("trim-paf did not converge")


# ------------------------------------------------------------------------------------------------------------------------------
# The same driver with the batch RESIDENT on the device across the passes: only pair lists go up and a few columns of the pair
# rows come down.  Array-based (numpy) so that it scales to SURVEY 8d config 4 (1e7 records); the per-pass bookkeeping of
# Paf::overlapping_paf_recs (paf.rs:223-301) is vectorised over the query groups:
#   - records are (stably) ordered by query name once; a pass looks at every pair (i < j) of one name (paf.rs:231-261),
#   - overlap == length of one of the two marks it contained, otherwise the pair is a candidate, left = smaller q_st (:244-256),
#   - candidates are stably sorted by overlap, descending, and the first of every name is cut in this pass (:262-284);
#     if a name had more than one candidate the whole thing runs again (:286-288),
#   - the contained flags of the LAST pass are what --remove-contained applies (:289-301).
# ------------------------------------------------------------------------------------------------------------------------------
def select_pairs(order, grp_sorted, q_st, q_en):
    """One pass of pair selection.  order: record indices stably sorted by query name; grp_sorted: the dense group id of each record in that
    order (non-decreasing).  Returns (left, right, unseen, contained) with left/right as RECORD indices."""
    n = len(order)
    contained = np.zeros(n, bool)              # by position in `order`
    if n < 2:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0, contained
    qs, qe = q_st[order].astype(np.int64), q_en[order].astype(np.int64)
    start = np.flatnonzero(np.r_[True, grp_sorted[1:] != grp_sorted[:-1]])
    size = np.diff(np.r_[start, n])
    pos_in = np.arange(n) - np.repeat(start, size)
    left_room = np.repeat(size, size) - pos_in - 1   # records of the same name behind this one
    ci, cj, cov = [], [], []
    for d in range(1, int(size.max())):
        i = np.flatnonzero(left_room >= d)
        j = i + d
        ov = np.minimum(qe[i], qe[j]) - np.maximum(qs[i], qs[j])
        keep = ov >= 1
        i, j, ov = i[keep], j[keep], ov[keep]
        c2 = ov == (qe[j] - qs[j])
        c1 = ~c2 & (ov == (qe[i] - qs[i]))
        contained[j[c2]] = True
        contained[i[c1]] = True
        cand = ~(c1 | c2)
        ci.append(i[cand]); cj.append(j[cand]); cov.append(ov[cand])
    if not ci:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0, contained
    ci, cj, cov = np.concatenate(ci), np.concatenate(cj), np.concatenate(cov)
    if len(ci) == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0, contained
    # the reference pushes candidates in (i, j) order and stable-sorts by overlap descending; the first of every name wins
    k = np.lexsort((cj, ci, -cov, grp_sorted[ci]))
    g = grp_sorted[ci][k]
    first = np.r_[True, g[1:] != g[:-1]]
    sel = k[first]
    i, j = ci[sel], cj[sel]
    swap = qs[i] > qs[j]
    li, ri = np.where(swap, j, i), np.where(swap, i, j)
    return order[li].astype(np.uint32), order[ri].astype(np.uint32), int(len(ci) - len(sel)), contained


class ResidentTrim:
    """trim-paf over a batch that stays in HBM.  `torch` supplies device memory; all compute goes through the C ABI."""

    def __init__(self, eng, torch, dev, ops, op_off, t_st, t_en, q_st, q_en, strand, group, room_factor=3.0):
        self.eng, self.torch, self.dev = eng, torch, dev
        self.n = len(op_off) - 1
        n_ops = int(op_off[-1])
        self.n_ops0 = n_ops
        self.op_off_host = np.ascontiguousarray(op_off, dtype=np.uint64)  # (as the batch came: what a plan for RB_LIFT_OP_STARTS is built from)
        cap = int(n_ops * (1.0 + room_factor)) + 4096
        self._own = capi.DevBuf(eng, torch, cap + 64, torch.int32, device=dev)         # [original ops | room for the clips of the passes], from the
        self.d_ops = self._own.t                                             # library's allocator (2 MB physical chunks: DESIGN.md section 3)
        self.d_ops[n_ops:].zero_()
        if isinstance(ops, np.ndarray):
            self.d_ops[:n_ops] = torch.from_numpy(np.ascontiguousarray(ops, dtype=np.uint32).view(np.int32)).to(dev)
        else:
            self.d_ops[:n_ops] = ops[:n_ops]                                # (a device tensor already)
        self.cap, self.cursor = cap, (n_ops + 31) // 32 * 32
        i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).to(dev)  # noqa: E731
        self.d_off = i64(op_off)
        self.d_c = [i64(x) for x in (t_st, t_en, q_st, q_en)]
        self.d_strand = torch.from_numpy(np.ascontiguousarray(strand, dtype=np.uint8)).to(dev)
        self.d_contig = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.d_norm = torch.zeros(max(self.n, 1) * 64, dtype=torch.uint8, device=dev)
        self.view = eng.batch_view(self.n, n_ops, self.d_ops.data_ptr(), self.d_off.data_ptr(), *[x.data_ptr() for x in self.d_c],
                                   self.d_strand.data_ptr(), self.d_contig.data_ptr())
        torch.cuda.synchronize()
        eng.dev_scan_records(self.view, 0, self.d_norm.data_ptr())            # remove_trailing_indels + check_integrity, once
        torch.cuda.synchronize()
        norm = self.d_norm.cpu().numpy().view(capi.NORM_DT)[:self.n]
        if (norm["status"] != 0).any():
            raise RuntimeError("remove_trailing_indels / check_integrity: the reference panics on this input")
        self.q_st, self.q_en = norm["q_st"].astype(np.uint64), norm["q_en"].astype(np.uint64)
        self.cur_n = norm["n_ops"].astype(np.uint64)
        group = np.asarray(group)
        self.order = np.argsort(group, kind="stable")
        gs = group[self.order]
        self.grp_sorted = np.cumsum(np.r_[0, gs[1:] != gs[:-1]])
        self.passes, self.pairs_done, self.pairs_by_wave = 0, 0, 0

    def run(self, scores=(1, 1, 1), policy=capi.BSEARCH_MODERN, max_passes=100000, check_host=False, fetch=True, events=None, on_pass=None):
        """the passes of Paf::overlapping_paf_recs (paf.rs:286-288).  Everything of a pass runs on the device: the pair scan and the
        selection (rb_dev_trim_select), the split + clip of the chosen pairs in place (rb_dev_overlap_split, rb_dev_apply_pairs), the
        status check (rb_dev_trim_check); the host reads 64 bytes per pass.  check_host: also run the numpy restatement of the
        selection (select_pairs) on the coordinates of every pass and compare (tests).
        fetch=False: the normalised rows stay on the device when the passes are done (fetch() brings them later: 64 bytes per record,
        640 MB for config 4 -- a consumer that goes on on the device never needs them).  events: a list that receives, per pass,
        (pairs, ms of the selection, ms of the pair kernels, ms of apply + check) from HIP events on the engine's stream.
        on_pass(i, k, d_left, d_right, d_rows): called after pass i has cut its k pairs (rows = k * 128 bytes, status checked), before the
        next selection -- tests compare every pair row of a pass with the op-space CPU port there."""
        torch, eng, dev = self.torch, self.eng, self.dev
        n_groups, d_order, d_grp, d_cont, d_l, d_r, d_po, d_rows, d_pass, d_scr = self._pass_buffers()
        d_cont.zero_(), d_pass.zero_()  # (the buffers are kept between calls: a second run() starts from zeros like the first)
        self._by_wave = None
        ev_log = []
        for _ in range(max_passes):
            if check_host:
                norm = self.d_norm.cpu().numpy().view(capi.NORM_DT)[:self.n]
                want = select_pairs(self.order, self.grp_sorted, norm["q_st"].astype(np.uint64), norm["q_en"].astype(np.uint64))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if events is not None else None
            if ev:
                ev[0].record()
            eng.dev_trim_select(self.n, n_groups, d_order.data_ptr(), d_grp.data_ptr(), self.d_norm.data_ptr(), self.cursor, d_cont.data_ptr(),
                                d_l.data_ptr(), d_r.data_ptr(), d_po.data_ptr(), d_pass.data_ptr(), d_scr.data_ptr())
            if ev:
                ev[1].record()
            eng.sync()
            ps = d_pass.cpu().numpy().view(capi.TRIM_PASS_DT)[0]
            k, deferred, end = int(ps["n_pairs"]), int(ps["n_deferred"]), int(ps["ops_end"])
            self.passes += 1
            if check_host:
                wl_, wr_, wun, wcont = want
                got = sorted(zip(d_l[:k].cpu().numpy().view(np.uint32).tolist(), d_r[:k].cpu().numpy().view(np.uint32).tolist()))
                assert got == sorted(zip(wl_.tolist(), wr_.tolist())) and deferred == wun, "device pass selection differs from the host restatement"
                c = np.zeros(self.n, bool)
                c[self.order] = wcont
                assert np.array_equal(d_cont[:self.n].cpu().numpy().astype(bool), c), "contained flags differ"
            if k:
                if end > self.cap:
                    raise RuntimeError("ResidentTrim: out of room for the clips (raise room_factor)")
                eng.dev_overlap_split(self.view, self.d_norm.data_ptr(), k, d_l.data_ptr(), d_r.data_ptr(), d_po.data_ptr(), scores,
                                      policy | (0 if os.environ.get("RB_TRIM_COPY") else capi.TRIM_IN_PLACE),   # regular records are cut where they are
                                      d_rows.data_ptr(), self.d_ops.data_ptr())
                if ev:
                    ev[2].record()
                eng.dev_apply_pairs(k, d_l.data_ptr(), d_r.data_ptr(), d_rows.data_ptr(), self.d_off.data_ptr(), self.d_norm.data_ptr())
                eng.dev_trim_check(k, d_rows.data_ptr(), d_pass.data_ptr())
                if ev:
                    ev[3].record()
                    ev_log.append((k, ev))
                eng.sync()
                bad = int(d_pass.cpu().numpy().view(capi.TRIM_PASS_DT)[0]["bad_status"])
                if bad:
                    raise RuntimeError(f"trim pair status {bad}: the reference panics")
                if on_pass is not None:
                    on_pass(self.passes - 1, k, d_l, d_r, d_rows)
                by_wave = (d_rows[: k * 128].view(torch.int64).view(k, 16)[:, 15] == 1).sum()  # (diagnostic word: 1 = wave-per-pair kernel; read at the end)
                self._by_wave = by_wave if getattr(self, "_by_wave", None) is None else self._by_wave + by_wave
                self.cursor = (end + 31) // 32 * 32
                self.pairs_done += k
            if deferred > 0:
                continue
            self._d_cont = d_cont
            if getattr(self, "_by_wave", None) is not None:
                self.pairs_by_wave += int(self._by_wave.item())
                self._by_wave = None
            if events is not None:
                torch.cuda.synchronize()
                for k_, e_ in ev_log:
                    events.append((k_, e_[0].elapsed_time(e_[1]), e_[1].elapsed_time(e_[2]), e_[2].elapsed_time(e_[3])))
            if fetch:
                self.fetch()
            return self
        raise RuntimeError("trim-paf did not converge")

    def _pass_buffers(self):
        """the query groups of the batch (host: one pass over the sorted group ids) and the device buffers of the passes, made once -- a
        resident host has them before the first pass, as it has the batch"""
        if getattr(self, "_pb", None) is None:
            torch, eng, dev = self.torch, self.eng, self.dev
            start = np.flatnonzero(np.r_[True, self.grp_sorted[1:] != self.grp_sorted[:-1]]) if self.n else np.zeros(0, np.int64)
            grp_off = np.r_[start, self.n].astype(np.uint64)
            n_groups = len(grp_off) - 1
            d_order = torch.from_numpy(np.ascontiguousarray(self.order, dtype=np.uint32).view(np.int32)).to(dev)
            d_grp = torch.from_numpy(grp_off.view(np.int64)).to(dev)
            d_cont = torch.zeros(self.n + 1, dtype=torch.uint8, device=dev)
            d_l, d_r = torch.zeros(n_groups + 1, dtype=torch.int32, device=dev), torch.zeros(n_groups + 1, dtype=torch.int32, device=dev)
            d_po = torch.zeros(n_groups + 1, dtype=torch.int64, device=dev)
            d_rows = torch.empty((n_groups + 1) * 128, dtype=torch.uint8, device=dev)
            d_pass = torch.zeros(64, dtype=torch.uint8, device=dev)
            d_scr = torch.zeros(eng.trim_select_scratch_bytes(n_groups), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            self._pb = (n_groups, d_order, d_grp, d_cont, d_l, d_r, d_po, d_rows, d_pass, d_scr)
        return self._pb

    def fetch(self):
        """the normalised rows and the contained flags of the finished passes, on the host (64 + 1 bytes per record)"""
        norm = self.d_norm.cpu().numpy().view(capi.NORM_DT)[:self.n]
        self.q_st, self.q_en = norm["q_st"].astype(np.uint64), norm["q_en"].astype(np.uint64)
        self.cur_n = norm["n_ops"].astype(np.uint64)
        self.contained = self._d_cont[:self.n].cpu().numpy().astype(bool)
        return self

    def release(self):
        """give the ops arena back (after gather(): the dense copy is what goes on)"""
        self.d_ops = None
        self._own.free()

    def gather(self):
        """The current records as a dense batch: (d_new_ops, new_op_off host, norm rows host)."""
        torch, eng, dev = self.torch, self.eng, self.dev
        d_new_off = torch.zeros(self.n + 1, dtype=torch.int64, device=dev)
        scratch = torch.empty(eng.text_scratch_bytes(self.n) + 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.dev_gather_records(self.n, self.d_ops.data_ptr(), self.d_off.data_ptr(), self.d_norm.data_ptr(), d_new_off.data_ptr(), 0, scratch.data_ptr())
        torch.cuda.synchronize()
        total = int(d_new_off[self.n].item())
        d_new = torch.zeros(total + 64, dtype=torch.int32, device=dev)
        eng.dev_gather_records(self.n, self.d_ops.data_ptr(), self.d_off.data_ptr(), self.d_norm.data_ptr(), d_new_off.data_ptr(), d_new.data_ptr(),
                               scratch.data_ptr())
        torch.cuda.synchronize()
        norm = self.d_norm.cpu().numpy().view(capi.NORM_DT)[:self.n].copy()
        return d_new, d_new_off.cpu().numpy().view(np.uint64), norm
