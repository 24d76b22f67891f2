"""Host-side pass driver of trim-paf (Paf::overlapping_paf_recs, paf.rs:210-305) over the device pair kernel.

Plumbing for tests: record bookkeeping (stable sort by query name, pair scan, one pair per query per pass,
recursion as a loop, contained flags) happens here exactly as in the reference; every pass's independent
pairs go to the GPU in one rb_host_overlap_split call."""
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
