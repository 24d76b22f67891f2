"""Record sharding across the GPUs of one node: contiguous record ranges balanced on
the op-count prefix; no data-path collective, host-side gather only.  Host logic, no device code."""
import numpy as np


def shard_bounds(op_off, n_shards):
    """Split records 0..n into n_shards contiguous ranges with (nearly) equal numbers of ops.
    Returns an int array b of length n_shards + 1; shard s owns records b[s]..b[s+1]."""
    op_off = np.asarray(op_off, dtype=np.uint64)
    n = len(op_off) - 1
    total = int(op_off[-1])
    b = np.zeros(n_shards + 1, dtype=np.int64)
    for s in range(1, n_shards):
        target = total * s // n_shards
        b[s] = int(np.searchsorted(op_off, np.uint64(target), side="left"))
    b[n_shards] = n
    return np.maximum.accumulate(np.minimum(b, n))


def shard_slice(arrs, op_off, lo, hi):
    """Views of one shard: per-record arrays sliced, ops sliced and op_off rebased to the shard."""
    op_off = np.asarray(op_off, dtype=np.uint64)
    o0, o1 = int(op_off[lo]), int(op_off[hi])
    out = {k: (v[o0:o1] if k == "ops" else v[lo:hi]) for k, v in arrs.items() if k != "op_off"}
    out["op_off"] = op_off[lo:hi + 1] - op_off[lo]
    return out


def gather_rows(parts, bounds, out_key="out_off"):
    """Concatenate per-shard (rows, out_ops) in shard order.  Because shards are contiguous record
    ranges this preserves the canonical order inside every contig; `rec` is shifted back to global
    record numbers and `out_off` to the concatenated op array."""
    rows, ops = [], []
    op_base = 0
    for s, (r, o) in enumerate(parts):
        r = r.copy()
        r["rec"] = r["rec"] + np.uint32(bounds[s])
        r[out_key] = r[out_key] + np.uint64(op_base)
        op_base += len(o)
        rows.append(r)
        ops.append(o)
    return (np.concatenate(rows) if rows else np.zeros(0)), (np.concatenate(ops) if ops else np.zeros(0, np.uint32))


def canonical_sort(rows, contig_of_rec):
    """Canonical liftover order for a multi-contig gather: contig first appearance, then record, then
    window order as produced (stable)."""
    contig_of_rec = np.asarray(contig_of_rec)
    first = {}
    for c in contig_of_rec:
        first.setdefault(int(c), len(first))
    rank = np.array([first[int(contig_of_rec[int(r)])] for r in rows["rec"]], dtype=np.int64)
    order = np.lexsort((np.arange(len(rows)), rows["rec"], rank))
    return rows[order]


# ---- nucfreq: partition by POSITION range, reads that straddle a boundary go to both ranks ----
def shard_regions(rg_tid, rg_st, rg_en, n_shards, tile=4096):
    """Split the positions of the regions into n_shards contiguous pieces of (nearly) equal size, cut on multiples of `tile`
    inside a region (whole tiles of the device kernel).  Returns per shard a list of (region index, st, en); concatenating the
    shards' pieces in shard order gives every region back in order."""
    rg_st, rg_en = np.asarray(rg_st, np.int64), np.asarray(rg_en, np.int64)
    total = int((rg_en - rg_st).sum())
    out = [[] for _ in range(n_shards)]
    if total == 0:
        return out
    done = 0
    for r, (st, en) in enumerate(zip(rg_st.tolist(), rg_en.tolist())):
        p = st
        while p < en:
            s = min(n_shards - 1, done * n_shards // total)
            limit = -(-(total * (s + 1)) // n_shards)            # first global position of the next shard
            q = min(en, p + max(limit - done, 1))
            if q < en:                                           # cut on a tile edge of this region
                q = min(en, st + -(-(q - st) // tile) * tile)
            out[s].append((r, p, q))
            done += q - p
            p = q
    return out


def shard_reads(tid, pos, end, pieces):
    """The contiguous range [lo, hi) of sorted reads that can reach any of the pieces (rg_tid resolved by the caller:
    pieces = [(tid, st, en)]); `end` = each read's reference end."""
    tid, pos, end = np.asarray(tid, np.int64), np.asarray(pos, np.int64), np.asarray(end, np.int64)
    key = tid * (1 << 32) + pos
    lo, hi = len(tid), 0
    pmax = np.maximum.accumulate(tid * (1 << 32) + end) if len(tid) else np.zeros(0, np.int64)
    for t, st, en in pieces:
        a = int(np.searchsorted(pmax, t * (1 << 32) + st, side="right"))
        b = int(np.searchsorted(key, t * (1 << 32) + en, side="left"))
        if b > a:
            lo, hi = min(lo, a), max(hi, b)
    return (lo, hi) if hi > lo else (0, 0)


def gather_counts(parts, pieces_per_shard, rg_st, rg_en):
    """Per-shard count arrays (positions of the shard's pieces, concatenated) back into one array per region list order."""
    rg_st, rg_en = np.asarray(rg_st, np.int64), np.asarray(rg_en, np.int64)
    off = np.concatenate([[0], np.cumsum(rg_en - rg_st)])
    out = np.zeros((int(off[-1]), 4), np.uint32)
    for counts, pieces in zip(parts, pieces_per_shard):
        o = 0
        for r, st, en in pieces:
            out[off[r] + (st - rg_st[r]):off[r] + (en - rg_st[r])] = counts[o:o + (en - st)]
            o += en - st
    return out


# ---- trim-paf: the dependency unit is the query name (paf.rs:223, :235): whole groups per rank ----
def shard_query_groups(q_names, weights, n_shards):
    """Records -> shards by query name: names are sorted (the order trim-paf prints in, paf.rs:223), cut into n_shards contiguous
    runs of whole groups with (nearly) equal total weight (ops).  Returns per shard the record indices in input order; running
    trim-paf on each shard and concatenating the outputs in shard order gives the output of the whole file."""
    names = np.asarray(q_names)
    w = np.asarray(weights, dtype=np.float64)
    uniq, inv = np.unique(names, return_inverse=True)           # sorted unique names; byte order = Rust's String order for ASCII
    gw = np.bincount(inv, weights=w, minlength=len(uniq))
    cum = np.cumsum(gw)
    total = cum[-1] if len(cum) else 0.0
    cuts = [0]
    for s in range(1, n_shards):
        cuts.append(int(np.searchsorted(cum, total * s / n_shards, side="left")))
    cuts.append(len(uniq))
    cuts = np.maximum.accumulate(np.minimum(cuts, len(uniq)))
    return [np.flatnonzero((inv >= cuts[s]) & (inv < cuts[s + 1])) for s in range(n_shards)]
