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
