"""Shared helpers for the tests: PAF/BED decode in numpy, adversarial CIGAR generators, row comparison."""
import re

import numpy as np

OPC = {c: i for i, c in enumerate("MIDNSHP=X")}
OPCH = "MIDNSHP=X"
REF = {0, 2, 3, 7, 8}
QRY = {0, 1, 4, 7, 8}
CG_RE = re.compile(rb"(\d+)([MIDNSHP=X])")


CONT = 14  # continuation word (include/rustybam_amd.h): bits 28.. of the length of the op in front of it


def pack(cigar):
    """'4M1I' -> uint32 array of packed words (a length of 2^28 and more takes two)"""
    if isinstance(cigar, str):
        cigar = cigar.encode()
    out = []
    for n, c in CG_RE.findall(cigar):
        n = int(n)
        out.append(((n & 0x0FFFFFFF) << 4) | OPC[c.decode()])
        if n >> 28:
            out.append(((n >> 28) << 4) | CONT)
    return np.array(out, dtype=np.uint32)


def unpack(ops):
    out, i, n = [], 0, len(ops)
    while i < n:
        v = int(ops[i])
        ln = v >> 4
        if i + 1 < n and (int(ops[i + 1]) & 15) == CONT and (v & 15) != CONT:
            ln += ((int(ops[i + 1]) >> 4) & 15) << 28
            i += 1
        out.append(f"{ln}{OPCH[v & 15] if (v & 15) < 9 else '?'}")
        i += 1
    return "".join(out)


class Recs:
    """SoA batch of PAF records (what the C ABI consumes) + the host-side strings."""

    def __init__(self):
        self.q_name, self.t_name = [], []
        self.q_len, self.t_len, self.mapq = [], [], []
        self.q_st, self.q_en, self.t_st, self.t_en, self.strand = [], [], [], [], []
        self.cigars = []

    def add(self, line):
        t = line.split()
        self.q_name.append(t[0]); self.q_len.append(int(t[1])); self.q_st.append(int(t[2])); self.q_en.append(int(t[3]))
        self.strand.append(ord(t[4])); self.t_name.append(t[5]); self.t_len.append(int(t[6]))
        self.t_st.append(int(t[7])); self.t_en.append(int(t[8])); self.mapq.append(int(t[11]))
        cg = [x for x in t[12:] if x.startswith("cg:Z:")]
        self.cigars.append(pack(cg[0][5:]) if cg else np.zeros(0, np.uint32))

    def finish(self):
        self.n = len(self.cigars)
        self.op_off = np.zeros(self.n + 1, np.uint64)
        self.op_off[1:] = np.cumsum([len(c) for c in self.cigars])
        self.ops = np.concatenate(self.cigars) if self.n else np.zeros(0, np.uint32)
        names = {}
        self.contig = np.array([names.setdefault(t, len(names)) for t in self.t_name], dtype=np.uint32)
        self.contig_names = names
        for k in ("q_st", "q_en", "t_st", "t_en"):
            setattr(self, k, np.array(getattr(self, k), dtype=np.uint64))
        self.strand = np.array(self.strand, dtype=np.uint8)
        return self

    def arrays(self):
        return (self.ops, self.op_off, self.t_st, self.t_en, self.q_st, self.q_en, self.strand)


def read_paf(path):
    r = Recs()
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        for line in f:
            r.add(line)
    return r.finish()


def recs_from_lines(lines):
    r = Recs()
    for ln in lines:
        r.add(ln)
    return r.finish()


def read_bed(path, contig_names):
    """-> (w_contig, w_st, w_en, ids); contigs unknown to the records get fresh ids"""
    wc, ws, we, ids = [], [], [], []
    names = dict(contig_names)
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        t = line.rstrip("\n").split("\t")
        try:
            st, en = int(t[1]), int(t[2])
        except ValueError:
            continue
        wc.append(names.setdefault(t[0], len(names)))
        ws.append(st); we.append(en)
        ids.append(t[3] if len(t) > 3 else f"{t[0]}:{st + 1}-{en}")
    return (np.array(wc, np.uint32), np.array(ws, np.uint64), np.array(we, np.uint64), ids)


# ------------------------------------------------------------------ random CIGARs
def sums(ops):
    R = sum(int(v) >> 4 for v in ops if (int(v) & 15) in REF)
    Q = sum(int(v) >> 4 for v in ops if (int(v) & 15) in QRY)
    return R, Q


def random_cigar(rng, n_ops, mode):
    """mode 'regular': = X M I D, no two adjacent of one type, starts/ends on a match op (like minimap2)
       mode 'indel_ends': regular body with leading/trailing I/D runs
       mode 'spliced': regular plus N ops (introns), which the streaming kernel treats like D but which are never stripped
       mode 'wild': anything incl. N S H P, zero lengths, adjacent duplicates"""
    ops = []
    if mode == "wild":
        for _ in range(n_ops):
            c = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8], p=[.1, .12, .12, .04, .03, .03, .03, .38, .15]))
            ln = int(rng.choice([0, 1, 2, 3, 7, 40], p=[.04, .4, .2, .16, .15, .05]))
            ops.append((ln << 4) | c)
        return np.array(ops, np.uint32)
    prev = -1
    for i in range(n_ops):
        while True:
            if mode == "spliced" and (i == 0 or i == n_ops - 1):
                c = int(rng.choice([7, 8, 0, 3], p=[.75, .1, .1, .05]))  # an N at an end is never stripped
            elif i == 0 or i == n_ops - 1:
                c = int(rng.choice([7, 8, 0], p=[.8, .1, .1]))
            elif mode == "spliced":
                c = int(rng.choice([7, 8, 0, 1, 2, 3], p=[.42, .18, .05, .12, .12, .11]))
            else:
                c = int(rng.choice([7, 8, 0, 1, 2], p=[.45, .2, .05, .15, .15]))
            if c != prev:
                break
        prev = c
        ln = int(rng.choice([1, 2, 3, 9, 150], p=[.45, .2, .15, .15, .05]))
        ops.append((ln << 4) | c)
    if mode == "indel_ends":
        lead = [((int(rng.integers(1, 5)) << 4) | int(rng.choice([1, 2]))) for _ in range(int(rng.integers(0, 3)))]
        trail = [((int(rng.integers(1, 5)) << 4) | int(rng.choice([1, 2]))) for _ in range(int(rng.integers(0, 3)))]
        ops = lead + ops + trail
    return np.array(ops, np.uint32)


def random_batch(rng, n_rec, mode="regular", n_contig=2, max_ops=40, long_frac=0.1, break_frac=0.0):
    cig, t_st, t_en, q_st, q_en, strand, contig = [], [], [], [], [], [], []
    for _ in range(n_rec):
        n_ops = int(rng.integers(1, max_ops))
        if rng.random() < long_frac:
            n_ops = int(rng.integers(200, 1500))
        m = mode if mode != "mixed" else str(rng.choice(["regular", "indel_ends", "spliced", "wild"]))
        c = random_cigar(rng, n_ops, m)
        R, Q = sums(c)
        ts, qs = int(rng.integers(0, 3000)), int(rng.integers(0, 3000))
        te, qe = ts + R, qs + Q
        if rng.random() < break_frac:
            te += int(rng.integers(1, 3))
        cig.append(c); t_st.append(ts); t_en.append(te); q_st.append(qs); q_en.append(qe)
        strand.append(ord("+") if rng.random() < .5 else ord("-"))
        contig.append(int(rng.integers(0, n_contig)))
    op_off = np.zeros(n_rec + 1, np.uint64)
    op_off[1:] = np.cumsum([len(c) for c in cig])
    ops = np.concatenate(cig) if cig else np.zeros(0, np.uint32)
    return dict(ops=ops, op_off=op_off, t_st=np.array(t_st, np.uint64), t_en=np.array(t_en, np.uint64),
                q_st=np.array(q_st, np.uint64), q_en=np.array(q_en, np.uint64), strand=np.array(strand, np.uint8),
                contig=np.array(contig, np.uint32))


def random_windows(rng, batch, n_win, monotone=True):
    hi = int(batch["t_en"].max()) + 50 if len(batch["t_en"]) else 100
    n_contig = int(batch["contig"].max()) + 1 if len(batch["contig"]) else 1
    wc = rng.integers(0, n_contig + 1, n_win).astype(np.uint32)  # one contig id may have no records
    st = rng.integers(0, hi, n_win)
    ln = rng.choice([1, 2, 5, 30, 400, 5000], n_win)
    # some windows snap exactly to record bounds (equality takes the long path, liftover.rs:23)
    for i in range(n_win):
        if rng.random() < .25 and len(batch["t_st"]):
            r = int(rng.integers(0, len(batch["t_st"])))
            st[i] = int(batch["t_st"][r]); ln[i] = max(1, int(batch["t_en"][r]) - int(batch["t_st"][r]))
            wc[i] = batch["contig"][r]
    en = st + ln
    if monotone:
        order = np.lexsort((en, st))
        st, en, wc = st[order], np.maximum.accumulate(en[order]), wc[order]
        # per-contig monotone in BED order: sort contig-stable
    return wc.astype(np.uint32), st.astype(np.uint64), en.astype(np.uint64)


def batch_args(b):
    return (b["ops"], b["op_off"], b["t_st"], b["t_en"], b["q_st"], b["q_en"], b["strand"])


def compare_hits(g_rows, g_ops, o_rows, o_ops, what=""):
    """GPU (rustybam_amd.HIT_DT) vs oracle (pyoracle.HIT_DT) rows, same canonical order."""
    assert len(g_rows) == len(o_rows), f"{what}: {len(g_rows)} rows vs oracle {len(o_rows)}"
    for k in ("rec", "win", "status"):
        bad = np.nonzero(g_rows[k].astype(np.int64) != o_rows[k].astype(np.int64))[0]
        assert len(bad) == 0, f"{what}: field {k} differs at rows {bad[:5]}: gpu {g_rows[k][bad[:5]]} oracle {o_rows[k][bad[:5]]}"
    ok = o_rows["status"] == 0
    for k in ("t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
        bad = np.nonzero(ok & (g_rows[k].astype(np.uint64) != o_rows[k].astype(np.uint64)))[0]
        assert len(bad) == 0, f"{what}: field {k} differs at rows {bad[:5]}: gpu {g_rows[k][bad[:5]]} oracle {o_rows[k][bad[:5]]} (rec {o_rows['rec'][bad[:5]]} win {o_rows['win'][bad[:5]]})"
    bad = np.nonzero(ok & ((g_rows["flags"] & 1) != (o_rows["flags"] & 1)))[0]
    assert len(bad) == 0, f"{what}: inside flag differs at rows {bad[:5]}"
    bad = np.nonzero(ok & (g_rows["out_n"] != o_rows["out_n"]))[0]
    assert len(bad) == 0, f"{what}: out_n differs at rows {bad[:5]}: gpu {g_rows['out_n'][bad[:5]]} oracle {o_rows['out_n'][bad[:5]]}"
    for i in np.nonzero(ok)[0]:
        a = g_ops[int(g_rows["out_off"][i]):int(g_rows["out_off"][i]) + int(g_rows["out_n"][i])]
        b = o_ops[int(o_rows["out_off"][i]):int(o_rows["out_off"][i]) + int(o_rows["out_n"][i])]
        assert np.array_equal(a, b), f"{what}: cigar differs at row {i} (rec {o_rows['rec'][i]} win {o_rows['win'][i]}): gpu {unpack(a[:12])}.. oracle {unpack(b[:12])}.."


# ------------------------------------------------------------------ numpy twin of rb_dev_digest_rows
def _sm64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def digest_rows(rows, ops, row_base=0, rec_base=0):
    """The digest include/rustybam_amd.h defines for rb_dev_digest_rows, over rows whose clips are copied ops in `ops`
    (oracle output, or GPU output in copied-ops mode)."""
    total = 0
    with np.errstate(over="ignore"):
        for i, h in enumerate(rows):
            x = _sm64(np.uint64(int(h["rec"]) + rec_base))
            x = _sm64(x ^ np.uint64(h["win"]))
            x = _sm64(x ^ np.uint64(h["status"]))
            if int(h["status"]) == 0:
                n, off = int(h["out_n"]), int(h["out_off"])
                w = ops[off:off + n].astype(np.uint64)
                hops = int(_sm64((np.arange(n, dtype=np.uint64) << np.uint64(32)) | w).sum(dtype=np.uint64)) if n else 0
                x = _sm64(x ^ np.uint64(int(h["flags"]) & 1))
                for k in ("out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
                    x = _sm64(x ^ np.uint64(h[k]))
                x = _sm64(x ^ np.uint64(hops))
            total = (total + int(x) * (2 * (row_base + i) + 1)) & ((1 << 64) - 1)
    return total
