"""profiles/r05_reclen_summary.md from the table of tools/r05_reclen.sh (gpurun_out/<tag>/table.txt) -- python tools/r05_reclen_md.py <tag>"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rows = [ln.split() for ln in open(os.path.join(ROOT, "gpurun_out", tag, "table.txt")).read().strip().splitlines()[1:]]
T = collections.OrderedDict()
for mean, rec, op, sm, step, kern, frac, unpl, dig in rows:
    T.setdefault((int(mean), int(rec)), {})[(op, sm)] = (float(step), float(kern), float(frac), dig)
out = []
out.append("| mean ops / record (uniform 0.6 .. 1.4 of it) | records | liftover, per record | liftover, tiles | break-paf, per record | break-paf, tiles | digests |")
out.append("|---|---|---|---|---|---|---|")
for (mean, rec), d in T.items():
    def cell(op, sm):
        v = d.get((op, sm))
        return f"{v[1]:.2f} ms ({v[2]:.3f}); step {v[0]:.2f}" if v else "-"
    sms = sorted({k[1] for k in d if k[1] != "0"})
    sm = sms[0] if sms else "0"
    same = all(d.get((op, "0"), (0, 0, 0, "a"))[3] == d.get((op, sm), (0, 0, 0, "b"))[3] for op in ("liftover", "break"))
    out.append(f"| {mean} | {rec:,} | {cell('liftover', '0')} | {cell('liftover', sm)} | {cell('break', '0')} | {cell('break', sm)} | {'equal' if same else 'DIFFER'} |")
print("\n".join(out))
