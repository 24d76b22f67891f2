"""Diagnostics (round 3): how much of the clip kernel's time is decided by how its buffers are pieced together physically?  The
same batch, its buffers from rb_dev_alloc under RB_ALLOC_MODE = default (plain hipMalloc), contiguous (hipDeviceMallocContiguous),
chunks / scatter (2 MB hipMemCreate chunks mapped in creation / pseudo-random order); kernel ms per allocation.
usage: python tools/contig_probe.py [mode ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rustybam_amd
from rustybam_amd import workload as wl
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
eng = rustybam_amd.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
seed, n_rec = 0x5EED0003, 1000000
nops = wl.n_ops(seed, 0, n_rec)
op_off = np.zeros(n_rec + 1, np.uint64); op_off[1:] = np.cumsum(nops)
total = int(op_off[-1])
d_off = torch.from_numpy(op_off.view(np.int64)).to(dev)
w_c, w_st, w_en = wl.sliding_windows(3000)
plan = eng.plan_create(op_off, np.zeros(n_rec, np.uint32), w_c, w_st, w_en)
rows_cap = 12609557
out_cap = max(4096, eng.plan_out_capacity(plan, False))
ops_bytes = ((total + 64) * 4 + 255) & ~255
out_bytes = ((out_cap + 64) * 4 + 255) & ~255
ws_bytes = (eng.plan_workspace_bytes(plan, rows_cap) + 255) & ~255
rows_bytes = ((rows_cap + 1) * 64 + 255) & ~255
d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)
d_norm = torch.empty(n_rec * 64, dtype=torch.uint8, device=dev)
hdr = None
modes = [m.split(":") for m in (sys.argv[1:] or ["default", "chunks", "default", "chunks", "scatter", "contiguous", "default", "chunks"])]
for it, mc in enumerate(modes):
    mode = mc[0]
    os.environ["RB_ALLOC_MODE"] = mode
    mode = ":".join(mc)
    import time
    ta = time.perf_counter()
    p_ops, p_out = eng.dev_alloc(ops_bytes), eng.dev_alloc(out_bytes)
    p_ws, p_rows = eng.dev_alloc(ws_bytes), eng.dev_alloc(rows_bytes)
    t_alloc = time.perf_counter() - ta
    eng.dev_synth_fill_ops(seed, 0, n_rec, d_off.data_ptr(), p_ops)
    torch.cuda.synchronize()
    if hdr is None:
        z = torch.zeros(n_rec, dtype=torch.int64, device=dev)
        d_contig = torch.zeros(n_rec, dtype=torch.int32, device=dev)
        d_s0 = torch.full((n_rec,), ord("+"), dtype=torch.uint8, device=dev)
        d_red = torch.empty(n_rec * 72, dtype=torch.uint8, device=dev)
        v0 = eng.batch_view(n_rec, total, p_ops, d_off.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), d_s0.data_ptr(), d_contig.data_ptr())
        eng.dev_scan_records(v0, d_red.data_ptr(), 0); torch.cuda.synchronize()
        red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
        t_st, t_en, q_st, q_en, strand = wl.headers(seed, 0, red["t_bases"], red["q_bases"], "uniform")
        hdr = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (t_st, t_en, q_st, q_en)] + [torch.from_numpy(strand).to(dev), d_contig]
        del d_red
    view = eng.batch_view(n_rec, total, p_ops, d_off.data_ptr(), *[x.data_ptr() for x in hdr[:4]], hdr[4].data_ptr(), hdr[5].data_ptr())
    pol = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
    eng.set_timing(True)
    for _ in range(7):
        eng.dev_liftover(plan, view, d_norm.data_ptr(), pol, p_ws, p_rows, rows_cap, p_out, out_cap, d_cnt.data_ptr())
    torch.cuda.synchronize()
    ks = np.sort(np.asarray(eng.get_timing()[-5:]))
    eng.set_timing(False)
    cnt = d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
    print(f"alloc {it} {mode:12s}: kernel ms min {ks[0]:.3f} median {ks[2]:.3f} max {ks[-1]:.3f}  ops 0x{p_ops:x} out 0x{p_out:x} overflow {int(cnt['overflow'])}  alloc {t_alloc:.2f} s", flush=True)
    for q in (p_ops, p_out, p_ws, p_rows):
        eng.dev_free(q)
