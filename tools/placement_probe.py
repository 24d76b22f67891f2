"""Diagnostics: is the run-to-run spread of the clip kernel a property of where the buffers land?  One process, the
same batch; the output / workspace / row buffers are re-allocated several times and the step timed each time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rustybam_amd
from rustybam_amd import workload as wl
dev = torch.device("cuda:0")
eng = rustybam_amd.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
seed, n_rec = 0x5EED0003, 1000000
nops = wl.n_ops(seed, 0, n_rec)
op_off = np.zeros(n_rec + 1, np.uint64); op_off[1:] = np.cumsum(nops)
total = int(op_off[-1])
d_off = torch.from_numpy(op_off.view(np.int64)).to(dev)
def run(tag, hold):
    d_ops = torch.empty(total + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(seed, 0, n_rec, d_off.data_ptr(), d_ops.data_ptr())
    z = torch.zeros(n_rec, dtype=torch.int64, device=dev)
    d_contig = torch.zeros(n_rec, dtype=torch.int32, device=dev)
    d_s0 = torch.full((n_rec,), ord("+"), dtype=torch.uint8, device=dev)
    d_red = torch.empty(n_rec * 72, dtype=torch.uint8, device=dev)
    d_norm = torch.empty(n_rec * 64, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n_rec, total, d_ops.data_ptr(), d_off.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), d_s0.data_ptr(), d_contig.data_ptr())
    eng.dev_scan_records(v0, d_red.data_ptr(), 0); torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    t_st, t_en, q_st, q_en, strand = wl.headers(seed, 0, red["t_bases"], red["q_bases"], "uniform")
    dd = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (t_st, t_en, q_st, q_en)]
    d_strand = torch.from_numpy(strand).to(dev)
    view = eng.batch_view(n_rec, total, d_ops.data_ptr(), d_off.data_ptr(), *[x.data_ptr() for x in dd], d_strand.data_ptr(), d_contig.data_ptr())
    w_c, w_st, w_en = wl.sliding_windows(3000)
    plan = eng.plan_create(op_off, np.zeros(n_rec, np.uint32), w_c, w_st, w_en)
    rows_cap, out_cap = 12609557, 7611312896
    d_ws = torch.empty(eng.plan_workspace_bytes(plan, rows_cap), dtype=torch.uint8, device=dev)
    d_rows = torch.empty((rows_cap + 1) * 64, dtype=torch.uint8, device=dev)
    d_out = torch.empty(out_cap + 64, dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)
    pol = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
    eng.set_timing(True)
    for _ in range(8):
        eng.dev_liftover(plan, view, d_norm.data_ptr(), pol, d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
    torch.cuda.synchronize()
    ks = np.sort(np.asarray(eng.get_timing()[-6:]))
    eng.set_timing(False)
    print(f"{tag}: kernel ms median {ks[len(ks)//2]:.3f}  d_ops 0x{d_ops.data_ptr():x} d_out 0x{d_out.data_ptr():x}", flush=True)
    if hold: return (d_ops, d_out, d_rows, d_ws)
held = []
for i in range(10):
    r = run(f"alloc {i}", hold=(i % 2 == 0))  # holding some buffers shifts where the next ones land
    if r: held.append(r)
    if len(held) > 1: held.pop(0)
    torch.cuda.empty_cache()
