"""Diagnostics (round 3): does the clip kernel's time depend on WHERE its input and output lie relative to each other?  One process,
one 72 GB arena allocated once; the ops and the output slots are carved from it at controlled offsets (the physical placement of
the arena stays what it is), the step is timed for every layout.  A second arena repeats the sweep (another physical placement).
If the time follows the relative offset, the library can choose a good one; if it follows the arena, it cannot."""
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
ops_bytes = (total + 64) * 4
out_bytes = (out_cap + 64) * 4
d_ws = torch.empty(eng.plan_workspace_bytes(plan, rows_cap), dtype=torch.uint8, device=dev)
d_rows = torch.empty((rows_cap + 1) * 64, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(64, dtype=torch.uint8, device=dev)
d_norm = torch.empty(n_rec * 64, dtype=torch.uint8, device=dev)
hdr = None
# LAYOUT_ARENA=contiguous: the arena from rb_dev_alloc under RB_ALLOC_MODE=contiguous (one physical block: offsets inside it ARE physical
# offsets); chunks: 2 MB physical chunks; default: torch's allocator (hipMalloc)
arena_mode = os.environ.get("LAYOUT_ARENA", "torch")
from rustybam_amd import capi
for arena_no in range(2):
    own = None
    if arena_mode == "torch":
        arena = torch.empty(ops_bytes + out_bytes + (6 << 30), dtype=torch.uint8, device=dev)
    else:
        os.environ["RB_ALLOC_MODE"] = arena_mode
        own = capi.DevBuf(eng, torch, ops_bytes + out_bytes + (6 << 30), torch.uint8, device=dev)
        arena = own.t
    base = arena.data_ptr()
    base_al = (base + (1 << 21) - 1) & ~((1 << 21) - 1)  # 2 MB aligned
    for name, a_off, gap in [("gap 0", 0, 0), ("gap 128 B", 0, 128), ("gap 4 KB", 0, 4096), ("gap 64 KB", 0, 65536), ("gap 1 MB", 0, 1 << 20),
                             ("gap 2 MB", 0, 2 << 20), ("gap 2 MB + 4 KB", 0, (2 << 20) + 4096), ("gap 1 GB", 0, 1 << 30), ("gap 1 GB + 68 KB", 0, (1 << 30) + 69632),
                             ("ops + 4 KB, gap 0", 4096, 0), ("ops + 1 MB, gap 512 KB", 1 << 20, 1 << 19), ("out first", -1, 0),
                             ("gap 8 KB", 0, 8192), ("gap 32 KB", 0, 32768), ("gap 256 KB", 0, 1 << 18), ("gap 683 KB", 0, 699392), ("gap 16 MB", 0, 16 << 20),
                             ("gap 256 MB", 0, 256 << 20), ("gap 3 GB + 1364 KB", 0, (3 << 30) + 1396736)]:
        if a_off >= 0:
            p_ops = base_al + a_off
            p_out = (p_ops + ops_bytes + gap + 127) & ~127
        else:
            p_out = base_al
            p_ops = (p_out + out_bytes + 127) & ~127
        o0 = p_ops - base
        d_ops = arena[o0:o0 + ops_bytes].view(torch.int32)
        o1 = p_out - base
        d_out = arena[o1:o1 + out_bytes].view(torch.int32)
        eng.dev_synth_fill_ops(seed, 0, n_rec, d_off.data_ptr(), d_ops.data_ptr())
        torch.cuda.synchronize()
        if hdr is None:
            z = torch.zeros(n_rec, dtype=torch.int64, device=dev)
            d_contig = torch.zeros(n_rec, dtype=torch.int32, device=dev)
            d_s0 = torch.full((n_rec,), ord("+"), dtype=torch.uint8, device=dev)
            d_red = torch.empty(n_rec * 72, dtype=torch.uint8, device=dev)
            v0 = eng.batch_view(n_rec, total, d_ops.data_ptr(), d_off.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), d_s0.data_ptr(), d_contig.data_ptr())
            eng.dev_scan_records(v0, d_red.data_ptr(), 0); torch.cuda.synchronize()
            red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
            t_st, t_en, q_st, q_en, strand = wl.headers(seed, 0, red["t_bases"], red["q_bases"], "uniform")
            hdr = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (t_st, t_en, q_st, q_en)] + [torch.from_numpy(strand).to(dev), d_contig]
            del d_red
        view = eng.batch_view(n_rec, total, d_ops.data_ptr(), d_off.data_ptr(), *[x.data_ptr() for x in hdr[:4]], hdr[4].data_ptr(), hdr[5].data_ptr())
        pol = rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN
        eng.set_timing(True)
        for _ in range(7):
            eng.dev_liftover(plan, view, d_norm.data_ptr(), pol, d_ws.data_ptr(), d_rows.data_ptr(), rows_cap, d_out.data_ptr(), out_cap, d_cnt.data_ptr())
        torch.cuda.synchronize()
        ks = np.sort(np.asarray(eng.get_timing()[-5:]))
        eng.set_timing(False)
        cnt = d_cnt.cpu().numpy().view(rustybam_amd.COUNTERS_DT)[0]
        print(f"arena {arena_no} (0x{base:x}) {name:24s}: kernel ms min {ks[0]:.3f} median {ks[2]:.3f} max {ks[-1]:.3f}  overflow {int(cnt['overflow'])}", flush=True)
    del arena, d_ops, d_out
    if own is not None:
        own.free()
    torch.cuda.empty_cache()
