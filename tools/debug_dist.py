import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch, bench
import ska_sdp_exec_swiftly_amd as sw
from ska_sdp_exec_swiftly_amd.distributed import DistributedForward
wl = bench.WORKLOADS["64k-sparse"]; p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p)
fcs = sw.make_full_facet_cover(cfg)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
waves = {}
for c in sgs: waves.setdefault(c.off0, []).append(c)
waves = list(waves.values())
yB = p["yB_size"]
data = [torch.randn((yB, yB), device="cuda", dtype=torch.complex64) for _ in fcs]
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dfw = DistributedForward(cfg, fcs, data, lru_forward=1, subgrid_configs=sgs)
    e0 = ev(); dfw.local._get_BF_Fs(); e1 = ev()
    marks = []
    pending = None
    for wave in waves:
        a = ev(); h = dfw.start_wave(wave); b = ev()
        if pending is not None: dfw.finish_wave(pending)
        c = ev(); marks.append((a, b, c)); pending = h
    dfw.finish_wave(pending); e2 = ev()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    # sequential variant
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for wave in waves:
        dfw.get_subgrid_wave(wave)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("sequential waves only ms", (t3 - t2) * 1e3, "mem GB", torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30)
    print("rep", rep, "wall ms", (t1 - t0) * 1e3, "K1", e0.elapsed_time(e1), "start sum", sum(a.elapsed_time(b) for a, b, c in marks), "finish sum", sum(b.elapsed_time(c) for a, b, c in marks))
