"""Forward + backward round trip on the 64k-sparse workload: stage timing of the wave-batched backward pass
(HIP events) and a sanity check (finite output).  GPU only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import ska_sdp_exec_swiftly_amd as sw  # noqa: E402

wl = bench.WORKLOADS["64k-sparse"]
p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p)
fcs = sw.make_full_facet_cover(cfg)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
yB = p["yB_size"]
data = [torch.randn((yB, yB), device="cuda", dtype=torch.complex64) for _ in fcs]
waves = {}
for c in sgs:
    waves.setdefault(c.off0, []).append(c)
fwd = sw.SwiftlyForward(cfg, list(zip(fcs, data)), subgrid_configs=sgs, wave_axis=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
subgrids = {k: fwd.get_wave(w) for k, w in waves.items()}
torch.cuda.synchronize()
t1 = time.perf_counter()
del fwd
timer = bench.StageTimer(torch)
for rep in range(2):
    bwd = sw.SwiftlyBackward(cfg, fcs, lru_backward=1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k, w in waves.items():
        e = timer.start()
        parts = bwd.wave_contributions(w, [subgrids[k][i] for i in range(len(w))])
        timer.stop(f"rep{rep}_B1-4_prepare_split", e)
        e = timer.start()
        bwd.accumulate_wave(w, parts)
        timer.stop(f"rep{rep}_B5-7_accumulate_column_and_evict", e)
    e = timer.start()
    out = bwd.finish()
    timer.stop(f"rep{rep}_B8_finish", e)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep", rep, "backward ms", (t2 - t1) * 1e3)
    del bwd
print("forward axis0 (incl. K1) ms", (t1 - t0) * 1e3)
for name, (cnt, ms) in timer.totals().items():
    print(f"  {name:<45} {cnt:4d} groups {ms:9.3f} ms")
print("finite", all(bool(torch.isfinite(torch.view_as_real(o)).all()) for o in out))
