"""Backward pass on the 64k-sparse workload: stage timing (HIP events) of both schedules of SwiftlyBackward
(wave_axis 0 = the reference's, 1 = band accumulators) on the subgrids a forward pass produced, agreement of the
two results, and the adjoint identity <B y, x> == <y, F x>-style sanity check against the forward pass is left to
the parity tests.  GPU only."""
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
fwd = sw.SwiftlyForward(cfg, list(zip(fcs, data)), subgrid_configs=sgs, wave_axis=1)
by1 = {}
for c in sgs:
    by1.setdefault(c.off1, []).append(c)
subgrids = {}
for k, w in by1.items():
    res = fwd.get_wave(w)
    for i, c in enumerate(w):
        subgrids[(c.off0, c.off1)] = res[i].clone()
torch.cuda.synchronize()
del fwd, data
torch.cuda.empty_cache()
timer = bench.StageTimer(torch)
outs = {}
AXES = [int(a) for a in os.environ.get("AXES", "1,0").split(",")]
REPS = int(os.environ.get("REPS", "3"))
for axis in AXES:
    key = "off1" if axis == 1 else "off0"
    waves = {}
    for c in sgs:
        waves.setdefault(getattr(c, key), []).append(c)
    for rep in range(REPS):
        bwd = sw.SwiftlyBackward(cfg, fcs, lru_backward=1, wave_axis=axis, subgrid_configs=sgs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        host = [0.0, 0.0]
        for k, w in waves.items():
            e = timer.start()
            h0 = time.perf_counter()
            parts = bwd.wave_contributions(w, [subgrids[(c.off0, c.off1)] for c in w])
            h1 = time.perf_counter()
            timer.stop(f"axis{axis}_rep{rep}_B1-4_prepare_split", e)
            e = timer.start()
            h2 = time.perf_counter()
            bwd.accumulate_wave(w, parts)
            host[1] += time.perf_counter() - h2
            host[0] += h1 - h0
            timer.stop(f"axis{axis}_rep{rep}_B5-7_accumulate", e)
        print(f"  host ms: contributions {host[0] * 1e3:.2f}, accumulate {host[1] * 1e3:.2f}")
        e = timer.start()
        out = bwd.finish()
        timer.stop(f"axis{axis}_rep{rep}_B8_finish", e)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"wave_axis {axis} rep {rep} backward ms {(t2 - t1) * 1e3:.2f}", flush=True)
        del bwd
    outs[axis] = [o[::7, ::5].clone() for o in out]
    del out
    torch.cuda.empty_cache()
for name, (cnt, ms) in timer.totals().items():
    print(f"  {name:<45} {cnt:4d} groups {ms:9.3f} ms")
if len(outs) == 2:
    err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs[1], outs[0]))
    print("max |band - reference schedule| / max:", err)
