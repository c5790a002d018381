"""One or more backward passes (band schedule of SwiftlyBackward) of a bench.py workload on RANDOM subgrids of its plan --
no forward pass in the process, so that a rocprofv3 trace of this command holds the backward kernels only
(tools/gpu_pmc.sh ... backward -> profiles/r5_pmc_kernels.json["<workload>:backward"]).

    python tools/run_backward.py [workload] [passes]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import ska_sdp_exec_swiftly_amd as sw  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "64k-sparse"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = bench.WORKLOADS[name]
p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p)
fcs = sw.make_full_facet_cover(cfg)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
waves = {}
for c in sgs:
    waves.setdefault(c.off1, []).append(c)
xA = p["xA_size"]
biggest = max(len(w) for w in waves.values())
data = torch.randn((biggest, xA, xA), device="cuda", dtype=torch.complex64)  # one wave tensor, reused by every wave
for rep in range(passes):
    bwd = sw.SwiftlyBackward(cfg, fcs, lru_backward=1, subgrid_configs=sgs, wave_axis=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in waves.values():
        bwd.add_new_subgrid_tasks(w, [data[k] for k in range(len(w))])
    out = bwd.finish()
    torch.cuda.synchronize()
    print(f"{name}: backward pass {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms, {len(sgs)} subgrids -> {len(out)} facets", flush=True)
    del out, bwd
