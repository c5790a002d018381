"""Subgrid side of the 64k-sparse workload (K3 + sum_finish + K5b per wave) timed over all waves with HIP events, for
same-box A/B runs of build variants (SWIFTLY_HIP_LIB)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import ska_sdp_exec_swiftly_amd as sw  # noqa: E402

wl = bench.WORKLOADS["64k-sparse"]
p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p)
core = cfg.core
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
facet_cfgs = sw.make_full_facet_cover(cfg)
F, m = len(facet_cfgs), core.xM_yN_size
waves = {}
for c in sgs:
    waves.setdefault(c.off1, []).append(c)
G = {k: torch.randn((F, len(v), m, m), dtype=torch.complex64, device="cuda") for k, v in list(waves.items())[:6]}
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 0
    for k, g in G.items():
        sw.api._finish_from_G(core, g, facet_cfgs, waves[k])  # pylint: disable=protected-access
        n += 1
    e1.record()
    torch.cuda.synchronize()
print(os.environ.get("SWIFTLY_HIP_LIB", "default"), f"sum_finish + K5b: {1e3 * e0.elapsed_time(e1) / n:.1f} us per wave")
