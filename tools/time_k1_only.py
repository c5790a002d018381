"""K1 alone (prepare_facet_band of one 22528^2 facet of the 64k-sparse workload), one HIP-event pair per 9 launches:
quick A/B of build variants (SWIFTLY_HIP_LIB) and environment switches."""
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
band = core.band_for_offsets([c.off1 for c in sgs])
facet = torch.randn((p["yB_size"], p["yB_size"]), device="cuda", dtype=torch.complex64)
n = 9
res = []
for off in (0, 22528):
    out = core.prepare_facet_band(facet, off, band)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            core.prepare_facet_band(facet, off, band, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    res.append(best)
print(sys.argv[1] if len(sys.argv) > 1 else "", "K1 ms per facet:", " ".join(f"{v:.4f}" for v in res))
