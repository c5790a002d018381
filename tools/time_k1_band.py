"""K1 of the contiguous-axis-first pipeline (prepare_facet_band of one 22528^2 facet of the 64k-sparse workload) timed
back to back with one HIP-event pair; used for A/B runs of build variants (SWIFTLY_HIP_LIB)."""
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
for off in (0, 22528, -22528):
    out = core.prepare_facet_band(facet, off, band)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            core.prepare_facet_band(facet, off, band, out=out)
        e1.record()
        torch.cuda.synchronize()
    print(os.environ.get("SWIFTLY_HIP_LIB", "default"), os.environ.get("SWIFTLY_ROW_SEGSKIP", ""), f"off {off}: K1 {e0.elapsed_time(e1) / n:.4f} ms per facet")

# the backward mirror: finish_facet along the contiguous axis of a band accumulator
acc = torch.randn((p["yB_size"], int(band[1])), device="cuda", dtype=torch.complex64)
for off in (0, 22528):
    fin = core.finish_facet_band(acc, band, off, p["yB_size"])
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            core.finish_facet_band(acc, band, off, p["yB_size"], out=fin)
        e1.record()
        torch.cuda.synchronize()
    print(os.environ.get("SWIFTLY_HIP_LIB", "default"), f"off {off}: finish_facet_band {e0.elapsed_time(e1) / n:.4f} ms per facet")
