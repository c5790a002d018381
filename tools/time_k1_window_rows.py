"""K1 of the axis-1-first pipeline (whole-row kernel with the window epilogue, core.prepare_facet_window_rows) on one 22528^2
facet of the 64k workload for different window counts, beside the band store of the two-workgroup kernel: what the epilogue
costs per round of eight windows.  HIP events around 5 launches, best of 5."""
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
keys = sorted({c.off1 for c in sgs})
band = core.band_for_offsets(keys)
yB = p["yB_size"]
facet = torch.randn((yB, yB), device="cuda", dtype=torch.complex64)


def timed(fn, reps=5, rounds=5):
    best = 1e9
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


out = core.prepare_facet_band(facet, 22528, band)
print(f"band store (two workgroups per row): {timed(lambda: core.prepare_facet_band(facet, 22528, band, out=out)):.3f} ms")
for n in [int(a) for a in sys.argv[1:]] or (1, 8, 9, 16, 24, 25):
    use = (keys * 2)[:n]
    starts = torch.tensor(core.window_starts(band, use), dtype=torch.int32, device="cuda")
    rows = torch.empty((n, yB, 512) if os.environ.get("LAYOUT", "wave") == "wave" else (yB, n * 512), dtype=torch.complex64, device="cuda")
    t = timed(lambda: core.prepare_facet_window_rows(facet, 22528, band, starts, rows))
    print(f"window rows, {n:2d} windows: {t:.3f} ms")
