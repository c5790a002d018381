"""The kernels of one forward wave of the 64k-sparse workload (contiguous-axis-first pipeline), launched a few times on
synthetic data: the target of rocprofv3 --pmc passes (per-kernel counters / HBM traffic)."""
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
facet_cfgs = sw.make_full_facet_cover(cfg)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
yB, m, xM, xA = p["yB_size"], core.xM_yN_size, p["xM_size"], p["xA_size"]
band = core.band_for_offsets([s.off1 for s in sgs])
facet = torch.randn((yB, yB), device="cuda", dtype=torch.complex64)
F = len(facet_cfgs)
bands = torch.empty((F, yB, core.band_columns(band)), dtype=torch.complex64, device="cuda")
reps = int(os.environ.get("REPS", "2"))
for _ in range(reps):
    for j in range(2):
        core.prepare_facet_band(facet, facet_cfgs[j].off1, band, out=bands[j])
bands.copy_(torch.randn(bands.shape, device="cuda", dtype=torch.complex64)) if os.environ.get("FILL") else None
wave = [s for s in sgs if s.off1 == 0]
rowmap, n_rows = core.subgrid_column_rows([s.off0 for s in wave])
off0s = [c.off0 for c in facet_cfgs]
off1s = [c.off1 for c in facet_cfgs]
for _ in range(reps):
    Q = core.prepare_facet_columns(bands, off0s, band, 0, rowmap, n_rows)
    G = core.transform_contributions(Q, 1, off0s, [s.off0 for s in wave], rowmap=rowmap)
    tmp = torch.empty((len(wave), xM, xA), dtype=torch.complex64, device="cuda")
    core.sum_finish_facets(G, off0s, off1s, tmp, [s.off1 for s in wave], xA)
    res = torch.empty((len(wave), xA, xA), dtype=torch.complex64, device="cuda")
    core.launch("finish_subgrid", tmp, xA, 1, xA, res, 1, xA, 0, size=xA, nbatch=len(wave), in_bs=xM * xA,
                out_bs=xA * xA, offs=[s.off0 for s in wave])
torch.cuda.synchronize()
print("band", band, "wave subgrids", len(wave), "rows kept", n_rows)
