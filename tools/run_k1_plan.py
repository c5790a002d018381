"""K1 exactly as bench.py runs it for the 64k-sparse workload (row-compacted BF_F): used under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to measure the HBM traffic per launch group."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch
import bench
import ska_sdp_exec_swiftly_amd as sw
wl = bench.WORKLOADS["64k-sparse"]; p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
core = cfg.core
rowmap, nrows = core.subgrid_column_rows([s.off0 for s in sgs])
facet = torch.randn((p["yB_size"], p["yB_size"]), device="cuda", dtype=torch.complex64)
out = core.prepare_facet_rows(facet, 22528, rowmap, nrows)
for _ in range(2):
    core.prepare_facet_rows(facet, 22528, rowmap, nrows, out=out)
torch.cuda.synchronize()
print("rows kept", nrows, "of", p["yN_size"])
