"""Timeline of the whole-row K1 (diagnostic build variants/wtrace.so, -DSWF_TRACE=1 for row_pass.hip + row_whole.hip): one
launch on a 22528^2 facet; thread 0 of every workgroup stamps the shader clock at 12 points of every row."""
import ctypes
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
os.environ.setdefault("SWIFTLY_K1_WHOLE", "1")  # the plain band store through the whole-row kernel (default: two workgroups per row)
os.environ.setdefault("SWIFTLY_HIP_LIB", os.path.join(ROOT, "variants", "wtrace.so"))
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
out = core.prepare_facet_band(facet, 22528, band)
for _ in range(3):
    core.prepare_facet_band(facet, 22528, band, out=out)
torch.cuda.synchronize()
NB, NP = 49152, 12
lib = ctypes.CDLL(os.environ["SWIFTLY_HIP_LIB"])
buf = numpy.zeros((NB, NP), dtype=numpy.uint64)
rc = lib.swiftly_hip_wtrace_fetch(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
rows = p["yB_size"]
t = buf[:rows].astype(numpy.int64)
names = ["fold (waits for the row's loads)", "scale / descriptors / P0(A) / barrier", "scatter A, twiddle B, P0(B), issue early loads",
         "barrier, gather A, barrier", "scatter B + P1(A)", "barrier, gather B, barrier", "scatter A' + P1(B)",
         "barrier, gather A', barrier", "scatter B', issue late loads, P2(A), store A", "barrier, gather B' (issue)", "P2(B), store B"]
life = t[:, 11] - t[:, 0]
ok = life > 0
print("rows", rows, "traced", int(ok.sum()))
print(f"row time cycles (s_memtime ticks): mean {life[ok].mean():.0f} median {numpy.median(life[ok]):.0f} p10 {numpy.percentile(life[ok], 10):.0f} p90 {numpy.percentile(life[ok], 90):.0f}")
for i, nm in enumerate(names):
    d = (t[:, i + 1] - t[:, i])[ok]
    print(f"  {i:2d} {nm:50s} mean {d.mean():8.0f}  median {numpy.median(d):8.0f}  p10 {numpy.percentile(d, 10):8.0f}  p90 {numpy.percentile(d, 90):8.0f}  ({100 * d.mean() / life[ok].mean():.1f} %)")
# gap between the end of a row and the start of the workgroup's next one
grid = 256
nxt = t[grid:, 0] - t[:-grid, 11]
print(f"row-to-row gap: mean {nxt.mean():.0f}")
span = t[ok][:, 11].max() - t[ok][:, 0].min()
print(f"kernel span {span} ticks; rows per workgroup {rows / grid:.0f}; mean row time x rows per workgroup = {life[ok].mean() * rows / grid:.0f}")
