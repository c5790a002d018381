"""Timeline of the K1 workgroups (diagnostic build variants/trace.so, -DSWF_TRACE=1): one K1 launch on a 22528^2 facet,
per-workgroup shader-clock stamps fetched with swiftly_hip_trace_fetch; prints phase durations and per-CU concurrency."""
import ctypes
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
os.environ.setdefault("SWIFTLY_HIP_LIB", os.path.join(ROOT, "variants", "trace.so"))
import torch  # noqa: E402

import bench  # noqa: E402
import ska_sdp_exec_swiftly_amd as sw  # noqa: E402
from ska_sdp_exec_swiftly_amd import _lib  # noqa: E402

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
rc = lib.swiftly_hip_trace_fetch(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
nblk = 2 * p["yB_size"]
t = buf[:nblk].astype(numpy.int64)
hw = buf[:nblk, 10]
xcc = (hw >> 32).astype(numpy.int64) & 0xF
hwid = (hw & 0xFFFFFFFF).astype(numpy.int64)
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
names = ["load", "twiddle", "compute0", "exchange0", "compute1", "exchange1", "compute2+store-issue", "store-drain"]
pts = [0, 1, 2, 3, 4, 5, 6, 7, 8]
t0 = t[:, 0].min()
print("blocks", nblk, "distinct CUs", len(numpy.unique(cuid)))
life = t[:, 8] - t[:, 0]
print(f"lifetime cycles: mean {life.mean():.0f} median {numpy.median(life):.0f} p10 {numpy.percentile(life, 10):.0f} p90 {numpy.percentile(life, 90):.0f}")
for i, nm in enumerate(names):
    d = t[:, pts[i + 1]] - t[:, pts[i]]
    print(f"  {nm:24s} mean {d.mean():8.0f}  median {numpy.median(d):8.0f}  p10 {numpy.percentile(d, 10):8.0f}  p90 {numpy.percentile(d, 90):8.0f}  ({100 * d.mean() / life.mean():.1f} %)")
# per-CU: what is the sibling doing while a workgroup loads?  fraction of each phase overlapped by another
# resident workgroup's compute phases (2..7)
sel = numpy.unique(cuid)[:8]
for c in sel[:2]:
    idx = numpy.where(cuid == c)[0]
    order = idx[numpy.argsort(t[idx, 0])]
    print("CU", c, "first 8 workgroups (start, load end, compute end, end) relative cycles:")
    for b in order[:8]:
        print("   blk", b, (t[b, [0, 1, 7, 8]] - t0).tolist())

