"""K2 row kernel duration vs number of rows (workgroups = 2*rows): tells how many workgroups are co-resident per CU."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch
import ska_sdp_exec_swiftly_amd as sw
P = sw.SWIFT_CONFIGS["64k[1]-n32k-1k"]
core = sw.SwiftlyConfig(backend="hip", **P).core
for R in (64, 128, 256, 384, 512, 1024):
    x = torch.randn((R, 22528), device="cuda", dtype=torch.complex64)
    out = core.prepare_facet(x, 0, axis=1)
    for _ in range(3): core.prepare_facet(x, 0, axis=1, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): core.prepare_facet(x, 0, axis=1, out=out)
    e1.record(); torch.cuda.synchronize()
    print("rows", R, "workgroups", 2 * R, "us", round(e0.elapsed_time(e1) / 20 * 1e3, 1))
