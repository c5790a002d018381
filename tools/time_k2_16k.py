"""K2 timing on the yN=16384 family (32k[1]-n16k-1k) to see how the lean row kernel does when two
workgroups fit per CU."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch
import ska_sdp_exec_swiftly_amd as sw
for name in ("32k[1]-n16k-1k", "64k[1]-n32k-1k"):
    P = sw.SWIFT_CONFIGS[name]
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    core = cfg.core
    yB, yN, m = P["yB_size"], P["yN_size"], core.xM_yN_size
    bf = torch.randn((yN, yB), device="cuda", dtype=torch.complex64)
    col = core.extract_column(bf, 928 * 3, 0)
    for _ in range(3): core.extract_column(bf, 928 * 3, 0, out=col)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): core.extract_column(bf, 928 * 3, 0, out=col)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(name, "K2 ms", round(ms, 4), "GB/s", round(8 * m * (yB + yN) / 1e9 / ms * 1e3, 1))
    del bf, col, core, cfg
