import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch, bench
import ska_sdp_exec_swiftly_amd as sw
wl = bench.WORKLOADS["64k-sparse"]; p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p); core = cfg.core
fcs = sw.make_full_facet_cover(cfg)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
by1 = {}
for c in sgs: by1.setdefault(c.off1, []).append(c)
w = max(by1.values(), key=len); S = len(w); xA = p["xA_size"]; xM = p["xM_size"]; m = core.xM_yN_size; F = len(fcs)
big = os.environ.get("BIG", "0") == "1"
if big:  # occupy memory like the round-trip tool does
    hold = [torch.empty((22528, 11648), dtype=torch.complex64, device="cuda") for _ in range(9)]
subs = [torch.randn((xA, xA), dtype=torch.complex64, device="cuda") for _ in w]
def T(label, fn, n=4):
    for i in range(n):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: print(f"{label}: gpu {e0.elapsed_time(e1):.3f} ms")
    return r
sub = torch.empty((S, xA, xA), dtype=torch.complex64, device="cuda")
T("stack", lambda: torch.stack(subs, out=sub))
tmp = torch.empty((S, xM, xA), dtype=torch.complex64, device="cuda")
T("prepare axis0", lambda: core.launch("prepare_subgrid", sub, xA, 1, xA, tmp, 1, xA, 0, size=xA, nbatch=S, in_bs=xA*xA, out_bs=xM*xA, offs=[sg.off0 for sg in w]))
parts = torch.empty((F, S, m, m), dtype=torch.complex64, device="cuda")
T("split+colpass", lambda: core.split_prepare_facets(tmp, [sg.off1 for sg in w], [c.off0 for c in fcs], [c.off1 for c in fcs], parts))
print("S =", S)
