"""cProfile of the host side of one backward pass (band schedule) on the 64k-sparse workload: shows which call the
host thread spends its time in (how the ~2 ms per-call cost of size-changing hipMallocAsync was found).  GPU only."""
import os, sys, time, cProfile, pstats
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
xA = p["xA_size"]
data = {(c.off0, c.off1): torch.randn((xA, xA), dtype=torch.complex64, device="cuda") for c in sgs}
def one_pass(prof=None):
    bwd = sw.SwiftlyBackward(cfg, fcs, wave_axis=1, subgrid_configs=sgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if prof: prof.enable()
    for k, w in by1.items():
        parts = bwd.wave_contributions(w, [data[(c.off0, c.off1)] for c in w])
        bwd.accumulate_wave(w, parts)
    if prof: prof.disable()
    t1 = time.perf_counter()
    out = bwd.finish(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host loop {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
one_pass(); one_pass()
pr = cProfile.Profile(); one_pass(pr)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
