"""Stream-level schedules of the per-wave stages of the 64k-sparse workload (r4 session 2): random band buffers, timing
only, existing entry points.  K2 per facet on 2 / 3 / 4 streams; the whole wave loop with the subgrid side of wave w on
its own stream next to the facet side of wave w + 1."""
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
facet_cfgs = sw.make_full_facet_cover(cfg)
F, m, yN = len(facet_cfgs), core.xM_yN_size, core.yN_size
yB = facet_cfgs[0].size
off0s = [c.off0 for c in facet_cfgs]
waves = {}
for c in sgs:
    waves.setdefault(int(c.off1), []).append(c)
keys = sorted(waves)  # all 25 waves: one pass worth of per-wave work
band = core.band_for_offsets([sg.off1 for sg in sgs])
bands = torch.randn((F, yB, core.band_columns(band)), dtype=torch.complex64, device="cuda")
rows = {k: core.subgrid_column_rows([sg.off0 for sg in waves[k]]) for k in keys}
Q = {k: torch.empty((F, rows[k][1], m), dtype=torch.complex64, device="cuda") for k in keys}
side = [torch.cuda.Stream() for _ in range(4)]
sub = torch.cuda.Stream()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best  # ms per pass worth of waves


def k2_wave(k, nstreams, group=1):
    """K2 of wave k issued from the CURRENT stream: all facets in one call (nstreams = 0) or `group` facets per call
    round-robin over `nstreams` side streams forked from / joined into the current stream"""
    if nstreams == 0:
        core.prepare_facet_columns(bands, off0s, band, k, rows[k][0], rows[k][1], out=Q[k])
        return
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    for n, j in enumerate(range(0, F, group)):
        s = side[n % nstreams]
        if n < nstreams:
            s.wait_event(ev)
        with torch.cuda.stream(s):
            core.prepare_facet_columns(bands[j:j + group], off0s[j:j + group], band, k, rows[k][0], rows[k][1],
                                       out=Q[k][j:j + group])
    for s in side[:nstreams]:
        e = torch.cuda.Event()
        e.record(s)
        cur.wait_event(e)


def subgrid_side(k):
    w = waves[k]
    return sw.api._finish_from_columns(core, Q[k], 1, facet_cfgs, w, [sg.off0 for sg in w], rowmap=rows[k][0])  # pylint: disable=protected-access


def loop(nstreams, overlap):
    main = torch.cuda.current_stream()
    keep = []
    for k in keys:
        k2_wave(k, nstreams)
        if not overlap:
            keep.append(subgrid_side(k))
            continue
        e = torch.cuda.Event()
        e.record(main)
        sub.wait_event(e)
        with torch.cuda.stream(sub):
            keep.append(subgrid_side(k))
    if overlap:
        e = torch.cuda.Event()
        e.record(sub)
        main.wait_event(e)
    return keep


def k2_only(nstreams, group=1):
    for k in keys:
        k2_wave(k, nstreams, group)


print("K2 only, ms per pass (25 waves):", flush=True)
for ns, g in ((0, 1), (2, 1), (3, 1), (4, 1), (2, 2)):
    print(f"  streams {ns} facets/call {g if ns else F}: {timed(lambda: k2_only(ns, g)):7.3f}", flush=True)
print("wave loop (K2 + K3 + sum_finish + K5b), ms per pass:", flush=True)
for ns in (0, 2, 3):
    for ov in (False, True):
        print(f"  K2 streams {ns}, subgrid side on its own stream {ov}: {timed(lambda: loop(ns, ov)):7.3f}", flush=True)
