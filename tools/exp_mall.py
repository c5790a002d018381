"""Infinity-Cache experiments on the per-wave stages of the 64k-sparse workload (r4 session 2), all through the
existing entry points, random band buffers (timing only):

  K2   -- one call for all 9 facets (default) | one call per facet on one stream (134 MB scratch, cacheable, reused) |
          per facet alternating between two streams (tails of one facet's passes under the other's)
  K3-5 -- the subgrid side of a wave for all its subgrids at once (default) | in chunks of 10 / 7 / 5 / 3 subgrids, so
          that G (2 MB per facet and subgrid) and the half-finished subgrids stay below the 256 MiB cache

Run it with the default library and with variants/ntoff.so (column passes with cacheable instead of non-temporal
accesses):  SWIFTLY_HIP_LIB=variants/ntoff.so python tools/exp_mall.py
"""
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
keys = sorted(waves)[8:14]  # six central waves (the longest ones)
band = core.band_for_offsets([sg.off1 for sg in sgs])
bands = torch.randn((F, yB, core.band_columns(band)), dtype=torch.complex64, device="cuda")
tag = os.environ.get("SWIFTLY_HIP_LIB", "default")
print(tag, "waves", keys, "subgrids per wave", [len(waves[k]) for k in keys], flush=True)


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
    return 1e3 * best / len(keys)  # us per wave


rows = {k: core.subgrid_column_rows([sg.off0 for sg in waves[k]]) for k in keys}
Q = {k: torch.empty((F, rows[k][1], m), dtype=torch.complex64, device="cuda") for k in keys}


def k2_all():
    for k in keys:
        core.prepare_facet_columns(bands, off0s, band, k, rows[k][0], rows[k][1], out=Q[k])


def k2_per_facet():
    for k in keys:
        for j in range(F):
            core.prepare_facet_columns(bands[j:j + 1], off0s[j:j + 1], band, k, rows[k][0], rows[k][1], out=Q[k][j:j + 1])


side = [torch.cuda.Stream(), torch.cuda.Stream()]


def k2_two_streams(group=1):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    for s in side:
        s.wait_event(ev)
    n = 0
    for k in keys:
        for j in range(0, F, group):
            with torch.cuda.stream(side[n % 2]):
                core.prepare_facet_columns(bands[j:j + group], off0s[j:j + group], band, k, rows[k][0], rows[k][1],
                                           out=Q[k][j:j + group])
            n += 1
    for s in side:
        e = torch.cuda.Event()
        e.record(s)
        main.wait_event(e)


print(f"{tag} K2 all facets per call          : {timed(k2_all):8.1f} us per wave", flush=True)
print(f"{tag} K2 per facet, one stream        : {timed(k2_per_facet):8.1f} us per wave", flush=True)
print(f"{tag} K2 per facet, two streams       : {timed(k2_two_streams):8.1f} us per wave", flush=True)
print(f"{tag} K2 3 facets per call, 2 streams : {timed(lambda: k2_two_streams(3)):8.1f} us per wave", flush=True)
k2_all()
torch.cuda.synchronize()


def subgrid_side(chunk):
    for k in keys:
        w = waves[k]
        for i in range(0, len(w), chunk):
            part = w[i:i + chunk]
            sw.api._finish_from_columns(core, Q[k], 1, facet_cfgs, part, [sg.off0 for sg in part], rowmap=rows[k][0])  # pylint: disable=protected-access


for chunk in (64, 10, 7, 5, 3):
    print(f"{tag} K3-5 in chunks of {chunk:2d} subgrids : {timed(lambda: subgrid_side(chunk)):8.1f} us per wave", flush=True)
