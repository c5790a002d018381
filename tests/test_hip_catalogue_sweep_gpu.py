"""
A sweep over the reference's parameter catalogue (swift_configs.py, 244 entries): one entry of every (odd factor of
yN, xM, m) kind and of every fused-kernel size pair, run through the streaming classes WITH THEIR DEFAULTS (the
pipeline `preferred_wave_axis` picks: fused band pipelines where the sizes have them, general launch sequences
elsewhere) on two facets of the entry's cover and a handful of subgrids, forward against oracle/separable.py (exact
for separable facets at any size) and backward against its backward counterpart.  Catches size-specific mistakes
(geometry tables, modulus arithmetic, band layouts, scratch sizing) that the per-kernel tests of fixed sizes cannot.
"""
import numpy
import pytest

import bench
from oracle import separable as sep
from oracle import swiftly_oracle as orc

pytestmark = pytest.mark.gpu

# key -> expected forward wave axis with a plan (1 = fused band pipeline)
SWEEP = {
    "1k[1]-n512-256": 1,       # all powers of two, smallest fused pair (m, xM) = (128, 256)
    "4k[1]-n2k-512": 1,        # (256, 512)
    "2k[1]-n1k-512": 1,        # yN = 1024: the backward gather-sum transform must take the four-step (r3 bug: the
                               # single-pass 32-column geometry has no gather-sum load and read the table as a row map)
    "16k[1]-n8k-1k": 1,        # (512, 1024), generic K1, plain band
    "32k[1]-n16k-1k": 1,       # two-workgroup K1 at 16384, split band
    "3k[1]-n1536-512": 1,      # yN = 3 * 512
    "5k[1]-n2560-512": 1,      # yN = 5 * 512
    "7k[1]-n3584-512": 1,      # yN = 7 * 512
    "72k[1]-n36k-512": 1,      # yN = 9 * 4096 (the catalogue's only factor 9), four-step sub-transforms
    "20k[1]-n10k-1k": 1,       # yN = 5 * 2048, (512, 1024)
    "12k[1]-n6k-384": 0,       # xM = 384, m = 192: general launch sequences (radix-3 pass on every length)
    "10k[1]-n5k-320": 0,       # xM = 320, m = 160
    "14k[1]-n7k-448": 0,       # xM = 448, m = 224
}


def relrms(a, b):
    return float(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2)))


@pytest.mark.parametrize("key", sorted(SWEEP))
def test_catalogue_entry_default_pipelines(key):
    import torch

    import ska_sdp_exec_swiftly_amd as sw
    from ska_sdp_exec_swiftly_amd.swift_configs import SWIFT_CONFIGS

    if key not in SWIFT_CONFIGS:
        pytest.skip(f"{key} not in the catalogue")
    p = {k: SWIFT_CONFIGS[key][k] for k in ("W", "fov", "N", "yB_size", "yN_size", "xA_size", "xM_size")}
    cfg = sw.SwiftlyConfig(backend="hip", **p)
    cover = sw.make_full_facet_cover(cfg)
    facet_cfgs = [cover[0], cover[len(cover) // 2]]  # two facets with different offsets along both axes
    step = cfg.core.subgrid_off_step
    all_sgs = sw.make_full_subgrid_cover(cfg)
    if p["xA_size"] % step:
        # an odd xA_size (12k[1]-n6k-384: 345) gives cover offsets that are not multiples of the subgrid offset step
        # N / yN and empty masks from the reference's cover rule (api.py:593-612): hand-placed subgrids instead, with
        # partial masks on some of them
        xA, N = p["xA_size"], p["N"]
        ones = numpy.ones(xA)
        part = numpy.concatenate([numpy.zeros(xA // 3), numpy.ones(xA - xA // 3)])
        all_sgs = [sw.SubgridConfig(o0, o1, xA, m0, m1) for o0, o1, m0, m1 in (
            (0, 0, ones, ones), (0, 2 * (xA // 2), ones, part), (step * (N // 3 // step), 0, part, ones),
            (step * (N // 2 // step), 2 * (xA // 2), part, part), (N - 2 * (xA // 2), N - 2 * (xA // 2), ones, ones))]
    picks = [all_sgs[i] for i in sorted({0, 1, len(all_sgs) // 3, len(all_sgs) // 2 + 1, len(all_sgs) - 1})]
    vectors = [sep.facet_vectors(4000 + j, p["yB_size"], rank=2) for j in range(len(facet_cfgs))]
    facets = [bench.separable_facet(torch, vectors[j], facet_cfgs[j]) for j in range(len(facet_cfgs))]
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=picks)
    assert fwd.wave_axis == SWEEP[key], (key, fwd.wave_axis)
    got = fwd.get_subgrid_tasks(picks)
    par = bench.verify_subgrids(p, facet_cfgs, vectors, picks, {i: g.cpu().numpy() for i, g in enumerate(got)})
    assert par["rel_rmse"] < 2e-5, (key, par["rel_rmse_each"])

    # backward: separable subgrids -> sampled rows of both facets, element by element
    sg_vectors = [sep.subgrid_vectors(5000 + i, c.size, rank=1) for i, c in enumerate(picks)]
    subgrids = [bench.separable_facet(torch, sg_vectors[i], c) for i, c in enumerate(picks)]  # u (x) v times the masks
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, subgrid_configs=picks)
    order = sorted(range(len(picks)), key=lambda k: (picks[k].off1, picks[k].off0))
    bwd.add_new_subgrid_tasks([picks[k] for k in order], [subgrids[k] for k in order])
    out = bwd.finish()
    bpar = bench.verify_facets(p, facet_cfgs, picks, sg_vectors, out, rows_per_facet=8)
    # W > 11 (here 11.125: max 1/pswf = 113 instead of 90) raises the float32 floor of the facet-side windows
    assert bpar["rel_rmse"] < (4e-5 if p["W"] <= 11.0 else 1e-4), (key, bpar["rel_rmse_each"])
