"""
GPU parity at the sizes of the other BASELINE.json configurations and for sparse FACET lists (rows g1, f2),
complex64, against the separable oracle (oracle/separable.py) on a few facets / subgrids each:

  * config 3 parameters (W=11, N=32768, yB=4096, yN=8192, xA=2048, xM=4096 -> m=1024): both the general launch
    sequence and (r3) the fused contiguous-axis-first pipeline;
  * config 5, catalogue 128k[1]-n64k-1k (N=131072, yB=45056, yN=65536, xM=1024, m=512): yN = 65536 runs through the
    contiguous-axis-first pipeline (band row kernel at 2 x 32768 points, 256 x 256 column passes);
  * a sparse facet list (scripts/demo_sparse_facet.py:34-134 style: only the facets that intersect a region) on the
    fused path: facets that do not form an off0 x off1 grid.
"""
import numpy
import pytest

import bench
from oracle import separable as sep
from oracle import swiftly_oracle as orc

pytestmark = pytest.mark.gpu


def relrms(a, b):
    return float(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2)))


def _run(P, facet_cfgs, sg_cfgs, wave_axis, tol, plan=True, seed=900):
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    cfg = sw.SwiftlyConfig(backend="hip", **P)
    yB = P["yB_size"]
    vectors = [sep.facet_vectors(seed + j, yB, rank=2) for j in range(len(facet_cfgs))]
    facets = [bench.separable_facet(torch, vectors[j], c) for j, c in enumerate(facet_cfgs)]
    if wave_axis is None:
        wave_axis = sw.api.preferred_wave_axis(cfg, torch.complex64)
    key = (lambda c: c.off1) if wave_axis == 1 else (lambda c: c.off0)
    ordered = sorted(sg_cfgs, key=lambda c: (key(c), c.off0, c.off1))
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs if plan else None,
                            wave_axis=wave_axis)
    dev = fwd.get_subgrid_tasks(ordered)
    got = [t.cpu().numpy() for t in dev]
    # the subgrid -> facet direction at the same sizes, tied to the forward pass (checked against the oracle below)
    # by the adjoint identity of tests/test_adjoint_cpu.py with y = forward(x): sum |F x|^2 == N^-2 sum conj(B y) x
    lhs = sum(float(t.to(torch.complex128).abs().pow(2).sum()) for t in dev)
    for baxis in (0, 1):
        bkey = (lambda c: c.off1) if baxis == 1 else (lambda c: c.off0)
        order = sorted(range(len(ordered)), key=lambda i: (bkey(ordered[i]), ordered[i].off0, ordered[i].off1))
        bwd = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=baxis, subgrid_configs=sg_cfgs if plan else None)
        bwd.add_new_subgrid_tasks([ordered[i] for i in order], [dev[i] for i in order])
        out = bwd.finish()
        rhs = sum(torch.sum(b.to(torch.complex128).conj() * x.to(torch.complex128)).item() for b, x in zip(out, facets))
        rhs /= float(P["N"]) ** 2
        print(f"N={P['N']} backward wave_axis={baxis}: |F x|^2 = {lhs:.6e}, N^-2 <B F x, x> = {rhs:.6e}")
        assert abs(lhs - rhs) <= 1e-5 * lhs, (baxis, lhs, rhs)
        del bwd, out
    ref = orc.OracleCore(P["W"], P["N"], P["xM_size"], P["yN_size"])
    items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    so = sep.SeparableOracle(ref, items, vectors)
    errs = []
    for g, c in zip(got, ordered):
        w = so.subgrid(orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1))
        assert g.dtype == numpy.complex64 and g.shape == w.shape
        errs.append(relrms(g, w))
    print(f"N={P['N']} wave_axis={wave_axis}: relRMSE {['%.2e' % e for e in errs]}")
    assert max(errs) < tol, errs
    return wave_axis


@pytest.mark.parametrize("wave_axis", [0, 1])
def test_config3_sizes_m1024_xM4096(wave_axis):
    """wave_axis=0: the general launch sequence; wave_axis=1 (r3): the contiguous-axis-first pipeline with the
    generic K1 for yN = 8192 (plain band layout), the single-pass 1024-point column transform for m = 1024 and the
    two-waves-per-row sum_finish / split_prepare instances for xM = 4096 -- forward against the oracle, both backward
    schedules through the adjoint identity."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    P = dict(W=11.0, fov=1.0, N=32768, yB_size=4096, yN_size=8192, xA_size=2048, xM_size=4096)
    cover = sw.api.make_full_cover_config(P["N"], P["yB_size"], sw.FacetConfig)
    facet_cfgs = [c for c in cover if (c.off0, c.off1) in ((0, 0), (4096, 4096 * 7))]
    sgs = sw.api.make_full_cover_config(P["N"], P["xA_size"], sw.SubgridConfig)
    sg_cfgs = [c for c in sgs if (c.off0 // 2048, c.off1 // 2048) in ((0, 0), (0, 15), (3, 2))]
    if wave_axis == 1:
        cfg = sw.SwiftlyConfig(backend="hip", **P)
        assert sw.api.preferred_wave_axis(cfg, torch.complex64, n_facets=64) == 1
        assert cfg.core.band_for_offsets([0, 2048]) == (0, 8192) and cfg.core.band_columns((0, 8192)) == 8192
    _run(P, facet_cfgs, sg_cfgs, wave_axis, 2e-5, plan=wave_axis == 1)


def test_config5_sizes_yN65536():
    import ska_sdp_exec_swiftly_amd as sw

    P = dict(W=10.875, fov=1.0, N=131072, yB_size=45056, yN_size=65536, xA_size=928, xM_size=1024)
    cover = sw.api.make_full_cover_config(P["N"], P["yB_size"], sw.FacetConfig)
    facet_cfgs = [c for c in cover if (c.off0, c.off1) in ((0, 45056), (90112, 0))]
    sgs = sw.api.make_full_cover_config(P["N"], P["xA_size"], sw.SubgridConfig)
    want = ((0, 0), (1, 0), (140, 3), (7, 141))
    sg_cfgs = [c for c in sgs if (c.off0 // 928, c.off1 // 928) in want]
    assert len(sg_cfgs) == len(want)
    axis = _run(P, facet_cfgs, sg_cfgs, None, 2e-5)
    assert axis == 1  # the yN = 65536 kernels exist only in the contiguous-axis-first pipeline


@pytest.mark.parametrize("wave_axis", [0, 1])
def test_sparse_facet_list_on_fused_path(wave_axis):
    """Five of the nine facets of the 64k configuration (a plus shape: no off0 x off1 grid) with small-ish data
    (yB = 1024 keeps the test cheap; yN = 32768 so that both pipelines exist)."""
    import ska_sdp_exec_swiftly_amd as sw

    yB = 1024
    P = dict(W=10.875, fov=1.0, N=65536, yB_size=yB, yN_size=32768, xA_size=928, xM_size=1024)
    offs = [(0, 0), (0, 22528), (22528, 0), (0, 45056), (45056, 0)]
    rng = numpy.random.default_rng(3)
    facet_cfgs = [sw.FacetConfig(o0, o1, yB, (rng.random(yB) > 0.05).astype(float), None) for o0, o1 in offs]
    sg_cfgs = [sw.SubgridConfig(i0 * 928, i1 * 928, 928) for i0, i1 in ((0, 0), (0, 2), (3, 0), (3, 2), (70, 2))]
    _run(P, facet_cfgs, sg_cfgs, wave_axis, 2e-5)


def _full_cover_wave(P, facet_cfgs, wave, check_idx, bwd_facets, tol_f, tol_b, seed=1200, column_precision=None):
    """One WHOLE wave (all subgrids of the cover that share an ``off1``) through the default pipelines with the given
    facets: forward against the separable oracle on the subgrids ``check_idx`` of the wave, backward (separable
    subgrids of the whole wave) element by element on sampled rows of the facets ``bwd_facets``."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    cfg = sw.SwiftlyConfig(backend="hip", column_precision=column_precision, **P)
    yB = P["yB_size"]
    vectors = [sep.facet_vectors(seed + j, yB, rank=2) for j in range(len(facet_cfgs))]
    facets = [bench.separable_facet(torch, vectors[j], c) for j, c in enumerate(facet_cfgs)]
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=wave)
    assert fwd.wave_axis == 1
    got = fwd.get_subgrid_tasks(wave)
    par = bench.verify_subgrids(P, facet_cfgs, vectors, wave, {i: got[i].cpu().numpy() for i in check_idx}, tol=tol_f)
    print(f"N={P['N']} full-cover wave of {len(wave)} subgrids, {len(facet_cfgs)} facets: forward {par['rel_rmse_each']}")
    assert par["ok"], par["rel_rmse_each"]
    del fwd, got, facets
    torch.cuda.empty_cache()
    sg_vectors = [sep.subgrid_vectors(seed + 500 + i, c.size, rank=1) for i, c in enumerate(wave)]
    subgrids = [bench.separable_facet(torch, sg_vectors[i], c) for i, c in enumerate(wave)]
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, subgrid_configs=wave)
    bwd.add_new_subgrid_tasks(wave, subgrids)
    assert bwd.wave_axis == 1
    out = bwd.finish()
    bpar = bench.verify_facets(P, [facet_cfgs[j] for j in bwd_facets], wave, sg_vectors, [out[j] for j in bwd_facets],
                               rows_per_facet=6, tol=tol_b)
    print(f"    backward rows of facets {bwd_facets}: {bpar['rel_rmse_each']}")
    assert bpar["ok"], bpar["rel_rmse_each"]


def test_config3_full_cover_wave_all_64_facets():
    """BASELINE config 3 at FULL facet cover (r3 review: only 2 of the 64 facets were driver-checked): 8 x 8 facets ->
    the 16 subgrids of one off1 column.  The subgrid side sums 64 facets in 8 off1 groups per padded row through the
    wave-parallel sum_finish / split_prepare instances <10, 12> with their host-side colouring rounds; yN = 8192 runs
    the generic K1, m = 1024 the single-pass column transform."""
    import ska_sdp_exec_swiftly_amd as sw

    P = dict(W=11.0, fov=1.0, N=32768, yB_size=4096, yN_size=8192, xA_size=2048, xM_size=4096)
    facet_cfgs = sw.api.make_full_cover_config(P["N"], P["yB_size"], sw.FacetConfig)
    assert len(facet_cfgs) == 64
    sgs = sw.api.make_full_cover_config(P["N"], P["xA_size"], sw.SubgridConfig)
    wave = [c for c in sgs if c.off1 == 5 * 2048]
    assert len(wave) == 16
    _full_cover_wave(P, facet_cfgs, wave, check_idx=[0, 7, 15], bwd_facets=[0, 13, 36, 63], tol_f=2e-5, tol_b=4e-5)


@pytest.mark.parametrize("bits", [32, 64])
def test_config5_full_wave_yN65536(bits):
    """BASELINE config 5 (catalogue 128k[1]-n64k-1k, yN = 65536) on the two facets a rank of an 8-GPU node holds: one
    whole wave (142 subgrids) forward against the oracle, and -- instead of the adjoint identity of r3 -- its backward
    pass element by element against the separable backward oracle; in both column precisions."""
    import ska_sdp_exec_swiftly_amd as sw

    P = dict(W=10.875, fov=1.0, N=131072, yB_size=45056, yN_size=65536, xA_size=928, xM_size=1024)
    cover = sw.api.make_full_cover_config(P["N"], P["yB_size"], sw.FacetConfig)
    facet_cfgs = [c for c in cover if (c.off0, c.off1) in ((0, 45056), (90112, 0))]
    sgs = sw.api.make_full_cover_config(P["N"], P["xA_size"], sw.SubgridConfig)
    wave = [c for c in sgs if c.off1 == 3 * 928]
    assert len(wave) == 142
    # column_precision = 64: float64 arithmetic in the 256 x 256 passes of K2, in K3 and in their backward mirrors
    tol_f, tol_b = (2e-5, 4e-5) if bits == 32 else (5e-6, 1e-5)
    _full_cover_wave(P, facet_cfgs, wave, check_idx=[0, 70, 141], bwd_facets=[0, 1], tol_f=tol_f, tol_b=tol_b,
                     column_precision=bits)
