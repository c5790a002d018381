"""
GPU parity of the kernels of the contiguous-axis-first forward pipeline
(include/swiftly_hip.h: prepare_facet_band, prepare_facet_columns,
transform_contributions, sum_finish_facets) against the oracle, primitive by
primitive, plus the whole pipeline against the oracle replica of the reference
dataflow.  complex64; tolerances: relative RMSE 2e-6 for a single transform of
un-amplified data, 2e-5 end to end.
"""
import numpy
import pytest

import bench
from oracle import separable as sep
from oracle import swiftly_oracle as orc

pytestmark = pytest.mark.gpu

W64, N64, xM64, yN64, yB64 = 10.875, 65536, 1024, 32768, 22528


def relrms(a, b):
    return float(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2)))


def band_cols(yN, band):
    """physical column of every logical (centred) column, -1 outside the band (parity-split layout)"""
    start, length = band
    half = ((length + 1) // 2 + 15) // 16 * 16  # swiftly_hip_band_columns(length) / 2
    d = (numpy.arange(yN) - start) % yN
    return numpy.where(d < length, (d & 1) * half + (d >> 1), -1)


_core64 = {}


def core64():
    from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip

    if "c" not in _core64:
        _core64["c"] = SwiftlyCoreHip(W64, N64, xM64, yN64)
        _core64["o"] = orc.OracleCore(W64, N64, xM64, yN64)
    return _core64["c"], _core64["o"]


@pytest.mark.parametrize("band", [(0, yN64), (10736, 11472), (32001, 2049)])
def test_prepare_facet_band(band):
    import torch

    core, ref = core64()
    assert core.supports_band_pipeline(torch.complex64)
    rng = numpy.random.default_rng(21)
    rows = 7
    x = (rng.standard_normal((rows, yB64)) + 1j * rng.standard_normal((rows, yB64))).astype(numpy.complex64)
    xt = torch.from_numpy(x).cuda()
    pc = band_cols(yN64, band)
    # offsets: the three of the 3x3 cover (22 of the 32 input segments hold data, three rotations: the NSEG = 22 kernel),
    # one that is not a multiple of the 1024-point segment (23 segments: NSEG = 24), an odd one (8-byte load path)
    for off, fold in ((0, False), (64 * 352, True), (-64 * 352, False), (-64 * 320, True), (64 * 351, False), (64 * 352 + 2, True), (4097, False)):
        got = core.prepare_facet_band(xt, off, band, fold_other_axis_window=fold).cpu().numpy()
        want = ref.prepare_facet(x.astype(complex), off, 1)
        if fold:
            want = want * ref.facet_window(rows)[:, None]
        assert got.shape == (rows, core.band_columns(band))
        keep = pc >= 0
        rel = relrms(got[:, pc[keep]], want[:, keep])
        assert rel < 2e-6, (band, off, rel)


def test_prepare_facet_band_16_segment_facets():
    """K1 on facets of 16384 columns in 32768-point rows (the 4x4 / 8x8 custom covers of bench.py): exactly 16 of the 32
    load segments hold data at aligned offsets -- the NSEG = 16 instances, whose re-laid-out window has no (r, r + 16)
    segment pairs at all -- and 17 at an unaligned one (NSEG = 22 with empty tail segments)."""
    import torch

    core, ref = core64()
    rng = numpy.random.default_rng(23)
    rows, yB = 5, 16384
    x = (rng.standard_normal((rows, yB)) + 1j * rng.standard_normal((rows, yB))).astype(numpy.complex64)
    xt = torch.from_numpy(x).cuda()
    band = (10736, 11472)
    pc = band_cols(yN64, band)
    keep = pc >= 0
    for off in (0, 16384, -16384, 3 * 8192, 16384 + 2 * 333):
        got = core.prepare_facet_band(xt, off, band, fold_other_axis_window=False).cpu().numpy()
        want = ref.prepare_facet(x.astype(complex), off, 1)
        assert got.shape == (rows, core.band_columns(band))
        rel = relrms(got[:, pc[keep]], want[:, keep])
        assert rel < 2e-6, (off, rel)


def test_band_for_offsets_and_supports():
    core, _ = core64()
    offs = [i * 928 for i in list(range(0, 13)) + list(range(59, 71))]
    start, length = core.band_for_offsets(offs)
    m, yN = core.xM_yN_size, core.yN_size
    cover = numpy.zeros(yN, dtype=bool)
    cover[(start + numpy.arange(length)) % yN] = True
    for off in offs:
        s = off * yN // core.N
        assert cover[(yN // 2 - m // 2 + numpy.arange(m) + s) % yN].all()
    assert length < yN // 2
    assert core.band_for_offsets([i * 928 for i in range(71)]) == (0, yN)


@pytest.mark.parametrize("band,use_rowmap", [((0, yN64), False), ((10736, 11472), True)])
def test_prepare_facet_columns(band, use_rowmap):
    """K2: window gather from band buffers + strided-axis prepare_facet (no window) for 2 facets at once."""
    import torch

    core, ref = core64()
    m, yN = core.xM_yN_size, core.yN_size
    rng = numpy.random.default_rng(22)
    yB0 = 176  # facet size along axis 0 (any size < yN works for the primitive)
    F = 2
    ncols = core.band_columns(band)
    pc = band_cols(yN, band)
    logical = (rng.standard_normal((F, yB0, yN)) + 1j * rng.standard_normal((F, yB0, yN))).astype(numpy.complex64)
    packed = numpy.zeros((F, yB0, ncols), dtype=numpy.complex64)
    packed[:, :, pc[pc >= 0]] = logical[:, :, pc >= 0]
    bands = torch.from_numpy(packed).cuda()
    off0s = [0, 22528]
    sub_off0s = [0, 3 * 928, -5 * 928]
    rowmap, n_rows = core.subgrid_column_rows(sub_off0s) if use_rowmap else (None, yN)
    rm = rowmap.cpu().numpy() if rowmap is not None else numpy.arange(yN)
    for off1 in (7 * 928, -11 * 928):
        got = core.prepare_facet_columns(bands, off0s, band, off1, rowmap, n_rows).cpu().numpy()
        assert got.shape == (F, n_rows, m)
        for f in range(F):
            win = ref.extract_from_facet(logical[f].astype(complex), off1, axis=1)  # [yB0, m]
            want = ref.prepare_facet(win / ref.facet_window(yB0)[:, None], off0s[f], axis=0)  # window NOT applied
            keep = rm >= 0
            rel = relrms(got[f][rm[keep]], want[keep])
            assert rel < 2e-6, (off1, f, rel)


P11 = dict(W=11.0, N=1024, yB=352, yN=512, xA=192, xM=256)  # m = 128: sum_finish instance (7, 8)


def _p11():
    from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip

    return (
        SwiftlyCoreHip(P11["W"], P11["N"], P11["xM"], P11["yN"]),
        orc.OracleCore(P11["W"], P11["N"], P11["xM"], P11["yN"]),
    )


def _G_want(ref, C, off0):
    """Fn * cfft_m(C, axis 0) rotated by the facet offset, without placement: rows of add_to_subgrid(axis 0)."""
    m, xM = ref.xM_yN_size, ref.xM_size
    placed = ref.add_to_subgrid(C, off0, axis=0)
    sp = off0 * xM // ref.N
    return placed[(numpy.arange(m) + xM // 2 - m // 2 + sp) % xM]


def test_transform_contributions_all_layouts():
    import torch

    core, ref = _p11()
    m, yN, N = core.xM_yN_size, P11["yN"], P11["N"]
    rng = numpy.random.default_rng(23)
    fstep, sstep = core.facet_off_step, core.subgrid_off_step
    foffs = [0, 88 * fstep, -40 * fstep]
    soffs = [0, 96 * sstep, -17 * sstep, N // 2 + 10 * sstep]
    F, S = len(foffs), len(soffs)
    # layout 0: column buffers [F, m, yN], windows along the contiguous axis
    cols = (rng.standard_normal((F, m, yN)) + 1j * rng.standard_normal((F, m, yN))).astype(numpy.complex64)
    G = core.transform_contributions(torch.from_numpy(cols).cuda(), 0, foffs, soffs).cpu().numpy()
    assert G.shape == (F, S, m, m)
    for f in range(F):
        for b in range(S):
            C = ref.extract_from_facet(cols[f].astype(complex), soffs[b], axis=1)
            assert relrms(G[f, b], _G_want(ref, C, foffs[f])) < 2e-6
    # layout 1: [F, rows, m] with a row map, windows along the strided axis
    rowmap, n_rows = core.subgrid_column_rows(soffs)
    rm = rowmap.cpu().numpy()
    full = (rng.standard_normal((F, yN, m)) + 1j * rng.standard_normal((F, yN, m))).astype(numpy.complex64)
    compact = numpy.zeros((F, n_rows, m), dtype=numpy.complex64)
    compact[:, rm[rm >= 0]] = full[:, rm >= 0]
    G1 = core.transform_contributions(torch.from_numpy(compact).cuda(), 1, foffs, soffs, rowmap=rowmap).cpu().numpy()
    G1b = core.transform_contributions(torch.from_numpy(full).cuda(), 1, foffs, soffs).cpu().numpy()
    for f in range(F):
        for b in range(S):
            C = ref.extract_from_facet(full[f].astype(complex), soffs[b], axis=0)
            want = _G_want(ref, C, foffs[f])
            assert relrms(G1[f, b], want) < 2e-6
            assert relrms(G1b[f, b], want) < 2e-6
    # layout 2: materialised contributions [F, S, m, m]
    contrib = (rng.standard_normal((F, S, m, m)) + 1j * rng.standard_normal((F, S, m, m))).astype(numpy.complex64)
    G2 = core.transform_contributions(torch.from_numpy(contrib).cuda(), 2, foffs, None, nsub=S).cpu().numpy()
    for f in range(F):
        for b in range(S):
            assert relrms(G2[f, b], _G_want(ref, contrib[f, b].astype(complex), foffs[f])) < 2e-6


def test_sum_finish_facets():
    import torch

    core, ref = _p11()
    m, xM, xA, N = core.xM_yN_size, P11["xM"], P11["xA"], P11["N"]
    rng = numpy.random.default_rng(24)
    fstep, sstep = core.facet_off_step, core.subgrid_off_step
    f_offs = [(0, 0), (0, 88 * fstep), (88 * fstep, 0), (88 * fstep, 88 * fstep), (-44 * fstep, 20 * fstep)]
    s_off1 = [0, 96 * sstep, -33 * sstep]
    F, S = len(f_offs), len(s_off1)
    contrib = (rng.standard_normal((F, S, m, m)) + 1j * rng.standard_normal((F, S, m, m))).astype(numpy.complex64)
    G = core.transform_contributions(torch.from_numpy(contrib).cuda(), 2, [o[0] for o in f_offs], None, nsub=S)
    mask = (rng.random((S, xA)) > 0.2).astype(numpy.float32)
    out = torch.empty((S, xM, xA), dtype=torch.complex64, device="cuda")
    core.sum_finish_facets(G, [o[0] for o in f_offs], [o[1] for o in f_offs], out, s_off1, xA,
                           mask=torch.from_numpy(mask).cuda())
    got = out.cpu().numpy()
    for b in range(S):
        acc = numpy.zeros((xM, xM), dtype=complex)
        for f, (o0, o1) in enumerate(f_offs):
            acc += ref.add_to_subgrid(ref.add_to_subgrid(contrib[f, b].astype(complex), o0, 0), o1, 1)
        want = numpy.array([ref.finish_subgrid(acc[r], s_off1[b], xA) for r in range(xM)]) * mask[b][None, :]
        rel = relrms(got[b], want)
        assert rel < 3e-6, (b, rel)


@pytest.mark.parametrize("axis", [0, 1])
def test_forward_pipelines_match_oracle_small_rows(axis):
    """Both forward pipelines through SwiftlyForward at yN = 32768 with SMALL facets (yB = 352 so that the
    2-D oracle is cheap): 4 facets, planned sparse subgrid set, complex64."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    yB, xA = 352, 928
    P = dict(W=W64, fov=1.0, N=N64, yB_size=yB, yN_size=yN64, xA_size=xA, xM_size=xM64)
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    if axis == 1 and sw.api.preferred_wave_axis(cfg, torch.complex64) != 1:
        pytest.skip("band pipeline not available")
    ref = core64()[1]
    fstep = cfg.facet_off_step
    rng = numpy.random.default_rng(25)
    facet_cfgs = [
        sw.FacetConfig(o0, o1, yB, (rng.random(yB) > 0.1).astype(float), None)
        for o0, o1 in ((0, 0), (0, 5 * fstep * 10), (-7 * fstep * 10, 0), (-7 * fstep * 10, 5 * fstep * 10))
    ]
    # separable dense facets (oracle/separable.py): the exact result per subgrid costs O(0.5 s) on the CPU
    vectors = [sep.facet_vectors(500 + j, yB, rank=2) for j in range(len(facet_cfgs))]
    facets = [bench.separable_facet(torch, vectors[j], c) for j, c in enumerate(facet_cfgs)]
    sg_cfgs = [
        sw.SubgridConfig(i0 * xA, i1 * xA, xA, None, (rng.random(xA) > 0.1).astype(float))
        for i0, i1 in ((0, 0), (0, 3), (2, 0), (2, 3), (69, 3), (69, 70), (2, 70))
    ]
    key = (lambda c: c.off1) if axis == 1 else (lambda c: c.off0)
    order = sorted(range(len(sg_cfgs)), key=lambda i: (key(sg_cfgs[i]), i))
    ordered = [sg_cfgs[i] for i in order]
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=axis, lru_forward=2)
    got = [t.cpu().numpy() for t in fwd.get_subgrid_tasks(ordered)]
    items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    so = sep.SeparableOracle(ref, items, vectors)
    for g, c in zip(got, ordered):
        w = so.subgrid(orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1))
        rel = relrms(g, w)
        assert rel < 2e-5, (axis, c.off0, c.off1, rel)
    # unplanned subgrid is refused
    with pytest.raises(ValueError):
        fwd.get_subgrid_task(sw.SubgridConfig(5 * xA, 5 * xA, xA))


def _small_rows_problem(seed=41):
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    yB, xA = 352, 928
    P = dict(W=W64, fov=1.0, N=N64, yB_size=yB, yN_size=yN64, xA_size=xA, xM_size=xM64)
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    fstep = cfg.facet_off_step
    facet_cfgs = [sw.FacetConfig(o0, o1, yB) for o0, o1 in ((0, 0), (0, 50 * fstep), (-70 * fstep, 0))]
    vectors = [sep.facet_vectors(seed + j, yB, rank=2) for j in range(len(facet_cfgs))]
    facets = [bench.separable_facet(torch, vectors[j], c) for j, c in enumerate(facet_cfgs)]
    # a 3 x 4 block of subgrids in the reference's natural (off0-major) order
    sg_cfgs = [sw.SubgridConfig(i0 * xA, i1 * xA, xA) for i0 in (0, 2, 69) for i1 in (0, 3, 4, 70)]
    return torch, sw, cfg, facet_cfgs, facets, sg_cfgs


def test_forward_default_axis_with_plan_serves_any_order():
    """SwiftlyForward(subgrid_configs=plan) without wave_axis picks the contiguous-axis-first pipeline, and the
    reference's one-subgrid-at-a-time off0-major loop costs one wave computation per off1 (finished subgrids of a
    planned wave are cached until they are asked for), with results identical to whole-wave requests."""
    torch, sw, cfg, facet_cfgs, facets, sg_cfgs = _small_rows_problem()
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs)
    if not cfg.core.supports_band_pipeline(torch.complex64):
        assert fwd.wave_axis == 0
        pytest.skip("band pipeline not available")
    assert fwd.wave_axis == 1
    calls = []
    inner = fwd.get_wave
    fwd.get_wave = lambda sgs, timer=None: (calls.append(len(sgs)), inner(sgs, timer))[1]
    got = [fwd.get_subgrid_task(c).cpu().numpy() for c in sg_cfgs]  # off0-major, one at a time
    assert calls == [3, 3, 3, 3]  # one launch sequence per off1 wave, each over the 3 planned subgrids
    assert not fwd._results and fwd._result_bytes == 0  # every cached subgrid was handed out
    ref = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
    by1 = sorted(range(len(sg_cfgs)), key=lambda i: (sg_cfgs[i].off1, i))
    want = ref.get_subgrid_tasks([sg_cfgs[i] for i in by1])
    for k, i in enumerate(by1):
        assert numpy.array_equal(got[i], want[k].cpu().numpy())
    # a second request for a subgrid that was already handed out recomputes (and caches the rest of its wave again)
    again = fwd.get_subgrid_task(sg_cfgs[5]).cpu().numpy()
    assert numpy.array_equal(again, got[5]) and len(fwd._results) == 2
    # without the budget the group is computed on its own
    fwd2 = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs)
    fwd2._result_budget = 0
    one = fwd2.get_subgrid_task(sg_cfgs[5]).cpu().numpy()
    assert not fwd2._results and relrms(one, got[5]) < 1e-6
    # the separable oracle agrees
    ref_core = core64()[1]
    so = sep.SeparableOracle(ref_core, [orc.CoverItem(c.off0, c.off1, c.size) for c in facet_cfgs],
                             [sep.facet_vectors(41 + j, 352, rank=2) for j in range(3)])
    c = sg_cfgs[5]
    assert relrms(got[5], so.subgrid(orc.CoverItem(c.off0, c.off1, c.size))) < 2e-5


@pytest.mark.parametrize("order", ["ascending", "descending", "shuffled"])
def test_forward_planned_wave_prefetch_is_bit_identical(order):
    """The planned-wave prefetch of SwiftlyForward (K2 of the predicted next wave on the core's side stream while the
    subgrid side of the current one runs) does not change a single bit, whatever order the waves are asked for in --
    a wrong prediction only leaves a prefetched buffer unused -- and the oracle agrees with what it serves."""
    torch, sw, cfg, facet_cfgs, facets, sg_cfgs = _small_rows_problem(seed=77)
    if not cfg.core.supports_band_pipeline(torch.complex64):
        pytest.skip("band pipeline not available")
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off1, []).append(c)
    keys = sorted(waves)
    if order == "descending":
        keys = keys[::-1]
    elif order == "shuffled":
        keys = [keys[i] for i in (2, 0, 3, 1)]

    def run(prefetch):
        old = sw.api._PREFETCH
        sw.api._PREFETCH = prefetch
        try:
            fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
            out = {k: fwd.get_wave(waves[k]).cpu().numpy() for k in keys}
            used = fwd.__dict__.get("_prefetched", "never") != "never"
        finally:
            sw.api._PREFETCH = old
        torch.cuda.synchronize()
        return out, used

    plain, used0 = run(False)
    ahead, used1 = run(True)
    assert not used0 and used1  # the second run really went through the prefetch
    for k in keys:
        assert numpy.array_equal(plain[k], ahead[k]), (order, k)
    so = sep.SeparableOracle(core64()[1], [orc.CoverItem(c.off0, c.off1, c.size) for c in facet_cfgs],
                             [sep.facet_vectors(77 + j, 352, rank=2) for j in range(3)])
    c = waves[keys[1]][1]
    assert relrms(ahead[keys[1]][1], so.subgrid(orc.CoverItem(c.off0, c.off1, c.size))) < 2e-5


def test_finish_axis1_rows_matches_oracle():
    """(r6) step R of the axis-1-first pipeline against the oracle primitives: for rows of a K1 band buffer, the window
    gather of extract_from_facet(axis 1) followed by add_to_subgrid(axis 1) (core.py:243-285) -- the latter read back
    from its placement in the padded subgrid -- laid out as a parity-split band that is exactly the wave's window."""
    import torch

    core, ref = core64()
    rng = numpy.random.default_rng(7)
    rows, m, xM = 6, 512, xM64
    band = (10736, 11472)
    pc = band_cols(yN64, band)
    full = (rng.standard_normal((rows, yN64)) + 1j * rng.standard_normal((rows, yN64))).astype(numpy.complex64)
    phys = numpy.zeros((2, rows, core.band_columns(band)), dtype=numpy.complex64)
    keep = pc >= 0
    phys[:, :, pc[keep]] = full[None][:, :, keep]
    bands = torch.from_numpy(phys).cuda()
    facet_off1s = [0, 64 * 352]
    for sub_off1 in (0, 3 * 928, -5 * 928):
        W, wband = core.finish_axis1_rows(bands, facet_off1s, band, sub_off1)
        s = sub_off1 * yN64 // N64
        assert wband == ((yN64 // 2 - m // 2 + s) % yN64, m) and tuple(W.shape) == (2, rows, m)
        got = W.cpu().numpy()
        wpc = band_cols(yN64, wband)
        for f, foff in enumerate(facet_off1s):
            contrib = ref.extract_from_facet(full.astype(complex), sub_off1, 1)          # [rows, m]
            placed = ref.add_to_subgrid(contrib, foff, 1)                                  # [rows, xM]
            sp = foff * xM // N64
            k = numpy.arange(m)
            Z = placed[:, (k + xM // 2 - m // 2 + sp) % xM]                                # Z[k], core.py:274-285
            # logical window element i holds Z[(i + s) mod m], at the parity-split position of logical column c0 + i
            i = numpy.arange(m)
            want_logical = Z[:, (i + s) % m]
            got_logical = got[f][:, wpc[(wband[0] + i) % yN64]]
            rel = relrms(got_logical, want_logical)
            assert rel < 2e-6, (sub_off1, f, rel)


def test_window_rows_store_of_the_whole_row_k1_matches_the_row_pass_per_wave():
    """(r6) the forward K1 with the contiguous-axis finish in its epilogue (swiftly_hip_prepare_facet_window_rows: one
    persistent workgroup per CU owns whole rows, stages the band in LDS and finishes every planned window): for every window
    what finish_axis1_rows (pinned on the oracle above) makes of the band store of the two-workgroup K1 for that wave --
    including windows at the very start / end of the band, odd and even window starts and a band that wraps around."""
    import torch

    core, _ = core64()
    rng = numpy.random.default_rng(17)
    rows, size, m = 300, 352, 512   # more rows than workgroups: the persistent loop (row prefetch, stage reuse) runs
    facet = (rng.standard_normal((rows, size)) + 1j * rng.standard_normal((rows, size))).astype(numpy.complex64)
    dev = torch.from_numpy(facet).cuda()
    for foff, wave_off1s in ((0, [0, 928, 3 * 928, -2 * 928]), (64 * 352, [-928, 5 * 928, 7 * 928 + 2, 12 * 928 + 6]),
                             (-32 * 352, [30 * 928, 33 * 928, 35 * 928 + 4] + [30 * 928 + 2 * k for k in range(1, 9)])):
        band = core.band_for_offsets(wave_off1s)
        assert core.supports_window_rows(band, size, [foff])
        starts = core.window_starts(band, wave_off1s)
        assert min(starts) >= 0 and max(starts) + m <= band[1]
        sd = torch.tensor(starts, dtype=torch.int32, device="cuda")
        # both output layouts: wave-major [windows, rows, m] (the pipeline's) and the windows of a row side by side
        full = torch.full((len(starts), rows, m), float("nan"), dtype=torch.complex64, device="cuda")
        core.prepare_facet_window_rows(dev, foff, band, sd, full)
        got = full.cpu().numpy()
        assert numpy.isfinite(got.view(numpy.float32)).all()
        side = torch.full((rows, len(starts) * m), float("nan"), dtype=torch.complex64, device="cuda")
        core.prepare_facet_window_rows(dev, foff, band, sd, side)
        assert numpy.array_equal(side.cpu().numpy().reshape(rows, len(starts), m).transpose(1, 0, 2), got)
        bands1 = core.prepare_facet_band(dev, foff, band)[None]
        for w, off1 in enumerate(wave_off1s):
            want, wband = core.finish_axis1_rows(bands1, [foff], band, off1)
            assert wband[0] == (band[0] + starts[w]) % yN64
            rel = relrms(got[w], want[0].cpu().numpy())
            assert rel < 5e-7, (foff, w, rel)
    # the 22- and 24-segment instances (full-size facets of the timed workload; a facet whose position in the padded row is not a
    # multiple of a segment touches 23 of them)
    big = (rng.standard_normal((3, 22528)) + 1j * rng.standard_normal((3, 22528))).astype(numpy.complex64)
    bdev = torch.from_numpy(big).cuda()
    wave_off1s = [0, 928, -928 * 3, 928 * 6]
    band = core.band_for_offsets(wave_off1s)
    sd = torch.tensor(core.window_starts(band, wave_off1s), dtype=torch.int32, device="cuda")
    _, ref = core64()
    k = numpy.arange(m)
    for foff in (22528, 22528 + 352, -22528 + 2):
        assert core.supports_window_rows(band, 22528, [foff], n_windows=4) and not core.supports_window_rows(band, 22528, [foff], 257)
        got = core.prepare_facet_window_rows(bdev, foff, band, sd, torch.empty((4, 3, m), dtype=torch.complex64, device="cuda"),
                                             fold_other_axis_window=False)
        bands1 = core.prepare_facet_band(bdev, foff, band, fold_other_axis_window=False)[None]
        prepared = ref.prepare_facet(big.astype(complex), foff, 1)                         # [3, yN], complex128
        sp = foff * xM64 // N64
        for w, off1 in enumerate(wave_off1s):
            rows_pass, wband = core.finish_axis1_rows(bands1, [foff], band, off1)
            # the oracle's window row (as in test_finish_axis1_rows_matches_oracle): both forms are float32 roundings of it
            s1 = off1 * yN64 // N64
            placed = ref.add_to_subgrid(ref.extract_from_facet(prepared, off1, 1), foff, 1)
            Z = placed[:, (k + xM64 // 2 - m // 2 + sp) % xM64]
            want_logical = Z[:, (k + s1) % m]
            wpc = band_cols(yN64, wband)
            cols = wpc[(wband[0] + k) % yN64]
            rel_fused = relrms(got[w].cpu().numpy()[:, cols], want_logical)
            rel_rows = relrms(rows_pass[0].cpu().numpy()[:, cols], want_logical)
            assert rel_fused < 2.5e-6 and rel_rows < 2.5e-6, (foff, w, rel_fused, rel_rows)   # measured 1.44e-6 / 1.45e-6
            assert rel_fused < 1.25 * rel_rows + 1e-8, (foff, w, rel_fused, rel_rows)
    # unsupported shapes are refused, not approximated: a band wider than the LDS stage
    wide = (0, core.WINDOW_ROWS_STAGE_COLUMNS + 64)
    assert not core.supports_window_rows(wide, size, [0])
    with pytest.raises(Exception):
        core.prepare_facet_window_rows(dev, 0, wide, torch.zeros(1, dtype=torch.int32, device="cuda"),
                                       torch.empty((rows, m), dtype=torch.complex64, device="cuda"))


def test_axis1_first_pipeline_matches_oracle_and_default_order():
    """(r6) SwiftlyConfig(axis1_first=True): the forward band pipeline with the contiguous axis finished before K2 / K3
    gives the oracle's subgrids (tighter than the default order: its float32 rounding acts on singly windowed data) in
    any request order, with and without the planned-wave prefetch -- in both of its forms: the finish in the epilogue of the
    whole-row K1 (what a planned pass on this configuration runs) and a row pass per wave (axis1_first="rows")."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    torch_, sw_, cfg0, facet_cfgs, facets, sg_cfgs = _small_rows_problem(seed=91)
    if not cfg0.core.supports_band_pipeline(torch.complex64):
        pytest.skip("band pipeline not available")
    P = dict(W=W64, fov=1.0, N=N64, yB_size=352, yN_size=yN64, xA_size=928, xM_size=xM64)
    cfg = sw.SwiftlyConfig(backend="hip", axis1_first=True, **P)
    assert cfg.core.axis1_first and not cfg0.core.axis1_first
    so = sep.SeparableOracle(core64()[1], [orc.CoverItem(c.off0, c.off1, c.size) for c in facet_cfgs],
                             [sep.facet_vectors(91 + j, 352, rank=2) for j in range(3)])
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off1, []).append(c)
    ref = sw.SwiftlyForward(cfg0, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
    for prefetch, fused in ((True, True), (False, True), (True, "rows"), (False, "rows")):
        old = sw.api._PREFETCH
        sw.api._PREFETCH = prefetch
        try:
            cfg.core.axis1_first = fused
            fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
            for key in (sorted(waves) if prefetch else sorted(waves)[::-1]):
                got = fwd.get_wave(waves[key]).cpu().numpy()
                base = ref.get_wave(waves[key]).cpu().numpy()
                for k, c in enumerate(waves[key]):
                    want = so.subgrid(orc.CoverItem(c.off0, c.off1, c.size))
                    assert relrms(got[k], want) < 4e-6, (prefetch, fused, key, k, relrms(got[k], want))
                    assert relrms(got[k], base[k]) < 3e-5
            assert fwd._axis1() == {True: 2, "rows": 1}[fused]  # pylint: disable=protected-access
        finally:
            sw.api._PREFETCH = old
    # without a plan there are no windows to fuse: the row pass per wave
    cfg.core.axis1_first = True
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), wave_axis=1)
    key = sorted(waves)[1]
    got = fwd.get_wave(waves[key]).cpu().numpy()
    assert fwd._axis1() == 1  # pylint: disable=protected-access
    for k, c in enumerate(waves[key]):
        assert relrms(got[k], so.subgrid(orc.CoverItem(c.off0, c.off1, c.size))) < 4e-6
    # pickling carries the switch (core.py:512-525: only parameters travel)
    import pickle

    assert pickle.loads(pickle.dumps(cfg.core)).axis1_first is True
    assert pickle.loads(pickle.dumps(sw.SwiftlyConfig(backend="hip", axis1_first="rows", **P).core)).axis1_first == "rows"


def test_axis1_first_through_the_multi_gpu_classes():
    """(r6) the axis-1-first order on the facet-sharded path: two virtual ranks (whole facets, in-process shuffle in place of
    the all-to-all) pack their blocks with finish_axis1_rows in front of K2 and the owners finish them in placed mode --
    the same subgrids as the single-process object in that mode."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw
    from ska_sdp_exec_swiftly_amd.distributed import DistributedForward

    _, _, cfg0, facet_cfgs, facets, sg_cfgs = _small_rows_problem(seed=93)
    if not cfg0.core.supports_band_pipeline(torch.complex64):
        pytest.skip("band pipeline not available")
    P = dict(W=W64, fov=1.0, N=N64, yB_size=352, yN_size=yN64, xA_size=928, xM_size=xM64)
    cfg = sw.SwiftlyConfig(backend="hip", axis1_first=True, **P)
    world = 2
    ref = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
    fwds = [DistributedForward(cfg, facet_cfgs, facets, dtype=torch.complex64, wave_axis=1, subgrid_configs=sg_cfgs,
                               rank_world=(r, world), cooperative=False) for r in range(world)]
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off1, []).append(c)
    for wave in waves.values():
        want = ref.get_wave(wave)
        packed = [f.pack_wave(wave) for f in fwds]
        # in-process all-to-all: rank r receives chunk r of every sender, in sender order
        recvs = []
        for r in range(world):
            parts = []
            for snd, (send, in_counts, _) in enumerate(packed):
                pos = sum(in_counts[:r])
                parts.append(send[pos : pos + in_counts[r]])
            recvs.append(torch.cat(parts))
        got = {}
        for r, f in enumerate(fwds):
            assert recvs[r].numel() == sum(packed[r][2])
            mine, res = f.unpack_wave(wave, recvs[r])
            for k, i in enumerate(mine):
                got[i] = res[k]
        assert sorted(got) == list(range(len(wave)))
        scale = float(want.abs().max())
        for i in range(len(wave)):
            assert float((got[i] - want[i]).abs().max()) <= 5e-6 * scale


def test_chained_k2_with_a_trailing_unchunked_facet_group():
    """(r5 advisor) More than 32 facets: K2 of a wave runs as one four-step per group of 32 facets, and only groups whose
    intermediate exceeds 256 MiB take the chunk-stream path.  36 facets at yN = 8192, m = 1024: the first group (2 GiB) is
    chunked, the trailing group of 4 facets (exactly 256 MiB) runs un-chunked on the side stream through slot 0 of the same
    workspace.  The chained K2 of the next prefetched wave (no fork behind the previous call) must still wait for it: the
    planned, prefetching pass gives the same BITS as the pass without prefetch."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    P = dict(W=11.0, fov=1.0, N=32768, yB_size=4096, yN_size=8192, xA_size=2048, xM_size=4096)
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    facet_cfgs = sw.api.make_full_cover_config(P["N"], P["yB_size"], sw.FacetConfig)[:36]
    cover = sw.api.make_full_cover_config(P["N"], P["xA_size"], sw.SubgridConfig)
    plan = [c for c in cover if c.off1 in (2048, 3 * 2048, 5 * 2048, 7 * 2048) and c.off0 in (0, 4 * 2048)]
    assert len(plan) == 8
    gen = torch.Generator(device="cuda").manual_seed(11)
    facets = [torch.view_as_complex(torch.randn((4096, 4096, 2), device="cuda", generator=gen)) for _ in facet_cfgs]
    waves = {}
    for c in plan:
        waves.setdefault(c.off1, []).append(c)

    def run(prefetch):
        old = sw.api._PREFETCH
        sw.api._PREFETCH = prefetch
        try:
            fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=plan, wave_axis=1)
            out = [fwd.get_wave(waves[k]).cpu().numpy() for k in sorted(waves)]
            used = fwd.__dict__.get("_prefetched", "never") != "never"
        finally:
            sw.api._PREFETCH = old
        torch.cuda.synchronize()
        return out, used

    plain, used0 = run(False)
    for _ in range(3):  # a race does not lose every time
        ahead, used1 = run(True)
        assert not used0 and used1
        for a, b in zip(plain, ahead):
            assert numpy.array_equal(a, b)


@pytest.mark.parametrize("lru_backward", [1, 2])
def test_backward_default_axis_with_plan_stages_single_adds(lru_backward):
    """SwiftlyBackward(subgrid_configs=plan) without wave_axis picks the band schedule for complex64 subgrids;
    subgrids added one by one are staged per off1 (LRUCache(lru_backward)) and each wave is folded in ONCE -- when
    complete, or on eviction / finish() for waves the caller leaves incomplete."""
    torch, sw, cfg, facet_cfgs, _, sg_cfgs = _small_rows_problem()
    vec = [sep.subgrid_vectors(900 + i, c.size, rank=2) for i, c in enumerate(sg_cfgs)]
    data = [bench.separable_facet(torch, vec[i], c) for i, c in enumerate(sg_cfgs)]
    by1 = sorted(range(len(sg_cfgs)), key=lambda i: (sg_cfgs[i].off1, i))
    ref = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs)
    ref.add_new_subgrid_tasks([sg_cfgs[i] for i in by1], [data[i] for i in by1])
    want = [t.cpu().numpy() for t in ref.finish()]
    for order_name, order in (("off0-major", list(range(len(sg_cfgs)))), ("off1-major", by1)):
        bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=lru_backward, subgrid_configs=sg_cfgs)
        folds = []
        inner = bwd._add_wave
        bwd._add_wave = lambda sgs, subs: (folds.append(len(sgs)), inner(sgs, subs))[1]
        for i in order:
            bwd.add_new_subgrid_task(sg_cfgs[i], data[i])
        assert bwd.wave_axis == 1
        got = [t.cpu().numpy() for t in bwd.finish()]
        if order_name == "off1-major" or lru_backward >= 4:
            assert folds == [3, 3, 3, 3], (order_name, folds)  # each planned wave folded exactly once, when complete
        assert sum(folds) == len(sg_cfgs)
        for a, b in zip(got, want):
            assert relrms(a, b) < 3e-6, order_name  # same kernels; summation order of the band columns may differ
    # no plan: single adds are staged until eviction / finish
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=lru_backward, wave_axis=1)
    for i in by1:
        bwd.add_new_subgrid_task(sg_cfgs[i], data[i])
    got = [t.cpu().numpy() for t in bwd.finish()]
    for a, b in zip(got, want):
        assert relrms(a, b) < 3e-6
    # complex128 subgrids keep the reference schedule
    b128 = sw.SwiftlyBackward(cfg, facet_cfgs, subgrid_configs=sg_cfgs)
    b128.add_new_subgrid_task(sg_cfgs[0], data[0].to(torch.complex128))
    assert b128.wave_axis == 0


def test_forward_plan_is_matched_by_value_and_partial_hits_release_their_bytes():
    """r3 advice: (i) requests with EQUAL but distinct config objects (a rebuilt cover) hit the planned-wave path;
    (ii) a group that is only partly cached is recomputed and the cached copies of its members are released, so the
    byte budget does not shrink; (iii) the wave_axis=0-only entry points say so when the object runs wave_axis=1."""
    torch, sw, cfg, facet_cfgs, facets, sg_cfgs = _small_rows_problem()
    if not cfg.core.supports_band_pipeline(torch.complex64):
        pytest.skip("band pipeline not available")
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs)
    clones = [sw.SubgridConfig(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in sg_cfgs]
    assert all(a is not b for a, b in zip(clones, sg_cfgs))
    calls = []
    inner = fwd.get_wave
    fwd.get_wave = lambda sgs, timer=None: (calls.append(len(sgs)), inner(sgs, timer))[1]
    first = fwd.get_subgrid_task(clones[0])
    assert calls == [3] and len(fwd._results) == 2  # the whole planned wave, two subgrids kept for later
    held = fwd._result_bytes
    assert held == 2 * clones[0].size ** 2 * 8
    # a group of the same wave with one cached and one already handed-out member: recomputed, nothing left behind
    wave = [c for c in clones if c.off1 == clones[0].off1]
    got = fwd.get_subgrid_tasks([wave[0], wave[1]])
    assert len(calls) == 2
    assert fwd._result_bytes == sum(t.numel() * t.element_size() for t in fwd._results.values())
    assert fwd._result_bytes <= held
    assert relrms(got[0].cpu().numpy(), first.cpu().numpy()) < 1e-6
    with pytest.raises(ValueError, match="wave_axis=0"):
        fwd.wave_contributions([sg_cfgs[0]])
    with pytest.raises(ValueError, match="wave_axis=0"):
        fwd.get_NMBF_BFs_off0(sg_cfgs[0].off0)


def test_backward_axis_is_fixed_by_the_first_entry_point():
    """r3 advice: SwiftlyBackward(subgrid_configs=plan) resolves its automatic schedule at whichever public entry
    point sees data first, and a later add cannot flip it."""
    torch, sw, cfg, facet_cfgs, _, sg_cfgs = _small_rows_problem()
    if not cfg.core.supports_backward_band(torch.complex64):
        pytest.skip("band schedule not available")
    xA = sg_cfgs[0].size
    data = [torch.randn((xA, xA), dtype=torch.complex64, device="cuda") for _ in sg_cfgs]
    by1 = sorted(range(len(sg_cfgs)), key=lambda i: (sg_cfgs[i].off1, i))
    wave = [i for i in by1 if sg_cfgs[i].off1 == sg_cfgs[by1[0]].off1]
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, subgrid_configs=sg_cfgs)
    parts = bwd.wave_contributions([sg_cfgs[i] for i in wave], [data[i] for i in wave])
    assert bwd.wave_axis == 1  # resolved here, not at the first add_new_subgrid_task
    bwd.accumulate_wave([sg_cfgs[i] for i in wave], parts)
    rest = [i for i in by1 if i not in wave]
    bwd.add_new_subgrid_tasks([sg_cfgs[i] for i in rest], [data[i] for i in rest])
    assert bwd.wave_axis == 1
    got = [t.cpu().numpy() for t in bwd.finish()]
    ref = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs)
    ref.add_new_subgrid_tasks([sg_cfgs[i] for i in by1], [data[i] for i in by1])
    want = [t.cpu().numpy() for t in ref.finish()]
    for a, b in zip(got, want):
        assert relrms(a, b) < 3e-6
    # complex128 data through the low-level entry keeps the reference schedule
    b128 = sw.SwiftlyBackward(cfg, facet_cfgs, subgrid_configs=sg_cfgs)
    b128.wave_contributions([sg_cfgs[0]], [data[0].to(torch.complex128)])
    assert b128.wave_axis == 0


def _virtual_all_to_all(sends, in_counts):
    """In-process stand-in for all_to_all_single: rank r receives, from every source s in order, the chunk s sent
    to r (``sends[s]`` flat buffers, ``in_counts[s][r]`` elements)."""
    import torch

    world = len(sends)
    starts = [numpy.concatenate([[0], numpy.cumsum(c)]) for c in in_counts]
    return [
        torch.cat([sends[s][int(starts[s][r]) : int(starts[s][r + 1])] for s in range(world)]) for r in range(world)
    ]


@pytest.mark.parametrize("world,whole_waves", [(2, False), (8, False), (2, True), (8, True)])
def test_cooperative_facets_virtual_ranks(world, whole_waves):
    """Facet counts that do not divide by the world size (r4): 3 facets on 2 ranks (one whole facet each + facet 2
    worked on by both) and on 8 ranks (all three cooperative).  Virtual ranks in one process, the three exchanges
    (band rows, forward blocks per wave, backward blocks per wave, finishing rows) replaced by in-process shuffles of
    the flat buffers; results against the single-process classes: the same kernels on the same numbers -- only the
    order of the facet sums (arrival order) and of the column overlaps differs.  ``whole_waves``: every wave's subgrids
    are finished / held by ONE rank (the exchange of a wave is a gather to its owner) instead of dealt out round-robin."""
    import torch

    from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward, DistributedForward

    torch_, sw, cfg, facet_cfgs, facets, sg_cfgs = _small_rows_problem()
    if not cfg.core.supports_band_pipeline(torch.complex64) or not cfg.core.supports_backward_band(torch.complex64):
        pytest.skip("band pipelines not available")
    ref = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
    fwds = [DistributedForward(cfg, facet_cfgs, facets, subgrid_configs=sg_cfgs, wave_axis=1, dtype=torch.complex64,
                               rank_world=(r, world), whole_waves=whole_waves) for r in range(world)]
    sh = fwds[0].sharding
    if whole_waves:  # 4 waves of 3 subgrids: every wave on one rank, the ranks' loads differ by at most one wave
        owners = [sh.wave_rank[int(c.off1)] for c in sg_cfgs]
        assert len(set(owners)) == min(world, 4) and all(len(f.subgrids_of([c])) in (0, 1) for f in fwds for c in sg_cfgs)
    assert sh.coop == ([2] if world == 2 else [0, 1, 2])
    assert sorted(k for r in range(world) for k in sh.keys_of[r]) == sorted({c.off1 for c in sg_cfgs})
    yB = facet_cfgs[0].size
    assert sum(sh.coop_rows(yB, r)[1] for r in range(world)) == yB
    for f in fwds:
        f.prepare_all_facets()
    for j in sh.coop:
        packed = [f.pack_coop(j) for f in fwds]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        for r, f in enumerate(fwds):
            assert recvs[r].numel() == sum(packed[r][2])
            f.unpack_coop(j, recvs[r])
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off1, []).append(c)
    full = {}
    if whole_waves:
        # one all-to-all per GROUP of waves with distinct owners: every rank sends to / receives from every rank
        assert sorted(k for g in sh.wave_groups for k in g) == sorted(waves) and all(len(g) <= world for g in sh.wave_groups)
        for group in sh.wave_groups:
            gw = [waves[k] for k in group]
            packed = [f.pack_group(gw) for f in fwds]
            recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
            seen = 0
            for r, f in enumerate(fwds):
                assert recvs[r].numel() == sum(packed[r][2])
                sgs, res = f.unpack_group(gw, recvs[r])
                if sgs is None:
                    continue
                seen += 1
                want = ref.get_wave(sgs)
                for i, c in enumerate(sgs):
                    assert float((res[i] - want[i]).abs().max()) <= 2e-5 * float(want.abs().max()), (c.off0, c.off1)
                    full[(c.off0, c.off1)] = want[i]
            assert seen == len(gw)
    for key, wave in ({} if whole_waves else waves).items():
        want = ref.get_wave(wave)
        packed = [f.pack_wave(wave) for f in fwds]
        # the cooperative facets' blocks come from the rank that owns the wave
        owner = sh.key_owner[int(key)]
        assert [len(sh.items_of(r, key)) for r in range(world)] == [
            len(sh.facets_of[r]) + (len(sh.coop) if r == owner else 0) for r in range(world)]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        got = {}
        for r, f in enumerate(fwds):
            assert recvs[r].numel() == sum(packed[r][2])
            mine, res = f.unpack_wave(wave, recvs[r])
            for k, i in enumerate(mine):
                got[i] = res[k]
        assert sorted(got) == list(range(len(wave)))
        scale = float(want.abs().max())
        for i, c in enumerate(wave):
            assert float((got[i] - want[i]).abs().max()) <= 2e-5 * scale, (key, i)
            full[(c.off0, c.off1)] = want[i]
    # backward, band schedule, the same wave ranges
    rb = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs)
    bwds = [DistributedBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs, dtype=torch.complex64,
                                rank_world=(r, world), whole_waves=whole_waves) for r in range(world)]
    assert bwds[0].sharding.coop == sh.coop
    if whole_waves:
        for group in bwds[0].sharding.wave_groups:
            gw = [waves[k] for k in group]
            for wave in gw:
                rb.add_new_subgrid_tasks(wave, [full[(c.off0, c.off1)] for c in wave])
            packed = []
            for b in bwds:
                held = [w for w in gw if b.sharding.wave_rank[b.wave_key(w)] == b.rank]
                packed.append(b.pack_group(gw, [full[(c.off0, c.off1)] for c in held[0]] if held else []))
            recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
            for r, b in enumerate(bwds):
                assert recvs[r].numel() == sum(packed[r][2])
                b.unpack_group(gw, recvs[r])
    for key, wave in ({} if whole_waves else waves).items():
        data = [full[(c.off0, c.off1)] for c in wave]
        rb.add_new_subgrid_tasks(wave, data)
        packed = [b.pack_wave(wave, [data[i] for i in b.subgrids_of(wave)]) for b in bwds]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        for r, b in enumerate(bwds):
            assert recvs[r].numel() == sum(packed[r][2])
            b.unpack_wave(wave, recvs[r])
    want = rb.finish()
    done = {}
    for b in bwds:
        idx, out = b.finish()
        for j, o in zip(idx, out):
            done[j] = o
    for j in sh.coop:
        packed = [b.pack_coop_finish(j) for b in bwds]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        rows = []
        for r, b in enumerate(bwds):
            assert recvs[r].numel() == sum(packed[r][2])
            jj, row0, piece = b.unpack_coop_finish(j, recvs[r])
            assert jj == j and row0 == sh.coop_rows(yB, r)[0]
            rows.append(piece)
        done[j] = torch.cat(rows)
    assert sorted(done) == list(range(len(facet_cfgs)))
    for j, w in enumerate(want):
        assert done[j].shape == w.shape
        assert float((done[j] - w).abs().pow(2).mean().sqrt()) <= 3e-6 * float(w.abs().pow(2).mean().sqrt()), j


_K1_PROBE = r"""
import hashlib, sys
import numpy, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip
core = SwiftlyCoreHip(10.875, 65536, 1024, 32768)
rng = numpy.random.default_rng(5)
x = torch.from_numpy((rng.standard_normal((5, 22528)) + 1j * rng.standard_normal((5, 22528))).astype(numpy.complex64)).cuda()
band = (10736, 11472)
h = hashlib.sha256()
for off in (0, 64 * 352, -64 * 352, -64 * 320, 64 * 351):
    h.update(core.prepare_facet_band(x, off, band, fold_other_axis_window=True).cpu().numpy().tobytes())
b = torch.from_numpy((rng.standard_normal((5, band[1])) + 1j * rng.standard_normal((5, band[1]))).astype(numpy.complex64)).cuda()
for off in (0, 64 * 352):
    h.update(core.finish_facet_band(b, band, off, 22528).cpu().numpy().tobytes())
# many rows, other bands and the 16-segment instance (r6)
xl = torch.from_numpy((rng.standard_normal((700, 22528)) + 1j * rng.standard_normal((700, 22528))).astype(numpy.complex64)).cuda()
for off, bnd in ((0, band), (64 * 352, band), (-64 * 320, (32001, 2049)), (64 * 352, (0, 32768))):
    h.update(core.prepare_facet_band(xl, off, bnd, fold_other_axis_window=False).cpu().numpy().tobytes())
xs = torch.from_numpy((rng.standard_normal((300, 16384)) + 1j * rng.standard_normal((300, 16384))).astype(numpy.complex64)).cuda()
h.update(core.prepare_facet_band(xs, 0, band).cpu().numpy().tobytes())   # 16 data segments
print("DIGEST", h.hexdigest())
"""


def test_k1_window_and_twiddle_tables_are_bit_identical(tmp_path):
    """(r5) the re-laid-out load window (16-byte loads) and the compact twiddle sections of the long-row kernels are
    copies of the plain tables: K1 and the backward finish give the same BITS with them switched off (SWIFTLY_K1_WIN4=0;
    the switch is read once per process, hence the subprocesses)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "k1_probe.py"
    script.write_text(_K1_PROBE)
    digests = []
    for mode in ("0", "1", "2"):
        env = dict(os.environ, SWIFTLY_K1_WIN4=mode)
        out = subprocess.run([sys.executable, str(script), root, os.path.join(root, "ska-sdp-distributed-fourier-transform_amd")],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1] == digests[2]
