"""
Element-wise GPU parity of the subgrid -> facet direction at the BENCHMARKED shape and of its two large
kernels at the long transform lengths.

* ``test_backward_64k_sparse_rows_match_oracle``: workload "64k-sparse" of bench.py (N = 65536, 3x3 facets of
  22528^2, 505 subgrids, complex64), subgrids = dense rank-2 outer products on the 1/8 grid times the subgrid
  cover masks (exact in float32), both backward schedules.  ``oracle/separable.py:SeparableBackwardOracle`` gives
  any facet row from 1-D oracle primitives (reference core.py:328-484, composition api_helper.py:115-197; pinned
  against the 2-D replica of SwiftlyBackward in tests/test_oracle_separable.py): 16 sampled rows of EVERY facet are
  compared element by element.
* primitive tests of ``swiftly_hip_accumulate_facet_columns`` (gather-sum column pass: add_to_facet axis 0 ->
  finish_facet axis 0 -> mask0 -> add_to_facet axis 1 into the band) and ``swiftly_hip_finish_facet_band`` (the
  ``ST = 2`` long-row kernel) at yN = 32768 and 65536 against the oracle primitives.

Tolerances (complex64, W = 10.875 family): single long transform of un-amplified data 2e-6
relative RMSE; the whole backward pass 3.5e-5 per facet row set with float32 arithmetic (measured 2.1e-5), 1e-5 with
float64 arithmetic in the column passes (measured 5.1e-6).
"""
import numpy
import pytest

import bench
from oracle import separable as sep
from oracle import swiftly_oracle as orc

pytestmark = pytest.mark.gpu

BACKWARD_TOL = bench.BACKWARD_PARITY_TOL


def relrms(a, b):
    return float(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2)))


@pytest.mark.parametrize("baxis,bits", [(1, 32), (0, 32), (1, 64)], ids=["band-f32", "reference-order-f32", "band-f64"])
def test_backward_64k_sparse_rows_match_oracle(baxis, bits):
    """bits = 64: float64 arithmetic in the column passes of the band schedule (gather-sum four-step + the m-point
    pass of extract_from_subgrid): 2.1e-5 -> 5.1e-6, bound 1e-5."""
    import torch

    import ska_sdp_exec_swiftly_amd as sw

    wl = bench.WORKLOADS["64k-sparse"]
    p = wl["params"]
    cfg = sw.SwiftlyConfig(backend="hip", column_precision=bits, **p)
    tol = BACKWARD_TOL if bits == 32 else bench.HIGH_PRECISION_BACKWARD_PARITY_TOL
    facet_cfgs = sw.make_full_facet_cover(cfg)
    sg_cfgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
    assert (len(facet_cfgs), len(sg_cfgs)) == (9, 505)
    if baxis == 1 and not cfg.core.supports_backward_band(torch.complex64):
        pytest.skip("band schedule not available")
    vectors = [sep.subgrid_vectors(4321 + i, c.size, rank=2) for i, c in enumerate(sg_cfgs)]
    bkey = (lambda c: c.off1) if baxis == 1 else (lambda c: c.off0)
    order = sorted(range(len(sg_cfgs)), key=lambda i: (bkey(sg_cfgs[i]), sg_cfgs[i].off0, sg_cfgs[i].off1))
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=baxis, subgrid_configs=sg_cfgs)
    # wave by wave, so that only one wave of subgrids (<= 25 x 6.9 MB) is alive at a time
    i = 0
    while i < len(order):
        j = i
        while j < len(order) and bkey(sg_cfgs[order[j]]) == bkey(sg_cfgs[order[i]]):
            j += 1
        wave = [sg_cfgs[k] for k in order[i:j]]
        data = [bench.separable_facet(torch, vectors[k], sg_cfgs[k]) for k in order[i:j]]
        bwd.add_new_subgrid_tasks(wave, data)
        i = j
    out = bwd.finish()
    torch.cuda.synchronize()
    par = bench.verify_facets(p, facet_cfgs, sg_cfgs, vectors, out, rows_per_facet=16)
    print(f"backward wave_axis={baxis} float{bits}: relRMSE per facet {par['rel_rmse_each']} max|err|/rms {par['max_abs_over_rms']:.2e}")
    assert par["facets"] == 9 and par["rows_per_facet"] == 16
    assert par["rel_rmse"] < tol, par
    assert par["max_abs_over_rms"] < 40 * tol, par


LONG = [
    dict(W=10.875, N=65536, xM=1024, yN=32768, yB=22528, xA=928),
    dict(W=10.875, N=131072, xM=1024, yN=65536, yB=45056, xA=928),
]


def _cores(p):
    from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip

    return SwiftlyCoreHip(p["W"], p["N"], p["xM"], p["yN"]), orc.OracleCore(p["W"], p["N"], p["xM"], p["yN"])


@pytest.mark.parametrize("p", LONG, ids=lambda p: f"yN{p['yN']}")
def test_finish_facet_band_long_rows(p):
    """finish_facet along the contiguous axis of a band accumulator (plain column order, zero outside the band)."""
    import torch

    core, ref = _cores(p)
    yN, yB = p["yN"], p["yB"]
    rng = numpy.random.default_rng(31)
    rows = 24
    mask = (rng.random(yB) > 0.15).astype(float)
    for band, off in (((yN // 2 - 5000, 11472), 0), ((yN - 3000, 7001), yB), ((0, yN), -yB)):
        start, length = band
        data = (rng.standard_normal((rows, length)) + 1j * rng.standard_normal((rows, length))).astype(numpy.complex64)
        full = numpy.zeros((rows, yN), dtype=complex)
        full[:, (start + numpy.arange(length)) % yN] = data
        want = ref.finish_facet(full, off, yB, axis=1) * mask[None, :]
        got = core.finish_facet_band(torch.from_numpy(data).cuda(), band, off, yB, mask=mask).cpu().numpy()
        assert got.shape == (rows, yB)
        rel = relrms(got, want)
        assert rel < 2e-6, (band, off, rel)
        # masked pixels are exactly zero
        assert not got[:, mask == 0].any()


@pytest.mark.parametrize("p", LONG, ids=lambda p: f"yN{p['yN']}")
def test_accumulate_facet_columns_long_columns(p):
    """Gather-sum column pass: two waves whose band columns overlap (first-write flags + read-modify-write), three
    subgrids per wave with overlapping row windows, two facets with masks; sampled columns against the oracle."""
    import torch

    core, ref = _cores(p)
    N, yN, yB, xA, m = p["N"], p["yN"], p["yB"], p["xA"], core.xM_yN_size
    rng = numpy.random.default_rng(32)
    F = 2
    facet_off0s = [0, yB]
    masks = (rng.random((F, yB)) > 0.1).astype(numpy.float32)
    waves = [  # (subgrid off1, [subgrid off0 ...]): adjacent off1 -> band windows overlap by m - xA*yN/N columns
        (3 * xA, [0, xA, -2 * xA]),
        (4 * xA, [xA, 2 * xA]),
    ]
    band = core.band_for_offsets([w[0] for w in waves] + [6 * xA])
    start, length = band
    bands = torch.full((F, yB, length), float("nan"), dtype=torch.complex64, device="cuda")  # uninitialised on purpose
    touched = torch.zeros((length,), dtype=torch.uint8, device="cuda")
    work = torch.empty((F, yN, m), dtype=torch.complex64, device="cuda")
    mask_t = torch.from_numpy(masks).cuda()
    def window(off1):
        s1 = off1 * yN // N
        return s1, (yN // 2 - m // 2 + numpy.arange(m) + s1) % yN  # logical column of window index i

    w_a, w_b = set(window(waves[0][0])[1].tolist()), set(window(waves[1][0])[1].tolist())
    both = sorted(w_a & w_b)
    assert len(both) == m - xA * yN // N  # adjacent waves share 48 band columns
    only_a, only_b = sorted(w_a - w_b), sorted(w_b - w_a)
    bigs = [only_a[0], only_a[len(only_a) // 2], only_b[-1], only_b[7], both[0], both[-1], both[len(both) // 2]]
    want = {big: numpy.zeros((F, yB), dtype=complex) for big in bigs}
    for off1, off0s in waves:
        S = len(off0s)
        parts = (rng.standard_normal((F, S, m, m)) + 1j * rng.standard_normal((F, S, m, m))).astype(numpy.complex64)
        pt = torch.from_numpy(parts).cuda()
        groups = core.column_row_sources(off0s)
        assert len(groups) == 1
        core.accumulate_facet_columns(pt, m, [0], [pt.stride(0)], groups[0][1], facet_off0s, yB, mask_t, off1, bands, band,
                                      workspace=work, touched=touched)
        s1, logical = window(off1)
        for big in bigs:  # add_to_facet along axis 1 (core.py:441-449): window index i holds contribution column (i + s1) mod m
            hit = numpy.flatnonzero(logical == big)
            if hit.size == 0:
                continue
            c = (int(hit[0]) + s1) % m
            for f in range(F):
                acc = numpy.zeros(yN, dtype=complex)
                for b, o0 in enumerate(off0s):
                    acc = ref.add_to_facet(parts[f, b][:, c].astype(complex), o0, axis=0, out=acc)
                want[big][f] += ref.finish_facet(acc, facet_off0s[f], yB, axis=0) * masks[f]
    core.band_zero_untouched(bands, touched)
    got = bands.cpu().numpy()
    assert numpy.isfinite(got.view(numpy.float32)).all()
    tch = touched.cpu().numpy()
    for big in bigs:
        d = (big - start) % yN
        assert d < length and tch[d] == 1
        for f in range(F):
            rel = relrms(got[f][:, d], want[big][f])
            assert rel < 2e-6, (big, f, rel)
    # columns no wave wrote are zero, and some exist
    assert (tch == 0).any() and not got[:, :, tch == 0].any()
