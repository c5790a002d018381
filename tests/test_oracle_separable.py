"""The separable-facet oracle (oracle/separable.py) against the plain 2-D oracle
replica of the reference dataflow (oracle.forward_all, itself pinned by the
reference-generated goldens) on a small configuration, two parameter families (full covers need even sizes: the reference cover masks are empty for odd ones),
with point sources mixed in; and the point-source part against the direct DFT."""
import numpy
import pytest

from oracle import separable as sep
from oracle import swiftly_oracle as orc

PARAMS = [
    dict(W=11.0, N=512, yB=176, yN=256, xA=96, xM=128),
    dict(W=13.5625, N=1024, yB=416, yN=512, xA=228, xM=256),
]


@pytest.mark.parametrize("p", PARAMS)
def test_separable_matches_2d_oracle(p):
    core = orc.OracleCore(p["W"], p["N"], p["xM"], p["yN"])
    facet_items = orc.make_full_cover(p["N"], p["yB"])
    sg_items = orc.make_full_cover(p["N"], p["xA"])
    sources = [(1.0, 3, -7), (0.5, p["N"] // 2 - 5, 40), (-0.25, -p["yB"], p["yB"] + 3)]
    vectors, pixels, facets = [], [], []
    for j, item in enumerate(facet_items):
        a, b = sep.facet_vectors(100 + j, item.size, rank=2)
        pix = sep.point_source_pixels(sources, p["N"], item)
        dense = sum(numpy.outer(a[r], b[r]) for r in range(a.shape[0]))
        dense = dense * item.mask0[:, None] * item.mask1[None, :]
        for p0, p1, val in pix:
            dense[p0, p1] += val
        # exactly representable in float32 (what the device facets rely on)
        assert numpy.array_equal(dense.astype(numpy.complex64).astype(complex), dense)
        vectors.append((a, b))
        pixels.append(pix)
        facets.append(dense)
    # every source landed on exactly one facet
    assert sum(len(px) for px in pixels) == len(sources)
    so = sep.SeparableOracle(core, facet_items, vectors, pixels)
    picks = sep.pick_subgrids(sg_items, 6)
    assert len(set(picks)) == 6
    # rounding only; W = 13.56 amplifies by 1/pswf ~ 5e3 and then cancels (c.f. DESIGN.md section 2)
    tol = 1e-12 if p["W"] < 12 else 1e-9
    want = orc.forward_all(core, facet_items, facets, [sg_items[i] for i in picks])
    for i, w in zip(picks, want):
        got = so.subgrid(sg_items[i])
        assert numpy.abs(got - w).max() <= tol * numpy.abs(w).max()
    # contributions too
    sg = sg_items[picks[1]]
    bf = core.prepare_facet(facets[4], facet_items[4].off0, axis=0)
    col = orc.extract_column(core, bf, sg.off0, facet_items[4].off1)
    c_want = core.extract_from_facet(col, sg.off1, axis=1)
    c_got = so.contribution(4, sg)
    assert numpy.abs(c_got - c_want).max() <= tol * numpy.abs(c_want).max()


def test_point_sources_match_dft():
    p = PARAMS[0]
    core = orc.OracleCore(p["W"], p["N"], p["xM"], p["yN"])
    facet_items = orc.make_full_cover(p["N"], p["yB"])
    sg_items = orc.make_full_cover(p["N"], p["xA"])
    sources = [(1, i + 1, i) for i in range(10)] + [(2.0, -200, 133), (0.5, 255, -256)]
    pixels = [sep.point_source_pixels(sources, p["N"], it) for it in facet_items]
    so = sep.SeparableOracle(core, facet_items, None, pixels)
    for i in sep.pick_subgrids(sg_items, 5):
        sg = sg_items[i]
        truth = orc.make_subgrid_from_sources(sources, p["N"], sg.size, [sg.off0, sg.off1], [sg.mask0, sg.mask1])
        got = so.subgrid(sg)
        # W = 11: the algorithm itself is accurate to ~1e-7 of the peak here (c.f. reference decimal=8 at W=13.56)
        assert numpy.abs(got - truth).max() < 3e-7 * numpy.abs(truth).max()


@pytest.mark.parametrize("p", PARAMS)
def test_separable_backward_matches_2d_oracle(p):
    """SeparableBackwardOracle against the 2-D replica of SwiftlyBackward (orc.backward_all, pinned by the
    reference-generated round-trip goldens), sparse subgrid subset, rank-2 masked subgrids."""
    core = orc.OracleCore(p["W"], p["N"], p["xM"], p["yN"])
    facet_items = orc.make_full_cover(p["N"], p["yB"])
    sg_items = orc.make_full_cover(p["N"], p["xA"])[::2]
    vectors = [sep.subgrid_vectors(700 + i, sg.size, rank=2) for i, sg in enumerate(sg_items)]
    subgrids = []
    for sg, (u, v) in zip(sg_items, vectors):
        dense = sum(numpy.outer(u[r], v[r]) for r in range(u.shape[0])) * sg.mask0[:, None] * sg.mask1[None, :]
        assert numpy.array_equal(dense.astype(numpy.complex64).astype(complex), dense)
        subgrids.append(dense)
    want = orc.backward_all(core, facet_items, sg_items, subgrids)
    so = sep.SeparableBackwardOracle(core, facet_items, sg_items, vectors)
    tol = 1e-12 if p["W"] < 12 else 2e-9
    for j, w in enumerate(want):
        got = so.facet(j)
        assert numpy.abs(got - w).max() <= tol * numpy.abs(w).max(), j
        rows = [0, 5, w.shape[0] - 1]
        assert numpy.array_equal(so.facet_rows(j, rows), got[rows])
    # contribution of one subgrid to one facet against prepare_and_split_subgrid
    parts = orc.prepare_and_split_subgrid(core, subgrids[3], [sg_items[3].off0, sg_items[3].off1], facet_items)
    c = sep.backward_contribution(core, *vectors[3], sg_items[3], facet_items[5])
    assert numpy.abs(c - parts[5]).max() <= tol * numpy.abs(parts[5]).max()
