"""
The oracle (oracle/swiftly_oracle.py) checked against golden vectors produced
by the reference implementation itself (tests/golden/make_golden.py).
Tolerance: the oracle and the reference both compute in complex128 with the
same numpy.fft; they differ only in how index shuffles are expressed, so
agreement is at rounding level (1e-12 relative to the array scale).
"""
import os

import numpy
import pytest

from oracle import swiftly_oracle as orc

TEST_PARAMS = dict(W=13.5625, N=1024, yB_size=416, yN_size=512, xA_size=228, xM_size=256)
SMALL_PARAMS = dict(W=13.5625, N=512, yB_size=208, yN_size=256, xA_size=100, xM_size=128)
BENCH8K_PARAMS = dict(W=11.0, N=8192, yB_size=1408, yN_size=2048, xA_size=1024, xM_size=2048)


def core_of(p):
    return orc.OracleCore(p["W"], p["N"], p["xM_size"], p["yN_size"])


def close(a, b, tol=1e-12):
    scale = max(1.0, float(numpy.max(numpy.abs(b))))
    assert a.shape == b.shape
    assert numpy.max(numpy.abs(a - b)) <= tol * scale


@pytest.mark.parametrize("name,p", [("test", TEST_PARAMS), ("small", SMALL_PARAMS), ("bench8k", BENCH8K_PARAMS)])
def test_constants(golden_dir, name, p):
    g = numpy.load(os.path.join(golden_dir, "constants.npz"))
    core = core_of(p)
    # bit-exact: same scipy call, same chunking
    assert numpy.array_equal(core.pswf, g[f"{name}_pswf"])
    assert numpy.array_equal(core.Fb, g[f"{name}_Fb"])
    assert numpy.array_equal(core.Fn, g[f"{name}_Fn"])
    assert len(core.Fn) == core.xM_yN_size


def test_primitives_1d(golden_dir):
    g = numpy.load(os.path.join(golden_dir, "prim1d.npz"))
    p = TEST_PARAMS
    core = core_of(p)
    fos, sos = g["facet_offs"], g["sg_offs"]
    for yB in (p["yB_size"], p["yB_size"] - 1):
        for i, fo in enumerate(fos):
            close(core.prepare_facet(g[f"facet_{yB}"], int(fo), 0), g[f"prepare_facet_{yB}_{i}"])
            close(core.finish_facet(g[f"facc_{yB}"], int(fo), yB, 0), g[f"finish_facet_{yB}_{i}"])
    for i, so in enumerate(sos):
        # pure gathers / scatters: bit exact
        assert numpy.array_equal(core.extract_from_facet(g["prep"], int(so), 0), g[f"extract_from_facet_{i}"])
        assert numpy.array_equal(core.add_to_facet(g["contrib"], int(so), 0), g[f"add_to_facet_{i}"])
    for i, fo in enumerate(fos):
        close(core.add_to_subgrid(g["contrib"], int(fo), 0), g[f"add_to_subgrid_{i}"])
        close(core.extract_from_subgrid(g["sacc"], int(fo), 0), g[f"extract_from_subgrid_{i}"])
    for xA in (p["xA_size"], p["xA_size"] - 1):
        for i, so in enumerate(sos):
            close(core.finish_subgrid(g["sacc"], int(so), xA), g[f"finish_subgrid_{xA}_{i}"])
            close(core.prepare_subgrid(g[f"subgrid_{xA}"], int(so)), g[f"prepare_subgrid_{xA}_{i}"])


def test_primitives_2d(golden_dir):
    g = numpy.load(os.path.join(golden_dir, "prim2d.npz"))
    p = SMALL_PARAMS
    core = core_of(p)
    fo0, fo1, so0, so1 = (int(v) for v in g["offs"])
    xA, yB = p["xA_size"], p["yB_size"]
    facet = g["facet"]
    BF = core.prepare_facet(facet, fo0, 0)
    close(BF, g["prepare_facet_a0"])
    close(core.prepare_facet(facet[:37], fo1, 1), g["prepare_facet_a1"])
    close(core.extract_from_facet(BF, so0, 0), g["extract_from_facet_a0"])
    col = orc.extract_column(core, BF, so0, fo1)
    close(col, g["extract_column"])
    contrib = core.extract_from_facet(col, so1, 1)
    close(contrib, g["contrib"])
    a0 = core.add_to_subgrid(g["contrib"], fo0, 0)
    close(a0, g["add_to_subgrid_a0"])
    a01 = core.add_to_subgrid(g["add_to_subgrid_a0"], fo1, 1)
    close(a01, g["add_to_subgrid_a01"])
    close(core.finish_subgrid(g["add_to_subgrid_a01"], [so0, so1], xA), g["finish_subgrid"])
    close(core.finish_subgrid(g["add_to_subgrid_a01"], [so0, so1], xA - 1), g["finish_subgrid_odd"])
    ps = core.prepare_subgrid(g["subgrid"], [so0, so1])
    close(ps, g["prepare_subgrid"])
    e0 = core.extract_from_subgrid(g["prepare_subgrid"], fo0, 0)
    close(e0, g["extract_from_subgrid_a0"])
    e01 = core.extract_from_subgrid(g["extract_from_subgrid_a0"], fo1, 1)
    close(e01, g["extract_from_subgrid_a01"])
    f1 = core.add_to_facet(g["extract_from_subgrid_a01"], so1, 1)
    assert numpy.array_equal(f1, g["add_to_facet_a1"])
    ff1 = core.finish_facet(g["add_to_facet_a1"], fo1, 61, 1)
    close(ff1, g["finish_facet_a1"])
    f0 = core.add_to_facet(g["finish_facet_a1"], so0, 0)
    assert numpy.array_equal(f0, g["add_to_facet_a0"])
    close(core.finish_facet(g["add_to_facet_a0"], fo0, yB, 0), g["finish_facet_a0"])


def small_problem(g):
    """Rebuild the cover + seeded facets of roundtrip2d.npz."""
    p = SMALL_PARAMS
    facet_items = [
        orc.CoverItem(o0, o1, p["yB_size"], m0, m1)
        for (o0, o1), m0, m1 in zip(g["facet_offs"], g["facet_mask0"], g["facet_mask1"])
    ]
    sg_items = [
        orc.CoverItem(o0, o1, p["xA_size"], m0, m1)
        for (o0, o1), m0, m1 in zip(g["sg_offs"], g["sg_mask0"], g["sg_mask1"])
    ]
    yB = p["yB_size"]
    facets = []
    for j, f in enumerate(facet_items):
        r = numpy.random.default_rng(1234 + j)
        d = (r.standard_normal((yB, yB)) + 1j * r.standard_normal((yB, yB))).astype(numpy.complex64).astype(complex)
        facets.append(d * f.mask0[:, None] * f.mask1[None, :])
    return facet_items, sg_items, facets


def test_cover_matches_reference(golden_dir):
    g = numpy.load(os.path.join(golden_dir, "roundtrip2d.npz"))
    p = SMALL_PARAMS
    for chunk, offs, m0, m1 in (
        (p["yB_size"], g["facet_offs"], g["facet_mask0"], g["facet_mask1"]),
        (p["xA_size"], g["sg_offs"], g["sg_mask0"], g["sg_mask1"]),
    ):
        cover = orc.make_full_cover(p["N"], chunk)
        assert numpy.array_equal(numpy.array([[c.off0, c.off1] for c in cover]), offs)
        assert numpy.array_equal(numpy.array([c.mask0 for c in cover]), m0)
        assert numpy.array_equal(numpy.array([c.mask1 for c in cover]), m1)


def test_roundtrip_2d(golden_dir):
    g = numpy.load(os.path.join(golden_dir, "roundtrip2d.npz"))
    core = core_of(SMALL_PARAMS)
    facet_items, sg_items, facets = small_problem(g)
    sgs = numpy.array(orc.forward_all(core, facet_items, facets, sg_items))
    close(sgs[:, ::7, ::5], g["subgrids_sample"])
    close(sgs[g["subgrid_full_idx"]], g["subgrids_full"])
    assert numpy.allclose(sgs.sum(axis=(1, 2)), g["subgrids_sum"], rtol=0, atol=1e-10)
    assert numpy.allclose((numpy.abs(sgs) ** 2).sum(axis=(1, 2)), g["subgrids_pow"], rtol=1e-12)
    fo = numpy.array(orc.backward_all(core, facet_items, sg_items, list(sgs)))
    close(fo[:, ::9, ::7], g["facets_out_sample"], tol=1e-11)
    close(fo[g["facet_full_idx"]], g["facets_out_full"], tol=1e-11)
    rmse = max(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2)) for a, b in zip(fo, facets))
    assert abs(rmse - float(g["roundtrip_rmse"])) < 1e-9
