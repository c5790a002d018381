"""
Known-answer checks restated from the reference's tests/test_core.py, written
against the duck-typed core surface so that the same
checks run on the oracle (CPU) and on SwiftlyCoreHip (GPU, `-m gpu`).

Every function takes ``make_core(params) -> core`` and the absolute tolerances
to use; the reference's own tolerances are the defaults (complex128).
Truth is the direct DFT of point sources (``make_subgrid_from_sources`` /
``make_facet_from_sources`` restated in oracle/swiftly_oracle.py and pinned by
tests/test_oracle_known_answers.py and tests/test_oracle_golden.py).
"""
import itertools

import numpy

from oracle import swiftly_oracle as orc

TEST_PARAMS = dict(W=13.5625, N=1024, yB_size=416, yN_size=512, xA_size=228, xM_size=256)


def _np(a):
    """Accept numpy arrays or torch tensors."""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return numpy.asarray(a)


def facet_to_subgrid_basic(make_core, xA, yB, tol=1.5e-15, dtype=complex, thin=1):
    """reference tests/test_core.py:93-136: a unit pixel at the image centre
    gives a constant subgrid val/N."""
    p = TEST_PARAMS
    N = p["N"]
    core = make_core(p)
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    for val, facet_off in itertools.product([0, 1, 0.1], numpy.arange(-5 * Ny, 5 * Ny // 2, Ny * thin)):
        facet_off = int(facet_off)
        facet = numpy.zeros(yB, dtype=dtype)
        facet[yB // 2 - facet_off] = val
        prepped = core.prepare_facet(facet, facet_off, axis=0)
        for sg_off in numpy.arange(0, 10 * Nx, Nx * thin):
            sg_off = int(sg_off)
            c = core.extract_from_facet(prepped, sg_off, axis=0)
            acc = core.add_to_subgrid(c, facet_off, axis=0)
            sg = _np(core.finish_subgrid(acc, sg_off, xA))
            assert sg.shape == (xA,)
            assert numpy.max(numpy.abs(sg - val / N)) < tol


SOURCES_1D = [
    [(1, 0)],
    [(2, 1)],
    [(1, -3)],
    [(-0.1, 5)],
    [(1 / 8, 20), (2 / 8, 5), (3 / 8, -4)],
    "border-",
    "border+",
    [(1 / 16, i) for i in range(-10, 10)],
]


def facet_to_subgrid_dft_1d(make_core, xA, yB, tol=1.5e-8, dtype=complex, thin=1):
    """reference tests/test_core.py:139-199."""
    p = TEST_PARAMS
    N = p["N"]
    core = make_core(p)
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    for sources, facet_off in itertools.product(SOURCES_1D, numpy.arange(-100 * Ny, 100 * Ny, 10 * Ny * thin)):
        facet_off = int(facet_off)
        if sources == "border-":
            sources = [(1, -yB)]
        elif sources == "border+":
            sources = [(1, yB)]
        min_x = -(yB - 1) // 2 + facet_off
        max_x = min_x + yB - 1
        sources = [(i, min(max(x, min_x), max_x)) for i, x in sources]
        facet = orc.make_facet_from_sources(sources, N, yB, [facet_off]).astype(dtype)
        assert numpy.isclose(numpy.sum(facet), sum(s[0] for s in sources))
        prepped = core.prepare_facet(facet, facet_off, axis=0)
        for sg_off in [0, Nx, -Nx, N]:
            c = core.extract_from_facet(prepped, sg_off, axis=0)
            acc = core.add_to_subgrid(c, facet_off, axis=0)
            sg = _np(core.finish_subgrid(acc, sg_off, xA))
            expected = orc.make_subgrid_from_sources(sources, N, xA, [sg_off])
            assert numpy.max(numpy.abs(sg - expected)) < tol, (sources, facet_off, sg_off)


def facet_to_subgrid_dft_2d(make_core, tol=1.5e-8, dtype=complex):
    """reference tests/test_core.py:202-254."""
    p = TEST_PARAMS
    N, xA, yB = p["N"], p["xA_size"], p["yB_size"]
    core = make_core(p)
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    for sources, fo in itertools.product(
        [[(1, 1, 2)], [(1 / 8, 20, 4), (2 / 8, 2, 5), (3 / 8, -5, -4)]],
        [[0, 0], [Ny, Ny], [-Ny, Ny], [0, -Ny]],
    ):
        facet = orc.make_facet_from_sources(sources, N, yB, fo).astype(dtype)
        p0 = core.prepare_facet(facet, fo[0], axis=0)
        pp = core.prepare_facet(p0, fo[1], axis=1)
        for so in [[0, 0], [0, Nx], [Nx, 0], [-Nx, -Nx]]:
            c0 = core.extract_from_facet(pp, so[0], axis=0)
            c = core.extract_from_facet(c0, so[1], axis=1)
            a0 = core.add_to_subgrid(c, fo[0], axis=0)
            a = core.add_to_subgrid(a0, fo[1], axis=1)
            sg = _np(core.finish_subgrid(a, list(so), xA))
            expected = orc.make_subgrid_from_sources(sources, N, xA, so)
            assert numpy.max(numpy.abs(sg - expected)) < tol


def subgrid_to_facet_basic(make_core, xA, yB, tol=1.5e-13, dtype=complex, thin=1):
    """reference tests/test_core.py:257-293."""
    p = TEST_PARAMS
    core = make_core(p)
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    sg_offs = Nx * numpy.arange(-9, 8, thin)
    facet_offs = Ny * numpy.arange(-9, 8, thin)
    for val, sg_off in itertools.product([0, 1, 0.1], sg_offs):
        sg_off = int(sg_off)
        prepped = core.prepare_subgrid(((val / xA) * numpy.ones(xA)).astype(dtype), sg_off)
        for facet_off in facet_offs:
            facet_off = int(facet_off)
            e = core.extract_from_subgrid(prepped, facet_off, axis=0)
            acc = core.add_to_facet(e, sg_off, axis=0)
            facet = _np(core.finish_facet(acc, facet_off, yB, axis=0))
            assert abs(facet[yB // 2 - facet_off] - val) < tol


def subgrid_to_facet_dft(make_core, xA, yB, tol=1.5e-11, dtype=complex, thin=1):
    """reference tests/test_core.py:296-361."""
    p = TEST_PARAMS
    N = p["N"]
    core = make_core(p)
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    sg_offs = Nx * numpy.arange(-9, 8, thin)
    facet_offs = Ny * numpy.arange(-9, 8, thin)
    for sources, sg_off in itertools.product([[(1, 0)], [(2, 1)], [(1, -3)], [(-0.1, 5)]], sg_offs):
        sg_off = int(sg_off)
        subgrid = (orc.make_subgrid_from_sources(sources, N, xA, [sg_off]) / xA * N).astype(dtype)
        prepped = core.prepare_subgrid(subgrid, sg_off)
        for facet_off in facet_offs:
            facet_off = int(facet_off)
            e = core.extract_from_subgrid(prepped, facet_off, axis=0)
            acc = core.add_to_facet(e, sg_off, axis=0)
            facet = _np(core.finish_facet(acc, facet_off, yB, axis=0))
            expected = orc.make_facet_from_sources(sources, N, yB, [facet_off])
            nz = expected != 0
            assert numpy.max(numpy.abs(facet[nz] - expected[nz]), initial=0) < tol
            if sources[0][0] > 0:
                assert numpy.all(facet[~nz].real < numpy.max(expected.real))
            else:
                assert numpy.all(-facet[~nz].real < numpy.max(-expected.real))


def subgrid_to_facet_dft_2d(make_core, tol=1.5e-11, dtype=complex):
    """reference tests/test_core.py:364-428."""
    p = TEST_PARAMS
    N, xA, yB = p["N"], p["xA_size"], p["yB_size"]
    core = make_core(p)
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    for sources, so in itertools.product(
        [[(1, 0, 0)], [(1, 20, 4)], [(3, -5, 4)]], [[0, 0], [0, Nx], [Nx, 0], [-Nx, -Nx]]
    ):
        subgrid = (orc.make_subgrid_from_sources(sources, N, xA, so) / xA / xA * N * N).astype(dtype)
        prepped = core.prepare_subgrid(subgrid, list(so))
        for fo in [[0, 0], [Ny, Ny], [-Ny, Ny], [0, -Ny]]:
            e0 = core.extract_from_subgrid(prepped, fo[0], axis=0)
            e1 = core.extract_from_subgrid(e0, fo[1], axis=1)
            a0 = core.add_to_facet(e1, so[0], axis=0)
            a1 = core.add_to_facet(a0, so[1], axis=1)
            f0 = core.finish_facet(a1, fo[0], yB, axis=0)
            f1 = _np(core.finish_facet(f0, fo[1], yB, axis=1))
            expected = orc.make_facet_from_sources(sources, N, yB, fo)
            nz = expected != 0
            assert numpy.max(numpy.abs(f1[nz] - expected[nz])) < tol
            assert numpy.all(f1[~nz].real < numpy.max(expected.real))
