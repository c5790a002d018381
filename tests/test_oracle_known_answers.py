"""
The oracle against the reference's own known-answer tests
(/root/reference/tests/test_core.py:93-428, restated in tests/kat.py) and the
hand-written truth-generator goldens of tests/test_fourier_algorithm.py.
"""
import numpy
import pytest

import kat
from oracle import swiftly_oracle as orc

P = kat.TEST_PARAMS


def make_core(p):
    return orc.OracleCore(p["W"], p["N"], p["xM_size"], p["yN_size"])


def test_params():
    # tests/test_core.py:43-79
    c = make_core(P)
    assert (c.W, c.N, c.yN_size, c.xM_size, c.xM_yN_size) == (13.5625, 1024, 512, 256, 128)
    bad = dict(P, N=1050)
    with pytest.raises(ValueError):
        make_core(bad)
    assert c.subgrid_off_step == 2 and c.facet_off_step == 4


@pytest.mark.parametrize("xA", [P["xA_size"], P["xA_size"] - 1])
@pytest.mark.parametrize("yB", [P["yB_size"], P["yB_size"] - 1])
def test_facet_to_subgrid_basic(xA, yB):
    kat.facet_to_subgrid_basic(make_core, xA, yB)


@pytest.mark.parametrize("xA", [P["xA_size"], P["xA_size"] - 1])
@pytest.mark.parametrize("yB", [P["yB_size"], P["yB_size"] - 1])
def test_facet_to_subgrid_dft_1d(xA, yB):
    kat.facet_to_subgrid_dft_1d(make_core, xA, yB)


def test_facet_to_subgrid_dft_2d():
    kat.facet_to_subgrid_dft_2d(make_core)


@pytest.mark.parametrize("xA", [P["xA_size"], P["xA_size"] - 1])
@pytest.mark.parametrize("yB", [P["yB_size"], P["yB_size"] - 1])
def test_subgrid_to_facet_basic(xA, yB):
    kat.subgrid_to_facet_basic(make_core, xA, yB)


@pytest.mark.parametrize("xA", [P["xA_size"], P["xA_size"] - 1])
@pytest.mark.parametrize("yB", [P["yB_size"], P["yB_size"] - 1])
def test_subgrid_to_facet_dft(xA, yB):
    kat.subgrid_to_facet_dft(make_core, xA, yB)


def test_subgrid_to_facet_dft_2d():
    kat.subgrid_to_facet_dft_2d(make_core)


# ---- truth generators: hand-written cases in the style of
# reference tests/test_fourier_algorithm.py:587-676
def test_make_facet_from_sources():
    f = orc.make_facet_from_sources([(1, 0)], 100, 10, [0])
    assert f[5] == 1 and f.sum() == 1
    f = orc.make_facet_from_sources([(2, 3)], 100, 10, [0])
    assert f[8] == 2
    f = orc.make_facet_from_sources([(2, 5)], 100, 10, [0])  # outside (index 10)
    assert f.sum() == 0
    f = orc.make_facet_from_sources([(1, -5)], 100, 10, [0])
    assert f[0] == 1
    f = orc.make_facet_from_sources([(1, 52)], 100, 10, [50])  # offset
    assert f[7] == 1
    f = orc.make_facet_from_sources([(1, -48)], 100, 10, [50])  # wraps modulo N
    assert f[7] == 1
    f = orc.make_facet_from_sources([(1, 1, -2)], 100, 9, [0, 0])  # odd size: centre 4
    assert f[5, 2] == 1 and f.sum() == 1
    f = orc.make_facet_from_sources([(1, 1, -2)], 100, 9, [0, 0], [numpy.ones(9), numpy.zeros(9)])
    assert f.sum() == 0


def test_make_subgrid_from_sources():
    # one source at the centre -> constant 1/N
    sg = orc.make_subgrid_from_sources([(1, 0)], 100, 10, [0])
    assert numpy.allclose(sg, 1 / 100)
    # general: equals the inverse DFT of the image
    N = 64
    img = numpy.zeros(N, dtype=complex)
    srcs = [(1.5, 3), (-0.5, -7), (0.25, 20)]
    for i, x in srcs:
        img[N // 2 + x] += i
    full = orc.cifft(img, 0)
    sg = orc.make_subgrid_from_sources(srcs, N, 11, [5])
    assert numpy.allclose(sg, numpy.take(full, numpy.arange(5 - 5, 5 + 6) + N // 2, mode="wrap"))
    sg2 = orc.make_subgrid_from_sources([(1, 2, -3)], N, 8, [4, -6])
    u0 = numpy.arange(4 - 4, 4 + 4)
    u1 = numpy.arange(-6 - 4, -6 + 4)
    exp = numpy.exp(2j * numpy.pi / N * (u0[:, None] * 2 + u1[None, :] * -3)) / N**2
    assert numpy.allclose(sg2, exp)
