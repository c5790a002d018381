"""
Size-independent property used for full-size backward parity: the subgrid -> facet transform is N^2 times the
adjoint of the facet -> subgrid transform,

    sum conj(y) * forward(M_f x)  ==  N**-2 * sum conj(backward(M_s y)) * x

(M_f / M_s: the facet / subgrid cover masks; the forward pass applies M_s, the backward pass M_f).  Per axis:
forward = (1/(yN xM)) C D_xM^H Pl Fn D_m S D_yN^H P Fb, backward = (1/m) Fb P^T D_yN S^T D_m^H Fn Pl^T D_xM C^T
with m = xM yN / N (reference core.py:189-510).  Checked here on the ORACLE (pinned to the reference's fixtures) in
complex128; tests/test_hip_bench_shape_gpu.py uses the identity at N = 65536 to tie the HIP backward pass to the
oracle-checked forward pass.
"""
import numpy
import pytest

from oracle import swiftly_oracle as orc

PARAMS = [
    dict(W=13.5625, N=512, yB_size=208, yN_size=256, xA_size=100, xM_size=128),
    dict(W=11.0, N=512, yB_size=176, yN_size=256, xA_size=96, xM_size=128),
]


@pytest.mark.parametrize("p", PARAMS)
def test_backward_is_scaled_adjoint_of_forward(p):
    core = orc.OracleCore(p["W"], p["N"], p["xM_size"], p["yN_size"])
    facet_items = orc.make_full_cover(p["N"], p["yB_size"])
    sg_items = orc.make_full_cover(p["N"], p["xA_size"])[::3]  # a sparse subset is enough: the identity is per pair
    rng = numpy.random.default_rng(0)

    def rnd(n):
        return rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))

    x = [rnd(p["yB_size"]) for _ in facet_items]
    y = [rnd(p["xA_size"]) for _ in sg_items]
    xm = [a * f.mask0[:, None] * f.mask1[None, :] for a, f in zip(x, facet_items)]
    ym = [a * s.mask0[:, None] * s.mask1[None, :] for a, s in zip(y, sg_items)]
    fx = orc.forward_all(core, facet_items, xm, sg_items)
    by = orc.backward_all(core, facet_items, sg_items, ym)
    lhs = sum(numpy.vdot(b, a) for a, b in zip(fx, y))                 # sum conj(y) * F(x)
    rhs = sum(numpy.vdot(b, a) for a, b in zip(x, by)) / p["N"] ** 2   # sum conj(B(y)) * x / N^2
    # random x, y: the sums are ~sqrt(n) cancellations of terms amplified by 1/pswf (4.9e3 for W = 13.56)
    assert abs(lhs - rhs) <= (1e-11 if p["W"] < 12 else 1e-9) * abs(lhs), (lhs, rhs)
