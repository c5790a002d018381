"""
CPU oracle for the SwiFTly facet<->subgrid hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference algorithm
(ska_sdp_exec_swiftly 1.0.0).  It is used exclusively as the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package
(``ska-sdp-distributed-fourier-transform_amd/``) never imports it and has no
CPU fallback.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
package itself (``/root/reference/src``, numpy backend) in the authoring
container, runs every primitive and the 2-D task bodies on seeded inputs and
commits the input/output vectors under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors,
and ``tests/test_oracle_known_answers.py`` re-derives the reference's own
known-answer tests (``tests/test_core.py``) against it.

Formulation.  The reference composes ``numpy.roll`` / ``pad_mid`` /
``extract_mid`` / ``fftshift`` calls.  The oracle instead uses the closed
index forms of those compositions (one modular gather or scatter per
primitive), which is also the form the HIP kernels
implement -- so a disagreement between oracle and kernel localises to an
index map or to the FFT, never to helper plumbing.

All arrays are "centred": index ``n // 2`` is the origin.  Notation:
``yN`` padded facet size, ``xM`` padded subgrid size, ``m = xM*yN/N``
contribution size, ``s = subgrid_off*yN//N``, ``sp = facet_off*xM//N``.
"""

import numpy
import scipy.special

__all__ = [
    "OracleCore",
    "make_facet_from_sources",
    "make_subgrid_from_sources",
    "CoverItem",
    "make_full_cover",
    "extract_column",
    "sum_and_finish_subgrid",
    "prepare_and_split_subgrid",
    "accumulate_column",
    "accumulate_facet",
    "finish_facet_2d",
    "forward_all",
    "backward_all",
]


# ---------------------------------------------------------------------------
# centred transforms (reference: fourier_algorithm.py:96-122)
# ---------------------------------------------------------------------------
def cfft(a, axis):
    """Centred forward FFT: fftshift(fft(ifftshift(a))).

    Reference fourier_algorithm.py:96-107.  For even and odd n alike
    ifftshift == roll(-(n//2)) and fftshift == roll(+(n//2)).
    """
    n = a.shape[axis]
    return numpy.roll(
        numpy.fft.fft(numpy.roll(a, -(n // 2), axis=axis), axis=axis),
        n // 2,
        axis=axis,
    )


def cifft(a, axis):
    """Centred inverse FFT (1/n normalised).  Reference
    fourier_algorithm.py:110-122."""
    n = a.shape[axis]
    return numpy.roll(
        numpy.fft.ifft(numpy.roll(a, -(n // 2), axis=axis), axis=axis),
        n // 2,
        axis=axis,
    )


def _bshape(vec, ndim, axis):
    """View a 1-D vector so that it broadcasts along ``axis`` of an
    ``ndim``-dimensional array (reference fourier_algorithm.py:38-50)."""
    shape = [1] * ndim
    shape[axis] = len(vec)
    return numpy.reshape(vec, shape)


def _gather(a, idx, axis):
    return numpy.take(a, idx, axis=axis)


def _scatter(a, idx, n, axis):
    """Zero array of length ``n`` along ``axis`` with ``a`` placed at the
    (distinct) positions ``idx``."""
    shape = list(a.shape)
    shape[axis] = n
    out = numpy.zeros(shape, dtype=a.dtype)
    sl = [slice(None)] * a.ndim
    sl[axis] = idx
    out[tuple(sl)] = a
    return out


# ---------------------------------------------------------------------------
# core primitives
# ---------------------------------------------------------------------------
class OracleCore:
    """Numpy restatement of ``SwiftlyCore`` (reference core.py:20-484).

    Constructor argument order follows core.py:39 -- ``(W, N, xM_size,
    yN_size)``.
    """

    def __init__(self, W, N, xM_size, yN_size):
        self.W = W
        self.N = N
        self.xM_size = xM_size
        self.yN_size = yN_size
        # core.py:55-74
        if N % yN_size != 0:
            raise ValueError(f"Image size {N} not divisible by facet size {yN_size}!")
        if N % xM_size != 0:
            raise ValueError(f"Image size {N} not divisible by subgrid size {xM_size}!")
        if (xM_size * yN_size) % N != 0:
            raise ValueError("Contribution size not integer!")
        self.xM_yN_size = xM_size * yN_size // N  # core.py:48
        self.pswf = self._pswf()
        # core.py:104-117
        self.Fb = 1.0 / self.pswf[1:]
        step = N // xM_size
        self.Fn = self.pswf[(yN_size // 2) % step :: step]

    # core.py:76-92
    @property
    def subgrid_off_step(self):
        return self.N // self.yN_size

    @property
    def facet_off_step(self):
        return self.N // self.xM_size

    def _pswf(self):
        """PSWF sampled at 2*(k - yN/2)/yN, k < yN, with pswf[0] = 0.

        Reference core.py:119-150: ``scipy.special.pro_ang1(0, 0, pi*W/2, x)``
        evaluated in chunks of 500 (scipy work-around kept so that chunk
        boundaries -- and hence results -- are identical).
        """
        yN = self.yN_size
        n2 = yN // 2
        if yN % 2 == 0:
            x = 2.0 * numpy.arange(-n2, n2) / yN
        else:
            x = 2.0 * numpy.arange(-n2, n2 + 1) / yN
        out = numpy.empty(yN, dtype=float)
        for lo in range(1, yN, 500):
            out[lo : lo + 500] = scipy.special.pro_ang1(
                0, 0, numpy.pi * self.W / 2, x[lo : lo + 500]
            )[0]
        out[0] = 0.0
        return out

    # -- window helpers ----------------------------------------------------
    def facet_window(self, facet_size):
        """``extract_mid(Fb, facet_size)`` (core.py:216, 476) in closed form:
        element y is ``1 / pswf[yN//2 - facet_size//2 + y]`` for both
        parities of ``facet_size``."""
        lo = self.yN_size // 2 - facet_size // 2
        return 1.0 / self.pswf[lo : lo + facet_size]

    def _s(self, subgrid_off):
        return subgrid_off * self.yN_size // self.N

    def _sp(self, facet_off):
        return facet_off * self.xM_size // self.N

    @staticmethod
    def _out(result, out, add=False):
        """Out-array convention of core.py:152-186."""
        if out is None:
            return result
        if out.shape != result.shape:
            raise ValueError(f"Output shape is {out.shape}, expected {result.shape}!")
        if add:
            out[:] += result
        else:
            out[:] = result
        return out

    # -- facet -> subgrid --------------------------------------------------
    def prepare_facet(self, facet, facet_off, axis, out=None):
        """core.py:189-222.  out[k] = cifft_yN( scatter_{(yN/2 - yB//2 + y +
        facet_off) mod yN} facet[y] / pswf[yN/2 - yB//2 + y] )."""
        yN = self.yN_size
        yB = facet.shape[axis]
        y = numpy.arange(yB)
        pos = (yN // 2 - yB // 2 + y + facet_off) % yN
        w = _bshape(self.facet_window(yB), facet.ndim, axis)
        return self._out(cifft(_scatter(facet * w, pos, yN, axis), axis), out)

    def extract_from_facet(self, prep, subgrid_off, axis, out=None):
        """core.py:224-253.  out[(i+s) mod m] = prep[(yN/2 - m/2 + i + s)
        mod yN], i < m."""
        yN, m = self.yN_size, self.xM_yN_size
        s = self._s(subgrid_off)
        j = numpy.arange(m)  # output index
        i = (j - s) % m
        src = (yN // 2 - m // 2 + i + s) % yN
        return self._out(_gather(prep, src, axis), out)

    def add_to_subgrid(self, contrib, facet_off, axis, out=None):
        """core.py:255-285.  out[(k + xM/2 - m/2 + sp) mod xM] +=
        Fn[k] * cfft_m(contrib)[(k + sp) mod m], k < m."""
        xM, m = self.xM_size, self.xM_yN_size
        sp = self._sp(facet_off)
        k = numpy.arange(m)
        F = cfft(contrib, axis)
        vals = _gather(F, (k + sp) % m, axis) * _bshape(self.Fn, contrib.ndim, axis)
        res = _scatter(vals, (k + xM // 2 - m // 2 + sp) % xM, xM, axis)
        return self._out(res, out, add=True)

    def finish_subgrid(self, summed, subgrid_off, subgrid_size, out=None):
        """core.py:287-325.  Per axis: out[i] = cifft_xM(acc)[(xM/2 - xA//2 +
        i + subgrid_off) mod xM], i < xA.  ``subgrid_off`` is an int for 1-D
        input and a list for 2-D (core.py:306-312)."""
        xM = self.xM_size
        dims = summed.ndim
        if not isinstance(subgrid_off, list):
            if dims != 1:
                raise ValueError("Subgrid offset must be given for every dimension!")
            subgrid_off = [subgrid_off]
        tmp = summed
        i = numpy.arange(subgrid_size)
        for axis in range(dims):
            src = (xM // 2 - subgrid_size // 2 + i + subgrid_off[axis]) % xM
            tmp = _gather(cifft(tmp, axis), src, axis)
        return self._out(tmp, out)

    # -- subgrid -> facet --------------------------------------------------
    def prepare_subgrid(self, subgrid, subgrid_off, out=None):
        """core.py:328-368.  Per axis: p[(xM/2 - xA//2 + i + off) mod xM] =
        sg[i]; out = cfft_xM(p)."""
        xM = self.xM_size
        dims = subgrid.ndim
        if dims == 1 and not isinstance(subgrid_off, (tuple, list)):
            subgrid_off = (subgrid_off,)
        if len(subgrid_off) != dims:
            raise ValueError("Dimensionality mismatch between subgrid and offsets!")
        tmp = subgrid
        for axis in range(dims):
            xA = tmp.shape[axis]
            pos = (xM // 2 - xA // 2 + numpy.arange(xA) + subgrid_off[axis]) % xM
            tmp = cfft(_scatter(tmp, pos, xM, axis), axis)
        return self._out(tmp, out)

    def extract_from_subgrid(self, FSi, facet_off, axis, out=None):
        """core.py:370-406.  g[(k+sp) mod m] = Fn[k] * FS[(k + xM/2 - m/2 +
        sp) mod xM]; out = cifft_m(g)."""
        xM, m = self.xM_size, self.xM_yN_size
        sp = self._sp(facet_off)
        k = numpy.arange(m)
        vals = _gather(FSi, (k + xM // 2 - m // 2 + sp) % xM, axis) * _bshape(
            self.Fn, FSi.ndim, axis
        )
        g = _scatter(vals, (k + sp) % m, m, axis)
        return self._out(cifft(g, axis), out)

    def add_to_facet(self, contrib, subgrid_off, axis, out=None):
        """core.py:408-449.  out[(yN/2 - m/2 + i + s) mod yN] +=
        contrib[(i + s) mod m], i < m."""
        yN, m = self.yN_size, self.xM_yN_size
        s = self._s(subgrid_off)
        i = numpy.arange(m)
        vals = _gather(contrib, (i + s) % m, axis)
        res = _scatter(vals, (yN // 2 - m // 2 + i + s) % yN, yN, axis)
        return self._out(res, out, add=True)

    def finish_facet(self, acc, facet_off, facet_size, axis, out=None):
        """core.py:452-484.  out[y] = cfft_yN(acc)[(yN/2 - yB//2 + y +
        facet_off) mod yN] / pswf[yN/2 - yB//2 + y], y < yB."""
        yN = self.yN_size
        y = numpy.arange(facet_size)
        src = (yN // 2 - facet_size // 2 + y + facet_off) % yN
        w = _bshape(self.facet_window(facet_size), acc.ndim, axis)
        return self._out(_gather(cfft(acc, axis), src, axis) * w, out)


# ---------------------------------------------------------------------------
# truth generators (reference fourier_algorithm.py:218-315)
# ---------------------------------------------------------------------------
def make_facet_from_sources(sources, image_size, facet_size, facet_offsets, facet_masks=None):
    """Place point sources on a facet (fourier_algorithm.py:218-264).

    A source at image coordinate c lands on pixel (c - off + size//2) mod N
    of each axis if that is < size.
    """
    dims = len(facet_offsets)
    facet = numpy.zeros(dims * [facet_size], dtype=complex)
    origin = numpy.asarray(facet_offsets, dtype=int) - facet_size // 2
    for intensity, *coord in sources:
        pix = numpy.mod(numpy.asarray(coord) - origin, image_size)
        if numpy.all(pix < facet_size):
            facet[tuple(pix)] += intensity
    for axis, mask in enumerate(facet_masks or []):
        if mask is not None:
            facet *= _bshape(numpy.asarray(mask), dims, axis)
    return facet


def make_subgrid_from_sources(sources, image_size, subgrid_size, subgrid_offsets, subgrid_masks=None):
    """Direct DFT of point sources on a subgrid (fourier_algorithm.py:267-315):
    sg[u] = sum_src I/N^dims * exp(2 pi i <u, c> / N), u = off - size//2 + i.
    """
    dims = len(subgrid_offsets)
    sg = numpy.zeros(dims * [subgrid_size], dtype=complex)
    us = [
        numpy.arange(off - subgrid_size // 2, off + (subgrid_size + 1) // 2)
        for off in subgrid_offsets
    ]
    for intensity, *coord in sources:
        term = numpy.asarray(intensity / image_size**dims, dtype=complex)
        for axis, (u, c) in enumerate(zip(us, coord)):
            term = term * _bshape(
                numpy.exp(2j * numpy.pi / image_size * u * c), dims, axis
            )
        sg += term
    for axis, mask in enumerate(subgrid_masks or []):
        if mask is not None:
            sg *= _bshape(numpy.asarray(mask), dims, axis)
    return sg


# ---------------------------------------------------------------------------
# covers (reference api_helper.py:213-253, api.py:39-104)
# ---------------------------------------------------------------------------
class CoverItem:
    """Offset/size/mask record; stands in for FacetConfig and SubgridConfig
    (api.py:39-104) with the masks already expanded to 0/1 float vectors."""

    def __init__(self, off0, off1, size, mask0=None, mask1=None):
        self.off0, self.off1, self.size = int(off0), int(off1), int(size)
        self.mask0, self.mask1 = mask0, mask1


def make_full_cover(N, chunk):
    """Full cover of an N x N plane with ``chunk``-sized pieces whose masks
    split overlaps at the midpoint between neighbouring offsets
    (api_helper.py:213-240)."""
    offs = chunk * numpy.arange(int(numpy.ceil(N / chunk)))
    border = (offs + numpy.hstack([offs[1:], [N + offs[0]]])) // 2
    masks = []
    for i, off in enumerate(offs):
        left = (border[i - 1] - off + chunk // 2) % N
        right = border[i] - off + chunk // 2
        mk = numpy.zeros(chunk)
        mk[left:right] = 1
        masks.append(mk)
    return [
        CoverItem(o0, o1, chunk, masks[i0], masks[i1])
        for i0, o0 in enumerate(offs)
        for i1, o1 in enumerate(offs)
    ]


# ---------------------------------------------------------------------------
# 2-D task bodies (reference api_helper.py:73-210)
# ---------------------------------------------------------------------------
def extract_column(core, BF_F, subgrid_off0, facet_off1):
    """api_helper.py:200-210."""
    return core.prepare_facet(
        core.extract_from_facet(BF_F, subgrid_off0, axis=0), facet_off1, axis=1
    )


def sum_and_finish_subgrid(core, contribs, facet_items, sg):
    """api_helper.py:73-112: group by facet off1, sum axis 0 within a group,
    then axis 1 over groups, finish, apply subgrid masks."""
    acc = None
    for off1 in sorted({f.off1 for f in facet_items}):
        col = None
        for f, c in zip(facet_items, contribs):
            if f.off1 == off1:
                col = core.add_to_subgrid(c, f.off0, axis=0, out=col)
        acc = core.add_to_subgrid(col, off1, axis=1, out=acc)
    res = core.finish_subgrid(acc, [sg.off0, sg.off1], sg.size)
    if sg.mask0 is not None:
        res = res * sg.mask0[:, None]
    if sg.mask1 is not None:
        res = res * sg.mask1[None, :]
    return res


def prepare_and_split_subgrid(core, subgrid, offs, facet_items):
    """api_helper.py:115-139."""
    prepared = core.prepare_subgrid(subgrid, list(offs))
    by_off0 = {
        off0: core.extract_from_subgrid(prepared, off0, axis=0)
        for off0 in {f.off0 for f in facet_items}
    }
    return [
        core.extract_from_subgrid(by_off0[f.off0], f.off1, axis=1) for f in facet_items
    ]


def accumulate_column(core, NAF_NAF, NAF_MNAF, subgrid_off1):
    """api_helper.py:142-152."""
    return core.add_to_facet(NAF_NAF, subgrid_off1, axis=1, out=NAF_MNAF)


def accumulate_facet(core, NAF_MNAF, MNAF_BMNAF, facet, sg_off0):
    """api_helper.py:155-179."""
    t = core.finish_facet(NAF_MNAF, facet.off1, facet.size, axis=1)
    if facet.mask1 is not None:
        t = t * facet.mask1[None, :]
    return core.add_to_facet(t, sg_off0, axis=0, out=MNAF_BMNAF)


def finish_facet_2d(core, MNAF_BMNAF, facet):
    """api_helper.py:182-197 (without the AttributeError of 184-187: a facet
    that received nothing is all zeros)."""
    if MNAF_BMNAF is None:
        return numpy.zeros((facet.size, facet.size), dtype=complex)
    t = core.finish_facet(MNAF_BMNAF, facet.off0, facet.size, axis=0)
    if facet.mask0 is not None:
        t = t * facet.mask0[:, None]
    return t


def forward_all(core, facet_items, facets, subgrid_items):
    """Serial replica of SwiftlyForward (api.py:238-324): returns the list of
    finished subgrids in the order of ``subgrid_items``."""
    BF_Fs = [core.prepare_facet(d, f.off0, axis=0) for f, d in zip(facet_items, facets)]
    cols = {}
    out = []
    for sg in subgrid_items:
        if sg.off0 not in cols:
            cols = {
                sg.off0: [
                    extract_column(core, BF, sg.off0, f.off1)
                    for f, BF in zip(facet_items, BF_Fs)
                ]
            }
        contribs = [
            core.extract_from_facet(c, sg.off1, axis=1) for c in cols[sg.off0]
        ]
        out.append(sum_and_finish_subgrid(core, contribs, facet_items, sg))
    return out


def backward_all(core, facet_items, subgrid_items, subgrids):
    """Serial replica of SwiftlyBackward (api.py:347-463): returns finished
    facets in the order of ``facet_items``."""
    F = len(facet_items)
    MNAF_BMNAFs = [None] * F
    columns = {}
    order = []
    for sg, data in zip(subgrid_items, subgrids):
        parts = prepare_and_split_subgrid(core, data, [sg.off0, sg.off1], facet_items)
        if sg.off0 not in columns:
            columns[sg.off0] = [None] * F
            order.append(sg.off0)
        columns[sg.off0] = [
            accumulate_column(core, p, old, sg.off1)
            for p, old in zip(parts, columns[sg.off0])
        ]
    for off0 in order:
        MNAF_BMNAFs = [
            accumulate_facet(core, col, acc, f, off0)
            for f, col, acc in zip(facet_items, columns[off0], MNAF_BMNAFs)
        ]
    return [finish_facet_2d(core, acc, f) for f, acc in zip(facet_items, MNAF_BMNAFs)]
