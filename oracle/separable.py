"""
Separable synthetic facets with a cheap exact oracle.  TEST INFRASTRUCTURE ONLY
(imported by ``tests/`` and by the parity leg of ``bench.py``; never by the
product package).

The SwiFTly forward pass is linear and every primitive acts along ONE axis
(reference core.py:189-325, composition api_helper.py:73-112, 200-210).  For a
facet that is a sum of outer products

    facet_j = sum_r  a_{j,r} (x) b_{j,r}        (dense, [yB, yB])

the reference result for a subgrid is therefore

    finish_subgrid( sum_j sum_r  A_{j,r} (x) B_{j,r} ),
    A = add_to_subgrid(extract_from_facet(prepare_facet(a * mask0, off0_j), sg.off0), off0_j)   (1-D, length xM)
    B = the same along axis 1 with (b * mask1, off1_j, sg.off1)

i.e. it only needs the 1-D oracle primitives on vectors plus one 2-D
``finish_subgrid`` of an [xM, xM] array -- O(seconds) even at N = 65536, where
the direct 2-D oracle would need ~10 GB and minutes per facet.  The facets are
nevertheless DENSE random-looking [yB, yB] arrays that exercise every row and
column of the HIP kernels at full size.

Vector components are multiples of 1/8 with |.| <= 3, so every product and the
rank-``R`` sum is exactly representable in float32: the device facet
(``torch.outer`` in complex64) and the oracle's complex128 facet are the SAME
numbers and no input-rounding floor enters the comparison.

Point sources (reference fourier_algorithm.py:218-264 placement rule) are
rank-1 terms with delta vectors; ``point_source_pixels`` gives their pixel
placement for building the device facet and ``make_subgrid_from_sources`` (the
direct DFT, independent of the algorithm) the truth.

The subgrid -> facet direction (reference core.py:328-484, composition
api_helper.py:115-197) is linear and axis-separable in exactly the same way:
for subgrids that are sums of outer products  sg_i = sum_r u_{i,r} (x) v_{i,r}
the finished facet j is

    facet_j = sum_i sum_r  A_{i,j,r} (x) B_{i,j,r},
    A = mask0_j * finish_facet(add_to_facet(extract_from_subgrid(prepare_subgrid(u, sg.off0), off0_j), sg.off0), off0_j)
    B = the same along axis 1 with (v, sg.off1, off1_j, mask1_j)                     (1-D, length yB)

(``SeparableBackwardOracle``): any facet row or pixel costs ``S * R`` multiply-adds
per output element once the 1-D chains are done, also at N = 65536 where the
2-D oracle replica of SwiftlyBackward would need tens of GB per facet.
"""
import numpy

from . import swiftly_oracle as orc

__all__ = [
    "facet_vectors",
    "subgrid_vectors",
    "SeparableOracle",
    "SeparableBackwardOracle",
    "backward_contribution",
    "point_source_pixels",
    "pick_subgrids",
]


def facet_vectors(seed, yB, rank=2):
    """``(a, b)`` complex128 arrays ``[rank, yB]`` with components on the 1/8 grid."""
    rng = numpy.random.default_rng(seed)

    def vec():
        re = numpy.clip(numpy.round(rng.standard_normal((rank, yB)) * 8) / 8, -3, 3)
        im = numpy.clip(numpy.round(rng.standard_normal((rank, yB)) * 8) / 8, -3, 3)
        return re + 1j * im

    return vec(), vec()


def subgrid_vectors(seed, xA, rank=1):
    """``(u, v)`` complex128 arrays ``[rank, xA]`` on the 1/8 grid (see :py:func:`facet_vectors`): a subgrid
    ``sum_r u_r (x) v_r`` built from them in complex64 is exact."""
    return facet_vectors(seed, xA, rank)


def point_source_pixels(sources, image_size, item):
    """Pixel placements ``[(p0, p1, value)]`` of 2-D point sources
    ``(intensity, x0, x1)`` on the facet ``item`` (CoverItem / FacetConfig),
    masks applied -- the sparse form of reference
    fourier_algorithm.py:218-264 (``make_facet_from_sources``)."""
    size = item.size
    out = []
    for intensity, c0, c1 in sources:
        p0 = (c0 - (item.off0 - size // 2)) % image_size
        p1 = (c1 - (item.off1 - size // 2)) % image_size
        if p0 < size and p1 < size:
            val = complex(intensity)
            if item.mask0 is not None:
                val *= item.mask0[p0]
            if item.mask1 is not None:
                val *= item.mask1[p1]
            if val != 0:
                out.append((int(p0), int(p1), val))
    return out


class SeparableOracle:
    """Oracle results for subgrids of a forward pass over separable facets.

    :param core: ``OracleCore``
    :param facet_items: facet cover items (offsets, size, masks)
    :param vectors: per facet ``(a[R, yB], b[R, yB])`` or None
    :param pixels: per facet list of ``(p0, p1, value)`` point placements (already masked) or None
    """

    def __init__(self, core, facet_items, vectors, pixels=None):
        self.core = core
        self.items = facet_items
        self.prep = []  # per facet: list of (prepared axis-0 vector, prepared axis-1 vector)
        for j, item in enumerate(facet_items):
            terms = []
            vec = vectors[j] if vectors is not None else None
            if vec is not None:
                a, b = vec
                m0 = item.mask0 if item.mask0 is not None else 1.0
                m1 = item.mask1 if item.mask1 is not None else 1.0
                for r in range(a.shape[0]):
                    terms.append((a[r] * m0, b[r] * m1))
            for p0, p1, val in (pixels[j] if pixels is not None and pixels[j] else []):
                da = numpy.zeros(item.size, dtype=complex)
                db = numpy.zeros(item.size, dtype=complex)
                da[p0], db[p1] = val, 1.0
                terms.append((da, db))
            self.prep.append(
                [
                    (core.prepare_facet(ta, item.off0, axis=0), core.prepare_facet(tb, item.off1, axis=0))
                    for ta, tb in terms
                ]
            )

    def contribution(self, j, sg):
        """The ``[m, m]`` contribution of facet ``j`` to subgrid ``sg`` (what the
        reference ships between workers, api.py:263-277)."""
        core = self.core
        m = core.xM_yN_size
        out = numpy.zeros((m, m), dtype=complex)
        for pa, pb in self.prep[j]:
            out += numpy.outer(
                core.extract_from_facet(pa, sg.off0, axis=0), core.extract_from_facet(pb, sg.off1, axis=0)
            )
        return out

    def subgrid(self, sg):
        """Finished, masked subgrid ``[size, size]`` (api_helper.py:73-112)."""
        core = self.core
        xM = core.xM_size
        acc = numpy.zeros((xM, xM), dtype=complex)
        for item, terms in zip(self.items, self.prep):
            for pa, pb in terms:
                A = core.add_to_subgrid(core.extract_from_facet(pa, sg.off0, axis=0), item.off0, axis=0)
                B = core.add_to_subgrid(core.extract_from_facet(pb, sg.off1, axis=0), item.off1, axis=0)
                acc += numpy.outer(A, B)
        res = core.finish_subgrid(acc, [sg.off0, sg.off1], sg.size)
        if sg.mask0 is not None:
            res = res * numpy.asarray(sg.mask0)[:, None]
        if sg.mask1 is not None:
            res = res * numpy.asarray(sg.mask1)[None, :]
        return res


class SeparableBackwardOracle:
    """Oracle results for the facets of a backward pass over separable subgrids.

    Follows the reference composition (api_helper.py:115-197: ``prepare_and_split_subgrid`` ->
    ``accumulate_column`` -> ``accumulate_facet`` -> ``finish_facet``, all 1-D linear maps applied along one axis
    each; facet masks after the ``finish_facet`` of their axis, api_helper.py:175-176, 195-196) with the sums over
    subgrids deferred to the very end, which linearity allows.

    :param core: ``OracleCore``
    :param facet_items: facet cover items (offsets, size, masks)
    :param sg_items: subgrid cover items (offsets, size, masks); the subgrid masks are applied to the vectors
        here (the reference's subgrids arrive masked)
    :param vectors: per subgrid ``(u[R, xA], v[R, xA])``
    """

    def __init__(self, core, facet_items, sg_items, vectors):
        self.core = core
        self.items = facet_items
        off0s = sorted({f.off0 for f in facet_items})
        off1s = sorted({f.off1 for f in facet_items})
        sizes = {f.size for f in facet_items}
        if len(sizes) != 1:
            raise ValueError("facets of one size expected")
        yB = sizes.pop()
        # A[off0_f][i*R + r] / B[off1_f][...]: the 1-D chains WITHOUT the facet masks (facets sharing an offset share them)
        self.A = {o: [] for o in off0s}
        self.B = {o: [] for o in off1s}
        for sg, (u, v) in zip(sg_items, vectors):
            m0 = sg.mask0 if sg.mask0 is not None else 1.0
            m1 = sg.mask1 if sg.mask1 is not None else 1.0
            for r in range(u.shape[0]):
                pu = core.prepare_subgrid(u[r] * m0, sg.off0)
                pv = core.prepare_subgrid(v[r] * m1, sg.off1)
                for o in off0s:
                    self.A[o].append(self._chain(pu, sg.off0, o, yB))
                for o in off1s:
                    self.B[o].append(self._chain(pv, sg.off1, o, yB))
        self.A = {o: numpy.array(rows) for o, rows in self.A.items()}  # [S*R, yB]
        self.B = {o: numpy.array(rows) for o, rows in self.B.items()}

    def _chain(self, prepared, sg_off, facet_off, yB):
        core = self.core
        contrib = core.extract_from_subgrid(prepared, facet_off, axis=0)   # core.py:370-406
        padded = core.add_to_facet(contrib, sg_off, axis=0)                # core.py:408-449
        return core.finish_facet(padded, facet_off, yB, axis=0)            # core.py:452-484

    def facet_rows(self, j, rows):
        """Rows ``rows`` of finished facet ``j``: ``[len(rows), yB]`` (facet masks applied)."""
        item = self.items[j]
        A, B = self.A[item.off0], self.B[item.off1]
        rows = numpy.asarray(rows)
        left = A[:, rows]
        if item.mask0 is not None:
            left = left * numpy.asarray(item.mask0)[rows][None, :]
        out = left.T @ B
        if item.mask1 is not None:
            out = out * numpy.asarray(item.mask1)[None, :]
        return out

    def facet(self, j):
        """The whole finished facet ``j`` (small configurations only)."""
        return self.facet_rows(j, numpy.arange(self.items[j].size))


def backward_contribution(core, u, v, sg, facet):
    """``[m, m]`` contribution of the rank-R subgrid ``sum_r u_r (x) v_r`` (masks of ``sg`` applied here) to ``facet``:
    ``prepare_and_split_subgrid`` (api_helper.py:115-139) for separable data."""
    m = core.xM_yN_size
    out = numpy.zeros((m, m), dtype=complex)
    m0 = sg.mask0 if sg.mask0 is not None else 1.0
    m1 = sg.mask1 if sg.mask1 is not None else 1.0
    for r in range(u.shape[0]):
        ca = core.extract_from_subgrid(core.prepare_subgrid(u[r] * m0, sg.off0), facet.off0, axis=0)
        cb = core.extract_from_subgrid(core.prepare_subgrid(v[r] * m1, sg.off1), facet.off1, axis=0)
        out += numpy.outer(ca, cb)
    return out


def pick_subgrids(sg_items, count=6):
    """A spread of ``count`` subgrids from a (sparse) set: the centre, the
    extreme offsets along both axes (these are the wrapped ones in a cover that
    wraps around the grid origin) and evenly spaced others."""
    n = len(sg_items)
    if n <= count:
        return list(range(n))
    key0 = [s.off0 for s in sg_items]
    key1 = [s.off1 for s in sg_items]
    picks = [
        min(range(n), key=lambda i: (key0[i], key1[i])),
        max(range(n), key=lambda i: (key0[i], key1[i])),
        min(range(n), key=lambda i: (key1[i], -key0[i])),
        max(range(n), key=lambda i: (key1[i], -key0[i])),
    ]
    step = max(1, n // (count + 1))
    for i in range(step // 2, n, step):
        if len(set(picks)) >= count:
            break
        picks.append(i)
    uniq = []
    for i in picks:
        if i not in uniq:
            uniq.append(i)
    return uniq[:count]
