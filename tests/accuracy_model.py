"""
complex64 error budget of the forward pipeline, emulated on the CPU (test infrastructure; uses the oracle).

The HIP forward pipeline stores six intermediates in complex64: K1 output (band buffer), the
four-step scratch inside K2, K2 output, K3 output, the half-finished subgrid, the subgrid.  This module replays
the same chain -- same axis order, same stored intermediates, windows in float32 -- on one facet and one subgrid
with every stage's ARITHMETIC selectable between float32 (scipy's pocketfft computes complex64 transforms in single
precision; numpy's does not do so consistently) and float64, and every stored intermediate rounded to complex64
explicitly.  With float64 arithmetic everywhere the result is the float32 STORAGE floor; switching single stages to
float32 shows which stage's arithmetic sets the end-to-end error (r3 review: the storage-floor figure of rounds 2/3
came from a probe that trusted numpy's complex64 ifft to be exact, and was 4x too high).
"""
import numpy
import scipy.fft

from oracle import swiftly_oracle as orc


def _c64(a):
    return a.astype(numpy.complex64)


def _fft(a, axis, bits, inverse=False):
    """unnormalised forward / normalised inverse transform in float32 or float64 arithmetic"""
    a = a.astype(numpy.complex64 if bits == 32 else numpy.complex128)
    f = scipy.fft.ifft if inverse else scipy.fft.fft
    return f(a, axis=axis)


def _centred(a, axis, bits, inverse, mid_round=None):
    n = a.shape[axis]
    a = numpy.roll(a, -(n // 2), axis=axis)
    if mid_round is None:
        r = _fft(a, axis, bits, inverse)
    else:
        r = _fourstep(a, axis, bits, inverse, *mid_round)
    return numpy.roll(r, n // 2, axis=axis)


def _fourstep(a, axis, bits, inverse, n1, n2):
    """four-step transform along `axis` (input index y1*n2 + y2, output k1 + n1*k2) with the intermediate rounded to
    complex64 -- the K2 scratch.  Inverse through conjugation."""
    a = numpy.moveaxis(a, axis, 0)
    n = a.shape[0]
    assert n == n1 * n2
    ct = numpy.complex64 if bits == 32 else numpy.complex128
    x = (numpy.conj(a) if inverse else a).astype(ct)
    rest = x.shape[1:]
    x = x.reshape((n1, n2) + rest)                        # [y1, y2]
    y = scipy.fft.fft(x, axis=0)                          # [k1, y2]
    k1 = numpy.arange(n1).reshape((n1, 1) + (1,) * len(rest))
    y2 = numpy.arange(n2).reshape((1, n2) + (1,) * len(rest))
    tw = numpy.exp(-2j * numpy.pi * ((k1 * y2) % n) / n).astype(ct)
    y = (y * tw).astype(numpy.complex64)                  # the scratch, stored in complex64
    z = scipy.fft.fft(y.astype(ct), axis=1)               # [k1, k2]
    out = numpy.moveaxis(z, 0, 1).reshape((n,) + rest)    # k1 + n1*k2
    if inverse:
        out = numpy.conj(out) / n
    return numpy.moveaxis(out, 0, axis)


def forward_chain(P, facet, fo, so, bits, scratch_split=None, round_stores=True):
    """One facet -> one subgrid through the HIP pipeline's dataflow.  bits = dict(k1=, k2=, k3=, sf=, k5=) of 32 | 64;
    scratch_split = (n1, n2) emulates the complex64 four-step scratch of K2.  Returns the finished [xA, xA] subgrid
    contribution of this facet (complex128 values of complex64-representable numbers when round_stores)."""
    c = orc.OracleCore(P["W"], P["N"], P["xM"], P["yN"])
    yB, yN, xA, xM, m = P["yB"], P["yN"], P["xA"], P["xM"], c.xM_yN_size
    rnd = _c64 if round_stores else (lambda a: a)
    w = c.facet_window(yB).astype(numpy.float32)
    wt = {32: numpy.float32, 64: numpy.float64}
    y = numpy.arange(yB)
    pos = lambda off: (yN // 2 - yB // 2 + y + off) % yN  # noqa: E731
    # K1: both windows, pad + shift + inverse transform along axis 1 (contiguous), stored in complex64
    t = bits["k1"]
    x = facet.astype(numpy.complex64 if t == 32 else numpy.complex128)
    x = x * w.astype(wt[t])[None, :] * w.astype(wt[t])[:, None]
    k1 = numpy.zeros((yB, yN), dtype=x.dtype)
    k1[:, pos(fo[1])] = x
    k1 = rnd(_centred(k1, 1, t, True))
    # K2: column window (extract_from_facet axis 1), pad + shift + inverse transform along axis 0, row window kept
    t = bits["k2"]
    col = c.extract_from_facet(k1, so[1], 1)                      # gather, exact
    k2 = numpy.zeros((yN, m), dtype=col.dtype)
    k2[pos(fo[0]), :] = col
    k2 = _centred(k2, 0, t, True, mid_round=scratch_split)
    q = rnd(c.extract_from_facet(k2, so[0], 0))                   # [m, m]
    # K3: axis-0 half of add_to_subgrid (m-point transform, Fn) -- kept unplaced (G), stored in complex64
    t = bits["k3"]
    sp = [c._sp(fo[0]), c._sp(fo[1])]
    k = numpy.arange(m)
    fn = c.Fn.astype(numpy.float32).astype(wt[t])
    g = _centred(q, 0, t, False)
    g = rnd(g[(k + sp[0]) % m, :] * fn[:, None])
    # sum_finish: axis-1 m-point transform, Fn, placement, xM-point inverse along axis 1, crop to xA
    t = bits["sf"]
    fn = c.Fn.astype(numpy.float32).astype(wt[t])
    h = _centred(g, 1, t, False)
    h = h[:, (k + sp[1]) % m] * fn[None, :]
    acc = numpy.zeros((m, xM), dtype=h.dtype)
    acc[:, (k + xM // 2 - m // 2 + sp[1]) % xM] = h
    i = numpy.arange(xA)
    acc = _centred(acc, 1, t, True)[:, (xM // 2 - xA // 2 + i + so[1]) % xM]
    acc = rnd(acc)
    # K5b: placement along axis 0, xM-point inverse, crop
    t = bits["k5"]
    full = numpy.zeros((xM, xA), dtype=acc.dtype)
    full[(k + xM // 2 - m // 2 + sp[0]) % xM, :] = acc
    out = _centred(full, 0, t, True)[(xM // 2 - xA // 2 + i + so[0]) % xM, :]
    return rnd(out).astype(complex)


def forward_chain_reordered(P, facet, fo, so, bits, scratch_split=None, round_stores=True, halves=False):
    """The r5 review's reordering (its item 3): axis 1 is FINISHED before the strided-axis work starts.  K1's epilogue takes
    the wave's m-column window of the transformed row, runs the axis-1 half of add_to_subgrid on it (m-point transform x Fn,
    core.py:255-285) and stores THAT; K2 and K3 then work on data that carries the axis-0 window only, and sum_finish loses
    its m-point transforms (placement + xM-point inverse + crop).  Same stored intermediates otherwise.  bits as in
    forward_chain (k1 covers the fused epilogue).

    halves=True: the form that can live in the epilogue of the TWO-workgroup K1 (built and measured in r6, then replaced by the
    whole-row K1 that finishes the window itself) -- each of a row's two workgroups holds the outputs of one parity, so that
    K1 can only store the two decimation-in-time half spectra of the window
    (m/2-point transforms of the even / odd window samples); K2 and K3 run on those, and the radix-2 step that joins them,
    the phase of the window rotation and Fn follow K3.  The halves are ALIASED (frequency u folded onto u + m/2): the column
    passes round at the level of the window's leakage that Fn would have suppressed."""
    c = orc.OracleCore(P["W"], P["N"], P["xM"], P["yN"])
    yB, yN, xA, xM, m = P["yB"], P["yN"], P["xA"], P["xM"], c.xM_yN_size
    rnd = _c64 if round_stores else (lambda a: a)
    w = c.facet_window(yB).astype(numpy.float32)
    wt = {32: numpy.float32, 64: numpy.float64}
    y = numpy.arange(yB)
    pos = lambda off: (yN // 2 - yB // 2 + y + off) % yN  # noqa: E731
    sp = [c._sp(fo[0]), c._sp(fo[1])]
    k = numpy.arange(m)
    # K1 + epilogue: both windows, axis-1 inverse transform (row stays in registers), window gather, m-point transform, Fn
    t = bits["k1"]
    x = facet.astype(numpy.complex64 if t == 32 else numpy.complex128)
    x = x * w.astype(wt[t])[None, :] * w.astype(wt[t])[:, None]
    k1 = numpy.zeros((yB, yN), dtype=x.dtype)
    k1[:, pos(fo[1])] = x
    k1 = _centred(k1, 1, t, True)
    col = c.extract_from_facet(k1, so[1], 1)                      # [yB, m], gather
    fn = c.Fn.astype(numpy.float32).astype(wt[t])
    if halves:
        s1 = (so[1] * yN // P["N"]) % m
        b = col[:, (k + s1) % m]                                  # window samples b[i]: col[(i + s) mod m] = b[i]
        ct = numpy.complex64 if t == 32 else numpy.complex128
        g0 = scipy.fft.fft(b[:, 0::2].astype(ct), axis=1)
        g1 = scipy.fft.fft(b[:, 1::2].astype(ct), axis=1)
        h = rnd(numpy.concatenate([g0, g1], axis=1))              # stored: [yB, m] = the two half spectra
    else:
        h = _centred(col, 1, t, False)
        h = rnd(h[:, (k + sp[1]) % m] * fn[None, :])              # stored: [yB, m] per (facet, wave)
    # K2: pad + shift + inverse transform along axis 0 on data that carries the axis-0 window only
    t = bits["k2"]
    k2 = numpy.zeros((yN, m), dtype=h.dtype)
    k2[pos(fo[0]), :] = h
    k2 = _centred(k2, 0, t, True, mid_round=scratch_split)
    q = rnd(c.extract_from_facet(k2, so[0], 0))                   # [m, m]
    # K3: axis-0 m-point transform, Fn
    t = bits["k3"]
    fn = c.Fn.astype(numpy.float32).astype(wt[t])
    g = _centred(q, 0, t, False)
    g = rnd(g[(k + sp[0]) % m, :] * fn[:, None])
    # sum_finish without its m-point transform: placement along axis 1, xM-point inverse, crop
    t = bits["sf"]
    if halves:  # join the halves: Fk[ck] = W^((s - m/2) u) (G0[u'] + W^u G1[u']), u = ck - m/2; Z[k] = Fn[k] Fk[(k + s'1) mod m]
        ct = numpy.complex64 if t == 32 else numpy.complex128
        ck = (k + sp[1]) % m
        u = ck - m // 2
        W = lambda e: numpy.exp(-2j * numpy.pi * ((e % m) / m)).astype(ct)  # noqa: E731
        fnt = c.Fn.astype(numpy.float32).astype(wt[t])
        gc = g.astype(ct)
        z = gc[:, u % (m // 2)] + W(u)[None, :] * gc[:, m // 2 + u % (m // 2)]
        g = z * (W((s1 - m // 2) * u) * fnt)[None, :]
    acc = numpy.zeros((m, xM), dtype=numpy.complex64 if t == 32 else numpy.complex128)
    acc[:, (k + xM // 2 - m // 2 + sp[1]) % xM] = g
    i = numpy.arange(xA)
    acc = rnd(_centred(acc, 1, t, True)[:, (xM // 2 - xA // 2 + i + so[1]) % xM])
    # K5b
    t = bits["k5"]
    full = numpy.zeros((xM, xA), dtype=acc.dtype)
    full[(k + xM // 2 - m // 2 + sp[0]) % xM, :] = acc
    out = _centred(full, 0, t, True)[(xM // 2 - xA // 2 + i + so[0]) % xM, :]
    return rnd(out).astype(complex)


def reference_chain(P, facet, fo, so):
    """the same facet -> subgrid contribution through the oracle primitives in complex128 (reference order)"""
    c = orc.OracleCore(P["W"], P["N"], P["xM"], P["yN"])
    t = c.prepare_facet(facet.astype(complex), fo[0], axis=0)
    t = c.prepare_facet(c.extract_from_facet(t, so[0], axis=0), fo[1], axis=1)
    t = c.extract_from_facet(t, so[1], axis=1)
    t = c.add_to_subgrid(c.add_to_subgrid(t, fo[0], axis=0), fo[1], axis=1)
    return c.finish_subgrid(t, [so[0], so[1]], P["xA"])


def rel_rmse(a, b):
    return float(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2)))


PROBE = dict(W=11.0, N=8192, yB=1408, yN=2048, xA=1024, xM=2048)
PROBE_FO, PROBE_SO = (4 * 16, 4 * 32), (4 * 4, -4 * 12)


def budget(P=PROBE, fo=PROBE_FO, so=PROBE_SO, seed=5, split=(32, 64)):
    rng = numpy.random.default_rng(seed)
    yB = P["yB"]
    facet = _c64(rng.standard_normal((yB, yB)) + 1j * rng.standard_normal((yB, yB)))
    want = reference_chain(P, facet, fo, so)
    rows = {}
    all64 = dict(k1=64, k2=64, k3=64, sf=64, k5=64)
    all32 = dict(k1=32, k2=32, k3=32, sf=32, k5=32)
    rows["float64 arithmetic, no rounding of intermediates"] = rel_rmse(forward_chain(P, facet, fo, so, all64, None, False), want)
    rows["storage floor: float64 arithmetic, complex64 intermediates (no scratch)"] = rel_rmse(forward_chain(P, facet, fo, so, all64), want)
    rows["storage floor incl. the complex64 four-step scratch of K2"] = rel_rmse(forward_chain(P, facet, fo, so, all64, split), want)
    rows["float32 arithmetic everywhere (r3 kernels)"] = rel_rmse(forward_chain(P, facet, fo, so, all32, split), want)
    for st in ("k1", "k2", "k3", "sf", "k5"):
        b = dict(all64)
        b[st] = 32
        rows[f"float32 arithmetic in {st} only"] = rel_rmse(forward_chain(P, facet, fo, so, b, split), want)
    b = dict(all32)
    b["k2"] = 64
    rows["float64 arithmetic in k2 only (r4 kernels)"] = rel_rmse(forward_chain(P, facet, fo, so, b, split), want)
    return rows


def chain_error(bits, P=PROBE, fo=PROBE_FO, so=PROBE_SO, seed=5, split=(32, 64), chain=None):
    """end-to-end relative RMSE of the emulated chain with the given per-stage arithmetic"""
    rng = numpy.random.default_rng(seed)
    yB = P["yB"]
    facet = _c64(rng.standard_normal((yB, yB)) + 1j * rng.standard_normal((yB, yB)))
    return rel_rmse((chain or forward_chain)(P, facet, fo, so, bits, split), reference_chain(P, facet, fo, so))


def halves_error(P=None, fo=None, so=None, seed=5, bits=None):
    """end-to-end error of the axis-1-first chain with the contiguous-axis finish split into half spectra (float32 everywhere
    unless ``bits`` says otherwise) on the probe configuration"""
    P, fo, so = P or PROBE, fo or PROBE_FO, so or PROBE_SO
    rng = numpy.random.default_rng(seed)
    yB = P["yB"]
    facet = _c64(rng.standard_normal((yB, yB)) + 1j * rng.standard_normal((yB, yB)))
    want = reference_chain(P, facet, fo, so)
    got = forward_chain_reordered(P, facet, fo, so, bits or dict(k1=32, k2=32, k3=32, sf=32, k5=32), scratch_split=(32, 64),
                                  halves=True)
    return rel_rmse(got, want)


def reordered_budget():
    """the reordered dataflow's figures: all-float32, storage floor, and single-stage float32"""
    all64 = dict(k1=64, k2=64, k3=64, sf=64, k5=64)
    all32 = dict(k1=32, k2=32, k3=32, sf=32, k5=32)
    rows = {"reordered: float32 arithmetic everywhere": chain_error(all32, chain=forward_chain_reordered),
            "reordered: storage floor (float64 arithmetic, complex64 intermediates + scratch)": chain_error(all64, chain=forward_chain_reordered)}
    for st in ("k1", "k2", "k3", "sf", "k5"):
        b = dict(all64)
        b[st] = 32
        rows[f"reordered: float32 arithmetic in {st} only"] = chain_error(b, chain=forward_chain_reordered)
    return rows


if __name__ == "__main__":
    for name, v in budget().items():
        print(f"{v:.3e}  {name}")
    for name, v in reordered_budget().items():
        print(f"{v:.3e}  {name}")
