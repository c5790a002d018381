"""
GPU tests of the robustness contract of the drop-in boundary:
  * pickling like SwiftlyCoreFunc (reference core.py:512-525): only the four parameters travel;
  * one core shared by many host threads (the reference scatters ONE core object to every Dask worker
    thread, api.py:145-147; 38 threads/worker in slurm_scripts/run_distr_single_csd3.slurm:71);
  * the accumulate entry points refuse batch items that share output elements;
  * extract_column with a row map on a BF_F whose transform axis is strided (the generic
    four-step path used to index its scratch through the row map).
"""
import pickle
import threading

import numpy
import pytest

from oracle import swiftly_oracle as orc

pytestmark = pytest.mark.gpu

P = dict(W=11.0, N=512, yB_size=176, yN_size=256, xA_size=96, xM_size=128)


def _core():
    from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip

    return SwiftlyCoreHip(P["W"], P["N"], P["xM_size"], P["yN_size"])


def _chain(core, facet, fo, so, xA):
    bf = core.prepare_facet(facet, fo, axis=0)
    c = core.extract_from_facet(bf, so, axis=0)
    acc = core.add_to_subgrid(c, fo, axis=0)
    return core.finish_subgrid(acc, so, xA)


def test_pickle_roundtrip():
    core = _core()
    clone = pickle.loads(pickle.dumps(core))
    assert (clone.W, clone.N, clone.xM_size, clone.yN_size) == (core.W, core.N, core.xM_size, core.yN_size)
    assert clone._handle and clone._handle.value != core._handle.value  # a fresh native handle
    rng = numpy.random.default_rng(1)
    facet = rng.standard_normal(P["yB_size"]) + 1j * rng.standard_normal(P["yB_size"])
    a = _chain(core, facet, 8, 6, P["xA_size"])
    b = _chain(clone, facet, 8, 6, P["xA_size"])
    assert numpy.array_equal(a, b)
    # the state carries no device pointers
    assert set(core.__getstate__()) == {"W", "N", "xM_size", "yN_size", "column_precision", "axis1_first"}
    core.column_precision = 64  # the precision setting travels with the pickle
    assert pickle.loads(pickle.dumps(core)).column_precision == 64
    core.column_precision = 32


def test_eight_threads_share_one_core():
    core = _core()
    ref = orc.OracleCore(P["W"], P["N"], P["xM_size"], P["yN_size"])
    nthreads, reps = 8, 25
    rng = numpy.random.default_rng(2)
    facets = rng.standard_normal((nthreads, P["yB_size"])) + 1j * rng.standard_normal((nthreads, P["yB_size"]))
    offs = [(4 * (3 * i - 7), 2 * (5 * i - 11)) for i in range(nthreads)]
    want = [_chain(ref, facets[i], offs[i][0], offs[i][1], P["xA_size"]) for i in range(nthreads)]
    errors = []
    barrier = threading.Barrier(nthreads)

    def work(i):
        try:
            barrier.wait()
            for _ in range(reps):
                got = _chain(core, facets[i], offs[i][0], offs[i][1], P["xA_size"])
                err = numpy.abs(got - want[i]).max()
                if not err <= 1e-11 * numpy.abs(want[i]).max():
                    errors.append((i, float(err)))
                    return
        except Exception as exc:  # pylint: disable=broad-except
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_handles_created_concurrently():
    """swiftly_hip_create from several threads at once (per-device one-time setup is locked)."""
    made, errors = [], []

    def work():
        try:
            made.append(_core())
        except Exception as exc:  # pylint: disable=broad-except
            errors.append(repr(exc))

    threads = [threading.Thread(target=work) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(made) == 6
    x = numpy.ones(P["yB_size"], dtype=complex)
    outs = [c.prepare_facet(x, 0, axis=0) for c in made]
    assert all(numpy.array_equal(outs[0], o) for o in outs[1:])


def test_accumulating_batches_must_not_overlap():
    import torch

    core = _core()
    m, xM = core.xM_yN_size, core.xM_size
    src = torch.zeros((2, 4, m), dtype=torch.complex128, device="cuda")
    dst = torch.zeros((4, xM), dtype=torch.complex128, device="cuda")
    with pytest.raises(ValueError):
        core.launch("add_to_subgrid", src, 4, m, 1, dst, xM, 1, 0, nbatch=2, in_bs=4 * m, out_bs=0)
    dstf = torch.zeros((4, core.yN_size), dtype=torch.complex128, device="cuda")
    with pytest.raises(ValueError):
        core.launch("add_to_facet", src, 4, m, 1, dstf, core.yN_size, 1, 0, nbatch=2, in_bs=4 * m, out_bs=0)
    # distinct outputs are fine
    dst2 = torch.zeros((2, 4, xM), dtype=torch.complex128, device="cuda")
    core.launch("add_to_subgrid", src, 4, m, 1, dst2, xM, 1, 0, nbatch=2, in_bs=4 * m, out_bs=4 * xM)


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128])
def test_extract_column_rowmap_on_strided_bf(dtype):
    """Row-compacted BF_F stored TRANSPOSED (transform axis strided): goes through the generic two-pass
    route for long transforms; result must equal the contiguous-layout call and the oracle."""
    import torch

    W, N, xM, yN, yB = 11.0, 4096, 512, 2048, 1408  # yN >= 512 on a strided axis -> four-step
    from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip

    core = SwiftlyCoreHip(W, N, xM, yN)
    ref = orc.OracleCore(W, N, xM, yN)
    m = core.xM_yN_size
    off0s = [0, 3 * 448, -5 * 448]
    rowmap, n_rows = core.subgrid_column_rows(off0s)
    rng = numpy.random.default_rng(3)
    ncol = 40  # "facet size" along the other axis of this slab
    full = (rng.standard_normal((yN, ncol)) + 1j * rng.standard_normal((yN, ncol))).astype(dtype)
    rm = rowmap.cpu().numpy()
    compact = numpy.zeros((n_rows, ncol), dtype=dtype)
    compact[rm[rm >= 0]] = full[rm >= 0]
    bf_c = torch.from_numpy(compact).cuda()                      # [n_rows, ncol], transform axis (1) contiguous
    bf_t = torch.from_numpy(numpy.ascontiguousarray(compact.T)).cuda().T  # same values, axis 1 strided
    assert bf_t.stride() == (1, n_rows)
    for off0 in off0s:
        a = core.extract_column(bf_c, off0, 1408, rowmap=rowmap).cpu().numpy()
        b = core.extract_column(bf_t, off0, 1408, rowmap=rowmap).cpu().numpy()
        want = orc.extract_column(ref, full.astype(complex), off0, 1408)
        tol = 2e-6 if dtype == numpy.complex64 else 1e-12
        for got in (a, b):
            assert got.shape == (m, yN)
            rel = numpy.sqrt(numpy.mean(numpy.abs(got - want) ** 2) / numpy.mean(numpy.abs(want) ** 2))
            assert rel < tol, rel
    # a window that asks for rows the compacted buffer does not hold (map entry -1): those rows read as zeros
    # (no out-of-bounds access through a negative map entry), on both layouts and every row kernel
    other = 2 * 448
    needs = numpy.zeros(yN, dtype=bool)
    s0 = other * yN // N
    needs[(yN // 2 - m // 2 + numpy.arange(m) + s0) % yN] = True
    assert (rm[needs] < 0).any() and (rm[needs] >= 0).any()
    held = full.astype(complex).copy()
    held[rm < 0] = 0
    want = orc.extract_column(ref, held, other, 1408)
    for src in (bf_c, bf_t):
        got = core.extract_column(src, other, 1408, rowmap=rowmap).cpu().numpy()
        rel = numpy.sqrt(numpy.mean(numpy.abs(got - want) ** 2) / numpy.mean(numpy.abs(want) ** 2))
        assert rel < tol, rel


def test_integration_binding_host_staging():
    """The reference-side binding of INTEGRATION.md section B, verbatim in spirit: numpy arrays staged through
    swiftly_hip_malloc / memcpy, one ABI call per `ska_sdp_func`-style method (2-D arrays, last axis, `.T` views for
    axis 0, accumulate ops read `out`), including prepare_subgrid_inplace[_2d] on pre-padded arrays."""
    import ctypes

    from ska_sdp_exec_swiftly_amd import _lib, calculate_pswf

    lib = _lib.load()
    W, N, xM, yN = P["W"], P["N"], P["xM_size"], P["yN_size"]
    ref = orc.OracleCore(W, N, xM, yN)
    h = ctypes.c_void_p()
    pswf = numpy.ascontiguousarray(calculate_pswf(W, yN))
    _lib.check(lib.swiftly_hip_create(ctypes.byref(h), N, yN, xM, float(W),
                                      pswf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 0))

    def run(name, a, out, *tail):
        dt = 0 if a.dtype == numpy.complex64 else 1
        d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.swiftly_hip_malloc(ctypes.byref(d_in), a.size * a.itemsize))
        _lib.check(lib.swiftly_hip_malloc(ctypes.byref(d_out), out.size * out.itemsize))
        ac, oc = numpy.ascontiguousarray(a), numpy.ascontiguousarray(out)
        lib.swiftly_hip_memcpy_h2d(d_in, ac.ctypes.data, ac.nbytes, None)
        lib.swiftly_hip_memcpy_h2d(d_out, oc.ctypes.data, oc.nbytes, None)
        args = [h, dt, d_in, a.shape[0]]
        if name in ("prepare_facet", "prepare_subgrid"):
            args.append(a.shape[1])
        args += [a.shape[1], 1, d_out, out.shape[1], 1, *tail, None]
        _lib.check(getattr(lib, "swiftly_hip_" + name)(*args))
        lib.swiftly_hip_memcpy_d2h(oc.ctypes.data, d_out, oc.nbytes, None)
        _lib.check(lib.swiftly_hip_stream_synchronize(None))
        out[...] = oc
        lib.swiftly_hip_free(d_in)
        lib.swiftly_hip_free(d_out)

    rng = numpy.random.default_rng(8)
    yB, m, xA = P["yB_size"], ref.xM_yN_size, P["xA_size"]
    facet = rng.standard_normal((5, yB)) + 1j * rng.standard_normal((5, yB))
    out = numpy.empty((5, yN), dtype=complex)
    run("prepare_facet", facet, out, 8)
    want = ref.prepare_facet(facet, 8, axis=1)
    assert numpy.abs(out - want).max() <= 1e-12 * numpy.abs(want).max()
    # axis 0 as a .T view + accumulation into a pre-filled output
    contrib = rng.standard_normal((m, 7)) + 1j * rng.standard_normal((m, 7))
    acc = rng.standard_normal((xM, 7)) + 1j * rng.standard_normal((xM, 7))
    want = acc + ref.add_to_subgrid(contrib, -12, axis=0)
    run("add_to_subgrid", contrib.T, acc.T, -12)
    assert numpy.abs(acc - want).max() <= 1e-12 * numpy.abs(want).max()
    # prepare_subgrid_inplace_2d on an array that pad_mid has already padded (reference core.py:849-853)
    sg = rng.standard_normal((xA, xA)) + 1j * rng.standard_normal((xA, xA))
    padded = numpy.zeros((xM, xM), dtype=complex)
    lo = xM // 2 - xA // 2
    padded[lo : lo + xA, lo : lo + xA] = sg
    run("prepare_subgrid", padded.copy(), padded, 6)           # axis 1
    run("prepare_subgrid", padded.T.copy(), padded.T, -4)      # axis 0
    want = ref.prepare_subgrid(sg, [-4, 6])
    assert numpy.abs(padded - want).max() <= 1e-12 * numpy.abs(want).max()

    # the single-call 2-D entries of the shim (core.py:752-778, 837-855) through the raw ABI, complex128 and complex64
    def staged(arr):
        d = ctypes.c_void_p()
        c = numpy.array(arr, order="C", copy=True)  # (fetch() overwrites it with the result)
        _lib.check(lib.swiftly_hip_malloc(ctypes.byref(d), c.nbytes))
        lib.swiftly_hip_memcpy_h2d(d, c.ctypes.data, c.nbytes, None)
        return d, c

    def fetch(d, like):
        lib.swiftly_hip_memcpy_d2h(like.ctypes.data, d, like.nbytes, None)
        _lib.check(lib.swiftly_hip_stream_synchronize(None))
        lib.swiftly_hip_free(d)
        return like

    for cdt, code, tol in ((numpy.complex128, 1, 1e-12), (numpy.complex64, 0, 2e-6)):
        p2 = numpy.zeros((xM, xM), dtype=cdt)
        p2[lo : lo + xA, lo : lo + xA] = sg
        d, c = staged(p2)
        _lib.check(lib.swiftly_hip_prepare_subgrid_inplace_2d(h, code, d, xM, 1, -4, 6, None))
        got = fetch(d, c)
        assert numpy.abs(got - want).max() <= tol * numpy.abs(want).max()
        # one axis, in place, on the first 9 rows (last axis) ...
        d, c = staged(p2[lo : lo + 9])
        _lib.check(lib.swiftly_hip_prepare_subgrid_inplace(h, code, d, 9, xM, 1, 6, None))
        got = fetch(d, c)
        w1 = numpy.stack([ref.prepare_subgrid(sg[r].astype(complex), 6) for r in range(9)])
        assert numpy.abs(got - w1).max() <= tol * numpy.abs(w1).max()
        # add_to_subgrid_2d accumulates into a pre-filled [xM, xM]
        c2 = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))).astype(cdt)
        a2 = (rng.standard_normal((xM, xM)) + 1j * rng.standard_normal((xM, xM))).astype(cdt)
        w2 = a2.astype(complex) + ref.add_to_subgrid(ref.add_to_subgrid(c2.astype(complex), -12, axis=0), 20, axis=1)
        d_in, _ = staged(c2)
        d_out, co = staged(a2)
        _lib.check(lib.swiftly_hip_add_to_subgrid_2d(h, code, d_in, m, 1, d_out, xM, 1, -12, 20, None))
        got = fetch(d_out, co)
        lib.swiftly_hip_free(d_in)
        assert numpy.abs(got - w2).max() <= tol * numpy.abs(w2).max() * (1 if code else 10)
    lib.swiftly_hip_destroy(h)


def test_python_mirror_of_the_2d_shim_calls():
    """SwiftlyCoreHip.add_to_subgrid_2d (one native call) and prepare_subgrid_inplace against the oracle."""
    import torch

    from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip

    W, N, xM, yN, xA = P["W"], P["N"], P["xM_size"], P["yN_size"], P["xA_size"]
    core = SwiftlyCoreHip(W, N, xM, yN)
    ref = orc.OracleCore(W, N, xM, yN)
    m = ref.xM_yN_size
    rng = numpy.random.default_rng(9)
    c2 = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
    want = ref.add_to_subgrid(ref.add_to_subgrid(c2, 12, axis=0), -4, axis=1)
    got = core.add_to_subgrid_2d(c2, 12, -4)
    assert isinstance(got, numpy.ndarray) and numpy.abs(got - want).max() <= 1e-12 * numpy.abs(want).max()
    acc = torch.from_numpy(want.copy()).cuda()
    core.add_to_subgrid_2d(torch.from_numpy(c2).cuda(), 12, -4, out=acc)
    assert numpy.abs(acc.cpu().numpy() - 2 * want).max() <= 1e-12 * numpy.abs(want).max()
    with pytest.raises(ValueError):
        core.add_to_subgrid_2d(c2[:-1], 12, -4)
    sg = rng.standard_normal((xA, xA)) + 1j * rng.standard_normal((xA, xA))
    padded = numpy.zeros((xM, xM), dtype=complex)
    lo = xM // 2 - xA // 2
    padded[lo : lo + xA, lo : lo + xA] = sg
    t = torch.from_numpy(padded).cuda()
    assert core.prepare_subgrid_inplace(t, [4, -6]) is t
    want = ref.prepare_subgrid(sg, [4, -6])
    assert numpy.abs(t.cpu().numpy() - want).max() <= 1e-12 * numpy.abs(want).max()
    rows = torch.from_numpy(padded[lo : lo + 3].copy()).cuda()
    core.prepare_subgrid_inplace(rows, 8)
    w1 = numpy.stack([ref.prepare_subgrid(sg[r], 8) for r in range(3)])
    assert numpy.abs(rows.cpu().numpy() - w1).max() <= 1e-12 * numpy.abs(w1).max()
