#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by running the REFERENCE
implementation itself (ska_sdp_exec_swiftly 1.0.0, numpy backend, imported
unchanged from /root/reference/src).

Run in the authoring container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The un-vendored / absent dependencies the reference imports at module top
(ska_sdp_func at core.py:8; dask + distributed at api.py:18-21) are replaced
by empty stub modules -- none of them is touched by the numpy code path.

Fixtures (all complex128 unless noted):
  constants.npz     pswf / Fb / Fn of SwiftlyCore for three parameter sets
  prim1d.npz        every primitive, 1-D, reference TEST_PARAMS
                    (tests/test_core.py:20-27), even and odd yB / xA, offsets
                    that are negative and >= N
  prim2d.npz        every primitive along both axes of 2-D arrays + the
                    api_helper task bodies, small parameter set
  roundtrip2d.npz   full forward (9 facets -> 36 subgrids) and backward
                    (-> 9 facets) pass through the reference api_helper
                    functions in the order SwiftlyForward / SwiftlyBackward
                    drive them (api.py:238-463), small parameter set
"""
import os
import sys
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def import_reference():
    a = _stub("ska_sdp_func")
    b = _stub("ska_sdp_func.fourier_transforms")
    c = _stub("ska_sdp_func.fourier_transforms.swiftly")
    a.fourier_transforms = b
    b.swiftly = c

    class _NoClient:
        @staticmethod
        def current():
            raise RuntimeError("dask is not installed")

    d = _stub("dask", delayed=lambda *x, **k: None)
    d.array = _stub("dask.array")
    d.distributed = _stub("dask.distributed", Client=_NoClient)
    _stub("distributed", Client=_NoClient)
    sys.path.insert(0, REF_SRC)
    from ska_sdp_exec_swiftly import api, api_helper
    from ska_sdp_exec_swiftly.fourier_transform import fourier_algorithm as fa
    from ska_sdp_exec_swiftly.fourier_transform.core import SwiftlyCore

    return SwiftlyCore, fa, api_helper, api


TEST_PARAMS = dict(W=13.5625, N=1024, yB_size=416, yN_size=512, xA_size=228, xM_size=256)
SMALL_PARAMS = dict(W=13.5625, N=512, yB_size=208, yN_size=256, xA_size=100, xM_size=128)
SMALL11_PARAMS = dict(W=11.0, N=512, yB_size=176, yN_size=256, xA_size=96, xM_size=128)
BENCH8K_PARAMS = dict(W=11.0, N=8192, yB_size=1408, yN_size=2048, xA_size=1024, xM_size=2048)


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def main():
    SwiftlyCore, fa, helper, api = import_reference()
    rng = numpy.random.default_rng(20240807)

    # ---------------------------------------------------------------- consts
    consts = {}
    for name, p in [("test", TEST_PARAMS), ("small", SMALL_PARAMS), ("small11", SMALL11_PARAMS), ("bench8k", BENCH8K_PARAMS)]:
        core = SwiftlyCore(p["W"], p["N"], p["xM_size"], p["yN_size"])
        pswf = core._calculate_pswf()
        consts[f"{name}_pswf"] = pswf
        consts[f"{name}_Fb"] = core._Fb
        consts[f"{name}_Fn"] = core._Fn
    numpy.savez_compressed(os.path.join(HERE, "constants.npz"), **consts)

    # ---------------------------------------------------------------- 1-D
    p = TEST_PARAMS
    N = p["N"]
    core = SwiftlyCore(p["W"], N, p["xM_size"], p["yN_size"])
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    m = core.xM_yN_size
    d1 = {}
    facet_offs = [0, Ny, -3 * Ny, 7 * Ny, N + 2 * Ny, -N - Ny]
    sg_offs = [0, Nx, -5 * Nx, 11 * Nx, N, -N + 3 * Nx]
    d1["facet_offs"] = numpy.array(facet_offs)
    d1["sg_offs"] = numpy.array(sg_offs)
    for yB in (p["yB_size"], p["yB_size"] - 1):
        facet = crand(rng, yB)
        d1[f"facet_{yB}"] = facet
        for i, fo in enumerate(facet_offs):
            d1[f"prepare_facet_{yB}_{i}"] = core.prepare_facet(facet, fo, axis=0)
        acc = crand(rng, p["yN_size"])
        d1[f"facc_{yB}"] = acc
        for i, fo in enumerate(facet_offs):
            d1[f"finish_facet_{yB}_{i}"] = core.finish_facet(acc, fo, yB, axis=0)
    prep = crand(rng, p["yN_size"])
    d1["prep"] = prep
    contrib = crand(rng, m)
    d1["contrib"] = contrib
    for i, so in enumerate(sg_offs):
        d1[f"extract_from_facet_{i}"] = core.extract_from_facet(prep, so, axis=0)
        d1[f"add_to_facet_{i}"] = core.add_to_facet(contrib, so, axis=0)
    for i, fo in enumerate(facet_offs):
        d1[f"add_to_subgrid_{i}"] = core.add_to_subgrid(contrib, fo, axis=0)
    sacc = crand(rng, p["xM_size"])
    d1["sacc"] = sacc
    for i, fo in enumerate(facet_offs):
        d1[f"extract_from_subgrid_{i}"] = core.extract_from_subgrid(sacc, fo, axis=0)
    for xA in (p["xA_size"], p["xA_size"] - 1):
        sg = crand(rng, xA)
        d1[f"subgrid_{xA}"] = sg
        for i, so in enumerate(sg_offs):
            d1[f"finish_subgrid_{xA}_{i}"] = core.finish_subgrid(sacc, so, xA)
            d1[f"prepare_subgrid_{xA}_{i}"] = core.prepare_subgrid(sg, so)
    numpy.savez_compressed(os.path.join(HERE, "prim1d.npz"), **d1)

    # ---------------------------------------------------------------- 2-D
    p = SMALL_PARAMS
    N = p["N"]
    core = SwiftlyCore(p["W"], N, p["xM_size"], p["yN_size"])
    Nx, Ny = core.subgrid_off_step, core.facet_off_step
    m, yN, xM, yB, xA = core.xM_yN_size, p["yN_size"], p["xM_size"], p["yB_size"], p["xA_size"]
    d2 = {}
    fo0, fo1 = 3 * Ny, -N // 2 + Ny
    so0, so1 = -7 * Nx, N + 5 * Nx
    d2["offs"] = numpy.array([fo0, fo1, so0, so1])
    facet = crand(rng, yB, 61)  # non-square, odd and small second axis
    d2["facet"] = facet
    BF = core.prepare_facet(facet, fo0, axis=0)
    d2["prepare_facet_a0"] = BF
    d2["prepare_facet_a1"] = core.prepare_facet(facet[:37], fo1, axis=1)  # first 37 rows
    d2["extract_from_facet_a0"] = core.extract_from_facet(BF, so0, axis=0)
    col = helper.extract_column(core, BF, so0, fo1)
    d2["extract_column"] = col
    contrib = core.extract_from_facet(col, so1, axis=1)
    d2["contrib"] = contrib
    a0 = core.add_to_subgrid(contrib, fo0, axis=0)
    d2["add_to_subgrid_a0"] = a0
    a01 = core.add_to_subgrid(a0, fo1, axis=1)
    d2["add_to_subgrid_a01"] = a01
    d2["finish_subgrid"] = core.finish_subgrid(a01, [so0, so1], xA)
    d2["finish_subgrid_odd"] = core.finish_subgrid(a01, [so0, so1], xA - 1)
    sg = crand(rng, xA, 33)
    d2["subgrid"] = sg
    ps = core.prepare_subgrid(sg, [so0, so1])
    d2["prepare_subgrid"] = ps
    e0 = core.extract_from_subgrid(ps, fo0, axis=0)
    d2["extract_from_subgrid_a0"] = e0
    e01 = core.extract_from_subgrid(e0, fo1, axis=1)
    d2["extract_from_subgrid_a01"] = e01
    f1 = core.add_to_facet(e01, so1, axis=1)
    d2["add_to_facet_a1"] = f1
    ff1 = core.finish_facet(f1, fo1, 61, axis=1)
    d2["finish_facet_a1"] = ff1
    f0 = core.add_to_facet(ff1, so0, axis=0)
    d2["add_to_facet_a0"] = f0
    d2["finish_facet_a0"] = core.finish_facet(f0, fo0, yB, axis=0)
    numpy.savez_compressed(os.path.join(HERE, "prim2d.npz"), **d2)

    # ---------------------------------------------------------------- round trip
    class Cfg:  # duck-types SwiftlyConfig for make_full_*_cover
        image_size = N
        max_facet_size = yB
        max_subgrid_size = xA

    facet_cfgs = api.make_full_facet_cover(Cfg)
    sg_cfgs = api.make_full_subgrid_cover(Cfg)
    rt = {}
    rt["facet_offs"] = numpy.array([[f.off0, f.off1] for f in facet_cfgs])
    rt["sg_offs"] = numpy.array([[s.off0, s.off1] for s in sg_cfgs])
    rt["facet_mask0"] = numpy.array([f.mask0 for f in facet_cfgs])
    rt["facet_mask1"] = numpy.array([f.mask1 for f in facet_cfgs])
    rt["sg_mask0"] = numpy.array([s.mask0 for s in sg_cfgs])
    rt["sg_mask1"] = numpy.array([s.mask1 for s in sg_cfgs])
    facets = []
    for j, f in enumerate(facet_cfgs):
        r = numpy.random.default_rng(1234 + j)
        g = (r.standard_normal((yB, yB)) + 1j * r.standard_normal((yB, yB))).astype(
            numpy.complex64
        ).astype(complex)
        g *= f.mask0[:, None]
        g *= f.mask1[None, :]
        facets.append(g)
    # Forward, exactly the call order of api.py:238-324
    BF_Fs = [core.prepare_facet(d, f.off0, axis=0) for f, d in zip(facet_cfgs, facets)]
    subgrids = []
    cache = {}
    for s in sg_cfgs:
        if s.off0 not in cache:
            cache = {
                s.off0: [
                    helper.extract_column(core, BF, s.off0, f.off1)
                    for f, BF in zip(facet_cfgs, BF_Fs)
                ]
            }
        contribs = [core.extract_from_facet(c, s.off1, axis=1) for c in cache[s.off0]]
        subgrids.append(helper.sum_and_finish_subgrid(core, contribs, facet_cfgs, s))
    sgs = numpy.array(subgrids)
    # keep fixtures small: strided samples + whole-array moments
    rt["subgrids_sample"] = sgs[:, ::7, ::5]
    rt["subgrids_sum"] = sgs.sum(axis=(1, 2))
    rt["subgrids_pow"] = (numpy.abs(sgs) ** 2).sum(axis=(1, 2))
    rt["subgrid_full_idx"] = numpy.array([0, 17, 35])
    rt["subgrids_full"] = sgs[[0, 17, 35]]
    # Backward, api.py:347-463 with an unbounded column cache
    F = len(facet_cfgs)
    MN = [None] * F
    cols, order = {}, []
    for s, data in zip(sg_cfgs, subgrids):
        parts = helper.prepare_and_split_subgrid(core, data, [s.off0, s.off1], facet_cfgs)
        if s.off0 not in cols:
            cols[s.off0] = [None] * F
            order.append(s.off0)
        cols[s.off0] = [
            helper.accumulate_column(core, pp, old, s.off1)
            for pp, old in zip(parts, cols[s.off0])
        ]
    for off0 in order:
        MN = [
            helper.accumulate_facet(core, c, acc, f, off0)
            for f, c, acc in zip(facet_cfgs, cols[off0], MN)
        ]
    out_facets = [helper.finish_facet(core, acc, f) for f, acc in zip(facet_cfgs, MN)]
    fo = numpy.array(out_facets)
    rt["facets_out_sample"] = fo[:, ::9, ::7]
    rt["facets_out_sum"] = fo.sum(axis=(1, 2))
    rt["facets_out_pow"] = (numpy.abs(fo) ** 2).sum(axis=(1, 2))
    rt["facet_full_idx"] = numpy.array([4])
    rt["facets_out_full"] = fo[[4]]
    err = max(
        numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2)) for a, b in zip(out_facets, facets)
    )
    print("reference round-trip RMSE (dense random facets, small params):", err)
    rt["roundtrip_rmse"] = numpy.array(err)
    # store facets as seeds only (regenerated in the test); outputs c128
    numpy.savez_compressed(os.path.join(HERE, "roundtrip2d.npz"), **rt)
    make_roundtrip11(SwiftlyCore, helper, api)
    for f in ("constants.npz", "prim1d.npz", "prim2d.npz", "roundtrip2d.npz", "roundtrip2d_w11.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


def make_roundtrip11(SwiftlyCore, helper, api):
    """Forward + backward through the reference for the W=11 small parameter set
    (max 1/pswf ~ 90, the regime the complex64 path is specified for)."""
    p = SMALL11_PARAMS
    N, yB, xA = p["N"], p["yB_size"], p["xA_size"]
    core = SwiftlyCore(p["W"], N, p["xM_size"], p["yN_size"])

    class Cfg:
        image_size = N
        max_facet_size = yB
        max_subgrid_size = xA

    facet_cfgs = api.make_full_facet_cover(Cfg)
    sg_cfgs = api.make_full_subgrid_cover(Cfg)
    facets = []
    for j, f in enumerate(facet_cfgs):
        r = numpy.random.default_rng(4321 + j)
        g = (r.standard_normal((yB, yB)) + 1j * r.standard_normal((yB, yB))).astype(numpy.complex64).astype(complex)
        facets.append(g * f.mask0[:, None] * f.mask1[None, :])
    BF_Fs = [core.prepare_facet(d, f.off0, axis=0) for f, d in zip(facet_cfgs, facets)]
    subgrids = []
    cache = {}
    for s in sg_cfgs:
        if s.off0 not in cache:
            cache = {s.off0: [helper.extract_column(core, BF, s.off0, f.off1) for f, BF in zip(facet_cfgs, BF_Fs)]}
        contribs = [core.extract_from_facet(c, s.off1, axis=1) for c in cache[s.off0]]
        subgrids.append(helper.sum_and_finish_subgrid(core, contribs, facet_cfgs, s))
    F = len(facet_cfgs)
    MN, cols, order = [None] * F, {}, []
    for s, data in zip(sg_cfgs, subgrids):
        parts = helper.prepare_and_split_subgrid(core, data, [s.off0, s.off1], facet_cfgs)
        if s.off0 not in cols:
            cols[s.off0] = [None] * F
            order.append(s.off0)
        cols[s.off0] = [helper.accumulate_column(core, pp, old, s.off1) for pp, old in zip(parts, cols[s.off0])]
    for off0 in order:
        MN = [helper.accumulate_facet(core, c, acc, f, off0) for f, c, acc in zip(facet_cfgs, cols[off0], MN)]
    out_facets = numpy.array([helper.finish_facet(core, acc, f) for f, acc in zip(facet_cfgs, MN)])
    sgs = numpy.array(subgrids)
    err = max(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2)) for a, b in zip(out_facets, facets))
    print("reference round-trip RMSE (W=11 small params):", err)
    numpy.savez_compressed(
        os.path.join(HERE, "roundtrip2d_w11.npz"),
        n_facets=numpy.array(F),
        n_subgrids=numpy.array(len(sg_cfgs)),
        subgrids_sample=sgs[:, ::5, ::3],
        subgrid_full_idx=numpy.array([0, 7, 20]),
        subgrids_full=sgs[[0, 7, 20]],
        facets_out_sample=out_facets[:, ::9, ::7],
        facet_full_idx=numpy.array([4]),
        facets_out_full=out_facets[[4]],
        roundtrip_rmse=numpy.array(err),
    )


if __name__ == "__main__":
    main()
