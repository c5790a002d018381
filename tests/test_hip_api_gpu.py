"""
GPU tests of the streaming classes (SwiftlyConfig / SwiftlyForward /
SwiftlyBackward with backend="hip") against
  * the reference-generated round-trip golden (tests/golden/roundtrip2d.npz),
  * the oracle's serial replica of the reference dataflow,
  * the reference's own end-to-end check (tests/test_api.py:56-125: one point
    source, per-facet RMSE < 3e-10 after facet -> subgrid -> facet), for the
    same (lru_forward, lru_backward, shuffle) combinations.
"""
import os
import random

import numpy
import pytest

from oracle import swiftly_oracle as orc

pytestmark = pytest.mark.gpu

TEST_PARAMS = dict(W=13.5625, fov=1.0, N=1024, yB_size=416, yN_size=512, xA_size=228, xM_size=256)
SMALL_PARAMS = dict(W=13.5625, fov=1.0, N=512, yB_size=208, yN_size=256, xA_size=100, xM_size=128)


SMALL11_PARAMS = dict(W=11.0, fov=1.0, N=512, yB_size=176, yN_size=256, xA_size=96, xM_size=128)


def small_problem(params, dtype, seed):
    import ska_sdp_exec_swiftly_amd as sw

    cfg = sw.SwiftlyConfig(backend="hip", **params)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    sg_cfgs = sw.make_full_subgrid_cover(cfg)
    yB = params["yB_size"]
    facets = []
    for j, f in enumerate(facet_cfgs):
        r = numpy.random.default_rng(seed + j)
        d = (r.standard_normal((yB, yB)) + 1j * r.standard_normal((yB, yB))).astype(numpy.complex64)
        facets.append((d * f.mask0[:, None] * f.mask1[None, :]).astype(dtype))
    return sw, cfg, facet_cfgs, sg_cfgs, facets


def relrms(a, b):
    return float(numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2)))


# (golden file, parameters, facet seed, dtype, forward relRMSE bound, backward relRMSE bound)
#  complex128: rounding only; the 1/pswf window (max ~4.9e3 at W=13.56) sets the level.
#  complex64 is specified for the W~11 family (max 1/pswf ~ 90): float32 arithmetic in the
#  length-yN transforms acts on window-amplified data, which puts the error at ~1e-5 relative
#  (tests/accuracy_model.py: float32 ARITHMETIC in K2 and K3 sets it; the float32 STORAGE floor is 3.3e-6).
CASES = [
    ("roundtrip2d.npz", SMALL_PARAMS, 1234, numpy.complex128, 1e-10, 1e-9),
    ("roundtrip2d_w11.npz", SMALL11_PARAMS, 4321, numpy.complex128, 1e-11, 1e-10),
    ("roundtrip2d_w11.npz", SMALL11_PARAMS, 4321, numpy.complex64, 2e-5, 4e-5),
]


@pytest.mark.parametrize("gfile,params,seed,dtype,ftol,btol", CASES)
def test_forward_backward_golden(golden_dir, gfile, params, seed, dtype, ftol, btol):
    g = numpy.load(os.path.join(golden_dir, gfile))
    sw, cfg, facet_cfgs, sg_cfgs, facets = small_problem(params, dtype, seed)
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), lru_forward=2)
    sgs = fwd.get_subgrid_tasks(sg_cfgs)  # one wave per subgrid column
    got = numpy.array([s.cpu().numpy() for s in sgs])
    assert got.dtype == dtype
    step = (7, 5) if gfile == "roundtrip2d.npz" else (5, 3)
    e1 = relrms(got[:, :: step[0], :: step[1]], g["subgrids_sample"])
    e2 = relrms(got[g["subgrid_full_idx"]], g["subgrids_full"])
    print(f"forward relRMSE {dtype.__name__}: sample {e1:.3e} full {e2:.3e}")
    assert e1 < ftol and e2 < ftol, (e1, e2)
    # one at a time gives the same answer as waves
    one = fwd.get_subgrid_task(sg_cfgs[7]).cpu().numpy()
    assert numpy.array_equal(one, got[7])
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=2)
    for sg_cfg, data in zip(sg_cfgs, sgs):
        bwd.add_new_subgrid_task(sg_cfg, data)
    out = numpy.array([f.cpu().numpy() for f in bwd.finish()])
    # wave-batched form (all subgrids at once, grouped by off0 internally) gives the same facets
    bwd2 = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=1, queue_size=2)
    bwd2.add_new_subgrid_tasks(sg_cfgs, sgs)
    out2 = numpy.array([f.cpu().numpy() for f in bwd2.finish()])
    assert relrms(out2, out) < (1e-12 if dtype == numpy.complex128 else 3e-6)
    e3 = relrms(out[:, ::9, ::7], g["facets_out_sample"])
    e4 = relrms(out[g["facet_full_idx"]], g["facets_out_full"])
    print(f"backward relRMSE {dtype.__name__}: sample {e3:.3e} full {e4:.3e}")
    assert e3 < btol and e4 < btol, (e3, e4)
    if dtype == numpy.complex64:
        # band schedule (wave_axis=1: strided-axis finish per off1 wave, contiguous-axis transform at the end):
        # subgrid by subgrid in the reference's order, wave-batched with a plan, and shuffled
        order = list(range(len(sg_cfgs)))
        random.Random(5).shuffle(order)
        runs = []
        for mode in ("single", "waves+plan", "shuffled"):
            plan = sg_cfgs if mode == "waves+plan" else None
            b1 = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=plan)
            if mode == "single":
                for sg_cfg, data in zip(sg_cfgs, sgs):
                    b1.add_new_subgrid_task(sg_cfg, data)
            elif mode == "waves+plan":
                by1 = sorted(range(len(sg_cfgs)), key=lambda i: (sg_cfgs[i].off1, sg_cfgs[i].off0))
                b1.add_new_subgrid_tasks([sg_cfgs[i] for i in by1], [sgs[i] for i in by1])
            else:
                for i in order:
                    b1.add_new_subgrid_task(sg_cfgs[i], sgs[i])
            o1 = numpy.array([f.cpu().numpy() for f in b1.finish()])
            e5 = relrms(o1[:, ::9, ::7], g["facets_out_sample"])
            e6 = relrms(o1[g["facet_full_idx"]], g["facets_out_full"])
            print(f"backward (band, {mode}) relRMSE: sample {e5:.3e} full {e6:.3e}")
            assert e5 < btol and e6 < btol, (mode, e5, e6)
            runs.append(o1)
        # different summation orders of float32 values that are each ~1.4e-5 from the truth
        assert relrms(runs[1], runs[0]) < 1.5e-5 and relrms(runs[2], runs[0]) < 1.5e-5


def test_backward_band_sparse_plan_and_overlaps():
    """wave_axis=1 backward with a sparse subgrid set (band shorter than the padded axis, wrapped offsets) and
    with more than two subgrids overlapping in a row (duplicates: the row tables split into groups): equal to the
    reference schedule (wave_axis=0) on the same inputs."""
    import torch

    params = dict(W=11.0, fov=1.0, N=1024, yB_size=352, yN_size=512, xA_size=192, xM_size=256)
    sw, cfg, facet_cfgs, sg_cfgs, _ = small_problem(params, numpy.complex64, 3)
    keep = [c for c in sg_cfgs if c.off1 in (0, 192, 960)][:14]
    keep = keep + keep[:3] + keep[:2]  # duplicates: up to three sources per padded row
    assert len({c.off1 for c in keep}) < len({c.off1 for c in sg_cfgs})
    gen = torch.Generator(device="cpu").manual_seed(9)
    data = [torch.randn((192, 192), dtype=torch.complex64, generator=gen).cuda() for _ in keep]
    b0 = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=0)
    b1 = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=keep)
    by1 = sorted(range(len(keep)), key=lambda i: keep[i].off1)
    b0.add_new_subgrid_tasks(keep, data)
    b1.add_new_subgrid_tasks([keep[i] for i in by1], [data[i] for i in by1])
    assert b1._band[1] < cfg.core.yN_size  # pylint: disable=protected-access
    # truth: the reference schedule in complex128; both complex64 schedules must sit at the float32 level
    b64 = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=0)
    b64.add_new_subgrid_tasks(keep, [d.to(torch.complex128) for d in data])
    truth = torch.stack(b64.finish())
    got1 = torch.stack(b1.finish()).to(torch.complex128)
    got0 = torch.stack(b0.finish()).to(torch.complex128)
    norm = float(truth.abs().pow(2).mean().sqrt())
    e1 = float((got1 - truth).abs().pow(2).mean().sqrt()) / norm
    e0 = float((got0 - truth).abs().pow(2).mean().sqrt()) / norm
    print(f"relRMSE vs complex128: band schedule {e1:.3e}, reference schedule {e0:.3e}")
    assert e1 < 2e-5 and e1 < 1.5 * e0 + 2e-6, (e1, e0)
    with pytest.raises(ValueError):
        b2 = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=keep[:1])
        other = next(c for c in sg_cfgs if c.off1 != keep[0].off1)
        b2.add_new_subgrid_task(other, data[0])
    with pytest.raises(ValueError):
        b3 = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1)
        b3.add_new_subgrid_task(keep[0], data[0].to(torch.complex128))


@pytest.mark.parametrize(
    "lru_forward,lru_backward,shuffle",
    [(1, 1, False), (2, 1, False), (1, 2, False), (1, 1, True), (2, 1, True), (1, 2, True)],
)
def test_swiftly_api_roundtrip(lru_forward, lru_backward, shuffle):
    """reference tests/test_api.py:42-125 on the HIP backend (complex128)."""
    import ska_sdp_exec_swiftly_amd as sw

    sources = [(1, 1, 0)]
    cfg = sw.SwiftlyConfig(backend="hip", **TEST_PARAMS)
    sg_cfgs = sw.make_full_subgrid_cover(cfg)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    facet_tasks = [(fc, sw.make_facet(cfg.image_size, fc, sources)) for fc in facet_cfgs]
    fwd = sw.SwiftlyForward(cfg, facet_tasks, lru_forward, 100)
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward, 100)
    if shuffle:
        random.Random(42).shuffle(sg_cfgs)
    for sg_cfg in sg_cfgs:
        sg = fwd.get_subgrid_task(sg_cfg)
        assert sw.check_subgrid(cfg.image_size, sg_cfg, sg, sources) < 1e-9
        bwd.add_new_subgrid_task(sg_cfg, sg)
    for fc, facet in zip(facet_cfgs, bwd.finish()):
        assert sw.check_facet(cfg.image_size, fc, facet, sources) < 3e-10


def test_forward_matches_oracle_c64_bench_shape():
    """A slice of the N=8192 benchmark shape (BASELINE config 2 parameters:
    W=11, yB=1408, yN=2048, xA=1024, xM=2048, m=512): 2 facets x one subgrid
    column of 3, complex64, against the oracle.  Tolerance: relative RMSE
    2e-5 (float32 arithmetic on data amplified by 1/pswf <= 90, see CASES above)."""
    import ska_sdp_exec_swiftly_amd as sw

    P = dict(W=11.0, fov=1.0, N=8192, yB_size=1408, yN_size=2048, xA_size=1024, xM_size=2048)
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    facet_cfgs = [c for c in sw.make_full_facet_cover(cfg) if c.off0 == 1408 and c.off1 in (0, 2816)]
    sg_cfgs = [c for c in sw.make_full_subgrid_cover(cfg) if c.off0 == 2048][:3]
    facets = []
    for j, f in enumerate(facet_cfgs):
        r = numpy.random.default_rng(77 + j)
        d = (r.standard_normal((1408, 1408)) + 1j * r.standard_normal((1408, 1408))).astype(numpy.complex64)
        facets.append(d * f.mask0[:, None].astype(numpy.float32) * f.mask1[None, :].astype(numpy.float32))
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)))
    got = [s.cpu().numpy() for s in fwd.get_subgrid_tasks(sg_cfgs)]
    ref = orc.OracleCore(P["W"], P["N"], P["xM_size"], P["yN_size"])
    items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    sitems = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in sg_cfgs]
    want = orc.forward_all(ref, items, [f.astype(complex) for f in facets], sitems)
    for a, b in zip(got, want):
        assert a.dtype == numpy.complex64
        rel = numpy.sqrt(numpy.mean(numpy.abs(a - b) ** 2) / numpy.mean(numpy.abs(b) ** 2))
        assert rel < 2e-5, rel  # measured 1.1e-5; numpy float32 path: 6.9e-6


def test_sparse_plan_is_bit_identical():
    """Row-compacted BF_F (subgrid_configs= plan) gives exactly the same subgrids as the full BF_F, and
    asking for an unplanned column is refused."""
    sw, cfg, facet_cfgs, sg_cfgs, facets = small_problem(SMALL11_PARAMS, numpy.complex64, 4321)
    wanted = [c for c in sg_cfgs if c.off0 in (0, 96 * 2, 96 * 5)]
    full = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)))
    plan = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=wanted)
    a = [t.cpu().numpy() for t in full.get_subgrid_tasks(wanted)]
    b = [t.cpu().numpy() for t in plan.get_subgrid_tasks(wanted)]
    assert plan._n_rows < cfg.internal_facet_size  # really compacted
    assert plan.BF_Fs_persist[0].shape[0] == plan._n_rows
    for x, y in zip(a, b):
        assert numpy.array_equal(x, y)
    # ... and the planned path agrees with the ORACLE (not only with the other HIP path)
    ref = orc.OracleCore(SMALL11_PARAMS["W"], SMALL11_PARAMS["N"], SMALL11_PARAMS["xM_size"], SMALL11_PARAMS["yN_size"])
    items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    sitems = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in wanted]
    want = orc.forward_all(ref, items, [f.astype(complex) for f in facets], sitems)
    for y, w in zip(b, want):
        assert relrms(y, w) < 2e-5
    with pytest.raises(ValueError):
        plan.get_subgrid_task([c for c in sg_cfgs if c.off0 == 96][0])


def test_forward_c64_test_params_fused_paths():
    """complex64 forward at the reference TEST_PARAMS sizes (m=128, xM=256): exercises the fused
    sum+finish row kernel instance (7, 8) and the batched K4a path against the oracle.  W=13.56 has
    max 1/pswf ~ 4.9e3, so float32 only reaches ~1e-2 here (measured r2 with a float32 numpy chain: 8e-3);
    the check is against that level, the tight complex64 checks use W=11."""
    import ska_sdp_exec_swiftly_amd as sw

    cfg = sw.SwiftlyConfig(backend="hip", **TEST_PARAMS)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    sg_cfgs = [c for c in sw.make_full_subgrid_cover(cfg) if c.off0 == 228 * 2]
    rng = numpy.random.default_rng(5)
    facets = []
    for f in facet_cfgs:
        d = (rng.standard_normal((416, 416)) + 1j * rng.standard_normal((416, 416))).astype(numpy.complex64)
        facets.append((d * f.mask0[:, None] * f.mask1[None, :]).astype(numpy.complex64))
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)))
    got = [s.cpu().numpy() for s in fwd.get_subgrid_tasks(sg_cfgs)]
    ref = orc.OracleCore(TEST_PARAMS["W"], TEST_PARAMS["N"], TEST_PARAMS["xM_size"], TEST_PARAMS["yN_size"])
    items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    sitems = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in sg_cfgs]
    want = orc.forward_all(ref, items, [f.astype(complex) for f in facets], sitems)
    for a, b in zip(got, want):
        assert relrms(a, b) < 3e-2
    # and the same data in complex128 (unfused fallback) agrees to rounding
    fwd2 = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, [f.astype(complex) for f in facets])))
    got2 = [s.cpu().numpy() for s in fwd2.get_subgrid_tasks(sg_cfgs)]
    for a, b in zip(got2, want):
        assert relrms(a, b) < 1e-10


@pytest.mark.parametrize("params,dtype", [(TEST_PARAMS, numpy.complex64), (SMALL11_PARAMS, numpy.complex128)])
def test_distributed_classes_world1(params, dtype):
    """DistributedForward / DistributedBackward with a single rank (no process group): same results as the
    single-process classes, through the exchange layouts (send buffer written in place, arrival-order consumption).
    TEST_PARAMS/complex64 takes the fused route (transformed blocks), SMALL11/complex128 the raw-contribution route."""
    import torch

    from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward, DistributedForward

    sw, cfg, facet_cfgs, sg_cfgs, facets = small_problem(params, dtype, 777)
    tdt = torch.complex64 if dtype == numpy.complex64 else torch.complex128
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), wave_axis=0)
    dfw = DistributedForward(cfg, facet_cfgs, [torch.from_numpy(f).cuda() for f in facets], dtype=tdt, wave_axis=0)
    assert dfw.fused == (dtype == numpy.complex64)
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=2)
    dbw = DistributedBackward(cfg, facet_cfgs, lru_backward=2, dtype=tdt)
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off0, []).append(c)
    pending = None
    results = []
    for wave in waves.values():  # pipelined like bench.py: start w+1 before finishing w
        handle = dfw.start_wave(wave)
        if pending is not None:
            results.append(dfw.finish_wave(pending))
        pending = handle
    results.append(dfw.finish_wave(pending))
    tol = 3e-6 if dtype == numpy.complex64 else 1e-12
    for wave, (mine, res) in zip(waves.values(), results):
        assert mine == list(range(len(wave)))
        want = fwd.get_wave(wave)
        scale = float(want.abs().max())
        assert float((res - want).abs().max()) <= tol * scale
        bwd.add_new_subgrid_tasks(wave, [want[k] for k in range(len(wave))])
        dbw.add_wave(wave, [want[k] for k in range(len(wave))])
    idx, got = dbw.finish()
    ref = bwd.finish()
    assert idx == list(range(len(facet_cfgs)))
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())


def _virtual_all_to_all(sends, in_counts):
    """In-process stand-in for all_to_all_single: rank r receives, from every source s in order, the chunk s sent
    to r (``sends[s]`` flat buffers, ``in_counts[s][r]`` elements)."""
    import torch

    world = len(sends)
    starts = [numpy.concatenate([[0], numpy.cumsum(c)]) for c in in_counts]
    return [
        torch.cat([sends[s][int(starts[s][r]) : int(starts[s][r + 1])] for s in range(world)]) for r in range(world)
    ]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_distributed_virtual_ranks(world):
    """The multi-rank layouts with REAL HIP kernels on one GPU: `world` virtual ranks in one process (facet sharding,
    send buffers written in place per destination, arrival-order consumption, weighted subgrid ownership, the mirror
    exchange of the backward pass), the all-to-all replaced by an in-process shuffle of the flat buffers.  Results
    must equal the single-process classes."""
    import torch

    from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward, DistributedForward

    # W = 11 family (well conditioned in float32) with sizes that have the fused subgrid kernels (m = 128, xM = 256)
    params = dict(W=11.0, fov=1.0, N=1024, yB_size=352, yN_size=512, xA_size=192, xM_size=256)
    sw, cfg, facet_cfgs, sg_cfgs, facets = small_problem(params, numpy.complex64, 31)
    dev_facets = [torch.from_numpy(f).cuda() for f in facets]
    ref_fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, dev_facets)), wave_axis=0)
    assert ref_fwd.supports_fused_subgrid_side()
    ref_bwd = sw.SwiftlyBackward(cfg, facet_cfgs)
    fwds = [
        DistributedForward(cfg, facet_cfgs, dev_facets, dtype=torch.complex64, wave_axis=0, rank_world=(r, world))
        for r in range(world)
    ]
    bwds = [DistributedBackward(cfg, facet_cfgs, rank_world=(r, world)) for r in range(world)]
    assert sorted(j for f in fwds for j in f.sharding.local_facets) == list(range(len(facet_cfgs)))
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off0, []).append(c)
    for wave in waves.values():
        want = ref_fwd.get_wave(wave)
        packed = [f.pack_wave(wave) for f in fwds]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        got = {}
        for r, f in enumerate(fwds):
            assert recvs[r].numel() == sum(packed[r][2])  # out_counts
            mine, res = f.unpack_wave(wave, recvs[r])
            for k, i in enumerate(mine):
                got[i] = res[k]
        assert sorted(got) == list(range(len(wave)))
        scale = float(want.abs().max())
        for i in range(len(wave)):
            # same kernels, different facet summation order (arrival order): float32 rounding only
            assert float((got[i] - want[i]).abs().max()) <= 2e-5 * scale
        # backward: every virtual rank sends the contributions of the subgrids it holds
        ref_bwd.add_new_subgrid_tasks(wave, [want[i] for i in range(len(wave))])
        packed = [b.pack_wave(wave, [want[i] for i in b.sharding.subgrids_of(len(wave))]) for b in bwds]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        for r, b in enumerate(bwds):
            assert recvs[r].numel() == sum(packed[r][2])
            b.unpack_wave(wave, recvs[r])
    ref = ref_bwd.finish()
    for b in bwds:
        idx, out = b.finish()
        for j, o in zip(idx, out):
            assert float((o - ref[j]).abs().max()) <= 2e-5 * float(ref[j].abs().max())
    # the band schedule of the backward pass (waves keyed by off1) through the same exchange layouts
    bwds1 = [DistributedBackward(cfg, facet_cfgs, rank_world=(r, world), wave_axis=1, subgrid_configs=sg_cfgs)
             for r in range(world)]
    full = {(c.off0, c.off1): ref_fwd.get_subgrid_task(c) for c in sg_cfgs}
    waves1 = {}
    for c in sg_cfgs:
        waves1.setdefault(c.off1, []).append(c)
    for wave in waves1.values():
        packed = [b.pack_wave(wave, [full[(wave[i].off0, wave[i].off1)] for i in b.sharding.subgrids_of(len(wave))])
                  for b in bwds1]
        recvs = _virtual_all_to_all([p[0] for p in packed], [p[1] for p in packed])
        for r, b in enumerate(bwds1):
            b.unpack_wave(wave, recvs[r])
    for b in bwds1:
        idx, out = b.finish()
        for j, o in zip(idx, out):
            # different kernels and summation order than `ref`: two float32 results, each ~1e-5 from the truth
            assert float((o - ref[j]).abs().pow(2).mean().sqrt()) <= 3e-5 * float(ref[j].abs().pow(2).mean().sqrt())


@pytest.mark.parametrize("params", [TEST_PARAMS, dict(W=11.0, fov=1.0, N=1024, yB_size=352, yN_size=512, xA_size=192, xM_size=256)])
def test_backward_fused_split_matches_primitives(params):
    """The fused subgrid side of the backward pass (prepare_subgrid axis 0 + split_prepare_facets) against the
    primitive-by-primitive route (reference api_helper.py:115-139) in complex128 on the same subgrids, including
    wrapped offsets and odd subgrid sizes."""
    import torch

    sw, cfg, facet_cfgs, sg_cfgs, _ = small_problem(params, numpy.complex64, 5)
    core = cfg.core
    assert core.supports_fused_subgrid(torch.complex64)
    xA = params["xA_size"]
    wave = [c for c in sg_cfgs if c.off0 == sg_cfgs[-1].off0]
    gen = torch.Generator(device="cpu").manual_seed(2)
    data = [torch.randn((xA, xA), dtype=torch.complex64, generator=gen).cuda() for _ in wave]
    b32 = sw.SwiftlyBackward(cfg, facet_cfgs)
    got = b32.wave_contributions(wave, data).to(torch.complex128)
    b64 = sw.SwiftlyBackward(cfg, facet_cfgs)
    want = b64.wave_contributions(wave, [d.to(torch.complex128) for d in data])  # complex128: primitive route
    assert want.dtype == torch.complex128 and got.shape == want.shape
    err = float((got - want).abs().pow(2).mean().sqrt() / want.abs().pow(2).mean().sqrt())
    print(f"fused split relRMSE vs complex128 primitives: {err:.3e}")
    assert err < 2e-6
    for f in range(len(facet_cfgs)):
        assert float((got[f] - want[f]).abs().max()) <= 2e-5 * float(want[f].abs().max())


def test_delayed_handles_roundtrip():
    """``delayed=True``: the streaming classes hand out ``DeviceTask`` handles (the counterpart of the reference's
    Dask Delayed / Future objects, api.py:238-253): ``compute()`` gives the numpy result, the handle itself can be fed
    to ``SwiftlyBackward`` without synchronising, and the round trip equals the plain-tensor one."""
    import torch

    sw, cfg, facet_cfgs, sg_cfgs, facets = small_problem(SMALL11_PARAMS, numpy.complex64, 4321)
    plain = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)))
    want = [t.cpu().numpy() for t in plain.get_subgrid_tasks(sg_cfgs)]
    fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), delayed=True, queue_size=3)
    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, delayed=True)
    tasks = []
    for sg in sg_cfgs:
        task = fwd.get_subgrid_task(sg)
        assert isinstance(task, sw.DeviceTask) and task.shape == (sg.size, sg.size) and task.dtype == torch.complex64
        bwd.add_new_subgrid_task(sg, task)  # a handle is accepted wherever subgrid data is
        tasks.append(task)
    for task, w in zip(tasks, want):
        assert numpy.array_equal(task.compute(), w) and task.done()
        assert numpy.array_equal(numpy.asarray(task), task.result())
    out = bwd.finish()
    ref = sw.SwiftlyBackward(cfg, facet_cfgs)
    for sg, w in zip(sg_cfgs, want):
        ref.add_new_subgrid_task(sg, w)
    for a, b in zip(out, ref.finish()):
        assert isinstance(a, sw.DeviceTask)
        assert numpy.array_equal(a.compute(), b.cpu().numpy())
    # facets may be handed over as handles too
    fwd2 = sw.SwiftlyForward(cfg, [(c, sw.DeviceTask(torch.from_numpy(f).cuda())) for c, f in zip(facet_cfgs, facets)])
    assert numpy.array_equal(fwd2.get_subgrid_task(sg_cfgs[3]).cpu().numpy(), want[3])
