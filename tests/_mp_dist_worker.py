"""Worker of tests/test_hip_multiprocess_gpu.py: one of WORLD_SIZE processes that share GPU 0, joined by a gloo
process group (RCCL cannot connect several ranks on one device): runs the facet-sharded forward and backward
transforms through the REAL process-group code path (torch.distributed collectives, split sizes, pending handles)
with real HIP kernels, and compares with the single-process classes on rank 0."""
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))


def main():
    import torch
    import torch.distributed as dist

    import ska_sdp_exec_swiftly_amd as sw
    from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward, DistributedForward

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = dict(W=11.0, fov=1.0, N=1024, yB_size=352, yN_size=512, xA_size=192, xM_size=256)
    cfg = sw.SwiftlyConfig(backend="hip", **params)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    sg_cfgs = sw.make_full_subgrid_cover(cfg)
    yB = params["yB_size"]
    facets = []
    for j, f in enumerate(facet_cfgs):
        r = numpy.random.default_rng(50 + j)
        d = (r.standard_normal((yB, yB)) + 1j * r.standard_normal((yB, yB))).astype(numpy.complex64)
        facets.append(torch.from_numpy(d * f.mask0[:, None] * f.mask1[None, :]).to(torch.complex64).cuda())
    local = [j for j in range(len(facet_cfgs)) if j % world == rank]
    data = [facets[j] if j in local else None for j in range(len(facet_cfgs))]
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off0, []).append(c)
    waves = list(waves.values())
    # forward: pipelined like bench.py
    dfw = DistributedForward(cfg, facet_cfgs, data, lru_forward=1, dtype=torch.complex64, wave_axis=0)
    assert (dfw.rank, dfw.world) == (rank, world)
    got, pending = {}, None
    for wave in waves + [None]:
        handle = (dfw.start_wave(wave), wave) if wave is not None else None
        if pending is not None:
            mine, res = dfw.finish_wave(pending[0])
            for k, i in enumerate(mine):
                c = pending[1][i]
                got[(c.off0, c.off1)] = res[k]
        pending = handle
    # every rank learns all subgrids (object gather of host copies), rank 0 checks against the single-process class
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: v.cpu().numpy() for k, v in got.items()})
    everything = {k: v for part in gathered for k, v in part.items()}
    assert sorted(everything) == sorted((c.off0, c.off1) for c in sg_cfgs)
    ref_fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), wave_axis=0)
    ref_sub = {}
    for wave in waves:
        res = ref_fwd.get_wave(wave)
        for k, c in enumerate(wave):
            ref_sub[(c.off0, c.off1)] = res[k]
            a, b = everything[(c.off0, c.off1)], res[k].cpu().numpy()
            assert numpy.abs(a - b).max() <= 2e-5 * numpy.abs(b).max(), (rank, c.off0, c.off1)
    # backward, both schedules: every rank feeds the subgrids it "holds"
    for axis in (0, 1):
        key = (lambda c: c.off1) if axis == 1 else (lambda c: c.off0)
        bw = {}
        for c in sg_cfgs:
            bw.setdefault(key(c), []).append(c)
        dbw = DistributedBackward(cfg, facet_cfgs, wave_axis=axis, subgrid_configs=sg_cfgs)
        ref_bwd = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=axis, subgrid_configs=sg_cfgs)
        pending = None
        for wave in bw.values():
            mine = dbw.sharding.subgrids_of(len(wave))
            handle = dbw.start_wave(wave, [ref_sub[(wave[i].off0, wave[i].off1)] for i in mine])
            if pending is not None:
                dbw.finish_wave(pending)
            pending = handle
            ref_bwd.add_new_subgrid_tasks(wave, [ref_sub[(c.off0, c.off1)] for c in wave])
        dbw.finish_wave(pending)
        idx, out = dbw.finish()
        ref = ref_bwd.finish()
        assert idx == local
        for j, o in zip(idx, out):
            rms = float(ref[j].abs().pow(2).mean().sqrt())
            assert float((o - ref[j]).abs().pow(2).mean().sqrt()) <= 3e-5 * rms, (rank, axis, j)
    # complex128 (reference schedule): with 9 facets the ranks that carry an extra facet hold NO subgrid in any wave, so
    # the dtype of their (empty) send buffer and of their receive buffer comes from the declared `dtype` only
    dbw = DistributedBackward(cfg, facet_cfgs, wave_axis=0, dtype=torch.complex128)
    ref_bwd = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=0)
    if world > 1 and len(facet_cfgs) % world:
        assert any(not dbw.sharding.subgrids_of(len(w), r) for w in waves for r in range(world))
    for wave in waves:
        mine = dbw.sharding.subgrids_of(len(wave))
        dbw.add_wave(wave, [ref_sub[(wave[i].off0, wave[i].off1)].to(torch.complex128) for i in mine])
        ref_bwd.add_new_subgrid_tasks(wave, [ref_sub[(c.off0, c.off1)].to(torch.complex128) for c in wave])
    idx, out = dbw.finish()
    ref = ref_bwd.finish()
    for j, o in zip(idx, out):
        assert o.dtype == torch.complex128
        assert float((o - ref[j]).abs().max()) <= 1e-10 * float(ref[j].abs().max()), (rank, "c128", j)
    cooperative_section(rank, world)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world}: ok", flush=True)


def cooperative_section(rank, world):
    """Cooperative facets (r4) through the real process group: a facet count that does not divide by the world size at
    yN = 32768 (split band layout), band pipelines with a plan -- K1 on row blocks, the band-row exchange, per-wave
    ownership of the leftover facet, the finishing exchange of the backward pass -- against the single-process classes."""
    import torch
    import torch.distributed as dist

    import ska_sdp_exec_swiftly_amd as sw
    from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward, DistributedForward

    yB, xA = 352, 928
    P = dict(W=10.875, fov=1.0, N=65536, yB_size=yB, yN_size=32768, xA_size=xA, xM_size=1024)
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    if not cfg.core.supports_band_pipeline(torch.complex64):
        return
    fstep = cfg.facet_off_step
    offs = [(0, 0), (0, 50 * fstep), (-70 * fstep, 0), (40 * fstep, -30 * fstep)][: (3 if world == 2 else 4)]
    facet_cfgs = [sw.FacetConfig(o0, o1, yB) for o0, o1 in offs]
    assert len(facet_cfgs) % world
    gen = torch.Generator(device="cpu").manual_seed(77)
    facets = [torch.randn((yB, yB), dtype=torch.complex64, generator=gen).cuda() for _ in facet_cfgs]
    sg_cfgs = [sw.SubgridConfig(i0 * xA, i1 * xA, xA) for i0 in (0, 2, 69) for i1 in (0, 3, 4, 70)]
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off1, []).append(c)
    dfw = DistributedForward(cfg, facet_cfgs, facets, subgrid_configs=sg_cfgs, wave_axis=1, dtype=torch.complex64)
    sh = dfw.sharding
    assert sh.coop == list(range((len(facet_cfgs) // world) * world, len(facet_cfgs)))
    dfw.prepare_all_facets()
    ref = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)
    full = {}
    pending = None
    for wave in list(waves.values()) + [None]:
        handle = dfw.start_wave(wave) if wave is not None else None
        if pending is not None:
            mine, res = dfw.finish_wave(pending)
            want = ref.get_wave(pending[0])
            for k, i in enumerate(mine or []):
                assert float((res[k] - want[i]).abs().max()) <= 2e-5 * float(want.abs().max()), (rank, "coop fwd", i)
            for i, c in enumerate(pending[0]):
                full[(c.off0, c.off1)] = want[i]
        pending = handle
    dbw = DistributedBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs, dtype=torch.complex64)
    rb = sw.SwiftlyBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs)
    for wave in waves.values():
        mine = dbw.sharding.subgrids_of(len(wave))
        dbw.add_wave(wave, [full[(wave[i].off0, wave[i].off1)] for i in mine])
        rb.add_new_subgrid_tasks(wave, [full[(c.off0, c.off1)] for c in wave])
    idx, out = dbw.finish()
    want = rb.finish()
    for j, o in zip(idx, out):
        rms = float(want[j].abs().pow(2).mean().sqrt())
        assert float((o - want[j]).abs().pow(2).mean().sqrt()) <= 3e-6 * rms, (rank, "coop bwd whole", j)
    assert sorted(p[0] for p in dbw.coop_pieces) == sh.coop
    for j, row0, piece in dbw.coop_pieces:
        w = want[j][row0 : row0 + piece.shape[0]]
        assert piece.shape[0] == sh.coop_rows(yB)[1]
        rms = float(want[j].abs().pow(2).mean().sqrt())
        assert float((piece - w).abs().pow(2).mean().sqrt()) <= 3e-6 * rms, (rank, "coop bwd piece", j)
    dist.barrier()
    # whole-wave ownership with one all-to-all per group of waves (bench.py's default multi-GPU schedule), streaming
    # forward -> backward like the round-trip leg of bench.py
    dfw = DistributedForward(cfg, facet_cfgs, facets, subgrid_configs=sg_cfgs, wave_axis=1, dtype=torch.complex64,
                             whole_waves=True)
    dbw = DistributedBackward(cfg, facet_cfgs, wave_axis=1, subgrid_configs=sg_cfgs, dtype=torch.complex64, whole_waves=True)
    dfw.prepare_all_facets()
    by_key = {int(k): w for k, w in waves.items()}
    pend_f = pend_b = None
    for group in dfw.sharding.wave_groups + [None]:
        gw = [by_key[k] for k in group] if group is not None else None
        hf = (gw, dfw.start_group(gw)) if gw is not None else None
        if pend_f is not None:
            sgs, res = dfw.finish_group(pend_f[1])
            if sgs is not None:
                for k, c in enumerate(sgs):
                    w = full[(c.off0, c.off1)]
                    assert float((res[k] - w).abs().max()) <= 2e-5 * float(w.abs().max()), (rank, "group fwd", c.off0, c.off1)
            hb = dbw.start_group(pend_f[0], [full[(c.off0, c.off1)] for c in sgs] if sgs is not None else [])
            if pend_b is not None:
                dbw.finish_group(pend_b)
            pend_b = hb
        pend_f = hf
    dbw.finish_group(pend_b)
    idx, out = dbw.finish()
    for j, o in zip(idx, out):
        rms = float(want[j].abs().pow(2).mean().sqrt())
        assert float((o - want[j]).abs().pow(2).mean().sqrt()) <= 3e-6 * rms, (rank, "group bwd whole", j)
    for j, row0, piece in dbw.coop_pieces:
        w = want[j][row0 : row0 + piece.shape[0]]
        rms = float(want[j].abs().pow(2).mean().sqrt())
        assert float((piece - w).abs().pow(2).mean().sqrt()) <= 3e-6 * rms, (rank, "group bwd piece", j)
    dist.barrier()


if __name__ == "__main__":
    main()
