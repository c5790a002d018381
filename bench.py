#!/usr/bin/env python3
"""
bench.py -- facet -> subgrid throughput of the MI355X SwiFTly path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one complete forward pass of the workload with the facets already
resident in HBM: prepare_facet(axis 0) of every facet (K1), then for every
subgrid column the column kernel (K2), the window extraction (K3), the
per-subgrid accumulation (K4) and finish (K5).  Nothing is cached across
steps.  Prints ONE JSON line (rank 0).

Workloads (BASELINE.json configs):
  64k-sparse  (default) catalogue "64k[1]-n32k-1k": N=65536, 3x3 facets of
              22528^2, sparse subgrid set (505 of 71^2 = 10.02 %, 25 columns),
              complex64 -- the configuration BASELINE.json's metric is quoted on
              (fits one 288 GB GPU: 36.5 GB facets + 53 GB BF_F).
  8k          BASELINE configs[1] parameters (N=8192, 6x6 facets, 8x8 subgrids).
  1k          reference TEST_PARAMS (N=1024), plumbing.

For N > 1 (launched by torch.distributed.run, one rank per GPU) facets are
sharded over ranks and the contributions go through an RCCL all-to-all per
wave; total work is fixed ("strong" scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)

WORKLOADS = {
    "64k-sparse": dict(
        params=dict(W=10.875, fov=1.0, N=65536, yB_size=22528, yN_size=32768, xA_size=928, xM_size=1024),
        sparse_radius=12.65,
        name="64k[1]-n32k-1k, 3x3 facets, sparse subgrid set 505/5041 (25 columns)",
    ),
    "8k": dict(
        params=dict(W=11.0, fov=1.0, N=8192, yB_size=1408, yN_size=2048, xA_size=1024, xM_size=2048),
        sparse_radius=None,
        name="N=8192 custom (W=11, yB=1408, yN=2048, xA=1024, xM=2048), 6x6 facets -> 8x8 subgrids",
    ),
    "1k": dict(
        params=dict(W=13.5625, fov=1.0, N=1024, yB_size=416, yN_size=512, xA_size=228, xM_size=256),
        sparse_radius=None,
        name="reference TEST_PARAMS N=1024, 3x3 facets -> 5x5 subgrids",
    ),
}


def select_subgrids(all_cfgs, N, xA, radius):
    """Sparse set: subgrids whose wrapped index distance from the grid origin is
    <= radius (mirrors scripts/demo_sparse_facet.py:63-134, which does the same
    for facets)."""
    if radius is None:
        return list(all_cfgs)
    n = -(-N // xA)
    out = []
    for c in all_cfgs:
        i0, i1 = c.off0 // xA, c.off1 // xA
        if min(i0, n - i0) ** 2 + min(i1, n - i1) ** 2 <= radius**2:
            out.append(c)
    return out


def algorithmic_bytes(p, F, S, C):
    """Compulsory-traffic model B_fwd of SURVEY.md section 8(d), complex64."""
    E = 8
    yB, yN, xA, xM, N = p["yB_size"], p["yN_size"], p["xA_size"], p["xM_size"], p["N"]
    m = xM * yN // N
    parts = dict(
        K1=E * F * (yB * yB + yN * yB),
        K2=E * F * C * (m * yB + m * yN),
        K3=E * F * S * 2 * m * m,
        K4=E * F * S * m * m,
        K5=E * S * xA * xA,
    )
    return sum(parts.values()), parts


class StageTimer:
    """HIP-event timing of launch groups on the stream the kernels run on
    (torch's current stream == the stream handed to the C ABI)."""

    def __init__(self, torch):
        self.torch = torch
        self.pairs = {}

    def start(self):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, name, ev0):
        ev1 = self.torch.cuda.Event(enable_timing=True)
        ev1.record()
        self.pairs.setdefault(name, []).append((ev0, ev1))

    def totals(self):
        """name -> (count, total ms); call after synchronize"""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.pairs.items()}


def run_forward(sw, torch, cfg, fwd_factory, waves, timer=None):
    """One full forward pass.  Returns the number of finished subgrids."""
    fwd = fwd_factory()
    core = cfg.core
    count = 0
    if timer is None:
        fwd._get_BF_Fs()  # pylint: disable=protected-access
        for wave in waves:
            res = fwd.get_subgrid_tasks(wave)
            count += len(res)
        return count
    # instrumented pass: same launches, events around each stage
    # K1 exactly as SwiftlyForward._get_BF_Fs issues it, one event pair per facet
    pre = fwd.dtype == torch.complex64
    fwd._prewindowed = pre  # pylint: disable=protected-access
    n_rows = fwd._n_rows if fwd._rowmap is not None else core.yN_size  # pylint: disable=protected-access
    bfs = []
    for c, data in zip(fwd.facet_configs, fwd._facets):  # pylint: disable=protected-access
        t0 = timer.start()
        bfs.append(
            core.prepare_facet_rows(data, c.off0, fwd._rowmap, n_rows, fold_axis1_window=pre)  # pylint: disable=protected-access
        )
        timer.stop("K1_prepare_facet_axis0", t0)
    fwd.BF_Fs_persist = bfs
    for wave in waves:
        t0 = timer.start()
        fwd.get_NMBF_BFs_off0(wave[0].off0)
        timer.stop("K2_extract_column", t0)
        t0 = timer.start()
        res = fwd._wave(wave)  # pylint: disable=protected-access
        timer.stop("K345_extract_sum_finish", t0)
        count += res.shape[0]
    return count


def cpu_baseline(p, F, S, C, budget_s=25.0):
    """Oracle (numpy restatement of the reference, complex128 like the
    reference's numpy path) timed on ONE host core on a bounded sample of the
    same workload, extrapolated linearly by unit counts."""
    from oracle import swiftly_oracle as orc  # checker / baseline only

    yB, yN, xA, xM, N = p["yB_size"], p["yN_size"], p["xA_size"], p["xM_size"], p["N"]
    core = orc.OracleCore(p["W"], N, xM, yN)
    m = core.xM_yN_size
    rng = numpy.random.default_rng(0)
    ncol = max(1, min(yB, int(8.0e6 // yN)))  # K1 slab: column-independent
    slab = (rng.standard_normal((yB, ncol)) + 1j * rng.standard_normal((yB, ncol))).astype(numpy.complex64)
    t0 = time.perf_counter()
    bf = core.prepare_facet(slab, 0, axis=0)
    t_k1 = (time.perf_counter() - t0) * (yB / ncol)
    # K2 on a row slab of one (facet, column)
    nrow = max(1, min(m, int(1.6e7 // yN)))
    rows = (rng.standard_normal((nrow, yB)) + 1j * rng.standard_normal((nrow, yB))).astype(numpy.complex64)
    t0 = time.perf_counter()
    col = core.prepare_facet(rows, 0, axis=1)
    t_k2 = (time.perf_counter() - t0) * (m / nrow)
    del bf
    # K3..K5 for one subgrid with all F contributions
    colfull = numpy.zeros((m, yN), dtype=complex)
    colfull[:nrow] = col
    items = orc.make_full_cover(N, yB)[:F]
    sg = orc.CoverItem(0, 0, xA, numpy.ones(xA), numpy.ones(xA))
    t0 = time.perf_counter()
    contribs = [core.extract_from_facet(colfull, 0, axis=1) for _ in range(F)]
    orc.sum_and_finish_subgrid(core, contribs, items, sg)
    t_sg = time.perf_counter() - t0
    total = F * t_k1 + F * C * t_k2 + S * t_sg
    return dict(
        value=F * S / total,
        unit="contributions/s",
        cores=1,
        kind="port",
        sample=(
            f"oracle (numpy, complex128) on 1 core: K1 on a {yB}x{ncol} column slab of one facet, "
            f"K2 on {nrow} of {m} rows of one (facet, column), K3-K5 for one subgrid with {F} contributions; "
            f"extrapolated linearly to {F} facets x {C} columns x {S} subgrids "
            f"(K1 {F * t_k1:.1f} s + K2 {F * C * t_k2:.1f} s + K3-5 {S * t_sg:.1f} s)"
        ),
        extrapolated_seconds=total,
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="64k-sparse", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import ska_sdp_exec_swiftly_amd as sw
    from ska_sdp_exec_swiftly_amd import api as sw_api
    from ska_sdp_exec_swiftly_amd.distributed import DistributedForward

    sw.api = sw_api
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    if args.gpus != world:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    wl = WORKLOADS[args.workload]
    p = wl["params"]
    cfg = sw.SwiftlyConfig(backend="hip", **p)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    sg_cfgs = select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off0, []).append(c)
    waves = list(waves.values())
    F, S, C = len(facet_cfgs), len(sg_cfgs), len(waves)

    # synthetic dense facets N(0,1)+iN(0,1), complex64, generated on the device
    # (seed 1234 + facet index), times the cover masks
    yB = p["yB_size"]
    local = [j for j in range(F) if j % world == rank]
    facet_data = [None] * F
    for j in local:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1234 + j)
        re = torch.randn((yB, yB), generator=gen, device="cuda", dtype=torch.float32)
        im = torch.randn((yB, yB), generator=gen, device="cuda", dtype=torch.float32)
        m0 = torch.from_numpy(facet_cfgs[j].mask0).to("cuda", torch.float32)
        m1 = torch.from_numpy(facet_cfgs[j].mask1).to("cuda", torch.float32)
        facet_data[j] = torch.complex(re, im) * m0[:, None] * m1[None, :]
        del re, im

    force_dist = os.environ.get("SWIFTLY_BENCH_FORCE_DIST") == "1"  # exercise the multi-GPU code path on 1 GPU
    if world == 1 and not force_dist:

        def factory():
            return sw.SwiftlyForward(
                cfg, [(facet_cfgs[j], facet_data[j]) for j in range(F)], lru_forward=1, subgrid_configs=sg_cfgs
            )

        def one_pass(timer=None):
            return run_forward(sw, torch, cfg, factory, waves, timer)

    else:

        def one_pass(timer=None):  # pylint: disable=unused-argument
            dfw = DistributedForward(cfg, facet_cfgs, facet_data, lru_forward=1, subgrid_configs=sg_cfgs)
            dfw.local._get_BF_Fs()  # pylint: disable=protected-access
            # software pipeline: the all-to-all of wave w runs while wave w+1's column/extract kernels do
            n = 0
            pending = None
            for wave in waves:
                handle = dfw.start_wave(wave)
                if pending is not None:
                    n += len(dfw.finish_wave(pending)[0])
                pending = handle
            n += len(dfw.finish_wave(pending)[0])
            return n

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)

    # per-stage HIP-event timing (separate instrumented pass, 1 GPU only)
    stages = {}
    roofline = None
    total_bytes, parts = algorithmic_bytes(p, F, S, C)
    if world == 1 and not force_dist:
        timer = StageTimer(torch)
        one_pass(timer)
        torch.cuda.synchronize()
        for name, (cnt, ms) in timer.totals().items():
            stages[name] = dict(launch_groups=cnt, total_ms=round(ms, 3), avg_ms=round(ms / cnt, 4))
        k1 = stages["K1_prepare_facet_axis0"]
        k1_bytes = parts["K1"] / F  # per facet = per launch group
        achieved = k1_bytes / (k1["avg_ms"] * 1e-3) / 1e9
        # HBM bytes per K1 launch group from rocprofv3 PMC passes of the same kernels (tools/run_k1_plan.py,
        # profiles/r1d_pmc_k1_traffic.txt): FETCH_SIZE doubled (gfx950 counts 64 B per 128 B request) + WRITE_SIZE
        k1_traffic = {"64k-sparse": 2 * (1994492 + 2885678) * 1024 + (5767168 + 2019072) * 1024}.get(args.workload)
        roofline = dict(
            kernel="K1 prepare_facet(axis=0) per facet = col_pass<n1=128, mapped load> + col_pass<n2=256, mapped store>",
            bound="hbm",
            achieved=round(achieved, 1),
            peak=HBM_PEAK_GBS,
            unit="GB/s",
            frac=round(achieved / HBM_PEAK_GBS, 4),
            traffic=k1_traffic,
            traffic_note="bytes per launch group measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)",
            algorithmic_bytes_per_launch=k1_bytes,
            avg_launch_ms=k1["avg_ms"],
        )

    if roofline is None and local:
        # multi-GPU: time K1 of one local facet on this rank (outside the timed region) for the roofline object
        core = cfg.core
        rowmap, n_rows = core.subgrid_column_rows([sg.off0 for sg in sg_cfgs])
        j0 = local[0]
        buf = core.prepare_facet_rows(facet_data[j0], facet_cfgs[j0].off0, rowmap, n_rows, fold_axis1_window=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            core.prepare_facet_rows(facet_data[j0], facet_cfgs[j0].off0, rowmap, n_rows, out=buf, fold_axis1_window=True)
        e1.record()
        torch.cuda.synchronize()
        k1_ms = e0.elapsed_time(e1) / 3
        k1_bytes = parts["K1"] / F
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
        roofline = dict(
            kernel="K1 prepare_facet(axis=0) per facet (rank 0) = col_pass<n1> + col_pass<n2>",
            bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
            algorithmic_bytes_per_launch=k1_bytes, avg_launch_ms=round(k1_ms, 4),
        )
        del buf

    line = dict(
        metric="facet_to_subgrid_contributions_per_s",
        value=round(F * S / (ms_per_step * 1e-3), 1),
        unit="contributions/s",
        n_gpus=world,
        steps=args.steps,
        warmup=args.warmup,
        ms_per_step=round(ms_per_step, 3),
        higher_is_better=True,
        scaling="strong",
        vs_baseline=None,
        dtype="complex64 (f32 arithmetic)",
        data="synthetic",
        config=dict(
            workload=wl["name"], facets=F, subgrids=S, subgrid_columns=C, contributions=F * S, params=p,
            parallelism=f"facets sharded over {world} rank(s), contribution all-to-all" if world > 1 else "1 GPU",
        ),
        hbm_algorithmic_gbs=round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
        hbm_algorithmic_frac_of_peak=round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4),
        algorithmic_bytes=dict(total=total_bytes, **parts),
        stages=stages,
        roofline=roofline,
    )
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(p, F, S, C)
            line["cpu_baseline"]["value"] = round(line["cpu_baseline"]["value"], 3)
            line["speedup_vs_cpu_core"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
