#!/usr/bin/env python3
"""
bench.py -- facet -> subgrid throughput of the MI355X SwiFTly path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-verify] [--no-cpu-baseline]

One "step" = one complete forward pass of the workload with the facets already
resident in HBM: the full-facet transform of every facet (K1), then for every
subgrid wave the per-wave facet transform (K2), the window extraction (K3), the
per-subgrid accumulation (K4) and finish (K5).  Nothing is cached across
steps.  Prints ONE JSON line (rank 0).

Workloads (BASELINE.json configs; `--workload`): the default is the one the metric is
quoted on; "32k-8x8" (configs[2]), "128k" / "128k-8x8" (configs[4], forward + backward round trip; one rank holds
a stated SUBSET of the facets -- `config.facets` of `config.facets_total` -- because 288 GB do not hold all of
them), "64k-sparse-4x4" (the headline problem with a cover that divides by 8):
  64k-sparse  (default) catalogue "64k[1]-n32k-1k": N=65536, 3x3 facets of
              22528^2, sparse subgrid set (505 of 71^2 = 10.02 %, 25 columns),
              complex64 -- the configuration BASELINE.json's metric is quoted on
              (fits one 288 GB GPU: 36.5 GB facets + intermediates).
  8k          BASELINE configs[1] parameters (N=8192, 6x6 facets, 8x8 subgrids).
  1k          reference TEST_PARAMS (N=1024), plumbing.

Synthetic data.  Facet j = sum_{r<2} a_{j,r} (x) b_{j,r} times the cover masks,
with seeded random vectors on the 1/8 grid (oracle/separable.py): DENSE
random-looking complex64 arrays whose exact forward result the CPU oracle can
evaluate per subgrid in O(1 s) even at N = 65536.  After the timed region the
subgrids the TIMED objects produce are compared with the oracle ("parity" in
the JSON line) -- the benchmarked code path is the verified code path.

--gpus N > 1: if not already running under torch.distributed.run, bench.py
re-launches itself with N ranks (one per GPU, RCCL); facets are sharded over
ranks and the contributions go through an RCCL all-to-all per wave; total work
is fixed ("strong" scaling).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# complex64 relative RMSE bounds vs the complex128 oracle (tests/accuracy_model.py).  float32
# arithmetic everywhere (default): measured 1.0e-5 ... 1.3e-5 forward, 2.1e-5 ... 2.6e-5 backward over the workloads.
# float64 arithmetic in the column passes K2 / K3 (column_precision = 64): 2.8e-6 forward, 5.1e-6 backward; the float32
# STORAGE floor of this dataflow is 3.3e-6 on the probe configuration (r3's "1.0e-5 floor" came from a faulty probe).
PARITY_TOL = 1.5e-5
BACKWARD_PARITY_TOL = 3.5e-5
HIGH_PRECISION_PARITY_TOL = 5e-6
# forward relRMSE bound of the axis-1-first pipeline (float32 arithmetic, SwiftlyConfig(axis1_first=True); model: 2.1e-6)
AXIS1_FIRST_PARITY_TOL = 4e-6
HIGH_PRECISION_BACKWARD_PARITY_TOL = 1e-5

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
    # BASELINE.json configs[2] (SURVEY section 8d cfg 3, literal): 64 facets -> one rank's share divides by 8
    "32k-8x8": dict(
        params=dict(W=11.0, fov=1.0, N=32768, yB_size=4096, yN_size=8192, xA_size=2048, xM_size=4096),
        sparse_radius=None,
        name="N=32768 custom (W=11, yB=4096, yN=8192, xA=2048, xM=4096 -> m=1024), 8x8 facets -> 16x16 subgrids",
        roundtrip=True,
    ),
    # the headline problem with a facet cover that divides by 8 (same kernels: yN = 32768, xM = 1024, m = 512)
    "64k-sparse-4x4": dict(
        params=dict(W=10.875, fov=1.0, N=65536, yB_size=16384, yN_size=32768, xA_size=928, xM_size=1024),
        sparse_radius=12.65,
        name="N=65536 custom cover (yB=16384, yN=32768): 4x4 facets, sparse subgrid set 505/5041 (25 columns)",
    ),
    # BASELINE.json configs[4]: catalogue 128k[1]-n64k-1k (reference swift_configs.py:998), forward + backward round trip.
    # 9 facets of 45056^2 (16.2 GB) + 23.6 GB band buffer each do not fit one 288 GB GPU: a rank holds at most 2
    "128k": dict(
        params=dict(W=10.875, fov=1.0, N=131072, yB_size=45056, yN_size=65536, xA_size=928, xM_size=1024),
        sparse_radius=None,
        name="128k[1]-n64k-1k, 3x3 facets of 45056^2, full subgrid cover 142^2",
        max_facets_per_rank=2, roundtrip=True, parity_radius=9.0,
    ),
    # SURVEY section 8d cfg 5 "preferred custom": 8x8 facets of 16384^2 (8 per rank on 8 GPUs), full subgrid cover
    "128k-8x8": dict(
        params=dict(W=10.875, fov=1.0, N=131072, yB_size=16384, yN_size=32768, xA_size=928, xM_size=1024),
        sparse_radius=None,
        name="N=131072 custom (yB=16384, yN=32768, xA=928, xM=1024 -> m=256), 8x8 facets -> 142^2 subgrids",
        max_facets_per_rank=8, roundtrip=True, parity_radius=9.0,
    ),
    # a catalogue entry whose facet side is not a power of two (yN = 3 * 2048): radix-3 pass + power-of-two kernels
    # (csrc/swiftly_mixed.h); SWIFTLY_NO_MIXED=1 times the Bluestein fallback instead
    "12k": dict(
        params=dict(W=11.0, fov=1.0, N=12288, yB_size=4224, yN_size=6144, xA_size=448, xM_size=512),
        sparse_radius=None,
        name="12k[1]-n6k-512 (yN = 6144 = 3 * 2^11), 3x3 facets -> 28x28 subgrids",
    ),
    # the 64k kernels' sizes (m = 512, xM = 1024) behind a facet side of 3 * 2^12
    "24k": dict(
        params=dict(W=10.875, fov=1.0, N=24576, yB_size=8448, yN_size=12288, xA_size=928, xM_size=1024),
        sparse_radius=None,
        name="24k[1]-n12k-1k (yN = 12288 = 3 * 2^12), 3x3 facets -> 27x27 subgrids",
    ),
    "1k": dict(
        params=dict(W=13.5625, fov=1.0, N=1024, yB_size=416, yN_size=512, xA_size=228, xM_size=256),
        sparse_radius=None,
        name="reference TEST_PARAMS N=1024, 3x3 facets -> 5x5 subgrids",
        # W = 13.56 has max 1/pswf ~ 4.9e3: float32 storage alone gives ~8e-3 here; plumbing only
        parity_tol=3e-2,
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


# --------------------------------------------------------------------------- synthetic facets
def separable_facet(torch, vec, cfg, pixels=None, device="cuda"):
    """Device facet ``sum_r a_r (x) b_r`` (times the cover masks, plus optional
    point-source pixels) as complex64 -- exactly the numbers the separable
    oracle uses (components on the 1/8 grid: products and sums are exact in
    float32)."""
    yB = cfg.size
    out = torch.zeros((yB, yB), dtype=torch.complex64, device=device)
    if vec is not None:
        a, b = vec
        m0 = cfg.mask0 if cfg.mask0 is not None else numpy.ones(yB)
        m1 = cfg.mask1 if cfg.mask1 is not None else numpy.ones(yB)
        for r in range(a.shape[0]):
            ta = torch.from_numpy((a[r] * m0).astype(numpy.complex64)).to(device)
            tb = torch.from_numpy((b[r] * m1).astype(numpy.complex64)).to(device)
            out.add_(torch.outer(ta, tb))
    for p0, p1, val in pixels or []:
        out[p0, p1] += complex(val)
    return out


def verify_subgrids(p, facet_cfgs, vectors, sg_cfgs, got_by_index, pixels=None, tol=None):
    """Compare finished subgrids (dict: index into sg_cfgs -> numpy array) with the separable oracle.
    Returns the "parity" object of the JSON line."""
    from oracle import separable as sep  # checker only
    from oracle import swiftly_oracle as orc

    core = orc.OracleCore(p["W"], p["N"], p["xM_size"], p["yN_size"])
    items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    so = sep.SeparableOracle(core, items, vectors, pixels)
    tol = PARITY_TOL if tol is None else tol
    rels, maxs = [], []
    for i, got in sorted(got_by_index.items()):
        c = sg_cfgs[i]
        want = so.subgrid(orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1))
        rms = float(numpy.sqrt(numpy.mean(numpy.abs(want) ** 2)))
        err = numpy.abs(got - want)
        rels.append(float(numpy.sqrt(numpy.mean(err**2))) / rms)
        maxs.append(float(err.max()) / rms)
    return dict(
        checker="oracle/separable.py (numpy, complex128) on the subgrids produced by the timed objects",
        subgrids=len(rels),
        subgrid_offsets=[[int(sg_cfgs[i].off0), int(sg_cfgs[i].off1)] for i in sorted(got_by_index)],
        rel_rmse=max(rels),
        rel_rmse_each=[float(f"{r:.3e}") for r in rels],
        max_abs_over_rms=max(maxs),
        tol_rel_rmse=tol,
        # BASELINE.md section 3 budgeted 1e-6 assuming float32 STORAGE rounding of un-amplified data; the facet-side
        # intermediates carry data amplified by 1/pswf (<= 90 per axis for W = 10.875): with float64 arithmetic and every
        # stored intermediate rounded to complex64 the chain gives 2.3e-6 on the probe configuration (N = 8192), 3.3e-6
        # with the complex64 four-step scratch of K2 (tests/accuracy_model.py, pinned by tests/test_accuracy_budget_cpu.py)
        storage_floor_rel_rmse=3.3e-6,
        storage_floor_source="tests/accuracy_model.py (probe configuration N = 8192), DESIGN.md section 2",
        ok=bool(max(rels) < tol),
    )


def verify_facets(p, facet_cfgs, sg_cfgs, sg_vectors, facets_out, rows_per_facet=16, tol=None):
    """Compare sampled rows of the finished facets of a backward pass over separable subgrids (device tensors
    ``facets_out``, one per facet config) with oracle/separable.py:SeparableBackwardOracle, element by element.
    Returns the ``backward.parity`` object of the JSON line."""
    from oracle import separable as sep  # checker only
    from oracle import swiftly_oracle as orc

    core = orc.OracleCore(p["W"], p["N"], p["xM_size"], p["yN_size"])
    f_items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in facet_cfgs]
    s_items = [orc.CoverItem(c.off0, c.off1, c.size, c.mask0, c.mask1) for c in sg_cfgs]
    so = sep.SeparableBackwardOracle(core, f_items, s_items, sg_vectors)
    tol = BACKWARD_PARITY_TOL if tol is None else tol
    rels, maxs, rows_used = [], [], []
    for j, (item, got_t) in enumerate(zip(f_items, facets_out)):
        yB = item.size
        # rows inside the facet's mask (the others are identically zero on both sides), spread over the facet
        inside = numpy.flatnonzero(numpy.asarray(item.mask0) != 0) if item.mask0 is not None else numpy.arange(yB)
        rng = numpy.random.default_rng(97 + j)
        n = min(rows_per_facet, inside.size)
        rows = numpy.sort(numpy.concatenate([inside[[0, -1]], rng.choice(inside[1:-1], size=n - 2, replace=False)]))
        want = so.facet_rows(j, rows)
        got = got_t[rows.tolist()].cpu().numpy()
        rms = float(numpy.sqrt(numpy.mean(numpy.abs(want) ** 2)))
        err = numpy.abs(got - want)
        rels.append(float(numpy.sqrt(numpy.mean(err**2))) / rms)
        maxs.append(float(err.max()) / rms)
        rows_used.append(int(n))
    return dict(
        checker="oracle/separable.py:SeparableBackwardOracle (numpy, complex128): sampled rows of every finished facet, element by element",
        facets=len(rels),
        rows_per_facet=min(rows_used),
        subgrids=len(sg_cfgs),
        rel_rmse=max(rels),
        rel_rmse_each=[float(f"{r:.3e}") for r in rels],
        max_abs_over_rms=max(maxs),
        tol_rel_rmse=tol,
        ok=bool(max(rels) < tol),
    )


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


# --------------------------------------------------------------------------- CPU baseline
REF_SRC = "/root/reference/src"  # only present in the authoring container, never on the GPU box


def _import_reference():
    """The reference's own numpy core + 2-D task bodies (``SwiftlyCore``, ``api_helper``), imported UNCHANGED from
    ``/root/reference/src`` with stub modules for its absent dependencies (ska_sdp_func, dask, distributed: none of
    them is on the numpy path) -- the recipe of SURVEY.md appendix D / tests/golden/make_golden.py.  None when the
    reference is not there (GPU box): the caller falls back to the oracle port."""
    import types

    if not os.path.isdir(REF_SRC):
        return None

    def stub(name, **attrs):
        mod = sys.modules.get(name)
        if mod is None:
            mod = types.ModuleType(name)
            sys.modules[name] = mod
        mod.__dict__.update(attrs)
        return mod

    try:
        a, b = stub("ska_sdp_func"), stub("ska_sdp_func.fourier_transforms")
        a.fourier_transforms, b.swiftly = b, stub("ska_sdp_func.fourier_transforms.swiftly")

        class _NoClient:
            @staticmethod
            def current():
                raise RuntimeError("dask is not installed")

        d = stub("dask", delayed=lambda *x, **k: None)
        d.array, d.distributed = stub("dask.array"), stub("dask.distributed", Client=_NoClient)
        stub("distributed", Client=_NoClient)
        if REF_SRC not in sys.path:
            sys.path.insert(0, REF_SRC)
        from ska_sdp_exec_swiftly import api_helper  # pylint: disable=import-outside-toplevel
        from ska_sdp_exec_swiftly.fourier_transform.core import SwiftlyCore  # pylint: disable=import-outside-toplevel

        return SwiftlyCore, api_helper
    except Exception:  # pylint: disable=broad-except
        return None


_CPU_CORE = {}
CPU_BASELINE_PROCS = int(os.environ.get("SWIFTLY_BENCH_CPU_PROCS", "32"))


def _cpu_pin(slot, pins):
    """worker initialiser: pin this process to one CPU of the spread (numpy's pocketfft is single-threaded)"""
    with slot.get_lock():
        i = slot.value
        slot.value += 1
    try:
        os.sched_setaffinity(0, {pins[i % len(pins)]})
    except (AttributeError, OSError):  # pragma: no cover
        pass


def _cpu_warm(args):
    """Untimed: bring one worker process up (imports, the PSWF and window tables of its core) so that the timed repeats
    of all workers start together and run under full contention from their first second."""
    p, _F, _seed = args
    from oracle import swiftly_oracle as orc  # checker / baseline only

    key = (p["W"], p["N"], p["xM_size"], p["yN_size"])
    if _CPU_CORE.get("key") != key:
        ref = _import_reference()
        if ref is not None:
            _CPU_CORE.update(key=key, core=ref[0](p["W"], p["N"], p["xM_size"], p["yN_size"]),
                             finish=ref[1].sum_and_finish_subgrid, kind="reference")
        else:
            _CPU_CORE.update(key=key, core=orc.OracleCore(p["W"], p["N"], p["xM_size"], p["yN_size"]),
                             finish=orc.sum_and_finish_subgrid, kind="port")
    one = numpy.zeros((p["yB_size"], 1), dtype=numpy.complex64)
    _CPU_CORE["core"].prepare_facet(one, 0, axis=0)  # first use of the length-yN transform in this process (plan, tables)
    time.sleep(0.5)  # keeps this worker busy until every other worker has taken its own warm-up task
    return _CPU_CORE["kind"]


def _cpu_sample(args):
    """One worker's share of the CPU sample (runs in a separate process): the three task kinds Dask would run --
    prepare_facet of a facet slab (api.py:281-298), extract_column of some rows (api_helper.py:200-210),
    extract + sum_and_finish_subgrid of one subgrid (api.py:255-279, api_helper.py:73-112)."""
    p, F, seed = args
    from oracle import swiftly_oracle as orc  # checker / baseline only

    yB, yN, xA, xM, N = p["yB_size"], p["yN_size"], p["xA_size"], p["xM_size"], p["N"]
    key = (p["W"], N, xM, yN)
    if _CPU_CORE.get("key") != key:  # one core per worker process, reused by the repeats (the PSWF takes seconds)
        ref = _import_reference()
        if ref is not None:
            _CPU_CORE.update(key=key, core=ref[0](p["W"], N, xM, yN),  # SwiftlyCore(W, N, xM_size, yN_size), core.py:39
                             finish=ref[1].sum_and_finish_subgrid, kind="reference")
        else:
            _CPU_CORE.update(key=key, core=orc.OracleCore(p["W"], N, xM, yN), finish=orc.sum_and_finish_subgrid, kind="port")
    core, finish, kind = _CPU_CORE["core"], _CPU_CORE["finish"], _CPU_CORE["kind"]
    m = core.xM_yN_size
    rng = numpy.random.default_rng(seed)
    ncol = max(1, min(yB, int(2.0e6 // yN)))  # K1 slab: column-independent
    slab = (rng.standard_normal((yB, ncol)) + 1j * rng.standard_normal((yB, ncol))).astype(numpy.complex64)
    t0 = time.perf_counter()
    bf = core.prepare_facet(slab, 0, axis=0)
    t_k1 = (time.perf_counter() - t0) * (yB / ncol)
    del bf
    nrow = max(1, min(m, int(4.0e6 // yN)))
    rows = (rng.standard_normal((nrow, yB)) + 1j * rng.standard_normal((nrow, yB))).astype(numpy.complex64)
    t0 = time.perf_counter()
    col = core.prepare_facet(rows, 0, axis=1)
    t_k2 = (time.perf_counter() - t0) * (m / nrow)
    colfull = numpy.zeros((m, yN), dtype=complex)
    colfull[:nrow] = col
    items = orc.make_full_cover(N, yB)[:F]
    sg = orc.CoverItem(0, 0, xA, numpy.ones(xA), numpy.ones(xA))
    t0 = time.perf_counter()
    contribs = [core.extract_from_facet(colfull, 0, axis=1) for _ in range(F)]
    finish(core, contribs, items, sg)
    t_sg = time.perf_counter() - t0
    return t_k1, t_k2, t_sg, ncol, nrow, kind


def cpu_baseline(p, F, S, C):
    """The reference's numpy path (``SwiftlyCore`` + ``api_helper`` imported unchanged from /root/reference/src
    when that exists: ``kind = "reference"``; otherwise -- on the GPU box, where the reference is absent -- the
    oracle port of the same algorithm: ``kind = "port"``; both compute in complex128) timed on ALL host cores: what Dask would
    parallelise -- independent facet column slabs, (facet, column) row slabs and
    subgrids -- runs as one process per core (numpy's pocketfft is single
    threaded), every process working on its own bounded sample concurrently (so
    memory-bandwidth contention between cores is in the numbers), extrapolated
    linearly by unit counts."""
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor

    # (r6, method 5) at most CPU_BASELINE_PROCS worker processes, each pinned to its own CPU, spread evenly over the CPUs
    # this process may use.  Methods 1-4 ran one process per CPU (256 on the GPU box): the repeats of one run spread
    # 1.6 - 2x, the value halved between rounds (24 / 189 / 192 / 145 / 90 contributions/s) and the leg took 84 s of the
    # driver's 130 -- 256 numpy processes contend for memory bandwidth and last-level cache in a different way every time.
    # `cores` in the line is the number of processes (= threads) actually used.
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = list(range(os.cpu_count() or 1))
    cores = max(1, min(CPU_BASELINE_PROCS, len(avail)))
    pins = [avail[(i * len(avail)) // cores] for i in range(cores)]
    repeats = 3
    t0 = time.perf_counter()
    per_repeat = []
    ctx = multiprocessing.get_context("spawn")  # the parent holds an initialised HIP runtime, which must not be forked
    slot = ctx.Value("i", 0)
    with ProcessPoolExecutor(cores, mp_context=ctx, initializer=_cpu_pin, initargs=(slot, pins)) as pool:
        # (r5) untimed warm-up: in r4 the first repeat ran while the workers were still being spawned -- staggered, i.e.
        # under less contention -- and came out 1.7-2.2x faster than the others (247 / 145 / 111 contributions/s)
        list(pool.map(_cpu_warm, [(p, F, i) for i in range(cores)]))
        # ... and one untimed repeat of the sample itself: the host reaches its sustained clocks only under load
        list(pool.map(_cpu_sample, [(p, F, 500 + i) for i in range(cores)]))
        for rep in range(repeats):  # every repeat keeps all workers busy at once; the pool (and each worker's core) is reused
            per_repeat.append(list(pool.map(_cpu_sample, [(p, F, 1000 + rep * cores + i) for i in range(cores)])))
    wall = time.perf_counter() - t0
    m = p["xM_size"] * p["yN_size"] // p["N"]
    values, splits = [], []
    for res in per_repeat:
        t_k1 = float(numpy.mean([r[0] for r in res]))
        t_k2 = float(numpy.mean([r[1] for r in res]))
        t_sg = float(numpy.mean([r[2] for r in res]))
        # per-unit times measured with all cores busy; units are independent, so `cores` of them run at a time
        total = (F * t_k1 + F * C * t_k2 + S * t_sg) / cores
        values.append(F * S / total)
        splits.append((total, F * t_k1 / cores, F * C * t_k2 / cores, S * t_sg / cores))
    ncol, nrow, kind = per_repeat[0][0][3], per_repeat[0][0][4], per_repeat[0][0][5]
    what = (
        "reference SwiftlyCore + api_helper (numpy backend, imported unchanged from /root/reference/src)"
        if kind == "reference" else "oracle port of the reference's numpy path"
    )
    order = sorted(range(repeats), key=lambda i: values[i])
    mid = order[repeats // 2]  # median of the repeats
    total, s1, s2, s3 = splits[mid]
    return dict(
        value=values[mid],
        unit="contributions/s",
        cores=cores,
        kind=kind,
        samples=[round(v, 1) for v in values],
        spread=round(max(values) / min(values), 3),
        host_cpus=len(avail),
        sample=(
            f"{what}, complex128, {cores} processes concurrently (each pinned to its own CPU, of {len(avail)} CPUs), each: K1 on a {p['yB_size']}x{ncol} "
            f"column slab of one facet, K2 on {nrow} of {m} rows of one (facet, column), K3-K5 for one subgrid "
            f"with {F} contributions; per-unit times averaged over processes and extrapolated linearly to {F} facets "
            f"x {C} columns x {S} subgrids spread over {cores} cores "
            f"(K1 {s1:.1f} s + K2 {s2:.1f} s + K3-5 {s3:.1f} s); MEDIAN of {repeats} repeats "
            f"({', '.join(f'{v:.0f}' for v in values)} contributions/s); wall time of all repeats {wall:.1f} s"
        ),
        extrapolated_seconds=total,
        # (r4 advice) what the sample was, so that values stay comparable across rounds: method 5 (r6) = method 4 on at most 32
        # pinned processes; method 4 (r5) = method 2 + an
        # untimed warm-up task per worker + one untimed repeat of the sample; method 2 (r4) = slabs of 2e6 / 4e6 elements and the median of three repeats;
        # method 1 (r1-r3) = 4e6 / 8e6 elements, one repeat
        method=dict(version=5, processes="min(32, CPUs), one per CPU, pinned, spread evenly over the CPUs of the host "
                                         "(methods 1-4: one process per CPU, 256 on the GPU box)", warmup="per worker: process start + PSWF tables + first padded-length transform, then ONE "
                                      "untimed repeat of the whole sample under full contention, before the timed repeats", k1_slab_elements=int(p["yB_size"]) * int(ncol), k2_slab_elements=int(nrow) * int(p["yB_size"]),
                    k1_slab_columns=int(ncol), k2_slab_rows=int(nrow), repeats=repeats, statistic="median"),
    )


# --------------------------------------------------------------------------- measured traffic (PMC passes)
PMC_FILE = os.path.join(ROOT, "profiles", "r6_pmc_kernels.json")


def _pmc_record(workload):
    """Per-kernel counter summary of THIS round's build for the workload (profiles/r6_pmc_kernels.json, written by
    tools/pmc_kernels.py from a rocprofv3 kernel trace and separate --pmc FETCH_SIZE / WRITE_SIZE passes), or None."""
    try:
        with open(PMC_FILE, encoding="utf-8") as fh:
            return json.load(fh).get(workload)
    except (OSError, ValueError):
        return None


def measured_traffic(workload):
    """HBM bytes per launch of the roofline kernel (K1) from the committed PMC summary, or (None, None)."""
    rec = _pmc_record(workload)
    if not rec or "K1" not in rec.get("kernels", {}):
        return None, None
    return rec["kernels"]["K1"]["counter_bytes_per_launch"], rec.get("note")


def traffic_build_state(workload):
    """Is the committed counter summary from THIS build?  {"state": "match" | "stale" | "unknown", "recorded": ...,
    "running": ...}: the summary records the source hash compiled into the library it was collected on
    (swiftly_hip_build_id), the hash of that .so and the git commit; "stale" = the running library was built from other
    kernel sources, i.e. `traffic` / `kernels` describe an older build."""
    from ska_sdp_exec_swiftly_amd import _lib  # pylint: disable=import-outside-toplevel

    rec = _pmc_record(workload)
    running = _lib.build_info()
    recorded = (rec or {}).get("build")
    if not recorded:
        state = "unknown"
    else:
        state = "match" if recorded.get("src_hash") == running["src_hash"] else "stale"
    return dict(state=state, recorded=recorded, running=running)


def kernel_table(workload, F, C, parts):
    """Per-kernel table of the forward pass: algorithmic bytes beside the counter bytes and the
    launch durations of the committed rocprofv3 summary of this build; both fractions are of the 8 TB/s HBM3E peak.
    `frac_algorithmic` is the effective figure (compulsory bytes / time), `frac_sustained` what the kernel really
    moves per second (FETCH_SIZE x 2 + WRITE_SIZE) -- a kernel that prunes its output has sustained < algorithmic,
    one that re-reads or writes scratch has sustained > algorithmic."""
    rec = _pmc_record(workload)
    if not rec:
        return None
    k = rec["kernels"]
    rows = []
    spec = [
        ("K1", ["K1"], F, parts["K1"], "prepare_facet along the contiguous axis, band-pruned store"),
        ("K2", ["K2a", "K2b"], C, parts["K2"], "window gather + strided-axis prepare_facet per wave (four-step: pass A + pass B)"),
        ("K3", ["K3"], C, parts["K3"], "transform_contributions (extract + add_to_subgrid axis 0)"),
        ("K4b+K5a", ["K4b5a"], C, parts["K4"], "sum_finish_facets (facet sum + add_to_subgrid / finish_subgrid axis 1)"),
        ("K5b", ["K5b"], C, parts["K5"], "finish_subgrid axis 0"),
    ]
    for stage, ids, per_pass, alg, what in spec:
        if not all(i in k for i in ids):
            continue
        us = sum(k[i]["avg_us"] for i in ids) * per_pass
        cnt = sum(k[i]["counter_bytes_per_launch"] for i in ids) * per_pass
        rows.append(dict(
            stage=stage, what=what, launches_per_pass=per_pass * len(ids),
            avg_us=[k[i]["avg_us"] for i in ids],
            us_per_pass=round(us, 1),
            algorithmic_bytes_per_pass=int(alg),
            counter_bytes_per_pass=int(cnt),
            fetch_bytes_per_pass=int(sum(k[i]["fetch_bytes_per_launch"] for i in ids) * per_pass),
            write_bytes_per_pass=int(sum(k[i]["write_bytes_per_launch"] for i in ids) * per_pass),
            frac_algorithmic=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            frac_sustained=round(cnt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        ))
    if not rows:
        return None
    tot_us = sum(r["us_per_pass"] for r in rows)
    return dict(
        source="profiles/r6_pmc_kernels.json: " + rec.get("note", ""),
        rows=rows,
        sum_us_per_pass=round(tot_us, 1),
        counter_bytes_per_pass=int(sum(r["counter_bytes_per_pass"] for r in rows)),
        frac_sustained_whole_pass=round(sum(r["counter_bytes_per_pass"] for r in rows) / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
    )


def backward_kernel_table(workload, F, C, parts):
    """The same table for the subgrid -> facet direction (band schedule): every stage is the mirror
    of a forward stage and moves the mirrored bytes, so the algorithmic figures are the forward model's
    (B8 finish_facet <-> K1, B5 gather-sum four-step <-> K2, B4 m-point column pass <-> K3, B2 split_prepare_facets <-> K4,
    B1 prepare_subgrid axis 0 <-> K5).  Counter bytes and durations: profiles/r6_pmc_kernels.json["<workload>:backward"]."""
    rec = _pmc_record(workload + ":backward")
    if not rec:
        return None
    k = rec["kernels"]
    spec = [
        ("B8", ["B8"], F, parts["K1"], "finish_facet along the contiguous axis from the band accumulator (crop x Fb x mask1 store)"),
        ("B5-7", ["B5a", "B5b"], C, parts["K2"], "gather-sum four-step per wave: add_to_facet axis 0 in the load, strided-axis finish_facet, column scatter-add"),
        ("B4", ["B4"], C, parts["K3"], "in-place m-point column pass (extract_from_subgrid axis 0)"),
        ("B2-3", ["B2"], C, parts["K4"], "split_prepare_facets (prepare_subgrid axis 1 + extract_from_subgrid axis 1 per facet)"),
        ("B1", ["B1"], C, parts["K5"], "prepare_subgrid axis 0"),
    ]
    rows = []
    for stage, ids, per_pass, alg, what in spec:
        if not all(i in k for i in ids):
            continue
        us = sum(k[i]["avg_us"] for i in ids) * per_pass
        cnt = sum(k[i]["counter_bytes_per_launch"] for i in ids) * per_pass
        rows.append(dict(
            stage=stage, what=what, launches_per_pass=per_pass * len(ids), avg_us=[k[i]["avg_us"] for i in ids],
            us_per_pass=round(us, 1), algorithmic_bytes_per_pass=int(alg), counter_bytes_per_pass=int(cnt),
            fetch_bytes_per_pass=int(sum(k[i]["fetch_bytes_per_launch"] for i in ids) * per_pass),
            write_bytes_per_pass=int(sum(k[i]["write_bytes_per_launch"] for i in ids) * per_pass),
            frac_algorithmic=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            frac_sustained=round(cnt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        ))
    if not rows:
        return None
    tot_us = sum(r["us_per_pass"] for r in rows)
    running = traffic_build_state(workload + ":backward")
    return dict(
        source="profiles/r6_pmc_kernels.json: " + rec.get("note", ""),
        build_state=running["state"],
        rows=rows,
        sum_us_per_pass=round(tot_us, 1),
        counter_bytes_per_pass=int(sum(r["counter_bytes_per_pass"] for r in rows)),
        frac_sustained_whole_pass=round(sum(r["counter_bytes_per_pass"] for r in rows) / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
    )


def quick_forward(torch, sw, sw_api, name, passes=5, warmups=2):
    """Forward pass of another BASELINE configuration beside the headline line (`other_workloads`, r4 review): the same
    objects as the default workload's timed region -- fresh planned SwiftlyForward per pass, K1 of every facet, every
    wave -- `passes` passes, each timed on its own, after `warmups` untimed ones (r6: the MEDIAN is reported with every
    pass in `step_ms_each`; r5 timed three passes behind one warm-up as a block and one configuration ranged 151 - 203 ms
    between boxes -- first passes of a workload pay the caching allocator's block splitting and the first-use tables),
    parity of the subgrids those objects produce against the oracle.  Never part of `value`."""
    from oracle import separable as sep  # data recipe shared with the checker

    wl = WORKLOADS[name]
    p = wl["params"]
    cfg = sw.SwiftlyConfig(backend="hip", **p)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    cap = wl.get("max_facets_per_rank")
    if cap is not None:
        facet_cfgs = facet_cfgs[:cap]
    sg_cfgs = select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
    wave_axis = sw_api.preferred_wave_axis(cfg, torch.complex64, n_facets=len(facet_cfgs))
    key = (lambda c: c.off1) if wave_axis == 1 else (lambda c: c.off0)
    waves = {}
    for i, c in enumerate(sg_cfgs):
        waves.setdefault(key(c), []).append(i)
    F, S, C = len(facet_cfgs), len(sg_cfgs), len(waves)
    vectors = [sep.facet_vectors(1234 + j, p["yB_size"], rank=2) for j in range(F)]
    data = [separable_facet(torch, vectors[j], facet_cfgs[j]) for j in range(F)]
    picks = sep.pick_subgrids(sg_cfgs, 3)

    def one_pass(keep=None):
        fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, data)), lru_forward=1, subgrid_configs=sg_cfgs, wave_axis=wave_axis)
        fwd.prepare_all_facets()
        for widx in waves.values():
            res = fwd.get_wave([sg_cfgs[i] for i in widx])
            if keep is not None:
                for k, i in enumerate(widx):
                    if i in picks:
                        keep[i] = res[k].cpu().numpy()

    warm_each, fwd_each = [], []
    for rep in range(warmups + passes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_pass()
        torch.cuda.synchronize()
        (warm_each if rep < warmups else fwd_each).append(1e3 * (time.perf_counter() - t0))
    ms = sorted(fwd_each)[len(fwd_each) // 2]
    kept = {}
    one_pass(kept)
    torch.cuda.synchronize()
    par = verify_subgrids(p, facet_cfgs, vectors, sg_cfgs, kept, tol=wl.get("parity_tol"))
    total_bytes, _parts = algorithmic_bytes(p, F, S, C)
    roundtrip = None
    baxis = 1 if cfg.core.supports_backward_band(torch.complex64) else 0
    if wl.get("roundtrip") and wave_axis == baxis:
        # BASELINE config 5 is a round trip: the streaming form of the workload's own run (every forward wave folded
        # straight into the backward accumulators), one warm-up + two timed passes
        def roundtrip_pass():
            fwd = sw.SwiftlyForward(cfg, list(zip(facet_cfgs, data)), lru_forward=1, subgrid_configs=sg_cfgs, wave_axis=wave_axis)
            fwd.prepare_all_facets()
            bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=1, subgrid_configs=sg_cfgs, wave_axis=baxis)
            for widx in waves.values():
                wave = [sg_cfgs[i] for i in widx]
                res = fwd.get_wave(wave)
                bwd.add_new_subgrid_tasks(wave, [res[k] for k in range(len(wave))])
            del fwd
            return bwd.finish()

        each = []
        for rep in range(3):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            out = roundtrip_pass()
            torch.cuda.synchronize()
            if rep:
                each.append(1e3 * (time.perf_counter() - tb))
            finite = all(bool(torch.isfinite(torch.view_as_real(o)).all()) for o in out)
            del out
        rt = sum(each) / len(each)
        roundtrip = dict(ms_per_pass=round(rt, 3), each_ms=[round(t, 2) for t in each], finite=finite,
                         contributions_per_s_both_directions=round(2 * F * S / (rt * 1e-3), 1))
    del data
    torch.cuda.empty_cache()
    return dict(
        roundtrip=roundtrip,
        workload=wl["name"], facets=F, subgrids=S, subgrid_columns=C, wave_axis=wave_axis, passes=passes,
        ms_per_step=round(ms, 3), statistic="median", step_ms_each=[round(t, 2) for t in fwd_each],
        warmup_ms=[round(t, 2) for t in warm_each], contributions_per_s=round(F * S / (ms * 1e-3), 1),
        hbm_algorithmic_frac_of_peak=round(total_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        parity={k: par[k] for k in ("rel_rmse", "subgrids", "tol_rel_rmse", "ok")},
    )


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio when a process group is created; with stdout redirected it sits in
    the C buffer until exit, i.e. AFTER the JSON line.  Flushing it as soon as it exists keeps the JSON line last."""
    import ctypes  # pylint: disable=import-outside-toplevel

    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="64k-sparse", choices=sorted(WORKLOADS))  # default: BASELINE.json's metric config
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="group", choices=["group", "wave"],
                    help="multi-GPU: 'group' (default) = every wave's subgrids are finished by ONE rank and the waves are "
                         "exchanged in groups of n_gpus waves with distinct owners (one balanced all-to-all per group); "
                         "'wave' = the subgrids of every wave dealt out round-robin, one all-to-all per wave")
    ap.add_argument("--axis1-first", nargs="?", const="fused", default=None, choices=["fused", "rows"],
                    help="time the axis-1-first forward pipeline (float32 arithmetic at ~5x smaller error): 'fused' = "
                         "SwiftlyConfig(axis1_first=True), the contiguous axis finished in the epilogue of K1; 'rows' = "
                         "axis1_first='rows', by a row pass per wave")
    ap.add_argument("--column-precision", type=int, default=32, choices=[32, 64],
                    help="arithmetic of the column passes K2 / K3 (complex64 data): 32 = float32 (default, the timed "
                         "configuration of every round), 64 = float64 butterflies (3.7x smaller error, 1.5x the time)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle parity check of the timed objects")
    ap.add_argument("--verify", action="store_true", help="(default) kept for explicitness")
    ap.add_argument("--no-backward", action="store_true",
                    help="skip the subgrid -> facet leg (reported beside the headline metric, outside its timed region)")
    ap.add_argument("--rccl-dry", action="store_true",
                    help="1 GPU only, unmeasured-on-links rehearsal of the multi-GPU path: a one-rank RCCL process group is "
                         "created and the pass runs through DistributedForward with the REAL all_to_all_single calls (split "
                         "sizes of one rank, the real send / receive buffers, RCCL's stream against the compute stream); "
                         "the line says so in `config.parallelism` and is not a scaling measurement")
    ap.add_argument("--strict-others", action="store_true",
                    help="exit non-zero when one of the secondary workloads of the default run errors or fails its parity")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short forward runs of BASELINE configs 2, 3 and 5 (`other_workloads`) that follow the default "
                         "workload's measurement")
    ap.add_argument("--wave-axis", type=int, default=None, choices=[0, 1],
                    help="force the forward pipeline: 0 = strided axis first (waves by off0), 1 = contiguous axis first")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one rank per GPU ourselves (one process per GPU over RCCL)
        cmd = [
            sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__),
        ] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

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
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a mislabelled number")
    if torch.cuda.device_count() < world and os.environ.get("SWIFTLY_BENCH_OVERSUBSCRIBE") != "1":
        raise SystemExit(f"bench.py: {world} ranks requested but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if args.rccl_dry:
        if world != 1:
            raise SystemExit("bench.py: --rccl-dry is a one-rank rehearsal (use it with --gpus 1)")
        from ska_sdp_exec_swiftly_amd import distributed as sw_dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        _flush_c_stdio()
        sw_dist.FORCE_COLLECTIVE = True  # one-rank exchanges go through RCCL instead of being short-circuited
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; SWIFTLY_BENCH_BACKEND=gloo (+ SWIFTLY_BENCH_OVERSUBSCRIBE=1) runs several ranks on ONE
        # GPU with a host-staged exchange -- a functional check of this launcher path, not a measurement
        torch.distributed.init_process_group(os.environ.get("SWIFTLY_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        _flush_c_stdio()

    wl = WORKLOADS[args.workload]
    p = wl["params"]
    cfg = sw.SwiftlyConfig(backend="hip", column_precision=args.column_precision, axis1_first={None: False, "rows": "rows", "fused": True}[args.axis1_first], **p)
    all_facet_cfgs = sw.make_full_facet_cover(cfg)
    # a rank holds at most `max_facets_per_rank` facets (HBM capacity): with too few ranks only the first cap * world
    # facets of the cover take part -- a stated subset; contributions are counted for those only
    cap = wl.get("max_facets_per_rank")
    n_active = len(all_facet_cfgs) if cap is None else min(len(all_facet_cfgs), cap * world)
    facet_cfgs = all_facet_cfgs[:n_active]
    all_sg_cfgs = sw.make_full_subgrid_cover(cfg)
    sg_cfgs = select_subgrids(all_sg_cfgs, p["N"], p["xA_size"], wl["sparse_radius"])
    force_dist = os.environ.get("SWIFTLY_BENCH_FORCE_DIST") == "1"  # exercise the multi-GPU code path on 1 GPU
    single = world == 1 and not force_dist and not args.rccl_dry
    wave_axis = sw_api.preferred_wave_axis(cfg, torch.complex64, n_facets=len(facet_cfgs))
    if args.wave_axis is not None:
        wave_axis = args.wave_axis
    key = (lambda c: c.off1) if wave_axis == 1 else (lambda c: c.off0)
    waves = {}
    for i, c in enumerate(sg_cfgs):
        waves.setdefault(key(c), []).append(i)
    wave_idx = list(waves.values())
    waves = [[sg_cfgs[i] for i in w] for w in wave_idx]
    F, S, C = len(facet_cfgs), len(sg_cfgs), len(waves)

    # synthetic dense facets: separable (rank 2) random vectors on the 1/8 grid, seed 1234 + facet index,
    # times the cover masks; built on the device
    from oracle import separable as sep  # data recipe shared with the checker

    local = [j for j in range(F) if j % world == rank]
    vectors = [sep.facet_vectors(1234 + j, p["yB_size"], rank=2) for j in range(F)]
    facet_data = [None] * F
    # facets that do not fill a round of ranks are worked on by all ranks (distributed.FacetSharding, band pipeline):
    # every rank then needs its rows of them -- the synthetic facet is simply built everywhere
    shared = list(range((F // world) * world, F)) if world > 1 and wave_axis == 1 else []
    for j in sorted(set(local) | set(shared)):
        facet_data[j] = separable_facet(torch, vectors[j], facet_cfgs[j])

    picks = sep.pick_subgrids(sg_cfgs, 6) if not args.no_verify else []
    kept = {}

    if single:

        def factory():
            return sw.SwiftlyForward(
                cfg, [(facet_cfgs[j], facet_data[j]) for j in range(F)], lru_forward=1, subgrid_configs=sg_cfgs,
                wave_axis=wave_axis,
            )

        def one_pass(timer=None, keep=None):
            fwd = factory()
            if timer is not None:
                t0 = timer.start()
            fwd.prepare_all_facets(timer)
            if timer is not None:
                timer.stop("K1_total", t0)
            count = 0
            for widx, wave in zip(wave_idx, waves):
                res = fwd.get_wave(wave, timer)
                count += len(wave)
                if keep is not None:
                    for k, i in enumerate(widx):
                        if i in picks:
                            keep[i] = res[k].cpu().numpy()
            return count

    else:

        grouped = args.exchange == "group" and wave_axis == 1 and (world > 1 or args.rccl_dry)
        index_of = {(c.off0, c.off1): i for i, c in enumerate(sg_cfgs)}

        def one_pass(timer=None, keep=None):  # pylint: disable=unused-argument
            dfw = DistributedForward(
                cfg, facet_cfgs, facet_data, lru_forward=1, subgrid_configs=sg_cfgs, wave_axis=wave_axis,
                dtype=torch.complex64, whole_waves=grouped,
            )
            dfw.prepare_all_facets()
            if grouped:
                # whole waves per rank, one all-to-all per group of `world` waves; the exchange of group g runs while
                # the facet side of group g+1 is computed
                by_key = {dfw.wave_key(w): w for w in waves}
                n = 0
                pending = None

                def finish_group(h):
                    sgs, res = dfw.finish_group(h)
                    if sgs is None:
                        return 0
                    if keep is not None:
                        for k, c in enumerate(sgs):
                            if index_of[(c.off0, c.off1)] in picks:
                                keep[index_of[(c.off0, c.off1)]] = res[k].cpu().numpy()
                    return len(sgs)

                for group in dfw.sharding.wave_groups:
                    handle = dfw.start_group([by_key[k] for k in group])
                    if pending is not None:
                        n += finish_group(pending)
                    pending = handle
                n += finish_group(pending)
                return n
            # software pipeline: the all-to-all of wave w runs while wave w+1's column/extract kernels do
            n = 0
            pending = None

            def finish(h, widx):
                mine, res = dfw.finish_wave(h)
                if keep is not None and res is not None:
                    for k, pos in enumerate(mine):
                        if widx[pos] in picks:
                            keep[widx[pos]] = res[k].cpu().numpy()
                return len(mine)

            for widx, wave in zip(wave_idx, waves):
                handle = dfw.start_wave(wave)
                if pending is not None:
                    n += finish(*pending)
                pending = (handle, widx)
            n += finish(*pending)
            return n

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    fence()
    # (r5) one HIP event behind every timed pass (no synchronisation: the timed region is unchanged): the per-step times
    # say whether a run's average hides an outlier -- the first process on a fresh box has shown single passes of + 10 ms
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_events[0].record()
    for i in range(args.steps):
        one_pass()
        step_events[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms_each = [round(step_events[i].elapsed_time(step_events[i + 1]), 3) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)

    # parity of the timed objects (same factory, same facets, same waves), outside the timed region
    parity = None
    if picks:
        one_pass(keep=kept)
        fence()
        if world > 1:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, kept)
            kept = {k: v for d in gathered for k, v in d.items()}
        if rank == 0:
            parity = verify_subgrids(p, facet_cfgs, vectors, sg_cfgs, kept,
                                     tol=AXIS1_FIRST_PARITY_TOL if args.axis1_first else wl.get("parity_tol"))

    # (r6) the axis-1-first pipeline beside the timed default order: same objects, same facets, float32 arithmetic -- the
    # contiguous axis finished (m-point transform x Fn per wave window) BEFORE K2 / K3, which then see ONE facet window
    accurate = None
    if single and picks and args.column_precision == 32 and not args.axis1_first and rank == 0:
        def axis1_leg(mode, tol):
            cfg.core.axis1_first = mode
            try:
                one_pass()
                fence()
                each = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    one_pass()
                    fence()
                    each.append(1e3 * (time.perf_counter() - t0))
                kept_a = {}
                one_pass(keep=kept_a)
                fence()
                a_par = verify_subgrids(p, facet_cfgs, vectors, sg_cfgs, kept_a, tol=tol)
                med = sorted(each)[1]
                return dict(ms_per_step=round(med, 3), each_ms=[round(t, 2) for t in each],
                            contributions_per_s=round(F * S / (med * 1e-3), 1),
                            hbm_algorithmic_frac_of_peak=round(algorithmic_bytes(p, F, S, C)[0] / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            parity={k: a_par[k] for k in ("rel_rmse", "rel_rmse_each", "max_abs_over_rms", "tol_rel_rmse", "ok")})
            except NotImplementedError as err:
                return dict(error=str(err))
            finally:
                cfg.core.axis1_first = False

        accurate = dict(
            mode="axis1_first",
            what="SwiftlyConfig(axis1_first=True): the contiguous-axis half of add_to_subgrid (window gather, m-point "
                 "transform, Fn) runs in the epilogue of K1 -- one persistent workgroup per CU owns whole rows, stages the band of "
                 "a row in LDS and stores the finished window rows of every planned wave (swiftly_hip_prepare_facet_window_rows; "
                 "no band buffer, no row pass per wave); K2 / K3 unchanged on singly windowed data, sum_finish_facets without "
                 "its m-point transforms; float32 arithmetic throughout; not part of `value`",
            **axis1_leg(True, AXIS1_FIRST_PARITY_TOL),
        )
        accurate["row_pass_per_wave"] = dict(
            what="SwiftlyConfig(axis1_first='rows'): the same order on top of the default K1 -- per wave, the rows of the band "
                 "buffers go through a pass of their own (swiftly_hip_finish_axis1_rows) before K2; what configurations "
                 "without the fused K1 (other sizes, no plan, cooperative facets of the multi-GPU pass) run",
            **axis1_leg("rows", AXIS1_FIRST_PARITY_TOL),
        )

    # the float64-arithmetic column passes (column_precision = 64) beside the timed float32 configuration: same objects,
    # same facets, three passes + the same parity check, outside the timed region (1 GPU, default precision only)
    high_precision = None
    if single and picks and args.column_precision == 32 and rank == 0:
        cfg.core.column_precision = 64
        try:
            one_pass()
            fence()
            t0 = time.perf_counter()
            for _ in range(3):
                one_pass()
            fence()
            hp_ms = 1e3 * (time.perf_counter() - t0) / 3
            kept64 = {}
            one_pass(keep=kept64)
            fence()
            hp_par = verify_subgrids(p, facet_cfgs, vectors, sg_cfgs, kept64, tol=HIGH_PRECISION_PARITY_TOL)
            high_precision = dict(
                column_precision=64, ms_per_step=round(hp_ms, 3),
                what="K2 (both four-step passes) and K3 with float64 windows / butterflies / exchanges / twiddles between "
                     "complex64 loads and stores (swiftly_hip_set_column_precision); not part of `value`",
                parity={k: hp_par[k] for k in ("rel_rmse", "rel_rmse_each", "max_abs_over_rms", "tol_rel_rmse", "ok")},
            )
        finally:
            cfg.core.column_precision = 32

    # per-stage HIP-event timing (separate instrumented pass, 1 GPU only)
    stages = {}
    roofline = None
    total_bytes, parts = algorithmic_bytes(p, F, S, C)
    if single:
        timer = StageTimer(torch)
        one_pass(timer)
        torch.cuda.synchronize()
        for name, (cnt, ms) in timer.totals().items():
            stages[name] = dict(launch_groups=cnt, total_ms=round(ms, 3), avg_ms=round(ms / cnt, 4))
        # K1's average launch duration: ONE HIP-event pair around the F back-to-back launches of the production
        # call (an event pair per launch adds ~0.2 ms of drain/flush to each 2 ms kernel and would not agree with
        # the rocprofv3 kernel durations in profiles/)
        k1_runs = []
        for _ in range(3):
            fwd = factory()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fwd.prepare_all_facets()
            e1.record()
            torch.cuda.synchronize()
            k1_runs.append(e0.elapsed_time(e1) / F)
            del fwd
        k1 = dict(avg_ms=round(sum(k1_runs[1:]) / len(k1_runs[1:]), 4))  # first run: allocator warm-up
        stages["K1_back_to_back"] = dict(launch_groups=F, avg_ms=k1["avg_ms"], runs_avg_ms=[round(t, 4) for t in k1_runs])
        k1_bytes = parts["K1"] / F  # per facet = per launch group
        achieved = k1_bytes / (k1["avg_ms"] * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic(args.workload)
        roofline = dict(
            # (what the kernel really moves per second leads: `frac` below is the EFFECTIVE figure -- algorithmic bytes
            # over time -- and the band store keeps only 35 % of the outputs)
            sustained_frac=round(traffic / (k1["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            kernel=sw_api.K1_DESCRIPTION[wave_axis],
            bound="hbm",
            achieved=round(achieved, 1),
            peak=HBM_PEAK_GBS,
            unit="GB/s",
            frac=round(achieved / HBM_PEAK_GBS, 4),
            traffic=traffic,
            # what the kernel really sustains: counter bytes per launch / the same live launch duration (the band
            # store keeps 35 % of the outputs, so this is below `achieved`)
            sustained=round(traffic / (k1["avg_ms"] * 1e-3) / 1e9, 1) if traffic else None,
            traffic_note=traffic_note
            or "null: no rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE summary of this build is committed for this workload",
            traffic_build=traffic_build_state(args.workload),
            algorithmic_bytes_per_launch=k1_bytes,
            avg_launch_ms=k1["avg_ms"],
        )

    if roofline is None and local:
        # multi-GPU: time K1 of one local facet on this rank (outside the timed region) for the roofline object
        core = cfg.core
        j0 = local[0]
        if wave_axis == 1:
            band = core.band_for_offsets([sg.off1 for sg in sg_cfgs])
            buf = core.prepare_facet_band(facet_data[j0], facet_cfgs[j0].off1, band)
            k1 = lambda: core.prepare_facet_band(facet_data[j0], facet_cfgs[j0].off1, band, out=buf)  # noqa: E731
        else:
            rowmap, n_rows = core.subgrid_column_rows([sg.off0 for sg in sg_cfgs])
            buf = core.prepare_facet_rows(facet_data[j0], facet_cfgs[j0].off0, rowmap, n_rows, fold_axis1_window=True)
            k1 = lambda: core.prepare_facet_rows(  # noqa: E731
                facet_data[j0], facet_cfgs[j0].off0, rowmap, n_rows, out=buf, fold_axis1_window=True
            )
        k1()  # warm-up: first touch of the fresh output buffer
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            k1()
        e1.record()
        torch.cuda.synchronize()
        k1_ms = e0.elapsed_time(e1) / 5
        k1_bytes = parts["K1"] / F
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
        roofline = dict(
            kernel=sw_api.K1_DESCRIPTION[wave_axis] + " (rank 0, one facet)",
            bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
            algorithmic_bytes_per_launch=k1_bytes, avg_launch_ms=round(k1_ms, 4),
        )
        del buf

    # subgrid -> facet direction: reported beside the headline metric, never part of `value`
    backward = None
    roundtrip = None
    # band schedule (waves by off1) wherever its kernels exist, else the reference schedule (waves by off0)
    baxis = 1 if cfg.core.supports_backward_band(torch.complex64) else 0
    bkey = (lambda c: c.off1) if baxis == 1 else (lambda c: c.off0)
    produced_bytes = S * p["xA_size"] ** 2 * 8
    streaming = bool(wl.get("roundtrip")) or produced_bytes > 24e9

    def backward_parity(plan):
        """element-wise parity of the band schedule on SEPARABLE subgrids (rank-1 outer products on the 1/8 grid times
        the subgrid masks: exact in float32) of the plan ``plan``: sampled rows of every facet against the oracle"""
        vec = [sep.subgrid_vectors(4321 + i, c.size, rank=1) for i, c in enumerate(plan)]
        bw = {}
        for i, c in enumerate(plan):
            bw.setdefault(bkey(c), []).append(i)
        bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=1, subgrid_configs=plan, wave_axis=baxis)
        for idx in bw.values():  # one wave of subgrids alive at a time
            bwd.add_new_subgrid_tasks([plan[i] for i in idx], [separable_facet(torch, vec[i], plan[i]) for i in idx])
        out = bwd.finish()
        torch.cuda.synchronize()
        par = verify_facets(p, facet_cfgs, plan, vec, out, rows_per_facet=16, tol=wl.get("parity_tol"))
        del out
        return par

    if single and not args.no_backward and not streaming:
        fwd = factory()
        fwd.prepare_all_facets()
        produced = [fwd.get_wave(wave).clone() for wave in waves]
        del fwd
        torch.cuda.synchronize()
        lookup = {(c.off0, c.off1): data[k] for wave, data in zip(waves, produced) for k, c in enumerate(wave)}
        bwaves = {}
        for c in sg_cfgs:
            bwaves.setdefault(bkey(c), []).append(c)
        bwaves = list(bwaves.values())

        def backward_pass():
            bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=1, subgrid_configs=sg_cfgs, wave_axis=baxis)
            for wave in bwaves:
                bwd.add_new_subgrid_tasks(wave, [lookup[(c.off0, c.off1)] for c in wave])
            return bwd.finish()

        try:
            # two untimed passes (r4 review: one was not enough -- the first timed pass of the driver's run took 162 ms
            # against 49 ms for the others: the caching allocator re-splitting its blocks after the float64 leg above),
            # then the MEDIAN of five
            warm = []
            out = None
            for _ in range(2):
                del out
                tb = time.perf_counter()
                out = backward_pass()
                torch.cuda.synchronize()
                warm.append(1e3 * (time.perf_counter() - tb))
            nb = 5
            each = []
            for _ in range(nb):
                del out  # the previous pass's facets go back to the allocator before the next pass asks for its own
                tb = time.perf_counter()
                out = backward_pass()
                torch.cuda.synchronize()
                each.append(1e3 * (time.perf_counter() - tb))
            b_ms = sorted(each)[nb // 2]
            finite = all(bool(torch.isfinite(torch.view_as_real(o)).all()) for o in out)
            backward = dict(
                ms_per_pass=round(b_ms, 3), statistic="median", passes=nb, each_ms=[round(t, 2) for t in each],
                warmup_ms=[round(t, 2) for t in warm],
                ratio_to_forward=round(b_ms / ms_per_step, 3),
                hbm_algorithmic_frac_of_peak=round(total_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                kernels=backward_kernel_table(args.workload, F, C, parts),
                schedule="band accumulators, waves by off1" if baxis == 1 else "reference schedule, waves by off0",
                subgrids=S, facets=F, finite=finite,
            )
            del out, produced, lookup
            produced = lookup = None
            bk = backward.get("kernels")
            if bk:
                # the dominant kernel group of the backward pass against the HBM peak (durations and counter bytes of the
                # committed summary of this build, like `kernels`; not a live measurement)
                top = max(bk["rows"], key=lambda r: r["us_per_pass"])
                secs = top["us_per_pass"] * 1e-6
                backward["roofline"] = dict(
                    kernel=f"{top['stage']}: {top['what']}", bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
                    achieved=round(top["algorithmic_bytes_per_pass"] / secs / 1e9, 1),
                    frac=top["frac_algorithmic"], traffic=top["counter_bytes_per_pass"] // top["launches_per_pass"],
                    sustained=round(top["counter_bytes_per_pass"] / secs / 1e9, 1), sustained_frac=top["frac_sustained"],
                    launches_per_pass=top["launches_per_pass"], source=bk["source"][:60] + "...", build_state=bk["build_state"],
                )
            if picks:
                backward["parity"] = backward_parity(sg_cfgs)
        except (ValueError, NotImplementedError) as err:  # sizes without the band kernels
            backward = dict(skipped=str(err))
        del produced, lookup

    if not args.no_backward and streaming:
        # forward + backward ROUND TRIP, streaming (facet -> subgrid wave -> folded straight back into the facet sums,
        # the flow of the reference's scripts/demo_api.py): the finished subgrids of a big cover (20164 x 6.9 MB for
        # N = 131072) are never all alive
        if wave_axis != baxis:
            roundtrip = dict(skipped="forward and backward pipelines use different wave keys for this configuration")
        else:
            from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward

            def roundtrip_pass():
                if single:
                    fwd = factory()
                    fwd.prepare_all_facets()
                    bwd = sw.SwiftlyBackward(cfg, facet_cfgs, lru_backward=1, subgrid_configs=sg_cfgs, wave_axis=baxis)
                    for wave in waves:
                        res = fwd.get_wave(wave)
                        bwd.add_new_subgrid_tasks(wave, [res[k] for k in range(len(wave))])
                    del fwd
                    return bwd.finish()
                dfw = DistributedForward(cfg, facet_cfgs, facet_data, lru_forward=1, subgrid_configs=sg_cfgs,
                                         wave_axis=wave_axis, dtype=torch.complex64, whole_waves=grouped)
                dbw = DistributedBackward(cfg, facet_cfgs, wave_axis=baxis, subgrid_configs=sg_cfgs, dtype=torch.complex64,
                                          whole_waves=grouped)
                dfw.prepare_all_facets()
                pend_f = pend_b = None
                if grouped:
                    by_key = {dfw.wave_key(w): w for w in waves}
                    for group in dfw.sharding.wave_groups + [None]:
                        gw = [by_key[k] for k in group] if group is not None else None
                        hf = (gw, dfw.start_group(gw)) if gw is not None else None
                        if pend_f is not None:
                            sgs, res = dfw.finish_group(pend_f[1])
                            hb = dbw.start_group(pend_f[0], [res[k] for k in range(len(sgs))] if sgs is not None else [])
                            if pend_b is not None:
                                dbw.finish_group(pend_b)
                            pend_b = hb
                        pend_f = hf
                    dbw.finish_group(pend_b)
                    del dfw
                    return dbw.finish()[1]
                for wave in waves + [None]:
                    hf = (wave, dfw.start_wave(wave)) if wave is not None else None
                    if pend_f is not None:
                        mine, res = dfw.finish_wave(pend_f[1])
                        hb = dbw.start_wave(pend_f[0], [res[k] for k in range(len(mine))] if res is not None else [])
                        if pend_b is not None:
                            dbw.finish_wave(pend_b)
                        pend_b = hb
                    pend_f = hf
                dbw.finish_wave(pend_b)
                del dfw
                return dbw.finish()[1]

            out = roundtrip_pass()
            fence()
            del out
            each = []
            for _ in range(2):
                fence()
                tb = time.perf_counter()
                out = roundtrip_pass()
                fence()
                each.append(1e3 * (time.perf_counter() - tb))
                finite = all(bool(torch.isfinite(torch.view_as_real(o)).all()) for o in out)
                del out
            rt = sum(each) / len(each)
            if world > 1:
                t = torch.tensor([rt], device="cuda", dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                rt = float(t.item())
            roundtrip = dict(
                ms_per_pass=round(rt, 3), passes=len(each), each_ms=[round(t, 2) for t in each],
                forward_ms=round(ms_per_step, 3), backward_ms_by_difference=round(rt - ms_per_step, 3),
                contributions_per_s_both_directions=round(2 * F * S / (rt * 1e-3), 1),
                schedule="streaming: every forward wave is folded straight into the backward band accumulators",
                subgrids=S, facets=F, finite=finite,
            )
            if single and picks:
                plan = select_subgrids(all_sg_cfgs, p["N"], p["xA_size"], wl.get("parity_radius", wl["sparse_radius"]))
                roundtrip["backward_parity"] = backward_parity(plan)
                roundtrip["backward_parity"]["note"] = (
                    f"band schedule on a plan of {len(plan)} of the {len(all_sg_cfgs)} subgrids (the oracle costs "
                    "one length-yN transform per subgrid and facet offset)"
                )

    line = dict(
        metric="facet_to_subgrid_contributions_per_s",
        value=round(F * S / (ms_per_step * 1e-3), 1),
        unit="contributions/s",
        n_gpus=world,
        rccl_ranks=world if (world > 1 or args.rccl_dry) and os.environ.get("SWIFTLY_BENCH_BACKEND", "nccl") == "nccl" else 0,
        steps=args.steps,
        warmup=args.warmup,
        ms_per_step=round(ms_per_step, 3),
        step_ms_each=step_ms_each,  # HIP events behind each timed pass on rank 0's stream (information; `value` uses the wall clock)
        higher_is_better=True,
        scaling="strong",
        vs_baseline=None,
        dtype=f"complex64 (f32 arithmetic, axis-1-first order: {args.axis1_first})" if args.axis1_first else "complex64 (f32 arithmetic)" if args.column_precision == 32 else "complex64 (f64 arithmetic in the column passes K2/K3, f32 elsewhere)",
        data="synthetic",
        config=dict(
            workload=wl["name"], facets=F, facets_total=len(all_facet_cfgs), subgrids=S, subgrid_columns=C,
            contributions=F * S, params=p,
            wave_axis=wave_axis,
            facet_data="separable rank-2 dense random (1/8 grid), seed 1234+j, times cover masks",
            parallelism=(f"facets sharded over {world} rank(s) (facets beyond a full round worked on cooperatively: K1 by "
                         f"row blocks, K2/K3 by wave ranges), contribution all-to-all per "
                         + ("group of waves with distinct owners" if args.exchange == "group" and wave_axis == 1 else "wave"))
            if world > 1 else ("1 GPU, ONE-RANK RCCL REHEARSAL of the multi-GPU path (--rccl-dry): every exchange is a real "
                               "all_to_all_single of one rank; not a scaling measurement" if args.rccl_dry else "1 GPU"),
        ),
        hbm_algorithmic_gbs=round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
        hbm_algorithmic_frac_of_peak=round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4),
        algorithmic_bytes=dict(total=total_bytes, **parts),
        stages=stages,
        roofline=roofline,
        kernels=kernel_table(args.workload, F, C, parts) if world == 1 else None,
        parity=parity,
        accurate=accurate,
        high_precision=high_precision,
        backward=backward,
        roundtrip=roundtrip,
    )
    if line["roofline"] is not None:
        # (r5 review) the object LEADS with the whole pass -- all stages' algorithmic bytes over the timed step, and what
        # the counters say the pass moved over the kernels' own durations -- and carries the dominant kernel (K1, 52 % of
        # the algorithmic bytes) inside: K1's own `frac` credits 3.8 GB per launch of band-pruned output it never writes
        # (`pruned_bytes`), so it is not the state of the pass.
        k1r = line["roofline"]
        ktab = line["kernels"] or {}
        k1r["frac_algorithmic"] = k1r.get("frac")
        k1r["frac_sustained"] = k1r.pop("sustained_frac", None)
        if k1r.get("traffic") and k1r.get("algorithmic_bytes_per_launch"):
            k1r["pruned_bytes"] = int(k1r["algorithmic_bytes_per_launch"] - k1r["traffic"])
        line["roofline"] = dict(
            scope="whole forward pass (all stages); the dominant kernel is inside",
            bound="hbm",
            achieved=line["hbm_algorithmic_gbs"],
            peak=HBM_PEAK_GBS,
            unit="GB/s",
            frac=line["hbm_algorithmic_frac_of_peak"],
            sustained_frac=ktab.get("frac_sustained_whole_pass"),
            traffic=ktab.get("counter_bytes_per_pass"),
            algorithmic_bytes=total_bytes,
            ms_per_step=line["ms_per_step"],
            what="achieved = algorithmic bytes of ALL stages (SURVEY section 8d) / ms_per_step; traffic = FETCH_SIZE x 2 + "
                 "WRITE_SIZE of every kernel of a pass (committed counter summary of this build); sustained_frac = traffic / "
                 "sum of the kernels' own durations / peak",
            traffic_build=k1r.get("traffic_build"),
            dominant_kernel=k1r,
        )
    if len(facet_cfgs) < len(all_facet_cfgs):
        line["scaling"] = "weak (a rank holds at most %d facets: the facet subset grows with the ranks)" % cap
    if single and rank == 0 and args.workload == "64k-sparse" and not args.no_other_workloads and not args.no_verify:
        # BASELINE configs 2, 3 and 5 in the driver's line (r4 review): three timed forward passes each, after the headline
        # measurement and outside its timed region; the default workload's facets go back to the allocator first
        del facet_data[:]
        torch.cuda.empty_cache()
        line["other_workloads"] = {}
        for other in ("8k", "32k-8x8", "128k"):  # BASELINE configs 2, 3 and 5 (5: the two facets one rank of eight holds)
            try:
                line["other_workloads"][other] = quick_forward(torch, sw, sw_api, other)
            except Exception as err:  # pylint: disable=broad-except
                line["other_workloads"][other] = dict(error=f"{type(err).__name__}: {err}")
        line["other_workloads_ok"] = all(not r.get("error") and r.get("parity", {}).get("ok", False)
                                         for r in line["other_workloads"].values())
    if world > 1 or args.rccl_dry:
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()
        _flush_c_stdio()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(p, F, S, C)
            line["cpu_baseline"]["value"] = round(line["cpu_baseline"]["value"], 3)
            line["speedup_vs_cpu_host"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if parity is not None and not parity["ok"]:
        raise SystemExit(f"bench.py: PARITY FAILURE rel_rmse={parity['rel_rmse']:.3e} >= {parity['tol_rel_rmse']}")
    # (r5 advisor) the secondary workloads report their failures INSIDE the line (`other_workloads[..].error` / `.parity.ok`,
    # `other_workloads_ok`); the exit status of the default run belongs to the headline measurement.  --strict-others turns
    # them into a failure again (the builder's own sessions).
    if args.strict_others:
        for other, rec in (line.get("other_workloads") or {}).items():
            if rec.get("error") or not rec.get("parity", {}).get("ok", False):
                raise SystemExit(f"bench.py: other_workloads[{other}] failed: {rec.get('error') or rec['parity']}")
    bpar = (backward or {}).get("parity") or (roundtrip or {}).get("backward_parity")
    if bpar is not None and not bpar["ok"]:
        raise SystemExit(f"bench.py: BACKWARD PARITY FAILURE rel_rmse={bpar['rel_rmse']:.3e} >= {bpar['tol_rel_rmse']}")


if __name__ == "__main__":
    main()
