"""
``SwiftlyForward``: the facet -> subgrid direction of the streaming API (reference src/ska_sdp_exec_swiftly/api.py:217-324,
task bodies api_helper.py:73-112, 200-210) on one MI355X, with the reference's schedule (``wave_axis=0``) and the
contiguous-axis-first band pipeline (``wave_axis=1``).
"""
import logging
import os

import numpy

from .core_hip import band_range
from .ingest import _FacetIngest, _mask_table
from .prefetch import WavePrefetch, _knobs
from .tasks import DeviceTask, LRUCache, TaskQueue, _torch, _unwrap

log = logging.getLogger("fourier-logger")


def preferred_wave_axis(swiftly_config, dtype=None, n_facets=None):
    """Which subgrid offset the forward engine should group "waves" by for
    row-major facets: 0 = ``off0`` (the reference's column cache key,
    api.py:300-324; full-facet transform along the strided axis 0 first),
    1 = ``off1`` (full-facet transform along the CONTIGUOUS axis first: one
    kernel instead of a four-step with a facet-sized scratch; the axis order is
    free because the transforms are separable).  1 when the kernels of that
    pipeline exist for the configuration's sizes, dtype and -- when given -- the TOTAL number of facets of the
    cover (the fused subgrid side sums all facets in one kernel, at most ``core.MAX_FUSED_FACETS``)."""
    return 1 if swiftly_config.core.supports_band_pipeline(dtype, n_facets) else 0


K1_DESCRIPTION = {
    0: "K1 prepare_facet(axis=0) per facet = col_pass<n1=128, mapped load> + col_pass<n2=256, mapped store>",
    1: "K1 prepare_facet(axis=1) of all facet rows, band-compacted parity-split store = row_pass_band_kernel (2 workgroups per row)",
}

class SwiftlyForward(WavePrefetch):
    """Facet -> subgrid streaming transform (reference api.py:217-324).

    :param swiftly_config: SwiftlyConfig
    :param facet_tasks: list of ``(FacetConfig, facet_data)``; data may be a
        numpy array or a torch tensor (complex64 or complex128; it is uploaded
        once and stays in HBM)
    :param lru_forward: number of subgrid columns (distinct ``off0``) whose
        prepared facet columns ``NMBF_BF`` are kept
    :param queue_size: bound on unfinished subgrid tasks (reference
        ``TaskQueue``, api.py:466-522): ``get_subgrid_task`` blocks the host
        while that many earlier results are still being computed
    :param subgrid_configs: optional (extension) list of all subgrids that will
        be requested; enables row-compacted ``BF_F`` for sparse subgrid sets
    """

    # pylint: disable=too-many-arguments,too-many-instance-attributes
    def __init__(
        self, swiftly_config, facet_tasks, lru_forward=1, queue_size=20, client=None, subgrid_configs=None,
        wave_axis=None, delayed=False,
    ):
        self.delayed = bool(delayed)  # hand out DeviceTask handles instead of bare device tensors
        facet_tasks = [(cfg, _unwrap(data)) for cfg, data in facet_tasks]
        self.config = swiftly_config
        self.core = swiftly_config.core
        self.facet_tasks = facet_tasks
        self.task_queue = TaskQueue(queue_size)
        self.facet_configs = [cfg for cfg, _ in facet_tasks]
        self.queue_size = queue_size
        self._client = client
        self.lru = LRUCache(lru_forward)
        self.BF_Fs_persist = None
        self._prewindowed = False
        torch = _torch()
        # facet ingestion (host <-> device edge): device tensors are used in place; host (numpy) facets are
        # uploaded on a dedicated copy stream through a small ring of pinned staging buffers, and the compute
        # stream waits per facet only when a kernel first needs it -- the PCIe transfer of facet j+1 overlaps
        # the full-facet transform of facet j
        self._ingest = _FacetIngest(self.core)
        self._facet_info = [self._ingest.add(data) for _, data in facet_tasks]
        self._ingest.prefetch(0)
        dtypes = {info[0] for info in self._facet_info}
        if len(dtypes) > 1:
            raise ValueError("all facets must have the same dtype")
        self.dtype = dtypes.pop() if dtypes else torch.complex64
        if wave_axis is None:
            # default: the reference's schedule (waves keyed by off0) -- unless the caller hands over the plan of
            # subgrids it is going to request AND the contiguous-axis-first kernels exist for this configuration:
            # then that pipeline is used, and requests that cover only part of a planned wave are served from a
            # bounded cache of finished subgrids (below), so that ANY request order stays cheap
            self.wave_axis = 1 if subgrid_configs is not None and self._band_pipeline_ok() else 0
        else:
            self.wave_axis = int(wave_axis)
        if self.wave_axis not in (0, 1):
            raise ValueError("wave_axis must be 0 or 1")
        # optional plan (extension): when the caller knows up front which subgrids it will ask for (sparse
        # covers, scripts/demo_sparse_facet.py style), the facet-sized intermediate only keeps what those read
        self._rowmap, self._n_rows = None, None
        self._plan = None
        if subgrid_configs is not None:
            self._plan = list(subgrid_configs)
            if self.wave_axis == 0:
                self._rowmap, self._n_rows = self.core.subgrid_column_rows([sg.off0 for sg in subgrid_configs])
            self._planned_keys = {int(self._key(sg)) for sg in subgrid_configs}
        self._band = None
        self._wave_rowmaps = {}
        # finished subgrids computed ahead of their request (see get_subgrid_tasks): (off0, off1, size, id) -> tensor
        self._results = {}
        self._result_bytes = 0
        self._result_budget = int(float(os.environ.get("SWIFTLY_RESULT_CACHE_GB", "16")) * 2**30)
        self._plan_waves = None

    def _band_pipeline_ok(self):
        """the contiguous-axis-first pipeline can serve these facets (sizes, dtype, layout, facet count)"""
        torch = _torch()
        sizes = {info[1] for info in self._facet_info}
        return (
            self.dtype == torch.complex64
            and len(sizes) == 1
            and all(info[2] for info in self._facet_info)
            and self.core.supports_band_pipeline(self.dtype, len(self._facet_info))
        )

    # -- stage 1: BF_F = prepare_facet(axis 0), once per facet (api.py:281-298)
    def _prepare_one_facet(self, j):
        """BF_F of facet ``j`` as the streaming classes keep it: (optionally) row-compacted and with the
        axis-1 window of extract_column already applied (it commutes with the axis-0 transform), so the
        column kernel has no window loads; complex128 and unsupported sizes use the plain primitive."""
        cfg, data = self.facet_configs[j], self._ingest.ready(j)
        n_rows = self._n_rows if self._rowmap is not None else self.core.yN_size
        if self._prewindowed or self._rowmap is not None:
            return self.core.prepare_facet_rows(
                data, cfg.off0, self._rowmap, n_rows, fold_axis1_window=self._prewindowed
            )
        return self.core.prepare_facet(data, cfg.off0, axis=0)

    def prepare_all_facets(self, timer=None):
        """Stage 1 for every facet (idempotent).  ``timer`` (optional, bench.py's
        StageTimer) brackets each facet's launch group with HIP events."""
        if self.wave_axis == 1:
            return self._prepare_all_bands(timer)
        if self.BF_Fs_persist is None:
            self._prewindowed = self.dtype == _torch().complex64
            out = []
            for j in range(len(self.facet_configs)):
                t0 = timer.start() if timer is not None else None
                out.append(self._prepare_one_facet(j))
                self._ingest.prefetch(j + 1)
                if timer is not None:
                    timer.stop("K1_full_facet_transform", t0)
            self.BF_Fs_persist = out
        return self.BF_Fs_persist

    def _get_BF_Fs(self):
        return self.prepare_all_facets()

    # -- stage 2: per subgrid column (api.py:300-324)
    def get_NMBF_BFs_off0(self, off0, BF_Fs=None):
        """prepared facet columns for subgrid column ``off0`` (LRU cached)"""
        if self.wave_axis != 0:
            raise ValueError(
                "get_NMBF_BFs_off0 belongs to the reference schedule (wave_axis=0); this SwiftlyForward runs the "
                "contiguous-axis-first pipeline (wave_axis=1, the default with a subgrid plan): construct it with wave_axis=0"
            )
        if BF_Fs is None:
            BF_Fs = self._get_BF_Fs()
        elif BF_Fs is not self.BF_Fs_persist:
            # BF_Fs_persist holds pre-windowed / row-compacted data (see _prepare_one_facet); a plain
            # prepare_facet(axis=0) result would silently miss the axis-1 window
            raise ValueError("get_NMBF_BFs_off0 only accepts the BF_Fs this object prepared itself")
        cols = self.lru.get(off0)
        if cols is None:
            if self._plan is not None and int(off0) not in self._planned_keys:
                raise ValueError(f"subgrid column off0={off0} was not in the subgrid_configs plan")
            torch = _torch()
            core = self.core
            cols = torch.empty(
                (len(BF_Fs), core.xM_yN_size, core.yN_size), dtype=self.dtype, device=core.device
            )
            for j, (cfg, BF_F) in enumerate(zip(self.facet_configs, BF_Fs)):
                core.extract_column(BF_F, off0, cfg.off1, out=cols[j], rowmap=self._rowmap, prewindowed=self._prewindowed)
            self.lru.set(off0, cols)
        return cols

    # -- stage 3: per subgrid (api.py:255-279 + api_helper.py:73-112)
    def get_subgrid_task(self, subgrid_config):
        """Finished (masked) subgrid ``[size, size]`` as a device tensor
        (reference api.py:238-253)."""
        return self.get_subgrid_tasks([subgrid_config])[0]

    def _key(self, sg):
        return sg.off1 if self.wave_axis == 1 else sg.off0

    def get_subgrid_tasks(self, subgrid_configs):
        """Finished subgrids for a list of configs; consecutive configs with the
        same wave key (``off0``, or ``off1`` when ``wave_axis == 1``) and
        ``size`` are processed as one wave.  Each result is registered with the
        task queue (``queue_size``).

        With a ``subgrid_configs`` plan, a request that covers only PART of a planned wave (e.g. the reference's
        natural ``off0``-major loop over a cover while the waves are keyed by ``off1``) computes the whole
        planned wave once and keeps the subgrids that were not asked for yet in a cache bounded by
        ``SWIFTLY_RESULT_CACHE_GB`` (default 16); each cached subgrid is handed out once.  Beyond the budget the
        request is computed on its own (correct, slower)."""
        out = []
        i = 0
        while i < len(subgrid_configs):
            j = i + 1
            while (
                j < len(subgrid_configs)
                and self._key(subgrid_configs[j]) == self._key(subgrid_configs[i])
                and subgrid_configs[j].size == subgrid_configs[i].size
            ):
                j += 1
            tasks = self._serve_group(list(subgrid_configs[i:j]))
            self.task_queue.process(tasks)
            out.extend(DeviceTask(t) for t in tasks) if self.delayed else out.extend(tasks)
            i = j
        return out

    @staticmethod
    def _rid(sg):
        """identity of a request: its VALUE (equal configs of a rebuilt cover match the plan, r3 advice)"""
        return (int(sg.off0), int(sg.off1), int(sg.size))

    def _planned_wave_of(self, sg):
        """the planned subgrids that share ``sg``'s wave key and size, in plan order, duplicates dropped (None without
        a plan or when ``sg`` is not in the plan)"""
        if self._plan is None:
            return None
        if self._plan_waves is None:
            waves, seen = {}, set()
            for c in self._plan:
                r = self._rid(c)
                if r in seen:
                    continue
                seen.add(r)
                waves.setdefault((int(self._key(c)), int(c.size)), []).append(c)
            self._plan_waves = (waves, seen)
        waves, seen = self._plan_waves
        if self._rid(sg) not in seen:
            return None
        return waves.get((int(self._key(sg)), int(sg.size)))

    def _drop_result(self, rid):
        hit = self._results.pop(rid, None)
        if hit is not None:
            self._result_bytes -= hit.numel() * hit.element_size()
        return hit

    def _serve_group(self, group):
        """results for consecutive requests sharing the wave key: cache hits, a whole planned wave computed ahead,
        or just the group"""
        rids = [self._rid(sg) for sg in group]
        distinct = len(set(rids)) == len(rids)
        if distinct and all(r in self._results for r in rids):
            return [self._drop_result(r) for r in rids]  # each cached subgrid is handed out once
        # partial hits: the whole group is recomputed below, so the cached copies of its members are released
        # (r3 advice: they used to stay behind and shrink the budget for good)
        for r in set(rids):
            self._drop_result(r)
        full = self._planned_wave_of(group[0])
        if full is not None and distinct:
            in_full = {self._rid(c) for c in full}
            if all(r in in_full for r in rids):
                asked = set(rids)
                extra = [c for c in full if self._rid(c) not in asked and self._rid(c) not in self._results]
                esize = _torch().empty((), dtype=self.dtype).element_size()
                nbytes = sum(c.size * c.size for c in extra) * esize
                if extra and self._result_bytes + nbytes <= self._result_budget:
                    # (the requested config objects take the place of their plan twins: their masks are the ones asked for)
                    req = dict(zip(rids, group))
                    full = [req.get(self._rid(c), c) for c in full]
                    res = self.get_wave(full)
                    by_rid = {self._rid(c): res[k] for k, c in enumerate(full)}
                    for c in extra:
                        self._results[self._rid(c)] = by_rid[self._rid(c)]
                    self._result_bytes += nbytes
                    return [by_rid[r] for r in rids]
        res = self.get_wave(group)
        return [res[k] for k in range(len(group))]

    def get_wave(self, sgs, timer=None):
        """Finished, masked subgrids ``[S, xA, xA]`` of one wave (configs sharing
        the wave key and size).  ``timer`` brackets the stages with HIP events."""
        self.prepare_all_facets()
        if timer is None and self.wave_axis == 1:
            return self._wave_b(sgs)  # two native calls per wave
        t0 = timer.start() if timer is not None else None
        if self.wave_axis == 1:
            self._get_wave_columns(sgs[0].off1)
        else:
            self.get_NMBF_BFs_off0(sgs[0].off0)
        if timer is not None:
            timer.stop("K2_wave_facet_transform", t0)
            t0 = timer.start()
        res = self._wave_b_staged(sgs) if self.wave_axis == 1 else self._wave(sgs)
        if timer is not None:
            timer.stop("K345_extract_sum_finish", t0)
        return res

    def wave_contributions(self, sgs):
        """Contributions of every (local) facet to the subgrids ``sgs`` (same
        ``off0``): tensor ``[F, S, m, m]`` -- the data the reference ships
        between Dask workers (api.py:263-277) and the multi-GPU path ships
        through the all-to-all."""
        if self.wave_axis != 0:
            raise ValueError(
                "wave_contributions belongs to the reference schedule (wave_axis=0); with wave_axis=1 use "
                "distributed.DistributedForward.pack_wave / core.wave_facet_side"
            )
        torch = _torch()
        core = self.core
        m, yN = core.xM_yN_size, core.yN_size
        S, F = len(sgs), len(self.facet_configs)
        cols = self.get_NMBF_BFs_off0(sgs[0].off0)
        contrib = torch.empty((F, S, m, m), dtype=self.dtype, device=core.device)
        off1s = [sg.off1 for sg in sgs]
        for j in range(F):
            core.launch("extract_from_facet", cols[j], m, yN, 1, contrib[j], m, 1,
                        nbatch=S, in_bs=0, out_bs=m * m, offs=off1s)
        return contrib

    def supports_fused_subgrid_side(self):
        """transform_contributions / sum_finish_facets available for this configuration and dtype"""
        return self.core.supports_fused_subgrid(self.dtype)

    def _wave_source(self, sgs):
        """(source tensor, layout, window offsets, row map) of the wave for transform_contributions"""
        if self.wave_axis == 1:
            Q, rowmap = self._get_wave_columns(sgs[0].off1)
            return Q, 1, [sg.off0 for sg in sgs], rowmap
        return self.get_NMBF_BFs_off0(sgs[0].off0), 0, [sg.off1 for sg in sgs], None

    def wave_blocks(self, sgs, out=None, transformed=True):
        """Per-(facet, subgrid) ``[m, m]`` blocks of THIS object's facets for the subgrids ``sgs`` of one wave,
        ``[F, S, m, m]`` written into ``out`` (e.g. a slice of an all-to-all send buffer).  ``transformed``:
        the axis-0-transformed blocks ``G`` of transform_contributions (what the fused subgrid side consumes)
        instead of the raw contributions (reference api.py:263-277)."""
        torch = _torch()
        core = self.core
        m = core.xM_yN_size
        F, S = len(self.facet_configs), len(sgs)
        if out is None:
            out = torch.empty((F, S, m, m), dtype=self.dtype, device=core.device)
        if transformed:
            src, layout, offs, rowmap = self._wave_source(sgs)
            core.transform_contributions(src, layout, [cfg.off0 for cfg in self.facet_configs], offs, out=out, rowmap=rowmap)
            return out
        if self.wave_axis != 0:
            raise NotImplementedError("raw contributions are only produced by the wave_axis=0 pipeline")
        cols = self.get_NMBF_BFs_off0(sgs[0].off0)
        off1s = [sg.off1 for sg in sgs]
        for j in range(F):
            core.launch("extract_from_facet", cols[j], m, core.yN_size, 1, out[j], m, 1,
                        nbatch=S, in_bs=0, out_bs=out.stride(1), offs=off1s)
        return out

    def wave_blocks_into(self, sgs, flat, layout):
        """:py:meth:`wave_blocks` (transformed) of a whole wave with per-subgrid placement inside the flat buffer
        ``flat``: block (f, i) at ``layout[0][i] + f * layout[1][i]`` elements -- one native call
        (contiguous-axis-first pipeline only)."""
        self._check_planned(sgs)
        bands = self.prepare_all_facets()
        Q, rowmap, n_rows, compute = self._wave_Q(sgs[0].off1)
        # (r6) the same free-running K2 chain as _wave_b: K2 of the next waves of the announced order (set_wave_order; the
        # plan's own order otherwise) goes to the side stream before this wave's K3 -- the multi-GPU pass used to compute
        # every K2 on the caller's stream (41.4 against 35.5 ms at world 1)
        nxt = self._predict_next_waves(sgs[0].off1, _knobs()._PREFETCH_DEPTH)
        if not compute:
            self._prefetch_waves(nxt)
        band = self._band
        if compute:
            bands, band = self._k2_source(sgs[0].off1)
        try:
            self.core.wave_facet_side(bands, [cfg.off0 for cfg in self.facet_configs], band, sgs[0].off1, rowmap,
                                      n_rows, Q, compute, [sg.off0 for sg in sgs], flat, g_layout=layout)
        except Exception:
            if compute:  # Q was registered before it was computed: a later request must not find garbage
                self.lru._items.pop(("b", sgs[0].off1), None)  # pylint: disable=protected-access
            raise
        if compute:
            self._prefetch_waves(nxt)

    def _wave(self, sgs):
        # single-GPU route: the [m, m] contributions are never materialised -- the window gather is
        # folded into the axis-0 transform kernel reading the column buffers directly
        cols = self.get_NMBF_BFs_off0(sgs[0].off0)
        try:
            return _finish_from_columns(self.core, cols, 0, self.facet_configs, sgs, [sg.off1 for sg in sgs])
        except NotImplementedError:
            pass
        try:
            colacc = _colacc_from_columns(self.core, cols, self.facet_configs, sgs)
        except NotImplementedError:
            return sum_and_finish_wave(self.core, self.wave_contributions(sgs), self.facet_configs, sgs)
        return _finish_from_colacc(self.core, colacc, self.facet_configs, sgs)

    # -- contiguous-axis-first pipeline (wave_axis == 1) ---------------------------
    def _check_band_pipeline(self):
        torch = _torch()
        if not self.core.supports_band_pipeline(self.dtype):
            raise ValueError("wave_axis=1 is not available for this configuration / dtype (see preferred_wave_axis)")
        if len(self.facet_configs) > self.core.MAX_FUSED_FACETS:
            raise ValueError(
                f"wave_axis=1 sums at most {self.core.MAX_FUSED_FACETS} facets per subgrid in its fused kernel, "
                f"got {len(self.facet_configs)}; use wave_axis=0 (preferred_wave_axis(config, dtype, n_facets=...))"
            )
        sizes = {info[1] for info in self._facet_info}
        if len(sizes) != 1 or not all(info[2] for info in self._facet_info) or self.dtype != torch.complex64:
            raise ValueError("wave_axis=1 needs equally sized row-major complex64 facets")

    def _prepare_all_bands(self, timer=None):
        """K1: band buffers ``[F, yB, band columns]`` -- prepare_facet along axis 1 of every facet row, only the
        columns some planned subgrid window reads, axis-0 window pre-applied.

        (r2's "facet-major schedule" -- K2 of all planned waves per facet on a second stream behind its K1 -- was measured
        no faster than the plain wave loop, K1 and K2 contend for the same HBM / fabric; removed in r4, numbers in
        DESIGN.md section 4.)"""
        if self.BF_Fs_persist is None:
            self._check_band_pipeline()
            torch = _torch()
            core = self.core
            self._band = (
                core.band_for_offsets([sg.off1 for sg in self._plan]) if self._plan is not None else (0, core.yN_size)
            )
            F, yB = len(self._facet_info), self._facet_info[0][1][0]
            mode = self.__dict__["_axis1_mode"] = self._choose_axis1_mode()
            if mode == 2:
                # axis-1-first pipeline, contiguous-axis finish fused into K1: per facet row, for every planned window, the
                # finished window row (core.prepare_facet_window_rows) instead of the band
                keys = sorted(self._planned_keys)
                self.__dict__["_window_of"] = {k: w for w, k in enumerate(keys)}
                starts = torch.tensor(core.window_starts(self._band, keys), dtype=torch.int32, device=core.device)
                # wave-major: K2 of wave w reads the contiguous block [f, w] (side by side in a row measured the same)
                bands = torch.empty((F, len(keys), yB, core.xM_yN_size), dtype=self.dtype, device=core.device)
            else:
                bands = torch.empty((F, yB, core.band_columns(self._band)), dtype=self.dtype, device=core.device)
            for j, cfg in enumerate(self.facet_configs):
                data = self._ingest.ready(j)
                t0 = timer.start() if timer is not None else None
                if mode == 2:
                    core.prepare_facet_window_rows(data, cfg.off1, self._band, starts, bands[j])
                else:
                    core.prepare_facet_band(data, cfg.off1, self._band, out=bands[j])
                if timer is not None:
                    timer.stop("K1_full_facet_transform", t0)
                self._ingest.prefetch(j + 1)
            self.BF_Fs_persist = bands
            self.__dict__["_k2_chain_forked"] = False  # new band buffers: the next prefetched K2 forks behind K1 again
            if self._plan is not None and _knobs()._PREFETCH and _knobs()._PREFETCH_DEPTH >= 2:
                ready = self.__dict__["_bands_ready"] = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(core.device))
        return self.BF_Fs_persist

    def _wave_rows(self, off1):
        """(rowmap, n_rows) of the axis-0 rows the planned subgrids of wave ``off1`` read (None = all rows)."""
        if self._plan is None:
            return None, self.core.yN_size
        key = int(off1)
        if key not in self._wave_rowmaps:
            by_key = self.__dict__.get("_plan_off0s")
            if by_key is None:  # one walk over the plan, not one per wave (505 configs x 25 waves on every pass)
                by_key = self.__dict__["_plan_off0s"] = {}
                for sg in self._plan:
                    by_key.setdefault(int(sg.off1), []).append(sg.off0)
            self._wave_rowmaps[key] = self.core.subgrid_column_rows(by_key.get(key, []))
        return self._wave_rowmaps[key]

    #: may the axis-1-first pipeline fuse the contiguous-axis finish into K1?  The multi-GPU classes switch this off for the
    #: objects whose band buffers come from the band-row exchange (cooperative facets).
    axis1_fused = True

    def _choose_axis1_mode(self):
        """0: default order; 1: axis-1-first with a row pass per wave (core.finish_axis1_rows; ``axis1_first="rows"``, or
        ``True`` where 2 is not available); 2: axis-1-first with the finish in the epilogue of K1 (``axis1_first=True``;
        needs a plan -- the windows are the plan's waves -- and a configuration with core.supports_window_rows)."""
        if not (self.wave_axis == 1 and bool(getattr(self.core, "axis1_first", False))):
            return 0
        if (self.core.axis1_first is True and self.axis1_fused and self._plan is not None and self._band is not None and
                self.core.supports_window_rows(self._band, self._facet_info[0][1][1], [cfg.off1 for cfg in self.facet_configs],
                                               n_windows=len(self._planned_keys))):
            return 2
        return 1

    def _axis1(self):
        """the axis-1-first mode of this object (``SwiftlyConfig(axis1_first=True)``, wave_axis = 1): 0 = off, 1 = row pass
        per wave, 2 = fused into K1 (decided when the facets are prepared)"""
        mode = self.__dict__.get("_axis1_mode")
        if mode is None:
            mode = self._choose_axis1_mode() if self.BF_Fs_persist is None else 0
        return mode

    def _placed(self):
        """the ``placed`` argument of the subgrid side: in both axis-1-first modes the blocks arrive finished along axis 1"""
        return bool(self._axis1())

    def _k2_source(self, off1):
        """``(bands, band)`` that K2 of wave ``off1`` reads: the K1 band buffers and the plan's band -- or, in the
        axis-1-first pipeline (``SwiftlyConfig(axis1_first=True)``), the rows finished along the contiguous axis for this
        wave (core.finish_axis1_rows, on the current stream) with the band that is exactly the wave's window."""
        bands = self.BF_Fs_persist
        mode = self._axis1()
        if not mode:
            return bands, self._band
        if mode == 2:  # the window's finished rows: m columns of the K1 output, read as a band that is exactly the window
            core = self.core
            m, w = core.xM_yN_size, self._window_of[int(off1)]
            start = (core.window_starts(self._band, [off1])[0] + self._band[0]) % core.yN_size
            return bands[:, w], (start, m)
        return self.core.finish_axis1_rows(bands, [cfg.off1 for cfg in self.facet_configs], self._band, off1)

    def _get_wave_columns(self, off1):
        """K2: ``Q[F, rows, m]`` for the subgrid wave ``off1`` (LRU cached like the reference's per-off0 columns)."""
        self._take_prefetched(off1)
        hit = self.lru.get(("b", off1))
        if hit is None:
            if self._plan is not None and int(off1) not in self._planned_keys:
                raise ValueError(f"subgrid wave off1={off1} was not in the subgrid_configs plan")
            self.prepare_all_facets()
            rowmap, n_rows = self._wave_rows(off1)
            bands, band = self._k2_source(off1)
            Q = self.core.prepare_facet_columns(
                bands, [cfg.off0 for cfg in self.facet_configs], band, off1, rowmap, n_rows
            )
            hit = (Q, rowmap)
            self.lru.set(("b", off1), hit)
        return hit

    def _check_planned(self, sgs):
        if self._plan is not None:
            allowed = self.__dict__.get("_plan_set")
            if allowed is None:  # (built once: a set comprehension as a setdefault argument would run on every call)
                allowed = self.__dict__["_plan_set"] = {(int(sg.off0), int(sg.off1)) for sg in self._plan}
            if any((int(sg.off0), int(sg.off1)) not in allowed for sg in sgs):
                raise ValueError("subgrid was not in the subgrid_configs plan")

    def _wave_b_staged(self, sgs):
        """stage-by-stage form (one ABI call per kernel group; used when the stages are timed separately)"""
        Q, rowmap = self._get_wave_columns(sgs[0].off1)
        self._check_planned(sgs)
        return _finish_from_columns(self.core, Q, 1, self.facet_configs, sgs, [sg.off0 for sg in sgs], rowmap=rowmap,
                                    placed=self._placed())

    def _wave_Q(self, off1):
        """(Q workspace, rowmap, n_rows, needs computing) of wave ``off1`` (LRU cached like _get_wave_columns)"""
        torch = _torch()
        self._take_prefetched(off1)
        hit = self.lru.get(("b", off1))
        if hit is not None:
            return hit[0], hit[1], hit[0].shape[1], False
        if self._plan is not None and int(off1) not in self._planned_keys:
            raise ValueError(f"subgrid wave off1={off1} was not in the subgrid_configs plan")
        rowmap, n_rows = self._wave_rows(off1)
        core = self.core
        Q = torch.empty((len(self.facet_configs), n_rows, core.xM_yN_size), dtype=self.dtype, device=core.device)
        self.lru.set(("b", off1), (Q, rowmap))
        return Q, rowmap, n_rows, True
    def _wave_b(self, sgs):
        """One wave = two native calls: facet side (K2 + K3 + K4a) and subgrid side (K4b + K5).  (r3's grouped subgrid
        side -- axis 0 finished first per off1 group, 79 -> 49 MB per subgrid at the same speed -- lives in
        tools/experiments/ since r4.)  With a plan, K2 of the next planned wave is issued on the side stream before
        this wave's subgrid side (_prefetch_wave)."""
        torch = _torch()
        core = self.core
        self._check_planned(sgs)
        bands = self.prepare_all_facets()
        Q, rowmap, n_rows, compute = self._wave_Q(sgs[0].off1)
        nxt = self._predict_next_waves(sgs[0].off1, _knobs()._PREFETCH_DEPTH)
        if not compute:
            self._prefetch_waves(nxt)
        m = core.xM_yN_size
        G = torch.empty((len(self.facet_configs), len(sgs), m, m), dtype=self.dtype, device=core.device)
        band = self._band
        if compute:
            bands, band = self._k2_source(sgs[0].off1)
        try:
            core.wave_facet_side(bands, [cfg.off0 for cfg in self.facet_configs], band, sgs[0].off1, rowmap,
                                 n_rows, Q, compute, [sg.off0 for sg in sgs], G)
        except Exception:
            if compute:
                self.lru._items.pop(("b", sgs[0].off1), None)  # pylint: disable=protected-access
            raise
        if compute:  # (this wave's own K2 was enqueued on the current stream just now: the next one goes behind it)
            self._prefetch_waves(nxt)
        return _finish_from_G(core, G, self.facet_configs, sgs, placed=self._placed())


def _finish_from_columns(core, src, layout, facet_configs, sgs, window_offs, rowmap=None, band=None, placed=False):
    """K3..K5 without any HBM accumulator: per-(facet, subgrid) axis-0 transforms gathered straight from the
    wave's facet buffers (``src``), facet sum + axis-1 finish on chip, axis-0 finish."""
    torch = _torch()
    xM, xA, S = core.xM_size, sgs[0].size, len(sgs)
    dt, dev = src.dtype, core.device
    if dt != torch.complex64:
        raise NotImplementedError("fused subgrid path is complex64 only")
    off0s = [cfg.off0 for cfg in facet_configs]
    off1s = [cfg.off1 for cfg in facet_configs]
    G = core.transform_contributions(src, layout, off0s, window_offs, rowmap=rowmap, band=band)
    return _finish_from_G(core, G, facet_configs, sgs, placed=placed)


def _finish_from_G(core, G, facet_configs, sgs, placed=False):
    """facet sum + axis-1 finish on chip (sum_finish_facets), then the axis-0 finish, for ``G[F, S, m, m]``.  ``placed``:
    blocks of the axis-1-first pipeline (their contiguous axis is already finished up to the placement)."""
    torch = _torch()
    xM, xA, S = core.xM_size, sgs[0].size, len(sgs)
    dt, dev = G.dtype, core.device
    off0s = [cfg.off0 for cfg in facet_configs]
    off1s = [cfg.off1 for cfg in facet_configs]
    mask1 = _mask_table(core, sgs, "mask1", xA, dt)
    mask0 = _mask_table(core, sgs, "mask0", xA, dt)
    tmp = torch.empty((S, xM, xA), dtype=dt, device=dev)
    res = torch.empty((S, xA, xA), dtype=dt, device=dev)
    return core.wave_subgrid_side(G, off0s, off1s, [sg.off0 for sg in sgs], [sg.off1 for sg in sgs], xA, mask0, mask1,
                                  tmp, res, placed=placed)


def finish_from_blocks(core, blocks, facet_configs, sgs, transformed=True, placed=False):
    """Finished, masked subgrids ``[S, xA, xA]`` from the per-(facet, subgrid) blocks ``[F, S, m, m]`` of ALL
    facets (``facet_configs`` in the blocks' facet order): the receiving side of the multi-GPU exchange.
    ``transformed`` as in :py:meth:`SwiftlyForward.wave_blocks`; ``placed``: the senders ran the axis-1-first pipeline."""
    if transformed:
        return _finish_from_G(core, blocks, facet_configs, sgs, placed=placed)
    return sum_and_finish_wave(core, blocks, facet_configs, sgs)


def _facet_grid(facet_configs):
    """(off0 values, off1 values) when the facets form an off0 x off1 grid in row-major order
    (make_full_facet_cover), else None."""
    off0s = sorted({cfg.off0 for cfg in facet_configs})
    off1s = sorted({cfg.off1 for cfg in facet_configs})
    if [(cfg.off0, cfg.off1) for cfg in facet_configs] == [(a, b) for a in off0s for b in off1s]:
        return off0s, off1s
    return None


def _colacc_from_columns(core, cols, facet_configs, sgs):
    """K3+K4a fused: per-off1-group axis-0 sums ``colacc[G, S, xM, m]`` straight from the column
    buffers ``cols[F, m, yN]``."""
    torch = _torch()
    grid = _facet_grid(facet_configs)
    if grid is None:
        raise NotImplementedError("fused path needs an off0 x off1 facet grid")
    off0s, groups = grid
    m, xM, S, G = core.xM_yN_size, core.xM_size, len(sgs), len(groups)
    colacc = torch.zeros((G, S, xM, m), dtype=cols.dtype, device=core.device)
    off1s = [sg.off1 for sg in sgs]
    for i, off0_f in enumerate(off0s):
        core.add_to_subgrid_from_columns(cols[i * G : (i + 1) * G], off0_f, colacc, off1s)
    return colacc


def sum_and_finish_wave(core, contrib, facet_configs, sgs):
    """``sum_and_finish_subgrid`` (reference api_helper.py:73-112) for a wave of
    subgrids that share ``off0`` and ``size``: ``contrib[F, S, m, m]`` (facet
    order = ``facet_configs``) -> finished, masked subgrids ``[S, xA, xA]``."""
    torch = _torch()
    m, xM = core.xM_yN_size, core.xM_size
    S = len(sgs)
    dev, dt = core.device, contrib.dtype
    groups = sorted({cfg.off1 for cfg in facet_configs})  # facets grouped by off1 (api_helper.py:83)
    # K4a: axis-0 transform + placement, summed over the facets of one off1 group
    colacc = torch.zeros((len(groups), S, xM, m), dtype=dt, device=dev)
    grid = _facet_grid(facet_configs)
    if grid is not None and contrib.is_contiguous():
        # facets with the same off0 belong to different groups, so ONE launch per off0 handles all
        # (group, subgrid) pairs -- batch item z = g*S + b reads contrib[i*G + g, b], adds into colacc[g, b]
        G = len(groups)
        for i, off0_f in enumerate(grid[0]):
            core.launch("add_to_subgrid", contrib[i * G], m, 1, m, colacc, 1, m, off0_f,
                        nbatch=G * S, in_bs=m * m, out_bs=xM * m)
    else:
        for j, cfg in enumerate(facet_configs):
            core.launch("add_to_subgrid", contrib[j], m, 1, m, colacc[groups.index(cfg.off1)], 1, m, cfg.off0,
                        nbatch=S, in_bs=m * m, out_bs=xM * m)
    return _finish_from_colacc(core, colacc, facet_configs, sgs)


def _finish_from_colacc(core, colacc, facet_configs, sgs):
    """axis-1 sum over groups + finish (fused where available), then finish along axis 0."""
    torch = _torch()
    m, xM = core.xM_yN_size, core.xM_size
    off0, xA, S = sgs[0].off0, sgs[0].size, len(sgs)
    dev, dt = core.device, colacc.dtype
    groups = sorted({cfg.off1 for cfg in facet_configs})
    off1s = [sg.off1 for sg in sgs]
    mask1 = _mask_table(core, sgs, "mask1", xA, dt)
    mask0 = _mask_table(core, sgs, "mask0", xA, dt)
    tmp = torch.empty((S, xM, xA), dtype=dt, device=dev)
    try:
        core.sum_finish_rows(colacc, groups, tmp, off1s, xA, mask=mask1)
    except NotImplementedError:
        acc = torch.zeros((S, xM, xM), dtype=dt, device=dev)
        for g, off1 in enumerate(groups):
            core.launch("add_to_subgrid", colacc[g], xM, m, 1, acc, xM, 1, off1,
                        nbatch=S, in_bs=xM * m, out_bs=xM * xM)
        core.launch("finish_subgrid", acc, xM, xM, 1, tmp, xA, 1, 0, size=xA, mask=mask1,
                    nbatch=S, in_bs=xM * xM, out_bs=xM * xA, offs=off1s, mask_bs=xA if mask1 is not None else 0)
    res = torch.empty((S, xA, xA), dtype=dt, device=dev)
    core.launch("finish_subgrid", tmp, xA, 1, xA, res, 1, xA, off0, size=xA, mask=mask0,
                nbatch=S, in_bs=xM * xA, out_bs=xA * xA, mask_bs=xA if mask0 is not None else 0)
    return res
