"""
``SwiftlyBackward``: the subgrid -> facet direction of the streaming API (reference src/ska_sdp_exec_swiftly/api.py:327-463,
task bodies api_helper.py:115-197) on one MI355X, with the reference's schedule (``wave_axis=0``) and the band schedule
(``wave_axis=1``).
"""
import logging

import numpy

from .core_hip import band_range
from .ingest import _mask_table
from .tasks import DeviceTask, LRUCache, TaskQueue, _torch, _unwrap

log = logging.getLogger("fourier-logger")


class SwiftlyBackward:
    """Subgrid -> facet streaming transform (reference api.py:327-463).

    :param swiftly_config: SwiftlyConfig
    :param facets_config_list: list of FacetConfig
    :param lru_backward: number of subgrid columns (distinct ``off0``) whose
        partial sums ``NAF_MNAF [m, yN]`` per facet stay in HBM before they are
        folded into the facet accumulators
    :param queue_size: bound on unfinished subgrid tasks (reference
        ``TaskQueue``, api.py:466-522)
    :param wave_axis: 0 (default) = the reference's schedule: partial sums per subgrid ``off0`` column, facet
        accumulators ``[yN, yB]``, strided-axis transform at the end.  1 (complex64, power-of-two sizes) = the
        mirror of the forward ``wave_axis=1`` pipeline: subgrids sharing ``off1`` form a wave, the strided-axis
        ``finish_facet`` runs per wave on ``m`` columns with ``add_to_facet`` fused into its load and its store
        (no column accumulator in HBM), the facet accumulators are bands ``[yB, band]`` and the full-facet
        transform at the end runs along the contiguous axis in one kernel.  Any request order is correct.
    :param subgrid_configs: (wave_axis=1) the subgrids that will be added: sizes the band accumulators to the
        columns they touch; without it the band is the whole padded axis
    """

    # pylint: disable=too-many-arguments,too-many-instance-attributes
    def __init__(self, swiftly_config, facets_config_list, lru_backward=1, queue_size=20, client=None,
                 subgrid_configs=None, wave_axis=None, delayed=False):
        self.delayed = bool(delayed)  # finish() hands out DeviceTask handles instead of bare device tensors
        # wave_axis=None: the reference's schedule, unless the caller hands over the plan of subgrids it will add and
        # the band kernels exist -- decided when the first subgrid shows the dtype (complex64 only)
        self._auto_axis = wave_axis is None
        self.wave_axis = 0 if wave_axis is None else int(wave_axis)
        if self.wave_axis not in (0, 1):
            raise ValueError("wave_axis must be 0 or 1")
        self._plan = list(subgrid_configs) if subgrid_configs is not None else None
        self._plan_counts = None
        self._wsbuf = {}
        self._ring = 0
        self._band = None
        self._bands = None
        self._work = None
        self.config = swiftly_config
        self.core = swiftly_config.core
        self.facets_config_list = facets_config_list
        self.queue_size = queue_size
        self.task_queue = TaskQueue(queue_size)
        self._client = client
        self.lru = LRUCache(lru_backward)
        self.MNAF_BMNAFs_persist = [None for _ in facets_config_list]
        self.dtype = None
        self._off0s = sorted({cfg.off0 for cfg in facets_config_list})
        self._off0_of = [self._off0s.index(cfg.off0) for cfg in facets_config_list]

    def add_new_subgrid_task(self, subgrid_config, new_subgrid_task):
        """Fold one subgrid into the facet sums (reference api.py:347-372)."""
        return self.add_new_subgrid_tasks([subgrid_config], [new_subgrid_task])

    def _resolve_axis(self, first_subgrid):
        """Fix the automatic schedule on the FIRST data this object sees, whichever public entry point it arrives
        through (add_new_subgrid_task(s), wave_contributions, accumulate_wave / accumulate_chunks); it never changes
        afterwards (r3 advice: the low-level entry points used to leave it open, and a later add flipped the schedule
        under accumulators of the other kind)."""
        if not self._auto_axis:
            return
        self._auto_axis = False
        torch = _torch()
        dt = first_subgrid.dtype
        is_c64 = dt in (torch.complex64, torch.float32) if isinstance(first_subgrid, torch.Tensor) else (
            numpy.asarray(first_subgrid).dtype in (numpy.complex64, numpy.float32)
        )
        sizes = {cfg.size for cfg in self.facets_config_list}
        if self._plan is not None and is_c64 and len(sizes) == 1 and self.core.supports_backward_band(torch.complex64):
            self.wave_axis = 1

    def add_new_subgrid_tasks(self, subgrid_configs, new_subgrid_tasks):
        """Fold a list of subgrids into the facet sums (extension: consecutive
        subgrids sharing the wave key -- ``off0``, or ``off1`` with ``wave_axis=1`` -- and ``size`` are processed
        as one wave with batched launches).

        Band schedule (``wave_axis=1``): the per-wave kernels run over ALL subgrids of a wave at once, so subgrids
        that arrive one by one (or in pieces of a planned wave) are first staged -- a device copy into a per-key
        buffer held in ``LRUCache(lru_backward)``, the counterpart of the reference's per-column partial sums
        (api.py:402-438) -- and the wave is folded into the band accumulators when it is complete (plan known),
        evicted from the cache, or at :py:meth:`finish`."""
        new_subgrid_tasks = [_unwrap(t) for t in new_subgrid_tasks]
        if len(subgrid_configs) and self._auto_axis:
            self._resolve_axis(new_subgrid_tasks[0])
        col = None
        i = 0
        key = "off1" if self.wave_axis == 1 else "off0"
        while i < len(subgrid_configs):
            j = i + 1
            while (
                j < len(subgrid_configs)
                and getattr(subgrid_configs[j], key) == getattr(subgrid_configs[i], key)
                and subgrid_configs[j].size == subgrid_configs[i].size
            ):
                j += 1
            if self.wave_axis == 1:
                col = self._add_band_group(subgrid_configs[i:j], new_subgrid_tasks[i:j])
            else:
                col = self._add_wave(subgrid_configs[i:j], new_subgrid_tasks[i:j])
            i = j
        return col

    # ---- wave_axis = 1: staging of partial waves (the role of lru_backward in the band schedule)
    def _planned_count(self, off1, size):
        if self._plan is None:
            return None
        if self._plan_counts is None:
            counts = {}
            for c in self._plan:
                k = (int(c.off1), int(c.size))
                counts[k] = counts.get(k, 0) + 1
            self._plan_counts = counts
        return self._plan_counts.get((int(off1), int(size)))

    def _add_band_group(self, sgs, subgrids):
        key = (int(sgs[0].off1), int(sgs[0].size))
        if self._plan is not None and not any(int(c.off1) == key[0] for c in self._plan):
            raise ValueError(f"subgrid off1={key[0]} is not in the subgrid_configs this SwiftlyBackward was planned for")
        staged = self.lru.get(key)
        planned = self._planned_count(*key)
        if staged is None and (len(sgs) == planned or (planned is None and len(sgs) > 1)):
            return self._add_wave(list(sgs), list(subgrids))  # a whole wave at once: no staging copy
        torch = _torch()
        core = self.core
        xA = sgs[0].size
        if staged is None:
            cap = planned if planned is not None else 8
            staged = dict(cfgs=[], buf=torch.empty((max(cap, len(sgs)), xA, xA), dtype=torch.complex64, device=core.device))
        need = len(staged["cfgs"]) + len(sgs)
        if need > staged["buf"].shape[0]:
            grown = torch.empty((max(need, 2 * staged["buf"].shape[0]), xA, xA), dtype=torch.complex64, device=core.device)
            grown[: len(staged["cfgs"])].copy_(staged["buf"][: len(staged["cfgs"])])
            staged["buf"] = grown
        for sg, data in zip(sgs, subgrids):
            ten, _ = core._as_device(data)  # pylint: disable=protected-access
            if tuple(ten.shape) != (xA, xA):
                raise ValueError(f"subgrid has shape {tuple(ten.shape)}, expected {(xA, xA)}")
            if ten.dtype != torch.complex64:
                raise ValueError("SwiftlyBackward(wave_axis=1) needs complex64 data and power-of-two yN_size / xM_yN_size")
            staged["buf"][len(staged["cfgs"])].copy_(ten)
            staged["cfgs"].append(sg)
        if planned is not None and len(staged["cfgs"]) >= planned:
            self.lru._items.pop(key, None)  # pylint: disable=protected-access
            return self._flush_staged(staged)
        old_key, old = self.lru.set(key, staged)
        if old_key is not None and old is not None:
            self._flush_staged(old)
        return self._bands

    def _flush_staged(self, staged):
        n = len(staged["cfgs"])
        return self._add_wave(staged["cfgs"], [staged["buf"][k] for k in range(n)])

    def _ws(self, name, shape, dtype):
        """Grow-only persistent workspace (per-wave allocations of changing size are kept away from the caching
        allocator: its misses are synchronous hipMallocs)."""
        torch = _torch()
        n = 1
        for d in shape:
            n *= int(d)
        buf = self._wsbuf.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = self._wsbuf[name] = torch.empty((n,), dtype=dtype, device=self.core.device)
        return buf[:n].view(*shape)

    def wave_contributions(self, sgs, subgrids):
        """``prepare_and_split_subgrid`` (reference api_helper.py:115-139) for a
        wave: contributions ``[F, S, m, m]`` of the subgrids ``sgs`` (same size)
        to every facet -- what the reference ships from the subgrid's worker to
        the facets' workers (api.py:357-364).  On the fused route the result lives in one of two alternating
        workspaces of this object: it stays valid until the second-next call."""
        if self._auto_axis and len(subgrids):
            self._resolve_axis(_unwrap(subgrids[0]))
        torch = _torch()
        core = self.core
        m, xM = core.xM_yN_size, core.xM_size
        F, S, D = len(self.facets_config_list), len(sgs), len(self._off0s)
        xA = sgs[0].size
        subs = []
        for data in subgrids:
            ten, _ = core._as_device(data)  # pylint: disable=protected-access
            if self.dtype is None:
                self.dtype = ten.dtype
            elif ten.dtype != self.dtype:
                ten = ten.to(self.dtype)
            if tuple(ten.shape) != (xA, xA):
                raise ValueError(f"subgrid has shape {tuple(ten.shape)}, expected {(xA, xA)}")
            subs.append(ten)
        dev, dt = core.device, self.dtype
        if core.supports_fused_subgrid(dt) and F <= 64:
            # prepare_subgrid along axis 0 on the xA columns, then ONE kernel per padded row for the contiguous-axis
            # half (prepare axis 1 + extract axis 1 for every facet, on chip) and one column pass for the rest
            step = xA * xA * subs[0].element_size()
            base = subs[0].untyped_storage().data_ptr()
            if all(
                t.is_contiguous() and t.data_ptr() == subs[0].data_ptr() + i * step
                and t.untyped_storage().data_ptr() == base  # views of ONE allocation, not neighbours by chance
                for i, t in enumerate(subs)
            ):
                # the subgrids already sit back to back (slices of one wave tensor, e.g. what get_wave returned)
                sub = torch.as_strided(subs[0], (S, xA, xA), (xA * xA, xA, 1))
            else:
                sub = self._ws("stack", (S, xA, xA), dt)
                torch.stack(subs, out=sub)
            work = self._ws("work", (2 * S * xM * xA,), dt)
            self._ring ^= 1
            parts = self._ws(f"parts{self._ring}", (F, S, m, m), dt)
            return core.wave_split_subgrids(sub, [sg.off0 for sg in sgs], [sg.off1 for sg in sgs],
                                            [c.off0 for c in self.facets_config_list],
                                            [c.off1 for c in self.facets_config_list], work, parts)
        sub = subs[0].unsqueeze(0) if S == 1 else torch.stack(subs)
        sub = sub.contiguous()
        # prepare_subgrid (core.py:328-368): axis 1 on the xA rows, then axis 0 on all xM columns
        tmp = torch.empty((S, xA, xM), dtype=dt, device=dev)
        core.launch("prepare_subgrid", sub, xA, xA, 1, tmp, xM, 1, 0, size=xA,
                    nbatch=S, in_bs=xA * xA, out_bs=xA * xM, offs=[sg.off1 for sg in sgs])
        prepared = torch.empty((S, xM, xM), dtype=dt, device=dev)
        core.launch("prepare_subgrid", tmp, xM, 1, xM, prepared, 1, xM, 0, size=xA,
                    nbatch=S, in_bs=xA * xM, out_bs=xM * xM, offs=[sg.off0 for sg in sgs])
        # extract_from_subgrid along axis 0 once per distinct facet off0 (api_helper.py:125-131) ...
        e0 = torch.empty((D, S, m, xM), dtype=dt, device=dev)
        for d, off0_f in enumerate(self._off0s):
            core.launch("extract_from_subgrid", prepared, xM, 1, xM, e0[d], 1, xM, off0_f,
                        nbatch=S, in_bs=xM * xM, out_bs=m * xM)
        # ... and along axis 1 per facet (api_helper.py:133-138)
        parts = torch.empty((F, S, m, m), dtype=dt, device=dev)
        for j, cfg in enumerate(self.facets_config_list):
            core.launch("extract_from_subgrid", e0[self._off0_of[j]], m, xM, 1, parts[j], m, 1, cfg.off1,
                        nbatch=S, in_bs=m * xM, out_bs=m * m)
        return parts

    def accumulate_wave(self, sgs, parts):
        """``accumulate_column`` (reference api_helper.py:142-152) for a wave:
        add the contributions ``parts[F, S, m, m]`` of subgrids sharing the wave key
        into that wave's partial sums.  Grouping key: ``off0`` with ``wave_axis=0`` (the reference's schedule: LRU
        cache keyed by ``off0``, reference api.py:402-438; evicted columns go to the facet accumulators), ``off1``
        with ``wave_axis=1`` (band schedule: the wave is folded straight into the band accumulators).  With
        ``wave_axis=None`` the schedule is fixed by the first data this object sees (:py:meth:`_resolve_axis`); a wave
        whose subgrids do not share the key of the resolved schedule raises ``ValueError``."""
        self._resolve_axis(parts)
        torch = _torch()
        core = self.core
        m, yN = core.xM_yN_size, core.yN_size
        F = len(self.facets_config_list)
        if self.wave_axis == 1:
            return self._accumulate_band(sgs[0].off1, [(sgs, parts)])
        off0 = sgs[0].off0
        if any(int(sg.off0) != int(off0) for sg in sgs):
            raise ValueError(f"reference schedule (wave_axis=0): all subgrids of a wave must share off0={off0}")
        col = self.lru.get(off0)
        if col is None:
            col = torch.zeros((F, m, yN), dtype=parts.dtype, device=core.device)
        # one launch per subgrid (batched over facets): launches are ordered on the stream, so subgrids whose
        # windows overlap never update the same element concurrently
        for b, sg in enumerate(sgs):
            core.launch("add_to_facet", parts[:, b], m, m, 1, col, yN, 1, sg.off1,
                        nbatch=F, in_bs=parts.stride(0), out_bs=m * yN)
        old_off0, old_col = self.lru.set(off0, col)
        if old_off0 is not None and old_col is not None:
            self.update_MNAF_BMNAFs(old_off0, old_col)
        return col

    def accumulate_chunks(self, off0, chunks):
        """:py:meth:`accumulate_wave` for contributions that arrive in several pieces (one per source rank of
        the multi-GPU exchange): ``chunks = [(subgrid configs, parts[F, S_c, m, m]), ...]``, all of wave ``off0`` --
        the wave KEY: the subgrids' ``off0`` with ``wave_axis=0``, their ``off1`` with ``wave_axis=1`` (checked)."""
        if self._auto_axis and len(chunks):
            self._resolve_axis(chunks[0][1])
        torch = _torch()
        core = self.core
        m, yN = core.xM_yN_size, core.yN_size
        F = len(self.facets_config_list)
        if self.wave_axis == 1:  # ``off0`` is the wave key: the subgrids' off1
            return self._accumulate_band(off0, chunks)
        col = self.lru.get(off0)
        for sgs, parts in chunks:
            if self.dtype is None:
                self.dtype = parts.dtype
            if col is None:
                col = torch.zeros((F, m, yN), dtype=parts.dtype, device=core.device)
            for b, sg in enumerate(sgs):
                core.launch("add_to_facet", parts[:, b], m, m, 1, col, yN, 1, sg.off1,
                            nbatch=F, in_bs=parts.stride(0), out_bs=m * yN)
        if col is None:
            return None
        old_off0, old_col = self.lru.set(off0, col)
        if old_off0 is not None and old_col is not None:
            self.update_MNAF_BMNAFs(old_off0, old_col)
        return col

    def _add_wave(self, sgs, subgrids):
        parts = self.wave_contributions(sgs, subgrids)
        col = self.accumulate_wave(sgs, parts)
        self.task_queue.process([col])
        return col

    # ---- wave_axis = 1: band accumulators
    def _band_state(self, dtype):
        """Band, accumulators ``[F, yB, band length]`` (zeros) and the facet mask table, created at first use."""
        torch = _torch()
        core = self.core
        if self._bands is None:
            if dtype != torch.complex64 or not core.supports_backward_band(dtype):
                raise ValueError("SwiftlyBackward(wave_axis=1) needs complex64 data and power-of-two yN_size / xM_yN_size")
            sizes = {cfg.size for cfg in self.facets_config_list}
            if len(sizes) != 1:
                raise ValueError("SwiftlyBackward(wave_axis=1) needs facets of one size")
            yB = sizes.pop()
            # (the backward accumulators are plain-order bands for every yN: band_range, not the forward layout rule)
            self._band = (
                band_range(core.N, core.yN_size, core.xM_yN_size, [sg.off1 for sg in self._plan])
                if self._plan else (0, core.yN_size)
            )
            self._planned = {sg.off1 for sg in self._plan} if self._plan else None
            F = len(self.facets_config_list)
            # uninitialised: first-write flags per band column replace the zero fill
            self._bands = torch.empty((F, yB, self._band[1]), dtype=dtype, device=core.device)
            self._touched = torch.zeros((self._band[1],), dtype=torch.uint8, device=core.device)
            self._masks0 = _mask_table(core, self.facets_config_list, "mask0", yB, dtype)
            self._facet_off0s = [cfg.off0 for cfg in self.facets_config_list]
            # four-step scratch of accumulate_facet_columns (+ the radix-Q pass's output when yN = Q * 2^k)
            self._work = torch.empty((core._k2_scratch_bytes(F) // 8,), dtype=dtype, device=core.device)
        return self._bands

    def _accumulate_band(self, off1, chunks):
        """accumulate_column + accumulate_facet (reference api_helper.py:142-179) with the axes swapped, for the
        contributions ``chunks = [(subgrid configs, parts[F, S_c, m, m]), ...]`` of subgrids sharing ``off1``."""
        core = self.core
        m = core.xM_yN_size
        chunks = [(sgs, parts) for sgs, parts in chunks if len(sgs)]
        if not chunks:
            return None
        for sgs, _parts in chunks:
            # (r4 advice) the band schedule folds a wave under ONE off1: a caller that follows the reference's per-off0
            # flow (accumulate_column, api_helper.py:142-152) on an object whose schedule resolved to wave_axis=1 must
            # hear about it instead of getting every subgrid placed at the first one's off1
            bad = [sg for sg in sgs if int(sg.off1) != int(off1)]
            if bad:
                raise ValueError(
                    f"band schedule (wave_axis=1): all subgrids of a wave must share off1={off1}, got off1={bad[0].off1}; "
                    "group the subgrids by off1, or construct SwiftlyBackward(wave_axis=0) for the reference's per-off0 flow"
                )
        if self.dtype is None:
            self.dtype = chunks[0][1].dtype
        bands = self._band_state(chunks[0][1].dtype)
        if self._planned is not None and off1 not in self._planned:
            raise ValueError(f"subgrid off1={off1} is not in the subgrid_configs this SwiftlyBackward was planned for")
        F = len(self.facets_config_list)
        dt0 = chunks[0][1].dtype
        fstr, off0s, locs = [], [], []
        fixed = []
        for c, (sgs, parts) in enumerate(chunks):
            if parts.shape[0] != F or parts.dtype != dt0:
                raise ValueError("contribution chunk does not match the facet list / dtype")
            if parts.stride(3) != 1 or parts.stride(2) != m or (parts.shape[1] > 1 and parts.stride(1) != m * m):
                parts = parts.contiguous()
            fixed.append(parts)  # keeps a contiguous copy alive until the launch is queued
            fstr.append(parts.stride(0) if F > 1 else 0)
            for b, sg in enumerate(sgs):
                off0s.append(sg.off0)
                locs.append((c, b))
        # chunk offsets are relative to the LOWEST chunk address: the gather-sum kernel reads a negative 64-bit offset
        # as "no source row", so a chunk allocated below the base (a contiguous copy, a separately allocated chunk
        # handed to accumulate_chunks) would otherwise be dropped silently
        base = min(fixed, key=lambda t: t.data_ptr())
        offs = [(t.data_ptr() - base.data_ptr()) // base.element_size() for t in fixed]
        if len(chunks) > core.GS_MAX_CHUNKS:
            raise ValueError(f"at most {core.GS_MAX_CHUNKS} contribution chunks per wave")
        for _members, table in core.column_row_sources(off0s, locs):
            core.accumulate_facet_columns(base, m, offs, fstr, table, self._facet_off0s, bands.shape[1], self._masks0,
                                          off1, bands, self._band, workspace=self._work, touched=self._touched)
        return bands

    def _finish_bands(self):
        torch = _torch()
        core = self.core
        out = []
        if self._bands is None:
            dt = self.dtype or torch.complex64
            return [torch.zeros((cfg.size, cfg.size), dtype=dt, device=core.device) for cfg in self.facets_config_list]
        core.band_zero_untouched(self._bands, self._touched)
        for j, cfg in enumerate(self.facets_config_list):
            out.append(core.finish_facet_band(self._bands[j], self._band, cfg.off1, cfg.size, mask=cfg.mask1))
        self._bands = None
        self._work = None
        return out

    def update_MNAF_BMNAFs(self, off0, NAF_MNAFs):
        """accumulate_facet for every facet (reference api.py:440-463,
        api_helper.py:155-179): finish axis 1 (+mask1), add along axis 0."""
        torch = _torch()
        core = self.core
        yN = core.yN_size
        dev, dt = core.device, NAF_MNAFs.dtype
        for j, cfg in enumerate(self.facets_config_list):
            yB = cfg.size
            t = core.finish_facet(NAF_MNAFs[j], cfg.off1, yB, axis=1, mask=cfg.mask1)
            if self.MNAF_BMNAFs_persist[j] is None:
                self.MNAF_BMNAFs_persist[j] = torch.zeros((yN, yB), dtype=dt, device=dev)
            core.launch("add_to_facet", t, yB, 1, yB, self.MNAF_BMNAFs_persist[j], 1, yB, off0)
        return self.MNAF_BMNAFs_persist

    def finish(self):
        """Flush the column cache and finish all facets (reference
        api.py:374-400, api_helper.py:182-197).  A facet that never received a
        contribution is all zeros (the reference raises AttributeError there,
        api_helper.py:184-187)."""
        torch = _torch()
        core = self.core
        if self.wave_axis == 1:
            for _key, staged in self.lru.pop_all():
                self._flush_staged(staged)
            out = self._finish_bands()
            self.task_queue.wait_all_done()
            return [DeviceTask(t) for t in out] if self.delayed else out
        for old_off0, old_col in self.lru.pop_all():
            self.update_MNAF_BMNAFs(old_off0, old_col)
        out = []
        for cfg, acc in zip(self.facets_config_list, self.MNAF_BMNAFs_persist):
            if acc is None:
                dt = self.dtype or torch.complex64
                out.append(torch.zeros((cfg.size, cfg.size), dtype=dt, device=core.device))
            else:
                out.append(core.finish_facet(acc, cfg.off0, cfg.size, axis=0, mask=cfg.mask0))
        self.task_queue.wait_all_done()
        return [DeviceTask(t) for t in out] if self.delayed else out
