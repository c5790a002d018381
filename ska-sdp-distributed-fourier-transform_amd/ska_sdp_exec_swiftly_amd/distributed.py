"""
Multi-GPU forward and backward passes: one process per GPU (torch.distributed;
backend "nccl" is RCCL on ROCm, xGMI underneath), facets sharded over ranks,
subgrids of a wave owned round-robin, and ONE exchange step per wave: the
per-(facet, subgrid) contribution shuffle that the reference leaves to Dask's
worker-to-worker transfers (forward: reference api.py:263-277; backward:
api.py:357-364, api_helper.py:133-139) becomes an ``all_to_all_single`` of
``[m, m]`` blocks.

Layout of the exchange (no staging copies on either side):

* forward, sender: the kernels that produce the blocks write them straight into
  the send buffer ``[dest rank][local facet][subgrid owned by dest][m, m]``
  (``transform_contributions`` takes output strides);
  receiver: the receive buffer ``[source rank][facet of source][my subgrid]
  [m, m]`` IS ``[facet in arrival order][my subgrid][m, m]`` -- the consumer
  (``sum_finish_facets``) just gets the facet offsets in arrival order.
* backward, sender: contributions are computed with the facets in owner-major
  order, so the result ``[facet][my subgrid][m, m]`` is already the send
  buffer; receiver: one chunk ``[my facet][subgrid of source][m, m]`` per
  source rank, consumed chunk by chunk.

The exchange logic is independent of who computes the blocks, so the
partitioning / ordering is covered by world_size-2/3 gloo tests on CPU
(tests/test_distributed_cpu.py) while the product wires in the HIP kernels
(:class:`DistributedForward`, :class:`DistributedBackward`).
"""
import numpy

__all__ = [
    "FacetSharding",
    "exchange_contributions",
    "start_exchange",
    "exchange_blocks",
    "DistributedForward",
    "DistributedBackward",
]


def _torch():
    import torch  # pylint: disable=import-outside-toplevel

    return torch


class FacetSharding:
    """Who owns what.  Facet ``j`` lives on rank ``j % world``.  The subgrids of
    a wave are finished (forward) / held (backward) round-robin by the ranks
    with the FEWEST facets: when the facet count does not divide by the world
    size (the catalogue's 3x3 facets on 8 GPUs) the ranks that carry an extra
    facet already have the longest facet-side critical path, so they take no
    subgrid-side work (``balance=False``: every rank takes subgrids).

    **Cooperative facets** (r4; ``wave_keys`` = the wave keys of the plan in band order): the ``F mod world`` facets
    that do not fill a round are not handed to single ranks (which bounds the 3x3 cover at 4.9x on 8 GPUs) but worked
    on by ALL ranks -- the reference balances at (facet, column) task granularity too (api.py:300-324):

    * K1 (the full-facet transform along the contiguous axis) is row-independent: rank ``r`` transforms the rows
      ``coop_rows(size, r)`` of every cooperative facet;
    * the per-wave facet-side work (K2, K3) of a cooperative facet belongs to ONE rank per wave, ``key_owner[key]``,
      contiguous ranges of waves per rank -- so each rank needs all rows but only the band columns of its waves: one
      all-to-all per pass moves the band rows (row blocks out, column ranges in);
    * the blocks of wave ``key`` therefore come from ``items_of(r, key)`` = the whole facets of ``r`` plus the
      cooperative facets it owns for that wave, and arrive in the order ``arrival(key)``."""

    def __init__(self, n_facets, rank, world, balance=True, wave_keys=None, wave_sizes=None):
        self.n_facets, self.rank, self.world = n_facets, rank, world
        self._memo = {}  # layouts by (direction, subgrids, block size, wave key): a plan's waves come back every pass
        # wave_sizes = {wave key: subgrids in the wave}: WHOLE waves are owned by single ranks (largest waves first,
        # always to the rank with the fewest subgrids so far) instead of every wave being dealt out subgrid by subgrid.
        # At 8 ranks a rank then finishes ~3 whole waves of ~20 subgrids instead of 25 x 2-3 subgrids: the subgrid-side
        # kernels run at their single-GPU efficiency (measured r4, 64k workload: 45 -> 24 us per wave and rank).  The
        # exchange of a wave becomes a gather to its owner; over consecutive waves the owners rotate.
        #   A gather per wave would leave six of a GPU's seven xGMI links idle, so the waves are exchanged in GROUPS of
        # `world` waves with distinct owners (``wave_groups``): one balanced all-to-all per group
        # (DistributedForward.start_group / finish_group).
        self.wave_rank, self.wave_groups = None, None
        if wave_sizes is not None:  # (world = 1 too: groups of one wave -- the one-rank rehearsal of bench.py --rccl-dry)
            load = [0] * world
            self.wave_rank, self.wave_groups = {}, []
            by_size = sorted(wave_sizes.items(), key=lambda kv: (-kv[1], kv[0]))
            for g0 in range(0, len(by_size), world):
                group = by_size[g0 : g0 + world]  # waves of similar size; the largest goes to the least loaded rank
                ranks = sorted(range(world), key=lambda q: (load[q], q))
                for (k, n), r in zip(group, ranks):
                    self.wave_rank[int(k)] = r
                    load[r] += n
                self.wave_groups.append([int(k) for k, _ in group])
        n_whole = n_facets
        self.coop = []
        if wave_keys is not None and world > 1 and n_facets % world:
            n_whole = (n_facets // world) * world
            self.coop = list(range(n_whole, n_facets))
        self.facets_of = [[j for j in range(n_whole) if j % world == r] for r in range(world)]
        self.local_facets = self.facets_of[rank]
        counts = [len(f) for f in self.facets_of]
        if balance and min(counts) < max(counts):
            self.subgrid_ranks = [r for r in range(world) if counts[r] == min(counts)]
        else:
            self.subgrid_ranks = list(range(world))
        # facet order after concatenating received blocks in source-rank order (= owner-major order)
        self.arrival_order = [j for r in range(world) for j in self.facets_of[r]]
        self.to_global = numpy.argsort(self.arrival_order)  # arrival position of global facet j
        self.wave_keys, self.key_owner, self.keys_of = None, {}, [[] for _ in range(world)]
        if self.coop:
            self.wave_keys = [int(k) for k in wave_keys]
            K = len(self.wave_keys)
            bounds = [(i * K) // world for i in range(world + 1)]
            for r in range(world):
                self.keys_of[r] = self.wave_keys[bounds[r] : bounds[r + 1]]
                for k in self.keys_of[r]:
                    self.key_owner[k] = r

    def subgrids_of(self, n_subgrids, rank=None, key=None):
        """indices (within the wave) of the subgrids of ``rank``; ``key`` = the wave key (needed when whole waves are
        owned by single ranks)"""
        rank = self.rank if rank is None else rank
        if self.wave_rank is not None:
            if key is None:
                raise ValueError("whole-wave subgrid ownership: subgrids_of needs the wave key")
            return list(range(n_subgrids)) if self.wave_rank[int(key)] == rank else []
        if rank not in self.subgrid_ranks:
            return []
        return list(range(self.subgrid_ranks.index(rank), n_subgrids, len(self.subgrid_ranks)))

    def coop_rows(self, size, rank=None):
        """``(row0, rows)`` of a cooperative facet with ``size`` rows that ``rank`` transforms in K1 (blocks of whole
        multiples of 8 rows, consecutive in rank order)"""
        rank = self.rank if rank is None else rank
        cut = [min(size, ((i * size) // self.world + 7) // 8 * 8) for i in range(self.world)] + [size]
        return cut[rank], cut[rank + 1] - cut[rank]

    def items_of(self, rank, key=None):
        """global facet indices whose blocks ``rank`` produces for wave ``key``"""
        own = list(self.facets_of[rank])
        if self.coop and key is not None and self.key_owner.get(int(key)) == rank:
            own += self.coop
        return own

    def arrival(self, key=None):
        """facet order of the blocks of wave ``key`` after concatenating the received chunks in source-rank order"""
        return [j for r in range(self.world) for j in self.items_of(r, key)]


class _Pending:
    """In-flight all-to-all (keeps the buffers alive until the collective is done)."""

    def __init__(self, work, recv, send):
        self.work, self.recv, self.send = work, recv, send

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.recv


def _all_to_all(send, in_counts, out_counts, group=None, async_op=True):
    """``all_to_all_single`` of a contiguous (complex or real) 1-D tensor whose per-destination chunks are
    consecutive; counts in elements.  RCCL has no complex type: complex data travels as interleaved reals."""
    torch = _torch()
    dist = torch.distributed
    recv = torch.empty(int(sum(out_counts)), dtype=send.dtype, device=send.device)
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # ranks without RCCL between them (several processes on ONE GPU, as in tests/test_hip_multiprocess_gpu.py):
        # stage through host memory, synchronously.  Same layouts, same counts -- only the transport differs.
        real = send.is_complex()
        h_send = (torch.view_as_real(send) if real else send).reshape(-1).cpu()
        h_recv = torch.empty((2 if real else 1) * int(sum(out_counts)), dtype=h_send.dtype)
        k = 2 if real else 1
        dist.all_to_all_single(h_recv, h_send, [k * int(n) for n in out_counts], [k * int(n) for n in in_counts], group=group)
        (torch.view_as_real(recv) if real else recv).reshape(-1).copy_(h_recv)
        return _Pending(None, recv, send)
    if send.is_complex():
        work = dist.all_to_all_single(
            torch.view_as_real(recv).reshape(-1), torch.view_as_real(send).reshape(-1),
            [2 * int(n) for n in out_counts], [2 * int(n) for n in in_counts], group=group, async_op=async_op,
        )
    else:
        work = dist.all_to_all_single(recv, send, [int(n) for n in out_counts], [int(n) for n in in_counts],
                                      group=group, async_op=async_op)
    return _Pending(work if async_op else None, recv, send)


def _shared_sharding(core, n_facets, rank, world, wave_keys, wave_sizes):
    """The sharding of a (cover, plan, rank) -- pure bookkeeping derived from the arguments -- is kept on the core, so
    that the objects of consecutive passes over one plan share it and its layout memo (r4 review: 3.8 ms of Python per
    pass and rank against a 6.4 ms GPU budget at 8 ranks)."""
    cache = core.__dict__.setdefault("_sharding_cache", {})
    ck = (n_facets, rank, world, None if wave_keys is None else tuple(int(k) for k in wave_keys),
          None if wave_sizes is None else tuple(sorted((int(k), int(v)) for k, v in wave_sizes.items())))
    hit = cache.get(ck)
    if hit is None:
        if len(cache) >= 64:
            cache.clear()
        hit = cache[ck] = FacetSharding(n_facets, rank, world, wave_keys=wave_keys, wave_sizes=wave_sizes)
    return hit


def forward_layout(sharding, n_subgrids, blk, key=None):
    """Element counts of the forward exchange: (per-destination subgrid index lists, in_counts, out_counts); ``key``
    = the wave key when facets are worked on cooperatively (the senders' item counts depend on the wave)."""
    memo = sharding._memo  # pylint: disable=protected-access
    hit = memo.get(("f", n_subgrids, blk, key))
    if hit is None:
        F_local = len(sharding.items_of(sharding.rank, key))
        mine = sharding.subgrids_of(n_subgrids, key=key)
        dests = [sharding.subgrids_of(n_subgrids, r, key) for r in range(sharding.world)]
        in_counts = [F_local * len(d) * blk for d in dests]
        out_counts = [len(sharding.items_of(r, key)) * len(mine) * blk for r in range(sharding.world)]
        hit = memo[("f", n_subgrids, blk, key)] = (dests, in_counts, out_counts)  # shared: callers do not modify them
    return hit


def backward_layout(sharding, n_subgrids, blk, key=None):
    """Element counts of the backward exchange (subgrid holder -> facet owner)."""
    memo = sharding._memo  # pylint: disable=protected-access
    hit = memo.get(("b", n_subgrids, blk, key))
    if hit is None:
        mine = sharding.subgrids_of(n_subgrids, key=key)
        F_local = len(sharding.items_of(sharding.rank, key))
        in_counts = [len(sharding.items_of(r, key)) * len(mine) * blk for r in range(sharding.world)]
        out_counts = [F_local * len(sharding.subgrids_of(n_subgrids, r, key)) * blk for r in range(sharding.world)]
        hit = memo[("b", n_subgrids, blk, key)] = (in_counts, out_counts)
    return hit


# One-rank rehearsal (bench.py --rccl-dry, tests): with FORCE_COLLECTIVE the exchange of a ONE-rank group still goes
# through the backend's all_to_all_single (a self-copy through RCCL on its own stream) instead of being short-circuited.
FORCE_COLLECTIVE = False


def exchange_blocks(send, in_counts, out_counts, group=None, async_op=True):
    """Start the all-to-all of an already laid-out send buffer; ``wait()`` returns the flat receive buffer."""
    torch = _torch()
    ready = torch.distributed.is_available() and torch.distributed.is_initialized()
    if not ready or (len(in_counts) == 1 and not FORCE_COLLECTIVE):
        return _Pending(None, send, send)
    return _all_to_all(send.reshape(-1), in_counts, out_counts, group, async_op)


# -- r1 interface (kept: generic tests and callers that materialise [F_local, S, m, m]) ------------------------
class _Reordered:
    def __init__(self, pending, sharding, n_mine, blk):
        self.pending, self.sharding, self.n_mine, self.blk = pending, sharding, n_mine, blk

    def wait(self):
        torch = _torch()
        recv = self.pending.wait()
        arrived = recv.reshape(self.sharding.n_facets, self.n_mine, *self.blk)  # arrival (owner-major) facet order
        return arrived[torch.as_tensor(self.sharding.to_global, device=arrived.device)]


class _Done:
    def __init__(self, value):
        self.value = value

    def wait(self):
        return self.value


def start_exchange(contrib_local, sharding, group=None, async_op=True):
    """Issue the forward all-to-all of one wave for contributions materialised as ``[F_local, S, m, m]`` and
    return a handle whose ``wait()`` gives ``[F, S_local, m, m]`` (all facets in GLOBAL order, the subgrids this
    rank owns).  This convenience form stages the data twice (pack per destination, reorder on arrival);
    :class:`DistributedForward` avoids both copies."""
    torch = _torch()
    world = sharding.world
    S = contrib_local.shape[1]
    blk = tuple(contrib_local.shape[2:])
    if world == 1:
        return _Done(contrib_local)
    nblk = int(numpy.prod(blk))
    dests, in_counts, out_counts = forward_layout(sharding, S, nblk)
    send = torch.cat([contrib_local[:, d].reshape(-1) for d in dests])
    pending = _all_to_all(send, in_counts, out_counts, group, async_op)
    return _Reordered(pending, sharding, len(sharding.subgrids_of(S)), blk)


def exchange_contributions(contrib_local, sharding, group=None):
    """Blocking form of :func:`start_exchange`."""
    return start_exchange(contrib_local, sharding, group, async_op=False).wait()


def _dist_info(group):
    torch = _torch()
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _wave_sizes(plan, wave_axis):
    """{wave key: number of subgrids} of a plan (whole-wave subgrid ownership needs a plan)"""
    if not plan:
        raise ValueError("whole_waves=True needs the plan of subgrids (subgrid_configs)")
    sizes = {}
    for sg in plan:
        k = int(sg.off1 if wave_axis == 1 else sg.off0)
        sizes[k] = sizes.get(k, 0) + 1
    return sizes


def _plan_wave_keys(core, plan):
    """Wave keys (subgrid ``off1``) of a plan in BAND order -- sorted by the position of their column window in the
    band of the whole plan -- so that contiguous key ranges are contiguous column ranges; ``(keys, band)``."""
    band = core.band_for_offsets([sg.off1 for sg in plan])
    yN, m, N = core.yN_size, core.xM_yN_size, core.N
    keys = sorted({int(sg.off1) for sg in plan}, key=lambda k: (yN // 2 - m // 2 + k * yN // N - band[0]) % yN)
    return keys, band


class _CoopBands:
    """Column bookkeeping of the cooperative facets of one rank set: the band of the whole plan, the sub-band of every
    rank's wave range and where it sits inside the parity-split layout of the whole band (swiftly_rowpass.h): a band that
    starts an even number of columns ``delta`` after the whole band has its even / odd runs at columns ``delta / 2``
    of the whole band's even / odd runs."""

    def __init__(self, core, sharding, band):
        self.band = band
        self.half = core.band_columns(band) // 2
        self.sub = []  # per rank: (band of its waves or None, half columns, delta / 2, columns copied per run)
        for r in range(sharding.world):
            keys = sharding.keys_of[r]
            if not keys:
                self.sub.append((None, 0, 0, 0))
                continue
            b = core.band_for_offsets(keys)
            delta = (b[0] - band[0]) % core.yN_size
            if b == (0, core.yN_size) or delta % 2 or delta + b[1] > band[1]:
                raise ValueError("internal: the band of a wave range is not a sub-band of the plan's band")
            h = core.band_columns(b) // 2
            self.sub.append((b, h, delta // 2, min(h, self.half - delta // 2)))


class DistributedForward:
    """Facet-sharded ``SwiftlyForward`` (HIP).  Every rank constructs it with
    the FULL list of facet configs but only the data of its own facets
    (``facet_data[j]`` for ``j in sharding.local_facets``; other entries are
    ignored and may be ``None``).

    Per wave each rank runs K1/K2 for its facets, computes the axis-0-transformed
    contributions ``G`` (``transform_contributions``) of its facets to ALL
    subgrids of the wave directly into the send buffer, the all-to-all moves the
    blocks to the subgrids' owners, and the owners run ``sum_finish_facets`` +
    the axis-0 finish.  ``dtype``: complex dtype of the pass (all ranks must
    agree; a rank without facets cannot infer it).

    **Facet counts that do not divide by the world size** (r4, band pipeline with a plan): the leftover facets are
    worked on cooperatively (:class:`FacetSharding`) -- every rank needs the rows ``sharding.coop_rows(size)`` of each
    of them in ``facet_data[j]`` (the whole facet is accepted too), :py:meth:`prepare_all_facets` runs K1 on them and the
    band-row exchange, and each wave's K2 / K3 of a cooperative facet runs on the rank that owns the wave.
    ``cooperative=False`` keeps whole-facet ownership (facet ``j`` on rank ``j % world``)."""

    def __init__(self, swiftly_config, facet_configs, facet_data, lru_forward=1, group=None, subgrid_configs=None,
                 wave_axis=None, dtype=None, rank_world=None, cooperative=True, whole_waves=False):
        from .api import SwiftlyForward, preferred_wave_axis  # pylint: disable=import-outside-toplevel

        torch = _torch()
        self.group = group
        # rank_world=(rank, world) overrides the process group: "virtual ranks" of one process, used with
        # pack_wave / unpack_wave and a caller-provided exchange (tests of the multi-rank layouts on one GPU)
        self.rank, self.world = rank_world if rank_world is not None else _dist_info(group)
        self._virtual = rank_world is not None
        self.config = swiftly_config
        self.core = swiftly_config.core
        self.facet_configs = list(facet_configs)
        self.dtype = dtype if dtype is not None else torch.complex64
        if wave_axis is None:
            wave_axis = preferred_wave_axis(swiftly_config, self.dtype, n_facets=len(self.facet_configs))
        self.wave_axis = wave_axis
        # the receiving side sums over ALL facets of the cover in one kernel (<= MAX_FUSED_FACETS of them)
        self.fused = (
            self.dtype == torch.complex64
            and self.core.supports_fused_subgrid(self.dtype, n_facets=len(self.facet_configs))
        )
        if self.wave_axis == 1 and not self.fused:
            raise ValueError(
                f"wave_axis=1 needs the fused subgrid side (complex64, <= {self.core.MAX_FUSED_FACETS} facets in total); "
                "use wave_axis=0 (preferred_wave_axis(config, dtype, n_facets=...) says which)"
            )
        keys = None
        self._plan = list(subgrid_configs) if subgrid_configs is not None else None
        if (cooperative and self.wave_axis == 1 and self._plan and self.world > 1 and len(self.facet_configs) % self.world
                and self.core.band_for_offsets([self._plan[0].off1]) != (0, self.core.yN_size)
                and len({c.size for c in self.facet_configs}) == 1):
            keys, band = _plan_wave_keys(self.core, self._plan)
            if band == (0, self.core.yN_size):
                keys = None  # the plan needs the whole padded axis: no sub-bands to hand out
        self.sharding = _shared_sharding(self.core, len(self.facet_configs), self.rank, self.world, keys,
                                         _wave_sizes(self._plan, self.wave_axis) if whole_waves else None)
        local = self.sharding.local_facets
        self.local = None
        if local or not self.sharding.coop:
            self.local = SwiftlyForward(
                swiftly_config,
                [(self.facet_configs[j], facet_data[j]) for j in local],
                lru_forward=lru_forward,
                subgrid_configs=subgrid_configs,
                wave_axis=wave_axis,
            )
            if local and self.local.dtype != self.dtype:
                raise ValueError(f"local facets are {self.local.dtype}, the pass was declared {self.dtype}")
            self.local.dtype = self.dtype
        self.arrival_cfgs = [self.facet_configs[j] for j in self.sharding.arrival_order]
        # cooperative facets: one single-facet SwiftlyForward per facet over the waves this rank owns; its band buffer
        # is assembled by the band-row exchange of prepare_all_facets instead of its own K1
        self._coop, self._coop_data, self._coop_bands, self._coop_ready, self._coop_seen = {}, {}, None, False, set()
        if self.sharding.coop:
            self._coop_bands = _CoopBands(self.core, self.sharding, band)
            mine = set(self.sharding.keys_of[self.rank])
            my_plan = [sg for sg in self._plan if int(sg.off1) in mine]
            for j in self.sharding.coop:
                cfg = self.facet_configs[j]
                self._coop_data[j] = facet_data[j]
                if my_plan:
                    ghost = torch.zeros(1, dtype=self.dtype, device=self.core.device).expand(cfg.size, cfg.size)
                    fwd = SwiftlyForward(swiftly_config, [(cfg, ghost)], lru_forward=lru_forward, subgrid_configs=my_plan,
                                         wave_axis=1)
                    fwd.dtype = self.dtype
                    fwd.axis1_fused = False  # (its band buffer comes from the band-row exchange)
                    self._coop[j] = fwd
        self._announce_wave_order()

    def _announce_wave_order(self):
        """With whole-wave ownership the waves are packed group by group, inside a group in the order of their owners
        (pack_group) -- not in the plan's order: tell the planned-wave predictors of the local objects, so that K2 of the
        next waves runs ahead on the side stream as it does in the single-GPU pass (r6)."""
        sh = self.sharding
        if self.wave_axis != 1 or not self._plan or getattr(sh, "wave_rank", None) is None or not getattr(sh, "wave_groups", None):
            return
        order = [k for group in sh.wave_groups for k in sorted(group, key=lambda k: sh.wave_rank[k])]
        if self.local is not None:
            self.local.set_wave_order(order)
        for fwd in self._coop.values():
            fwd.set_wave_order(order)

    # -- cooperative facets: K1 on this rank's rows + the band-row exchange ----------------------------------------
    def pack_coop(self, j):
        """K1 of this rank's rows of cooperative facet ``j`` over the band of the whole plan, cut into the column
        ranges of the ranks' wave ranges: ``(send, in_counts, out_counts)`` -- chunk ``d`` = ``[my rows, band columns
        of rank d]``, received chunk ``s`` = ``[rows of rank s, my band columns]``."""
        torch = _torch()
        core, cb, sh = self.core, self._coop_bands, self.sharding
        cfg = self.facet_configs[j]
        yB = cfg.size
        row0, rows = sh.coop_rows(yB)
        data = self._coop_data[j]
        if data is None or tuple(data.shape) not in ((yB, yB), (rows, yB)):
            raise ValueError(f"rank {self.rank} needs the rows [{row0}, {row0 + rows}) of cooperative facet {j} "
                             f"(shape ({rows}, {yB})) or the whole facet in facet_data[{j}]")
        if not isinstance(data, torch.Tensor):
            data = torch.as_tensor(data)
        block = data[row0 : row0 + rows] if data.shape[0] == yB and rows != yB else data
        block = block.to(device=core.device, dtype=self.dtype)
        in_counts = [rows * 2 * cb.sub[d][1] for d in range(self.world)]
        out_counts = [sh.coop_rows(yB, s)[1] * 2 * cb.sub[self.rank][1] for s in range(self.world)]
        send = torch.empty(sum(in_counts), dtype=self.dtype, device=core.device)
        if rows:
            part = core.prepare_facet_band(block, cfg.off1, cb.band, rows_of=(yB, row0))  # [rows, 2 * half]
            pos = 0
            for d in range(self.world):
                _, h, d2, w = cb.sub[d]
                if h:
                    chunk = send[pos : pos + rows * 2 * h].view(rows, 2 * h)
                    chunk[:, :w].copy_(part[:, d2 : d2 + w])
                    chunk[:, h : h + w].copy_(part[:, cb.half + d2 : cb.half + d2 + w])
                    if w < h:  # (r4 advice) the padding columns of the two parity runs travel too: defined values
                        chunk[:, w:h].zero_()
                        chunk[:, h + w :].zero_()
                pos += rows * 2 * h
        return send, in_counts, out_counts

    def unpack_coop(self, j, recv):
        """hand the assembled band buffer ``[facet rows, my band columns]`` (received chunks in rank order = row
        order) to the cooperative facet's forward object"""
        self._coop_seen.add(j)
        self._coop_ready = self._coop_seen >= set(self.sharding.coop)
        if j not in self._coop:
            return
        b, h, _, _ = self._coop_bands.sub[self.rank]
        fwd = self._coop[j]
        yB = self.facet_configs[j].size
        fwd.BF_Fs_persist = recv.view(1, yB, 2 * h)
        fwd._band = b  # pylint: disable=protected-access

    def prepare_all_facets(self):
        """K1 for the local facets; cooperative facets: K1 on this rank's rows and the exchange of the band rows"""
        if self.local is not None and self.sharding.local_facets:
            self.local.prepare_all_facets()
        if self.sharding.coop and not self._coop_ready and not self._virtual:
            # (virtual ranks of one process: the caller moves the buffers of pack_coop / unpack_coop itself)
            for j in self.sharding.coop:
                send, in_counts, out_counts = self.pack_coop(j)
                self.unpack_coop(j, exchange_blocks(send, in_counts, out_counts, self.group).wait())

    def wave_key(self, sgs):
        """key of the wave ``sgs`` (``off1``, or ``off0`` in the reference schedule)"""
        return int(sgs[0].off1 if self.wave_axis == 1 else sgs[0].off0)

    def subgrids_of(self, sgs, rank=None):
        """indices within the wave ``sgs`` of the subgrids ``rank`` (default: this rank) finishes"""
        return self.sharding.subgrids_of(len(sgs), rank, self.wave_key(sgs))

    def _arrival_cfgs(self, key):
        if not self.sharding.coop:
            return self.arrival_cfgs
        return [self.facet_configs[j] for j in self.sharding.arrival(key)]

    def pack_wave(self, sgs, into=None):
        """Compute this rank's blocks for the subgrids ``sgs`` (one wave: same wave key and size) straight into
        the send buffer ``[dest][local facet][subgrid of dest][m, m]``; returns ``(send, in_counts, out_counts)``.
        ``into``: flat buffer of the right size to fill instead of a fresh one (a slice of a group's send buffer)."""
        torch = _torch()
        core = self.core
        m = core.xM_yN_size
        key = self.wave_key(sgs)
        dests, in_counts, out_counts = forward_layout(self.sharding, len(sgs), m * m, key)
        F_whole = len(self.sharding.local_facets)
        items = self.sharding.items_of(self.rank, key)
        if into is not None and into.numel() != sum(in_counts):
            raise ValueError("destination buffer does not match the wave's send size")
        send = into if into is not None else torch.empty(sum(in_counts), dtype=self.dtype, device=core.device)
        if self.fused and self.wave_axis == 1 and items:
            # one native call for the whole wave: block (f, i) of subgrid i = dests[d][k] goes to
            # chunk_base[d] + f * len(dests[d]) * m^2 + k * m^2
            placed = self.sharding._memo.get(("p", len(sgs), key))  # pylint: disable=protected-access
            if placed is None:
                offs, fstr = [0] * len(sgs), [0] * len(sgs)
                base = 0
                for d, cnt in zip(dests, in_counts):
                    for k, i in enumerate(d):
                        offs[i] = base + k * m * m
                        fstr[i] = len(d) * m * m
                    base += cnt
                placed = self.sharding._memo[("p", len(sgs), key)] = (offs, fstr)  # pylint: disable=protected-access
            offs, fstr = placed
            if F_whole:
                self.local.wave_blocks_into(sgs, send, (offs, fstr))
            for n, j in enumerate(items[F_whole:]):  # cooperative facets this rank owns for this wave: items F_whole + n
                if not self._coop_ready:
                    raise RuntimeError("prepare_all_facets() (the band-row exchange) must run before the first wave")
                shifted = [o + (F_whole + n) * f for o, f in zip(offs, fstr)]
                self._coop[j].wave_blocks_into(sgs, send, (shifted, fstr))
            return send, in_counts, out_counts
        pos = 0
        for d, cnt in zip(dests, in_counts):
            if cnt:
                block = send[pos : pos + cnt].view(F_whole, len(d), m, m)
                self.local.wave_blocks([sgs[i] for i in d], block, transformed=self.fused)
            pos += cnt
        return send, in_counts, out_counts

    def unpack_wave(self, sgs, recv):
        """Finish the subgrids of the wave that this rank owns from the flat receive buffer ``[source][facet of
        source][my subgrid][m, m]``: returns ``(indices within sgs, tensor [S_local, xA, xA] or None)``."""
        from .api import finish_from_blocks  # pylint: disable=import-outside-toplevel

        key = self.wave_key(sgs)
        mine = self.sharding.subgrids_of(len(sgs), key=key)
        if not mine:
            return mine, None
        m = self.core.xM_yN_size
        cfgs = self._arrival_cfgs(key)
        blocks = recv.view(len(cfgs), len(mine), m, m)  # facets in arrival order
        res = finish_from_blocks(self.core, blocks, cfgs, [sgs[i] for i in mine], transformed=self.fused,
                                 placed=int(self.wave_axis == 1 and bool(getattr(self.core, "axis1_first", False))))
        return mine, res

    def start_wave(self, sgs):
        """:py:meth:`pack_wave` + start of the all-to-all; returns a handle for :py:meth:`finish_wave`.
        Starting wave w+1 before finishing wave w overlaps the exchange (RCCL's stream) with the facet-side
        kernels (compute stream)."""
        send, in_counts, out_counts = self.pack_wave(sgs)
        return sgs, exchange_blocks(send, in_counts, out_counts, self.group)

    # -- whole-wave ownership: one balanced all-to-all per GROUP of waves with distinct owners --------------------
    def pack_group(self, waves):
        """``waves``: the waves (lists of subgrid configs) of one ``sharding.wave_groups`` entry, in any order.  Every
        wave goes to its owner as ONE chunk ``[my items][its subgrids][m, m]``: ``(send, in_counts, out_counts)`` of a
        single all-to-all that moves the whole group -- each rank sends to and receives from every other rank."""
        torch = _torch()
        sh = self.sharding
        if sh.wave_rank is None:
            raise ValueError("pack_group needs whole-wave subgrid ownership (whole_waves=True)")
        m2 = self.core.xM_yN_size ** 2
        owner = {}
        for sgs in waves:
            r = sh.wave_rank[self.wave_key(sgs)]
            if r in owner:
                raise ValueError("two waves of a group have the same owner")
            owner[r] = sgs
        in_counts = [len(sh.items_of(self.rank, self.wave_key(owner[d]))) * len(owner[d]) * m2 if d in owner else 0
                     for d in range(self.world)]
        mine = owner.get(self.rank)
        out_counts = [len(sh.items_of(s, self.wave_key(mine))) * len(mine) * m2 if mine is not None else 0
                      for s in range(self.world)]
        send = torch.empty(sum(in_counts), dtype=self.dtype, device=self.core.device)
        pos = 0
        for d in range(self.world):
            if in_counts[d]:
                sub_send, sub_in, _ = self.pack_wave(owner[d], into=send[pos : pos + in_counts[d]])
                assert sub_send.numel() == in_counts[d] and sub_in[d] == in_counts[d]
            pos += in_counts[d]
        return send, in_counts, out_counts

    def unpack_group(self, waves, recv):
        """finish the wave of the group this rank owns: ``(its subgrid configs or None, tensor [S, xA, xA] or None)``"""
        for sgs in waves:
            if self.sharding.wave_rank[self.wave_key(sgs)] == self.rank:
                _, res = self.unpack_wave(sgs, recv)
                return sgs, res
        return None, None

    def start_group(self, waves):
        """:py:meth:`pack_group` + start of the group's all-to-all; handle for :py:meth:`finish_group`"""
        send, in_counts, out_counts = self.pack_group(waves)
        return waves, exchange_blocks(send, in_counts, out_counts, self.group)

    def finish_group(self, handle):
        """wait for the exchange of a started group and :py:meth:`unpack_group`"""
        waves, pending = handle
        return self.unpack_group(waves, pending.wait())

    def finish_wave(self, handle):
        """Wait for the exchange of a started wave and :py:meth:`unpack_wave`."""
        sgs, pending = handle
        return self.unpack_wave(sgs, pending.wait())

    def get_subgrid_wave(self, sgs):
        """start_wave + finish_wave"""
        return self.finish_wave(self.start_wave(sgs))


class DistributedBackward:
    """Facet-sharded ``SwiftlyBackward`` (HIP): subgrid ``i`` of a wave is held
    by rank ``i % world`` (where the forward pass left it), its contributions to
    every facet are computed there (``prepare_and_split_subgrid``, reference
    api_helper.py:115-139) in owner-major facet order -- which makes the result
    the send buffer -- and the mirror all-to-all delivers them to the facets'
    owners, who accumulate (api_helper.py:142-179) and finally finish their
    facets (api_helper.py:182-197).  ``dtype``: complex dtype of the pass; all ranks must agree on it (a rank
    that holds no subgrid of a wave -- with ``balance`` the ranks carrying an extra facet never do -- cannot infer
    it from data, and the all-to-all needs matching element sizes on every rank); default complex64.

    At most ``MAX_IN_FLIGHT`` waves may be started and not yet finished: the send buffer of the fused route is
    one of two alternating workspaces, so :py:meth:`start_wave` makes the compute stream wait for the exchange
    that last read the slot it is about to overwrite.

    **Cooperative facets** (r4; band schedule with a plan, facet count not a multiple of the world size; mirror of
    :class:`DistributedForward`): the contributions of wave ``key`` to a leftover facet go to ``key_owner[key]``, which
    accumulates them into a band accumulator over the columns of ITS waves; :py:meth:`finish` zero-fills the untouched
    columns, moves row blocks to the ranks that own them (one all-to-all: column ranges out, row blocks in, overlapping
    columns of neighbouring ranges added) and runs the contiguous-axis ``finish_facet`` on this rank's rows.  The pieces
    are in ``coop_pieces`` = ``[(facet index, row0, tensor [rows, size])]`` after :py:meth:`finish`."""

    MAX_IN_FLIGHT = 2

    # pylint: disable=too-many-arguments
    def __init__(self, swiftly_config, facet_configs, lru_backward=1, group=None, rank_world=None, wave_axis=0,
                 subgrid_configs=None, dtype=None, cooperative=True, whole_waves=False):
        from .api import SwiftlyBackward  # pylint: disable=import-outside-toplevel
        from .core_hip import band_range  # pylint: disable=import-outside-toplevel

        torch = _torch()
        self.dtype = dtype if dtype is not None else torch.complex64
        if self.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("dtype must be torch.complex64 or torch.complex128")
        self._started = []  # exchanges started and not yet finished, oldest first
        self.group = group
        self.wave_axis = int(wave_axis)
        self.rank, self.world = rank_world if rank_world is not None else _dist_info(group)
        self._virtual = rank_world is not None
        self.config = swiftly_config
        self.core = core = swiftly_config.core
        self.facet_configs = list(facet_configs)
        plan = list(subgrid_configs) if subgrid_configs is not None else None
        keys = None
        if (cooperative and self.wave_axis == 1 and plan and self.world > 1 and len(self.facet_configs) % self.world
                and self.dtype == torch.complex64 and core.band_for_offsets([plan[0].off1]) != (0, core.yN_size)
                and len({c.size for c in self.facet_configs}) == 1):
            keys, band = _plan_wave_keys(core, plan)  # the same wave ranges as DistributedForward
            if band == (0, core.yN_size):
                keys = None
        self.sharding = _shared_sharding(core, len(self.facet_configs), self.rank, self.world, keys,
                                         _wave_sizes(plan, self.wave_axis) if whole_waves else None)
        sh = self.sharding
        # facet owner side: accumulators of the local facets
        self.local = None
        if sh.local_facets or not sh.coop:
            self.local = SwiftlyBackward(
                swiftly_config, [self.facet_configs[j] for j in sh.local_facets], lru_backward=lru_backward,
                wave_axis=self.wave_axis, subgrid_configs=subgrid_configs,
            )
            self.local.dtype = self.dtype
        # subgrid holder side: contributions to ALL facets, owner-major order (cooperative facets last)
        self._base_order = sh.arrival_order + sh.coop
        self.splitter = SwiftlyBackward(
            swiftly_config, [self.facet_configs[j] for j in self._base_order], lru_backward=1
        )
        self.splitter.dtype = self.dtype
        self._splitters = {tuple(self._base_order): self.splitter}
        self._coop, self.coop_pieces = {}, []
        self._coop_band = None
        if sh.coop:
            N, yN, m = core.N, core.yN_size, core.xM_yN_size
            self._coop_band = band_range(N, yN, m, [sg.off1 for sg in plan])  # plain column order (backward layout)
            self._coop_sub = []  # per rank: (first column inside the whole band, columns)
            for r in range(self.world):
                if sh.keys_of[r]:
                    b = band_range(N, yN, m, sh.keys_of[r])
                    self._coop_sub.append(((b[0] - self._coop_band[0]) % yN, b[1]))
                else:
                    self._coop_sub.append((0, 0))
            mine = set(sh.keys_of[self.rank])
            my_plan = [sg for sg in plan if int(sg.off1) in mine]
            if my_plan:
                for j in sh.coop:
                    cb = SwiftlyBackward(swiftly_config, [self.facet_configs[j]], lru_backward=lru_backward, wave_axis=1,
                                         subgrid_configs=my_plan)
                    cb.dtype = self.dtype
                    self._coop[j] = cb

    def wave_key(self, sgs):
        """key of the wave ``sgs`` (``off1`` in the band schedule, else ``off0``)"""
        return int(sgs[0].off1 if self.wave_axis == 1 else sgs[0].off0)

    def subgrids_of(self, sgs, rank=None):
        """indices within the wave ``sgs`` of the subgrids ``rank`` (default: this rank) holds"""
        return self.sharding.subgrids_of(len(sgs), rank, self.wave_key(sgs))

    def pack_wave(self, sgs, subgrids_mine):
        """``sgs``: all subgrid configs of the wave (same size and same ``off0`` -- ``off1`` with ``wave_axis=1``,
        where the facet owners take the strided-axis transform per wave and the contiguous one at the end --
        identical on every rank);
        ``subgrids_mine``: data of the subgrids this rank holds, in the order of
        ``sharding.subgrids_of(len(sgs))``.  Returns ``(send, in_counts, out_counts)``: the contributions of my
        subgrids to all facets, owner-major, flat."""
        torch = _torch()
        core = self.core
        m = core.xM_yN_size
        S = len(sgs)
        sh = self.sharding
        key = self.wave_key(sgs)
        mine = sh.subgrids_of(S, key=key)
        if len(subgrids_mine) != len(mine):
            raise ValueError(f"rank {self.rank} holds {len(mine)} subgrids of this wave, got {len(subgrids_mine)}")
        in_counts, out_counts = backward_layout(sh, S, m * m, key)
        if mine:
            # the contributions come out in the facet order of the splitter = the arrival order of this wave: with
            # cooperative facets that order depends on the rank that owns the wave, so there is one splitter per owner
            order = tuple(sh.arrival(key))
            splitter = self._splitters.get(order)
            if splitter is None:
                from .api import SwiftlyBackward  # pylint: disable=import-outside-toplevel

                splitter = self._splitters[order] = SwiftlyBackward(
                    self.config, [self.facet_configs[j] for j in order], lru_backward=1)
                splitter.dtype = self.dtype
            parts = splitter.wave_contributions([sgs[i] for i in mine], subgrids_mine)
            if parts.dtype != self.dtype:
                raise ValueError(f"subgrids are {parts.dtype}, the pass was declared {self.dtype}")
            send = parts.reshape(-1)
        else:
            send = torch.empty(0, dtype=self.dtype, device=core.device)
        return send, in_counts, out_counts

    def unpack_wave(self, sgs, recv, only_source=None):
        """Accumulate the received contributions (one chunk ``[my facet][subgrid of source][m, m]`` per source
        rank) into this rank's facets.  ``only_source``: ``recv`` holds the chunk of that source only (group exchange)."""
        sh = self.sharding
        key = self.wave_key(sgs)
        items = sh.items_of(self.rank, key)
        F_local = len(items)
        if not F_local:
            return
        F_whole = len(sh.local_facets)
        m = self.core.xM_yN_size
        S = len(sgs)
        chunks = []
        pos = 0
        for r in range(self.world):
            if only_source is not None and r != only_source:
                continue
            idx = sh.subgrids_of(S, r, key)
            cnt = F_local * len(idx) * m * m
            if cnt:
                chunks.append(([sgs[i] for i in idx], recv[pos : pos + cnt].view(F_local, len(idx), m, m)))
            pos += cnt
        wave_key = sgs[0].off1 if self.wave_axis == 1 else sgs[0].off0
        if F_whole:
            self.local.accumulate_chunks(wave_key, [(c, t[:F_whole]) for c, t in chunks])
        for n, j in enumerate(items[F_whole:]):
            self._coop[j].accumulate_chunks(wave_key, [(c, t[F_whole + n : F_whole + n + 1]) for c, t in chunks])

    def start_wave(self, sgs, subgrids_mine):
        """:py:meth:`pack_wave` + start of the mirror all-to-all; returns a handle for :py:meth:`finish_wave`."""
        # the send buffer about to be written may still be read by the exchange started MAX_IN_FLIGHT waves ago
        while len(self._started) >= self.MAX_IN_FLIGHT:
            self._started.pop(0).wait()  # stream-side wait on the collective (no host block for RCCL)
        send, in_counts, out_counts = self.pack_wave(sgs, subgrids_mine)
        pending = exchange_blocks(send, in_counts, out_counts, self.group)
        self._started.append(pending)
        return sgs, pending

    def finish_wave(self, handle):
        """Wait for the exchange and :py:meth:`unpack_wave`."""
        sgs, pending = handle
        recv = pending.wait()
        self._started = [q for q in self._started if q is not pending]
        self.unpack_wave(sgs, recv)

    def add_wave(self, sgs, subgrids_mine):
        """start_wave + finish_wave"""
        self.finish_wave(self.start_wave(sgs, subgrids_mine))

    # -- whole-wave ownership: one balanced all-to-all per GROUP of waves with distinct holders ---------------------
    def pack_group(self, waves, subgrids_mine):
        """``waves``: the waves of one ``sharding.wave_groups`` entry; ``subgrids_mine``: data of ALL subgrids of the wave
        of the group this rank holds (``[]`` if none).  ``(send, in_counts, out_counts)`` of the group's all-to-all:
        this rank sends its wave's contributions to every facet owner and receives, from the holder of every other wave
        of the group, that wave's contributions to its own facets."""
        sh = self.sharding
        if sh.wave_rank is None:
            raise ValueError("pack_group needs whole-wave subgrid ownership (whole_waves=True)")
        m2 = self.core.xM_yN_size ** 2
        holder = {sh.wave_rank[self.wave_key(sgs)]: sgs for sgs in waves}
        if len(holder) != len(waves):
            raise ValueError("two waves of a group have the same holder")
        mine = holder.get(self.rank)
        if mine is not None:
            send, in_counts, _ = self.pack_wave(mine, subgrids_mine)
        else:
            if len(subgrids_mine):
                raise ValueError(f"rank {self.rank} holds no wave of this group")
            send = _torch().empty(0, dtype=self.dtype, device=self.core.device)
            in_counts = [0] * self.world
        out_counts = [len(sh.items_of(self.rank, self.wave_key(holder[s]))) * len(holder[s]) * m2 if s in holder else 0
                      for s in range(self.world)]
        return send, in_counts, out_counts

    def unpack_group(self, waves, recv):
        """accumulate the received contributions of every wave of the group (one chunk per holder) into this rank's facets"""
        sh = self.sharding
        m = self.core.xM_yN_size
        holder = {sh.wave_rank[self.wave_key(sgs)]: sgs for sgs in waves}
        pos = 0
        for s in range(self.world):
            sgs = holder.get(s)
            if sgs is None:
                continue
            cnt = len(sh.items_of(self.rank, self.wave_key(sgs))) * len(sgs) * m * m
            if cnt:
                # the chunk of source s is exactly what unpack_wave expects for a wave whose only holder is s
                self.unpack_wave(sgs, recv[pos : pos + cnt], only_source=s)
            pos += cnt

    def start_group(self, waves, subgrids_mine):
        """:py:meth:`pack_group` + start of the group's all-to-all; handle for :py:meth:`finish_group`"""
        while len(self._started) >= self.MAX_IN_FLIGHT:
            self._started.pop(0).wait()
        send, in_counts, out_counts = self.pack_group(waves, subgrids_mine)
        pending = exchange_blocks(send, in_counts, out_counts, self.group)
        self._started.append(pending)
        return waves, pending

    def finish_group(self, handle):
        """wait for the exchange of a started group and :py:meth:`unpack_group`"""
        waves, pending = handle
        recv = pending.wait()
        self._started = [q for q in self._started if q is not pending]
        self.unpack_group(waves, recv)

    # -- cooperative facets: column ranges out, row blocks in, contiguous-axis finish on this rank's rows -------------
    def pack_coop_finish(self, j):
        """``(send, in_counts, out_counts)`` of the finishing exchange of cooperative facet ``j``: this rank's band
        accumulator ``[facet rows, my columns]`` (untouched columns zero-filled) IS the send buffer -- the row blocks
        of the ranks are consecutive -- and rank ``s`` sends ``[my rows, columns of s]``."""
        torch = _torch()
        core, sh = self.core, self.sharding
        yB = self.facet_configs[j].size
        ncol = self._coop_sub[self.rank][1]
        cb = self._coop.get(j)
        if ncol == 0:
            acc = torch.empty(0, dtype=self.dtype, device=core.device)
        elif cb is None or cb._bands is None:  # pylint: disable=protected-access
            acc = torch.zeros((yB, ncol), dtype=self.dtype, device=core.device)
        else:
            core.band_zero_untouched(cb._bands, cb._touched)  # pylint: disable=protected-access
            acc = cb._bands[0]  # pylint: disable=protected-access
            if tuple(acc.shape) != (yB, ncol):
                raise ValueError("internal: cooperative accumulator does not match the wave range's band")
        in_counts = [sh.coop_rows(yB, d)[1] * ncol for d in range(self.world)]
        rows = sh.coop_rows(yB)[1]
        out_counts = [rows * self._coop_sub[s][1] for s in range(self.world)]
        return acc.reshape(-1), in_counts, out_counts

    def unpack_coop_finish(self, j, recv):
        """assemble ``[my rows, band]`` from the received column ranges (overlaps of neighbouring ranges ADD) and finish
        the rows along the contiguous axis: ``(facet index, row0, tensor [rows, size])``"""
        torch = _torch()
        core, sh = self.core, self.sharding
        cfg = self.facet_configs[j]
        row0, rows = sh.coop_rows(cfg.size)
        if rows == 0:
            return (j, row0, torch.empty((0, cfg.size), dtype=self.dtype, device=core.device))
        buf = torch.zeros((rows, self._coop_band[1]), dtype=self.dtype, device=core.device)
        pos = 0
        for s in range(self.world):
            c0, n = self._coop_sub[s]
            if n:
                buf[:, c0 : c0 + n] += recv[pos : pos + rows * n].view(rows, n)
            pos += rows * n
        out = core.finish_facet_band(buf, self._coop_band, cfg.off1, cfg.size, mask=cfg.mask1)
        return (j, row0, out)

    def finish(self):
        """Finished facets of this rank: ``(global facet indices, list of tensors)``; the rows of cooperative facets
        this rank finished are in ``coop_pieces`` (virtual ranks: the caller moves the buffers of pack_coop_finish /
        unpack_coop_finish itself)."""
        sh = self.sharding
        if sh.coop and not self._virtual:
            self.coop_pieces = []
            for j in sh.coop:
                send, in_counts, out_counts = self.pack_coop_finish(j)
                recv = exchange_blocks(send, in_counts, out_counts, self.group).wait()
                self.coop_pieces.append(self.unpack_coop_finish(j, recv))
        return sh.local_facets, (self.local.finish() if sh.local_facets else [])
