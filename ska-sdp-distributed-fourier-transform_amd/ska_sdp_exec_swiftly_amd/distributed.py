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
    subgrid-side work (``balance=False``: every rank takes subgrids)."""

    def __init__(self, n_facets, rank, world, balance=True):
        self.n_facets, self.rank, self.world = n_facets, rank, world
        self.facets_of = [[j for j in range(n_facets) if j % world == r] for r in range(world)]
        self.local_facets = self.facets_of[rank]
        counts = [len(f) for f in self.facets_of]
        if balance and min(counts) < max(counts):
            self.subgrid_ranks = [r for r in range(world) if counts[r] == min(counts)]
        else:
            self.subgrid_ranks = list(range(world))
        # facet order after concatenating received blocks in source-rank order (= owner-major order)
        self.arrival_order = [j for r in range(world) for j in self.facets_of[r]]
        self.to_global = numpy.argsort(self.arrival_order)  # arrival position of global facet j

    def subgrids_of(self, n_subgrids, rank=None):
        """indices (within the wave) of the subgrids of ``rank``"""
        rank = self.rank if rank is None else rank
        if rank not in self.subgrid_ranks:
            return []
        return list(range(self.subgrid_ranks.index(rank), n_subgrids, len(self.subgrid_ranks)))


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


def forward_layout(sharding, n_subgrids, blk):
    """Element counts of the forward exchange: (per-destination subgrid index lists, in_counts, out_counts)."""
    F_local = len(sharding.local_facets)
    mine = sharding.subgrids_of(n_subgrids)
    dests = [sharding.subgrids_of(n_subgrids, r) for r in range(sharding.world)]
    in_counts = [F_local * len(d) * blk for d in dests]
    out_counts = [len(sharding.facets_of[r]) * len(mine) * blk for r in range(sharding.world)]
    return dests, in_counts, out_counts


def backward_layout(sharding, n_subgrids, blk):
    """Element counts of the backward exchange (subgrid holder -> facet owner)."""
    mine = sharding.subgrids_of(n_subgrids)
    F_local = len(sharding.local_facets)
    in_counts = [len(sharding.facets_of[r]) * len(mine) * blk for r in range(sharding.world)]
    out_counts = [F_local * len(sharding.subgrids_of(n_subgrids, r)) * blk for r in range(sharding.world)]
    return in_counts, out_counts


def exchange_blocks(send, in_counts, out_counts, group=None, async_op=True):
    """Start the all-to-all of an already laid-out send buffer; ``wait()`` returns the flat receive buffer."""
    torch = _torch()
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or len(in_counts) == 1:
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
    agree; a rank without facets cannot infer it)."""

    def __init__(self, swiftly_config, facet_configs, facet_data, lru_forward=1, group=None, subgrid_configs=None,
                 wave_axis=None, dtype=None, rank_world=None):
        from .api import SwiftlyForward, preferred_wave_axis  # pylint: disable=import-outside-toplevel

        torch = _torch()
        self.group = group
        # rank_world=(rank, world) overrides the process group: "virtual ranks" of one process, used with
        # pack_wave / unpack_wave and a caller-provided exchange (tests of the multi-rank layouts on one GPU)
        self.rank, self.world = rank_world if rank_world is not None else _dist_info(group)
        self.config = swiftly_config
        self.core = swiftly_config.core
        self.facet_configs = list(facet_configs)
        self.sharding = FacetSharding(len(self.facet_configs), self.rank, self.world)
        local = self.sharding.local_facets
        self.dtype = dtype if dtype is not None else torch.complex64
        if wave_axis is None:
            wave_axis = preferred_wave_axis(swiftly_config, self.dtype, n_facets=len(self.facet_configs))
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
        self.wave_axis = wave_axis
        self.arrival_cfgs = [self.facet_configs[j] for j in self.sharding.arrival_order]
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

    def prepare_all_facets(self):
        """K1 for the local facets"""
        if self.sharding.local_facets:
            self.local.prepare_all_facets()

    def pack_wave(self, sgs):
        """Compute this rank's blocks for the subgrids ``sgs`` (one wave: same wave key and size) straight into
        the send buffer ``[dest][local facet][subgrid of dest][m, m]``; returns ``(send, in_counts, out_counts)``."""
        torch = _torch()
        core = self.core
        m = core.xM_yN_size
        dests, in_counts, out_counts = forward_layout(self.sharding, len(sgs), m * m)
        F_local = len(self.sharding.local_facets)
        send = torch.empty(sum(in_counts), dtype=self.dtype, device=core.device)
        if self.fused and self.wave_axis == 1 and F_local:
            # one native call for the whole wave: block (f, i) of subgrid i = dests[d][k] goes to
            # chunk_base[d] + f * len(dests[d]) * m^2 + k * m^2
            offs, fstr = [0] * len(sgs), [0] * len(sgs)
            base = 0
            for d, cnt in zip(dests, in_counts):
                for k, i in enumerate(d):
                    offs[i] = base + k * m * m
                    fstr[i] = len(d) * m * m
                base += cnt
            self.local.wave_blocks_into(sgs, send, (offs, fstr))
            return send, in_counts, out_counts
        pos = 0
        for d, cnt in zip(dests, in_counts):
            if cnt:
                block = send[pos : pos + cnt].view(F_local, len(d), m, m)
                self.local.wave_blocks([sgs[i] for i in d], block, transformed=self.fused)
            pos += cnt
        return send, in_counts, out_counts

    def unpack_wave(self, sgs, recv):
        """Finish the subgrids of the wave that this rank owns from the flat receive buffer ``[source][facet of
        source][my subgrid][m, m]``: returns ``(indices within sgs, tensor [S_local, xA, xA] or None)``."""
        from .api import finish_from_blocks  # pylint: disable=import-outside-toplevel

        mine = self.sharding.subgrids_of(len(sgs))
        if not mine:
            return mine, None
        m = self.core.xM_yN_size
        blocks = recv.view(len(self.facet_configs), len(mine), m, m)  # facets in arrival order
        res = finish_from_blocks(self.core, blocks, self.arrival_cfgs, [sgs[i] for i in mine], transformed=self.fused)
        return mine, res

    def start_wave(self, sgs):
        """:py:meth:`pack_wave` + start of the all-to-all; returns a handle for :py:meth:`finish_wave`.
        Starting wave w+1 before finishing wave w overlaps the exchange (RCCL's stream) with the facet-side
        kernels (compute stream)."""
        send, in_counts, out_counts = self.pack_wave(sgs)
        return sgs, exchange_blocks(send, in_counts, out_counts, self.group)

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
    that last read the slot it is about to overwrite."""

    MAX_IN_FLIGHT = 2

    # pylint: disable=too-many-arguments
    def __init__(self, swiftly_config, facet_configs, lru_backward=1, group=None, rank_world=None, wave_axis=0,
                 subgrid_configs=None, dtype=None):
        from .api import SwiftlyBackward  # pylint: disable=import-outside-toplevel

        torch = _torch()
        self.dtype = dtype if dtype is not None else torch.complex64
        if self.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("dtype must be torch.complex64 or torch.complex128")
        self._started = []  # exchanges started and not yet finished, oldest first
        self.group = group
        self.wave_axis = int(wave_axis)
        self.rank, self.world = rank_world if rank_world is not None else _dist_info(group)
        self.config = swiftly_config
        self.core = swiftly_config.core
        self.facet_configs = list(facet_configs)
        self.sharding = FacetSharding(len(self.facet_configs), self.rank, self.world)
        # facet owner side: accumulators of the local facets
        self.local = SwiftlyBackward(
            swiftly_config, [self.facet_configs[j] for j in self.sharding.local_facets], lru_backward=lru_backward,
            wave_axis=self.wave_axis, subgrid_configs=subgrid_configs,
        )
        # subgrid holder side: contributions to ALL facets, owner-major order
        self.splitter = SwiftlyBackward(
            swiftly_config, [self.facet_configs[j] for j in self.sharding.arrival_order], lru_backward=1
        )
        self.local.dtype = self.splitter.dtype = self.dtype

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
        mine = self.sharding.subgrids_of(S)
        if len(subgrids_mine) != len(mine):
            raise ValueError(f"rank {self.rank} holds {len(mine)} subgrids of this wave, got {len(subgrids_mine)}")
        in_counts, out_counts = backward_layout(self.sharding, S, m * m)
        if mine:
            send = self.splitter.wave_contributions([sgs[i] for i in mine], subgrids_mine).reshape(-1)
            if send.dtype != self.dtype:
                raise ValueError(f"subgrids are {send.dtype}, the pass was declared {self.dtype}")
        else:
            send = torch.empty(0, dtype=self.dtype, device=core.device)
        return send, in_counts, out_counts

    def unpack_wave(self, sgs, recv):
        """Accumulate the received contributions (one chunk ``[my facet][subgrid of source][m, m]`` per source
        rank) into this rank's facets."""
        F_local = len(self.sharding.local_facets)
        if not F_local:
            return
        m = self.core.xM_yN_size
        S = len(sgs)
        chunks = []
        pos = 0
        for r in range(self.world):
            idx = self.sharding.subgrids_of(S, r)
            cnt = F_local * len(idx) * m * m
            if cnt:
                chunks.append(([sgs[i] for i in idx], recv[pos : pos + cnt].view(F_local, len(idx), m, m)))
            pos += cnt
        self.local.accumulate_chunks(sgs[0].off1 if self.wave_axis == 1 else sgs[0].off0, chunks)

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

    def finish(self):
        """Finished facets of this rank: ``(global facet indices, list of tensors)``."""
        return self.sharding.local_facets, (self.local.finish() if self.sharding.local_facets else [])
