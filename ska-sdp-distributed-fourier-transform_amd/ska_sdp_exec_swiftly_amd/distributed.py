"""
Multi-GPU forward pass: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm, xGMI underneath), facets block-cyclically sharded over ranks,
subgrids of a wave owned round-robin, and ONE exchange step per wave: the
per-(facet, subgrid) contribution shuffle that the reference leaves to Dask's
worker-to-worker transfers (reference api.py:263-277) becomes an
``all_to_all_single`` of ``[m, m]`` blocks.

The exchange logic is independent of who computes the blocks: it takes a
``make_contributions(subgrid_configs) -> [F_local, S, m, m]`` callable and a
``finish(contrib[F, S_local, m, m], subgrid_configs_local)`` callable, so the
partitioning / ordering is covered by world_size-2 gloo tests on CPU while the
product wires in the HIP kernels (:class:`DistributedForward`).
"""
import numpy

__all__ = ["FacetSharding", "exchange_contributions", "start_exchange", "DistributedForward"]


def _torch():
    import torch  # pylint: disable=import-outside-toplevel

    return torch


class FacetSharding:
    """Who owns what.  Facet ``j`` lives on rank ``j % world``; subgrid number
    ``i`` of a wave is finished on rank ``i % world``."""

    def __init__(self, n_facets, rank, world):
        self.n_facets, self.rank, self.world = n_facets, rank, world
        self.facets_of = [[j for j in range(n_facets) if j % world == r] for r in range(world)]
        self.local_facets = self.facets_of[rank]
        # facet order after concatenating received blocks in source-rank order
        self.arrival_order = [j for r in range(world) for j in self.facets_of[r]]
        self.to_global = numpy.argsort(self.arrival_order)  # arrival position of global facet j

    def subgrids_of(self, n_subgrids, rank=None):
        """indices (within the wave) of the subgrids rank finishes"""
        rank = self.rank if rank is None else rank
        return list(range(rank, n_subgrids, self.world))


class _Exchange:
    """In-flight all-to-all of one wave (``wait()`` returns the re-ordered contributions)."""

    def __init__(self, work, recv, send, sharding, n_mine, blk):
        self.work, self.recv, self.send = work, recv, send  # send is kept alive until the collective is done
        self.sharding, self.n_mine, self.blk = sharding, n_mine, blk

    def wait(self):
        torch = _torch()
        if self.work is not None:
            self.work.wait()
        arrived = self.recv.reshape(self.sharding.n_facets, self.n_mine, *self.blk)  # source-rank (arrival) order
        return arrived[torch.as_tensor(self.sharding.to_global, device=arrived.device)]


class _Done:
    def __init__(self, value):
        self.value = value

    def wait(self):
        return self.value


def start_exchange(contrib_local, sharding, group=None, async_op=True):
    """Issue the all-to-all of one wave and return a handle whose ``wait()`` gives
    ``[F, S_local, m, m]`` (all facets in global order, the subgrids this rank owns).

    With ``async_op`` the collective runs on RCCL's own stream: kernels launched
    afterwards on the compute stream (the next wave's column / extract kernels)
    overlap with it, and only ``wait()`` orders the compute stream behind it.
    """
    torch = _torch()
    dist = torch.distributed
    world = sharding.world
    S = contrib_local.shape[1]
    blk = tuple(contrib_local.shape[2:])
    mine = sharding.subgrids_of(S)
    if world == 1:
        return _Done(contrib_local)
    # send buffer: for every destination rank the blocks [F_local, S_dest, m, m]
    pieces = [contrib_local[:, sharding.subgrids_of(S, r)].reshape(-1) for r in range(world)]
    send = torch.cat(pieces)
    nblk = int(numpy.prod(blk))
    in_split = [p.numel() for p in pieces]
    out_split = [len(sharding.facets_of[r]) * len(mine) * nblk for r in range(world)]
    recv = torch.empty(sum(out_split), dtype=contrib_local.dtype, device=contrib_local.device)
    if contrib_local.is_complex():
        # RCCL has no complex type: ship as interleaved reals
        work = dist.all_to_all_single(
            torch.view_as_real(recv).reshape(-1),
            torch.view_as_real(send).reshape(-1),
            [2 * n for n in out_split],
            [2 * n for n in in_split],
            group=group,
            async_op=async_op,
        )
    else:
        work = dist.all_to_all_single(recv, send, out_split, in_split, group=group, async_op=async_op)
    return _Exchange(work if async_op else None, recv, send, sharding, len(mine), blk)


def exchange_contributions(contrib_local, sharding, group=None):
    """Blocking all-to-all of one wave.

    :param contrib_local: ``[F_local, S, m, m]`` contributions of this rank's
        facets to all ``S`` subgrids of the wave
    :return: ``[F, S_local, m, m]`` contributions of ALL facets (global facet
        order) to the subgrids this rank owns (``sharding.subgrids_of(S)``)
    """
    return start_exchange(contrib_local, sharding, group, async_op=False).wait()


class DistributedForward:
    """Facet-sharded ``SwiftlyForward`` (HIP).  Every rank constructs it with
    the FULL list of facet configs but only the data of its own facets
    (``facet_data[j]`` for ``j in sharding.local_facets``; other entries are
    ignored and may be ``None``)."""

    def __init__(self, swiftly_config, facet_configs, facet_data, lru_forward=1, group=None, subgrid_configs=None):
        from .api import SwiftlyForward  # pylint: disable=import-outside-toplevel

        torch = _torch()
        dist = torch.distributed
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.config = swiftly_config
        self.facet_configs = list(facet_configs)
        self.sharding = FacetSharding(len(self.facet_configs), self.rank, self.world)
        local = self.sharding.local_facets
        self.local = SwiftlyForward(
            swiftly_config,
            [(self.facet_configs[j], facet_data[j]) for j in local],
            lru_forward=lru_forward,
            subgrid_configs=subgrid_configs,
        )

    def start_wave(self, sgs):
        """Compute this rank's contributions to the subgrids ``sgs`` (same
        ``off0`` / ``size``) and start their all-to-all; returns a handle for
        :py:meth:`finish_wave`.  Starting wave w+1 before finishing wave w
        overlaps the exchange with the column and extract kernels."""
        torch = _torch()
        if self.sharding.local_facets:
            contrib = self.local.wave_contributions(sgs)
        else:
            core = self.config.core
            m = core.xM_yN_size
            contrib = torch.empty((0, len(sgs), m, m), dtype=self.local.dtype, device=core.device)
        return sgs, start_exchange(contrib, self.sharding, self.group)

    def finish_wave(self, handle):
        """Finish the subgrids of a started wave that this rank owns: returns
        ``(indices within sgs, tensor [S_local, xA, xA] or None)``."""
        from .api import sum_and_finish_wave  # pylint: disable=import-outside-toplevel

        sgs, exch = handle
        mine = self.sharding.subgrids_of(len(sgs))
        allc = exch.wait()
        if not mine:
            return mine, None
        res = sum_and_finish_wave(self.config.core, allc, self.facet_configs, [sgs[i] for i in mine])
        return mine, res

    def get_subgrid_wave(self, sgs):
        """start_wave + finish_wave"""
        return self.finish_wave(self.start_wave(sgs))
