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

__all__ = ["FacetSharding", "exchange_contributions", "DistributedForward"]


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


def exchange_contributions(contrib_local, sharding, group=None):
    """All-to-all of one wave.

    :param contrib_local: ``[F_local, S, m, m]`` contributions of this rank's
        facets to all ``S`` subgrids of the wave
    :return: ``[F, S_local, m, m]`` contributions of ALL facets (global facet
        order) to the subgrids this rank owns (``sharding.subgrids_of(S)``)
    """
    torch = _torch()
    dist = torch.distributed
    world, rank = sharding.world, sharding.rank
    F_local, S = contrib_local.shape[0], contrib_local.shape[1]
    blk = contrib_local.shape[2:]
    mine = sharding.subgrids_of(S)
    if world == 1:
        return contrib_local
    # send buffer: for every destination rank the blocks [F_local, S_dest, m, m]
    pieces = [contrib_local[:, sharding.subgrids_of(S, r)].reshape(-1) for r in range(world)]
    send = torch.cat(pieces)
    nblk = int(numpy.prod(blk))
    in_split = [p.numel() for p in pieces]
    out_split = [len(sharding.facets_of[r]) * len(mine) * nblk for r in range(world)]
    recv = torch.empty(sum(out_split), dtype=contrib_local.dtype, device=contrib_local.device)
    if contrib_local.is_complex():
        # RCCL has no complex type: ship as interleaved reals
        dist.all_to_all_single(
            torch.view_as_real(recv).reshape(-1),
            torch.view_as_real(send).reshape(-1),
            [2 * n for n in out_split],
            [2 * n for n in in_split],
            group=group,
        )
    else:
        dist.all_to_all_single(recv, send, out_split, in_split, group=group)
    arrived = recv.reshape(sharding.n_facets, len(mine), *blk)  # source-rank (arrival) order
    return arrived[torch.as_tensor(sharding.to_global, device=arrived.device)]


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

    def get_subgrid_wave(self, sgs):
        """Finish the subgrids of ``sgs`` (same ``off0`` / ``size``) this rank
        owns: returns ``(indices within sgs, tensor [S_local, xA, xA])``."""
        from .api import sum_and_finish_wave  # pylint: disable=import-outside-toplevel

        torch = _torch()
        mine = self.sharding.subgrids_of(len(sgs))
        if self.sharding.local_facets:
            contrib = self.local.wave_contributions(sgs)
        else:
            core = self.config.core
            m = core.xM_yN_size
            contrib = torch.empty((0, len(sgs), m, m), dtype=self.local.dtype, device=core.device)
        allc = exchange_contributions(contrib, self.sharding, self.group)
        if not mine:
            return mine, None
        res = sum_and_finish_wave(self.config.core, allc, self.facet_configs, [sgs[i] for i in mine])
        return mine, res
