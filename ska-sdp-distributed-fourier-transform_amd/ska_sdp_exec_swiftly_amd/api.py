"""
Streaming API: ``SwiftlyConfig`` / ``SwiftlyForward`` / ``SwiftlyBackward``
with the constructor signatures, method names and semantics of the reference
(src/ska_sdp_exec_swiftly/api.py:107-463), executed on one MI355X per process
instead of a Dask cluster.

What changes relative to the reference, and why:

* ``backend="hip"`` selects :class:`SwiftlyCoreHip` at the reference's own seam
  (api.py:137-143); any other value raises ``ValueError("Unknown SwiFTly
  backend")`` like the reference does.  No Dask client is needed
  (``dask_client`` / ``client`` are accepted and ignored).
* "Tasks" are device tensors: work is enqueued asynchronously on the current
  HIP stream, so ``get_subgrid_task`` returns immediately with a tensor whose
  contents are ready in stream order (the counterpart of a Dask future).
  ``queue_size`` keeps the meaning it has in the reference's ``TaskQueue``
  (api.py:466-522): at most that many submitted tasks may be unfinished; the
  submitting host thread blocks on the oldest one beyond that
  (:class:`TaskQueue`, HIP events instead of ``distributed.wait``).
* facets and all intermediates (``BF_F`` per facet, the per-``off0`` column
  cache of ``lru_forward`` / ``lru_backward`` entries) live in HBM for the
  whole run.
* ``get_subgrid_tasks`` / ``add_new_subgrid_tasks`` (extensions) process a
  whole subgrid column ("wave") per launch sequence; the single-subgrid
  methods are the same code with a wave of one.
* ``backend="numpy"`` (the reference default) is not available here -- this
  package has no CPU path by design -- and raises a ``ValueError`` that says
  so; pass ``backend="hip"`` (the default of THIS package).
"""
# pylint: disable=unused-import
import logging

from . import prefetch as _prefetch_mod
from .backward import SwiftlyBackward
from .config import (FacetConfig, SubgridConfig, SwiftlyConfig, _ChunkConfig, make_full_cover_config,
                     make_full_facet_cover, make_full_subgrid_cover, make_mask_from_slice)
from .forward import (K1_DESCRIPTION, SwiftlyForward, _colacc_from_columns, _facet_grid, _finish_from_colacc,
                      _finish_from_columns, _finish_from_G, finish_from_blocks, preferred_wave_axis, sum_and_finish_wave)
from .ingest import _FacetIngest, _mask_table
from .tasks import DeviceTask, LRUCache, TaskQueue, _torch, _unwrap

__all__ = [
    "FacetConfig",
    "SubgridConfig",
    "SwiftlyConfig",
    "SwiftlyForward",
    "SwiftlyBackward",
    "LRUCache",
    "TaskQueue",
    "DeviceTask",
    "preferred_wave_axis",
    "make_full_facet_cover",
    "make_full_subgrid_cover",
    "make_full_cover_config",
    "make_mask_from_slice",
    "sum_and_finish_wave",
    "finish_from_blocks",
]

log = logging.getLogger("fourier-logger")

# The live values of the prefetch knobs (prefetch.py reads them from HERE; tests and A/B runs switch them at run time):
# SWIFTLY_PREFETCH=0 turns the planned-wave prefetch off, SWIFTLY_PREFETCH_DEPTH = planned waves K2 may run ahead (2),
# SWIFTLY_CHAIN_K2=0 makes every prefetched K2 fork its chunk streams again
_PREFETCH = _prefetch_mod.PREFETCH_DEFAULT
_PREFETCH_DEPTH = _prefetch_mod.PREFETCH_DEPTH_DEFAULT
_CHAIN_K2 = _prefetch_mod.CHAIN_K2_DEFAULT
