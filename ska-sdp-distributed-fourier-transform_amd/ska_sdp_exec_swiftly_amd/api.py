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
import logging
import os

import numpy

from .core_hip import SwiftlyCoreHip, band_range

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


def make_mask_from_slice(slice_list, mask_size):
    """0/1 float vector that is 1 inside the given slices (reference
    api_helper.py:243-253)."""
    mask = numpy.zeros((mask_size,))
    for piece in slice_list:
        mask[piece] = 1
    return mask


class _ChunkConfig:
    """Offsets, size and masks of one facet or subgrid (reference
    api.py:39-104).  A mask may be an array, ``None`` or ``[[slices], size]``."""

    def __init__(self, off0, off1, size, mask0=None, mask1=None):
        self.off0 = off0
        self.off1 = off1
        self.size = size
        self._mask0 = mask0
        self._mask1 = mask1

    @staticmethod
    def _expand(mask):
        if isinstance(mask, list):
            return make_mask_from_slice(mask[0], mask[1])
        return mask

    @property
    def mask0(self):
        """mask along axis 0"""
        return self._expand(self._mask0)

    @property
    def mask1(self):
        """mask along axis 1"""
        return self._expand(self._mask1)


class FacetConfig(_ChunkConfig):
    """Facet configuration (reference api.py:39-70)"""


class SubgridConfig(_ChunkConfig):
    """Subgrid configuration (reference api.py:73-104)"""


def make_full_cover_config(N, chunk_size, class_name):
    """Cover the N x N plane with ``chunk_size`` pieces at multiples of
    ``chunk_size``; where neighbours overlap (also across the wrap-around) the
    masks hand each pixel to exactly one piece by cutting half way between the
    two offsets (reference api_helper.py:213-240)."""
    count = -(-N // chunk_size)
    offsets = [chunk_size * i for i in range(count)]
    cuts = [(offsets[i] + (offsets[i + 1] if i + 1 < count else N + offsets[0])) // 2 for i in range(count)]
    spans = []
    for i, off in enumerate(offsets):
        lo = (cuts[i - 1] - off + chunk_size // 2) % N
        hi = cuts[i] - off + chunk_size // 2
        spans.append((lo, hi))
    return [
        class_name(o0, o1, chunk_size, [[slice(*spans[i0])], chunk_size], [[slice(*spans[i1])], chunk_size])
        for i0, o0 in enumerate(offsets)
        for i1, o1 in enumerate(offsets)
    ]


def make_full_subgrid_cover(swiftlyconfig):
    """Subgrid configs covering the whole grid (reference api.py:593-601)"""
    return make_full_cover_config(swiftlyconfig.image_size, swiftlyconfig.max_subgrid_size, SubgridConfig)


def make_full_facet_cover(swiftlyconfig):
    """Facet configs covering the whole image (reference api.py:604-612)"""
    return make_full_cover_config(swiftlyconfig.image_size, swiftlyconfig.max_facet_size, FacetConfig)


class SwiftlyConfig:
    """SwiFTly parameters + the core that implements them (reference
    api.py:107-214)."""

    # pylint: disable=too-many-arguments,too-many-instance-attributes
    def __init__(
        self, W, fov, N, yB_size, yN_size, xA_size, xM_size, dask_client=None, backend="hip", column_precision=None,
        **_other_args
    ):
        self._W = W
        self._fov = fov
        self._N = N
        self._yB_size = yB_size
        self._yN_size = yN_size
        self._xA_size = xA_size
        self._xM_size = xM_size
        self.dask_client = dask_client  # unused: there is no Dask in this backend
        if backend == "hip":
            self._core = SwiftlyCoreHip(W, N, xM_size, yN_size, column_precision=column_precision)
        elif backend in ("numpy", "ska_sdp_func"):
            # reference api.py:137-141 -- those cores live in the reference package; this one is GPU only
            raise ValueError(
                f"SwiFTly backend {backend!r} is provided by ska_sdp_exec_swiftly itself; "
                "ska_sdp_exec_swiftly_amd only implements backend='hip' (no CPU fallback)"
            )
        else:
            raise ValueError(f"Unknown SwiFTly backend: {backend}")
        # the reference wraps a scattered core in dask.delayed (api.py:145-147)
        self.core_task = self._core

    @property
    def core(self):
        """the SwiftlyCoreHip instance"""
        return self._core

    @property
    def image_size(self):
        """Size of the entire (virtual) image in pixels"""
        return self._N

    @property
    def max_facet_size(self):
        """Maximum size of a facet in pixels"""
        return self._yB_size

    @property
    def max_subgrid_size(self):
        """Maximum size of a subgrid in pixels"""
        return self._xA_size

    @property
    def pswf_parameter(self):
        """PSWF window parameter W"""
        return self._W

    @property
    def internal_facet_size(self):
        """Padded facet size used internally"""
        return self._yN_size

    @property
    def internal_subgrid_size(self):
        """Padded subgrid size used internally"""
        return self._xM_size

    @property
    def facet_off_step(self):
        """All facet offsets must be divisible by this"""
        return self._core.facet_off_step

    @property
    def subgrid_off_step(self):
        """All subgrid offsets must be divisible by this"""
        return self._core.subgrid_off_step


class LRUCache:
    """Least-recently-used cache with the interface of reference
    api.py:525-590: ``get`` refreshes, ``set`` returns the evicted
    ``(key, value)`` or ``(None, None)``, ``pop_all`` drains oldest first."""

    def __init__(self, cache_size):
        self.cache_size = cache_size
        self._items = {}  # insertion order == recency order

    def get(self, key):
        """value or None; marks the key most recently used"""
        if key not in self._items:
            return None
        val = self._items.pop(key)
        self._items[key] = val
        return val

    def set(self, key, value):
        """insert / refresh; returns evicted (key, value) or (None, None)"""
        self._items.pop(key, None)
        self._items[key] = value
        if len(self._items) <= self.cache_size:
            return None, None
        old_key = next(iter(self._items))
        return old_key, self._items.pop(old_key)

    def pop_all(self):
        """yield and remove all entries, least recently used first"""
        while self._items:
            old_key = next(iter(self._items))
            yield old_key, self._items.pop(old_key)


class TaskQueue:
    """Bounded queue of in-flight tasks (reference api.py:466-522).

    The reference submits Dask tasks and, once ``max_task`` of them are
    unfinished, blocks in ``distributed.wait(..., FIRST_COMPLETED)``.  Here a
    task is a device tensor whose producing kernels have been enqueued on a HIP
    stream; "finished" means a HIP event recorded right after them has
    completed.  ``process`` records such an event per task and, while
    ``max_task`` or more are unfinished, blocks the submitting host thread on
    the OLDEST one (stream order makes the oldest the first to complete), so
    the host never runs more than ``max_task`` tasks ahead of the GPU and the
    scratch memory those tasks pin stays bounded.

    :param max_task: queue size
    :param event_factory: callable returning an object with ``record()``,
        ``query() -> bool`` and ``synchronize()`` (default: ``torch.cuda.Event``)
    """

    def __init__(self, max_task, event_factory=None):
        self.max_task = max(1, int(max_task))
        self.task_queue = []  # [(event, task)], oldest first
        self._event_factory = event_factory

    def _new_event(self):
        if self._event_factory is not None:
            return self._event_factory()
        return _torch().cuda.Event()

    def empty_done(self):
        """drop finished tasks from the queue (reference api.py:497-509)"""
        self.task_queue = [(ev, task) for ev, task in self.task_queue if not ev.query()]

    def process(self, task_list):
        """submit tasks; blocks while the queue is full (reference api.py:478-495)"""
        for task in task_list:
            while len(self.task_queue) >= self.max_task:
                self.task_queue[0][0].synchronize()
                self.empty_done()
            ev = self._new_event()
            ev.record()
            self.task_queue.append((ev, task))
        return task_list

    def wait_all_done(self):
        """block until every submitted task has finished (reference api.py:511-522)"""
        for ev, _ in self.task_queue:
            ev.synchronize()
        self.empty_done()
        if self.task_queue:
            raise RuntimeError("Some tasks did not finish")


# tuning knob: SWIFTLY_PREFETCH=0 turns the planned-wave prefetch of SwiftlyForward off (A/B runs)
_PREFETCH = os.environ.get("SWIFTLY_PREFETCH", "1") != "0"
# how many planned waves K2 may run ahead of the wave being served (SwiftlyForward._prefetch_wave): 1 = the r4 schedule
_PREFETCH_DEPTH = max(1, int(os.environ.get("SWIFTLY_PREFETCH_DEPTH", "2")))
# SWIFTLY_CHAIN_K2=0: every prefetched K2 forks its chunk streams behind the side stream again (A/B runs)
_CHAIN_K2 = os.environ.get("SWIFTLY_CHAIN_K2", "1") != "0"


def _torch():
    import torch  # pylint: disable=import-outside-toplevel

    return torch


class DeviceTask:
    """Handle of one asynchronous result: the counterpart of the ``dask.delayed`` / future objects the reference's
    streaming classes hand out (api.py:238-253, 347-400).  It wraps the device tensor whose producing kernels have
    been enqueued plus a HIP event recorded right behind them.

    * ``tensor`` -- the device tensor, valid in stream order (pass it, or the task itself, to
      ``SwiftlyBackward.add_new_subgrid_task``: no synchronisation happens);
    * ``done()`` -- has the GPU finished it?  ``wait()`` blocks the host until it has;
    * ``compute()`` / ``result()`` -- host copy as a numpy array (what ``Delayed.compute()`` / ``Future.result()``
      give a caller of the reference); ``numpy.asarray(task)`` works too.

    ``SwiftlyForward(..., delayed=True)`` / ``SwiftlyBackward(..., delayed=True)`` return these instead of bare tensors.
    """

    def __init__(self, tensor):
        self.tensor = tensor
        self._event = None
        if getattr(tensor, "is_cuda", False):
            # on the current stream of the TENSOR's device (where the producing kernels were enqueued), which need not
            # be the process's current device
            torch = _torch()
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream(tensor.device))

    def done(self):
        """True once the producing kernels have completed"""
        return self._event is None or self._event.query()

    def wait(self):
        """block the calling host thread until the result is complete"""
        if self._event is not None:
            self._event.synchronize()
        return self

    def compute(self):
        """host copy of the result (numpy)"""
        self.wait()
        return self.tensor.cpu().numpy()

    result = compute

    def __array__(self, dtype=None, copy=None):
        arr = self.compute()
        return arr.astype(dtype) if dtype is not None else arr

    @property
    def shape(self):
        """shape of the result"""
        return tuple(self.tensor.shape)

    @property
    def dtype(self):
        """torch dtype of the result"""
        return self.tensor.dtype


def _unwrap(data):
    """the device tensor of a :class:`DeviceTask`, anything else unchanged"""
    return data.tensor if isinstance(data, DeviceTask) else data


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


_MASK_CACHE = {}
_MASK_CACHE_MAX = 4096


def _mask_cache_put(key, value):
    if len(_MASK_CACHE) >= _MASK_CACHE_MAX:
        _MASK_CACHE.pop(next(iter(_MASK_CACHE)))
    _MASK_CACHE[key] = value


def _mask_table(core, configs, which, size, cdtype):
    """[len(configs), size] real device table of the masks (ones where a config
    has no mask), or None when no config has one.  Tables are cached per
    (device, precision, mask contents): an upload from pageable host memory is
    ordered behind everything already queued on the stream, i.e. it would stall
    the host once per wave."""
    torch = _torch()
    # fast path: the same config objects as last time (waves are re-requested every pass); the cache entry
    # keeps the configs alive, so their ids cannot be recycled while it exists
    idkey = (str(core.device), str(cdtype), which, size, tuple(id(c) for c in configs))
    hit = _MASK_CACHE.get(idkey)
    if hit is not None:
        return hit[1]
    masks = [getattr(c, which) for c in configs]
    if all(m is None for m in masks):
        _mask_cache_put(idkey, (tuple(configs), None))
        return None
    tab = numpy.ones((len(configs), size))
    for i, m in enumerate(masks):
        if m is not None:
            tab[i] = numpy.asarray(m, dtype=float)
    key = (str(core.device), str(cdtype), tab.shape, tab.tobytes())
    hit = _MASK_CACHE.get(key)
    if hit is None:
        rdtype = torch.float32 if cdtype == torch.complex64 else torch.float64
        hit = (None, torch.from_numpy(tab).to(device=core.device, dtype=rdtype).contiguous())
        _mask_cache_put(key, hit)
    _mask_cache_put(idkey, (tuple(configs), hit[1]))
    return hit[1]


class _FacetIngest:
    """Host -> device facet upload overlapped with compute.

    Device tensors are used in place.  Host (numpy) facets are uploaded LAZILY on a separate HIP stream in slabs of
    ``SLAB`` bytes through two pinned staging buffers (pageable memory cannot be copied asynchronously, and pinning
    a whole 4 GB facet would cost more than the copy): ``ready(j)`` makes the CURRENT stream wait for facet ``j``
    (a stream-side wait) and returns the tensor; ``prefetch(j)`` starts the upload of facet ``j`` -- the streaming
    classes call it for facet j+1 right after queueing the full-facet transform of facet j, so the transfer runs
    under that kernel."""

    SLAB = 128 << 20

    def __init__(self, core):
        self.core = core
        self.host, self.tensors, self.events = [], [], []
        self._stream = None
        self._staging = None
        self._staging_free = None

    def add(self, data):
        """register a facet; returns (dtype, shape, is_row_major)"""
        torch = _torch()
        if isinstance(data, torch.Tensor):
            ten, _ = self.core._as_device(data)  # pylint: disable=protected-access
            self.host.append(None)
            self.tensors.append(ten)
        else:
            arr = numpy.asarray(data)
            if not numpy.iscomplexobj(arr):
                arr = arr.astype(numpy.complex64 if arr.dtype == numpy.float32 else numpy.complex128)
            elif arr.dtype not in (numpy.complex64, numpy.complex128):
                arr = arr.astype(numpy.complex128)
            self.host.append(numpy.ascontiguousarray(arr))
            self.tensors.append(None)
        self.events.append(None)
        j = len(self.tensors) - 1
        src = self.tensors[j] if self.tensors[j] is not None else self.host[j]
        tdt = src.dtype if self.tensors[j] is not None else (
            torch.complex64 if src.dtype == numpy.complex64 else torch.complex128
        )
        row_major = src.stride(-1) == 1 if self.tensors[j] is not None else True
        return tdt, tuple(src.shape), row_major

    def prefetch(self, j):
        """start the upload of facet ``j`` (no-op for device facets / out of range / already started)"""
        torch = _torch()
        if j < 0 or j >= len(self.tensors) or self.tensors[j] is not None:
            return
        core = self.core
        arr = self.host[j]
        tdt = torch.complex64 if arr.dtype == numpy.complex64 else torch.complex128
        dev = torch.empty(arr.shape, dtype=tdt, device=core.device)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=core.device)
            self._staging = [torch.empty(self.SLAB, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._staging_free = [None, None]
        src = torch.from_numpy(arr.reshape(-1).view(numpy.uint8))
        dst = dev.reshape(-1).view(torch.uint8)
        nbytes = src.numel()
        # the new tensor's memory may still be in use by work queued on the current stream (caching allocator)
        self._stream.wait_stream(torch.cuda.current_stream(core.device))
        with torch.cuda.stream(self._stream):
            for k, pos in enumerate(range(0, nbytes, self.SLAB)):
                n = min(self.SLAB, nbytes - pos)
                slot = k % 2
                if self._staging_free[slot] is not None:
                    self._staging_free[slot].synchronize()  # staging slot still in flight
                self._staging[slot][:n].copy_(src[pos : pos + n])  # host memcpy into pinned memory
                dst[pos : pos + n].copy_(self._staging[slot][:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
                self._staging_free[slot] = ev
            done = torch.cuda.Event()
            done.record(self._stream)
        dev.record_stream(self._stream)
        self.tensors[j] = dev
        self.events[j] = done
        self.host[j] = None

    def ready(self, j):
        """facet ``j`` on the device with the current stream ordered behind its upload"""
        self.prefetch(j)
        ev = self.events[j]
        if ev is not None:
            _torch().cuda.current_stream(self.core.device).wait_event(ev)
            self.events[j] = None
        return self.tensors[j]


class SwiftlyForward:
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
        try:
            self.core.wave_facet_side(bands, [cfg.off0 for cfg in self.facet_configs], self._band, sgs[0].off1, rowmap,
                                      n_rows, Q, compute, [sg.off0 for sg in sgs], flat, g_layout=layout)
        except Exception:
            if compute:  # Q was registered before it was computed: a later request must not find garbage
                self.lru._items.pop(("b", sgs[0].off1), None)  # pylint: disable=protected-access
            raise

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
            bands = torch.empty((F, yB, core.band_columns(self._band)), dtype=self.dtype, device=core.device)
            for j, cfg in enumerate(self.facet_configs):
                data = self._ingest.ready(j)
                t0 = timer.start() if timer is not None else None
                core.prepare_facet_band(data, cfg.off1, self._band, out=bands[j])
                if timer is not None:
                    timer.stop("K1_full_facet_transform", t0)
                self._ingest.prefetch(j + 1)
            self.BF_Fs_persist = bands
            self.__dict__["_k2_chain_forked"] = False  # new band buffers: the next prefetched K2 forks behind K1 again
            if self._plan is not None and _PREFETCH and _PREFETCH_DEPTH >= 2:
                ready = self.__dict__["_bands_ready"] = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(core.device))
        return self.BF_Fs_persist

    def _wave_rows(self, off1):
        """(rowmap, n_rows) of the axis-0 rows the planned subgrids of wave ``off1`` read (None = all rows)."""
        if self._plan is None:
            return None, self.core.yN_size
        key = int(off1)
        if key not in self._wave_rowmaps:
            off0s = [sg.off0 for sg in self._plan if int(sg.off1) == key]
            self._wave_rowmaps[key] = self.core.subgrid_column_rows(off0s)
        return self._wave_rowmaps[key]

    def _get_wave_columns(self, off1):
        """K2: ``Q[F, rows, m]`` for the subgrid wave ``off1`` (LRU cached like the reference's per-off0 columns)."""
        self._take_prefetched(off1)
        hit = self.lru.get(("b", off1))
        if hit is None:
            if self._plan is not None and int(off1) not in self._planned_keys:
                raise ValueError(f"subgrid wave off1={off1} was not in the subgrid_configs plan")
            bands = self.prepare_all_facets()
            rowmap, n_rows = self._wave_rows(off1)
            Q = self.core.prepare_facet_columns(
                bands, [cfg.off0 for cfg in self.facet_configs], self._band, off1, rowmap, n_rows
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
        return _finish_from_columns(self.core, Q, 1, self.facet_configs, sgs, [sg.off0 for sg in sgs], rowmap=rowmap)

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

    # -- planned-wave prefetch (r4): K2 of the NEXT planned wave(s) on the core's side stream ------------------------
    def _predict_next_waves(self, off1, depth):
        """the planned waves a caller that walks the plan asks for after ``off1``, nearest first, at most ``depth`` of
        them ([]: no plan / end / prefetch off).  Positions are those of the PLAN (order of first appearance of the wave
        keys in ``subgrid_configs``): a caller that walks its own plan forwards or backwards is predicted whatever the
        numeric order of the keys; a repeated key keeps the direction of the walk."""
        if self._plan is None or not _PREFETCH or self.__dict__.get("_prefetch_off"):
            return []
        order = self.__dict__.get("_wave_order")
        if order is None:
            order = self.__dict__["_wave_order"] = list(dict.fromkeys(int(sg.off1) for sg in self._plan))
            self.__dict__["_wave_pos"] = {k: i for i, k in enumerate(order)}
        pos = self._wave_pos.get(int(off1))
        if pos is None:
            return []
        last = self.__dict__.get("_last_wave_pos")
        step = self.__dict__.get("_wave_step", 1)
        if last is not None and pos != last:
            step = 1 if pos > last else -1
        self.__dict__["_last_wave_pos"] = pos
        self.__dict__["_wave_step"] = step
        out = []
        for d in range(1, int(depth) + 1):
            nxt = pos + d * step
            if not 0 <= nxt < len(order):
                break
            out.append(order[nxt])
        return out

    def _predict_next_wave(self, off1):
        """the nearest of :py:meth:`_predict_next_waves` (None: nothing to predict)"""
        nxt = self._predict_next_waves(off1, 1)
        return nxt[0] if nxt else None

    def _take_prefetched(self, off1):
        """hand a prefetched ``Q`` of wave ``off1`` over to the LRU cache (the current stream waits for its K2).  When a
        wave that is neither prefetched nor cached has to be computed, the prefetched ones were mispredictions: their
        buffers are dropped, and after two such misses the prefetch is switched off for this object (a wasted K2 per
        wave costs more than the overlap gains)."""
        pending = self.__dict__.get("_prefetched")
        if not pending:
            return
        pf = pending.pop(int(off1), None)
        if pf is None:
            if self.lru.get(("b", off1)) is None:  # a different wave has to be computed: the guess was wrong
                pending.clear()
                missed = self.__dict__["_prefetch_missed"] = self.__dict__.get("_prefetch_missed", 0) + 1
                if missed >= 2:
                    self.__dict__["_prefetch_off"] = True
            return
        if self.lru.get(("b", off1)) is None:
            cur = _torch().cuda.current_stream(self.core.device)
            cur.wait_event(pf[2])
            # Q was allocated under the side stream and is read by kernels of the caller's stream from now on: tell the
            # caching allocator, so that a freed Q is not handed to the next side-stream allocation while `cur` reads it
            pf[0].record_stream(cur)
            self.lru.set(("b", off1), (pf[0], pf[1]))

    def _prefetch_wave(self, off1):
        """Enqueue K2 of planned wave ``off1`` on the side stream: it runs next to the subgrid side (K3-K5) of the wave
        the caller is being served now.  The bandwidth-bound column passes and the issue-bound ``sum_finish`` share
        the chip better than they follow each other (measured r4, 64k workload: 25.5 -> 24.2 ms for the 25 waves).

        Depth 1 (r4): the side stream starts behind everything queued on the caller's stream so far, i.e. K2 of wave
        w + 1 begins when K2 of wave w AND the subgrid side of wave w - 1 have finished -- one cross-stream hand-over
        (a 20-50 us idle gap, tools/trace_timeline.py) per wave.  Depth >= 2 (r5, SWIFTLY_PREFETCH_DEPTH): the side
        stream waits for the band buffers only (an event recorded behind K1), so the K2s of consecutive waves follow
        each other without a hand-over, up to ``depth`` waves ahead of the wave being served; ``Q`` is allocated under
        the side stream and handed over with ``record_stream``, which is what keeps a recycled block from being
        written while the caller's stream still reads it."""
        torch = _torch()
        core = self.core
        pending = self.__dict__.setdefault("_prefetched", {})
        if off1 is None or int(off1) in pending or self.lru.get(("b", off1)) is not None:
            return
        rowmap, n_rows = self._wave_rows(off1)
        main, side = torch.cuda.current_stream(core.device), core.side_stream()
        ready = self.__dict__.get("_bands_ready")
        if _PREFETCH_DEPTH >= 2 and ready is not None:
            side.wait_event(ready)  # K1 of every facet (recorded by _prepare_all_bands)
        else:
            ev = torch.cuda.Event()
            ev.record(main)  # bands ready; every reader of a Q buffer that the allocator may hand out again has been enqueued
            side.wait_event(ev)
        # (r5) second and later K2 of the free-running chain: the chunk streams of the four-step run on from the previous
        # wave's chunks instead of being forked behind its join -- the band buffers were complete before the first (forking)
        # call of this object, Q is a fresh block (swiftly_hip_chain_chunk_streams; 40 us of idle GPU per wave otherwise)
        chain = _PREFETCH_DEPTH >= 2 and ready is not None and _CHAIN_K2 and self.__dict__.get("_k2_chain_forked", False)
        with torch.cuda.stream(side):
            Q = torch.empty((len(self.facet_configs), n_rows, core.xM_yN_size), dtype=self.dtype, device=core.device)
            core.chain_chunk_streams(chain)
            try:
                core.prepare_facet_columns(
                    self.BF_Fs_persist, [cfg.off0 for cfg in self.facet_configs], self._band, off1, rowmap, n_rows, out=Q
                )
            finally:
                core.chain_chunk_streams(False)
            done = torch.cuda.Event()
            done.record(side)
        self.__dict__["_k2_chain_forked"] = True
        pending[int(off1)] = (Q, rowmap, done)

    def _prefetch_waves(self, waves):
        """:py:meth:`_prefetch_wave` for the predicted waves, nearest first, at most SWIFTLY_PREFETCH_DEPTH in flight"""
        for off1 in waves:
            if len(self.__dict__.get("_prefetched") or ()) >= _PREFETCH_DEPTH:
                break
            self._prefetch_wave(off1)

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
        nxt = self._predict_next_waves(sgs[0].off1, _PREFETCH_DEPTH)
        if not compute:
            self._prefetch_waves(nxt)
        m = core.xM_yN_size
        G = torch.empty((len(self.facet_configs), len(sgs), m, m), dtype=self.dtype, device=core.device)
        try:
            core.wave_facet_side(bands, [cfg.off0 for cfg in self.facet_configs], self._band, sgs[0].off1, rowmap,
                                 n_rows, Q, compute, [sg.off0 for sg in sgs], G)
        except Exception:
            if compute:
                self.lru._items.pop(("b", sgs[0].off1), None)  # pylint: disable=protected-access
            raise
        if compute:  # (this wave's own K2 was enqueued on the current stream just now: the next one goes behind it)
            self._prefetch_waves(nxt)
        return _finish_from_G(core, G, self.facet_configs, sgs)


def _finish_from_columns(core, src, layout, facet_configs, sgs, window_offs, rowmap=None, band=None):
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
    return _finish_from_G(core, G, facet_configs, sgs)


def _finish_from_G(core, G, facet_configs, sgs):
    """facet sum + axis-1 finish on chip (sum_finish_facets), then the axis-0 finish, for ``G[F, S, m, m]``."""
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
                                  tmp, res)


def finish_from_blocks(core, blocks, facet_configs, sgs, transformed=True):
    """Finished, masked subgrids ``[S, xA, xA]`` from the per-(facet, subgrid) blocks ``[F, S, m, m]`` of ALL
    facets (``facet_configs`` in the blocks' facet order): the receiving side of the multi-GPU exchange.
    ``transformed`` as in :py:meth:`SwiftlyForward.wave_blocks`."""
    if transformed:
        return _finish_from_G(core, blocks, facet_configs, sgs)
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
