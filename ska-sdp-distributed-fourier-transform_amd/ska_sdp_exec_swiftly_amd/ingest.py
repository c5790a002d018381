"""
Host -> device edges of the streaming classes: cached device tables of the config masks and the pinned staging ring that
uploads host facets one ahead of the kernels that use them.
"""
import numpy

from .tasks import _torch

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
