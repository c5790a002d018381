"""
Bounded caches and task handles of the streaming classes: ``LRUCache`` / ``TaskQueue`` with the reference's contracts
(src/ska_sdp_exec_swiftly/api.py:466-590), HIP events instead of Dask futures, and ``DeviceTask`` -- the counterpart of
a ``dask.delayed`` result.
"""
import logging

log = logging.getLogger("fourier-logger")


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
