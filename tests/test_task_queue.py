"""TaskQueue (reference api.py:466-522) semantics with fake events: bounded number of unfinished tasks,
blocking on the oldest, wait_all_done."""
import pytest

from ska_sdp_exec_swiftly_amd.api import LRUCache, TaskQueue


class FakeEvent:
    log = []
    clock = 0

    def __init__(self):
        self.done_at = None
        self.synced = False

    def record(self):
        FakeEvent.clock += 1
        self.done_at = FakeEvent.clock + 2  # finishes two ticks after submission unless waited for

    def query(self):
        return self.synced or FakeEvent.clock >= self.done_at

    def synchronize(self):
        FakeEvent.log.append(self.done_at)
        self.synced = True


def test_queue_bounds_unfinished_tasks():
    FakeEvent.log, FakeEvent.clock = [], 0
    q = TaskQueue(3, event_factory=FakeEvent)
    for i in range(10):
        q.process([f"task{i}"])
        assert len(q.task_queue) <= 3
    q.wait_all_done()
    assert q.task_queue == []
    # had to block at least once, always on the oldest unfinished task (monotone order)
    assert FakeEvent.log and FakeEvent.log == sorted(FakeEvent.log)


def test_queue_of_one_serialises():
    FakeEvent.log, FakeEvent.clock = [], 0
    q = TaskQueue(1, event_factory=FakeEvent)
    q.process(["a", "b", "c"])
    assert len(q.task_queue) == 1 and len(FakeEvent.log) == 2


def test_wait_all_done_raises_for_stuck_task():
    class Stuck(FakeEvent):
        def query(self):
            return False

    q = TaskQueue(4, event_factory=Stuck)
    q.process(["x"])
    with pytest.raises(RuntimeError):
        q.wait_all_done()


def test_lru_cache_contract():
    """reference api.py:525-590"""
    c = LRUCache(2)
    assert c.set(1, "a") == (None, None)
    assert c.set(2, "b") == (None, None)
    assert c.get(1) == "a"  # refresh 1 -> 2 is now the oldest
    assert c.set(3, "c") == (2, "b")
    assert c.get(2) is None
    assert list(c.pop_all()) == [(1, "a"), (3, "c")]
