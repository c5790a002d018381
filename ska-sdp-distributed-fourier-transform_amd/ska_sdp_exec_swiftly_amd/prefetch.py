"""
Planned-wave prefetch of the contiguous-axis-first forward pipeline (r4 / r5): K2 of the waves a caller that walks its
plan will ask for next, enqueued on the core's side stream while the subgrid side of the wave being served runs.  A
mix-in of :class:`SwiftlyForward`; the multi-GPU classes drive the same object through ``wave_blocks_into``
(distributed.DistributedForward), so they get the same free-running K2 chain.

The knobs live in the ``api`` module namespace (``api._PREFETCH`` ...; the tests switch them at run time).
"""
import logging
import os
import sys

from .tasks import _torch

log = logging.getLogger("fourier-logger")

# tuning knob: SWIFTLY_PREFETCH=0 turns the planned-wave prefetch of SwiftlyForward off (A/B runs)
PREFETCH_DEFAULT = os.environ.get("SWIFTLY_PREFETCH", "1") != "0"


def _env_int(name, default, lowest):
    """integer knob from the environment; anything unparsable falls back to the default (r5 advisor)"""
    try:
        return max(lowest, int(os.environ.get(name, default)))
    except (TypeError, ValueError):
        return default


# how many planned waves K2 may run ahead of the wave being served (_prefetch_wave): 1 = the r4 schedule
PREFETCH_DEPTH_DEFAULT = _env_int("SWIFTLY_PREFETCH_DEPTH", 2, 1)
# SWIFTLY_CHAIN_K2=0: every prefetched K2 forks its chunk streams behind the side stream again (A/B runs)
CHAIN_K2_DEFAULT = os.environ.get("SWIFTLY_CHAIN_K2", "1") != "0"


_REARM_AFTER = 4  # requests in plan order after which a switched-off prefetch is switched on again


def _knobs():
    """the module whose ``_PREFETCH`` / ``_knobs()._PREFETCH_DEPTH`` / ``_knobs()._CHAIN_K2`` are the live values"""
    return sys.modules[__package__ + ".api"]


class WavePrefetch:
    """prediction + side-stream K2 of the next planned waves; state lives in the host object's ``__dict__``"""

    # -- planned-wave prefetch (r4): K2 of the NEXT planned wave(s) on the core's side stream ------------------------
    def _predict_next_waves(self, off1, depth):
        """the planned waves a caller that walks the plan asks for after ``off1``, nearest first, at most ``depth`` of
        them ([]: no plan / end / prefetch off).  Positions are those of the PLAN (order of first appearance of the wave
        keys in ``subgrid_configs``): a caller that walks its own plan forwards or backwards is predicted whatever the
        numeric order of the keys; a repeated key keeps the direction of the walk."""
        if self._plan is None or not _knobs()._PREFETCH:
            return []
        if self.__dict__.get("_prefetch_off"):
            # switched off after two mispredictions (_take_prefetched): a caller that follows the order again for
            # _REARM_AFTER requests gets the prefetch back (r5 advisor: the switch used to be for the life of the object)
            pos, last = self.__dict__.get("_wave_pos", {}).get(int(off1)), self.__dict__.get("_last_wave_pos")
            follows = pos is not None and last is not None and pos - last == self.__dict__.get("_wave_step", 1)
            streak = self.__dict__["_prefetch_streak"] = (self.__dict__.get("_prefetch_streak", 0) + 1) if follows else 0
            if pos is not None:
                self.__dict__["_last_wave_pos"] = pos
            if streak < _REARM_AFTER:
                return []
            self.__dict__["_prefetch_off"] = False
            self.__dict__["_prefetch_missed"] = 0
            self.__dict__["_prefetch_streak"] = 0
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
            # a jump from one end of the order to the other is the next PASS of the same walk (an object that is reused
            # for several passes), not a turn: the direction is kept (r5 advisor)
            wrapped = len(order) > 2 and {pos, last} == {0, len(order) - 1} and (pos == 0) == (step > 0)
            if not wrapped:
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

    def set_wave_order(self, keys):
        """Tell the predictor the order in which the caller will ask for the planned waves (wave keys = ``off1``), when
        it is not the order of first appearance in ``subgrid_configs`` -- e.g. the group order of the multi-GPU pass
        (distributed.DistributedForward).  Keys outside the plan are ignored; the walk restarts."""
        planned = getattr(self, "_planned_keys", None)
        order = [int(k) for k in dict.fromkeys(int(k) for k in keys) if planned is None or int(k) in planned]
        self.__dict__["_wave_order"] = order
        self.__dict__["_wave_pos"] = {k: i for i, k in enumerate(order)}
        self.__dict__.pop("_last_wave_pos", None)
        self.__dict__["_wave_step"] = 1

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
                # the K2 kernels of the dropped waves may still be running (they read the band buffers and write these Q
                # blocks): keep the blocks referenced until their `done` events have fired
                parked = self.__dict__.setdefault("_prefetch_parked", [])
                parked[:] = [p for p in parked if p[2] is not None and not p[2].query()]
                parked.extend(pending.values())
                pending.clear()
                missed = self.__dict__["_prefetch_missed"] = self.__dict__.get("_prefetch_missed", 0) + 1
                if missed >= 2 and not self.__dict__.get("_prefetch_off"):
                    self.__dict__["_prefetch_off"] = True
                    log.info("SwiftlyForward: two mispredicted waves in a row -- planned-wave prefetch switched off "
                             "until %d requests have followed the plan again", _REARM_AFTER)
            return
        self.__dict__["_prefetch_missed"] = 0  # the walk follows the plan
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
        side = core.side_stream()
        ready = self.__dict__.get("_bands_ready")
        if _knobs()._PREFETCH_DEPTH >= 2 and ready is not None:
            if self.__dict__.get("_side_waited") is not ready:  # once per pass: the side stream is in order behind it
                side.wait_event(ready)  # K1 of every facet (recorded by _prepare_all_bands)
                self.__dict__["_side_waited"] = ready
        else:
            ev = torch.cuda.Event()
            # bands ready; every reader of a Q buffer that the allocator may hand out again has been enqueued
            ev.record(torch.cuda.current_stream(core.device))
            side.wait_event(ev)
        # (r5) second and later K2 of the free-running chain: the chunk streams of the four-step run on from the previous
        # wave's chunks instead of being forked behind its join -- the band buffers were complete before the first (forking)
        # call of this object, Q is a fresh block (swiftly_hip_chain_chunk_streams; 40 us of idle GPU per wave otherwise)
        # (axis-1-first pipeline with a row pass per wave: K2 reads rows that finish_axis1_rows has just written on the side
        # stream -- its chunk streams must fork behind them every time)
        chain = (_knobs()._PREFETCH_DEPTH >= 2 and ready is not None and _knobs()._CHAIN_K2
                 and self.__dict__.get("_k2_chain_forked", False) and self._axis1() != 1)
        with torch.cuda.stream(side):
            Q = torch.empty((len(self.facet_configs), n_rows, core.xM_yN_size), dtype=self.dtype, device=core.device)
            src, band = self._k2_source(off1)
            core.chain_chunk_streams(chain)
            try:
                core.prepare_facet_columns(
                    src, [cfg.off0 for cfg in self.facet_configs], band, off1, rowmap, n_rows, out=Q
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
            if len(self.__dict__.get("_prefetched") or ()) >= _knobs()._PREFETCH_DEPTH:
                break
            self._prefetch_wave(off1)
