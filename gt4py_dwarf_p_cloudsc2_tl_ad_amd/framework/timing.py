"""Timing that brackets device work (`with timing("run_0") as timer: ...`,
/root/reference/drivers/run_nonlinear.py:116-119; `Timer.reset()/get_time`, run_taylor_test.py:95-99).

The reference brackets every group of stencil calls with a timer (TaylorTest.run opens 22 of them per run,
tangent_linear/validation.py:151-176).  Bracketing by `hipDeviceSynchronize` on both sides - the first version of this
module - makes each bracket a pipeline drain: the GPU idles while the host walks the Python + ctypes path of the next
launch.  Here a bracket costs no synchronisation: it records a HIP event pair on the current stream next to the host clock,
and the interval is resolved when somebody ASKS for it (`Timer.get_time`), with ONE synchronisation for everything pending.
A bracket's time is max(host wall time, device time between its events): asynchronous GPU work is measured by its events
(the host returned early), host-side work (the test-only CPU backend, a blocking copy) by the host clock.  Inside a HIP-graph
capture nothing may be recorded or synchronised: brackets are no-ops there and the replay is timed by the caller."""
from __future__ import annotations

import contextlib
import time
from typing import Dict, List, Tuple

import torch


def _gpu() -> bool:
    return torch.cuda.is_available()


_ENABLED = True


def set_enabled(on: bool) -> bool:
    """Switch the brackets off (they become no-ops) / on again; returns the previous setting.  Used by internal objectives
    that call bracketed code hundreds of times without ever asking for the times (storage.tune_placement)."""
    global _ENABLED
    was, _ENABLED = _ENABLED, bool(on)
    return was


class Timer:
    times: Dict[str, float] = {}
    _pending: List[Tuple[str, float, object, object]] = []      # label, host seconds, start event, end event

    @classmethod
    def reset(cls) -> None:
        cls.times = {}
        cls._pending = []

    @classmethod
    def add(cls, label: str, seconds: float) -> None:
        cls.times[label] = cls.times.get(label, 0.0) + seconds

    @classmethod
    def _resolve(cls) -> None:
        if not cls._pending:
            return
        pending, cls._pending = cls._pending, []
        torch.cuda.synchronize()                                 # once, for every bracket recorded since the last query
        for label, host_s, a, b in pending:
            cls.add(label, max(host_s, a.elapsed_time(b) * 1e-3))

    @classmethod
    def get_time(cls, label: str, units: str = "ms") -> float:
        cls._resolve()
        scale = {"s": 1.0, "ms": 1e3, "us": 1e6}[units]
        return cls.times.get(label, 0.0) * scale


@contextlib.contextmanager
def timing(label: str):
    """A bracket's time is max(host wall time, device time between its events): per-label sums can exceed the run's wall time
    when brackets overlap on the device (the reference reports host wall time around synchronous work)."""
    if not _ENABLED:
        yield Timer
        return
    if not _gpu():                                               # CPU-only process: the host clock is the whole story
        t0 = time.perf_counter()
        try:
            yield Timer
        finally:
            Timer.add(label, time.perf_counter() - t0)
        return
    if torch.cuda.is_current_stream_capturing():                 # HIP-graph capture: no events, no synchronisation
        yield Timer
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    t0 = time.perf_counter()
    try:
        yield Timer
    finally:
        host_s = time.perf_counter() - t0
        b.record()
        Timer._pending.append((label, host_s, a, b))
        if len(Timer._pending) >= 4096:                          # nobody is asking: do not hoard events without bound
            Timer._resolve()
