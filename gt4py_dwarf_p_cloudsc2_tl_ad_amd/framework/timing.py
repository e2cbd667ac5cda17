"""Wall-clock timing that brackets device work (`with timing("run_0") as timer: ...`,
/root/reference/drivers/run_nonlinear.py:116-119; `Timer.reset()/get_time`, run_taylor_test.py:95-99)."""
from __future__ import annotations

import contextlib
import time
from typing import Dict

import torch


def _sync() -> None:
    # while a HIP graph is being captured (harness / driver `--graph` modes) nothing may synchronise: the timers then
    # measure nothing, and the replay is timed by the caller
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()


class Timer:
    times: Dict[str, float] = {}

    @classmethod
    def reset(cls) -> None:
        cls.times = {}

    @classmethod
    def add(cls, label: str, seconds: float) -> None:
        cls.times[label] = cls.times.get(label, 0.0) + seconds

    @classmethod
    def get_time(cls, label: str, units: str = "ms") -> float:
        scale = {"s": 1.0, "ms": 1e3, "us": 1e6}[units]
        return cls.times.get(label, 0.0) * scale


@contextlib.contextmanager
def timing(label: str):
    _sync()
    t0 = time.perf_counter()
    try:
        yield Timer
    finally:
        _sync()
        Timer.add(label, time.perf_counter() - t0)
