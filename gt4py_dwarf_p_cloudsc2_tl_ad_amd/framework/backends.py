"""Stencil-backend registry.  The product registers exactly one backend: "hip" (the hand-written
gfx950 kernels).  There is deliberately NO CPU backend here; the test-suite registers an
oracle-backed "numpy" backend of its own (tests/oracle_backend.py) to run the reference's drivers in
the GPU-less build container.  Asking for an unregistered backend raises."""
from __future__ import annotations

from typing import Any, Callable, Dict, Mapping, Optional

import torch

_BACKENDS: Dict[str, Dict[str, Any]] = {}


def register_backend(name: str, compile_fn: Callable[[str, Mapping[str, Any]], Callable], device: Any) -> None:
    _BACKENDS[name] = {"compile": compile_fn, "device": device}


def _hip_compile(name: str, externals: Optional[Mapping[str, Any]]):
    from ..stencils import compile_stencil

    return compile_stencil(name, externals or {})


def _hip_device():
    if not torch.cuda.is_available():
        raise RuntimeError("backend 'hip' needs an AMD GPU (torch.cuda.is_available() is False); "
                           "this build has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


register_backend("hip", _hip_compile, _hip_device)


def _get(backend: str) -> Dict[str, Any]:
    try:
        return _BACKENDS[backend]
    except KeyError:
        raise ValueError(
            f"stencil backend {backend!r} is not available: this build provides {sorted(_BACKENDS)} "
            "(GT4Py backends such as 'numpy' / 'gt:gpu' need GT4Py, which this build replaces)") from None


def compile_stencil(name: str, gt4py_config, externals: Optional[Mapping[str, Any]] = None):
    return _get(gt4py_config.backend)["compile"](name, externals or {})


def backend_device(gt4py_config):
    dev = _get(gt4py_config.backend)["device"]
    return dev() if callable(dev) else dev
