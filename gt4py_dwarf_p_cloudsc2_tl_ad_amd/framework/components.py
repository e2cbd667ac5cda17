"""Component base classes with the call protocol the reference's physics classes rely on
(`DiagnosticComponent.__call__(state, out=...)`, `ImplicitTendencyComponent.__call__(state, timestep,
out_tendencies=..., out_diagnostics=...)`, `array_call(...)`, `compile_stencil(name, externals)`):
/root/reference/src/cloudsc2_gt4py/physics/common/saturation.py:33-76, nonlinear/microphysics.py:43-172."""
from __future__ import annotations

from datetime import timedelta
from typing import Any, Dict, Mapping, Optional, Tuple

from .backends import compile_stencil as _compile
from .fields import DataArray, allocate


class _Component:
    def __init__(self, computational_grid, *, enable_checks: bool = True, gt4py_config) -> None:
        self.computational_grid = computational_grid
        self.enable_checks = enable_checks
        self.gt4py_config = gt4py_config

    def compile_stencil(self, name: str, externals: Optional[Mapping[str, Any]] = None):
        return _compile(name, self.gt4py_config, externals)

    # -- helpers ---------------------------------------------------------------------------------
    def _raw_inputs(self, state: Mapping[str, Any]) -> Dict[str, Any]:
        raw = {}
        for name, props in self.input_grid_properties.items():
            if name not in state:
                raise KeyError(f"{type(self).__name__}: state lacks input field '{name}'")
            f = state[name]
            if self.enable_checks and isinstance(f, DataArray):
                want = tuple(props["grid_dims"])
                if tuple(f.dims) != want:
                    raise ValueError(f"{type(self).__name__}: field '{name}' has dims {f.dims}, expected {want}")
            raw[name] = f.data if isinstance(f, DataArray) else f
        return raw

    def _outputs(self, props: Mapping[str, Mapping[str, Any]], out: Optional[Dict[str, Any]]) -> Dict[str, Any]:
        out = {} if out is None else out
        for name, p in props.items():
            if name not in out:
                data = allocate(self.computational_grid, p["grid_dims"], self.gt4py_config,
                                p.get("dtype_name", "float"))
                out[name] = DataArray(data, p["grid_dims"], p.get("units", ""))
        return out

    @staticmethod
    def _raw(out: Mapping[str, Any], props) -> Dict[str, Any]:
        return {n: (out[n].data if isinstance(out[n], DataArray) else out[n]) for n in props}


class DiagnosticComponent(_Component):
    def __call__(self, state: Dict[str, Any], *, out: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        raw_state = self._raw_inputs(state)
        out = self._outputs(self.diagnostic_grid_properties, out)
        self.array_call(raw_state, self._raw(out, self.diagnostic_grid_properties))
        return out


class ImplicitTendencyComponent(_Component):
    def __call__(self, state: Dict[str, Any], timestep: timedelta, *,
                 out_tendencies: Optional[Dict[str, Any]] = None,
                 out_diagnostics: Optional[Dict[str, Any]] = None,
                 overwrite_tendencies: Optional[Dict[str, bool]] = None) -> Tuple[Dict[str, Any], Dict[str, Any]]:
        raw_state = self._raw_inputs(state)
        tends = self._outputs(self.tendency_grid_properties, out_tendencies)
        diags = self._outputs(self.diagnostic_grid_properties, out_diagnostics)
        overwrite = overwrite_tendencies or {n: True for n in self.tendency_grid_properties}
        self.array_call(raw_state, timestep, self._raw(tends, self.tendency_grid_properties),
                        self._raw(diags, self.diagnostic_grid_properties), overwrite)
        return tends, diags
