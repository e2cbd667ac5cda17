"""Physics components of the MI355X build - the host-side mirror of the reference's component classes,
for use where the reference checkout is not available (the GPU box) and as the in-repo API:

  EtaLevels        /root/reference/src/cloudsc2_gt4py/physics/common/diagnostics.py:28-45
  Saturation       .../physics/common/saturation.py:33-76
  StateIncrement   .../physics/common/increment.py:32-132
  PerturbedState   .../physics/common/increment.py:135-261
  Cloudsc2NL       .../physics/nonlinear/microphysics.py:43-172
  Cloudsc2TL       .../physics/tangent_linear/microphysics.py:46-242
  Cloudsc2AD       .../physics/adjoint/microphysics.py:46-238

Same class names, constructor arguments, state / tendency / diagnostic field names, units and call
protocol (`component(state, timestep, out_tendencies=..., out_diagnostics=...)`); the bodies are
table-driven: the field lists of include/cloudsc2_hip.h generate the property dicts and the keyword
arguments of the stencil calls.  Parameter groups may be the reference's pydantic models or plain
mappings (anything with `.dict()` or `.items()`).
"""
from __future__ import annotations

from datetime import timedelta
from functools import cached_property
from typing import Any, Dict

from .framework.components import DiagnosticComponent, ImplicitTendencyComponent
from .framework.grid import I, J, K
from .stencils import INC, NL_IN, NL_OUT

_UNITS = {
    "ap": "Pa", "aph": "Pa", "lu": "g g^-1", "lude": "kg m^-3 s^-1", "mfd": "kg m^-2 s^-1", "mfu": "kg m^-2 s^-1",
    "q": "g g^-1", "qi": "g g^-1", "ql": "g g^-1", "qsat": "g g^-1", "supsat": "g g^-1", "t": "K",
    "tnd_cml_q": "g g^-1 s^-1", "tnd_cml_qi": "g g^-1 s^-1", "tnd_cml_ql": "g g^-1 s^-1", "tnd_cml_t": "K s^-1",
    "clc": "", "covptot": "", "fhpsl": "J m^-2 s^-1", "fhpsn": "J m^-2 s^-1", "fplsl": "kg m^-2 s^-1",
    "fplsn": "kg m^-2 s^-1", "tnd_q": "g g^-1 s^-1", "tnd_qi": "g g^-1 s^-1", "tnd_ql": "g g^-1 s^-1",
    "tnd_t": "K s^-1",
}
_HALF = {"aph", "fhpsl", "fhpsn", "fplsl", "fplsn"}
_DIAG_OUT = ("clc", "covptot", "fhpsl", "fhpsn", "fplsl", "fplsn")      # NL outputs kept as diagnostics
_TEND_OUT = ("tnd_q", "tnd_qi", "tnd_ql", "tnd_t")                       # NL outputs kept as tendencies


def _prop(stencil_name: str) -> Dict[str, Any]:
    kdim = K - 1 / 2 if stencil_name in _HALF else K
    return {"grid_dims": (I, J, kdim), "units": _UNITS[stencil_name]}


def _as_dict(group: Any) -> Dict[str, Any]:
    if group is None:
        return {}
    if hasattr(group, "dict"):
        return dict(group.dict())
    return dict(group)


def _externals(*groups: Any, **literals: Any) -> Dict[str, Any]:
    ext: Dict[str, Any] = {}
    for g in groups:
        ext.update(_as_dict(g))
    ext.update(literals)
    return ext


def _stencil_common(component) -> Dict[str, Any]:
    cfg = component.gt4py_config
    return dict(origin=(0, 0, 0), validate_args=cfg.validate_args, exec_info=cfg.exec_info)


# ---------------------------------------------------------------------------------- small components
class EtaLevels(DiagnosticComponent):
    """eta[k] = ap[column 0, k] / aph[column 0, nz] - one device slice operation instead of the
    reference's nz-step Python loop (diagnostics.py:42-45)."""

    @cached_property
    def input_grid_properties(self):
        return {"f_ap": _prop("ap"), "f_aph": _prop("aph")}

    @cached_property
    def diagnostic_grid_properties(self):
        return {"f_eta": {"grid_dims": (K,), "units": ""}}

    def array_call(self, state, out) -> None:
        nz = self.computational_grid.nz
        out["f_eta"][:nz] = state["f_ap"][0, 0, :nz] / state["f_aph"][0, 0, nz]


class Saturation(DiagnosticComponent):
    def __init__(self, computational_grid, kflag: int, lphylin: bool, yoethf_params, yomcst_params, *,
                 enable_checks: bool = True, gt4py_config) -> None:
        super().__init__(computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        ext = _externals(yoethf_params, yomcst_params, KFLAG=kflag, LPHYLIN=lphylin, QMAX=0.5)
        self.saturation = self.compile_stencil("saturation", ext)

    @cached_property
    def input_grid_properties(self):
        return {"f_ap": _prop("ap"), "f_t": _prop("t")}

    @cached_property
    def diagnostic_grid_properties(self):
        return {"f_qsat": _prop("qsat")}

    def array_call(self, state, out) -> None:
        g = self.computational_grid
        self.saturation(in_ap=state["f_ap"], in_t=state["f_t"], out_qsat=out["f_qsat"],
                        domain=(g.nx, 1, g.nz), **_stencil_common(self))


class StateIncrement(DiagnosticComponent):
    def __init__(self, computational_grid, factor: float, ignore_supsat: bool = False, *,
                 enable_checks: bool = True, gt4py_config) -> None:
        super().__init__(computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        self.f = gt4py_config.dtypes.float(factor)
        self.increment = self.compile_stencil("state_increment", {"IGNORE_SUPSAT": ignore_supsat})

    @cached_property
    def input_grid_properties(self):
        return {"f_" + n: _prop(n) for n in INC}

    @cached_property
    def diagnostic_grid_properties(self):
        return {"f_" + n + "_i": _prop(n) for n in INC}

    def array_call(self, state, out) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in INC}
        kw.update({"out_" + n + "_i": out["f_" + n + "_i"] for n in INC})
        self.increment(**kw, f=self.f, domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class PerturbedState(DiagnosticComponent):
    def __init__(self, computational_grid, factor: float, *, enable_checks: bool = True, gt4py_config) -> None:
        super().__init__(computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        self.f = gt4py_config.dtypes.float(factor)
        self.perturbed_state = self.compile_stencil("perturbed_state", {})

    @cached_property
    def input_grid_properties(self):
        props = {"f_" + n: _prop(n) for n in INC}
        props.update({"f_" + n + "_i": _prop(n) for n in INC})
        return props

    @cached_property
    def diagnostic_grid_properties(self):
        return {"f_" + n: _prop(n) for n in INC}

    def array_call(self, state, out) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in INC}
        kw.update({"in_" + n + "_i": state["f_" + n + "_i"] for n in INC})
        kw.update({"out_" + n: out["f_" + n] for n in INC})
        self.perturbed_state(**kw, f=self.f, domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


# ---------------------------------------------------------------------------------- microphysics
def _tend_name(stencil_name: str) -> str:
    """NL tendency outputs are published as f_q / f_qi / f_ql / f_t (nonlinear/microphysics.py:103-108)."""
    return "f_" + stencil_name[len("tnd_"):]


class Cloudsc2NL(ImplicitTendencyComponent):
    def __init__(self, computational_grid, lphylin: bool, ldrain1d: bool, yoethf_params, yomcst_params,
                 yrecldp_params, yrephli_params, yrphnc_params, *, enable_checks: bool = True, gt4py_config) -> None:
        super().__init__(computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        ext = _externals(yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrphnc_params,
                         ICALL=0, LPHYLIN=lphylin, LDRAIN1D=ldrain1d, ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9)
        self.cloudsc2 = self.compile_stencil("cloudsc2_nl", ext)

    @cached_property
    def input_grid_properties(self):
        props = {"f_" + n: _prop(n) for n in NL_IN}
        props["f_eta"] = {"grid_dims": (K,), "units": ""}
        return props

    @cached_property
    def tendency_grid_properties(self):
        return {_tend_name(n): _prop(n) for n in _TEND_OUT}

    @cached_property
    def diagnostic_grid_properties(self):
        return {"f_" + n: _prop(n) for n in _DIAG_OUT}

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in NL_IN}
        kw.update({"out_" + n: out_diagnostics["f_" + n] for n in _DIAG_OUT})
        kw.update({"out_" + n: out_tendencies[_tend_name(n)] for n in _TEND_OUT})
        self.cloudsc2(**kw, in_eta=state["f_eta"], dt=self.gt4py_config.dtypes.float(timestep.total_seconds()),
                      domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class Cloudsc2NLSaturation(Cloudsc2NL):
    """BUILD EXTENSION: `Saturation` + `Cloudsc2NL` as ONE kernel launch (stencil `cloudsc2_nl_saturation`).
    Same inputs as `Cloudsc2NL` minus `f_qsat`, which becomes an additional diagnostic output; the result is
    bit-identical to calling the two components in sequence (tests/test_hip_nl.py)."""

    def __init__(self, computational_grid, lphylin: bool, ldrain1d: bool, yoethf_params, yomcst_params,
                 yrecldp_params, yrephli_params, yrphnc_params, *, enable_checks: bool = True, gt4py_config) -> None:
        ImplicitTendencyComponent.__init__(self, computational_grid, enable_checks=enable_checks,
                                           gt4py_config=gt4py_config)
        ext = _externals(yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrphnc_params,
                         ICALL=0, LPHYLIN=lphylin, LDRAIN1D=ldrain1d, ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9,
                         KFLAG=1, QMAX=0.5)
        self.cloudsc2 = self.compile_stencil("cloudsc2_nl_saturation", ext)

    @cached_property
    def input_grid_properties(self):
        props = {"f_" + n: _prop(n) for n in NL_IN if n != "qsat"}
        props["f_eta"] = {"grid_dims": (K,), "units": ""}
        return props

    @cached_property
    def diagnostic_grid_properties(self):
        props = {"f_" + n: _prop(n) for n in _DIAG_OUT}
        props["f_qsat"] = _prop("qsat")
        return props

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in NL_IN if n != "qsat"}
        kw["out_qsat"] = out_diagnostics["f_qsat"]
        kw.update({"out_" + n: out_diagnostics["f_" + n] for n in _DIAG_OUT})
        kw.update({"out_" + n: out_tendencies[_tend_name(n)] for n in _TEND_OUT})
        self.cloudsc2(**kw, in_eta=state["f_eta"], dt=self.gt4py_config.dtypes.float(timestep.total_seconds()),
                      domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class Cloudsc2NLPerturbed(Cloudsc2NL):
    """BUILD EXTENSION: `PerturbedState(factor)` + `Cloudsc2NL` as ONE kernel launch (stencil
    `cloudsc2_nl_perturbed`): the state fields are read as x + factor * x_i on the fly."""

    def __init__(self, computational_grid, factor: float, lphylin: bool, ldrain1d: bool, yoethf_params, yomcst_params,
                 yrecldp_params, yrephli_params, yrphnc_params, *, enable_checks: bool = True, gt4py_config) -> None:
        ImplicitTendencyComponent.__init__(self, computational_grid, enable_checks=enable_checks,
                                           gt4py_config=gt4py_config)
        self.f = gt4py_config.dtypes.float(factor)
        ext = _externals(yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrphnc_params,
                         ICALL=0, LPHYLIN=lphylin, LDRAIN1D=ldrain1d, ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9)
        self.cloudsc2 = self.compile_stencil("cloudsc2_nl_perturbed", ext)

    @cached_property
    def input_grid_properties(self):
        props = {"f_eta": {"grid_dims": (K,), "units": ""}}
        for n in NL_IN:
            props["f_" + n] = _prop(n)
            props["f_" + n + "_i"] = _prop(n)
        return props

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {}
        for n in NL_IN:
            kw["in_" + n] = state["f_" + n]
            kw["in_" + n + "_i"] = state["f_" + n + "_i"]
        kw.update({"out_" + n: out_diagnostics["f_" + n] for n in _DIAG_OUT})
        kw.update({"out_" + n: out_tendencies[_tend_name(n)] for n in _TEND_OUT})
        self.cloudsc2(**kw, in_eta=state["f_eta"], f=self.f,
                      dt=self.gt4py_config.dtypes.float(timestep.total_seconds()),
                      domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class Cloudsc2TL(ImplicitTendencyComponent):
    def __init__(self, computational_grid, lphylin: bool, ldrain1d: bool, yoethf_params, yomcst_params,
                 yrecldp_params, yrephli_params, yrncl_params, yrphnc_params, *, enable_checks: bool = True,
                 gt4py_config) -> None:
        super().__init__(computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        ext = _externals(yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrncl_params, yrphnc_params,
                         ICALL=0, LPHYLIN=lphylin, LDRAIN1D=ldrain1d, NLEV=computational_grid.nz,
                         ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9)
        self.cloudsc2 = self.compile_stencil("cloudsc2_tl", ext)

    @cached_property
    def input_grid_properties(self):
        props = {"f_eta": {"grid_dims": (K,), "units": ""}}
        for n in NL_IN:
            props["f_" + n] = _prop(n)
            props["f_" + n + "_i"] = _prop(n)
        return props

    @cached_property
    def tendency_grid_properties(self):
        props = {}
        for n in _TEND_OUT:
            props[_tend_name(n)] = _prop(n)
            props[_tend_name(n) + "_i"] = _prop(n)
        return props

    @cached_property
    def diagnostic_grid_properties(self):
        props = {}
        for n in _DIAG_OUT:
            props["f_" + n] = _prop(n)
            props["f_" + n + "_i"] = _prop(n)
        return props

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {}
        for n in NL_IN:
            kw["in_" + n] = state["f_" + n]
            kw["in_" + n + "_i"] = state["f_" + n + "_i"]
        for n in _DIAG_OUT:
            kw["out_" + n] = out_diagnostics["f_" + n]
            kw["out_" + n + "_i"] = out_diagnostics["f_" + n + "_i"]
        for n in _TEND_OUT:
            kw["out_" + n] = out_tendencies[_tend_name(n)]
            kw["out_" + n + "_i"] = out_tendencies[_tend_name(n) + "_i"]
        self.cloudsc2(**kw, in_eta=state["f_eta"], dt=self.gt4py_config.dtypes.float(timestep.total_seconds()),
                      domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class Cloudsc2TLIncremented(Cloudsc2TL):
    """BUILD EXTENSION: `StateIncrement(factor, ignore_supsat)` + `Cloudsc2TL` as ONE launch (stencil
    `cloudsc2_tl_incremented`): the state needs no `f_*_i` fields, the perturbations are factor * state, formed in the
    kernel.  Outputs are those of the two components called one after the other (up to the compiler's fma contraction of
    the shared level function: ulps)."""

    def __init__(self, computational_grid, factor: float, ignore_supsat: bool, lphylin: bool, ldrain1d: bool,
                 yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrncl_params, yrphnc_params, *,
                 enable_checks: bool = True, gt4py_config) -> None:
        ImplicitTendencyComponent.__init__(self, computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        ext = _externals(yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrncl_params, yrphnc_params,
                         ICALL=0, LPHYLIN=lphylin, LDRAIN1D=ldrain1d, NLEV=computational_grid.nz,
                         ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9, IGNORE_SUPSAT=ignore_supsat)
        self.f = gt4py_config.dtypes.float(factor)
        self.cloudsc2 = self.compile_stencil("cloudsc2_tl_incremented", ext)

    @cached_property
    def input_grid_properties(self):
        props = {"f_eta": {"grid_dims": (K,), "units": ""}}
        for n in NL_IN:
            props["f_" + n] = _prop(n)
        return props

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in NL_IN}
        for n in _DIAG_OUT:
            kw["out_" + n] = out_diagnostics["f_" + n]
            kw["out_" + n + "_i"] = out_diagnostics["f_" + n + "_i"]
        for n in _TEND_OUT:
            kw["out_" + n] = out_tendencies[_tend_name(n)]
            kw["out_" + n + "_i"] = out_tendencies[_tend_name(n) + "_i"]
        self.cloudsc2(**kw, in_eta=state["f_eta"], f=self.f, dt=self.gt4py_config.dtypes.float(timestep.total_seconds()),
                      domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class Cloudsc2AD(ImplicitTendencyComponent):
    """State in: the 16 trajectory fields + the adjoint forcings `f_{clc,...}_i`, `f_tnd_{t,q,ql,qi}_i`
    (adjoint/microphysics.py:91-121).  Out: NL tendencies/diagnostics + `f_cml_{t,q,ql,qi}_i` (tendency
    dict) and the 12 adjoint state fields (diagnostic dict), :123-157."""

    def __init__(self, computational_grid, lphylin: bool, ldrain1d: bool, yoethf_params, yomcst_params,
                 yrecldp_params, yrephli_params, yrncl_params, yrphnc_params, *, enable_checks: bool = True,
                 gt4py_config, ad_traj_fix: bool = False) -> None:
        super().__init__(computational_grid, enable_checks=enable_checks, gt4py_config=gt4py_config)
        ext = _externals(yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrncl_params, yrphnc_params,
                         ICALL=0, LPHYLIN=lphylin, LDRAIN1D=ldrain1d, NLEV=computational_grid.nz,
                         ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9, AD_TRAJ_FIX=int(ad_traj_fix))
        self.cloudsc2 = self.compile_stencil(self._stencil_name, ext)

    _stencil_name = "cloudsc2_ad"
    _ADJ_STATE = ("ap", "aph", "lu", "lude", "mfd", "mfu", "q", "qi", "ql", "qsat", "supsat", "t")

    @cached_property
    def input_grid_properties(self):
        props = {"f_eta": {"grid_dims": (K,), "units": ""}}
        props.update({"f_" + n: _prop(n) for n in NL_IN})
        props.update({"f_" + n + "_i": _prop(n) for n in _DIAG_OUT})
        props.update({"f_" + n + "_i": _prop(n) for n in _TEND_OUT})   # f_tnd_t_i, ...
        return props

    @cached_property
    def tendency_grid_properties(self):
        props = {_tend_name(n): _prop(n) for n in _TEND_OUT}
        props.update({"f_cml_" + n[len("tnd_"):] + "_i": _prop(n) for n in _TEND_OUT})
        return props

    @cached_property
    def diagnostic_grid_properties(self):
        props = {"f_" + n: _prop(n) for n in _DIAG_OUT}
        props.update({"f_" + n + "_i": _prop(n) for n in self._ADJ_STATE})
        return props

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in NL_IN}
        kw.update({"in_" + n + "_i": state["f_" + n + "_i"] for n in NL_OUT})
        kw.update({"out_" + n: out_diagnostics["f_" + n] for n in _DIAG_OUT})
        kw.update({"out_" + n: out_tendencies[_tend_name(n)] for n in _TEND_OUT})
        kw.update({"out_" + n + "_i": out_diagnostics["f_" + n + "_i"] for n in self._ADJ_STATE})
        for n in ("q", "qi", "ql", "t"):
            kw["out_tnd_cml_" + n + "_i"] = out_tendencies["f_cml_" + n + "_i"]
        self.cloudsc2(**kw, in_eta=state["f_eta"], dt=self.gt4py_config.dtypes.float(timestep.total_seconds()),
                      domain=(g.nx, 1, g.nz + 1), **_stencil_common(self))


class Cloudsc2ADFromTrajectory(Cloudsc2AD):
    """BUILD EXTENSION: `Cloudsc2AD` called right after `Cloudsc2TL` on the same state - the symmetry test's sequence
    (adjoint/validation.py:135-151) - without the forward sweep that recomputes the NL trajectory (stencil
    `cloudsc2_ad_from_trajectory`).  The state must carry the TL call's NL flux outputs `f_fplsl` / `f_fplsn` (the harness
    puts the TL diagnostics into the state anyway, :149-150).  Only the adjoint fields are written; the NL tendencies /
    diagnostics of the output dicts are NOT (they are the TL call's).  Driver switches only."""

    _stencil_name = "cloudsc2_ad_from_trajectory"

    @cached_property
    def input_grid_properties(self):
        props = dict(super().input_grid_properties)
        props.update({"f_fplsl": _prop("fplsl"), "f_fplsn": _prop("fplsn")})
        return props

    def array_call(self, state, timestep: timedelta, out_tendencies, out_diagnostics, overwrite_tendencies) -> None:
        g = self.computational_grid
        kw = {"in_" + n: state["f_" + n] for n in NL_IN}
        kw.update({"in_" + n + "_i": state["f_" + n + "_i"] for n in NL_OUT})
        kw.update({"out_" + n + "_i": out_diagnostics["f_" + n + "_i"] for n in self._ADJ_STATE})
        for n in ("q", "qi", "ql", "t"):
            kw["out_tnd_cml_" + n + "_i"] = out_tendencies["f_cml_" + n + "_i"]
        self.cloudsc2(**kw, traj_fplsl=state["f_fplsl"], traj_fplsn=state["f_fplsn"], in_eta=state["f_eta"],
                      dt=self.gt4py_config.dtypes.float(timestep.total_seconds()), domain=(g.nx, 1, g.nz + 1),
                      **_stencil_common(self))
