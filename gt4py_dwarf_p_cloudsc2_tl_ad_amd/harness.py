"""Validation harnesses of the TL and AD schemes, on the device.

Counterparts of `TaylorTest` (/root/reference/src/cloudsc2_gt4py/physics/tangent_linear/validation.py:46-261)
and `SymmetryTest` (/root/reference/src/cloudsc2_gt4py/physics/adjoint/validation.py:44-231): same
constructor arguments, same sequence of component calls, same norms, same verdict strings (the
observable contract).  Differences, all on the measurement side:
  * every reduction runs on the GPU (the reference copies ~40 fields to the host per call);
  * with `torch.distributed` initialised the Taylor sums are all-reduced (SUM) and the symmetry
    maximum all-reduced (MAX) over the column shards - the only communication of the whole path;
  * every reduction is ONE kernel launch per group of fields (`reductions.field_sums` / `column_dots`) and the host
    reads the sums back ONCE per run (one all-reduce of (1 + number of step sizes) x 10 doubles across ranks);
  * `TaylorTest(..., fused=True)` applies the perturbation inside the NL kernel's loads AND forms the ten sums of
    NL(x + f x_i) - NL(x) in that kernel's epilogue (stencil `cloudsc2_nl_taylor`: no perturbed outputs are stored, no
    separate difference / sum launches - they were 23 % of the run's kernel time, r03); `fused_norms=True` is the same
    thing under its older name; `fused=True, store_perturbed=True` keeps the r02/r03 behaviour (stencil
    `cloudsc2_nl_perturbed`: the perturbed outputs of the last step size stay available in `tends_nl_p` / `diags_nl_p`, as
    in the reference, and the sums are a separate launch per step size);
    `fused_all=True` evaluates ALL step sizes in ceil(n / 5) launches that share the loads of a level (stencil
    `cloudsc2_nl_taylor_multi`): the ten perturbed runs become bound by arithmetic instead of re-streaming the state -
    and `state_increment` is fused into cloudsc2_tl and into that kernel (stencil `cloudsc2_tl_incremented`, `f_inc=`):
    the increments f1 * state are formed in the kernels and never stored (`state_i` stays empty in this mode);
  * `SymmetryTest(..., fused=True)`: the TIMED call (validation off) runs `cloudsc2_tl_incremented` instead of
    state_increment + cloudsc2_tl, and (r04) `cloudsc2_ad_from_trajectory` instead of cloudsc2_ad - the adjoint sweep alone,
    fed with the TL call's flux outputs instead of recomputing the NL trajectory; the validated call keeps the separate,
    complete launches (its norms need `state_i`, and it comes first, so the AD output storages exist);
  * `graph=True` (both harnesses) captures the run's kernel sequence once into a HIP graph and replays it: one host
    call per run instead of the Python + ctypes path of ~35 launches (the caller must pass the same state storages);
  * `SymmetryTest(..., ad_traj_fix=True)` selects the AD kernel variant whose freezing tests match
    NL/TL (include/cloudsc2_hip.h, `AD_TRAJ_FIX`); default False = the reference's literal behaviour.
"""
from __future__ import annotations

import sys
from datetime import timedelta
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .framework.timing import timing
from .reductions import column_dots, field_sums
from .physics import (Cloudsc2AD, Cloudsc2ADFromTrajectory, Cloudsc2NL, Cloudsc2NLPerturbed, Cloudsc2TL,
                      Cloudsc2TLIncremented, PerturbedState, Saturation, StateIncrement)

_TENDS = ("f_t", "f_q", "f_ql", "f_qi")
_DIAGS = ("f_clc", "f_fhpsl", "f_fhpsn", "f_fplsl", "f_fplsn", "f_covptot")


class _GraphedRun:
    """Capture `enqueue()` (device work only, returns device tensors) once into a HIP graph; `replay()` re-runs it and
    returns the same output tensors.  The stencils launch on torch's current stream, so `torch.cuda.graph` records them."""

    def __init__(self, enqueue, gt4py_config) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("graph=True replays a HIP graph: it needs the hip backend on a GPU")
        saved = gt4py_config.exec_info
        gt4py_config.exec_info = None              # per-stencil HIP events cannot be recorded inside a capture
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                enqueue()                          # warm-up on the capture stream: every output storage exists afterwards
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.out = enqueue()
            torch.cuda.synchronize()
        finally:
            gt4py_config.exec_info = saved

    def replay(self):
        self.graph.replay()
        return self.out


def _allreduce(t: torch.Tensor, op: str) -> torch.Tensor:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t


def taylor_verdict(norms: Sequence[float]) -> Tuple[bool, str]:
    """Scoring rule of TaylorTest.validate (tangent_linear/validation.py:183-217)."""
    e = np.abs(1.0 - np.asarray(norms, dtype=np.float64))
    start = next((i for i in range(e.size) if e[i] < 0.5), -1)
    if start == -1 or start > 3:
        return False, "The test failed with error 13."
    test, negat = -10, 1
    for i in range(start, e.size - 1):
        tmp = int(e[i + 1] < e[i])
        if negat > tmp:
            test += 10
        negat = tmp
    if test == -10:
        test = 11
    if e[start:].min() > 1e-5:
        test += 7
    if e[start:].min() > 1e-6:
        test += 5
    if test > 5:
        return False, f"The test failed with error {test}."
    return True, f"The test passed with penalty {test}. HOORAY!"


class TaylorTest:
    def __init__(self, computational_grid, factor1: float, factor2s: Tuple[float, ...], kflag: int, lphylin: bool,
                 ldrain1d: bool, yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrncl_params,
                 yrphnc_params, *, enable_checks: bool = True, gt4py_config, fused: bool = False,
                 fused_norms: bool = False, fused_all: bool = False, graph: bool = False,
                 store_perturbed: bool = False) -> None:
        self.f1, self.f2s = factor1, tuple(factor2s)
        self.fused_all = fused_all
        self.fused_norms = fused_norms or fused_all or (fused and not store_perturbed)
        self.fused = fused or self.fused_norms
        self.graph = graph
        self._graphed: Optional[_GraphedRun] = None
        self._gt4py_config = gt4py_config
        self._taylor = None
        if self.fused_norms:
            from .physics import _externals
            from .stencils import compile_stencil

            self._taylor = compile_stencil("cloudsc2_nl_taylor_multi" if fused_all else "cloudsc2_nl_taylor", _externals(
                yoethf_params, yomcst_params, yrecldp_params, yrephli_params, yrphnc_params, ICALL=0, LPHYLIN=lphylin,
                LDRAIN1D=ldrain1d, ZEPS1=1e-12, ZEPS2=1e-10, ZQMAX=0.5, ZSCAL=0.9))
        # no regularization in the Taylor test (validation.py:84-85)
        yrncl = dict(yrncl_params.dict() if hasattr(yrncl_params, "dict") else yrncl_params)
        yrncl["LREGCL"] = False
        kw = dict(enable_checks=enable_checks, gt4py_config=gt4py_config)
        self.saturation = Saturation(computational_grid, kflag, lphylin, yoethf_params, yomcst_params, **kw)
        self.cloudsc2_nl = Cloudsc2NL(computational_grid, lphylin, ldrain1d, yoethf_params, yomcst_params,
                                      yrecldp_params, yrephli_params, yrphnc_params, **kw)
        if fused_all:
            self.cloudsc2_tl = Cloudsc2TLIncremented(computational_grid, factor1, False, lphylin, ldrain1d, yoethf_params,
                                                     yomcst_params, yrecldp_params, yrephli_params, yrncl, yrphnc_params, **kw)
        else:
            self.cloudsc2_tl = Cloudsc2TL(computational_grid, lphylin, ldrain1d, yoethf_params, yomcst_params,
                                          yrecldp_params, yrephli_params, yrncl, yrphnc_params, **kw)
        self.state_increment = StateIncrement(computational_grid, factor1, **kw)
        if self.fused_norms:
            self.perturbed_states = [None] * len(self.f2s)
        elif self.fused:
            # build extension: perturbation applied inside the NL kernel's loads (no perturbed copy of the state)
            self.perturbed_nls = [Cloudsc2NLPerturbed(computational_grid, f2, lphylin, ldrain1d, yoethf_params,
                                                      yomcst_params, yrecldp_params, yrephli_params, yrphnc_params,
                                                      **kw) for f2 in self.f2s]
            self.perturbed_states = [None] * len(self.f2s)
        else:
            self.perturbed_states = [PerturbedState(computational_grid, f2, **kw) for f2 in self.f2s]
        self.diags_sat: Dict[str, Any] = {}
        self.state_i: Dict[str, Any] = {}
        self.state_p: Dict[str, Any] = {}
        self.tends_nl: Dict[str, Any] = {}
        self.diags_nl: Dict[str, Any] = {}
        self.tends_tl: Dict[str, Any] = {}
        self.diags_tl: Dict[str, Any] = {}
        self.tends_nl_p: Dict[str, Any] = {}
        self.diags_nl_p: Dict[str, Any] = {}

    def __call__(self, state, timestep: timedelta) -> bool:
        return self.validate(self.run(state, timestep))

    def run(self, state, timestep: timedelta) -> np.ndarray:
        """validation.py:150-181: saturation, NL, increment, TL, then 10 x (perturb, NL, norm)."""
        if self.graph:
            if self._graphed is None:
                self._graphed = _GraphedRun(lambda: self._enqueue(state, timestep), self._gt4py_config)
            with timing("run"):
                sums = self._graphed.replay()
        else:
            sums = self._enqueue(state, timestep)
        return self._finish(sums)

    _NAMES = [("tends", n) for n in _TENDS] + [("diags", n) for n in _DIAGS]

    def _enqueue(self, state, timestep: timedelta) -> torch.Tensor:
        """Everything of a run that happens on the device, enqueued without a host synchronisation (so that it can be
        captured into a HIP graph): returns the (1 + number of step sizes, 10) float64 DEVICE tensor whose row 0 holds the
        sums of the TL perturbation fields and row 1 + i the sums of NL(x + f2s[i] x_i) - NL(x), in `_NAMES` order."""
        names = self._NAMES
        with timing("run"):
            self.diags_sat = self.saturation(state, out=self.diags_sat)
            state.update(self.diags_sat)
            self.tends_nl, self.diags_nl = self.cloudsc2_nl(state, timestep, out_tendencies=self.tends_nl,
                                                            out_diagnostics=self.diags_nl)
            if not self.fused_all:       # fused_all: the increments f1 * state are formed inside the kernels that use them
                self.state_i = self.state_increment(state, out=self.state_i)
                state.update(self.state_i)
            self.tends_tl, self.diags_tl = self.cloudsc2_tl(state, timestep, out_tendencies=self.tends_tl,
                                                            out_diagnostics=self.diags_tl)
        with timing("norms"):
            # denominators: sum of each TL perturbation field (one launch, one small device vector)
            rows = [field_sums([getattr(self, k + "_tl")[n + "_i"].data for k, n in names])]
        if self.fused_all:
            with timing("run"):
                rows.append(self._fused_diffs_all(state, timestep, names))
            return torch.cat([rows[0].reshape(1, -1), rows[1]])
        for i, perturbed_state in enumerate(self.perturbed_states):
            if self.fused_norms:
                with timing("run"):
                    rows.append(self._fused_diffs(state, timestep, self.f2s[i], names))
                continue
            with timing("run"):
                if self.fused:
                    self.tends_nl_p, self.diags_nl_p = self.perturbed_nls[i](
                        state, timestep, out_tendencies=self.tends_nl_p, out_diagnostics=self.diags_nl_p)
                else:
                    self.state_p = perturbed_state(state, out=self.state_p)
                    self.state_p["time"] = state.get("time")
                    self.state_p["f_eta"] = state["f_eta"]
                    self.tends_nl_p, self.diags_nl_p = self.cloudsc2_nl(
                        self.state_p, timestep, out_tendencies=self.tends_nl_p, out_diagnostics=self.diags_nl_p)
            with timing("norms"):
                rows.append(field_sums([getattr(self, k + "_nl_p")[n].data for k, n in names],
                                       [getattr(self, k + "_nl")[n].data for k, n in names]))
        return torch.stack(rows)

    def _finish(self, sums: torch.Tensor) -> np.ndarray:
        """The host side of a run: ONE all-reduce over the column shards, ONE copy to the host, then get_norm per step
        size (validation.py:219-237)."""
        with timing("norms"):
            host = _allreduce(sums.clone() if self.graph else sums, "sum").cpu().numpy()
            return np.array([self._norm(f2, host[1 + i], host[0]) for i, f2 in enumerate(self.f2s)])

    def _nl_fields(self, state, increments: bool = True):
        from .stencils import NL_IN, NL_OUT

        kw = {}
        for n in NL_IN:
            kw["in_" + n] = state["f_" + n].data
            if increments:
                kw["in_" + n + "_i"] = state["f_" + n + "_i"].data
        for n in NL_OUT:   # unperturbed outputs: tendencies are published as f_q / f_qi / f_ql / f_t
            kw["ref_" + n] = (self.tends_nl["f_" + n[len("tnd_"):]] if n.startswith("tnd_") else self.diags_nl["f_" + n]).data
        any_f = state["f_ap"].data
        kw["in_eta"] = state["f_eta"].data if hasattr(state["f_eta"], "data") else state["f_eta"]
        return kw, any_f.shape[0], any_f.shape[2] - 1, any_f.device

    def _out_index(self, names, device) -> torch.Tensor:
        """positions of `names` in NL_OUT order, as a DEVICE index (made once: a host list would be copied to the device
        on every use, which a HIP-graph capture does not allow)"""
        from .stencils import NL_OUT

        idx = self.__dict__.get("_idx")
        if idx is None or idx.device != device:
            idx = torch.tensor([NL_OUT.index(("tnd_" + n[2:]) if k == "tends" else n[2:]) for k, n in names],
                               dtype=torch.long, device=device)
            self._idx = idx
        return idx

    def _fused_diffs(self, state, timestep: timedelta, f2: float, names) -> torch.Tensor:
        """sum(NL(x + f2 x_i) - NL(x)) per output field, formed in the epilogue of ONE kernel launch."""
        from .stencils import NL_OUT, taylor_blocks

        kw, nx, nz, device = self._nl_fields(state)
        part = torch.empty((taylor_blocks(nx), len(NL_OUT)), dtype=torch.float64, device=device)
        cfg = self._gt4py_config
        self._taylor(**kw, out_partials=part, f=f2, dt=float(timestep.total_seconds()), origin=(0, 0, 0),
                     domain=(nx, 1, nz + 1), validate_args=cfg.validate_args, exec_info=cfg.exec_info)
        return part.sum(dim=0).index_select(0, self._out_index(names, device))      # fixed block order: deterministic

    def _fused_diffs_all(self, state, timestep: timedelta, names) -> torch.Tensor:
        """(number of step sizes, 10): the sums of every step size from ceil(n / 5) launches that share their loads."""
        from .stencils import NL_OUT, taylor_blocks

        kw, nx, nz, device = self._nl_fields(state, increments=False)
        part = torch.empty((taylor_blocks(nx), len(self.f2s), len(NL_OUT)), dtype=torch.float64, device=device)
        cfg = self._gt4py_config
        self._taylor(**kw, out_partials=part, fs=self.f2s, f_inc=float(cfg.dtypes.float(self.f1)),
                     dt=float(timestep.total_seconds()), origin=(0, 0, 0),
                     domain=(nx, 1, nz + 1), validate_args=cfg.validate_args, exec_info=cfg.exec_info)
        return part.sum(dim=0).index_select(1, self._out_index(names, device))

    @staticmethod
    def _norm(f2: float, diffs: np.ndarray, sums_tl: np.ndarray) -> float:
        """get_norm / get_field_norm (validation.py:219-261)."""
        total, count = 0.0, 0
        for d, s in zip(diffs, sums_tl):
            den = abs(f2 * s)
            norm = abs(d) / den if den > sys.float_info.epsilon else 0.0
            count += norm > 0
            total += norm
        return total / count if count > 0 else 0.0

    def validate(self, norms: np.ndarray) -> bool:
        print(">>> Taylor test: Start")
        for f2, n in zip(self.f2s, norms):
            print(f"  factor1 = {self.f1:.3e}, factor2 = {f2:.3e}, norm = {n:.10f}")
        ok, log = taylor_verdict(norms)
        print("<<< Taylor test: End")
        print(log)
        return ok


class SymmetryTest:
    def __init__(self, computational_grid, factor: float, kflag: int, lphylin: bool, ldrain1d: bool, yoethf_params,
                 yomcst_params, yrecldp_params, yrephli_params, yrncl_params, yrphnc_params, *,
                 enable_checks: bool = True, gt4py_config, ad_traj_fix: bool = False, graph: bool = False,
                 fused: bool = False) -> None:
        self.f = factor
        self.graph = graph
        self.fused = fused
        self._graphed: Optional[_GraphedRun] = None
        kw = dict(enable_checks=enable_checks, gt4py_config=gt4py_config)
        self.gt4py_config = gt4py_config
        self.saturation = Saturation(computational_grid, kflag, lphylin, yoethf_params, yomcst_params, **kw)
        self.cloudsc2_tl = Cloudsc2TL(computational_grid, lphylin, ldrain1d, yoethf_params, yomcst_params,
                                      yrecldp_params, yrephli_params, yrncl_params, yrphnc_params, **kw)
        self.cloudsc2_ad = Cloudsc2AD(computational_grid, lphylin, ldrain1d, yoethf_params, yomcst_params,
                                      yrecldp_params, yrephli_params, yrncl_params, yrphnc_params,
                                      ad_traj_fix=ad_traj_fix, **kw)
        self.state_increment = StateIncrement(computational_grid, factor, ignore_supsat=True, **kw)
        self.cloudsc2_tl_incremented = None
        self.cloudsc2_ad_from_trajectory = None
        if fused:
            self.cloudsc2_tl_incremented = Cloudsc2TLIncremented(
                computational_grid, factor, True, lphylin, ldrain1d, yoethf_params, yomcst_params, yrecldp_params,
                yrephli_params, yrncl_params, yrphnc_params, **kw)
            if not ldrain1d:        # (the trajectory variant covers the driver switches: no evaporation block)
                self.cloudsc2_ad_from_trajectory = Cloudsc2ADFromTrajectory(
                    computational_grid, lphylin, ldrain1d, yoethf_params, yomcst_params, yrecldp_params, yrephli_params,
                    yrncl_params, yrphnc_params, ad_traj_fix=ad_traj_fix, **kw)
        self.diags_sat: Dict[str, Any] = {}
        self.state_i: Dict[str, Any] = {}
        self.tends_tl: Dict[str, Any] = {}
        self.diags_tl: Dict[str, Any] = {}
        self.tends_ad: Dict[str, Any] = {}
        self.diags_ad: Dict[str, Any] = {}
        self.last: Optional[Dict[str, Any]] = None

    def __call__(self, state, timestep: timedelta, enable_validation: bool = True) -> Optional[bool]:
        """validation.py:132-165."""
        if self.graph and not enable_validation:
            # the timed call of the reference's driver (run_symmetry_test.py:94-98: validation off) = four launches;
            # captured once, replayed per call (the caller must keep passing the same state storages)
            if self._graphed is None:
                self._graphed = _GraphedRun(lambda: self._stencils(state, timestep, False), self.gt4py_config)
            self._graphed.replay()
            return None
        norm1 = self._stencils(state, timestep, enable_validation)
        if not enable_validation:
            return None
        norm2 = self._norm2()
        eps = float(np.finfo(self.gt4py_config.dtypes.float).eps)
        diff = (norm1 - norm2).abs()
        norm3 = torch.where(norm2 == 0, diff / eps, diff / (eps * norm2))
        worst = _allreduce(norm3.max().reshape(1), "max")
        passed_cols = _allreduce((norm3 < 1e4).sum().to(torch.float64).reshape(1), "sum")
        total_cols = _allreduce(torch.tensor([float(norm3.numel())], dtype=torch.float64, device=norm3.device), "sum")
        worst = float(worst.item())
        self.last = {"max_error_eps": worst, "columns_passing": int(passed_cols.item()),
                     "columns": int(total_cols.item())}
        ok = worst < 1e4
        print("The symmetry test passed. HOORAY!" if ok else "The symmetry test failed.")
        print(f"The maximum error is {worst:.10e} times the machine epsilon.")
        if not ok:
            print(f"  ({self.last['columns_passing']} of {self.last['columns']} columns pass; the others are columns "
                  "whose saturation adjustment crosses RTT, where the reference's AD differs from its TL - "
                  "docs/DESIGN_r03_detail.md 3.3, quirks Q4/Q5)")
        return ok

    def _stencils(self, state, timestep: timedelta, enable_validation: bool) -> Optional[torch.Tensor]:
        """saturation, state_increment, cloudsc2_tl, [norm1,] cloudsc2_ad (:135-151); returns norm1 when validating"""
        self.diags_sat = self.saturation(state, out=self.diags_sat)
        state.update(self.diags_sat)
        if self.fused and not enable_validation:
            # timed call: the increments factor * state are formed inside the TL kernel (nothing else reads them here)
            self.tends_tl, self.diags_tl = self.cloudsc2_tl_incremented(state, timestep, out_tendencies=self.tends_tl,
                                                                        out_diagnostics=self.diags_tl)
        else:
            self.state_i = self.state_increment(state, out=self.state_i)
            state.update(self.state_i)
            self.tends_tl, self.diags_tl = self.cloudsc2_tl(state, timestep, out_tendencies=self.tends_tl,
                                                            out_diagnostics=self.diags_tl)
        norm1 = self._norm1() if enable_validation else None
        for n in _TENDS:                                        # add_tendencies_to_state (:222-231)
            state["f_tnd_" + n[2:]] = self.tends_tl[n]
            state["f_tnd_" + n[2:] + "_i"] = self.tends_tl[n + "_i"]
        state.update(self.diags_tl)
        # timed call of the fused mode: cloudsc2_ad without its forward sweep - the two fluxes it would recompute are the
        # TL call's f_fplsl / f_fplsn, in the state since the line above (the NL outputs of tends_ad / diags_ad are then
        # not written: they are the TL call's)
        ad = (self.cloudsc2_ad_from_trajectory if self.fused and not enable_validation and self.tends_ad
              and self.cloudsc2_ad_from_trajectory is not None else self.cloudsc2_ad)
        self.tends_ad, self.diags_ad = ad(state, timestep, out_tendencies=self.tends_ad, out_diagnostics=self.diags_ad)
        return norm1

    def _norm1(self) -> torch.Tensor:
        """get_norm1 (:167-181): per-column sum over levels and fields of (TL output perturbation)^2 - one launch."""
        return column_dots([dct[n + "_i"].data for dct, names in ((self.tends_tl, _TENDS), (self.diags_tl, _DIAGS))
                            for n in names])

    def _norm2(self) -> torch.Tensor:
        """get_norm2 (:183-215): per-column <delta x, AD output> - one launch for the 16 pairs."""
        pairs: List[Tuple[torch.Tensor, torch.Tensor]] = []
        for n in ("t", "q", "ql", "qi"):
            pairs.append((self.state_i["f_tnd_cml_" + n + "_i"].data, self.tends_ad["f_cml_" + n + "_i"].data))
        for n in ("ap", "aph", "t", "q", "qsat", "ql", "qi", "lu", "lude", "mfd", "mfu", "supsat"):
            pairs.append((self.state_i["f_" + n + "_i"].data, self.diags_ad["f_" + n + "_i"].data))
        return column_dots([a for a, _ in pairs], [b for _, b in pairs])
