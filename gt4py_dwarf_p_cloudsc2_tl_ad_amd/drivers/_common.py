"""Shared set-up of the three drivers: options, configuration, grid, state, parameters
(the part of /root/reference/drivers/run_nonlinear.py:51-94 that all drivers repeat)."""
from __future__ import annotations

import argparse
import os
from datetime import datetime, timedelta
from typing import Any, Dict, Tuple

import numpy as np
import torch

from ..framework.config import DataTypes, GridConfig, GT4PyConfig, IOConfig, PythonConfig
from ..framework.fields import DataArray
from ..framework.grid import ComputationalGrid, I, J, K
from ..framework.iox import HDF5GridOperator, HDF5Operator
from ..framework.backends import backend_device
from ..params import default_externals
from ..physics import EtaLevels
from .. import storage, synthetic

DATA_DIR = os.environ.get("CLOUDSC2_DATA_DIR", "/root/reference/data")

_STATE_H5 = {  # state name -> (HDF5 dataset, species index or None, half-level?)   (setup.py:48-65)
    "f_ap": ("PAP", None, False), "f_aph": ("PAPH", None, True), "f_lu": ("PLU", None, False),
    "f_lude": ("PLUDE", None, False), "f_mfd": ("PMFD", None, False), "f_mfu": ("PMFU", None, False),
    "f_qi": ("PCLV", 1, False), "f_ql": ("PCLV", 0, False), "f_q": ("PQ", None, False),
    "f_supsat": ("PSUPSAT", None, False), "f_t": ("PT", None, False),
    "f_tnd_cml_qi": ("TENDENCY_CML_CLD", 1, False), "f_tnd_cml_ql": ("TENDENCY_CML_CLD", 0, False),
    "f_tnd_cml_q": ("TENDENCY_CML_Q", None, False), "f_tnd_cml_t": ("TENDENCY_CML_T", None, False),
}
# parameter groups = the reference's pydantic models (iox.py:25-209), restricted to what the stencils import
_PARAM_GROUPS = {
    "yoethf": ("R2ES", "R3IES", "R3LES", "R4IES", "R4LES", "R5ALSCP", "R5ALVCP", "R5IES", "R5LES", "RALFDCP",
               "RALSDCP", "RALVDCP", "RKOOP1", "RKOOP2", "RTICE", "RTICECU", "RTWAT", "RTWAT_RTICECU_R",
               "RTWAT_RTICE_R", "RVTMP2"),
    "yomcst": ("RCPD", "RD", "RETV", "RG", "RLMLT", "RLSTT", "RLVTT", "RTT", "RV"),
    "yrecldp": ("RLMIN", "RKCONV", "RCLCRIT", "RPECONS"),
    "yrephli": ("RLPTRC", "LPHYLIN"),
    "yrncl": ("LREGCL",),
    "yrphnc": ("LEVAPLS2",),
}


def add_common_options(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--backend", default="hip", help="stencil backend (this build provides: hip)")
    ap.add_argument("--enable-checks", dest="enable_checks", action="store_true", default=False)
    ap.add_argument("--disable-checks", dest="enable_checks", action="store_false")
    ap.add_argument("--enable-validation", dest="enable_validation", action="store_true", default=True)
    ap.add_argument("--disable-validation", dest="enable_validation", action="store_false")
    ap.add_argument("--num-cols", type=int, default=1)
    ap.add_argument("--num-runs", type=int, default=1)
    ap.add_argument("--precision", default="double", choices=["double", "single"])
    ap.add_argument("--host-alias", default=None)
    ap.add_argument("--output-csv-file", default=None)
    ap.add_argument("--output-csv-file-stencils", default=None)
    ap.add_argument("--input", default="auto",
                    help="'auto': <data dir>/input.h5 if readable, else its 100-column synthetic stand-in, tiled "
                         "to --num-cols like the real file; 'synthetic': --num-cols DISTINCT seeded columns "
                         "(mixed regimes) generated on the device; anything else: the path of an HDF5 file with the "
                         "datasets of the reference's data/input.h5 (it must exist), tiled to --num-cols")


def make_config(args) -> Tuple[PythonConfig, IOConfig]:
    cfg = PythonConfig(
        num_cols=args.num_cols, enable_validation=args.enable_validation,
        input_file=os.path.join(DATA_DIR, "input.h5"), num_runs=args.num_runs, precision="double",
        data_types=DataTypes(bool=bool, float=np.float64, int=np.int64),
        gt4py_config=GT4PyConfig(backend=args.backend, rebuild=False, validate_args=True, verbose=True),
        sympl_enable_checks=True,
    ).with_precision(args.precision).with_backend(args.backend).with_checks(args.enable_checks)
    cfg = cfg.with_validation(args.enable_validation, getattr(args, "atol", None), getattr(args, "rtol", None))
    io = IOConfig(output_csv_file=None, host_name="").with_output_csv_file(args.output_csv_file) \
        .with_host_name(args.host_alias)
    return cfg, io


def _groups_from_defaults() -> Dict[str, Dict[str, Any]]:
    ext = default_externals()
    return {g: {k: ext[k] for k in keys} for g, keys in _PARAM_GROUPS.items()}


def setup(args) -> Dict[str, Any]:
    """Grid, state (with f_eta), timestep and parameter groups.  Input source: see `--input`."""
    cfg, io = make_config(args)
    gcfg = cfg.gt4py_config
    device = backend_device(gcfg)
    # "auto": the reader path (real input.h5 if readable, otherwise the 100-column synthetic stand-in
    # dataset of framework.iox, tiled to --num-cols exactly as the real file would be)
    use_file = args.input != "synthetic"
    if use_file and args.input != "auto":
        if not os.path.isfile(args.input):
            raise FileNotFoundError(f"--input {args.input}: no such file ('auto' falls back to the synthetic stand-in, "
                                    "an explicit path does not)")
        cfg = cfg._with(input_file=os.path.abspath(args.input))
    if use_file:
        op = HDF5Operator(cfg.input_file, gt4py_config=gcfg)
        nz = int(np.asarray(op.f["KLEV"]).reshape(-1)[0])
        nx = cfg.num_cols or int(np.asarray(op.f["KLON"]).reshape(-1)[0])
        grid = ComputationalGrid(GridConfig(nx=nx, ny=1, nz=nz))
        from ..framework.grid import D5, IJ, ExpandedDim

        def read_state(g, names, col0):
            gop = HDF5GridOperator(cfg.input_file, g, gt4py_config=gcfg, column_offset=col0)
            out: Dict[str, Any] = {}
            for name in names:
                h5, idx, half = _STATE_H5[name]
                kdim = K - 1 / 2 if half else K
                dims_map = (IJ, ExpandedDim, kdim) if idx is None else (IJ, ExpandedDim, K, D5[idx])
                h5_dims = (kdim, IJ) if idx is None else (D5, K, IJ)
                out[name] = gop.get_field((I, J, kdim), "float", "", h5, h5_dims, dims_map)
            return out

        # this rank's slice [rank * nx, (rank + 1) * nx) of the tiled global problem
        rank, world = _rank_world()
        state = read_state(grid, _STATE_H5, rank * nx)
        dt = timedelta(seconds=float(np.asarray(op.f["PTSPHY"]).reshape(-1)[0]))
        names = {k: np.asarray(op.f[k]).reshape(-1)[0] for k in op.f.keys() if np.asarray(op.f[k]).size == 1}
        base = default_externals()
        groups = _groups_from_defaults()
        for k, v in names.items():
            key = k.split("_", 1)[1] if k.startswith(("YRECLDP_", "YREPHLI_")) else k
            for g, keys in _PARAM_GROUPS.items():
                if key in keys:
                    groups[g][key] = type(base[key])(v)
        real = os.path.exists(cfg.input_file) and not type(op.f).__name__ == "_Defaulting"
        source = (f"{cfg.input_file}" if real else
                  "synthetic 100-column stand-in for data/input.h5 (cold regime), tiled + synthetic-parameters")
    else:
        nz = 137
        nx = cfg.num_cols or 100
        grid = ComputationalGrid(GridConfig(nx=nx, ny=1, nz=nz))
        rank, world = _rank_world()
        s = synthetic.make_state(nx * world, nz, col0=rank * nx, ncols=nx, dtype=gcfg.dtypes.float, device=device)
        state = {}
        for name, t in s.items():
            kdim = K - 1 / 2 if name == "f_aph" else K
            # (copied into a storage of this build: every field of a call must share one level pitch, storage.level_pitch)
            state[name] = DataArray(storage.from_klayout(t, gcfg.dtypes.float, device), (I, J, kdim), "")
        dt = timedelta(seconds=3600.0)
        groups = _groups_from_defaults()
        source = "synthetic columns (seed 20240807) + synthetic-parameters"
    state["time"] = datetime(1970, 1, 1)
    # eta from GLOBAL column 0 (common/diagnostics.py:42-45 reads column 0 of the whole domain)
    if _rank_world()[1] == 1 or (use_file and _rank_world()[0] == 0):
        eta_levels = EtaLevels(grid, enable_checks=cfg.sympl_enable_checks, gt4py_config=gcfg)
        state.update(eta_levels(state))
    elif use_file:
        # a rank whose first column is not global column 0 reads that one column of the file for it
        g1 = ComputationalGrid(GridConfig(nx=1, ny=1, nz=nz))
        eta_levels = EtaLevels(g1, enable_checks=cfg.sympl_enable_checks, gt4py_config=gcfg)
        state.update(eta_levels(read_state(g1, ("f_ap", "f_aph"), 0)))
    else:
        eta = torch.as_tensor(synthetic.eta_levels(nz, dtype=gcfg.dtypes.float), device=device)
        state["f_eta"] = DataArray(eta, (K,), "")
    print(f"[cloudsc2-hip] input: {source}; {nx} columns x {nz} levels, {cfg.precision}, backend {gcfg.backend}")
    return dict(config=cfg, io_config=io, grid=grid, state=state, dt=dt, params=groups, nx=nx, nz=nz, source=source)


def _rank_world() -> Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed_from_env() -> None:
    """One process per GPU under torch.distributed.run (RCCL); no-op for a single process."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))


def tune_field_placement(field_dicts, objective, *, _any_device: bool = False, **tuner_kw) -> Dict[str, Any]:
    """Re-place every 3-D GPU field found in `field_dicts` (dicts name -> DataArray; a field shared by several dicts or
    DataArrays is one field) at the placement `storage.tune_placement` measures to be fastest for `objective()` - the
    caller's own timed region, run on the candidate placement.  Contents are preserved; the DataArrays are re-pointed at
    the new storages in place, so every dict keeps working.  Returns the tuner's report (docs/TUNING_LOG.md 3.7).
    Exception-safe: if the tuner or the objective raises (out of memory, no room for the arena, a failing stencil), every
    DataArray is pointed back at its original storage with its original contents and the report carries `error` - the
    caller goes on untuned, as bench.py does.  (`_any_device`: lets the CPU unit test drive this with host tensors.)"""
    from ..framework.fields import FieldTensor

    by_ptr: Dict[int, list] = {}
    for d in field_dicts:
        for v in d.values():
            t = getattr(v, "data", None)
            if isinstance(v, DataArray) and isinstance(t, torch.Tensor) and t.dim() == 3 and (t.is_cuda or _any_device) and t.shape[1] == 1:
                by_ptr.setdefault(t.data_ptr(), [])
                if all(v is not w for w in by_ptr[t.data_ptr()]):
                    by_ptr[t.data_ptr()].append(v)
    if not by_ptr:
        return {"fields": 0}
    groups = list(by_ptr.values())
    first = groups[0][0].data
    nx, nz = first.shape[0], first.shape[2] - 1
    if any(g[0].data.shape != first.shape or g[0].data.dtype != first.dtype for g in groups):
        raise ValueError("tune_field_placement: the fields must share one shape and dtype")
    order = [f"f{i}" for i in range(len(groups))]
    sources = {n: storage.klayout(g[0].data.as_subclass(torch.Tensor)).clone() for n, g in zip(order, groups)}

    originals = [[da.data for da in g] for g in groups]     # kept until the tuner has returned: a failure restores them

    def point_at(fields):
        for n, g in zip(order, groups):
            for da in g:
                da.data = fields[n].as_subclass(FieldTensor)

    def launch(fields):
        point_at(fields)
        objective()

    np_dtype = {torch.float64: np.float64, torch.float32: np.float32}[first.dtype]
    import time

    t_tune = time.perf_counter()
    try:
        fields, report = storage.tune_placement(nx, nz, np_dtype, first.device, order, sources, launch, **tuner_kw)
    except Exception as exc:  # noqa: BLE001 - e.g. out of memory, "do not fit the arena cap", a failing objective
        # the state must not be left pointing into a half-tuned arena with its inout fields modified by candidate runs:
        # every DataArray goes back to its original storage with its original contents, and the caller runs untuned
        for g, olds, n in zip(groups, originals, order):
            storage.klayout(olds[0].as_subclass(torch.Tensor)).copy_(sources[n])
            for da, old in zip(g, olds):
                da.data = old
        if first.is_cuda:
            torch.cuda.empty_cache()
        return {"fields": len(groups), "error": f"{type(exc).__name__}: {exc}"[:300]}
    point_at(fields)
    report["fields"] = len(groups)
    report["tuning_s"] = time.perf_counter() - t_tune
    return report


def report_placement(rep: Dict[str, Any], unit: str = "run") -> None:
    """the line the drivers print after `--tune-placement`"""
    if rep.get("error"):
        print(f"[cloudsc2-hip] field placement NOT tuned ({rep['error']}): the {rep.get('fields')} fields stay where they "
              "were, the run continues on the untuned placement")
        return
    # BOTH figures, always: what the same timed region takes on the untuned placement and on the tuned one, and what the
    # opt-in cost (the arena stays allocated as long as the fields live; the tuning time is outside the timed runs)
    st2 = rep.get("second_stage")
    stage = ("" if st2 is None else
             f"; second stage {'chosen' if st2.get('chosen') else 'searched, not chosen'}"
             f" ({st2.get('candidates')} candidates at shifts up to {st2.get('shift_MB')} MB)" if "error" not in st2 else
             f"; second stage skipped ({st2['error']})")
    print(f"[cloudsc2-hip] field placement: UNTUNED {rep.get('default_ms', 0):.4f} ms per {unit} -> TUNED "
          f"{rep.get('tuned_ms', 0):.4f} ms per {unit} ({100.0 * (1.0 - rep.get('tuned_ms', 0) / max(rep.get('default_ms', 0), 1e-30)):+.1f} % "
          f"faster; {rep.get('candidates')} candidates, +{rep.get('extra_spacing_x2MB')} x 2 MB slab spacing, stagger "
          f"{rep.get('stagger_bytes')} B, shift {rep.get('shift_MB')} MB, {rep.get('fields')} fields; cost: a "
          f"{rep.get('arena_bytes', 0) / 1e9:.1f} GB arena and {rep.get('tuning_s', 0):.1f} s of tuning{stage})")
