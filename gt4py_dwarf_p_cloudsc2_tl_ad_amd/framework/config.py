"""Configuration objects: names and methods as used by /root/reference/drivers/config.py:22-48 and
/root/reference/drivers/run_nonlinear.py:197-232 (`with_*` builders returning new objects)."""
from __future__ import annotations

import socket
from typing import Any, Dict, Optional

import numpy as np
from pydantic import BaseModel, ConfigDict


class _Model(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True, extra="allow")

    def dict(self, **kw) -> Dict[str, Any]:  # pydantic-v1 spelling used by the reference
        return {k: getattr(self, k) for k in type(self).model_fields}

    def _with(self, **changes):
        args = self.dict()
        args.update(changes)
        return type(self)(**args)


class DataTypes(_Model):
    bool: Any
    float: Any
    int: Any

    def with_precision(self, precision: str) -> "DataTypes":
        if precision == "double":
            return DataTypes(bool=bool, float=np.float64, int=np.int64)
        if precision == "single":
            return DataTypes(bool=bool, float=np.float32, int=np.int32)
        raise ValueError(f"precision must be 'double' or 'single', got {precision!r}")


class GT4PyConfig(_Model):
    backend: str
    backend_opts: Dict[str, Any] = {}
    build_info: Optional[Dict[str, Any]] = None
    device_sync: bool = True
    dtypes: DataTypes = DataTypes(bool=bool, float=np.float64, int=np.int64)
    exec_info: Optional[Dict[str, Any]] = None
    managed: Any = "gt4py"
    rebuild: bool = False
    validate_args: bool = False
    verbose: bool = True

    def with_backend(self, backend: Optional[str]) -> "GT4PyConfig":
        return self._with(backend=backend or self.backend)

    def with_dtypes(self, dtypes: DataTypes) -> "GT4PyConfig":
        return self._with(dtypes=dtypes)

    def with_validate_args(self, flag: bool) -> "GT4PyConfig":
        return self._with(validate_args=flag)

    def reset_exec_info(self) -> None:
        self.exec_info = {}


class PythonConfig(_Model):
    num_cols: Optional[int] = None
    enable_validation: bool = True
    input_file: str = ""
    num_runs: int = 1
    precision: str = "double"
    data_types: DataTypes = DataTypes(bool=bool, float=np.float64, int=np.int64)
    gt4py_config: GT4PyConfig
    sympl_enable_checks: bool = True
    num_threads: int = 1
    atol: Optional[float] = None
    rtol: Optional[float] = None

    def with_precision(self, precision: str) -> "PythonConfig":
        dtypes = self.data_types.with_precision(precision)
        return self._with(precision=precision, data_types=dtypes,
                          gt4py_config=self.gt4py_config.with_dtypes(dtypes))

    def with_backend(self, backend: Optional[str]) -> "PythonConfig":
        return self._with(gt4py_config=self.gt4py_config.with_backend(backend))

    def with_checks(self, enabled: bool) -> "PythonConfig":
        return self._with(sympl_enable_checks=enabled,
                          gt4py_config=self.gt4py_config.with_validate_args(enabled))

    def with_validation(self, enabled: bool, atol: Optional[float] = None,
                        rtol: Optional[float] = None) -> "PythonConfig":
        # defaults (build's choice; the upstream ones are not visible): fp64 1e-12 / 1e-18,
        # fp32 1e-5 / 1e-10, as proposed in SURVEY.md Appendix E
        single = self.precision == "single"
        return self._with(enable_validation=enabled,
                          atol=atol if atol is not None else (1e-10 if single else 1e-18),
                          rtol=rtol if rtol is not None else (1e-5 if single else 1e-12))

    def with_num_cols(self, n: Optional[int]) -> "PythonConfig":
        return self._with(num_cols=n)

    def with_num_runs(self, n: Optional[int]) -> "PythonConfig":
        return self._with(num_runs=n or self.num_runs)

    def with_num_threads(self, n: Optional[int]) -> "PythonConfig":
        return self._with(num_threads=n or self.num_threads)


class IOConfig(_Model):
    output_csv_file: Optional[str] = None
    host_name: Optional[str] = ""

    def with_output_csv_file(self, path: Optional[str]) -> "IOConfig":
        return self._with(output_csv_file=path)

    def with_host_name(self, name: Optional[str]) -> "IOConfig":
        return self._with(host_name=name or socket.gethostname())


class GridConfig(_Model):
    nx: int
    ny: int = 1
    nz: int
