"""Field-by-field comparison (`validate(fields, reference, atol=, rtol=)`, run_nonlinear.py:139-147)."""
from __future__ import annotations

from typing import Any, Mapping, Optional

import numpy as np

from .fields import DataArray, to_numpy


def validate(fields: Mapping[str, Any], reference: Mapping[str, Any], *, atol: Optional[float] = None,
             rtol: Optional[float] = None, report: Optional[dict] = None) -> bool:
    """Prints one line per field present in both dicts (max abs / max rel error, pass/fail) and returns
    True iff all compared fields pass.  A reference name without a counterpart is reported, with the
    `f_qv` -> `f_q` fix-up of SURVEY.md 4.2 (nonlinear/reference.py:32 vs microphysics.py:106).  `report`, when
    given, receives the per-field figures (build extension for the tests; the reference prints only)."""
    atol = 0.0 if atol is None else atol
    rtol = 0.0 if rtol is None else rtol
    ok = True
    for name, ref in reference.items():
        if name == "time":
            continue
        key = name if name in fields else {"f_qv": "f_q"}.get(name)
        if key is None or key not in fields:
            print(f"  {name:12s}: no counterpart among the computed fields - skipped")
            continue
        a = to_numpy(fields[key].data if isinstance(fields[key], DataArray) else fields[key])
        b = to_numpy(ref.data if isinstance(ref, DataArray) else ref)
        n = min(a.shape[-1], b.shape[-1])
        a, b = a[..., :n].astype(np.float64), b[..., :n].astype(np.float64)
        err = np.abs(a - b)
        good = bool(np.all(err <= atol + rtol * np.abs(b)))
        nzb = np.abs(b) > 0
        rel = float(np.max(err[nzb] / np.abs(b[nzb]))) if nzb.any() else 0.0
        print(f"  {name:12s}{'' if key == name else ' (as ' + key + ')'}: max abs err {err.max():.3e}, "
              f"max rel err {rel:.3e} -> {'OK' if good else 'MISMATCH'}")
        ok = ok and good
        if report is not None:
            report[name] = {"as": key, "max_abs_err": float(err.max()), "max_rel_err": rel, "ok": good,
                            "shape": tuple(a.shape), "ref_shape": tuple(b.shape)}
    return ok
