"""A minimal executor for the gtscript subset the CLOUDSC2 reference stencils use (TEST TOOLING).

Purpose: pin the oracle and the HIP kernels to the reference's OWN stencil definitions.  GT4Py is not
installed here, so the reference cannot be imported; but its stencils are plain Python source in a
small DSL.  This module parses that source text (`ast`) and executes it with NumPy under the gtscript
semantics summarised in SURVEY.md 8(c):

  * `with computation(FORWARD|BACKWARD|PARALLEL), interval(a, b):` over the (nz+1)-level domain:
    FORWARD visits k ascending, BACKWARD descending; several `interval` blocks inside one
    `computation` are visited in that same k order; `interval(...)` = all levels, negative bounds
    count from nz+1;
  * every statement is evaluated for all columns at once; a field `if` masks the assignments of its
    body; a condition built from externals only is resolved statically (like GT4Py's compile-time ifs);
  * names assigned inside the stencil that are not parameters are 3-D temporaries that keep their
    value at level k between computations; an unassigned temporary reads as 0;
  * `f[0, 0, dk]` / `f[0, 0]` / `f[dk]` address 3-D / IJ / K fields; a bare field name means offset 0;
  * `@gtscript.function`s are inlined (arguments by value, `from __externals__ import ...` resolved
    from the same externals dict), including functions returning tuples;
  * `**` is `np.power`, `exp/tanh/cosh/sqrt/min/max/abs` map to NumPy;
  * precision (`Executor(dtype=...)`): fields, temporaries, function locals and scalar arguments are held in the field
    dtype; externals and literals are Python floats, which NumPy treats as weak scalars (they take the precision of the
    array they meet).  In float64 this is plain double arithmetic; in float32 it is the "every field-valued quantity in
    single precision" reading of GT4Py's numpy backend.

It is deliberately NOT a GT4Py re-implementation: no backends, no code generation, no storage
classes - just enough to run `/root/reference/src/cloudsc2_gt4py/physics/**/_stencils/*.py`
unmodified in the build container (see make_reference_exec.py, which writes the fixtures the tests
use; the reference source itself never enters this repo).
"""
from __future__ import annotations

import ast
import os
from typing import Any, Dict, List, Mapping, Optional, Tuple

import numpy as np

_MATH = {
    "exp": np.exp, "tanh": np.tanh, "cosh": np.cosh, "sqrt": np.sqrt, "abs": np.abs,
    "log": np.log, "sin": np.sin, "cos": np.cos,
    "min": np.minimum, "max": np.maximum,
}


class StencilDef:
    def __init__(self, name: str, node: ast.FunctionDef, is_stencil: bool):
        self.name, self.node, self.is_stencil = name, node, is_stencil


def load_definitions(paths: List[str]) -> Dict[str, StencilDef]:
    """Parse gtscript stencil / function definitions (by Python function name and by registered name)."""
    defs: Dict[str, StencilDef] = {}
    for path in paths:
        tree = ast.parse(open(path).read(), filename=path)
        for node in tree.body:
            if not isinstance(node, ast.FunctionDef):
                continue
            reg_name, is_stencil = node.name, False
            for dec in node.decorator_list:
                if isinstance(dec, ast.Call) and getattr(dec.func, "id", "") in ("stencil_collection", "function_collection"):
                    reg_name = dec.args[0].value if dec.args else dec.keywords[0].value.value
                    is_stencil = dec.func.id == "stencil_collection"
            d = StencilDef(reg_name, node, is_stencil)
            defs[node.name] = d
            defs[reg_name] = d
    return defs


class _Return(Exception):
    def __init__(self, value):
        self.value = value


class Executor:
    """Runs one stencil on fields laid out [level][column]."""

    def __init__(self, defs: Mapping[str, StencilDef], externals: Mapping[str, Any], dtype=np.float64):
        self.defs = defs
        self.ext = dict(externals)
        self.dtype = np.dtype(dtype)

    # ------------------------------------------------------------------ public
    def run(self, name: str, fields: Dict[str, np.ndarray], scalars: Mapping[str, Any], nz: int,
            domain_levels: Optional[int] = None) -> Dict[str, np.ndarray]:
        """Execute stencil `name` in place on `fields` (3-D: (nz+1, nx); IJ: (nx,); K: (nz+1,)).
        `domain_levels` = number of levels of the domain (nz+1 by default, nz for `saturation`).
        Returns the dict of temporaries (for inspection)."""
        d = self.defs[name]
        assert d.is_stencil, name
        self.nlev = domain_levels if domain_levels is not None else nz + 1
        self.fields = fields
        self.kinds = {}
        any3d = None
        for arg in d.node.args.args + d.node.args.kwonlyargs:
            ann = ast.unparse(arg.annotation) if arg.annotation is not None else ""
            if "gtscript.IJ" in ann:
                self.kinds[arg.arg] = "IJ"
            elif "gtscript.K" in ann:
                self.kinds[arg.arg] = "K"
            elif "Field" in ann:
                self.kinds[arg.arg] = "IJK"
                any3d = arg.arg
            else:
                self.kinds[arg.arg] = "scalar"
        self.nx = fields[any3d].shape[1]
        self.scalars = {k: self._scalar(v) for k, v in scalars.items()}
        self.temps: Dict[str, np.ndarray] = {}
        self.scope_ext: Dict[str, Any] = {}
        with np.errstate(all="ignore"):
            self._run_body(d)
        return self.temps

    def _run_body(self, d: StencilDef) -> None:
        for stmt in d.node.body:
            if isinstance(stmt, ast.ImportFrom):
                self._import_externals(stmt, self.scope_ext)
            elif isinstance(stmt, ast.With):
                self._computation(stmt)
            elif isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant):
                continue  # docstring
            else:
                raise NotImplementedError(ast.dump(stmt)[:200])

    # ------------------------------------------------------------------ helpers
    def _scalar(self, v):
        if isinstance(v, (bool, np.bool_)):
            return bool(v)
        if isinstance(v, (int, np.integer)):
            return int(v)
        return self.dtype.type(v)

    def _import_externals(self, stmt: ast.ImportFrom, scope: Dict[str, Any]) -> None:
        assert stmt.module == "__externals__", stmt.module
        for alias in stmt.names:
            if alias.name not in self.ext:
                raise KeyError(f"external {alias.name} not provided")
            scope[alias.name] = self.ext[alias.name]

    def _bounds(self, call: ast.Call) -> Tuple[int, int]:
        args = call.args
        if len(args) == 1 and isinstance(args[0], ast.Constant) and args[0].value is Ellipsis:
            return 0, self.nlev

        def val(a):
            if isinstance(a, ast.Constant):
                return a.value
            if isinstance(a, ast.UnaryOp) and isinstance(a.op, ast.USub):
                return -a.operand.value
            raise NotImplementedError(ast.dump(a))

        lo, hi = val(args[0]), val(args[1])
        lo = self.nlev + lo if lo < 0 else lo
        hi = self.nlev if hi is None else (self.nlev + hi if hi < 0 else hi)
        return lo, hi

    def _computation(self, w: ast.With) -> None:
        order, interval = None, None
        for item in w.items:
            c = item.context_expr
            if c.func.id == "computation":
                order = c.args[0].id
            elif c.func.id == "interval":
                interval = self._bounds(c)
        blocks: List[Tuple[Tuple[int, int], List[ast.stmt]]] = []
        if interval is not None:
            blocks.append((interval, w.body))
        else:
            for inner in w.body:
                assert isinstance(inner, ast.With), "expected `with interval(...)`"
                blocks.append((self._bounds(inner.items[0].context_expr), inner.body))
        ks = range(self.nlev - 1, -1, -1) if order == "BACKWARD" else range(self.nlev)
        for k in ks:
            for (lo, hi), body in blocks:
                if lo <= k < hi:
                    self.k = k
                    self._exec_block(body, {}, None, top=True)

    # ---- statements ----------------------------------------------------------
    def _exec_block(self, body, local: Optional[Dict[str, Any]], mask, top: bool = False) -> None:
        for stmt in body:
            self._exec(stmt, local, mask, top)

    def _exec(self, stmt, local, mask, top):
        if isinstance(stmt, ast.Assign):
            assert len(stmt.targets) == 1
            value = self._eval(stmt.value, local)
            self._assign(stmt.targets[0], value, local, mask, top)
        elif isinstance(stmt, ast.AugAssign):
            cur = self._eval(stmt.target, local)
            value = self._binop(stmt.op, cur, self._eval(stmt.value, local))
            self._assign(stmt.target, value, local, mask, top)
        elif isinstance(stmt, ast.If):
            cond = self._eval(stmt.test, local)
            if isinstance(cond, (bool, np.bool_)):
                self._exec_block(stmt.body if cond else stmt.orelse, local, mask, top)
            else:
                cond = np.broadcast_to(np.asarray(cond, dtype=bool), (self.nx,))
                m_true = cond if mask is None else (mask & cond)
                m_false = ~cond if mask is None else (mask & ~cond)
                self._exec_block(stmt.body, local, m_true, top)
                if stmt.orelse:
                    self._exec_block(stmt.orelse, local, m_false, top)
        elif isinstance(stmt, ast.Return):
            raise _Return(self._eval(stmt.value, local))
        elif isinstance(stmt, ast.ImportFrom):
            self._import_externals(stmt, local if not top else self.scope_ext)
        elif isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant):
            pass
        else:
            raise NotImplementedError(ast.dump(stmt)[:200])

    def _masked(self, old, new, mask):
        if mask is None:
            return new
        return np.where(mask, new, old)

    def _assign(self, target, value, local, mask, top):
        if isinstance(target, ast.Tuple):
            assert isinstance(value, tuple) and len(value) == len(target.elts)
            for t, v in zip(target.elts, value):
                self._assign(t, v, local, mask, top)
            return
        if isinstance(target, ast.Subscript):
            name, off = target.value.id, self._offset(target.slice)
        else:
            name, off = target.id, 0
        assert off == 0, "writes with a vertical offset are not part of the subset"
        if not top:  # inside an inlined function: plain local, held in the field precision like every temporary
            old = local.get(name, 0.0)
            new = self._masked(old, value, mask)
            if isinstance(new, np.ndarray) and new.dtype.kind == "f" and new.dtype != self.dtype:
                new = new.astype(self.dtype)
            local[name] = new
            return
        kind = self.kinds.get(name)
        if kind == "IJK":
            arr = self.fields[name]
            arr[self.k] = self._masked(arr[self.k], value, mask)
        elif kind == "IJ":
            arr = self.fields[name]
            arr[...] = self._masked(arr, value, mask)
        elif kind in ("K", "scalar"):
            raise NotImplementedError(f"assignment to {kind} argument {name}")
        else:
            if name not in self.temps:
                self.temps[name] = np.zeros((self.nlev + 1, self.nx), self.dtype)
            arr = self.temps[name]
            arr[self.k] = self._masked(arr[self.k], value, mask)

    # ---- expressions -----------------------------------------------------------
    def _offset(self, sl) -> int:
        elts = sl.elts if isinstance(sl, ast.Tuple) else [sl]
        vals = []
        for e in elts:
            if isinstance(e, ast.UnaryOp) and isinstance(e.op, ast.USub):
                vals.append(-e.operand.value)
            else:
                vals.append(e.value)
        if len(vals) == 3:
            assert vals[0] == 0 and vals[1] == 0, "horizontal offsets are not part of the subset"
            return vals[2]
        if len(vals) == 2:
            assert vals == [0, 0]
            return 0
        return vals[0]

    def _read(self, name: str, off: int, local):
        if local is not None and name in local:
            assert off == 0
            return local[name]
        kind = self.kinds.get(name)
        if kind == "IJK":
            return self.fields[name][self.k + off]
        if kind == "IJ":
            return self.fields[name]
        if kind == "K":
            return self.fields[name][self.k + off]
        if kind == "scalar":
            return self.scalars[name]
        if name in self.temps:
            return self.temps[name][self.k + off]
        if name in self.scope_ext:
            return self.scope_ext[name]
        # a temporary that was never assigned at this level: reads as 0
        self.temps.setdefault(name, np.zeros((self.nlev + 1, self.nx), self.dtype))
        return self.temps[name][self.k + off]

    def _binop(self, op, a, b):
        if isinstance(op, ast.Add):
            return a + b
        if isinstance(op, ast.Sub):
            return a - b
        if isinstance(op, ast.Mult):
            return a * b
        if isinstance(op, ast.Div):
            return a / b
        if isinstance(op, ast.Pow):
            return a ** b
        raise NotImplementedError(op)

    def _eval(self, node, local):
        if isinstance(node, ast.Constant):
            return node.value
        if isinstance(node, ast.Name):
            if local is not None and node.id in local:
                return local[node.id]
            if local is not None and node.id in self._fn_ext_stack[-1]:
                return self._fn_ext_stack[-1][node.id]
            return self._read(node.id, 0, None if local is None else local)
        if isinstance(node, ast.Subscript):
            return self._read(node.value.id, self._offset(node.slice), local)
        if isinstance(node, ast.BinOp):
            return self._binop(node.op, self._eval(node.left, local), self._eval(node.right, local))
        if isinstance(node, ast.UnaryOp):
            v = self._eval(node.operand, local)
            if isinstance(node.op, ast.USub):
                return -v
            if isinstance(node.op, ast.Not):
                return (not v) if isinstance(v, (bool, np.bool_)) else np.logical_not(v)
            raise NotImplementedError(node.op)
        if isinstance(node, ast.Compare):
            assert len(node.ops) == 1
            a, b = self._eval(node.left, local), self._eval(node.comparators[0], local)
            op = node.ops[0]
            if isinstance(op, ast.Lt):
                return a < b
            if isinstance(op, ast.LtE):
                return a <= b
            if isinstance(op, ast.Gt):
                return a > b
            if isinstance(op, ast.GtE):
                return a >= b
            if isinstance(op, ast.Eq):
                return a == b
            if isinstance(op, ast.NotEq):
                return a != b
            raise NotImplementedError(op)
        if isinstance(node, ast.BoolOp):
            vals = [self._eval(v, local) for v in node.values]
            if all(isinstance(v, (bool, np.bool_)) for v in vals):
                return all(vals) if isinstance(node.op, ast.And) else any(vals)
            out = vals[0]
            for v in vals[1:]:
                out = np.logical_and(out, v) if isinstance(node.op, ast.And) else np.logical_or(out, v)
            return out
        if isinstance(node, ast.Call):
            fname = node.func.id
            args = [self._eval(a, local) for a in node.args]
            if fname in _MATH:
                return _MATH[fname](*args)
            return self._call(self.defs[fname], args)
        if isinstance(node, ast.Tuple):
            return tuple(self._eval(e, local) for e in node.elts)
        raise NotImplementedError(ast.dump(node)[:200])

    _fn_ext_stack: List[Dict[str, Any]] = [{}]

    def _call(self, d: StencilDef, args):
        params = [a.arg for a in d.node.args.args]
        assert len(params) == len(args), d.name
        local: Dict[str, Any] = dict(zip(params, args))
        fext: Dict[str, Any] = {}
        self._fn_ext_stack = self._fn_ext_stack + [fext]
        try:
            for stmt in d.node.body:
                if isinstance(stmt, ast.ImportFrom):
                    self._import_externals(stmt, fext)
                else:
                    self._exec(stmt, local, None, top=False)
        except _Return as r:
            return r.value
        finally:
            self._fn_ext_stack = self._fn_ext_stack[:-1]
        return None


def reference_stencil_files(root: str) -> List[str]:
    base = os.path.join(root, "src", "cloudsc2_gt4py", "physics")
    out = []
    for sub in ("common", "nonlinear", "tangent_linear", "adjoint"):
        d = os.path.join(base, sub, "_stencils")
        out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".py") and f != "__init__.py"]
    return out
