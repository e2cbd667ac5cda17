"""Grid symbols and the computational grid (names from /root/reference/src/cloudsc2_gt4py/setup.py:21-43,
physics/*/microphysics.py: `I, J, K`, `K - 1/2`, `IJ`, `D5[i]`, `ExpandedDim`, `ComputationalGrid.grids[dims].shape`)."""
from __future__ import annotations

from typing import Dict, Tuple


class DimSymbol:
    def __init__(self, name: str, offset: float = 0.0, index=None):
        self.name, self.offset, self.index = name, float(offset), index

    def __sub__(self, other: float) -> "DimSymbol":
        return DimSymbol(self.name, self.offset - float(other), self.index)

    def __add__(self, other: float) -> "DimSymbol":
        return DimSymbol(self.name, self.offset + float(other), self.index)

    def __getitem__(self, index: int) -> "DimSymbol":
        return DimSymbol(self.name, self.offset, index)

    def __eq__(self, other) -> bool:
        return isinstance(other, DimSymbol) and (self.name, self.offset) == (other.name, other.offset)

    def __hash__(self) -> int:
        return hash((self.name, self.offset))

    def __repr__(self) -> str:
        s = self.name
        if self.offset:
            s += f"{self.offset:+g}"
        if self.index is not None:
            s += f"[{self.index}]"
        return s


I, J, K = DimSymbol("I"), DimSymbol("J"), DimSymbol("K")
IJ, D5, ExpandedDim = DimSymbol("IJ"), DimSymbol("D5"), DimSymbol("ExpandedDim")


class _SubGrid:
    def __init__(self, shape: Tuple[int, ...]):
        self.shape = shape


class _Grids(dict):
    def __init__(self, nx: int, ny: int, nz: int):
        super().__init__()
        self._n = {"I": nx, "J": ny, "K": nz}

    def __missing__(self, dims):
        if not isinstance(dims, tuple):
            dims = (dims,)
        shape = tuple(self._n[d.name] + (1 if (d.name == "K" and d.offset != 0) else 0) for d in dims)
        self[dims] = _SubGrid(shape)
        return self[dims]


class ComputationalGrid:
    """`grids[(I, J, K)].shape == (nx, ny, nz)`, `grids[(I, J, K - 1/2)].shape == (nx, ny, nz + 1)`."""

    def __init__(self, grid_config):
        self.nx, self.ny, self.nz = int(grid_config.nx), int(grid_config.ny), int(grid_config.nz)
        if self.ny != 1:
            raise ValueError("the column engine works on ny == 1 (the reference drivers use ny = 1)")
        self.grids: Dict = _Grids(self.nx, self.ny, self.nz)
