"""A slice of the shape / switch fuzz (`profiles/fuzz_shapes.py`: 400- and 800-case runs are kept under profiles/r02/) inside the GPU
suite: random nx (whole waves, ragged, 1-2 columns), very short and mid-size columns, both precisions, random externals switches;
NL always, TL with general increments and AD with general forcings in fp64 - each held to the NumPy oracle."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 11])
def test_fuzz_slice_agrees_with_the_oracle(gpu, seed, monkeypatch, capsys):
    spec = importlib.util.spec_from_file_location("fuzz_shapes", os.path.join(ROOT, "profiles", "fuzz_shapes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fuzz_shapes.py", "40", str(seed)])
    mod.main()
    out = capsys.readouterr().out
    assert "fuzz: 40 cases agree with the oracle" in out
    import re

    assert int(re.search(r"\((\d+) TL-incremented fields held to the oracle directly\)", out).group(1)) >= 50
