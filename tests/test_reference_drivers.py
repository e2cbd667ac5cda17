"""The reference's three drivers run UNMODIFIED against this build's `ifs_physics_common` / `gt4py`
import shim (shim/) - BASELINE configs[0]: plumbing on the host CPU, no GPU.  Only possible where
/root/reference exists (the build container); the stencil backend here is the test-only oracle
backend registered as "numpy" (tests/oracle_backend.py), the inputs are the synthetic 100-column
dataset because data/input.h5 is not shipped with the reference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "drivers")),
                                reason="reference checkout not present (GPU box)")


def _run(driver, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_reference_driver.py"), driver, *args],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))   # nothing is written into /root/reference
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


def test_run_nonlinear_unmodified(tmp_path):
    csv = tmp_path / "perf.csv"
    out = _run("run_nonlinear.py", "--num-cols", "64", "--num-runs", "2", "--output-csv-file", str(csv),
               "--output-csv-file-stencils", str(tmp_path / "stencils.csv"))
    assert "synthetic-parameters" in out            # the input source is stated
    assert "Performance: 64 columns, 2 runs" in out
    assert "== Validation:" in out and "f_qv         (as f_q)" in out   # golden names mapped
    assert "f_covptot   : max abs err 0.000e+00" in out                 # the one field inputs cannot change
    assert csv.exists() and "nl-numpy" in csv.read_text()
    assert "cloudsc2_nl" in (tmp_path / "stencils.csv").read_text()


def test_config1_run_nonlinear_unmodified_at_16384_columns():
    """BASELINE configs[0] at its stated size: the UNMODIFIED reference driver, 16 384 columns x 137 levels fp64 on the host
    CPU (plumbing, no GPU), with its golden comparison step - the stand-in dataset tiled to 16 384 columns."""
    out = _run("run_nonlinear.py", "--num-cols", "16384", "--num-runs", "1")
    assert "Performance: 16384 columns, 1 runs" in out and "== Validation:" in out
    assert "f_covptot   : max abs err 0.000e+00" in out and "f_qv         (as f_q)" in out


def test_run_taylor_test_unmodified():
    out = _run("run_taylor_test.py", "--num-cols", "64", "--num-runs", "1")
    assert ">>> Taylor test: Start" in out and "<<< Taylor test: End" in out
    assert "The test passed with penalty 0. HOORAY!" in out


def test_run_symmetry_test_unmodified():
    out = _run("run_symmetry_test.py", "--num-cols", "64", "--num-runs", "1")
    assert "The symmetry test passed. HOORAY!" in out
