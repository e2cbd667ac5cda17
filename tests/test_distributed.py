"""Multi-process path on the CPU (`gloo`, world_size 2): columns shard embarrassingly, the ONLY
communication is the all-reduce of the validation scalars (harness._allreduce; RCCL on the GPUs,
gloo here).  Two ranks with 48 columns each of a global 96-column problem must reproduce the
single-process 96-column Taylor norms and symmetry verdict; `eta` comes from GLOBAL column 0 on every
rank (common/diagnostics.py:42-45 reads column 0 of the whole domain)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _result(stdout):
    line = [l for l in stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_two_rank_shards_reproduce_the_single_process_result():
    env = dict(os.environ, OMP_NUM_THREADS="2")
    single = subprocess.run([sys.executable, WORKER, "96"], capture_output=True, text=True, timeout=900,
                            env={**env, "WORLD_SIZE": "1", "RANK": "0"})
    assert single.returncode == 0, single.stderr[-3000:]
    ref = _result(single.stdout)
    port = str(_free_port())
    procs = []
    for rank in range(2):
        e = {**env, "WORLD_SIZE": "2", "RANK": str(rank), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
             "MASTER_PORT": port}
        procs.append(subprocess.Popen([sys.executable, WORKER, "48"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=e))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    got = _result(outs[0][0])
    np.testing.assert_allclose(got["norms"][:8], ref["norms"][:8], rtol=1e-9)
    assert got["symmetry"]["columns"] == ref["symmetry"]["columns"] == 96
    assert got["symmetry"]["columns_passing"] == ref["symmetry"]["columns_passing"] == 96
    assert abs(got["symmetry"]["max_error_eps"] - ref["symmetry"]["max_error_eps"]) < 1e-6


def test_shards_are_slices_of_the_global_problem():
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    whole = make_state(96)
    for rank in range(2):
        part = make_state(96, col0=rank * 48, ncols=48)
        for k in whole:
            assert np.array_equal(part[k], whole[k][:, rank * 48:(rank + 1) * 48]), k
    s0 = make_state(96, col0=0, ncols=1)
    eta = eta_levels()
    assert np.array_equal(eta[:137], s0["f_ap"][:137, 0] / s0["f_aph"][137, 0])   # global column 0, any shard count


def test_reader_path_shards_are_slices_of_the_tiled_global_problem():
    """`--input auto` (the reference's reader path): GLOBAL column j reads file column j mod KLON.  A rank that owns the
    global columns [c0, c0 + nx) must read exactly that slice (every rank used to tile from column 0)."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.config import DataTypes, GT4PyConfig, GridConfig
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.grid import ComputationalGrid, I, IJ, J, K, ExpandedDim
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.iox import SYNTHETIC_KLON, HDF5GridOperator

    import oracle_backend

    oracle_backend.register("numpy")
    cfg = GT4PyConfig(backend="numpy", rebuild=False, validate_args=True, verbose=False,
                      dtypes=DataTypes(bool=bool, float=np.float64, int=np.int64))
    nx_total, nx = 260, 130                        # 2 ranks x 130 columns over a 100-column file: both wrap around
    whole = HDF5GridOperator("/nonexistent/input.h5", ComputationalGrid(GridConfig(nx=nx_total, ny=1, nz=137)),
                             gt4py_config=cfg).get_field((I, J, K), "float", "", "PT", (K, IJ), (IJ, ExpandedDim, K))
    w = whole.data[:, 0, :].numpy()
    assert np.array_equal(w[100:200], w[0:100]) and SYNTHETIC_KLON == 100      # the tiling rule itself
    for rank in range(2):
        part = HDF5GridOperator("/nonexistent/input.h5", ComputationalGrid(GridConfig(nx=nx, ny=1, nz=137)),
                                gt4py_config=cfg, column_offset=rank * nx) \
            .get_field((I, J, K), "float", "", "PT", (K, IJ), (IJ, ExpandedDim, K))
        assert np.array_equal(part.data[:, 0, :].numpy(), w[rank * nx:(rank + 1) * nx]), rank
    assert not np.array_equal(w[:nx], w[nx:2 * nx])                               # the two shards differ


def test_reader_path_under_two_gloo_ranks():
    """The drivers' own set-up (`drivers/_common.setup`, `--input auto`) under torch.distributed with two ranks: rank r holds
    the global columns [r nx, (r+1) nx) of the tiled 100-column dataset, and both ranks hold the SAME eta (global column 0)."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    port = str(_free_port())
    nx = 130
    procs = []
    for rank in range(2):
        e = {**env, "WORLD_SIZE": "2", "RANK": str(rank), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
             "MASTER_PORT": port}
        procs.append(subprocess.Popen([sys.executable, WORKER, str(nx), "reader"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    res = sorted((_result(o) for o, _ in outs), key=lambda r: r["rank"])
    single = subprocess.run([sys.executable, WORKER, str(2 * nx), "reader"], capture_output=True, text=True, timeout=600,
                            env={**env, "WORLD_SIZE": "1", "RANK": "0"})
    assert single.returncode == 0, single.stderr[-3000:]
    whole = _result(single.stdout)
    t = np.array(whole["t_level50"])
    assert np.array_equal(np.array(res[0]["t_level50"]), t[:nx]) and np.array_equal(np.array(res[1]["t_level50"]), t[nx:])
    assert np.array_equal(t[100:200], t[0:100]) and not np.array_equal(t[:nx], t[nx:])
    assert res[0]["eta"] == res[1]["eta"] == whole["eta"]
