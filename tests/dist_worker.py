"""Worker of tests/test_distributed.py: one rank of a world_size-N `gloo` job on the CPU.  Runs the
build's Taylor and symmetry drivers on this rank's column shard of a GLOBAL synthetic problem (oracle
backend - there is no GPU here) and, on rank 0, prints the all-reduced results as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist

    import oracle_backend

    oracle_backend.register("numpy")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

    cols = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "reader":
        # reader path (`--input auto`): report this rank's first columns and eta, for the shard check of the test
        import argparse

        import numpy as np

        from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
        from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers._common import add_common_options, setup

        ap = argparse.ArgumentParser()
        add_common_options(ap)
        ctx = setup(ap.parse_args(["--backend", "numpy", "--num-cols", cols, "--input", "auto"]))
        st = ctx["state"]
        t = storage.klayout(st["f_t"].data).numpy()
        print("RESULT " + json.dumps({"rank": int(os.environ.get("RANK", "0")), "t_level50": t[50].tolist(),
                                      "eta": np.asarray(st["f_eta"].data).tolist()}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    t = run_taylor_test.main(["--backend", "numpy", "--num-cols", cols, "--input", "synthetic", "--disable-validation"])
    s = run_symmetry_test.main(["--backend", "numpy", "--num-cols", cols, "--input", "synthetic", "--ad-traj-fix"])
    if int(os.environ.get("RANK", "0")) == 0:
        print("RESULT " + json.dumps({"norms": list(map(float, t["norms"])), "symmetry": s["detail"]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
