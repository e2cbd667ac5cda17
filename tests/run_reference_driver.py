"""Run one UNMODIFIED reference driver (/root/reference/drivers/run_*.py) against this build's shim
with the test-only oracle backend registered as "numpy" (build container only; see oracle_backend.py).

  python tests/run_reference_driver.py run_nonlinear.py --num-cols 64 --num-runs 2
"""
import os
import runpy
import sys

sys.dont_write_bytecode = True   # importing the reference package must not leave __pycache__ in the read-only checkout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("CLOUDSC2_REFERENCE", "/root/reference")


def main() -> None:
    driver = os.path.join(REFERENCE, "drivers", sys.argv[1])
    for p in (os.path.join(REFERENCE, "drivers"), os.path.join(REFERENCE, "src"), os.path.join(ROOT, "shim"), ROOT,
              os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_backend

    oracle_backend.register("numpy")
    sys.argv = [driver] + sys.argv[2:]
    runpy.run_path(driver, run_name="__main__")


if __name__ == "__main__":
    main()
