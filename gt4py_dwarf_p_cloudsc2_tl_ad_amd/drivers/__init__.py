"""Command-line drivers of the MI355X build: counterparts of /root/reference/drivers/run_nonlinear.py,
run_taylor_test.py and run_symmetry_test.py (same options, same printed verdicts), usable where the
reference checkout is absent.  `python -m gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers.run_nonlinear --help`."""
