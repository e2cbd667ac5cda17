from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.iox import HDF5GridOperator, HDF5Operator  # noqa: F401
