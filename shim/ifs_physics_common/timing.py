from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.timing import Timer, timing  # noqa: F401
