from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.validation import validate  # noqa: F401
