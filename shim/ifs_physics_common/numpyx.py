from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.fields import assign, to_numpy  # noqa: F401
