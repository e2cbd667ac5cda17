from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.fields import gt_zeros, managed_temporary_storage  # noqa: F401
