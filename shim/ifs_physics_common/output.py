from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.output import (  # noqa: F401
    print_performance, write_performance_to_csv, write_stencils_performance_to_csv)
