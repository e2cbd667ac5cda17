from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.config import (  # noqa: F401
    DataTypes, GridConfig, GT4PyConfig, IOConfig, PythonConfig)
