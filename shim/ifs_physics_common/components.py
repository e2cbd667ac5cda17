from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.components import (  # noqa: F401
    DiagnosticComponent, ImplicitTendencyComponent)
