"""Host-side counterpart of the `ifs_physics_common` surface the reference's drivers and components
use (last row).  The upstream package is not available in the build container; this
is a from-scratch implementation written from the reference's CALL SITES only.  Behaviours the call
sites do not determine (column tiling, default tolerances, MFLOPS, CSV columns) are this build's
documented choices.  `shim/ifs_physics_common/*` re-exports these modules under the upstream names
so that the unmodified reference drivers import them.
"""
