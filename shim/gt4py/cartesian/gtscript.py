"""Just enough of `gt4py.cartesian.gtscript` for the reference's stencil MODULES to import: the
`Field[...]` type subscripts in the signatures and the `@gtscript.function` decorator.  Nothing here
can compile or run a stencil."""


class _Axis:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


I, J, K = _Axis("I"), _Axis("J"), _Axis("K")
IJ, IK, JK, IJK = (I, J), (I, K), (J, K), (I, J, K)


class _FieldType:
    def __getitem__(self, item):
        return ("Field", item)


Field = _FieldType()


def function(fn):
    return fn


def stencil(*args, **kwargs):
    raise RuntimeError("gt4py is not available in this build: stencils are prebuilt HIP kernels "
                       "(gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils.compile_stencil)")
