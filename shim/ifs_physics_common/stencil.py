"""`@stencil_collection(name)` / `@function_collection(name)`: the reference registers its gtscript
definitions under these names.  The native build never executes those bodies - it only remembers
which names were registered (the kernels are selected by name in framework.backends)."""
REGISTERED_STENCILS = {}
REGISTERED_FUNCTIONS = {}


def stencil_collection(name):
    def deco(fn):
        REGISTERED_STENCILS[name] = fn
        return fn
    return deco


def function_collection(name):
    def deco(fn):
        REGISTERED_FUNCTIONS[name] = fn
        return fn
    return deco
