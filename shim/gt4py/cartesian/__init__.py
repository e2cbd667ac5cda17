from . import gtscript  # noqa: F401

StencilObject = object
