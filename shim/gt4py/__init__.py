"""Import-name stub (see shim/README.md): NOT GT4Py."""
