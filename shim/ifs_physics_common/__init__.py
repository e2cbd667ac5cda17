"""Import-name shim: see shim/README.md."""
