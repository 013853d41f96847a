"""Built pybind11 modules land here: pyvector, pymadtree, pymadicp, pypeline (the reference's module names)."""
