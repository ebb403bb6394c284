"""Importable alias of the product package.

The package directory required by the build contract,
`graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/`, is not a valid Python
identifier; this shim makes it importable as `mtmc_mpn` (sub-modules included) without copies
or symlinks: it points `__path__` at that directory and runs its `__init__.py` in this namespace.
"""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "graph-convolutional-network-for-multi-camera-vehicle-tracking_amd")
__path__ = [_impl]
with open(_os.path.join(_impl, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_impl, "__init__.py"), "exec"))
del _f
