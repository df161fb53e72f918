"""Importable alias of the `human-body-reconstruction_amd/` package directory (a hyphen cannot appear
in a Python module name).  `import hbr_amd` == the package in `human-body-reconstruction_amd/`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "human-body-reconstruction_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
