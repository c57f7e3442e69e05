"""Importable alias of the package directory `neural-sim-nerf_amd/` (a hyphen is not a valid module name).

`import neural_sim_nerf_amd` executes neural-sim-nerf_amd/__init__.py with this package's name, so every
sub-module (`neural_sim_nerf_amd.pack`, `.run_nerf_noscale`, ...) resolves to the files in that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "neural-sim-nerf_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
