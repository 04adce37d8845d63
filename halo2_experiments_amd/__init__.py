"""Import alias: the package directory is named ``halo2-experiments_amd`` (not a valid Python
identifier), so this shim makes it importable as ``halo2_experiments_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "halo2-experiments_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
