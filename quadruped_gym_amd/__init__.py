"""Import shim: makes the hyphenated source directory ``quadruped-gym_amd/``
importable as ``quadruped_gym_amd`` (same modules, same files, no copies)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "quadruped-gym_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _fh
