"""Import alias: the product package lives in the directory
``learning-continuous-implicit-representation-for-near-periodic-patterns_amd`` (not a valid
Python identifier); ``import npp_amd`` loads it under this name."""
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "learning-continuous-implicit-representation-for-near-periodic-patterns_amd")
__path__ = [_pkg]
with open(_os.path.join(_pkg, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg, "__init__.py"), "exec"))
del _f
