"""Import shim: the package directory is ``probing-rag_amd/`` (not a valid
Python identifier), so ``import probing_rag_amd`` resolves here, turns this
module into a package whose ``__path__`` is that directory, and executes its
``__init__.py``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "probing-rag_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f, _os
