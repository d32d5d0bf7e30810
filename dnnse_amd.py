"""Import shim: the package directory is named `dnn-for-speech-enhancement_amd` (not a valid
Python identifier), so load it by path and expose it as module `dnnse_amd`."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dnn-for-speech-enhancement_amd")
_name = "dnn_for_speech_enhancement_amd"
if _name in sys.modules:
    _pkg = sys.modules[_name]
else:
    _spec = importlib.util.spec_from_file_location(_name, os.path.join(_d, "__init__.py"),
                                                   submodule_search_locations=[_d])
    _pkg = importlib.util.module_from_spec(_spec)
    sys.modules[_name] = _pkg
    _spec.loader.exec_module(_pkg)
globals().update({k: getattr(_pkg, k) for k in dir(_pkg) if not k.startswith("__")})
