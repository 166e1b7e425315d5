"""Import shim: the package directory is named `yolo-v3-tf2_amd/` (not a valid Python
identifier), so `import yolo_v3_tf2_amd` lands here and is redirected to that directory."""
import importlib.util as _u
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "yolo-v3-tf2_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_pkg_dir, "__init__.py"),
                                   submodule_search_locations=[_pkg_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
