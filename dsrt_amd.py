"""Import alias: `import dsrt_amd` loads the package in ./deep-space-ray-tracer_amd/ (a directory name with hyphens
cannot appear in an import statement)."""
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_root, "deep-space-ray-tracer_amd")
_spec = importlib.util.spec_from_file_location("dsrt_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dsrt_amd"] = _mod
_spec.loader.exec_module(_mod)
