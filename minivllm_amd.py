"""Import alias: ``import minivllm_amd`` loads the package that lives in the
hyphenated directory ``mini-vllm_amd/`` (a hyphen is not importable by name)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mini-vllm_amd")
_spec = _ilu.spec_from_file_location(
    "minivllm_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["minivllm_amd"] = _mod
_spec.loader.exec_module(_mod)
