"""Import alias: the package directory is ``f5-tts_amd/`` (not a valid identifier), so this
one-file loader registers it under the importable name ``f5_tts_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "f5-tts_amd")
_spec = importlib.util.spec_from_file_location(
    "f5_tts_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["f5_tts_amd"] = _mod
_spec.loader.exec_module(_mod)
