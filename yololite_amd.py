"""Import shim: the package directory is named `yololite-official-repo_amd` (not a valid Python
identifier), so it is exposed under the importable name `yololite_amd`:

    import yololite_amd
    from yololite_amd.model import load_model_names_imgsize_from_ckpt
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "yololite-official-repo_amd")
_spec = importlib.util.spec_from_file_location("yololite_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["yololite_amd"] = _mod
_spec.loader.exec_module(_mod)
