"""Loader for the package directory ``decentralopf.jl_amd/`` (a dot is not importable by name)."""
import importlib.util
import os
import sys

_NAME = "decentralopf_jl_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decentralopf.jl_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
