"""Importable alias of the package directory ``conditional-flow-matching_amd/`` (its name is
not a Python identifier): ``import cfm_amd`` loads that directory as the package
``cfm_amd`` so ``cfm_amd.optimal_transport`` etc. resolve inside it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conditional-flow-matching_amd")
_spec = importlib.util.spec_from_file_location(
    "cfm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cfm_amd"] = _mod
_spec.loader.exec_module(_mod)
