"""Importable alias of the package directory `contracts-marl-aam-corridors_amd/`.

The directory name (fixed by the project layout) contains hyphens and cannot be written in an
`import` statement, so `import gmpe` loads that directory as the package `gmpe`.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "contracts-marl-aam-corridors_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
