"""Import alias: the package directory is ``sin-inn_amd/`` (hyphen, per the repository layout contract), which
Python cannot import by name.  ``import sin_inn_amd`` loads that directory as the package ``sin_inn_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sin-inn_amd')
_spec = importlib.util.spec_from_file_location('sin_inn_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['sin_inn_amd'] = _mod
_spec.loader.exec_module(_mod)
