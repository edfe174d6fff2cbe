"""Import shim: `import addk` loads the package that lives in ./auto-dynamic-deeplab_amd/ (a directory name that is
not a valid Python identifier) and registers it as the package `addk`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'auto-dynamic-deeplab_amd')
_spec = importlib.util.spec_from_file_location('addk', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['addk'] = _mod
_spec.loader.exec_module(_mod)
