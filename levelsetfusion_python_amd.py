"""Import alias: the package directory is named `levelsetfusion-python_amd` (hyphen, per the build contract),
which is not a Python identifier.  `import levelsetfusion_python_amd` loads that directory as a package under
this name (sub-modules resolve through its __path__)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "levelsetfusion-python_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
