"""Import shim: `import cuda_learn_notes_amd` from the repository root.

The package directory is named `cuda-learn-notes_amd` (the name of the project it mirrors; a hyphen is not importable), so this
one-file module loads that directory AS the package `cuda_learn_notes_amd` -- submodules included (`from cuda_learn_notes_amd
import bench_utils`). An installed copy (pyproject.toml maps the directory to the same package name) does not need it.
Reference boundary: the scripts' `import toy_hgemm` / `load(name=..., sources=...)` (kernels/hgemm/tools/utils.py:116-132)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cuda-learn-notes_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
