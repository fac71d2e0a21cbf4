"""Helpers of the HGEMM bench driver, same names and call signatures as reference kernels/hgemm/tools/utils.py:
    get_device_name() :7-12, get_device_capability() :15-17, pretty_print_line() :96-101,
    build_from_sources(verbose) :104-113, try_load_hgemm_library(force_build, verbose) :116-132, as_col_major(x) :135-140.
The reference imports a pip-installed `toy_hgemm` or JIT-builds its CUDA sources through torch.utils.cpp_extension;
here the library is the C-ABI libcln_amd.so built in-tree by hipcc for gfx950 (cuda-learn-notes_amd/_build.py), and the
object returned exposes the same 38 function names. There is no CPU fallback: without a built library and without
hipcc this raises."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import __graft_entry__ as _entry  # noqa: E402

_pkg = _entry.load_package()
from cuda_learn_notes_amd import bench_utils as _bu  # noqa: E402
from cuda_learn_notes_amd import _loader  # noqa: E402

get_device_name = _bu.get_device_name
pretty_print_line = _bu.pretty_print_line
as_col_major = _bu.as_col_major


def get_device_capability():
    """(major, minor) of the current device; gfx950 reports (9, 5)."""
    return torch.cuda.get_device_capability(torch.cuda.current_device()) if torch.cuda.is_available() else (0, 0)


def build_from_sources(verbose: bool = False):
    """Compile csrc/*.hip for gfx950 into lib/libcln_amd*.so (hipcc, in-tree) and return the loaded hgemm module."""
    pretty_print_line(f"Loading hgemm lib on device: {get_device_name()}, capability: {get_device_capability()}, "
                      f"Arch ENV: {os.environ.get('PYTORCH_ROCM_ARCH', 'gfx950')}")
    _pkg.build(verbose=verbose, force=True)
    _loader._cache.clear()  # dlopen the freshly linked objects, not the ones cached by an earlier load
    return _pkg.hgemm_lib()


def try_load_hgemm_library(force_build: bool = False, verbose: bool = False):
    """Prebuilt library if there is one, else (or with force_build) build from sources -- reference utils.py:116-132."""
    if not force_build:
        try:
            import toy_hgemm as hgemm  # the module name the reference imports (utils.py:120); toy_hgemm.py at the repository root
            hgemm.hgemm_mma_m16n8k16_naive  # noqa: B018 -- first attribute access dlopens libcln_amd.so (LibraryMissing when not built)
            pretty_print_line("Import toy_hgemm (prebuilt libcln_amd.so) done, use it!")
        except _loader.LibraryMissing:
            pretty_print_line("Can't load prebuilt libcln_amd.so, force build from source (hipcc --offload-arch=gfx950)")
            hgemm = build_from_sources(verbose=verbose)
    else:
        pretty_print_line("Force hgemm lib build from sources")
        hgemm = build_from_sources(verbose=verbose)
    return hgemm
