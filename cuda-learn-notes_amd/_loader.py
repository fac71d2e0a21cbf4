"""ctypes loader for the C-ABI libraries. There is NO fallback: if the HIP library is missing the
product path raises (the oracle / torch are never substituted for it).

Reference boundary being replaced: `lib = load(name=..., sources=[...])` returning a pybind module
(kernels/elementwise/elementwise.py:10-22) and `try_load_hgemm_library`
(kernels/hgemm/tools/utils.py:116-132).
"""
import ctypes
import os

# torch FIRST: the PyTorch-ROCm wheel bundles its own HIP runtime (torch/lib/libamdhip64.so). If libcln_amd.so is
# dlopen'ed before torch, the system runtime (/opt/rocm/lib/libamdhip64.so.7) is mapped first and the process ends
# up with two HIP runtimes; the second one reports "no ROCm-capable device is detected" on the first launch.
import torch  # noqa: F401

from . import manifest

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(PKG_DIR, "lib")

c_void_p, c_int, c_float, c_longlong = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

ARGTYPES = {
    "G3": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "G6": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "H0": [],
    "S3": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "S6": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "FA": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "P3": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p],
    "R1": [c_void_p, c_void_p, c_longlong, c_void_p],
    "SG": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p],
    "XY": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "LN": [c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_void_p],
    "RN": [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p],
    "RP": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "HI": [c_void_p, c_void_p, c_longlong, c_int, c_void_p],
    "UN": [c_void_p, c_void_p, c_longlong, c_void_p],
    "D2": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p],
    "GV": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "TR": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "EM": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
}

_cache = {}


class LibraryMissing(RuntimeError):
    pass


def so_path(so_name):
    return os.path.join(LIBDIR, so_name)


def load_so(so_name):
    """dlopen one of the shared objects and set prototypes for every manifest entry it holds."""
    if so_name in _cache:
        return _cache[so_name]
    path = so_path(so_name)
    if not os.path.exists(path):
        raise LibraryMissing(
            "%s not built. Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback for the kernel path." % path)
    lib = ctypes.CDLL(path)
    for e in manifest.ENTRIES:
        if manifest.SO_OF_LIB[e.lib] != so_name:
            continue
        if e.lib in manifest.OPTIONAL_LIBS and not hasattr(lib, e.name):
            continue  # an optional comparison row that this build does not carry (image without hipBLASLt / ck_tile headers)
        fn = getattr(lib, e.name)  # AttributeError here == ABI hole; let it propagate
        fn.argtypes = ARGTYPES[e.sig]
        fn.restype = c_int
    if so_name == "libcln_amd_probe.so":  # TEST-ONLY library: tuning / ablation hooks (tests/, tools/)
        fn = lib.cln_hgemm_variant
        fn.argtypes = [c_int] * 5 + [c_void_p] * 3 + [c_int] * 5 + [c_void_p]
        fn.restype = c_int
        fn = lib.cln_fa2_variant
        fn.argtypes = [c_int] * 5 + [c_void_p] * 4 + [c_int] * 3 + [c_void_p]
        fn.restype = c_int
    if so_name == "libcln_amd.so":
        fn = lib.cln_describe
        fn.argtypes = [ctypes.c_char_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_char_p, c_int]
        fn.restype = c_int
        # split-K workspace entry points (include/cln_amd.h, round 5)
        lib.cln_hgemm_workspace_bytes.argtypes, lib.cln_hgemm_workspace_bytes.restype = [c_int, c_int, c_int], ctypes.c_size_t
        lib.cln_hgemm_set_workspace.argtypes, lib.cln_hgemm_set_workspace.restype = [c_void_p, ctypes.c_size_t, c_void_p], c_int
        lib.cln_release_workspaces.argtypes, lib.cln_release_workspaces.restype = [], ctypes.c_size_t
        lib.cln_hgemm_workspace_held.argtypes, lib.cln_hgemm_workspace_held.restype = [], ctypes.c_size_t
        lib.cln_hgemm_library_workspace.argtypes, lib.cln_hgemm_library_workspace.restype = [c_int], c_int
    _cache[so_name] = lib
    return lib


def symbol(name):
    e = manifest.BY_NAME[name]
    return getattr(load_so(manifest.SO_OF_LIB[e.lib]), name)


def has_symbol(name):
    """False only for an optional comparison row that the built vendor library does not carry."""
    e = manifest.BY_NAME[name]
    return hasattr(load_so(manifest.SO_OF_LIB[e.lib]), name)
