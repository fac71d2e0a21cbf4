"""`import toy_hgemm` -- the module name the reference's HGEMM scripts import after `python setup.py install`
(kernels/hgemm/setup.py; kernels/hgemm/tools/utils.py:116-132 `try_load_hgemm_library`). Every function the reference's pybind
module exports (kernels/hgemm/pybind/hgemm.cc:58-107) is an attribute of this module, bound to the C-ABI of libcln_amd.so on first
use (the library is dlopen'ed lazily, so importing this module needs neither a GPU nor a finished build)."""
import cuda_learn_notes_amd as _pkg

_lib = None
__all__ = [e.name for e in _pkg.manifest.ENTRIES if e.lib in ("hgemm", "hgemm_vendor")]


_EXTRA = [e.name for e in _pkg.manifest.ENTRIES if e.lib == "hgemm_vendor_lt"]  # our own hipBLASLt comparison rows (cln_ prefix): not reference names
_lt = None


def __getattr__(name):
    global _lib, _lt
    if name in __all__:
        if _lib is None:
            _lib = _pkg.load("hgemm", "hgemm_vendor")
        return getattr(_lib, name)
    if name in _EXTRA:
        if _lt is None:
            _lt = _pkg.load("hgemm_vendor_lt")
        return getattr(_lt, name)
    raise AttributeError("module 'toy_hgemm' has no attribute %r" % name)


def __dir__():
    return sorted(__all__)
