"""cuda-learn-notes_amd: MI355X (gfx950) native drop-in for the HGEMM / FlashAttention-2 hot path
of DefTruth/CUDA-Learn-Notes and its supporting reduce/softmax/layer-norm/rms-norm/rope/elementwise
kernels. Host code is Python on PyTorch-ROCm calling hand-written HIP kernels through a C-ABI
(see include/cln_amd.h)."""
from . import manifest  # noqa: F401


def build(verbose=False, force=False):
    from . import _build
    return _build.build(verbose=verbose, force=force)


def load(*groups):
    """`lib = load('elementwise')` ~ reference `lib = load(name='elementwise_lib', sources=[...])`."""
    from . import host
    return host.load_lib(*groups)


def hgemm_lib():
    """Mirror of `import toy_hgemm` / try_load_hgemm_library (kernels/hgemm/tools/utils.py:116-132)."""
    return load("hgemm", "hgemm_vendor", "hgemm_vendor_lt")


def flash_attn_lib():
    return load("flash_attn")
