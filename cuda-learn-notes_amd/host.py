"""Host-side mirror of the reference's per-kernel PyTorch-extension API.

Every reference `lib.<name>(tensors..., knobs...)` exists here with the same name, argument order
and error behaviour, implemented as: dtype/shape checks -> `extern "C" int <name>(ptrs, dims, knobs,
stream)` through ctypes -> status code mapped back to RuntimeError. PyTorch is only plumbing
(device memory + current HIP stream). Tensors must live on the GPU: there is no CPU path.

Reference bindings mirrored (checks and messages):
  CHECK_TORCH_TENSOR_DTYPE -> RuntimeError("values must be torch::kHalf")   kernels/hgemm/naive/hgemm.cu:772-777
  CHECK_TORCH_TENSOR_SHAPE -> RuntimeError("Tensor size mismatch!")          kernels/hgemm/naive/hgemm.cu:778-782
  head-dim switch default  -> RuntimeError("headdim not support!")           flash_attn_mma_share_qkv.cu:860,:882
"""
import os
import threading
import types

import torch

from . import _loader, manifest

_TH_NAME = {
    torch.float16: "torch::kHalf", torch.float32: "torch::kFloat32", torch.bfloat16: "torch::kBFloat16",
    torch.int8: "torch::kInt8", torch.int32: "torch::kInt32",
}
if hasattr(torch, "float8_e4m3fn"):
    _TH_NAME[torch.float8_e4m3fn] = "torch::kFloat8_e4m3fn"
    _TH_NAME[torch.float8_e5m2] = "torch::kFloat8_e5m2"

_STATUS_TEXT = {
    -1: "bad argument (null/misaligned pointer or non-positive size)",
    -2: "unsupported shape",
    -3: "HIP launch failed (hipGetLastError)",
    -4: "rocBLAS row failed (call init_cublas_handle() first?)",
}


def _check_dtype(t, dtype):
    if t.dtype != dtype:
        print("Tensor Info:", t.dtype, t.device, tuple(t.shape))
        raise RuntimeError("values must be %s" % _TH_NAME.get(dtype, str(dtype)))


# The wrappers below run once per launch: for a 4-10 us kernel their cost IS the launch rate (profiles/r04_bw_rows_probe.log: 8.0 us per call
# through torch.cuda.current_stream() and per-tensor device objects, 4 us for torch's own dispatcher). The raw getters of torch._C do the same
# lookups without building Stream / device objects; they exist in every torch 2.x build (Triton's launcher uses them) -- the public API is
# the fallback.
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_device():
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def _check_dev(*ts):
    """Every tensor must be a contiguous HIP tensor on the CURRENT device: the launch goes to
    torch.cuda.current_stream() of the current device with raw data_ptr()s, so a tensor living on another GPU
    would be dereferenced from the wrong device (fault or silent peer access)."""
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("expected a GPU (HIP) tensor, got device=%s: the kernel library has no CPU path"
                               % t.device)
    cur = _current_device() if ts else -1
    for t in ts:
        if t.get_device() != cur:
            raise RuntimeError("tensor on %s but the current device is cuda:%d: wrap the call in "
                               "`with torch.cuda.device(tensor.device):`" % (t.device, cur))
        if not t.is_contiguous():
            raise RuntimeError("tensor must be contiguous (the kernels take raw data_ptr())")


def _check_shape(t, *shape):
    if t.shape != shape:  # torch.Size is a tuple
        raise RuntimeError("Tensor size mismatch!")


def _stream():
    """The raw hipStream_t of torch's current stream on the current device."""
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


# ---- the CPython entry in front of the C-ABI (csrc/pyext/cln_fastcall.c, round 5): argument checks, data_ptr()s, the raw stream and the C call
# in one vectorcall, falling back to the pure-Python wrapper below on ANY failed check or non-zero status (so the error texts stay those of
# this file). Absent module, CPU-only torch build or $CLN_AMD_NO_FASTCALL=1: the wrappers call through ctypes as before.
def _load_fastcall():
    if os.environ.get("CLN_AMD_NO_FASTCALL", "0") == "1" or _raw_device is None or _raw_stream is None:
        return None
    import glob
    import importlib.util
    for path in glob.glob(os.path.join(_loader.LIBDIR, "_cln_fastcall*.so")):
        try:
            spec = importlib.util.spec_from_file_location("_cln_fastcall", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.setup(_raw_device, _raw_stream)
            return mod
        except Exception:  # noqa: BLE001 -- an optional accelerator: any trouble means ctypes
            continue
    return None


_fastcall = _load_fastcall()


def _fast(kind, name, fn, slow, dtype, vt=False, out_dtype=None):
    """`slow` wrapped by the vectorcall entry of kind `kind` (a key of _cln_fastcall.KINDS) when the extension is there."""
    if _fastcall is None:
        return slow
    import ctypes
    addr = ctypes.cast(fn, ctypes.c_void_p).value
    return _fastcall.bind(addr, _fastcall.KINDS[kind], dtype, slow, name, int(bool(vt)), out_dtype)


def _raise(name, rc, unsupported_msg=None):
    if rc == 0:
        return
    if rc == -2 and unsupported_msg:
        raise RuntimeError(unsupported_msg)
    raise RuntimeError("%s: %s (status %d)" % (name, _STATUS_TEXT.get(rc, "error"), rc))


# ------------------------------------------------------------------------------------------------
def _make_g3(name, dtype=torch.float16):
    fn = _loader.symbol(name)

    def f(a, b, c):
        for t in (a, b, c):
            _check_dtype(t, dtype)
        _check_dev(a, b, c)
        M, K = a.size(0), a.size(1)
        N = b.size(1)  # TN operands keep the [K,N] shape (reference as_col_major)
        _check_shape(a, M, K)
        _check_shape(b, K, N)
        _check_shape(c, M, N)
        _raise(name, fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, _stream()),
               "%s: M/N/K must be multiples of the block tile" % name)
    f.__name__ = name
    return _fast("G3", name, fn, f, dtype)


def _make_g6(name, dtype=torch.float16):
    fn = _loader.symbol(name)

    def f(a, b, c, stages, swizzle=False, swizzle_stride=1):
        for t in (a, b, c):
            _check_dtype(t, dtype)
        _check_dev(a, b, c)
        M, K = a.size(0), a.size(1)
        N = b.size(1)
        _check_shape(a, M, K)
        _check_shape(b, K, N)
        _check_shape(c, M, N)
        _raise(name, fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, int(stages), int(bool(swizzle)),
                        int(swizzle_stride), _stream()),
               "%s: M/N/K must be multiples of the block tile" % name)
    f.__name__ = name
    call = _fast("G6", name, fn, f, dtype)
    if not manifest.BY_NAME[name].impl.startswith("best<"):
        return call

    # the run-time dispatched names may split K: the stream's workspace is a tensor of torch's caching allocator, handed to the library before
    # the launch (the library itself never allocates: include/cln_amd.h, workspace block)
    def g(a, b, c, stages, swizzle=False, swizzle_stride=1):
        try:
            need = _ws_need[(a.shape[0], b.shape[1], a.shape[1])]
        except KeyError:
            need = _ws_need_of(a.shape[0], b.shape[1], a.shape[1])
        except Exception:  # noqa: BLE001 -- not tensors / wrong rank: the wrapper below raises the reference's error
            need = 0
        if need:
            _ensure_workspace(need)
        return call(a, b, c, stages, swizzle, swizzle_stride)
    g.__name__ = name
    return g


def _make_h0(name):
    fn = _loader.symbol(name)

    def f():
        _raise(name, fn())
    f.__name__ = name
    return f


def _make_fa(name):
    fn = _loader.symbol(name)
    vt = name in manifest.FA_V_TRANSPOSED

    def f(Q, K, V, O, stages):
        for t in (Q, K, V, O):
            _check_dtype(t, torch.float16)
        _check_dev(Q, K, V, O)
        B, H, N, D = Q.shape
        _check_shape(K, B, H, N, D)
        _check_shape(O, B, H, N, D)
        if vt:
            _check_shape(V, B, H, D, N)
        else:
            _check_shape(V, B, H, N, D)
        rc = fn(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), B, H, N, D, int(stages), _stream())
        if rc == -2:
            if N % 64 != 0 or (D == 256 and N % 128 != 0) or (D > 256 and N % 128 != 0):
                raise RuntimeError("%s: seqlen must be a multiple of 64 (128 for headdim >= 256)" % name)
            raise RuntimeError("headdim not support!")
        _raise(name, rc)
    f.__name__ = name
    return _fast("FA", name, fn, f, torch.float16, vt=vt)


def _make_p3(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if "_f32" in name else torch.float16

    def f(a, b, c):
        for t in (a, b, c):
            _check_dtype(t, dtype)
        _check_dev(a, b, c)
        _check_shape(b, *a.shape)
        _check_shape(c, *a.shape)
        _raise(name, fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.numel(), _stream()))
    f.__name__ = name
    return _fast("P3", name, fn, f, dtype)


def _make_r1(name):
    fn = _loader.symbol(name)
    in_dt, out_dt = (getattr(torch, n) for n in manifest.REDUCE_DTYPES[name])

    def f(x):
        _check_dtype(x, in_dt)
        _check_dev(x)
        # the reference binding allocates a ZEROED result (block_all_reduce.cu:737-738) and its blocks add into it: two dispatches per call. Here the
        # launch overwrites y (csrc/stream_scratch.h: the last block moves the total out of a self-resetting per-stream scratch word): no fill kernel
        y = torch.empty(1, dtype=out_dt, device=x.device)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), x.numel(), _stream()))
        return y
    f.__name__ = name
    return _fast("R1", name, fn, f, in_dt, out_dtype=out_dt)


_SOFTMAX_ONE_BLOCK_MAX = 65536  # == SOFTMAX_ONE_BLOCK_MAX in csrc/softmax.hip (tests/test_host_logic.py holds them equal)


def _make_sg(name):
    fn = _loader.symbol(name)

    def f(x, y):
        _check_dtype(x, torch.float32)
        _check_dtype(y, torch.float32)
        _check_dev(x, y)
        _check_shape(y, *x.shape)
        # reference softmax.cu:419 allocates the zeroed accumulator in the binding; up to _SOFTMAX_ONE_BLOCK_MAX elements ONE workgroup
        # does both passes and overwrites it (csrc/softmax.hip SOFTMAX_ONE_BLOCK_MAX), so no fill kernel is launched for it
        n = x.numel()
        total = (torch.empty if n <= _SOFTMAX_ONE_BLOCK_MAX else torch.zeros)(1, dtype=torch.float32, device=x.device)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), total.data_ptr(), x.numel(), _stream()))
    f.__name__ = name
    return f


def _make_xy(name):
    fn = _loader.symbol(name)
    dtype = torch.float16 if "_f16" in name else torch.float32

    def f(x, y):
        _check_dtype(x, dtype)
        _check_dtype(y, dtype)
        _check_dev(x, y)
        _check_shape(y, *x.shape)
        S, H = x.size(0), x.size(1)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), S, H, _stream()),
               "%s: unsupported H=%d (must be a multiple of the pack width and fit 8 packs x 1024 lanes)"
               % (name, H))
    f.__name__ = name
    return _fast("XY", name, fn, f, dtype)


def _make_ln(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if name.startswith("layer_norm_f32") else torch.float16

    def f(x, y, g, b):
        _check_dtype(x, dtype)
        _check_dtype(y, dtype)
        _check_dev(x, y)
        _check_shape(y, *x.shape)
        N, K = x.size(0), x.size(1)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), float(g), float(b), N, K, _stream()),
               "%s: unsupported K=%d" % (name, K))
    f.__name__ = name
    return _fast("LN", name, fn, f, dtype)


def _make_rn(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if name.startswith("rms_norm_f32") else torch.float16

    def f(x, y, g):
        _check_dtype(x, dtype)
        _check_dtype(y, dtype)
        _check_dev(x, y)
        _check_shape(y, *x.shape)
        N, K = x.size(0), x.size(1)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), float(g), N, K, _stream()), "%s: unsupported K=%d" % (name, K))
    f.__name__ = name
    return _fast("RN", name, fn, f, dtype)


def _make_rp(name):
    fn = _loader.symbol(name)

    def f(x, out, ref_quirk=None):
        """ref_quirk=False (default): the script's torch semantics, pair i of token t rotated by t * theta^(-2i/hidden)
        (rope.py:68-88). ref_quirk=True: the reference CUDA KERNELS' behaviour -- the exponent is an integer division
        that is always 0, so every pair is rotated by t radians (rope.cu:26,:41,:55). Keyword argument added to the
        reference signature `rope_*(x, out)`; $CLN_AMD_ROPE_REF_QUIRK=1 sets the default for callers that cannot
        pass it (read once at import)."""
        _check_dtype(x, torch.float32)
        _check_dtype(out, torch.float32)
        _check_dev(x, out)
        _check_shape(out, *x.shape)
        quirk = int(_ROPE_QUIRK_DEFAULT if ref_quirk is None else bool(ref_quirk))
        _raise(name, fn(x.data_ptr(), out.data_ptr(), x.size(0), x.size(1), quirk, _stream()),
               "%s: hidden size must be a multiple of the pack width" % name)
    f.__name__ = name
    return f


_ROPE_QUIRK_DEFAULT = os.environ.get("CLN_AMD_ROPE_REF_QUIRK", "0") == "1"


def _make_hi(name):
    fn = _loader.symbol(name)

    def f(a):
        _check_dtype(a, torch.int32)
        _check_dev(a)
        # reference binding: M = max(a) on the host, y = zeros(M + 1) (histogram.cu:60-66)
        nbins = int(a.max().item()) + 1 if a.numel() else 1
        y = torch.zeros(max(nbins, 1), dtype=torch.int32, device=a.device)
        _raise(name, fn(a.data_ptr(), y.data_ptr(), a.numel(), max(nbins, 1), _stream()))
        return y
    f.__name__ = name
    return f


def _make_em(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if "_f32" in name else torch.float16

    def f(a, weight, o):
        _check_dtype(a, torch.int32)
        _check_dtype(weight, dtype)
        _check_dtype(o, dtype)
        _check_dev(a, weight, o)
        n, emb = a.size(0), weight.size(1)
        _check_shape(o, n, emb)
        _raise(name, fn(a.data_ptr(), weight.data_ptr(), o.data_ptr(), n, emb, weight.size(0), _stream()),
               "%s: embedding size must be a multiple of the pack width" % name)
    f.__name__ = name
    return f


def _make_un(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if "_f32" in name else torch.float16

    def f(x, y):
        _check_dtype(x, dtype)
        _check_dtype(y, dtype)
        _check_dev(x, y)
        _check_shape(y, *x.shape)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), x.numel(), _stream()))
    f.__name__ = name
    return _fast("UN", name, fn, f, dtype)


def _make_d2(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if name.startswith("dot_prod_f32") else torch.float16

    def f(a, b):
        _check_dtype(a, dtype)
        _check_dtype(b, dtype)
        _check_dev(a, b)
        _check_shape(b, *a.shape)
        prod = torch.empty(1, dtype=torch.float32, device=a.device)  # reference dot_product.cu:236-238 zeroes it; here the launch overwrites it (stream_scratch.h)
        _raise(name, fn(a.data_ptr(), b.data_ptr(), prod.data_ptr(), a.numel(), _stream()))
        return prod
    f.__name__ = name
    return _fast("D2", name, fn, f, dtype, out_dtype=torch.float32)


def _make_gv(name):
    fn = _loader.symbol(name)
    dtype = torch.float32 if name.startswith("sgemv") else torch.float16
    kmsg = {"k32": "K must be multiple of 32", "k128": "K must be multiple of 128", "k16": "K must be 16"}[name.split("_")[1]]

    def f(a, x, y):
        for t in (a, x, y):
            _check_dtype(t, dtype)
        _check_dev(a, x, y)
        M, K = a.size(0), a.size(1)
        _check_shape(x, K, 1)
        _check_shape(y, M, 1)
        _raise(name, fn(a.data_ptr(), x.data_ptr(), y.data_ptr(), M, K, _stream()), kmsg)  # sgemv.cu:134-137
    f.__name__ = name
    return f


def _make_tr(name):
    fn = _loader.symbol(name)

    def f(x, y):
        _check_dtype(x, torch.float32)
        _check_dtype(y, torch.float32)
        _check_dev(x, y)
        M, N = x.size(0), x.size(1)
        _check_shape(y, N, M)
        _raise(name, fn(x.data_ptr(), y.data_ptr(), M, N, _stream()),
               "%s: rows/cols must be multiples of 4 (x4 rungs) / 64 (shared rungs)" % name)
    f.__name__ = name
    return f


def _make_s3(name):
    return _make_g3(name, torch.float32)


def _make_s6(name):
    return _make_g6(name, torch.float32)


_MAKERS = {"S3": _make_s3, "S6": _make_s6, "D2": _make_d2, "GV": _make_gv, "TR": _make_tr, "UN": _make_un, "HI": _make_hi, "EM": _make_em, "G3": _make_g3, "G6": _make_g6, "H0": _make_h0, "FA": _make_fa, "P3": _make_p3, "R1": _make_r1,
           "SG": _make_sg, "XY": _make_xy, "LN": _make_ln, "RN": _make_rn, "RP": _make_rp}


def load_lib(*groups):
    """Return a module-like object whose attributes are the exported functions of the given lib
    groups ('hgemm' also pulls in the vendor row, as the reference's single hgemm module does)."""
    ns = types.SimpleNamespace()
    for e in manifest.ENTRIES:
        if e.lib in groups:
            if e.lib in manifest.OPTIONAL_LIBS and not _loader.has_symbol(e.name):
                continue  # optional comparison row absent from this build: the attribute is simply not there (callers test with hasattr / try)
            setattr(ns, e.name, _MAKERS[e.sig](e.name))
    return ns


# ---- split-K workspace of the best-dispatch HGEMM names (include/cln_amd.h; not part of the reference surface) -----------------------------
# Round 6 ("no hidden workspace"): the LIBRARY owns nothing. Every (device, stream) that runs a split-K shape through this module gets
# ONE uint8 tensor from torch's caching allocator (allocated on that stream, so its lifetime follows torch's stream semantics), registered with
# cln_hgemm_set_workspace and grown when a larger shape shows up. At most _WS_MAX of them are kept (least recently used withdrawn first); one that a
# stream capture has used is pinned -- the graph holds its address -- until release_workspaces(). A tensor the CALLER registered
# (hgemm_set_workspace) is never replaced or evicted.
_WS_MAX = 8
_ws_need = {}     # (M, N, K) -> bytes the plan of the shape uses (0: single-pass)
_workspaces = {}  # (device, raw stream) -> [tensor, bytes, user, pinned, clock]
_ws_clock = 0
_ws_lock = threading.Lock()  # two host threads on one stream: one of them registers the tensor, the other finds it


def hgemm_workspace_bytes(M, N, K):
    """Bytes of fp32 workspace the split-K / tail-split plan of (M, N, K) uses; 0 = the shape runs single-pass."""
    return int(_loader.load_so("libcln_amd.so").cln_hgemm_workspace_bytes(int(M), int(N), int(K)))


def _ws_need_of(M, N, K):
    need = _ws_need[(M, N, K)] = hgemm_workspace_bytes(M, N, K)
    if len(_ws_need) > 4096:
        _ws_need.clear()
    return need


def _ws_withdraw(key):
    lib = _loader.load_so("libcln_amd.so")
    with torch.cuda.device(key[0]):
        lib.cln_hgemm_set_workspace(None, 0, key[1])
    _workspaces.pop(key, None)


def _ensure_workspace(need):
    """The current stream's workspace holds at least `need` bytes after this call -- or the launch runs single-pass (growth is not allowed while
    the stream is being captured: an allocation made under capture belongs to the graph's private pool)."""
    global _ws_clock
    key = (_current_device(), _stream())
    ent = _workspaces.get(key)
    _ws_clock += 1
    if ent is not None and (ent[1] >= need or ent[2]):
        ent[4] = _ws_clock
        if not ent[3] and torch.cuda.is_current_stream_capturing():
            ent[3] = True
        return
    with _ws_lock:
        _grow_workspace(key, need)


def _grow_workspace(key, need):
    ent = _workspaces.get(key)
    if ent is not None and (ent[1] >= need or ent[2]):
        return
    if torch.cuda.is_current_stream_capturing() or (ent is not None and ent[3]):
        return
    if ent is None:
        free = [k for k, e in _workspaces.items() if not e[2] and not e[3]]
        while len(free) >= _WS_MAX:
            lru = min(free, key=lambda k: _workspaces[k][4])
            free.remove(lru)
            _ws_withdraw(lru)
    size = max(16 << 20, (need + (2 << 20) - 1) // (2 << 20) * (2 << 20))  # whole 2-MiB pages of the caching allocator, 16 MiB at least (few regrowths)
    buf = torch.empty(size, dtype=torch.uint8, device="cuda:%d" % key[0])
    lib = _loader.load_so("libcln_amd.so")
    _raise("cln_hgemm_set_workspace", lib.cln_hgemm_set_workspace(buf.data_ptr(), size, key[1]))
    _workspaces[key] = [buf, size, False, False, _ws_clock]  # the old tensor (if any) goes back to the allocator: launches queued on ITS stream run before any reuse


def hgemm_set_workspace(buf):
    """Give the launches on torch's CURRENT stream a caller-owned workspace: `buf` is a contiguous GPU tensor (any dtype; e.g.
    torch.empty(nbytes, dtype=torch.uint8, device="cuda")), or None to go back to the one this module keeps per stream. The library zeroes the
    first 4 KiB on the stream; shapes that need more than the buffer holds run single-pass."""
    lib = _loader.load_so("libcln_amd.so")
    key = (_current_device(), _stream())
    if buf is None:
        _raise("cln_hgemm_set_workspace", lib.cln_hgemm_set_workspace(None, 0, key[1]))
        _workspaces.pop(key, None)
        return
    _check_dev(buf)
    nbytes = buf.numel() * buf.element_size()
    _raise("cln_hgemm_set_workspace", lib.cln_hgemm_set_workspace(buf.data_ptr(), nbytes, key[1]))
    _workspaces[key] = [buf, nbytes, True, False, _ws_clock]


def release_workspaces():
    """Withdraw every workspace this module registered (their tensors go back to torch's allocator), and free what the library itself holds: the
    scalar-result kernels' ticket slabs and -- after cln_hgemm_library_workspace(1) only -- library-owned split-K buffers. Returns the bytes the
    LIBRARY freed. Call it only when no graph that captured a split-K launch will be replayed again."""
    _workspaces.clear()
    return int(_loader.load_so("libcln_amd.so").cln_release_workspaces())


def hgemm_workspace_held():
    """Bytes of LIBRARY-owned split-K workspace this process currently holds (0 unless cln_hgemm_library_workspace(1) was called)."""
    return int(_loader.load_so("libcln_amd.so").cln_hgemm_workspace_held())


def hgemm_workspace_tensors():
    """(device, raw stream) -> bytes of the workspace tensors this module currently keeps registered (torch-owned)."""
    return {k: e[1] for k, e in _workspaces.items()}


def hgemm_library_workspace(enable):
    """C callers' opt-in to library-owned (hipMalloc) workspaces; this module never needs it. Returns the previous setting."""
    return bool(_loader.load_so("libcln_amd.so").cln_hgemm_library_workspace(int(bool(enable))))


def hgemm_variant(kind, layout, tile, bk, stages, a, b, c, swizzle=0, swizzle_stride=1):
    """Tuning hook (not part of the reference surface; lives in the TEST-ONLY libcln_amd_probe.so): run an explicit
    tile/BK/stage variant."""
    _check_dev(a, b, c)
    M, K = a.size(0), a.size(1)
    N = b.size(1)
    fn = _loader.load_so("libcln_amd_probe.so").cln_hgemm_variant
    rc = fn(kind, layout, tile, bk, stages, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, int(swizzle),
            int(swizzle_stride), _stream())
    _raise("cln_hgemm_variant", rc, "variant not available for this shape/LDS budget")


def fa2_variant(D_nw_vt_opt_abl, Q, K, V, O):
    """Tuning hook (not part of the reference surface; lives in the TEST-ONLY libcln_amd_probe.so): run an explicit
    FlashAttention kernel variant."""
    _check_dev(Q, K, V, O)
    nw, vt, opt, abl = D_nw_vt_opt_abl
    B, H, N, D = Q.shape
    fn = _loader.load_so("libcln_amd_probe.so").cln_fa2_variant
    rc = fn(D, nw, vt, opt, abl, Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), B, H, N, _stream())
    _raise("cln_fa2_variant", rc, "variant not instantiated / shape not supported")
