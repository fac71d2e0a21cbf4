"""The CPython entry in front of the C-ABI (cuda-learn-notes_amd/csrc/pyext/cln_fastcall.c, round 5) without a GPU: a fake C-ABI library
(gcc, built here) records what it is called with; CPU tensors stand in for device tensors (the entry only uses the tensors' Python
methods -- dtype, get_device(), is_contiguous(), data_ptr(), shape -- and the two getters handed to setup()).

Held here: every signature class passes pointers / sizes / knobs / stream exactly as host.py's ctypes path does, and ANY failed check,
keyword argument, wrong arity or non-zero status goes to the pure-Python fallback (which owns the reference's error texts)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE_C = r"""
#include <string.h>
long long g_i[16]; double g_f[4]; void* g_p[8]; int g_rc = 0; int g_calls = 0;
#define REC() (g_calls++)
int f_p3(void* a, void* b, void* c, long long n, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_p[2]=c; g_i[0]=n; g_p[7]=s; return g_rc; }
int f_un(void* a, void* b, long long n, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_i[0]=n; g_p[7]=s; return g_rc; }
int f_xy(void* a, void* b, int S, int H, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_i[0]=S; g_i[1]=H; g_p[7]=s; return g_rc; }
int f_ln(void* a, void* b, float g, float bb, int N, int K, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_f[0]=g; g_f[1]=bb; g_i[0]=N; g_i[1]=K; g_p[7]=s; return g_rc; }
int f_rn(void* a, void* b, float g, int N, int K, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_f[0]=g; g_i[0]=N; g_i[1]=K; g_p[7]=s; return g_rc; }
int f_g3(void* a, void* b, void* c, int M, int N, int K, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_p[2]=c; g_i[0]=M; g_i[1]=N; g_i[2]=K; g_p[7]=s; return g_rc; }
int f_g6(void* a, void* b, void* c, int M, int N, int K, int st, int sw, int stride, void* s) {
  REC(); g_p[0]=a; g_p[1]=b; g_p[2]=c; g_i[0]=M; g_i[1]=N; g_i[2]=K; g_i[3]=st; g_i[4]=sw; g_i[5]=stride; g_p[7]=s; return g_rc; }
int f_fa(void* q, void* k, void* v, void* o, int B, int H, int N, int D, int st, void* s) {
  REC(); g_p[0]=q; g_p[1]=k; g_p[2]=v; g_p[3]=o; g_i[0]=B; g_i[1]=H; g_i[2]=N; g_i[3]=D; g_i[4]=st; g_p[7]=s; return g_rc; }
int f_r1(void* a, void* y, long long n, void* s) { REC(); g_p[0]=a; g_p[1]=y; g_i[0]=n; g_p[7]=s; if (!g_rc) *(float*)y = 42.0f; return g_rc; }
int f_d2(void* a, void* b, void* y, long long n, void* s) { REC(); g_p[0]=a; g_p[1]=b; g_p[2]=y; g_i[0]=n; g_p[7]=s; if (!g_rc) *(float*)y = 7.0f; return g_rc; }
"""
STREAM = 0x5150


@pytest.fixture(scope="module")
def fx(tmp_path_factory):
    from cuda_learn_notes_amd import host
    if host._fastcall is None:
        pytest.skip("_cln_fastcall not built (no C compiler / Python.h)")
    d = tmp_path_factory.mktemp("fake")
    (d / "fake.c").write_text(FAKE_C)
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", str(d / "fake.c"), "-o", str(d / "libfake.so")], check=True)
    lib = ctypes.CDLL(str(d / "libfake.so"))
    fc = host._fastcall
    fc.setup(lambda: -1, lambda dev: STREAM)  # CPU tensors report device -1
    yield fc, lib
    fc.setup(host._raw_device, host._raw_stream)


class Rec:
    def __init__(self, lib):
        self.lib = lib
        self.i = (ctypes.c_longlong * 16).in_dll(lib, "g_i")
        self.f = (ctypes.c_double * 4).in_dll(lib, "g_f")
        self.p = (ctypes.c_void_p * 8).in_dll(lib, "g_p")
        self.rc = ctypes.c_int.in_dll(lib, "g_rc")
        self.calls = ctypes.c_int.in_dll(lib, "g_calls")


def bind(fc, lib, sym, kind, dtype, slow, **kw):
    addr = ctypes.cast(getattr(lib, sym), ctypes.c_void_p).value
    return fc.bind(addr, fc.KINDS[kind], dtype, slow, sym, int(kw.get("vt", 0)), kw.get("out_dtype"))


def test_every_signature_class_passes_the_same_arguments_as_the_ctypes_path(fx):
    fc, lib = fx
    r = Rec(lib)
    slow_calls = []
    slow = lambda *a, **k: slow_calls.append((a, k)) or "slow"
    f32 = torch.float32
    a, b, c = torch.zeros(6, 10), torch.ones(6, 10), torch.empty(6, 10)
    assert bind(fc, lib, "f_p3", "P3", f32, slow)(a, b, c) is None
    assert (r.p[0], r.p[1], r.p[2], r.i[0], r.p[7]) == (a.data_ptr(), b.data_ptr(), c.data_ptr(), 60, STREAM)
    assert bind(fc, lib, "f_un", "UN", f32, slow)(a, c) is None and (r.p[0], r.p[1], r.i[0]) == (a.data_ptr(), c.data_ptr(), 60)
    assert bind(fc, lib, "f_xy", "XY", f32, slow)(a, c) is None and (r.i[0], r.i[1]) == (6, 10)
    assert bind(fc, lib, "f_ln", "LN", f32, slow)(a, c, 1.5, -0.25) is None and (r.f[0], r.f[1], r.i[0], r.i[1]) == (1.5, -0.25, 6, 10)
    assert bind(fc, lib, "f_rn", "RN", f32, slow)(a, c, 2) is None and (r.f[0], r.i[0], r.i[1]) == (2.0, 6, 10)
    A, B, C = torch.zeros(8, 5), torch.zeros(5, 3), torch.zeros(8, 3)
    assert bind(fc, lib, "f_g3", "G3", f32, slow)(A, B, C) is None and (r.i[0], r.i[1], r.i[2]) == (8, 3, 5)
    g6 = bind(fc, lib, "f_g6", "G6", f32, slow)
    assert g6(A, B, C, 3, True, 2048) is None and [r.i[k] for k in range(6)] == [8, 3, 5, 3, 1, 2048]
    assert g6(A, B, C, 2) is None and [r.i[k] for k in range(3, 6)] == [2, 0, 1]  # defaults swizzle=False, swizzle_stride=1
    q = torch.zeros(2, 3, 16, 8)
    vt = torch.zeros(2, 3, 8, 16)
    assert bind(fc, lib, "f_fa", "FA", f32, slow)(q, q, q, q.clone(), 2) is None and [r.i[k] for k in range(5)] == [2, 3, 16, 8, 2]
    assert bind(fc, lib, "f_fa", "FA", f32, slow, vt=1)(q, q, vt, q.clone(), 1) is None and r.p[2] == vt.data_ptr()
    y = bind(fc, lib, "f_r1", "R1", f32, slow, out_dtype=torch.float32)(a)
    assert y.shape == (1,) and y.dtype == torch.float32 and y.item() == 42.0 and r.p[1] == y.data_ptr() and r.i[0] == 60
    yi = bind(fc, lib, "f_r1", "R1", torch.int8, slow, out_dtype=torch.int32)(torch.zeros(33, dtype=torch.int8))
    assert yi.dtype == torch.int32 and r.i[0] == 33
    d = bind(fc, lib, "f_d2", "D2", f32, slow, out_dtype=torch.float32)(a, b)
    assert d.item() == 7.0 and (r.p[0], r.p[1], r.p[2]) == (a.data_ptr(), b.data_ptr(), d.data_ptr())
    assert slow_calls == []


def test_any_failed_check_goes_to_the_python_fallback(fx):
    fc, lib = fx
    r = Rec(lib)
    seen = []
    slow = lambda *a, **k: seen.append(len(a)) or "slow"
    f = bind(fc, lib, "f_p3", "P3", torch.float32, slow)
    a, b, c = torch.zeros(4, 4), torch.zeros(4, 4), torch.zeros(4, 4)
    n0 = r.calls.value
    assert f(a.half(), b, c) == "slow"                 # dtype
    assert f(a, b.t()[:, :2], c) == "slow"            # not contiguous / shape
    assert f(a, torch.zeros(4, 5), c) == "slow"       # shape mismatch
    assert f(a, b) == "slow"                           # arity
    assert f(a, b, c=c) == "slow"                      # keyword argument
    assert f(a, b, "x") == "slow"                      # not a tensor
    assert r.calls.value == n0                          # the C function was never reached
    r.rc.value = -2
    try:
        assert f(a, b, c) == "slow" and r.calls.value == n0 + 1   # a non-zero status: the fallback re-runs the call and raises the reference's text
    finally:
        r.rc.value = 0
    g6 = bind(fc, lib, "f_g6", "G6", torch.float32, slow)
    assert g6(torch.zeros(8, 5), torch.zeros(6, 3), torch.zeros(8, 3), 2) == "slow"   # b is not [K, N]
    ln = bind(fc, lib, "f_ln", "LN", torch.float32, slow)
    assert ln(a, c, "g", 0.0) == "slow"
    assert len(seen) == 9
    assert f.__name__ == "f_p3" and f.__wrapped__ is slow and "f_p3" in repr(f)


def test_device_mismatch_is_a_fallback(fx):
    fc, lib = fx
    r = Rec(lib)
    fc.setup(lambda: 0, lambda dev: STREAM)  # "current device 0": CPU tensors (device -1) must not reach the C function
    try:
        n0 = r.calls.value
        f = bind(fc, lib, "f_un", "UN", torch.float32, lambda *a: "slow")
        assert f(torch.zeros(3), torch.zeros(3)) == "slow" and r.calls.value == n0
    finally:
        fc.setup(lambda: -1, lambda dev: STREAM)


def test_host_wrappers_are_fast_entries_and_keep_the_reference_errors():
    from cuda_learn_notes_amd import host
    import cuda_learn_notes_amd as pkg
    if host._fastcall is None:
        pytest.skip("_cln_fastcall not built")
    ew = pkg.load("elementwise")
    assert type(ew.elementwise_add_f32).__name__ == "FastFn"
    x = torch.zeros(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ew.elementwise_add_f32(x, x, x)
    with pytest.raises(RuntimeError, match="values must be torch::kFloat32"):
        ew.elementwise_add_f32(x.half(), x, x)
