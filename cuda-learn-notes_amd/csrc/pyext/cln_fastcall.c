/* _cln_fastcall -- a CPython entry in front of the C-ABI (round 5).
 *
 * The reference binds every kernel through pybind11 (e.g. kernels/elementwise/elementwise.cu:170-177): one C++ call per launch. Here
 * the C-ABI library was reached through ctypes: ~1 us of argument marshalling per call plus the Python-level checks of host.py --
 * 5.1-5.4 us per call against torch's own 3.5-4.1 us, which is what the reference's timing protocol (elementwise.py:25-56: a host clock
 * around N launches) measures once a kernel is shorter than that. This module is the same boundary WITHOUT torch headers and without
 * ctypes: `bind(addr, kind, dtype, slow, name)` returns a vectorcall object that
 *     1. checks the tensor arguments the way host.py does (dtype singleton, GPU + current device, contiguity, shapes) through the
 *        tensors' own Python methods (PyObject_GetAttr / CallMethodNoArgs on interned names -- no torch C++ API),
 *     2. takes data_ptr()s, the sizes and the current raw HIP stream, and
 *     3. calls the C-ABI function pointer directly.
 * ANY failed check, keyword argument, unexpected arity or non-zero status falls back to `slow` -- the pure-Python wrapper of host.py --
 * which repeats the checks and raises the reference's RuntimeError texts (or handles the case): the fast path never invents behaviour.
 * Build: gcc -O2 -shared -fPIC (cuda-learn-notes_amd/_build.py); optional -- host.py uses ctypes when the module is absent.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stddef.h>
#include <stdint.h>

enum { K_P3 = 1, K_UN, K_XY, K_LN, K_RN, K_G3, K_G6, K_FA, K_R1, K_D2, K_RAW };

typedef int (*fn_p3)(void*, void*, void*, long long, void*);
typedef int (*fn_un)(void*, void*, long long, void*);
typedef int (*fn_xy)(void*, void*, int, int, void*);
typedef int (*fn_ln)(void*, void*, float, float, int, int, void*);
typedef int (*fn_rn)(void*, void*, float, int, int, void*);
typedef int (*fn_g3)(void*, void*, void*, int, int, int, void*);
typedef int (*fn_g6)(void*, void*, void*, int, int, int, int, int, int, void*);
typedef int (*fn_fa)(void*, void*, void*, void*, int, int, int, int, int, void*);
typedef int (*fn_r1)(void*, void*, long long, void*);

typedef struct {
  PyObject_HEAD
  vectorcallfunc vc;
  void* fn;
  int kind;
  int vt;            /* K_FA: V is [B,H,D,N] */
  PyObject* dtype;   /* expected torch.dtype singleton of the tensor arguments */
  PyObject* out_dtype; /* K_R1 / K_D2: dtype of the [1] result */
  PyObject* slow;    /* host.py's pure-Python wrapper */
  PyObject* name;
} FastFn;

static PyObject *s_dtype, *s_data_ptr, *s_get_device, *s_is_contiguous, *s_shape, *s_numel, *s_new_empty, *s_dtype_kw;
static PyObject *g_get_device, *g_get_stream; /* torch._C._cuda_getDevice, torch._C._cuda_getCurrentRawStream */
static PyObject *g_one_tuple;                 /* ((1,),) */

typedef struct {
  void* ptr;
  PyObject* shape; /* new reference (tuple subclass torch.Size) */
} TInfo;

/* 0 = ok; -1 = check failed or Python error (error cleared): take the slow path */
static int tensor_info(PyObject* t, PyObject* want_dtype, long cur_dev, TInfo* out) {
  out->shape = NULL;
  PyObject* d = PyObject_GetAttr(t, s_dtype);
  if (!d) goto fail;
  int ok = d == want_dtype;
  Py_DECREF(d);
  if (!ok) return -1;
  PyObject* r = PyObject_CallMethodNoArgs(t, s_get_device); /* -1 for a CPU tensor */
  if (!r) goto fail;
  long dev = PyLong_AsLong(r);
  Py_DECREF(r);
  if (dev != cur_dev) return -1;
  r = PyObject_CallMethodNoArgs(t, s_is_contiguous);
  if (!r) goto fail;
  ok = r == Py_True;
  Py_DECREF(r);
  if (!ok) return -1;
  r = PyObject_CallMethodNoArgs(t, s_data_ptr);
  if (!r) goto fail;
  out->ptr = PyLong_AsVoidPtr(r);
  Py_DECREF(r);
  if (PyErr_Occurred()) goto fail;
  out->shape = PyObject_GetAttr(t, s_shape);
  if (!out->shape || !PyTuple_Check(out->shape)) { Py_CLEAR(out->shape); goto fail; }
  /* every extent a plain int that fits a long long (symbolic sizes, overflow: slow path) -- dim() below can then never fail or leave an exception set */
  for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(out->shape); ++i) {
    PyObject* e = PyTuple_GET_ITEM(out->shape, i);
    if (!PyLong_CheckExact(e) || (PyLong_AsLongLong(e) == -1 && PyErr_Occurred())) { Py_CLEAR(out->shape); goto fail; }
  }
  return 0;
fail:
  PyErr_Clear();
  return -1;
}
static inline long long dim(PyObject* shape, Py_ssize_t i) { return PyLong_AsLongLong(PyTuple_GET_ITEM(shape, i)); }
static long long numel_of(PyObject* shape) {
  long long n = 1;
  for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(shape); ++i) n *= dim(shape, i);
  return n;
}
static int same_shape(PyObject* a, PyObject* b) {
  Py_ssize_t n = PyTuple_GET_SIZE(a);
  if (PyTuple_GET_SIZE(b) != n) return 0;
  for (Py_ssize_t i = 0; i < n; ++i)
    if (dim(a, i) != dim(b, i)) return 0;
  return 1;
}
static int shape_is(PyObject* s, int nd, long long d0, long long d1, long long d2, long long d3) {
  if (PyTuple_GET_SIZE(s) != nd) return 0;
  const long long want[4] = {d0, d1, d2, d3};
  for (int i = 0; i < nd; ++i)
    if (dim(s, i) != want[i]) return 0;
  return 1;
}
static int as_float(PyObject* o, float* out) {
  double v = PyFloat_AsDouble(o);
  if (v == -1.0 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
  *out = (float)v;
  return 0;
}
static int as_int(PyObject* o, int* out) { /* int(o): bools and ints */
  long v = PyLong_AsLong(o);
  if (v == -1 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
  if (v < INT32_MIN || v > INT32_MAX) return -1; /* the slow path's int() hands ctypes the same value and ctypes raises */
  *out = (int)v;
  return 0;
}

/* The C-ABI call runs WITHOUT the GIL: the callee may block -- workspace bookkeeping under its mutex, a hipDeviceSynchronize when a 65th
 * stream asks for a scratch slot -- and ctypes, the path this entry replaces, released it too. */
#define CLN_CALL(rc, expr) do { Py_BEGIN_ALLOW_THREADS (rc) = (expr); Py_END_ALLOW_THREADS } while (0) /* `expr` touches no Python object */

static PyObject* fast_call(PyObject* self_, PyObject* const* args, size_t nargsf, PyObject* kwnames) {
  FastFn* self = (FastFn*)self_;
  const Py_ssize_t nargs = PyVectorcall_NARGS(nargsf);
  TInfo t[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  int nt = 0, rc = -100;
  PyObject* result = NULL; /* K_R1 / K_D2: the [1] tensor handed back */
  if (kwnames && PyTuple_GET_SIZE(kwnames) > 0) goto slow;
  {
    static const int n_tensors[] = {0, 3, 2, 2, 2, 2, 3, 3, 4, 1, 2, 0};
    static const int n_args[] = {0, 3, 2, 2, 4, 3, 3, -6, 5, 1, 2, 0}; /* -6: 4..6 (defaults swizzle=False, swizzle_stride=1) */
    const int need = n_args[self->kind];
    if (need >= 0 ? nargs != need : (nargs < 4 || nargs > 6)) goto slow;
    PyObject* r = PyObject_CallNoArgs(g_get_device);
    if (!r) { PyErr_Clear(); goto slow; }
    const long cur = PyLong_AsLong(r);
    Py_DECREF(r);
    nt = n_tensors[self->kind];
    for (int i = 0; i < nt; ++i)
      if (tensor_info(args[i], self->dtype, cur, &t[i]) != 0) { nt = i + 1; goto slow; }
    PyObject* dv = PyLong_FromLong(cur);
    r = dv ? PyObject_CallOneArg(g_get_stream, dv) : NULL;
    Py_XDECREF(dv);
    if (!r) { PyErr_Clear(); goto slow; }
    void* stream = PyLong_AsVoidPtr(r);
    Py_DECREF(r);
    if (PyErr_Occurred()) { PyErr_Clear(); goto slow; }
    /* extents read while the GIL is held (the calls below run without it) */
    const long long n0 = nt > 0 ? numel_of(t[0].shape) : 0;
    const int two_d = nt > 0 && PyTuple_GET_SIZE(t[0].shape) == 2;
    const long long r0l = two_d ? dim(t[0].shape, 0) : 0, c0l = two_d ? dim(t[0].shape, 1) : 0;
    if (r0l > INT32_MAX || c0l > INT32_MAX) goto slow;
    const int r0 = (int)r0l, c0 = (int)c0l;
    switch (self->kind) {
      case K_P3:
        if (!same_shape(t[0].shape, t[1].shape) || !same_shape(t[0].shape, t[2].shape)) goto slow;
        CLN_CALL(rc, ((fn_p3)self->fn)(t[0].ptr, t[1].ptr, t[2].ptr, n0, stream));
        break;
      case K_UN:
        if (!same_shape(t[0].shape, t[1].shape)) goto slow;
        CLN_CALL(rc, ((fn_un)self->fn)(t[0].ptr, t[1].ptr, n0, stream));
        break;
      case K_XY:
        if (PyTuple_GET_SIZE(t[0].shape) != 2 || !same_shape(t[0].shape, t[1].shape)) goto slow;
        CLN_CALL(rc, ((fn_xy)self->fn)(t[0].ptr, t[1].ptr, r0, c0, stream));
        break;
      case K_LN: {
        float g, b;
        if (PyTuple_GET_SIZE(t[0].shape) != 2 || !same_shape(t[0].shape, t[1].shape) || as_float(args[2], &g) || as_float(args[3], &b)) goto slow;
        CLN_CALL(rc, ((fn_ln)self->fn)(t[0].ptr, t[1].ptr, g, b, r0, c0, stream));
        break;
      }
      case K_RN: {
        float g;
        if (PyTuple_GET_SIZE(t[0].shape) != 2 || !same_shape(t[0].shape, t[1].shape) || as_float(args[2], &g)) goto slow;
        CLN_CALL(rc, ((fn_rn)self->fn)(t[0].ptr, t[1].ptr, g, r0, c0, stream));
        break;
      }
      case K_G3:
      case K_G6: {
        if (PyTuple_GET_SIZE(t[0].shape) != 2 || PyTuple_GET_SIZE(t[1].shape) != 2) goto slow;
        const long long M = dim(t[0].shape, 0), K = dim(t[0].shape, 1), N = dim(t[1].shape, 1);
        if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || !shape_is(t[1].shape, 2, K, N, 0, 0) || !shape_is(t[2].shape, 2, M, N, 0, 0)) goto slow;
        if (self->kind == K_G3) {
          CLN_CALL(rc, ((fn_g3)self->fn)(t[0].ptr, t[1].ptr, t[2].ptr, (int)M, (int)N, (int)K, stream));
        } else {
          int stages, stride = 1, sw = 0;
          if (as_int(args[3], &stages)) goto slow;
          if (nargs > 4) { sw = PyObject_IsTrue(args[4]); if (sw < 0) { PyErr_Clear(); goto slow; } }
          if (nargs > 5 && as_int(args[5], &stride)) goto slow;
          CLN_CALL(rc, ((fn_g6)self->fn)(t[0].ptr, t[1].ptr, t[2].ptr, (int)M, (int)N, (int)K, stages, sw, stride, stream));
        }
        break;
      }
      case K_FA: {
        int stages;
        if (PyTuple_GET_SIZE(t[0].shape) != 4 || as_int(args[4], &stages)) goto slow;
        const long long B = dim(t[0].shape, 0), H = dim(t[0].shape, 1), N = dim(t[0].shape, 2), D = dim(t[0].shape, 3);
        if (!same_shape(t[0].shape, t[1].shape) || !same_shape(t[0].shape, t[3].shape)) goto slow;
        if (!(self->vt ? shape_is(t[2].shape, 4, B, H, D, N) : same_shape(t[0].shape, t[2].shape))) goto slow;
        CLN_CALL(rc, ((fn_fa)self->fn)(t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, (int)B, (int)H, (int)N, (int)D, stages, stream));
        break;
      }
      case K_R1:
      case K_D2: { /* x [, b same shape] -> fresh [1] tensor of out_dtype; the kernels need NO zeroed result (self-resetting stream scratch) */
        if (self->kind == K_D2 && !same_shape(t[0].shape, t[1].shape)) goto slow;
        PyObject* kw = PyDict_New();
        if (!kw || PyDict_SetItem(kw, s_dtype_kw, self->out_dtype) != 0) { Py_XDECREF(kw); PyErr_Clear(); goto slow; }
        PyObject* meth = PyObject_GetAttr(args[0], s_new_empty);
        result = meth ? PyObject_Call(meth, g_one_tuple, kw) : NULL;
        Py_XDECREF(meth);
        Py_DECREF(kw);
        if (!result) { PyErr_Clear(); goto slow; }
        PyObject* p = PyObject_CallMethodNoArgs(result, s_data_ptr);
        void* yp = p ? PyLong_AsVoidPtr(p) : NULL;
        Py_XDECREF(p);
        if (!p || PyErr_Occurred()) { PyErr_Clear(); Py_CLEAR(result); goto slow; }
        if (self->kind == K_R1) CLN_CALL(rc, ((fn_r1)self->fn)(t[0].ptr, yp, n0, stream));
        else CLN_CALL(rc, ((fn_p3)self->fn)(t[0].ptr, t[1].ptr, yp, n0, stream));
        break;
      }
      default: goto slow;
    }
  }
  for (int i = 0; i < nt; ++i) Py_XDECREF(t[i].shape);
  nt = 0;
  if (rc == 0) {
    if (result) return result;
    Py_RETURN_NONE;
  }
  Py_CLEAR(result);
slow: /* the pure-Python wrapper repeats the checks, raises the reference's error texts, or handles what this path does not */
  for (int i = 0; i < nt; ++i) Py_XDECREF(t[i].shape);
  return PyObject_Vectorcall(self->slow, args, nargsf, kwnames);
}

static void fast_dealloc(FastFn* self) {
  Py_XDECREF(self->dtype);
  Py_XDECREF(self->out_dtype);
  Py_XDECREF(self->slow);
  Py_XDECREF(self->name);
  Py_TYPE(self)->tp_free((PyObject*)self);
}
static PyObject* fast_repr(FastFn* self) { return PyUnicode_FromFormat("<cln fast entry %U>", self->name); }
static PyObject* fast_get_name(FastFn* self, void* c) { (void)c; Py_INCREF(self->name); return self->name; }
static PyObject* fast_get_slow(FastFn* self, void* c) { (void)c; Py_INCREF(self->slow); return self->slow; }
static PyGetSetDef fast_getset[] = {{"__name__", (getter)fast_get_name, NULL, NULL, NULL}, {"__wrapped__", (getter)fast_get_slow, NULL, NULL, NULL}, {NULL}};

static PyTypeObject FastFnType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "_cln_fastcall.FastFn",
    .tp_basicsize = sizeof(FastFn),
    .tp_dealloc = (destructor)fast_dealloc,
    .tp_vectorcall_offset = offsetof(FastFn, vc),
    .tp_call = PyVectorcall_Call,
    .tp_repr = (reprfunc)fast_repr,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL,
    .tp_getset = fast_getset,
};

/* bind(addr, kind, dtype, slow, name, vt=0, out_dtype=None) */
static PyObject* mod_bind(PyObject* m, PyObject* a) {
  (void)m;
  unsigned long long addr;
  int kind, vt = 0;
  PyObject *dtype, *slow, *name, *out_dtype = Py_None;
  if (!PyArg_ParseTuple(a, "KiOOU|iO", &addr, &kind, &dtype, &slow, &name, &vt, &out_dtype)) return NULL;
  if (kind < K_P3 || kind >= K_RAW || !addr || !PyCallable_Check(slow)) {
    PyErr_SetString(PyExc_ValueError, "bind: bad kind / address / fallback");
    return NULL;
  }
  if (!g_get_device || !g_get_stream) {
    PyErr_SetString(PyExc_RuntimeError, "call setup(get_device, get_raw_stream) first");
    return NULL;
  }
  FastFn* f = PyObject_New(FastFn, &FastFnType);
  if (!f) return NULL;
  f->vc = fast_call;
  f->fn = (void*)(uintptr_t)addr;
  f->kind = kind;
  f->vt = vt;
  Py_INCREF(dtype); f->dtype = dtype;
  Py_INCREF(out_dtype); f->out_dtype = out_dtype;
  Py_INCREF(slow); f->slow = slow;
  Py_INCREF(name); f->name = name;
  return (PyObject*)f;
}
/* setup(torch._C._cuda_getDevice, torch._C._cuda_getCurrentRawStream) */
static PyObject* mod_setup(PyObject* m, PyObject* a) {
  (void)m;
  PyObject *gd, *gs;
  if (!PyArg_ParseTuple(a, "OO", &gd, &gs)) return NULL;
  if (!PyCallable_Check(gd) || !PyCallable_Check(gs)) {
    PyErr_SetString(PyExc_TypeError, "setup: two callables");
    return NULL;
  }
  Py_INCREF(gd); Py_XSETREF(g_get_device, gd);
  Py_INCREF(gs); Py_XSETREF(g_get_stream, gs);
  Py_RETURN_NONE;
}
static PyMethodDef methods[] = {{"bind", mod_bind, METH_VARARGS, "bind(addr, kind, dtype, slow, name, vt=0, out_dtype=None) -> fast entry"},
                                {"setup", mod_setup, METH_VARARGS, "setup(get_device, get_raw_stream)"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_cln_fastcall", "vectorcall entries in front of the cln_amd C-ABI", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__cln_fastcall(void) {
  if (PyType_Ready(&FastFnType) < 0) return NULL;
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return NULL;
  s_dtype = PyUnicode_InternFromString("dtype");
  s_data_ptr = PyUnicode_InternFromString("data_ptr");
  s_get_device = PyUnicode_InternFromString("get_device");
  s_is_contiguous = PyUnicode_InternFromString("is_contiguous");
  s_shape = PyUnicode_InternFromString("shape");
  s_numel = PyUnicode_InternFromString("numel");
  s_new_empty = PyUnicode_InternFromString("new_empty");
  s_dtype_kw = PyUnicode_InternFromString("dtype");
  g_one_tuple = Py_BuildValue("((i))", 1);
  if (g_one_tuple) { /* ((1,),) */
    PyObject* inner = Py_BuildValue("(i)", 1);
    Py_DECREF(g_one_tuple);
    g_one_tuple = inner ? PyTuple_Pack(1, inner) : NULL;
    Py_XDECREF(inner);
  }
  if (!s_dtype || !s_data_ptr || !s_get_device || !s_is_contiguous || !s_shape || !s_numel || !s_new_empty || !g_one_tuple) return NULL;
  static const struct { const char* n; int v; } kinds[] = {{"P3", K_P3}, {"UN", K_UN}, {"XY", K_XY}, {"LN", K_LN}, {"RN", K_RN}, {"G3", K_G3},
                                                            {"G6", K_G6}, {"FA", K_FA}, {"R1", K_R1}, {"D2", K_D2}};
  PyObject* d = PyDict_New();
  for (size_t i = 0; i < sizeof(kinds) / sizeof(kinds[0]); ++i) {
    PyObject* v = PyLong_FromLong(kinds[i].v);
    PyDict_SetItemString(d, kinds[i].n, v);
    Py_DECREF(v);
  }
  PyModule_AddObject(m, "KINDS", d);
  Py_INCREF(&FastFnType);
  PyModule_AddObject(m, "FastFn", (PyObject*)&FastFnType);
  return m;
}
