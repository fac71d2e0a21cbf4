"""Build the gfx950 shared objects in-tree with hipcc (no torch headers, no pybind, no hipify).

  csrc/*.hip  --hipcc -c-->  build/*.o  --hipcc -shared-->  lib/libcln_amd.so         (the product: reference names only)
  csrc/hgemm_vendor*.hip, fa2_vendor_ck.hip, yardstick_vendor.hip  ->  lib/libcln_amd_vendor.so  (comparison rows: rocBLAS, hipBLASLt, ck_tile FMHA;
                                                           bench yardsticks: hipMemcpyDtoD, rocprim::reduce)
  csrc/*_probe.hip  ------------------------------------>  lib/libcln_amd_probe.so   (TEST-ONLY: tuning hooks, ablation and
                                                           probe instantiations; nothing in the product path loads it)
  csrc/pyext/cln_fastcall.c  --gcc-->  lib/_cln_fastcall*.so  (CPython vectorcall entries in front of the C-ABI; optional)

Replaces the reference's JIT `torch.utils.cpp_extension.load(...)` at script import
(kernels/hgemm/tools/utils.py:104-113, kernels/elementwise/elementwise.py:10-22).
hipcc cross-compiles without a GPU, so this runs in the CPU-only container too.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
BUILD = os.path.join(PKG_DIR, "build")
LIBDIR = os.path.join(PKG_DIR, "lib")

ARCH = "gfx950"
KERNEL_SOURCES = [
    "elementwise.hip", "activation.hip", "blas1.hip", "indexing.hip", "reduce.hip", "softmax.hip", "norm.hip", "rope.hip",
    "sgemm.hip", "stream_scratch.hip", "hgemm.hip", "hgemm_ring_nn.hip", "hgemm_ring_tn.hip", "flash_attn.hip", "flash_attn_m16x.hip", "describe.hip",
]
VENDOR_SOURCES = ["hgemm_vendor.hip", "hgemm_vendor_lt.hip", "fa2_vendor_ck.hip", "yardstick_vendor.hip"]  # the last: ck_tile FMHA instances (~1 min of hipcc)
# a comparison row whose sources are the ROCm image's ck_tile headers: if they are missing or do not compile, the vendor
# library is linked without it (the callers treat the row as absent) instead of failing the whole build
OPTIONAL_SOURCES = {"fa2_vendor_ck.hip", "hgemm_vendor_lt.hip", "yardstick_vendor.hip"}  # (the hipBLASLt row too: an image without hipBLASLt still builds the product)
# test-only library; it re-links the two ring compile units for the explicit (tile, BK, stages) hook
# (csrc/probe/: the probe compile units and the kernels that only they instantiate -- nothing under it is linked into libcln_amd.so)
PROBE_SOURCES = ["probe/hgemm_probe.hip", "probe/flash_attn_probe.hip", "probe/flash_attn_m16x_probe.hip"]
PROBE_SHARED = ["hgemm_ring_nn.hip", "hgemm_ring_tn.hip"]
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=fast",
          "-I" + CSRC, "-I" + os.path.join(CSRC, "probe")]
# per-file additions (the reason is stated at the top of each file)
EXTRA_FLAGS = {"flash_attn_m16x.hip": ["-fno-slp-vectorize"], "probe/flash_attn_m16x_probe.hip": ["-fno-slp-vectorize"],
               "fa2_vendor_ck.hip": ["-I/opt/rocm/include", "-Wno-everything"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _deps_digest():
    """Digest of every header/include file: any change rebuilds all objects."""
    h = hashlib.sha256()
    for sub in ("", "probe"):
        for fn in sorted(os.listdir(os.path.join(CSRC, sub))):
            if fn.endswith((".h", ".cuh", ".inc")):
                with open(os.path.join(CSRC, sub, fn), "rb") as f:
                    h.update(fn.encode())
                    h.update(f.read())
    h.update(" ".join(CFLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _compile_one(src, hdr_digest, verbose):
    obj = os.path.join(BUILD, os.path.basename(src).replace(".hip", ".o"))
    stamp = obj + ".stamp"
    with open(os.path.join(CSRC, src), "rb") as f:
        digest = hashlib.sha256(f.read() + hdr_digest.encode()).hexdigest()
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj, False
    cmd = [hipcc()] + CFLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if src in OPTIONAL_SOURCES:
            print("warning: optional comparison row %s did not compile, building without it:\n%s" % (src, r.stderr[-600:]), file=sys.stderr)
            return None, False
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    with open(stamp, "w") as f:
        f.write(digest)
    return obj, True


def _link(objs, out, extra, verbose):
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", out] + extra
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed for %s:\n%s" % (out, r.stderr[-4000:]))


def build_pyext(verbose=False, force=False):
    """csrc/pyext/cln_fastcall.c --gcc--> lib/_cln_fastcall<EXT_SUFFIX>: the CPython entry in front of the C-ABI (vectorcall, no ctypes
    marshalling, no torch headers). Optional: without a C compiler / Python.h the host layer keeps calling through ctypes."""
    import sysconfig
    src = os.path.join(CSRC, "pyext", "cln_fastcall.c")
    out = os.path.join(LIBDIR, "_cln_fastcall" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    stamp = os.path.join(BUILD, "cln_fastcall.stamp")
    with open(src, "rb") as f:
        digest = hashlib.sha256(f.read() + out.encode()).hexdigest()
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == digest:
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    inc = sysconfig.get_paths().get("include")
    if not cc or not inc or not os.path.exists(os.path.join(inc, "Python.h")):
        print("warning: no C compiler / Python.h: building without the _cln_fastcall entry (host.py falls back to ctypes)", file=sys.stderr)
        return None
    cmd = [cc, "-O2", "-shared", "-fPIC", "-Wall", "-I" + inc, src, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("warning: _cln_fastcall did not compile, building without it (host.py falls back to ctypes):\n%s" % r.stderr[-800:], file=sys.stderr)
        return None
    with open(stamp, "w") as f:
        f.write(digest)
    return out


def build_harness(verbose=False):
    """harness/hgemm_bench (the C++ caller of the C-ABI; tools/round_evidence.sh runs it): rebuilt when older than its source or the
    libraries. Optional: a failure here never fails the library build."""
    hdir = os.path.join(PKG_DIR, "harness")
    src, out = os.path.join(hdir, "hgemm_bench.cpp"), os.path.join(hdir, "hgemm_bench")
    deps = [src, os.path.join(LIBDIR, "libcln_amd.so"), os.path.join(LIBDIR, "libcln_amd_vendor.so")]
    if not all(os.path.exists(d) for d in deps):
        return None
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    cmd = [hipcc(), "-O2", "-std=c++17", src, "-I" + os.path.join(os.path.dirname(PKG_DIR), "include"), "-L" + LIBDIR, "-lcln_amd", "-lcln_amd_vendor",
           "-Wl,-rpath,$ORIGIN/../lib", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("warning: harness/hgemm_bench did not build:\n%s" % r.stderr[-600:], file=sys.stderr)
        return None
    return out


def build(verbose=False, force=False):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for fn in os.listdir(BUILD):
            os.remove(os.path.join(BUILD, fn))
    hd = _deps_digest()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        allsrc = KERNEL_SOURCES + VENDOR_SOURCES + PROBE_SOURCES
        res = list(ex.map(lambda s: _compile_one(s, hd, verbose), allsrc))
    objs = dict(zip(allsrc, res))
    main_so = os.path.join(LIBDIR, "libcln_amd.so")
    vend_so = os.path.join(LIBDIR, "libcln_amd_vendor.so")
    probe_so = os.path.join(LIBDIR, "libcln_amd_probe.so")
    if force or not os.path.exists(main_so) or any(objs[s][1] for s in KERNEL_SOURCES):
        _link([objs[s][0] for s in KERNEL_SOURCES], main_so, [], verbose)
    if force or not os.path.exists(vend_so) or any(objs[s][1] for s in VENDOR_SOURCES):
        libs = ["-L/opt/rocm/lib", "-lrocblas"] + (["-lhipblaslt"] if objs["hgemm_vendor_lt.hip"][0] is not None else [])
        _link([objs[s][0] for s in VENDOR_SOURCES if objs[s][0] is not None], vend_so, libs, verbose)
    if force or not os.path.exists(probe_so) or any(objs[s][1] for s in PROBE_SOURCES + PROBE_SHARED):
        _link([objs[s][0] for s in PROBE_SOURCES + PROBE_SHARED], probe_so, [], verbose)
    build_pyext(verbose, force)
    build_harness(verbose)
    return main_so, vend_so, probe_so


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv, force="-f" in sys.argv))
