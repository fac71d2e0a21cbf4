"""Per-kernel register / spill / LDS report of a HIP source, from the gfx950 assembly hipcc emits (no GPU needed).

  python kernel_resources.py csrc/flash_attn.hip [substring ...]     # table: vgpr agpr sgpr spill scratch lds

Used by tests/test_no_spills.py (every kernel a C-ABI name can dispatch must have .vgpr_spill_count 0 and no
scratch) and by hand while shaping a kernel (the `.s` stays in --keep <dir> for reading the instruction stream).
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
sys.path.insert(0, PKG)


def compile_asm(src, out_s):
    import _build
    cmd = [_build.hipcc()] + _build.CFLAGS + ["--cuda-device-only", "-S", src, "-o", out_s]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr[-3000:]))
    return out_s


def demangle(names):
    import shutil
    filt = shutil.which("c++filt")
    if not filt:
        return {n: n for n in names}
    # binutils' c++filt does not know the _Float16 mangling (DF16_): present it as `half` (Dh) for the listing
    r = subprocess.run([filt], input="\n".join(n.replace("DF16_", "Dh") for n in names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.strip().split("\n")))


def parse(asm_text):
    """-> list of dicts (name, demangled, vgpr, agpr, sgpr, spill, scratch, lds) from the amdhsa.kernels metadata."""
    md = asm_text[asm_text.rfind("amdhsa.kernels"):]
    out = []
    for blk in re.split(r"\n  - \.agpr_count", md)[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
        out.append({"name": re.search(r"\.name:\s+(\S+)", blk).group(1), "agpr": int(re.match(r":\s+(\d+)", blk).group(1)),
                    "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "spill": g("vgpr_spill_count"),
                    "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"),
                    "lds": g("group_segment_fixed_size")})
    dm = demangle([k["name"] for k in out])
    for k in out:
        k["demangled"] = dm[k["name"]]
    return out


def report(src, keep=None):
    d = keep or tempfile.mkdtemp(prefix="cln_asm_")
    os.makedirs(d, exist_ok=True)
    s = os.path.join(d, os.path.basename(src).replace(".hip", ".s"))
    compile_asm(src, s)
    return parse(open(s).read()), s


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    keep = None
    if "--keep" in sys.argv:
        keep = sys.argv[sys.argv.index("--keep") + 1]
        args.remove(keep)
    src, subs = args[0], args[1:]
    ks, s = report(src, keep)
    for k in ks:
        if subs and not any(x in k["demangled"] for x in subs):
            continue
        print("%-120s vgpr=%3d agpr=%3d sgpr=%3d spill=%d scratch=%d" % (
            k["demangled"].replace("void fa2::", "").replace("void hgemm::", "")[:120], k["vgpr"], k["agpr"], k["sgpr"],
            k["spill"], k["scratch"]))
    print("asm:", s)


if __name__ == "__main__":
    main()
