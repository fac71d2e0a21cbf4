"""Scan gfx950 assembly for MFMA instructions whose DESTINATION overlaps their A or B source registers (and is not the tied
accumulator). hipcc allows that for the 4-register-destination shapes (no early-clobber on v_mfma_f32_16x16x32_*); on MI355X
the K-doubled 16x16x32 shapes then return wrong results when another wave's MFMAs interleave on the same SIMD (round 3:
the D = 256 / 512 attention kernels with a constant-zero C operand; profiles/r03_fa_pair_mfma_overlap_bisect.log).
  python mfma_overlap_scan.py file.s [kernel-substring]       -> one line per offending instruction, 'total N'"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return (tok[0], set(range(int(m.group(1)), int(m.group(2)) + 1)))
    m = re.match(r"([va])(\d+)$", tok)
    return (m.group(1), {int(m.group(2))}) if m else (None, set())


def scan(text, sub=""):
    out = []
    for km in re.finditer(r"^(_Z\w+):.*?s_endpgm", text, re.M | re.S):
        name = km.group(1)
        if sub not in name:
            continue
        for l in km.group(0).split("\n"):
            l = l.split(";")[0].strip()
            if not l.startswith("v_mfma"):
                continue
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            d, a, b, c = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3].split()[0])
            for src in (a, b):
                if src[0] == d[0] and d[1] & src[1] and not (c[0] == d[0] and c[1] == d[1]):
                    out.append((name, l))
                    break
    return out


def scan_shared_object(so_path, workdir):
    """The same scan over the gfx950 code objects INSIDE a built shared library (llvm-objdump --offloading + -d: seconds, no
    recompilation): -> ([(mangled kernel name, instruction)], number of MFMA instructions seen)."""
    import glob
    import os
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump")
    if not objdump:
        raise RuntimeError("llvm-objdump not found")
    os.makedirs(workdir, exist_ok=True)
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)  # --offloading writes the extracted bundles next to its input
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=workdir)
    out, n_mfma = [], 0
    for co in sorted(glob.glob(local + ".*gfx950")):
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        # objdump: "<addr> <_Zname>:" headers, "\tinsn operands // addr: encoding" lines -> the hipcc -S shape scan() reads
        text = re.sub(r"^[0-9a-f]+ <(_Z\w+)>:", r"\1:", dis, flags=re.M)
        text = re.sub(r"//.*$", "", text, flags=re.M)
        n_mfma += text.count("v_mfma")
        out += scan(text)
    return out, n_mfma


def shared_object_spills(so_path, workdir):
    """Kernel metadata of the gfx950 code objects inside a built shared library (llvm-readelf --notes): -> (number of kernels,
    [(kernel name, vgpr spills, sgpr spills, scratch bytes)] for every kernel that spills or uses scratch)."""
    import glob
    import os
    import shutil
    import subprocess
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    os.makedirs(workdir, exist_ok=True)
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=workdir)
    n, bad = 0, []
    for co in sorted(glob.glob(local + ".*gfx950")):
        notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:] if "- .agpr_count" in notes else notes.split(".args:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vs, ss, sc = (re.search(r"\.%s:\s+(\d+)" % k, blk) for k in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"))
            if not (name and vs and ss and sc):
                continue
            n += 1
            if int(vs.group(1)) or int(ss.group(1)) or int(sc.group(1)):
                bad.append((name.group(1), int(vs.group(1)), int(ss.group(1)), int(sc.group(1))))
    return n, bad


if __name__ == "__main__":
    res = scan(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    from collections import Counter
    for (name, l), n in Counter(res).items():
        print("OVERLAP x%d | %s | %s" % (n, l, name[:70]))
    print("total", len(res))
