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


if __name__ == "__main__":
    res = scan(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    from collections import Counter
    for (name, l), n in Counter(res).items():
        print("OVERLAP x%d | %s | %s" % (n, l, name[:70]))
    print("total", len(res))
