"""Instruction budget of a kernel's hot loop by PURPOSE, from hipcc's own assembly of the product compile unit.
Every instruction of the loop blocks (the basic blocks that hold MFMAs and are inside a loop) is put in one class:

  mfma | exp (v_exp_f32) | row-sum add (v_add_f32) | cvt (v_cvt_pk_f16_f32) | scale fma (v_fma_f32 / v_fmac) | max / compare (overflow check of the
  deferred blocks) | other VALU (address arithmetic, selects) | LDS read | LDS-DMA | SALU | waitcnt / nop / barrier / setprio / branch

    python isa_itemise.py <file.hip under csrc/> <kernel-name substring> [more substrings ...]
"""
import os
import re
import sys
from collections import Counter, OrderedDict

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import kernel_resources as kr  # noqa: E402

CLASSES = OrderedDict([
    ("mfma", lambda o: o.startswith("v_mfma")),
    ("exp", lambda o: o.startswith("v_exp")),
    ("row-sum add", lambda o: o.startswith("v_add_f32")),
    ("cvt f32->f16 pair", lambda o: o.startswith("v_cvt_pk_f16")),
    ("scale fma", lambda o: o.startswith(("v_fma_f32", "v_fmac_f32", "v_pk_fma", "v_mul_f32", "v_pk_mul"))),
    ("max / compare", lambda o: o.startswith(("v_max", "v_cmp", "v_min"))),
    ("other VALU", lambda o: o.startswith("v_")),
    ("LDS read", lambda o: o.startswith("ds_")),
    ("LDS-DMA / global", lambda o: o.startswith(("global_", "buffer_"))),
    ("waitcnt", lambda o: o.startswith("s_waitcnt")),
    ("nop", lambda o: o.startswith("s_nop")),
    ("barrier / setprio / branch", lambda o: o.startswith(("s_barrier", "s_setprio", "s_cbranch", "s_branch"))),
    ("SALU", lambda o: o.startswith("s_")),
])


def classify(op):
    for name, pred in CLASSES.items():
        if pred(op):
            return name
    return "?"


def loop_blocks(asm, mangled):
    m = re.search(r"^%s:" % re.escape(mangled), asm, re.M)
    body = asm[m.end():asm.index("s_endpgm", m.end())]
    blocks, cur, in_loop = OrderedDict(), "entry", False
    for ln in body.split("\n"):
        s = ln.strip()
        mm = re.match(r"^(\.LBB\d+_\d+):(.*)", s)
        if mm:
            cur, in_loop = mm.group(1), "Loop" in mm.group(2)
            blocks[cur] = {"loop": in_loop, "ins": []}
            continue
        if not s or s.startswith((";", ".")) or cur not in blocks:
            continue
        blocks[cur]["ins"].append(s.split()[0])
    return OrderedDict((k, v["ins"]) for k, v in blocks.items() if v["loop"] and any(i.startswith("v_mfma") for i in v["ins"]))


def main():
    src = sys.argv[1] if os.path.isabs(sys.argv[1]) else os.path.join(kr.PKG, "csrc", sys.argv[1])
    out_s = "/tmp/itemise_%s.s" % os.path.basename(src)
    kr.compile_asm(src, out_s)
    asm = open(out_s).read()
    names = re.findall(r"^(_Z\w+):", asm, re.M)
    dem = kr.demangle(names)
    for sub in sys.argv[2:]:
        hits = [(n, dem[n]) for n in names if sub in dem[n] or sub in n]
        for mangled, d in hits:
            blocks = loop_blocks(asm, mangled)
            tot = Counter()
            print("\n## %s" % d.split("(")[0])
            print("| loop block | instr | " + " | ".join(CLASSES) + " |")
            print("|---|---|" + "---|" * len(CLASSES))
            for b, ins in blocks.items():
                c = Counter(classify(i) for i in ins)
                tot.update(c)
                print("| %s | %d | " % (b, len(ins)) + " | ".join(str(c.get(k, 0)) for k in CLASSES) + " |")
            n = sum(tot.values())
            print("| **per KV tile** | %d | " % n + " | ".join(str(tot.get(k, 0)) for k in CLASSES) + " |")
            valu = sum(tot[k] for k in ("exp", "row-sum add", "cvt f32->f16 pair", "scale fma", "max / compare", "other VALU"))
            soft = tot["exp"] + tot["row-sum add"] + tot["cvt f32->f16 pair"] + tot["scale fma"]
            print("\nnon-MFMA VALU %d per wave and KV tile, of which exp + row sum + convert (+ fp32 scale) = %d (%.0f %%); the rest: overflow check of "
                  "the deferred key blocks %d, address / select %d. Issue slots besides the %d MFMAs: %d (%.1f per MFMA)."
                  % (valu, soft, 100.0 * soft / max(valu, 1), tot["max / compare"], tot["other VALU"], tot["mfma"], n - tot["mfma"], (n - tot["mfma"]) / max(tot["mfma"], 1)))


if __name__ == "__main__":
    main()
