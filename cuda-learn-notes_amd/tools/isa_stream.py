"""Instruction-stream summary of one kernel in a hipcc .s file: per basic block, the mnemonic sequence with the
count of non-MFMA issues between consecutive MFMAs (the in-order wave hides ~5 behind each 32-cycle MFMA).

  python isa_stream.py /tmp/asm/flash_attn.s 'fa2_fwd_rb_kernelILi64' [--full]
"""
import re
import sys
from collections import Counter


def kernel_body(text, sub):
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(sub), text, re.M)
    if not m:
        raise SystemExit("kernel not found: " + sub)
    start = m.end()
    end = text.index(".end_amdhsa_kernel", start) if ".end_amdhsa_kernel" in text[start:] else len(text)
    end2 = text.find("s_endpgm", start)
    return m.group(1), text[start:end2 if end2 > 0 else end]


def main():
    text = open(sys.argv[1]).read()
    name, body = kernel_body(text, sys.argv[2])
    full = "--full" in sys.argv
    print(name)
    blocks, cur, label = [], [], "entry"
    for ln in body.split("\n"):
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                blocks.append((label, cur))
                label, cur = s.rstrip(":"), []
            continue
        if re.match(r"^\.?LBB\d+_\d+:", s):
            blocks.append((label, cur))
            label, cur = s.rstrip(":"), []
            continue
        cur.append(s.split()[0] if not full else s.split(";")[0].strip())
    blocks.append((label, cur))
    for label, ins in blocks:
        if not ins:
            continue
        nm = sum(1 for i in ins if i.startswith("v_mfma"))
        c = Counter(i.split("_e")[0] if False else i for i in (x.split()[0] for x in ins))
        top = ", ".join("%s x%d" % kv for kv in c.most_common(14))
        print("== %s: %d instr, %d mfma | %s" % (label, len(ins), nm, top))
        if nm >= 8:
            gaps, g, kinds = [], 0, Counter()
            seq = []
            for x in ins:
                op = x.split()[0]
                if op.startswith("v_mfma"):
                    gaps.append(g)
                    seq.append("M")
                    g = 0
                else:
                    g += 1
                    k = ("T" if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")) else
                         "L" if op.startswith("ds_") else "G" if op.startswith(("global_", "buffer_")) else
                         "w" if op.startswith("s_waitcnt") else "b" if op.startswith("s_barrier") else
                         "n" if op.startswith("s_nop") else "s" if op.startswith("s_") else "v")
                    seq.append(k)
            print("   gaps before each mfma:", gaps)
            line = "".join(seq)
            for i in range(0, len(line), 150):
                print("   " + line[i:i + 150])


if __name__ == "__main__":
    main()
