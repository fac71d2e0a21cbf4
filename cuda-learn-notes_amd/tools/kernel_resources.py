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
    rel = os.path.relpath(os.path.abspath(src), _build.CSRC).replace(os.sep, "/")  # EXTRA_FLAGS is keyed by the path under csrc/ ("probe/x.hip")
    cmd = [_build.hipcc()] + _build.CFLAGS + _build.EXTRA_FLAGS.get(rel, _build.EXTRA_FLAGS.get(os.path.basename(src), [])) + ["--cuda-device-only", "-S", src, "-o", out_s]
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


def _asm_cached(src):
    """The assembly of `src` under build/asm_cache/, keyed by the digest of the source, of every header under csrc/ and of the flags (the key
    _build.py stamps its objects with): several tests read the same 100-second compile of hgemm.hip -- one compile per tree state, not one per test."""
    import hashlib
    import _build
    with open(src, "rb") as f:
        digest = hashlib.sha256(f.read() + _build._deps_digest().encode() + b"-S").hexdigest()[:24]
    cache = os.path.join(_build.BUILD, "asm_cache")
    os.makedirs(cache, exist_ok=True)
    out = os.path.join(cache, "%s.%s.s" % (os.path.basename(src).replace(".hip", ""), digest))
    if not os.path.exists(out):
        for old in os.listdir(cache):  # one entry per source: a stale digest of the same file goes
            if old.startswith(os.path.basename(src).replace(".hip", "") + "."):
                os.remove(os.path.join(cache, old))
        tmp = out + ".tmp%d" % os.getpid()
        compile_asm(src, tmp)
        os.replace(tmp, out)
    return out


def report(src, keep=None):
    """(kernels, path of the .s). `keep` = directory that receives a copy of the assembly for reading; the compile itself is cached."""
    cached = _asm_cached(os.path.abspath(src))
    s = cached
    if keep:
        import shutil
        os.makedirs(keep, exist_ok=True)
        s = os.path.join(keep, os.path.basename(src).replace(".hip", ".s"))
        shutil.copyfile(cached, s)
    return parse(open(s).read()), s


def _vregs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def asm_mfma_stream_check(asm_text, mangled_name):
    """Hand-checks for a kernel whose MFMAs are INLINE ASM (csrc/hgemm_w4.cuh): hipcc's hazard pass cannot see them, so
    the wait states it would insert around them have to hold by construction. Returns a list of problems (empty = ok):
      * no v_accvgpr_* and no VALU write of an MFMA operand between the first and the last MFMA (VALU write -> MFMA
        read needs wait states; the accumulators must already sit in their AGPRs);
      * an `s_nop 7` between the last accumulator zero-fill and the first MFMA, an `s_nop 15` between the last MFMA and
        the first v_accvgpr_read of the epilogue;
      * M0 (the LDS-DMA destination walks it across asm statements) is never saved / restored inside the stream, i.e.
        the compiler found no use of its own for it there."""
    m = re.search(r"^%s:" % re.escape(mangled_name), asm_text, re.M)
    if not m:
        return ["kernel not found: " + mangled_name]
    body = asm_text[m.end():asm_text.index("s_endpgm", m.end())]
    ins = [ln.split(";")[0].strip() for ln in body.split("\n")]
    ins = [i for i in ins if i and not i.startswith(".") and not i.endswith(":")]
    mf = [k for k, i in enumerate(ins) if i.startswith("v_mfma")]
    if not mf:
        return ["no MFMA in " + mangled_name]
    first, last = mf[0], mf[-1]
    problems = []
    for k in range(first, last + 1):
        i = ins[k]
        if i.startswith("v_accvgpr"):
            problems.append("accvgpr traffic inside the MFMA stream: " + i)
        elif i.startswith("v_") and not i.startswith("v_mfma"):
            dst = _vregs(i.split(None, 1)[1].split(",")[0])
            for j in range(k + 1, min(k + 6, last + 1)):
                if ins[j].startswith("v_mfma"):
                    ops = set()
                    for t in ins[j].split(None, 1)[1].split(", "):
                        ops |= _vregs(t)
                    if dst & ops:
                        problems.append("VALU write %r feeds an MFMA %d instructions later" % (i, j - k))
        if re.match(r"s_mov_b32 s\d+, m0", i):
            problems.append("M0 saved inside the MFMA stream (somebody else uses it): " + i)
    pre = ins[:first]
    lastw = max([k for k, i in enumerate(pre) if i.startswith("v_accvgpr")] or [-1])
    if not any(i.startswith("s_nop 7") for i in pre[lastw + 1:]):
        problems.append("no s_nop 7 between the accumulator zero-fill and the first MFMA")
    post = ins[last + 1:]
    firstr = min([k for k, i in enumerate(post) if i.startswith("v_accvgpr_read")] or [len(post)])
    if not any(i.startswith("s_nop 15") for i in post[:firstr]):
        problems.append("no s_nop 15 between the last MFMA and the first v_accvgpr_read")
    return problems


def asm_mfma_operand_hazards(asm_text, mangled_name):
    """For a kernel whose MFMAs are inline asm (csrc/flash_attn_dw4.cuh, flash_attn_dw4b.cuh): a VGPR written by a VALU instruction needs TWO wait states
    before an MFMA reads it as A / B operand, and hipcc's hazard pass does not look into inline asm. Walks every basic block: for each v_mfma, the
    instructions of the two wait states before it (an `s_nop N` counts N + 1 states, every other instruction one) must not be VALU writes of its A / B
    registers. Returns the offending (VALU instruction, MFMA) pairs."""
    m = re.search(r"^%s:" % re.escape(mangled_name), asm_text, re.M)
    if not m:
        return ["kernel not found: " + mangled_name]
    body = asm_text[m.end():asm_text.index("s_endpgm", m.end())]
    ins = []
    for ln in body.split("\n"):
        i = ln.split(";")[0].strip()
        if not i or i.startswith("."):
            continue
        ins.append(i)
    problems = []
    for k, i in enumerate(ins):
        if not i.startswith("v_mfma"):
            continue
        ops = i.split(None, 1)[1].split(", ")
        ab = _vregs(ops[1]) | _vregs(ops[2])
        states, j = 0, k - 1
        while j >= 0 and states < 2:
            p = ins[j]
            if p.endswith(":"):
                break  # block boundary: predecessors not followed (the pads of the kernels sit inside the block of their MFMAs)
            if p.startswith("s_nop"):
                states += int(p.split()[1]) + 1
            else:
                if p.startswith("v_") and not p.startswith("v_mfma") and not p.startswith("v_accvgpr"):
                    if _vregs(p.split(None, 1)[1].split(",")[0]) & ab:
                        problems.append((p, i))
                states += 1
            j -= 1
    return problems


def asm_inflight_load_hazards(asm_text, mangled_name, load_re=r"global_load_dwordx4 .* sc1"):
    """Loads issued from INLINE ASM with an "=v" output (csrc/hgemm_w4.cuh, the split-K fix-up's agent-scope loads): the compiler regards the output as
    defined right behind the asm statement although the data lands only at the later `s_waitcnt vmcnt(0)` (a separate asm statement) -- a copy or a
    spill of the destination placed between the two would read registers the load has not filled yet. Walks every kernel body: from each
    matching load to the next `s_waitcnt` whose vmcnt is 0, no instruction other than another load may name one of its destination registers.
    Returns the offending (load, instruction) pairs."""
    m = re.search(r"^%s:" % re.escape(mangled_name), asm_text, re.M)
    if not m:
        return ["kernel not found: " + mangled_name]
    body = asm_text[m.end():asm_text.index("s_endpgm", m.end())]
    ins = [ln.split(";")[0].strip() for ln in body.split("\n")]
    ins = [i for i in ins if i and not i.startswith(".")]
    problems, seen = [], 0
    for k, i in enumerate(ins):
        if not re.match(load_re, i):
            continue
        seen += 1
        dst = _vregs(i.split(None, 1)[1].split(",")[0])
        for j in range(k + 1, len(ins)):
            p = ins[j]
            if p.startswith("s_waitcnt") and ("vmcnt(0)" in p or re.fullmatch(r"s_waitcnt 0x?0*", p) or p.strip() == "s_waitcnt lgkmcnt(0) vmcnt(0)"):
                break
            if p.endswith(":") or p.startswith("s_cbranch") or p.startswith("s_branch"):
                problems.append((i, "control flow before the wait: " + p))
                break
            if re.match(load_re, p):
                continue
            toks = p.split(None, 1)
            regs = set()
            for t in (toks[1].split(", ") if len(toks) > 1 else []):
                regs |= _vregs(t.split(" ")[0])
            if regs & dst:
                problems.append((i, p))
    return problems if seen else ["no load matching %r in %s" % (load_re, mangled_name)]


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
