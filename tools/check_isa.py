#!/usr/bin/env python3
"""ISA lint of the built HIP libraries: reject instruction forms that miscompute on gfx950, and check the hand-scheduled
inline-asm sequences whose correctness rests on things the compiler cannot see.

Rule PK-OPSEL (round 2, reproducer tools/hazard/pk_probe.hip): in a wave that has MFMAs in flight, with a
second wave resident on the SIMD, a packed-fp32 VALU op whose LOW result lane selects the HIGH dword of a source
(`v_pk_{mul,fma,add}_f32 ... op_sel:[..1..]`) returns 0 in that lane for lanes 48-63, sporadically.  hipcc's SLP vectoriser
creates the form on its own, so the source cannot rule it out: the build does.  Any kernel that contains an MFMA must not
contain the form.

Rule LDS-DMA-M0 (csrc/lds_dma.h): `global_load_lds_dwordx4` takes its LDS address from m0, which the inline asm sets itself
with `s_mov_b32 m0, ...` -- a register the compiler believes reserved for its own use and does not know the asm writes.
Every such instruction must be preceded by its OWN `s_mov_b32 m0`, with nothing but `s_nop` in between (nothing that reads or
writes m0 can have been scheduled, or emitted by the compiler, between the two).

Rule COUNTED-WAIT (for any kernel that marks such a sequence; round 2's pass_c_mfma_kernel did, round 3 rewrote it with plain
loads): two `global_load_dword` issued from inline asm are waited for with a COUNTED `s_waitcnt vmcnt(N)` that leaves the N
younger prefetch loads in flight.  That is only right if, between
the pair and the wait, EXACTLY N vector-memory loads were issued, no vector-memory store or atomic (they share the counter),
and no instruction touches the pair's destination VGPRs (a register copy made before the wait would be stale).  The asm marks
the two places with `s_mov_b32 sX, 0xc0de0001` (before the pair) and `s_mov_b32 sX, 0xc0de0002` (before the wait); the rule
checks every marked region of every kernel.

    python tools/check_isa.py [path/to/lib.so]       exit status 1 and a listing if a rule is violated
The objdump used: $MTMC_OBJDUMP, else llvm-objdump next to $HIPCC / under $ROCM_PATH / /opt/rocm, else the one on PATH.
MTMC_SKIP_ISA_LINT=1 makes the build proceed without the lint (a box without llvm-objdump); tests/test_isa_lint.py then skips.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile


def find_objdump():
    cands = [os.environ.get("MTMC_OBJDUMP")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"))
    cands.append(shutil.which("llvm-objdump"))
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


OBJDUMP = find_objdump()
PK = re.compile(r"\bv_pk_(mul|fma|add|min|max)_f32\b")
OPSEL_LO = re.compile(r"\bop_sel:\[([01,]+)\]")
MARK_PAIR, MARK_WAIT = "0xc0de0001", "0xc0de0002"
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
VMEM_LOAD = re.compile(r"^(global|buffer|flat|scratch)_load")


def kernels_of(lib):
    """yield (kernel name, [instruction text]) for every function of every gfx950 code object bundled in lib"""
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(os.path.abspath(lib), os.path.join(tmp, "lib.so"))      # bundles are extracted next to the input
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            yield from functions_of(dis)


class Ins(str):
    """instruction text with its address and, for a branch, the address it jumps to (None where the listing has none)"""
    addr = None
    target = None


def functions_of(dis):
    """(name, [Ins]) per function of one disassembly listing"""
    name, start, body = None, 0, []
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            if name:
                yield name, body
            name, start, body = m.group(2), int(m.group(1), 16), []
        elif name and line.startswith((" ", "\t")):
            text, _, comment = line.partition("//")
            ins = Ins(text.strip())
            if ins:
                ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
                ins.addr = int(ma.group(1), 16) if ma else None
                mt = re.search(r"<[^>]*\+0x([0-9A-Fa-f]+)>\s*$", comment)
                if mt and re.match(r"^s_(cbranch|branch)", ins):
                    ins.target = start + int(mt.group(1), 16)
                body.append(ins)
    if name:
        yield name, body


def vgprs_of(ins):
    """VGPR numbers an instruction names (v7, v[4:7])"""
    regs = set()
    for m in re.finditer(r"\bv(\d+)\b", ins):
        regs.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", ins):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def check_pk_opsel(name, body):
    if not any("v_mfma" in i or "v_smfma" in i for i in body):
        return []
    out = []
    for i in body:
        if PK.search(i):
            m = OPSEL_LO.search(i)
            if m and "1" in m.group(1):
                out.append(("PK-OPSEL", name, i))
    return out


def check_lds_dma_m0(name, body):
    out = []
    for k, ins in enumerate(body):
        if not ins.startswith("global_load_lds_"):
            continue
        j = k - 1
        while j >= 0 and body[j].startswith("s_nop"):
            j -= 1
        if j < 0 or not re.match(r"^s_mov_b32 m0\b", body[j]):
            out.append(("LDS-DMA-M0", name, f"{ins}   <- preceded by `{body[j] if j >= 0 else '(start)'}`, not by its own s_mov_b32 m0"))
    return out


BRANCH = re.compile(r"^s_(cbranch\w*|branch)\b")
LEAVES = re.compile(r"^s_(endpgm|setpc|swappc|trap|call)")


def check_counted_wait(name, body):
    """Every region [pair marker .. wait marker]: the two asm loads right behind the first marker; then EXACTLY N
    vector-memory loads, all of them ahead of the region's first branch (so they are issued on every path), no other
    vector-memory instruction anywhere in the region, nothing that names the pair's destination VGPRs, and control flow that
    neither leaves the region nor enters it from outside; `s_waitcnt vmcnt(N)` right behind the second marker."""
    out = []

    def bad(what):
        out.append(("COUNTED-WAIT", name, what))
    is_mark = lambda ins, lit: ins.startswith("s_mov_b32") and lit in ins
    k = 0
    while k < len(body):
        if not is_mark(body[k], MARK_PAIR):
            if is_mark(body[k], MARK_WAIT):
                bad("wait marker without a pair marker before it")
            k += 1
            continue
        pair = body[k + 1:k + 3]
        if len(pair) < 2 or not all(p.startswith("global_load_dword ") for p in pair):
            bad(f"pair marker not followed by the two global_load_dword: {list(pair)}")
            k += 1
            continue
        dst = set()
        for p in pair:
            dst |= vgprs_of(p.split(",")[0])
        j = k + 3
        while j < len(body) and not is_mark(body[j], MARK_WAIT) and not is_mark(body[j], MARK_PAIR):
            j += 1
        if j >= len(body) or not is_mark(body[j], MARK_WAIT):
            bad("pair marker without a wait marker behind it")
            k += 1
            continue
        region = body[k + 3:j]
        loads, seen_branch = 0, False
        for ins in region:
            if BRANCH.match(ins):
                seen_branch = True
            if LEAVES.match(ins):
                bad(f"the region can be left before its wait: {ins}")
            if VMEM.match(ins):
                if not VMEM_LOAD.match(ins):
                    bad(f"vector-memory store/atomic between the pair and its wait: {ins}")
                elif seen_branch:
                    bad(f"a vector-memory load behind a branch of the region (not issued on every path): {ins}")
                else:
                    loads += 1
            if vgprs_of(ins) & dst:
                bad(f"the pair's destination registers {sorted(dst)} are touched before the wait: {ins}")
        # control flow: closed region (addresses known only for real listings; patched test listings carry none)
        lo, hi = getattr(body[k], "addr", None), getattr(body[j], "addr", None)
        for idx, ins in enumerate(body):
            tgt = getattr(ins, "target", None)
            if not BRANCH.match(ins):
                continue
            inside = k < idx < j
            if tgt is None or lo is None or hi is None:
                if inside and getattr(ins, "addr", None) is None:
                    bad(f"control flow between the pair and its wait, target unknown: {ins}")
                continue
            if inside and not (lo <= tgt <= hi):
                bad(f"a branch leaves the region before its wait: {ins} -> {tgt:#x}")
            if not inside and lo < tgt <= hi:
                bad(f"a branch from outside enters the region behind its loads: {ins} -> {tgt:#x}")
        wait = body[j + 1] if j + 1 < len(body) else ""
        m = re.match(r"^s_waitcnt vmcnt\((\d+)\)$", wait)
        if not m:
            bad(f"wait marker not followed by a bare `s_waitcnt vmcnt(N)`: {wait}")
        elif int(m.group(1)) != loads:
            bad(f"{wait} but {loads} vector-memory load(s) are issued on every path behind the pair")
        k = j + 1
    return out


RULES = (check_pk_opsel, check_lds_dma_m0, check_counted_wait)


def violations_in(functions):
    out = []
    for name, body in functions:
        for rule in RULES:
            out += rule(name, body)
    return out


def violations(lib):
    return violations_in(kernels_of(lib))


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, "..", "graph-convolutional-network-for-multi-camera-vehicle-tracking_amd", "csrc", "libmtmc_mpn.so")
    if os.environ.get("MTMC_SKIP_ISA_LINT"):
        print("check_isa: skipped (MTMC_SKIP_ISA_LINT is set)")
        return 0
    if OBJDUMP is None:
        print("check_isa: no llvm-objdump found (looked at $MTMC_OBJDUMP, next to hipcc, under $ROCM_PATH and /opt/rocm, on PATH).\n"
              "           Point MTMC_OBJDUMP at one, or set MTMC_SKIP_ISA_LINT=1 to build without the lint.", file=sys.stderr)
        return 1
    funcs = list(kernels_of(lib))
    bad = violations_in(funcs)
    for rule, name, what in bad:
        print(f"{rule}  {name}:  {what}")
    n_dma = sum(1 for _, b in funcs for i in b if i.startswith("global_load_lds_"))
    n_cw = sum(1 for _, b in funcs for i in b if MARK_PAIR in i)
    print(f"check_isa: {len(funcs)} device functions, {n_dma} LDS-DMA instructions, {n_cw} counted-wait region(s), "
          f"{len(bad)} violation(s) of PK-OPSEL / LDS-DMA-M0 / COUNTED-WAIT")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
