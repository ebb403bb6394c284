#!/usr/bin/env python3
"""ISA lint of the built HIP library: reject instruction forms that miscompute on gfx950.

Rule PK-OPSEL (found in round 2, reproducer tools/hazard/pk_probe.hip): in a wave that has MFMAs in
flight, with a second wave resident on the SIMD, a packed-fp32 VALU op whose LOW result lane selects the HIGH dword of
a source (`v_pk_{mul,fma,add}_f32 ... op_sel:[..1..]`) returns 0 in that lane for lanes 48-63, sporadically.  hipcc's
SLP vectoriser creates the form on its own (e.g. two products with one scale that sits in the odd register of a pair),
so the source cannot rule it out: the build does.  Any kernel that contains an MFMA must not contain the form.

    python tools/check_isa.py [path/to/libmtmc_mpn.so]       exit status 1 and a listing if the rule is violated
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\bv_pk_(mul|fma|add|min|max)_f32\b")
OPSEL_LO = re.compile(r"\bop_sel:\[([01,]+)\]")


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
            name, body = None, []
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if name:
                        yield name, body
                    name, body = m.group(1), []
                elif name and line.startswith((" ", "\t")):
                    body.append(line.split("//")[0].strip())
            if name:
                yield name, body


def violations(lib):
    out = []
    for name, body in kernels_of(lib):
        has_mfma = any("v_mfma" in i or "v_smfma" in i for i in body)
        for i in body:
            if PK.search(i):
                m = OPSEL_LO.search(i)
                if m and "1" in m.group(1) and has_mfma:
                    out.append((name, i))
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, "..", "graph-convolutional-network-for-multi-camera-vehicle-tracking_amd", "csrc", "libmtmc_mpn.so")
    bad = violations(lib)
    for name, ins in bad:
        print(f"PK-OPSEL  {name}:  {ins}")
    n_k = sum(1 for _ in kernels_of(lib))
    print(f"check_isa: {n_k} device functions, {len(bad)} violation(s) of PK-OPSEL")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
