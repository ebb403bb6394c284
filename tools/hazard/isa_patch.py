#!/usr/bin/env python3
"""Insert wait states into the device ISA of one kernel (root-cause bisection of the f16x3 RNE store failure).

    isa_patch.py in.s out.s KERNEL_SUBSTR MODE

MODE: after:<regex>[:N|:text]   -- `s_nop N` (default 1) or the literal instruction text after every instruction of the kernel matching <regex>
      before:<regex>[:N]  -- ... before ...
"""
import re
import sys

src, dst, kern, mode = sys.argv[1:5]
if mode.startswith("sub@@"):            # sub@@<regex>@@<replacement (re.sub syntax, \n allowed)>
    _, rx, repl = mode.split("@@")
    rx = re.compile(rx)
    out, inside, hits = [], False, 0
    for line in open(src):
        if not inside and line.startswith("_ZN") and kern in line.split(":")[0]:
            inside = True
        elif inside and line.startswith(".Lfunc_end"):
            inside = False
        if inside and rx.search(line):
            line = rx.sub(repl.replace("\\n", "\n"), line)
            hits += 1
        out.append(line)
    open(dst, "w").writelines(out)
    print(f"{dst}: {hits} sites substituted", file=sys.stderr)
    sys.exit(0)
kind, rx, *rest = mode.split(":")
pad = "s_nop 1"
if rest:
    pad = f"s_nop {rest[0]}" if rest[0].isdigit() else rest[0].replace("_", " ", 1) if False else rest[0]
rx = re.compile(rx)
out, inside, hits = [], False, 0
for line in open(src):
    if not inside and line.startswith("_ZN") and kern in line.split(":")[0] and line.rstrip().split(";")[0].rstrip().endswith(":"):
        inside = True
    elif inside and line.startswith(".Lfunc_end"):
        inside = False
    body = line.split(";")[0].strip()
    hit = inside and body and not body.endswith(":") and not body.startswith(".") and rx.search(body)
    if hit and kind == "before":
        out.append(f"\t{pad}\n")
    out.append(line)
    if hit and kind == "after":
        out.append(f"\t{pad}\n")
    hits += bool(hit)
open(dst, "w").writelines(out)
print(f"{dst}: {hits} sites patched ({mode})", file=sys.stderr)
