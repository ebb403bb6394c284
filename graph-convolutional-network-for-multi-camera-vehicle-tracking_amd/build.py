"""Build csrc/libmtmc_mpn.so for gfx950 with hipcc (in-tree, so it travels with the source snapshot).

    python -m mtmc_mpn.build [--force]
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(force: bool = False, verbose: bool = False) -> str:
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    jobs = str(min(4, os.cpu_count() or 1))
    r = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("mtmc_mpn: hipcc build of csrc/ failed")
    if verbose:
        print(r.stdout)
    return os.path.join(CSRC, "libmtmc_mpn.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
