import os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for env in ({}, {"MTMC_PASSC_VALU": "1"}):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "phase_times.py"), "cfg4"], env=e, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if "pass_c" in l][0]
        import json; d = json.loads(line.strip())
        print(env or "mfma", {k: d[k] for k in ("pass_a_kernel", "pass_b_kernel", "pass_c_kernel")})
