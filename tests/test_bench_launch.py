"""`python bench.py --gpus N` must never sit in a rendezvous nobody else joins: without
a launcher's environment it starts its own N ranks before anything touches a GPU, or refuses at once."""
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""          # no GPU for this check, wherever it runs
    return env


def test_more_gpus_than_visible_is_refused_at_once():
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    assert time.time() - t0 < 240                                              # (first `import torch` of a fresh box is slow)


def test_a_launchers_world_size_must_match():
    env = dict(_env(), RANK="0", WORLD_SIZE="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_self_launch_command_is_the_drivers():
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(gpus=4, steps=7, warmup=2, workload="auto", no_stress=False, no_cpu=True)
    cmd = bench.self_launch_command(args, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "auto", "--no-cpu"]
