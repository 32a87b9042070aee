"""bench.py's launch logic without a GPU: `--gpus N` must never fall back to one silent rank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ESCAT_BENCH_REHEARSAL")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=120,
                          env=env, cwd=ROOT)


def test_gpus_n_without_enough_devices_exits_nonzero():
    import torch
    if torch.cuda.device_count() >= 2:
        return                                                   # (a multi-GPU box would really launch)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_world_size_mismatch_exits_nonzero():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
