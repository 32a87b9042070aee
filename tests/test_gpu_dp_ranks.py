"""
BASELINE cfg4 with more than one rank: two fresh child processes share cuda:0 over gloo (a one-GPU box; RCCL needs one
GPU per rank), run three DPTrainer steps on their halves of a batch, and must end with bit-identical parameters that
equal the single-rank run on the whole batch -- the all-reduce mean of the flat gradient bucket is what the reference
gets from DDP in acc.backward (training/train.py:408).
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = str(tmp_path / f"w{world}_r{r}.pt")
        outs.append(out)
        logf = open(str(tmp_path / f"w{world}_r{r}.log"), "w")     # a file, not a pipe: nobody reads while we poll
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out], env=env, cwd=ROOT,
                                      stdout=logf, stderr=subprocess.STDOUT, text=True))
        logf.close()
    logs = [""] * world
    try:
        # wait for every rank before judging any: a rank that died leaves its siblings inside a collective
        import time
        deadline = time.monotonic() + 240
        while any(p.poll() is None for p in procs):
            if time.monotonic() > deadline or any(p.poll() not in (None, 0) for p in procs):
                break
            time.sleep(0.1)
    finally:
        for p in procs:                                             # nothing of ours stays on the GPU past this point
            if p.poll() is None:
                p.kill()
        for i, p in enumerate(procs):
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
            with open(str(tmp_path / f"w{world}_r{i}.log")) as f:
                logs[i] = f.read()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} exit {p.returncode}\n" + "\n".join(l[-3000:] for l in logs)
    return [torch.load(o, weights_only=True) for o in outs]


def test_two_ranks_match_each_other_and_the_single_rank_run(tmp_path):
    two = _launch(2, tmp_path)
    one = _launch(1, tmp_path)[0]
    assert two[0]["world"] == 2 and two[1]["world"] == 2 and one["world"] == 1
    assert two[0]["graphs"] == (0, 48) and two[1]["graphs"] == (48, 96) and one["graphs"] == (0, 96)
    for a, b in zip(two[0]["params"], two[1]["params"]):
        assert torch.equal(a, b)                                   # same averaged gradient, same update, bit for bit
    for a, c in zip(two[0]["params"], one["params"]):
        assert (a - c).abs().max() <= 1e-5 * max(float(c.abs().max()), 1.0), float((a - c).abs().max())
    # the loss each rank reports is its shard's; their mean is the whole batch's
    for s in range(3):
        m = 0.5 * (two[0]["losses"][s] + two[1]["losses"][s])
        assert abs(m - one["losses"][s]) <= 1e-4 * max(abs(one["losses"][s]), 1e-3)
