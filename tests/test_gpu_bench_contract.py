"""bench.py prints ONE JSON line with the contract's fields (small sizes; the numbers themselves are not checked)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--graphs", "64", "--nodes", "128", "--k", "5",
                        "--steps", "2", "--warmup", "1", "--cpu-sample-graphs", "4", "--gen-workers", "1", "--cfg3-nodes", "30000"],
                       capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "graphs/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 0.02 * d["value"]
    # the synthetic batch is builder-shaped: the two tracks pair and every direction is one launch, timed as such
    assert d["route_step"].startswith("both tracks in one launch") and set(d["kernels_ms"]) == {"both_fwd", "both_bwd"}
    assert "four_launches" in d["variants"] and set(d["variants"]["four_launches"]["kernels_ms"]) == {"P_fwd", "Q_fwd", "P_bwd", "Q_bwd"}
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["frac"] <= 1.0 and "lds" in rf and rf["lds"]["frac"] <= 1.0 and "algorithmic_GBs" in rf
    assert d["steady_state"]["steps"] == 200 and d["steady_state"]["ms_per_step"] > 0 and d["steady_state"]["unit"] == "graphs/s"
    c3 = d["cfg3"]
    assert c3["nodes"] == 30000 and c3["fwd_ms"] > 0 and c3["roofline"]["bound"] == "hbm" and c3["telescoping_max_abs_err"] < 1e-4
    assert c3["gpu_build"]["pattern_equal_to_host_kdtree"] is True and c3["gpu_build"]["seconds"] > 0
    # the default route of the same blocks: frames recovered across the device, the chain agrees with the block chain;
    # the builder's Q carries its frames and needs the check only
    fc = c3["factored"]
    assert fc["route"].startswith("factored form recovered") and fc["factor_err_rel"] <= 1e-6
    assert fc["max_rel_diff_vs_block_chain"] <= 1e-5 and fc["fwd_ms"] > 0 and fc["recover_s"] > 0
    assert c3["gpu_build"]["frames_hint"]["factored_form_from_the_builders_frames"] is True
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["value"] > 0


def test_bench_two_rank_rehearsal_end_to_end():
    """`bench.py --gpus 2` without a launcher around it starts its own two ranks before anything
    touches the GPU; in rehearsal mode (ESCAT_BENCH_REHEARSAL=1: both ranks share this box's one card, gloo instead of
    RCCL) the whole N > 1 path runs end to end -- shard per rank, barrier, max-over-ranks time, the collective's own
    count of ranks -- and rank 0 alone prints the one JSON line."""
    env = dict(os.environ, ESCAT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--graphs", "64", "--nodes", "128", "--k", "5",
                        "--steps", "2", "--warmup", "1", "--no-cfg3", "--no-cpu-baseline", "--gen-workers", "1"],
                       capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen_by_allreduce"] == 2 and d["collective_backend"] == "gloo"
    assert d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    # weak scaling: 64 graphs per rank, value = all ranks' graphs / max-over-ranks time
    assert abs(d["value"] - 2 * 64 / (d["ms_per_step"] * 1e-3)) <= 0.02 * d["value"]
