"""GPU tests of bench.py's contract: one JSON line with the driver's fields, and the N>1 launch path
(two ranks sharing the one visible GPU over gloo -- the real runs use one GPU per rank over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _json_line(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["unit"] == "images/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"] + 1e-9
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "workload" in d["config"]
    # the timed mode is the product default and its parity is MEASURED in the run (against the reference's own 64-step runs, all three of them)
    assert d["precision"]["timed_mode"] == "strict"
    pm = d["precision_modes"]
    assert pm["strict"]["timed"] and pm["strict"]["parity"]["positions"] == 84284 and pm["strict"]["parity"]["token_mismatch"] <= 7e-4
    others = pm["strict"]["parity_other_runs"]
    assert sorted(o["positions"] for o in others.values()) == [84284, 168568] and all(o["token_mismatch"] <= 7e-4 for o in others.values())
    # the faster modes below it, one untimed-region batch each: the differential form without the correction pass (AT the bound) and single fp16
    assert not pm["diff"]["timed"] and pm["diff"]["images_per_s"] > 0 and pm["diff"]["parity"]["token_mismatch"] <= 1.3e-3
    assert 0 < pm["diff"]["ffn_up_frac_of_peak"] < 1
    assert not pm["fp16"]["timed"] and pm["fp16"]["images_per_s"] > 0


def test_bench_two_ranks_on_one_gpu():
    """The bare entry the driver uses: `python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE -- bench.py starts its
    own ranks.  (Two ranks share the one visible GPU over gloo here; on a real node it is one GPU per rank over RCCL.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MB_BENCH_FORCE_DEVICE="0", MB_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-prof"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8
    assert abs(d["value"] - 8 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"] + 1e-9
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["gather_ms"] is not None and d["gather_ms"] >= 0


def test_bench_two_gpus_over_rccl_when_two_devices_are_visible():
    """`bench.py --gpus 2` over RCCL (backend "nccl") on two real devices: skipped on the 1-GPU boxes, the first thing an 8-GPU node runs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MB_BENCH_FORCE_DEVICE",
                                                            "MB_BENCH_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "nccl" and d["config"]["global_batch"] == 16 and d["gather_ms"] > 0


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MB_BENCH_FORCE_DEVICE", "MB_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and f"--gpus {n}" in (r.stderr + r.stdout) and "GPU(s)" in (r.stderr + r.stdout)
