"""bench.py contract (task prompt section 4): one JSON line on rank 0 with the required keys, run directly and through
torch.distributed.run with one rank (the driver's launch line), on a small configuration so it finishes in seconds."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--batch", "4", "--arch", "dinov2_vits14", "--image-size", "224", "--no-episode",
         "--cpu-batches", "1,4", "--cpu-runs", "1", "--no-alt"]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(line, with_cpu):
    d = json.loads(line)
    missing = (REQUIRED - ({"cpu_baseline"} if not with_cpu else set())) - set(d)
    assert not missing, missing
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02      # pairs per step / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    assert isinstance(d["pipelined"], bool)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["launches_timed"] == 3      # sampled: one QKV launch per step
    assert isinstance(d["env"], dict) and all(k.startswith("EC_") for k in d["env"]) and len(d["library_source_hash"]) == 16
    assert "traffic" in r and (r["traffic"] is not None or "traffic_note" in r)      # never a number from another library build
    if with_cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and 1 <= c["cores"] <= c["logical_cpus"] and "sample" in c
        assert set(c["per_batch_size"]) == {"1", "4"} and c["value"] == c["per_batch_size"]["4"]["pairs_per_s"]
        ps = d["parity_sample"]                                         # fp16 backbone + bf16x3 head: inside the stated tolerance
        assert ps["pairs"] == 4 and ps["tolerance"] == 1e-3 and ps["within_tolerance"] and ps["argmax_flips"] == 0
        assert ps["pck@0.2_hip_vs_oracle_pred"] == 1.0


@pytest.mark.gpu
def test_bench_direct_small():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    _check(lines[0], with_cpu=True)


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline"] + SMALL
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    _check(lines[0], with_cpu=False)


@pytest.mark.gpu
def test_bench_unpipelined_flag_and_companion_run():
    """--no-pipeline times ec_forward; the default line times ec_forward_pipelined and carries the ec_forward figure beside it."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline"] + [a for a in SMALL if a != "--no-alt"]
    out = subprocess.run(base, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["pipelined"] is True and d["unpipelined"]["value"] > 0 and d["unpipelined"]["ms_per_step"] > 0 and "bf16_mode" in d
    # which number is which: roofline.* belongs to the timed region of `value`, the chip-alone figure sits beside it
    r = d["roofline"]
    assert "the timed region of `value`" in r["measured_in"] and r["launches_timed"] == d["steps"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel_isolated"]["frac"] > 0 and r["kernel_isolated"]["avg_launch_ms"] > 0
    assert d["value_reference_contract"] == d["unpipelined"]["value"]
    c = d["conforming_mode"]                                 # the tolerance-conforming mode, measured in the same process
    assert c["value"] > 0 and c["roofline"]["peak"] == 1250.0 and 0 < c["roofline"]["frac"] < 1 and c["precision"].startswith("fp16x2 backbone / bf16x3")
    assert c["bf16x3_bf16x3"]["value"] > 0                   # round 5's conforming mode, measured beside it
    assert c["value"] > 0.9 * c["bf16x3_bf16x3"]["value"]    # two MFMA units per product against three (this 8-image step is launch-bound: no more than a sanity bound)
    out = subprocess.run(base + ["--no-pipeline", "--no-alt"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["pipelined"] is False and "unpipelined" not in d
