"""GPU: the driver-facing contract of bench.py -- ONE JSON line on stdout with the metric BASELINE.json names, the
whole-job value, the live roofline leg (HIP events on the kernels' own streams) and, when asked, the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(*flags, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_line_contract():
    j = _run("--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"] == base["metric"] and j["unit"] == "clips/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["dtype"] == "f32"
    assert j["data"] == "synthetic" and j["vs_baseline"] is None and "model" not in j["config"] and "configs[1]" in j["config"]["workload"]
    assert abs(j["value"] - 8 * 1000.0 / j["ms_per_step"]) < 1e-6 * j["value"]          # whole-job clips/s at bs = 8
    assert 100.0 < j["value"] < 1000.0
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["launches_per_step"] > 50 and abs(r["avg_launch_ms"] * r["launches_per_step"] - r["kernel_ms_per_step"]) < 1e-6 * r["kernel_ms_per_step"]
    assert r["kernel_ms_per_step"] < j["ms_per_step"]                                    # event time of the conv kernels fits inside the step
    assert r["traffic"] is None or r["traffic"] > 1e6
    # the numerator is what the kernels execute (host walk of every launch's tiles); the older, larger bookings stay beside it
    assert r["frac_valid"] <= r["frac"] <= r["frac_mfma_issued"] < 1.0 and r["frac"] < min(r["frac_descriptor_counted"], r["frac_reference_counted"])
    w = j["roofline_winograd"]                               # the second GEMM family, timed in its own replays
    assert w["launches_per_step"] >= 8 and 0.2 < w["frac"] <= w["frac_mfma_issued"] < 1.0 and w["direct_equivalent_tflops"] > w["achieved"]
    assert r["frac_reference_counted"] < 1.0
    st = j["staged"]                                        # the same step with per-step input staging inside the timed region
    assert st["unit"] == "clips/s" and st["steps"] == 3 and abs(st["value"] - 8 * 1000.0 / st["ms_per_step"]) < 1e-6 * st["value"]
    assert st["ms_per_step"] < 1.5 * j["ms_per_step"] + 5.0 and st["loss_total"] == st["loss_total"]
    assert j["ranks_observed"] == 1 and "resident" in j["config"]["inputs"]
    assert all(k in j["loss"] for k in ("total", "loc", "cls", "cons")) and j["cpu_baseline"] is None


def test_bench_without_timing_leg_is_not_slower():
    a = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--resident-inputs")
    assert a["roofline"] is None and a["ms_per_step"] < 60.0 and a["staged"] is None


def test_bench_dp_schedule_through_rccl_on_one_rank():
    """PICONS_FORCE_REDUCER=1: the N > 1 code path (one-rank RCCL group, segmented backward, bucket all-reduces, rank count from an
    all-reduce of ones) and still exactly one stdout line -- RCCL's version banner must not land on stdout."""
    j = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing",
             env={"PICONS_FORCE_REDUCER": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert j["ranks_observed"] == 1 and j["reducer"]["backend"] == "nccl" and j["reducer"]["buckets"] >= 3 and j["reducer"]["forced_single_rank"]
    assert j["ms_per_step"] < 60.0 and j["loss"]["total"] == j["loss"]["total"]
