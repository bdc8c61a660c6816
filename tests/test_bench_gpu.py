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
    j = _run("--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--leg-steps", "3")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"] == base["metric"] and j["unit"] == "clips/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["dtype"].startswith("f32") and "3 bf16 terms" in j["dtype"] and j["config"]["bf16_split_conv"] is True     # says exactly what runs
    assert j["data"] == "synthetic" and j["vs_baseline"] is None and "model" not in j["config"] and "configs[1]" in j["config"]["workload"]
    assert abs(j["value"] - 8 * 1000.0 / j["ms_per_step"]) < 1e-6 * j["value"]          # whole-job clips/s at bs = 8
    assert 100.0 < j["value"] < 1000.0
    # the headline is the step WITH its per-step input staging (a1 starts at main_ucf101.py:52); the resident figure stands beside it
    assert "staged inside the timed region" in j["config"]["inputs"]
    rs = j["resident"]
    assert rs["unit"] == "clips/s" and abs(rs["value"] - 8 * 1000.0 / rs["ms_per_step"]) < 1e-6 * rs["value"] and j["value_resident"] == rs["value"]
    assert j["ms_per_step"] < 1.5 * rs["ms_per_step"] + 5.0
    fams = [j[k] for k in ("roofline_conv_x6", "roofline_fp32_conv", "roofline_winograd", "roofline_wgrad_x6", "roofline_wgrad_fp32")]
    assert all(f.get("invalid") is None and f["timed_steps"] == 3 for f in fams)             # every leg: three measured replays that agree within 1.5x
    d = j["roofline"]                                       # = the family with the largest single-stream kernel time per step among all FIVE
    assert d["bound"] == "mfma" and d["kernel_ms_per_step"] == max(f["kernel_ms_per_step"] for f in fams) and "dominant_by" in d
    assert abs(d["frac"] - d["achieved"] / d["peak"]) < 1e-9 and 0.1 < d["frac"] < 1.0 and d["kernel_ms_per_step"] < j["ms_per_step"]
    r = j["roofline_conv_x6"]                               # fp32 on the bf16 matrix cores, roof = bf16 peak / 6 products
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["peak"] - 2500.0 / 6) < 1e-6 and "conv_x6_kernel" in r["kernel"]
    assert r["in_region_timed_steps"] >= 1 and 0.9 < r["in_region_kernel_ms_per_step"] / r["kernel_ms_per_step"] < 4.0      # in-step durations are stretched by the other lanes
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.1 < r["frac"] < 1.0
    assert r["launches_per_step"] > 30 and abs(r["avg_launch_ms"] * r["launches_per_step"] - r["kernel_ms_per_step"]) < 1e-6 * r["kernel_ms_per_step"]
    assert r["kernel_ms_per_step"] < j["ms_per_step"]                                    # event time of the conv kernels fits inside the step
    assert r["traffic"] is None or r["traffic"] > 1e6
    assert r["frac_valid"] <= r["frac"] <= r["frac_mfma_issued"] < 1.0
    f = j["roofline_fp32_conv"]                              # the launches that stay on the fp32 MFMA kernels
    assert f["peak"] == 157.3 and f["launches_per_step"] >= 5 and 0.1 < f["frac"] <= f["frac_mfma_issued"] < 1.0
    w = j["roofline_winograd"]                               # the Winograd family, timed in its own replays
    assert w["launches_per_step"] >= 8 and 0.2 < w["frac"] <= w["frac_mfma_issued"] < 1.0 and w["direct_equivalent_tflops"] > w["achieved"]
    gx, gf = j["roofline_wgrad_x6"], j["roofline_wgrad_fp32"]  # the weight gradients: bf16-split launches against bf16 peak / 6, the fp32 ones against 157.3
    assert abs(gx["peak"] - 2500.0 / 6) < 1e-6 and "wgrad3_x6_kernel" in gx["kernel"] and gx["launches_per_step"] >= 30 and 0.1 < gx["frac"] <= gx["frac_mfma_issued"] < 1.0
    assert gf["peak"] == 157.3 and "wgrad3_kernel<.., 9>" in gf["kernel"] and 2 <= gf["launches_per_step"] <= 16 and 0.1 < gf["frac"] <= gf["frac_mfma_issued"] < 1.0
    assert "wgrad4_x6_kernel" in gx["kernel"]            # the stem's weight gradient runs on the bf16 split since round 6
    assert gx["launches_per_step"] + gf["launches_per_step"] >= 46
    wg_gf = (gx["flops_per_launch"] * gx["launches_per_step"] + gf["flops_per_launch"] * gf["launches_per_step"]) / 1e9
    assert abs(wg_gf - j["roofline_step"]["gflop_by_family"]["wgrad"]) < 1e-3 * wg_gf            # the two legs' numerators are the whole family's FLOPs
    assert j["config"]["winograd_launches"] == {"F(4x4,3x3)": 5, "F(2x2,3x3)": 21} and j["config"]["experiment_switches"] == [] and "no atomics" in j["config"]["weight_gradients"]
    fam = j["roofline_step"]["gflop_by_family"]
    assert fam["conv_bf16_split"] > fam["conv_fp32_mfma"] > 0 and fam["wgrad"] > 0 and fam["winograd_conv"] > 0
    dc = j["dict_contract"]                                 # the reference's float64 host dicts inside the timed region
    assert dc["steps"] == 3 and abs(dc["value"] - 8 * 1000.0 / dc["ms_per_step"]) < 1e-6 * dc["value"] and dc["loss_total"] == dc["loss_total"]
    so = j["split_off"]                                     # PICONS_SPLIT=0 in the same line
    assert so["steps"] == 3 and so["ms_per_step"] > 0 and so["loss_total"] == so["loss_total"]
    assert j["ranks_observed"] == 1 and j["busy_steps_outside_timed_regions"] >= 20
    assert all(k in j["loss"] for k in ("total", "loc", "cls", "cons")) and j["cpu_baseline"] is None


def test_bench_split_off_keeps_the_native_path():
    """PICONS_SPLIT=0: every conv launch on the fp32 MFMA kernels, dtype and roofline say so."""
    j = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs", env={"PICONS_SPLIT": "0"})
    assert j["dtype"] == "f32" and j["config"]["bf16_split_conv"] is False and j["split_off"] is None
    r = j["roofline_fp32_conv"]
    assert r["peak"] == 157.3 and "conv_gemm" in r["kernel"] and r["launches_per_step"] > 50 and 0.3 < r["frac"] < 1.0
    assert j["roofline_conv_x6"] is None and j["roofline"]["kernel_ms_per_step"] >= r["kernel_ms_per_step"]
    assert j["roofline_step"]["gflop_by_family"]["conv_bf16_split"] == 0


def test_bench_without_timing_leg_is_not_slower():
    a = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--resident-inputs", "--no-extra-legs")
    assert a["roofline"] is None and a["ms_per_step"] < 60.0 and a["resident"] is None and "resident in HBM" in a["config"]["inputs"]


def test_bench_dp_schedule_through_rccl_on_one_rank():
    """PICONS_FORCE_REDUCER=1: the N > 1 code path (one-rank RCCL group, segmented backward, bucket all-reduces, rank count from an
    all-reduce of ones) and still exactly one stdout line -- RCCL's version banner must not land on stdout."""
    j = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-extra-legs",
             env={"PICONS_FORCE_REDUCER": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert j["ranks_observed"] == 1 and j["reducer"]["backend"] == "nccl" and j["reducer"]["buckets"] >= 3 and j["reducer"]["forced_single_rank"]
    assert j["ms_per_step"] < 60.0 and j["loss"]["total"] == j["loss"]["total"]
    # the diagnostics a SCALE line is explained with: host time in reducer.wait(), device time the main stream sat behind its last backward
    # kernel until the last collective finished, and the bucket table (bytes, ready-op index)
    rd = j["reducer"]
    assert rd["steps"] == 3 and 0.0 <= rd["comm_wait_ms"] < 50.0 and 0.0 <= rd["exposed_ms"] < 50.0 and rd["exposed_ms_max"] >= rd["exposed_ms"]
    tab = rd["bucket_table"]
    assert len(tab) == rd["buckets"] and all(t["bytes"] > 0 and t["ready_op"] > 0 for t in tab)
    assert [t["ready_op"] for t in tab] == sorted(t["ready_op"] for t in tab) and sum(t["bytes"] for t in tab) > 150e6
