#!/usr/bin/env python3
"""Headline benchmark: train clips/sec of the full semi-supervised step (I3D + capsule head + decoder,
forward x2 + bv consistency + dice/BCE/spread losses + backward + Adam) on synthetic UCF101-24-shaped
data, bs=8 per GPU, clips 8x224x224 (= the reference's 16-frame span at stride 2, SURVEY finding 1).

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel =
the fp32-MFMA gather-GEMM conv kernel, timed with hipEvents on the launch stream) and `cpu_baseline`
(the CPU oracle = plain-PyTorch port of the reference, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak


def _baseline_metric():
    """The metric string is BASELINE.json's own (repo root; it travels with the snapshot)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "train clips/sec (bs=8, 16\u00d7224\u00d7224, I3D+caps+bv cons) at 1/2/4/8 GPU"


METRIC = _baseline_metric()


def cpu_baseline(seconds_budget=30.0):
    """CPU oracle (oracle/, kind 'port') on a bounded sample: one bs=2 train step (2 forward passes,
    losses, backward, Adam) at 8x224x224 with all host threads."""
    from oracle import step as ostep
    from picons_amd import synthetic
    from picons_amd.step import exp_rampup
    # a bounded thread count: with all 256 host threads of the GPU box torch's CPU convolutions oversubscribe
    # (the same step took 448 s there); 16 threads finish it in tens of seconds
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    P = ostep.as_torch_params(synthetic.init_state(47, 24))
    a = ostep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    lab, unl, perm, drops = synthetic.make_step_inputs(2, rank=0, step=0)
    t0 = time.time()
    r = ostep.train_step(P, a, lab, unl, 1, exp_rampup(100)(1), perm, drops)
    r["total"].backward()
    ostep.adam_step(P, {}, {}, 1, 1e-4)
    dt = time.time() - t0
    return {"value": 2.0 / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "1 full train step at bs=2 (1 labeled + 1 unlabeled), 8x224x224, --bv --n_frames 5, %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)        # >= 20 timed steps after >= 5 warm-up
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--gv", action="store_true", help="BASELINE config 3 (--gv instead of --bv)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    import picons_amd  # noqa: F401
    from picons_amd import capi, dist as pdist, step as pstep, synthetic
    rank, world, local = pdist.init_from_env()
    if world != a.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)

    args = pstep.default_args(bv=not a.gv, gv=a.gv, n_frames=5, wt_cons=0.1, lr=1e-4, epochs=100, thresh_epoch=11)
    eng = pstep.StepEngine(args, bs=a.bs, hw=224, num_classes=24, device=dev)
    lab, unl, perm, drops = synthetic.make_step_inputs(a.bs, rank=rank, step=0)
    eng.stage(lab, unl, perm, drops)                     # inputs resident in HBM before the timed region
    reducer = eng.make_reducer() if world > 1 else None
    ramp = pstep.exp_rampup(100)(1)
    kind = None if a.no_kernel_timing else capi.OP_CONV

    for _ in range(a.warmup):
        eng.run_staged(1, ramp, reducer=reducer)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    eng.kind_ms, eng.kind_count = 0.0, 0
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last = eng.run_staged(1, ramp, reducer=reducer, timed_kind=kind, collect=False)   # events recorded here, read below
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    ms_local = (time.perf_counter() - t0) * 1e3
    if kind is not None:
        eng.collect_timing()
    ms_total = pdist.barrier_max_ms(ms_local, device=dev)
    ms_step = ms_total / a.steps
    value = world * a.bs * a.steps / (ms_total / 1e3)

    fl = eng.plan.flops()
    fc = eng.plan.flops(capi.OP_CONV)
    conv_flops_step = fc["fwd"] + fc["bwd"]              # algorithmic (real channel counts), DESIGN.md §4
    # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in their own runs, tools/summarize_pmc.py); counters cannot be read from inside a run
    traffic = None
    for tag in ("r01",):
        tp = os.path.join(ROOT, "profiles", tag + "_traffic.json")
        if os.path.exists(tp) and a.bs == 8:
            try:
                traffic = json.load(open(tp))["conv_gemm_hbm_bytes_per_launch"]
            except Exception:
                traffic = None
    roof = None
    if kind is not None and eng.kind_count:
        avg_ms = eng.kind_ms / eng.kind_count
        flops_per_launch = conv_flops_step * a.steps / eng.kind_count
        ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_gemm_kernel (fp32 MFMA gather-GEMM: conv fwd / dgrad / convT)",
                "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                "traffic": traffic, "launches_per_step": eng.kind_count // a.steps, "avg_launch_ms": avg_ms,
                "kernel_ms_per_step": eng.kind_ms / a.steps,
                "flops_per_launch": flops_per_launch}
    out = {
        "metric": METRIC,
        "value": value, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: UCF101-24-shaped synthetic, I3D+caps, 8 frames (16-frame span, stride 2) x224x224, "
                               "bs=%d/GPU (bs/2 labeled + bs/2 unlabeled), %s consistency, dice+BCE loc loss, spread cls loss, Adam"
                               % (a.bs, "--gv" if a.gv else "--bv --n_frames 5 L2"),
                   "global_batch": world * a.bs, "clip": [3, 8, 224, 224], "parallelism": "dp%d" % world,
                   "algorithmic_gflop_per_step_per_gpu": (fl["fwd"] + fl["bwd"]) / 1e9},
        "loss": last,
        "roofline": roof,
    }
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
