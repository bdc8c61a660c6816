#!/usr/bin/env python3
"""Headline benchmark: train clips/sec of the full semi-supervised step (I3D + capsule head + decoder,
forward x2 + bv consistency + dice/BCE/spread losses + backward + Adam) on synthetic UCF101-24-shaped
data, bs=8 per GPU, clips 8x224x224 (= the reference's 16-frame span at stride 2, SURVEY finding 1).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a torchrun environment (no RANK): this process touches no GPU, starts N child
processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), relays rank 0's single JSON
line and exits non-zero if any child does.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
the ranks are taken from the environment as before.  `--dry-launch` makes the ranks only rendezvous and all-reduce ones (gloo
when the box has fewer than N GPUs): the launcher's own test, runnable without GPUs (tests/test_bench_launch_cpu.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel =
the fp32-MFMA gather-GEMM conv kernel, timed with hipEvents on the launch stream) and `cpu_baseline`
(the CPU oracle = plain-PyTorch port of the reference, timed on this host's cores on a bounded sample).
`value` is measured with the minibatch resident in HBM (contract); `staged` in the same line is the same step with every
step's minibatch prepared from host uint8 frames by the device input pipeline (upload + crop / flip / mask kernels + the
reference's cat / shuffle), overlapped with the previous step (main_ucf101.py:52-79 inside the timed region).
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


def cpu_baseline(timed=3):
    """CPU oracle (oracle/, kind 'port') on a bounded sample of the same workload: full train steps (2 forward passes, losses,
    backward, Adam) at bs=2, 8x224x224 -- one untimed warm-up step, then `timed` steps, median reported."""
    from oracle import step as ostep
    from picons_amd import synthetic
    from picons_amd.step import exp_rampup
    # a bounded thread count: with all 256 host threads of the GPU box torch's CPU convolutions oversubscribe
    # (the same step took 448 s there); 16 threads finish it in tens of seconds
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    P = ostep.as_torch_params(synthetic.init_state(47, 24))
    a = ostep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    m, v, times = {}, {}, []
    for it in range(1 + timed):
        lab, unl, perm, drops = synthetic.make_step_inputs(2, rank=0, step=it)
        t0 = time.time()
        for p in P.values():
            p.grad = None
        r = ostep.train_step(P, a, lab, unl, 1, exp_rampup(100)(1), perm, drops)
        r["total"].backward()
        ostep.adam_step(P, m, v, it + 1, 1e-4)
        times.append(time.time() - t0)
    dt = float(np.median(times[1:]))
    return {"value": 2.0 / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "median of %d full train steps after 1 warm-up at bs=2 (1 labeled + 1 unlabeled), 8x224x224, --bv --n_frames 5; "
                      "step times %s s" % (timed, ["%.1f" % t for t in times])}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, timeout=0.0):
    """Parent of an N-rank run: NO GPU call happens in this process (a process that has initialised the GPU must not be the one
    that turns into / spawns the ranks' interpreter on this pool).  Starts n fresh interpreters of this file, one per GPU, with a
    torchrun-style environment; rank 0's stdout is captured and its JSON line relayed, every other stream goes to stderr.
    Returns the exit code: 0 only if every rank exited 0 and rank 0 printed its line."""
    import signal
    import subprocess
    import tempfile
    port = _free_port()
    procs, out0 = [], tempfile.TemporaryFile(mode="w+b")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PICONS_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=out0 if r == 0 else sys.stderr, stderr=sys.stderr))

    def stop_all():
        for q in procs:                       # exact PIDs we started, never a pattern
            if q.poll() is None:
                q.terminate()
        t_end = time.time() + 10.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
    signal.signal(signal.SIGTERM, lambda *_: (stop_all(), sys.exit(143)))
    rc, t0 = 0, time.time()
    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % bad[0], file=sys.stderr)
                rc = bad[0][1] if bad[0][1] > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if timeout and time.time() - t0 > timeout:
                print("bench.py: ranks still running after %.0f s; stopping them" % timeout, file=sys.stderr)
                rc = 124
                break
            time.sleep(0.1)
    except KeyboardInterrupt:
        rc = 130
    finally:
        stop_all()
    if rc == 0:
        out0.seek(0)
        lines = [ln for ln in out0.read().decode(errors="replace").splitlines() if ln.strip()]
        line = None
        for ln in lines:
            try:
                j = json.loads(ln)
            except ValueError:
                continue
            if isinstance(j, dict) and "n_gpus" in j:
                line = (ln, j)
        if line is None:
            print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
            return 1
        if line[1].get("n_gpus") != n or line[1].get("ranks_observed") != n:
            print("bench.py: asked for %d ranks, the line says n_gpus=%s ranks_observed=%s" % (n, line[1].get("n_gpus"), line[1].get("ranks_observed")),
                  file=sys.stderr)
            return 1
        sys.stdout.write(line[0] + "\n")
        sys.stdout.flush()
    return rc


def dry_launch(a, json_fd):
    """--dry-launch: the ranks rendezvous, all-reduce ones and barrier -- nothing else.  Exercises exactly the launch / environment /
    rank-count logic of an N-rank run (tests/test_bench_launch_cpu.py runs it with 2 ranks over gloo on CPU)."""
    import torch.distributed as dist
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PICONS_BENCH_FAIL_RANK") == str(rank):       # test hook: this rank dies before the rendezvous
        sys.exit(7)
    backend = a.dry_backend
    if backend == "auto":                 # device_count() does not initialise the GPU on this image
        backend = "nccl" if torch.cuda.device_count() >= max(world, 1) and torch.cuda.device_count() > 0 else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    ones = torch.ones(1, device=("cuda:%d" % local) if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    dist.barrier()
    out = {"metric": METRIC, "dry_launch": True, "n_gpus": world, "ranks_observed": int(ones.item()), "value": None, "unit": "clips/s",
           "reducer": {"backend": dist.get_backend()}, "local_rank": local, "master": "%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])}
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()
    return 0 if out["ranks_observed"] == world else 1


class StagedInputs:
    """The reference's per-step input work inside the timed region (main_ucf101.py:52-79: casts, cat, randperm shuffle, H2D), done
    the MI355X way: the decoded uint8 frames of the step's 8 samples go up (1.8 MB each), pc_clip_from_u8 writes data / aug_data /
    loc_msk on the device, StepEngine.stage concatenates + shuffles into the arena.  The next step's samples are prepared on a
    side stream while the current step runs (what tools/bench_step_u8.py measures as 'overlapped')."""

    def __init__(self, eng, bs, ncls, rank):
        from picons_amd import inputpipe, synthetic
        self.eng, self.bs, self.ip = eng, bs, inputpipe
        self.vids = [synthetic.make_decoded_video(100 + 97 * rank + i, labeled=(i % bs) < bs // 2, num_classes=ncls) for i in range(2 * bs)]
        self.side = torch.cuda.Stream(device=eng.dev)
        self.g = torch.Generator().manual_seed(1234 + rank)
        np.random.seed(1234 + rank)          # the loader's own np.random draws (crop offsets, frame choice)

    @staticmethod
    def collate(samples):
        return {'data': torch.stack([s['data'] for s in samples]), 'aug_data': torch.stack([s['aug_data'] for s in samples]),
                'loc_msk': torch.stack([s['loc_msk'] for s in samples]), 'action': torch.stack([s['action'] for s in samples]),
                'label_vid': torch.tensor([s['label_vid'] for s in samples])}

    def prep(self, i):
        n, nv, h = self.bs, len(self.vids), self.bs // 2
        with torch.cuda.stream(self.side):
            lab = self.collate([self.ip.get_item(*self.vids[(n * i + j) % nv], train=True) for j in range(h)])
            unl = self.collate([self.ip.get_item(*self.vids[(n * i + h + j) % nv], train=True) for j in range(h)])
            ev = torch.cuda.Event()
            ev.record(self.side)
        perm = torch.randperm(n, generator=self.g).numpy()
        drops = [(torch.rand(n, c, generator=self.g) < 0.5).float().numpy() * 2 for c in (832, 128, 832, 128)]
        return lab, unl, perm, drops, ev

    def run(self, steps, epoch, ramp, reducer, lr):
        """`steps` full steps, each on a fresh minibatch; returns (seconds, last losses)."""
        eng = self.eng
        nxt = self.prep(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            lab, unl, perm, drops, ev = nxt
            torch.cuda.current_stream().wait_event(ev)
            eng.stage(lab, unl, perm, drops)
            eng.arm_early_adam(lr, on=reducer is None or not reducer.active)
            eng.forward_backward(epoch, ramp, reducer)
            gscale = 1.0
            if reducer is not None:
                reducer.wait()
                gscale = reducer.gscale
            eng.adam(lr, gscale)
            nxt = self.prep(i + 1)            # host decisions + uploads + pc_clip_from_u8 overlap the step enqueued above
            out = eng.read_scalars()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # >= 20 timed steps after >= 5 warm-up; 200 x 30 ms
                                                            # keeps the GPU busy long enough for an external sampler to see it
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--gv", action="store_true", help="BASELINE config 3 (--gv instead of --bv)")
    ap.add_argument("--jhmdb", action="store_true", help="BASELINE config 5's shape: 21 classes, main_jhmdb.py's step")
    ap.add_argument("--epoch", type=int, default=1, help="epoch the step runs at (>= 11 = --thresh_epoch: argmax pseudo-labels "
                    "for the unlabeled rows, capsules_ucf101.py:463)")
    ap.add_argument("--time-every", type=int, default=40, help="attach hipEvent pairs to the conv kernels of every n-th timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--resident-inputs", action="store_true", help="skip the `staged` leg (every step's minibatch prepared from host uint8 frames)")
    ap.add_argument("--staged-steps", type=int, default=0, help="timed steps of the `staged` leg (default: min(steps, 50))")
    ap.add_argument("--dry-launch", action="store_true", help="ranks only rendezvous + all-reduce ones (launcher test; no GPU work)")
    ap.add_argument("--dry-backend", default="auto", choices=["auto", "gloo", "nccl"])
    ap.add_argument("--launch-timeout", type=float, default=0.0, help="seconds after which the launcher stops its ranks (0 = never)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # the launcher: nothing above or below in this branch touches the GPU (importing torch does not)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:], a.launch_timeout))

    # stdout carries ONE JSON line: RCCL writes a version banner to C stdio's stdout (flushed at exit, i.e. after anything printed
    # here), so fd 1 is pointed at stderr for the whole run and the line goes out through a saved copy of the real stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != a.gpus:
        # a line labelled with a rank count that is not the one asked for would be recorded as the wrong configuration
        print("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to run (launch with matching values, or without a torchrun environment "
              "so that bench.py starts the ranks itself)" % (a.gpus, env_world), file=sys.stderr)
        sys.exit(2)
    if a.dry_launch:
        sys.exit(dry_launch(a, json_fd))

    import picons_amd  # noqa: F401
    from picons_amd import capi, dist as pdist, step as pstep, synthetic
    if env_world > 1 and int(os.environ.get("LOCAL_RANK", "0")) >= torch.cuda.device_count():
        print("bench.py: LOCAL_RANK %s but this node has %d GPU(s)" % (os.environ.get("LOCAL_RANK"), torch.cuda.device_count()), file=sys.stderr)
        sys.exit(3)
    rank, world, local = pdist.init_from_env()
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)

    args = pstep.default_args(bv=not a.gv, gv=a.gv, n_frames=5, wt_cons=0.1, lr=1e-4, epochs=100, thresh_epoch=11)
    ncls = 21 if a.jhmdb else 24
    eng = pstep.StepEngine(args, bs=a.bs, hw=224, num_classes=ncls, jhmdb=a.jhmdb, device=dev)
    lab, unl, perm, drops = synthetic.make_step_inputs(a.bs, rank=rank, step=0, num_classes=ncls)
    eng.stage(lab, unl, perm, drops)                     # inputs resident in HBM before the timed region
    # PICONS_FORCE_REDUCER=1 on one GPU: a one-rank RCCL group and the whole DP schedule (segmented backward, bucket all-reduces on
    # the comm stream behind events, 1/world in Adam) -- the N > 1 code path executed through RCCL where only one GPU exists
    forced = world == 1 and os.environ.get("PICONS_FORCE_REDUCER", "0") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    reducer = eng.make_reducer(force=forced) if (world > 1 or forced) else None
    ramp = pstep.exp_rampup(100)(a.epoch)
    kind = None if a.no_kernel_timing else capi.OP_CONV
    ranks_observed = 1
    if world > 1 or forced:                              # an all-reduce of ones: the rank count the collective really spans
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        ranks_observed = int(ones.item())
    if ranks_observed != world:
        print("bench.py: the collective spans %d ranks, WORLD_SIZE is %d" % (ranks_observed, world), file=sys.stderr)
        sys.exit(4)

    for _ in range(a.warmup):
        eng.run_staged(a.epoch, ramp, reducer=reducer)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    eng.kind_ms, eng.kind_count = 0.0, 0
    t0 = time.perf_counter()
    last = None
    timed_steps = 0
    for it in range(a.steps):
        tk = kind if (kind is not None and it % max(1, a.time_every) == 0) else None
        timed_steps += tk is not None
        last = eng.run_staged(a.epoch, ramp, reducer=reducer, timed_kind=tk, collect=False)   # events recorded here, read below
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    ms_local = (time.perf_counter() - t0) * 1e3
    if kind is not None:
        eng.collect_timing()
    ms_total = pdist.barrier_max_ms(ms_local, device=dev)
    ms_step = ms_total / a.steps
    value = world * a.bs * a.steps / (ms_total / 1e3)

    # ---- the second GEMM family, the Winograd conv kernel: its launches get hipEvent pairs in two extra single-stream steps (outside the
    # timed region above, whose event legs belong to the gather-GEMM family: one op kind per timed replay)
    conv_kind_ms, conv_kind_count = eng.kind_ms, eng.kind_count
    wino_ms = wino_count = wino_steps = 0
    if kind is not None:
        eng.kind_ms, eng.kind_count = 0.0, 0
        for _ in range(2):
            eng.run_staged(a.epoch, ramp, reducer=reducer, timed_kind=capi.OP_WINO_CONV, collect=False)
            wino_steps += 1
        torch.cuda.synchronize()
        eng.collect_timing()
        wino_ms, wino_count = eng.kind_ms, eng.kind_count
        eng.kind_ms, eng.kind_count = conv_kind_ms, conv_kind_count

    # ---- the same step with the reference's per-step input work inside the timed region (every rank, same barrier / max rule)
    staged = None
    if not a.resident_inputs and a.bs % 2 == 0 and not a.jhmdb:
        n_st = a.staged_steps or min(a.steps, 50)
        si = StagedInputs(eng, a.bs, ncls, rank)
        si.run(min(3, n_st), a.epoch, ramp, reducer, args.lr)                      # warm-up (pinned buffers, first uploads)
        if world > 1:
            torch.distributed.barrier()
        sec, last_st = si.run(n_st, a.epoch, ramp, reducer, args.lr)
        if world > 1:
            torch.distributed.barrier()
        ms_st = pdist.barrier_max_ms(sec * 1e3, device=dev)
        staged = {"value": world * a.bs * n_st / (ms_st / 1e3), "unit": "clips/s", "ms_per_step": ms_st / n_st, "steps": n_st,
                  "loss_total": last_st["total"],
                  "what": "every step stages a fresh minibatch from host uint8 frames: upload of the 8 selected frames per sample, "
                          "pc_clip_from_u8 (crop / flip / mask / normalise on the device), cat + shuffle into the arena; the next step's "
                          "samples are prepared on a side stream while the current step runs (main_ucf101.py:52-79 inside the metric)"}

    fl = eng.plan.flops()                                # FLOPs of the emitted (tap-trimmed) descriptors, real channel counts
    fe = eng.plan.conv_flops_executed()                  # conv / dgrad FLOPs as the kernels run them (host walk of every launch's tiles)
    fw = eng.plan.wgrad_flops_executed()
    fc = eng.plan.flops(capi.OP_CONV)
    fr = eng.plan.flops_reference_counted()
    lists = ("fwd", "bwd")
    conv_exec = sum(fe[n]["executed"] for n in lists)
    conv_mfma = sum(fe[n]["mfma"] for n in lists)
    conv_valid = sum(fe[n]["valid"] for n in lists)
    conv_desc = sum(fc[n] for n in lists)                # 2*M*N*K of the trimmed descriptors: round 2's numerator (over-books: the kernel
                                                         # skips, per tile, every tap that is padding for the whole tile)
    conv_ref = sum(fr[n] for n in lists)                 # all taps incl. zero padding, dgrad at its layer's forward FLOPs
    fz = eng.plan.wino_flops_executed()                  # the Winograd launches: transform-domain FLOPs (2.25x fewer than the direct form's)
    wino_exec = sum(fz[n]["executed"] for n in lists)
    step_exec = conv_exec + wino_exec + sum(fw[n]["executed"] for n in lists)
    # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in their own runs, tools/summarize_pmc.py); counters cannot be read from inside a run, so this
    # is OFFLINE data from the named file, not a measurement of this run
    traffic, traffic_src = None, None
    for tag in ("r03", "r02", "r01"):
        tp = os.path.join(ROOT, "profiles", tag + "_traffic.json")
        if traffic is None and os.path.exists(tp) and a.bs == 8:
            try:
                traffic = json.load(open(tp))["conv_gemm_hbm_bytes_per_launch"]
                traffic_src = "profiles/%s_traffic.json (offline rocprofv3 --pmc passes, not this run)" % tag
            except Exception:
                traffic = None
    roof = None
    if kind is not None and eng.kind_count:
        avg_ms = eng.kind_ms / eng.kind_count
        per = lambda f: f * timed_steps / eng.kind_count / (avg_ms * 1e-3) / 1e12     # TFLOP/s of the average launch
        ach = per(conv_exec)
        roof = {"bound": "mfma", "kernel": "conv_gemm_glds_kernel / conv_gemm_kernel (fp32 MFMA gather-GEMM: conv fwd / dgrad / convT)",
                "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                "flops_counted": "executed: per launch the host walks the kernel's tiles (pc_conv_work) and counts real rows x real columns x the K "
                                 "each block walks -- taps that are padding for a whole tile are skipped by the kernel and NOT counted",
                "frac_mfma_issued": per(conv_mfma) / PEAK_FP32_MFMA_TFLOPS,          # whole tiles: what an MFMA instruction counter sees
                "frac_valid": per(conv_valid) / PEAK_FP32_MFMA_TFLOPS,               # non-padding MACs only
                "frac_descriptor_counted": per(conv_desc) / PEAK_FP32_MFMA_TFLOPS,   # round 2's `frac`
                "frac_reference_counted": per(conv_ref) / PEAK_FP32_MFMA_TFLOPS,     # round 1's `frac`
                "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": eng.kind_count // max(1, timed_steps), "avg_launch_ms": avg_ms,
                "kernel_ms_per_step": eng.kind_ms / max(1, timed_steps), "timed_steps": timed_steps,
                "flops_per_launch": conv_exec * timed_steps / eng.kind_count}
    roof_wino = None
    if wino_count:
        fzw = sum(fz[n]["executed"] for n in lists)
        fzm = sum(fz[n]["mfma"] for n in lists)
        fzd = sum(eng.plan.flops_reference_counted_wino()[n] for n in lists)
        wms = wino_ms / wino_steps
        roof_wino = {"bound": "mfma", "kernel": "wino_conv_kernel (Winograd F(2x2,3x3) conv / input gradient, fp32 MFMA in the transform domain)",
                     "achieved": fzw / (wms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": fzw / (wms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                     "frac_mfma_issued": fzm / (wms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                     "direct_equivalent_tflops": fzd / (wms * 1e-3) / 1e12,      # the 3x3x3 formulation's FLOPs over the same time (2.25x the transform-domain work)
                     "launches_per_step": wino_count // wino_steps, "kernel_ms_per_step": wms, "timed_steps": wino_steps,
                     "flops_counted": "executed transform-domain FLOPs on real tiles / channels (pc_wino_work)"}
    roof_step = {"executed_gflop_per_step": step_exec / 1e9, "achieved": step_exec / (ms_step * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                 "unit": "TFLOP/s", "frac": step_exec / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                 "descriptor_gflop_per_step": sum(fl[n] for n in lists) / 1e9,
                 "gflop_by_family": {"conv_gemm": conv_exec / 1e9, "winograd_conv": wino_exec / 1e9, "wgrad": sum(fw[n]["executed"] for n in lists) / 1e9},
                 "note": "whole step (all kernels, host gaps and the loss read-back included) against the fp32 MFMA roof, executed GEMM FLOPs "
                         "(gather-GEMM conv / dgrad, Winograd conv / dgrad in the transform domain, wgrad; pc_conv_work + pc_wino_work + pc_wgrad_work); the reference's own formulation would need 6 185 GFLOP/step, "
                         "most of which are removed algebraically (DESIGN.md 3)"}
    out = {
        "metric": METRIC,
        "value": value, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[4] on one rank: JHMDB-21-shaped" if a.jhmdb else "configs[1]: UCF101-24-shaped") + " synthetic, I3D+caps, 8 frames (16-frame span, stride 2) x224x224, "
                               "bs=%d/GPU (bs/2 labeled + bs/2 unlabeled), %s consistency, dice+BCE loc loss, spread cls loss, Adam"
                               % (a.bs, "--gv" if a.gv else "--bv --n_frames 5 L2"),
                   "global_batch": world * a.bs, "clip": [3, 8, 224, 224], "parallelism": "dp%d" % world,
                   "epoch": a.epoch, "thresh_epoch": 11, "executed_gflop_per_step_per_gpu": step_exec / 1e9,
                   "inputs": "resident in HBM before the timed region (`staged` = the same step with per-step input staging inside it)"},
        "loss": last,
        "staged": staged,
        "roofline": roof,
        "roofline_winograd": roof_wino,
        "roofline_step": roof_step,
        "ranks_observed": ranks_observed,
        "reducer": None if reducer is None else {"buckets": len(reducer.buckets), "backend": torch.distributed.get_backend(),
                                                 "forced_single_rank": forced},
    }
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or forced:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
