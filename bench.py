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
`value` is the step as SURVEY.md 8(d) defines it, i.e. WITH the per-step input work of main_ucf101.py:52-79 inside the timed region
(every step's minibatch prepared from host uint8 frames by the device input pipeline and staged -- cat / shuffle -- into the arena, the
next step's samples on a side stream); `resident` / `value_resident` is the same step on a minibatch already in HBM, `dict_contract`
the same step fed the reference's float64 host dicts, `split_off` the step with PICONS_SPLIT=0 (all convolutions on fp32 MFMA).
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
    the MI355X way: the decoded uint8 frames of the step's 8 samples go up (1.8 MB each) and pc_clip_from_u8 writes data / aug_data /
    loc_msk of every sample straight into ITS PLACE (after cat + shuffle) of the next step's minibatch, in the layout the first conv reads
    ([2 bs][T][H][W][4]) -- a double buffer the stem's conv and weight gradient are pointed at (StepEngine.sample_stager) -- on a side stream
    while the current step runs.  (Rounds 2-3 stacked, concatenated, gathered and copied the samples into the arena and converted NCDHW ->
    NDHWC on the main stream at the head of the step: five passes over the clips.)"""

    def __init__(self, eng, bs, ncls, rank):
        from picons_amd import inputpipe, synthetic
        self.eng, self.bs, self.ip = eng, bs, inputpipe
        self.vids = [synthetic.make_decoded_video(100 + 97 * rank + i, labeled=(i % bs) < bs // 2, num_classes=ncls) for i in range(2 * bs)]
        self.st = eng.sample_stager()
        self.g = torch.Generator().manual_seed(1234 + rank)
        np.random.seed(1234 + rank)          # the loader's own np.random draws (crop offsets, frame choice)

    def prep(self, i, slot):
        n, nv = self.bs, len(self.vids)
        perm = torch.randperm(n, generator=self.g).numpy()
        drops = [(torch.rand(n, c, generator=self.g) < 0.5).float().numpy() * 2 for c in (832, 128, 832, 128)]
        self.st.prepare(slot, lambda j, out: self.ip.get_item(*self.vids[(n * i + j) % nv], train=True, out=out, ndhwc4=True), n // 2, perm, drops)

    def run(self, steps, epoch, ramp, reducer, lr, timed_kind=None, time_every=0, warm=0, at_start=None):
        """`warm` untimed steps flowing straight into `steps` timed ones, each on a fresh minibatch; returns (seconds, last losses, steps that
        carried kernel-timing events).  The clock starts behind a torch.cuda.synchronize() (and at_start(): the ranks' barrier) between the last
        warm step and the first timed one: the first timed step's minibatch was prepared under the last warm step, as every later one is under
        its predecessor -- no host-only phase (sample decoding, first uploads: tens of ms in which the chip clocks down) sits in front of the clock."""
        eng, st = self.eng, self.st
        self.prep(-warm, (-warm) & 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, ntimed = None, 0
        for i in range(-warm, steps):
            slot = i & 1
            if i == 0:
                torch.cuda.synchronize()
                if at_start is not None:
                    at_start()
                t0 = time.perf_counter()
            st.commit(slot)                   # the main stream waits for the slot; the clip conversion reads it in place
            tk = timed_kind if (i >= 0 and timed_kind is not None and time_every and i % time_every == 0) else None
            ntimed += tk is not None
            # a timed step of THIS leg is the ordinary four-lane step with event pairs riding in the dominant kernel's dispatches
            # (timed_on_lanes): it costs what every other step costs; the kernels' own durations come from the single-stream legs
            if reducer is not None and reducer.active:
                eng.arm_early_adam_dp(lr, reducer)
            else:
                eng.arm_early_adam(lr, on=True)
            eng.forward_backward(epoch, ramp, reducer, timed_kind=tk, timed_on_lanes=True)
            gscale = 1.0
            if reducer is not None:
                reducer.wait()
                gscale = reducer.gscale
            eng.adam(lr, gscale)
            st.release(slot)
            self.prep(i + 1, slot ^ 1)        # host decisions + uploads + pc_clip_from_u8 overlap the step enqueued above (on lane 1's stream: SampleStager.prepare)
            out = eng.read_scalars()
        torch.cuda.synchronize()
        eng._restore_input_ops()
        return time.perf_counter() - t0, out, ntimed


class DictInputs:
    """The reference's minibatch contract as the DataLoader hands it over (main_ucf101.py:52-79; datasets/ucf_dataloader.py:179-191:
    pageable float64 host tensors, 180 MB per bs-8 step) inside the timed region, through StepEngine.host_stager(): page-locked double
    buffer, uploads on a copy stream one step ahead, f64 -> f32 on the device."""

    def __init__(self, eng, bs, ncls, rank, nmb=3):
        from picons_amd import synthetic
        self.eng, self.st = eng, eng.host_stager()
        self.mbs = []
        for i in range(nmb):
            lab, unl, perm, drops = synthetic.make_step_inputs(bs, rank=rank, step=100 + i, num_classes=ncls)
            self.mbs.append(({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in lab.items()},
                             {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in unl.items()}, perm, drops))

    def run(self, steps, epoch, ramp, reducer, lr, warm=0, at_start=None):
        """As StagedInputs.run: `warm` untimed steps flow straight into the timed ones."""
        eng, st = self.eng, self.st
        nm = len(self.mbs)
        st.prepare((-warm) & 1, *self.mbs[(-warm) % nm])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, host_wait = None, 0.0
        for i in range(-warm, steps):
            slot = i & 1
            if i == 0:
                torch.cuda.synchronize()
                if at_start is not None:
                    at_start()
                t0, host_wait = time.perf_counter(), 0.0
            st.commit(slot)
            if reducer is not None and reducer.active:
                eng.arm_early_adam_dp(lr, reducer)
            else:
                eng.arm_early_adam(lr, on=True)
            eng.forward_backward(epoch, ramp, reducer)
            gscale = 1.0
            if reducer is not None:
                reducer.wait()
                gscale = reducer.gscale
            eng.adam(lr, gscale)
            st.release(slot)
            st.prepare(slot ^ 1, *self.mbs[(i + 1) % nm])      # the host's 180 MB gather + the uploads run under the step enqueued above
            tw = time.perf_counter()
            out = eng.read_scalars()
            host_wait += time.perf_counter() - tw
        torch.cuda.synchronize()
        eng._restore_input_ops()
        return time.perf_counter() - t0, out, host_wait


PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak; a bf16-split fp32 product costs six of them


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # >= 20 timed steps after >= 5 warm-up; 200 x 21 ms
                                                            # keeps the GPU busy long enough for an external sampler to see it
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--gv", action="store_true", help="BASELINE config 3 (--gv instead of --bv)")
    ap.add_argument("--jhmdb", action="store_true", help="BASELINE config 5's shape: 21 classes, main_jhmdb.py's step")
    ap.add_argument("--epoch", type=int, default=1, help="epoch the step runs at (>= 11 = --thresh_epoch: argmax pseudo-labels "
                    "for the unlabeled rows, capsules_ucf101.py:463)")
    ap.add_argument("--time-every", type=int, default=40, help="attach hipEvent pairs to the dominant conv kernel's launches of every n-th timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--resident-inputs", action="store_true", help="time the step with the minibatch resident in HBM ONLY (no per-step input staging: "
                    "`value` is then the resident figure and says so)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the dict_contract / split_off legs and the busy phase")
    ap.add_argument("--leg-steps", type=int, default=0, help="timed steps of the secondary legs (default: min(steps, 50))")
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
    t_gpu0 = time.perf_counter()

    args = pstep.default_args(bv=not a.gv, gv=a.gv, n_frames=5, wt_cons=0.1, lr=1e-4, epochs=100, thresh_epoch=11)
    ncls = 21 if a.jhmdb else 24
    eng = pstep.StepEngine(args, bs=a.bs, hw=224, num_classes=ncls, jhmdb=a.jhmdb, device=dev)
    split_on = bool(eng.plan.x6)
    lab, unl, perm, drops = synthetic.make_step_inputs(a.bs, rank=rank, step=0, num_classes=ncls)
    eng.stage(lab, unl, perm, drops)
    # PICONS_FORCE_REDUCER=1 on one GPU: a one-rank RCCL group and the whole DP schedule (segmented backward, bucket all-reduces on
    # the comm stream behind events, 1/world in Adam) -- the N > 1 code path executed through RCCL where only one GPU exists
    forced = world == 1 and os.environ.get("PICONS_FORCE_REDUCER", "0") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    reducer = eng.make_reducer(force=forced) if (world > 1 or forced) else None
    if reducer is not None:
        reducer.diagnostics = True            # comm_wait_ms / exposed_ms of the `reducer` block (off in a training loop)
    ramp = pstep.exp_rampup(100)(a.epoch)
    main_kind = None if a.no_kernel_timing else (capi.OP_CONV_X6 if split_on else capi.OP_CONV)
    ranks_observed = 1
    if world > 1 or forced:                              # an all-reduce of ones: the rank count the collective really spans
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        ranks_observed = int(ones.item())
    if ranks_observed != world:
        print("bench.py: the collective spans %d ranks, WORLD_SIZE is %d" % (ranks_observed, world), file=sys.stderr)
        sys.exit(4)
    staged_ok = not a.resident_inputs and a.bs % 2 == 0 and not a.jhmdb

    def timed_resident(e, steps, kind=None, every=0, red=None):
        """`steps` steps on the minibatch resident in HBM, bracketed as the contract says; -> (ms over all ranks' max, last losses, timed steps)."""
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        last, nt = None, 0
        for it in range(steps):
            tk = kind if (kind is not None and every and it % every == 0) else None
            nt += tk is not None
            last = e.run_staged(a.epoch, ramp, reducer=red, timed_kind=tk, collect=False, timed_on_lanes=True)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        return pdist.barrier_max_ms((time.perf_counter() - t0) * 1e3, device=dev), last, nt

    # ---- warm-up, then the HEADLINE leg: exactly `steps` steps between barriers + synchronize.  The step is SURVEY 8(d)'s: zero_grad ->
    # a1 .. Adam -> loss read-back, and a1 starts with the per-step input work (main_ucf101.py:52-79), so the minibatch is staged inside
    # the timed region (fresh samples every step from host uint8 frames through the device input pipeline, the next step's on a side stream)
    for _ in range(a.warmup):
        eng.run_staged(a.epoch, ramp, reducer=reducer)
    eng.kind_ms, eng.kind_count = 0.0, 0
    timed_steps = 0
    if staged_ok:
        si = StagedInputs(eng, a.bs, ncls, rank)

        def headline_start():              # between the last warm step and the first timed one, behind torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            if reducer is not None:
                reducer.stats()            # reset: the diagnostics below cover the headline leg only
        sec, last, timed_steps = si.run(a.steps, a.epoch, ramp, reducer, args.lr, timed_kind=main_kind, time_every=max(1, a.time_every), warm=3,
                                        at_start=headline_start)
        if world > 1:
            torch.distributed.barrier()
        ms_total = pdist.barrier_max_ms(sec * 1e3, device=dev)
        inputs_note = ("staged inside the timed region: every step prepares a fresh minibatch from host uint8 frames (upload of the 8 selected frames per sample, "
                       "pc_clip_from_u8, cat + shuffle into the arena; the next step's samples on a side stream) -- main_ucf101.py:52-79 is part of the step")
    else:
        if reducer is not None:
            torch.cuda.synchronize()
            reducer.stats()
        ms_total, last, timed_steps = timed_resident(eng, a.steps, main_kind, max(1, a.time_every), reducer)
        inputs_note = "resident in HBM before the timed region (--resident-inputs / --jhmdb: no per-step input staging in this run)"
    if main_kind is not None:
        eng.collect_timing()
    # per-step diagnostics of the headline leg's collectives (host time in reducer.wait(), device time the main stream sat behind its last
    # backward kernel until the last collective finished): what a SCALE line is explained with
    reducer_stats = {}
    if reducer is not None:
        torch.cuda.synchronize()
        reducer_stats = reducer.stats()
    ms_step = ms_total / a.steps
    value = world * a.bs * a.steps / (ms_total / 1e3)
    main_ms, main_count = eng.kind_ms, eng.kind_count

    n_leg = a.leg_steps or min(a.steps, 50)
    # ---- the same step on the minibatch already resident in HBM (what rounds 1-3 quoted as `value`)
    resident = None
    if staged_ok:
        eng.stage(lab, unl, perm, drops)
        for _ in range(3):                     # back on the arena path: untimed steps first, as in front of every other leg
            eng.run_staged(a.epoch, ramp, reducer=reducer)
        ms_r, last_r, _ = timed_resident(eng, n_leg, red=reducer)
        resident = {"value": world * a.bs * n_leg / (ms_r / 1e3), "unit": "clips/s", "ms_per_step": ms_r / n_leg, "steps": n_leg, "loss_total": last_r["total"]}

    # ---- every GEMM family's kernels get their event pairs in single-stream replays of their own (one op kind per timed replay): one warm
    # replay that carries events and is thrown away, then `reps` measured ones.  A leg whose replays disagree by more than 1.5x is reported
    # as null with the reason (one driver-side line carried a 4.8x outlier for one leg).
    def kind_leg(kind, reps=3):
        eng.run_staged(a.epoch, ramp, reducer=reducer, timed_kind=kind, collect=False)
        torch.cuda.synchronize()
        eng.collect_timing()
        per = []
        for _ in range(reps):
            eng.kind_ms, eng.kind_count = 0.0, 0
            eng.run_staged(a.epoch, ramp, reducer=reducer, timed_kind=kind, collect=False)
            torch.cuda.synchronize()
            eng.collect_timing()
            per.append((eng.kind_ms, eng.kind_count))
        return per

    def leg_summary(per, also_ms=None):
        """-> (total ms, total launches, replays, None) or (0, 0, 0, reason)."""
        if not per or not per[0][1]:
            return 0.0, 0, 0, "no launch of this kind in the step"
        ms = [m for m, _c in per] + ([also_ms] if also_ms else [])
        if len({c for _m, c in per}) != 1:
            return 0.0, 0, 0, "replays timed different launch counts: %s" % [c for _m, c in per]
        if max(ms) > 1.5 * min(ms):
            return 0.0, 0, 0, "replays disagree by more than 1.5x (kernel ms per step: %s)" % ["%.3f" % m for m in ms]
        return sum(m for m, _c in per), sum(c for _m, c in per), len(per), None
    legs = {}
    if main_kind is not None:
        eng.stage(lab, unl, perm, drops)
        legs["main"] = leg_summary(kind_leg(main_kind))
        legs["wino"] = leg_summary(kind_leg(capi.OP_WINO_CONV))
        if split_on:
            legs["f32c"] = leg_summary(kind_leg(capi.OP_CONV))
        # the weight gradients (round 6; the largest family of the step): the launches on the bf16 matrix cores and the fp32-MFMA ones apart
        # (pc_run_ops_timed's sub-filter: kind | 1 << 16 / kind | 2 << 16), event pairs in their own dispatches as for the conv kernels
        legs["wgx6"] = leg_summary(kind_leg(capi.OP_WGRAD | (1 << 16)))
        legs["wgf32"] = leg_summary(kind_leg(capi.OP_WGRAD | (2 << 16)))

    # ---- the reference's own minibatch contract inside the timed region: float64 host dicts -> pinned double buffer -> copy stream
    dict_leg = None
    if staged_ok and not a.no_extra_legs:
        di = DictInputs(eng, a.bs, ncls, rank)
        sec, last_d, host_wait = di.run(n_leg, a.epoch, ramp, reducer, args.lr, warm=3, at_start=(torch.distributed.barrier if world > 1 else None))
        ms_d = pdist.barrier_max_ms(sec * 1e3, device=dev)
        dict_leg = {"value": world * a.bs * n_leg / (ms_d / 1e3), "unit": "clips/s", "ms_per_step": ms_d / n_leg, "steps": n_leg, "loss_total": last_d["total"],
                    "host_wait_ms_per_step": host_wait * 1e3 / n_leg,
                    "what": "the reference's minibatch contract (main_ucf101.py:52-79, datasets/ucf_dataloader.py:179-191): two dicts of pageable float64 host tensors "
                            "per step (180 MB), gathered in shuffled order into a page-locked double buffer, uploaded on a copy stream one step ahead, "
                            "f64 -> f32 on the device (pc_ncdhw_to_ndhwc reads the float64 staging); host_wait = time the host spends in read_scalars()"}

    # ---- the same engine with every convolution on the fp32 MFMA kernels (PICONS_SPLIT=0), minibatch resident: what the bf16-split conv kernel buys
    split_off = None
    if split_on and not a.no_extra_legs and reducer is None and rank == 0:
        # a child process of its own: a second engine in THIS process would find the four hardware queues taken by the first one's lanes and
        # run 3 - 5 % slower for that reason alone; this process is idle meanwhile
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(n_leg), "--warmup", "3", "--bs", str(a.bs), "--epoch", str(a.epoch),
               "--resident-inputs", "--no-extra-legs", "--no-kernel-timing", "--no-cpu-baseline"] + (["--gv"] if a.gv else []) + (["--jhmdb"] if a.jhmdb else [])
        try:
            cp = subprocess.run(cmd, env=dict(os.environ, PICONS_SPLIT="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
            j0 = json.loads([ln for ln in cp.stdout.splitlines() if ln.strip()][-1])
            split_off = {"value": j0["value"], "unit": "clips/s", "ms_per_step": j0["ms_per_step"], "steps": j0["steps"], "loss_total": j0["loss"]["total"],
                         "what": "PICONS_SPLIT=0 in a process of its own: every conv / dgrad launch on v_mfma_f32_32x32x2_f32 (the round-3 kernels), minibatch "
                                 "resident in HBM -- compare with `resident`"}
        except Exception as e:          # the leg is evidence, not the metric: say what happened and go on
            split_off = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- keep the GPU visibly busy for an external sampler: at least 3 s of back-to-back steps in this process, outside every timed region
    busy_steps = 0
    if not a.no_extra_legs:
        eng.stage(lab, unl, perm, drops)
        t_busy0 = time.perf_counter()
        while time.perf_counter() - t_busy0 < 3.0 or busy_steps < 20:      # >= 3 s of GPU work measured from the phase's own start
            eng.run_staged(a.epoch, ramp, reducer=reducer)
            busy_steps += 1
            if busy_steps >= 600:
                break
        torch.cuda.synchronize()

    pl = eng.plan
    fl = pl.flops()                                      # FLOPs of the emitted (tap-trimmed) descriptors, real channel counts
    fe = pl.conv_flops_executed()                        # fp32-MFMA conv / dgrad FLOPs as the kernels run them (host walk of every launch's tiles)
    fx = pl.x6_flops_executed()                          # the same for the launches on the bf16-split kernel
    fw = pl.wgrad_flops_executed()
    fz = pl.wino_flops_executed()                        # the Winograd launches: transform-domain FLOPs (2.25x fewer than the direct form's)
    fr = pl.flops_reference_counted()
    lists = ("fwd", "bwd")
    tot = lambda f, k: sum(f[n][k] for n in lists)
    conv_exec, x6_exec, wino_exec, wg_exec = tot(fe, "executed"), tot(fx, "executed"), tot(fz, "executed"), tot(fw, "executed")
    step_exec = conv_exec + x6_exec + wino_exec + wg_exec
    # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # their own runs, tools/summarize_pmc.py); counters cannot be read from inside a run, so this is OFFLINE data from the named file
    traffic_of, traffic_src = {}, None
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tp = os.path.join(ROOT, "profiles", tag + "_traffic.json")
        if not traffic_of and os.path.exists(tp) and a.bs == 8:
            try:
                tj = json.load(open(tp))
                fam_avg = lambda pred: (lambda ws: (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ws) / max(sum(v["launches"] for v in ws), 1)) if ws else None)(
                    [v for k, v in tj.get("kernels", {}).items() if pred(k)])
                traffic_of = {"x6": tj.get("conv_x6_hbm_bytes_per_launch"), "f32": tj.get("conv_gemm_hbm_bytes_per_launch"),
                              "wgx6": fam_avg(lambda k: "wgrad3_x6_kernel" in k or "wgrad_x6_kernel" in k or "wgrad4_x6_kernel" in k),
                              "wgf32": fam_avg(lambda k: ("wgrad4_kernel" in k or "wgrad3_kernel" in k or "wgrad_kernel" in k) and "_x6" not in k),
                              "wino": (lambda ws: (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ws) / max(sum(v["launches"] for v in ws), 1)) if ws else None)(
                                  [v for k, v in tj.get("kernels", {}).items() if "wino_conv_kernel" in k or "wino4_conv_kernel" in k])}
                traffic_src = "profiles/%s_traffic.json (offline rocprofv3 --pmc passes, not this run)" % tag
            except Exception:
                traffic_of = {}
    traffic = traffic_of.get("x6" if split_on else "f32")
    def roof(kernel, exec_f, issued_f, valid_f, leg, peak, extra=None):
        kms, count, steps_t, why = leg if leg is not None else (0.0, 0, 0, "leg not run")
        if why is not None or not count or not steps_t:
            return {"bound": "mfma", "kernel": kernel, "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None, "invalid": why}
        ms_s = kms / steps_t
        tf = lambda f: f / (ms_s * 1e-3) / 1e12
        r = {"bound": "mfma", "kernel": kernel, "achieved": tf(exec_f), "peak": peak, "unit": "TFLOP/s", "frac": tf(exec_f) / peak,
             "frac_mfma_issued": tf(issued_f) / peak, "frac_valid": tf(valid_f) / peak,
             "launches_per_step": count // steps_t, "avg_launch_ms": kms / count, "kernel_ms_per_step": ms_s, "timed_steps": steps_t,
             "flops_per_launch": exec_f * steps_t / count, "traffic": None,
             "timing": "hipEvent pair in every launch's own dispatch, single-stream replays of the step (one warm replay discarded, %d measured; replays within 1.5x)" % steps_t,
             "flops_counted": "executed: per launch the host walks the kernel's tiles (pc_conv_work) and counts real rows x real columns x the K each block "
                              "walks -- taps that are padding for a whole tile are skipped by the kernel and NOT counted"}
        r.update(extra or {})
        return r
    roof_x6 = roof_f32conv = roof_wino = None
    if main_kind is not None and split_on:
        roof_x6 = roof("conv_x6_kernel (fp32 conv / dgrad / convT multiplied on the bf16 matrix cores: 3 bf16 terms per operand, 6 products on "
                       "v_mfma_f32_32x32x16_bf16, fp32 accumulate)", x6_exec, tot(fx, "mfma"), tot(fx, "valid"), legs.get("main"),
                       PEAK_BF16_MFMA_TFLOPS / 6.0,
                       {"peak_note": "dense bf16 MFMA peak / 6 products per fp32 multiply-accumulate = %.1f TFLOP/s of fp32-equivalent work; the chip holds "
                                     "1.35 - 1.5 GHz of its 2.4 GHz in this kernel (in-kernel clock stamps, profiles/r04_x6_tile_probe.txt)" % (PEAK_BF16_MFMA_TFLOPS / 6.0),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "in_region_kernel_ms_per_step": (main_ms / timed_steps) if timed_steps else None, "in_region_timed_steps": timed_steps,
                        "in_region_note": "hipEvent pairs in the same kernel's dispatches during the headline leg's timed steps (the ordinary four-lane step: durations "
                                          "as stretched by the kernels running beside them on the other lanes); the fractions above come from the single-stream replays"})
        roof_f32conv = roof("conv_gemm_glds_kernel / conv_gemm_kernel (fp32 MFMA gather-GEMM: the launches that stay on v_mfma_f32_32x32x2_f32 -- the RGB stem, "
                            "channel counts that are not multiples of 32, launches too small to gain)", conv_exec, tot(fe, "mfma"), tot(fe, "valid"), legs.get("f32c"),
                            PEAK_FP32_MFMA_TFLOPS, {"traffic": traffic_of.get("f32"), "traffic_source": traffic_src})
        if roof_x6.get("achieved"):
            roof_x6["frac_of_fp32_mfma_peak"] = roof_x6["achieved"] / PEAK_FP32_MFMA_TFLOPS
    elif main_kind is not None:
        roof_f32conv = roof("conv_gemm_glds_kernel / conv_gemm_kernel (fp32 MFMA gather-GEMM: conv fwd / dgrad / convT)", conv_exec, tot(fe, "mfma"), tot(fe, "valid"),
                            legs.get("main"), PEAK_FP32_MFMA_TFLOPS, {"traffic": traffic, "traffic_source": traffic_src})
    if main_kind is not None:
        fzd = sum(pl.flops_reference_counted_wino()[n] for n in lists)
        roof_wino = roof("wino4_conv_kernel + wino_conv_kernel (Winograd F(4x4,3x3) on five 112x112 / 56x56 launches, F(2x2,3x3) on the 28x28 layers and the trunk forward at 56x56: conv / input gradient, fp32 MFMA in the transform domain)", wino_exec, tot(fz, "mfma"), wino_exec,
                         legs.get("wino"), PEAK_FP32_MFMA_TFLOPS,
                         {"flops_counted": "executed transform-domain FLOPs on real tiles / channels (pc_wino_work)",
                          "traffic": traffic_of.get("wino"), "traffic_source": traffic_src})
        if roof_wino.get("kernel_ms_per_step"):
            roof_wino["direct_equivalent_tflops"] = fzd / (roof_wino["kernel_ms_per_step"] * 1e-3) / 1e12      # the 3x3x3 formulation's FLOPs over the same time
    roof_wgx6 = roof_wgf32 = None
    if main_kind is not None:
        wx, wf = pl.wgrad_flops_by_family()
        roof_wgx6 = roof("wgrad3_x6_kernel + wgrad_x6_kernel + wgrad4_x6_kernel (weight gradients of the 3x3x3 / 1x1x1 / transposed layers and the RGB stem on the bf16 matrix "
                         "cores: both operands split into 3 bf16 terms in registers, 6 products, hi / lo fp32 accumulators; K slices summed in slice order by the gradient "
                         "re-layout, no atomics)",
                         wx["executed"], wx["mfma"], wx["valid"], legs.get("wgx6"), PEAK_BF16_MFMA_TFLOPS / 6.0,
                         {"flops_counted": "executed: real rows x real columns x the positions each K slice walks (pc_wgrad_work)",
                          "peak_note": "dense bf16 MFMA peak / 6 products per fp32 multiply-accumulate", "traffic": traffic_of.get("wgx6"), "traffic_source": traffic_src})
        roof_wgf32 = roof("wgrad3_kernel<.., 9> (fp32 MFMA weight gradients: the 9-tap spectral PrimaryCaps / upsample1 planes -- 20-position rows, a k16 step would be "
                          "37 % padding; with PICONS_SPLIT=0 also wgrad4_kernel / wgrad3_kernel / wgrad_kernel for every other layer)", wf["executed"], wf["mfma"], wf["valid"], legs.get("wgf32"), PEAK_FP32_MFMA_TFLOPS,
                          {"flops_counted": "executed (pc_wgrad_work)", "traffic": traffic_of.get("wgf32"), "traffic_source": traffic_src})
    # `roofline` = the GEMM family with the LARGEST single-stream kernel time per step (the dominant kernel) among ALL FIVE families; the others keep their own blocks
    fam = [r for r in (roof_x6, roof_f32conv, roof_wino, roof_wgx6, roof_wgf32) if r is not None and r.get("kernel_ms_per_step")]
    roofline = max(fam, key=lambda r: r["kernel_ms_per_step"]) if fam else None
    if roofline is not None:
        roofline = dict(roofline, dominant_by="largest single-stream kernel time per step among the five GEMM families (%s)"
                        % ", ".join("%s %.2f ms" % (r["kernel"].split(" ")[0], r["kernel_ms_per_step"]) for r in fam))
    ms_for_step = resident["ms_per_step"] if resident else ms_step
    roof_step = {"executed_gflop_per_step": step_exec / 1e9, "achieved": step_exec / (ms_step * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                 "unit": "TFLOP/s", "frac": step_exec / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                 "descriptor_gflop_per_step": sum(fl[n] for n in lists) / 1e9, "reference_formulation_conv_gflop_per_step": sum(fr[n] for n in lists) / 1e9,
                 "gflop_by_family": {"conv_bf16_split": x6_exec / 1e9, "conv_fp32_mfma": conv_exec / 1e9, "winograd_conv": wino_exec / 1e9, "wgrad": wg_exec / 1e9},
                 "note": "whole step (all kernels, host gaps, input staging and the loss read-back included) against the fp32 MFMA roof, executed GEMM FLOPs in fp32-equivalent "
                         "multiply-accumulates (pc_conv_work + pc_wino_work + pc_wgrad_work); the launches on the bf16-split kernel have a higher roof (`roofline.peak`), "
                         "so this fraction can exceed what an all-fp32-MFMA step could reach; resident-input step: %.3f ms" % ms_for_step}
    wg_split = split_on and os.environ.get("PICONS_SPLIT_WGRAD", "1") != "0"
    n_x6_ops = sum(1 for n in lists for op in pl.lists[n] if op[0] == capi.OP_CONV_X6)
    n_f32_ops = sum(1 for n in lists for op in pl.lists[n] if op[0] == capi.OP_CONV)
    dtype = ("f32 (fp32 operands and fp32 accumulation everywhere; %d of the %d conv / input-gradient launches%s multiply on the bf16 matrix cores: every fp32 operand "
             "as the exact sum of 3 bf16 terms, 6 products on v_mfma_f32_32x32x16_bf16, hi / lo fp32 accumulators -- error vs fp64 <= the fp32-MFMA kernel's on the same "
             "launch, tests/test_x6_gpu.py; the trunk's forward convs, the Winograd layers, the stem, the spectral-plane weight gradient and every non-GEMM kernel: "
             "fp32 MFMA / VALU)" % (n_x6_ops, n_x6_ops + n_f32_ops, " and the 3x3x3 / 1x1x1 weight gradients" if wg_split else "")) if split_on else "f32"
    out = {
        "metric": METRIC,
        "value": value, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": ("configs[4] on one rank: JHMDB-21-shaped" if a.jhmdb else "configs[1]: UCF101-24-shaped") + " synthetic, I3D+caps, 8 frames (16-frame span, stride 2) x224x224, "
                               "bs=%d/GPU (bs/2 labeled + bs/2 unlabeled), %s consistency, dice+BCE loc loss, spread cls loss, Adam"
                               % (a.bs, "--gv" if a.gv else "--bv --n_frames 5 L2"),
                   "global_batch": world * a.bs, "clip": [3, 8, 224, 224], "parallelism": "dp%d" % world,
                   "epoch": a.epoch, "thresh_epoch": 11, "executed_gflop_per_step_per_gpu": step_exec / 1e9, "inputs": inputs_note,
                   "bf16_split_conv": split_on,
                   # which Winograd form each launch takes (a line must say what arithmetic ran) and whether any experiment switch was in force
                   "winograd_launches": {"F(4x4,3x3)": sum(1 for n in lists for op in pl.lists[n] if op[0] == capi.OP_WINO_CONV and op[1][15] == 4),
                                         "F(2x2,3x3)": sum(1 for n in lists for op in pl.lists[n] if op[0] == capi.OP_WINO_CONV and op[1][15] != 4)},
                   "weight_gradients": "K-slice images added in slice order (no atomics)" if pl.wg_ordered else "fp32 atomics between K slices (PICONS_WGRAD_ATOMIC=1)",
                   "experiment_switches": sorted(k for k in __import__("picons_amd.switches", fromlist=["x"]).EXPERIMENTS
                                                 if os.environ.get(k) and os.environ.get("PICONS_DIAG_LIB", "0") not in ("", "0"))},
        "loss": last,
        "value_resident": None if resident is None else resident["value"],
        "resident": resident,
        "dict_contract": dict_leg,
        "split_off": split_off,
        "roofline": roofline,
        "roofline_conv_x6": roof_x6,
        "roofline_fp32_conv": roof_f32conv,
        "roofline_winograd": roof_wino,
        "roofline_wgrad_x6": roof_wgx6,
        "roofline_wgrad_fp32": roof_wgf32,
        "roofline_step": roof_step,
        "ranks_observed": ranks_observed,
        "busy_steps_outside_timed_regions": busy_steps,
        "reducer": None if reducer is None else dict(reducer_stats, buckets=len(reducer.buckets), backend=torch.distributed.get_backend(),
                                                     forced_single_rank=forced, bucket_table=reducer.bucket_table()),
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
