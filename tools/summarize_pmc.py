#!/usr/bin/env python3
"""rocprofv3 evidence -> the per-kernel tables kept under profiles/ (HBM GB/s for the HBM-bound kernels, MFMA utilisation for the
GEMM kernels) and the per-launch HBM traffic figure bench.py reports.

    export TMPDIR=/tmp
    B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_X -o X -- $B            # durations (own run)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- $B      # separate passes:
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- $B      # TCC has 4 slots, FETCH_SIZE
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \\
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o sq -- $B
    python tools/summarize_pmc.py --stats gpurun_out/prof_X/X_kernel_stats.csv --fetch gpurun_out/pmc_f/f_counter_collection.csv \\
        --write gpurun_out/pmc_w/w_counter_collection.csv --sq gpurun_out/pmc_sq/sq_counter_collection.csv --steps 3 --tag r02

(steps = launches of adam_kernel in the traced run = warm-up + timed steps.)  Writes profiles/<tag>_traffic.json (read by
bench.py), profiles/<tag>_hbm_table.md and profiles/<tag>_pmc_sq_gemm.csv.

gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies the 128-byte requests of 16-B-per-lane
streams as 64 B, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores and float atomics.  Both are in KiB.
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are counted, so this is fabric-side traffic, an upper bound on DRAM
traffic.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 shader engines (rocprofv3 sums a
counter over its instances)."""
import argparse
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK, HBM_ACHIEVABLE = 8.0e12, 6.3e12


def clean(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def load_counter(path, counters):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in counters:
            continue
        k = clean(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return per, {k: len(v) for k, v in cnt.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--sq")
    ap.add_argument("--mfma", help="counter CSV of a pass with SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16")
    ap.add_argument("--steps", type=int, required=True); ap.add_argument("--tag", default="r02")
    a = ap.parse_args()
    stats = {}
    for r in csv.DictReader(open(a.stats)):
        stats[clean(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
    fetch, nf = load_counter(a.fetch, {"FETCH_SIZE"})
    write, nw = load_counter(a.write, {"WRITE_SIZE"})
    out = {"note": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md; per launch, bench.py workload (bs=8); "
                   "offline rocprofv3 --pmc passes of the committed code, not a measurement of a bench run", "kernels": {}}
    conv = [0, 0.0]
    convx = [0, 0.0]
    rows = []
    for k in sorted(set(fetch) | set(write)):
        nl = max(nf.get(k, 0), nw.get(k, 0))
        b = (2.0 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024.0
        per_launch = b / max(nl, 1)
        out["kernels"][k] = {"launches": nl, "fetch_kib": fetch[k]["FETCH_SIZE"], "write_kib": write[k]["WRITE_SIZE"], "hbm_bytes_per_launch": per_launch}
        if k.startswith("conv_gemm"):
            conv[0] += nl; conv[1] += b
        if k.startswith("conv_x6_kernel"):
            convx[0] += nl; convx[1] += b
        if k in stats:
            calls, avg_ns = stats[k]
            rate = per_launch / (avg_ns * 1e-9)
            rows.append((calls * avg_ns / a.steps / 1e6, k, calls / a.steps, avg_ns / 1e3, per_launch / 1e6, rate / 1e12))
    out["conv_gemm_hbm_bytes_per_launch"] = conv[1] / max(conv[0], 1)
    out["conv_gemm_launches"] = conv[0]
    out["conv_x6_hbm_bytes_per_launch"] = convx[1] / max(convx[0], 1)        # the bf16-split conv / dgrad kernel (bench.py's `roofline.traffic` since round 4)
    out["conv_x6_launches"] = convx[0]
    json.dump(out, open(os.path.join(ROOT, "profiles", a.tag + "_traffic.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", a.tag + "_hbm_table.md"), "w") as f:
        f.write("# Per-kernel HBM traffic and rate (%s; bs = 8 bench workload; durations from the --stats run, bytes from the PMC passes)\n\n" % a.tag)
        f.write("HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction); fabric-side, Infinity-Cache hits included.  Peak 8 TB/s, achievable ~6.3 TB/s.\n\n")
        f.write("| kernel | ms / step | launches / step | avg us | MB / launch | TB/s | of 6.3 TB/s |\n|---|---|---|---|---|---|---|\n")
        for ms, k, n, us, mb, tbs in sorted(rows, reverse=True):
            f.write("| `%s` | %.3f | %.1f | %.1f | %.1f | %.2f | %.2f |\n" % (k, ms, n, us, mb, tbs, tbs * 1e12 / HBM_ACHIEVABLE))
    print("conv_gemm: %d launches, %.1f MB HBM per launch" % (conv[0], out["conv_gemm_hbm_bytes_per_launch"] / 1e6))
    print("conv_x6: %d launches, %.1f MB HBM per launch" % (convx[0], out["conv_x6_hbm_bytes_per_launch"] / 1e6))
    if a.mfma:
        # SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = fp32 MFMA FLOPs the hardware issued; the host-side walk of every launch's tiles
        # (pc_conv_work / pc_wino_work / pc_wgrad_work, what bench.py's roofline numerator is made of) must agree with it per kernel family
        import sys
        sys.path.insert(0, ROOT)
        import picons_amd  # noqa: F401
        from picons_amd import step as pstep
        from picons_amd.plan import Plan
        mf, _n = load_counter(a.mfma, {"SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE"})
        p = Plan(24, 224, n=8, groups=2, lanes=1)
        p.build_forward(); p.build_loss(pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)); p.build_backward(); p.build_adam()
        fams = {"conv_gemm": p.conv_flops_executed(), "conv_x6": p.x6_flops_executed(), "wino_conv": p.wino_flops_executed(), "wgrad": p.wgrad_flops_executed()}
        host = {k: sum(v["mfma"] for v in f.values()) for k, f in fams.items()}
        execd = {k: sum(v["executed"] for v in f.values()) for k, f in fams.items()}
        fam = lambda k: "wino_conv" if k.startswith("wino4_conv") else next((f for f in ("wino_conv", "conv_gemm", "conv_x6", "wgrad") if k.startswith(f)), None)
        # one MOPS count = 512 FLOPs of the counter's own dtype; a bf16-split kernel issues six bf16 products per fp32 product, so its
        # fp32-EQUIVALENT FLOPs (what the host books) are the bf16 count / 6.  The wgrad family holds kernels of both kinds.
        cnt = collections.defaultdict(float)
        for k, v in mf.items():
            if fam(k):
                cnt[fam(k)] += (v["SQ_INSTS_VALU_MFMA_MOPS_F32"] + v["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / 6.0) * 512.0 / a.steps
        with open(os.path.join(ROOT, "profiles", a.tag + "_mfma_counter_check.txt"), "w") as f:
            f.write("# fp32(-equivalent) MFMA FLOPs per step and kernel family: hardware counters (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 + SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / 6 --\n"
                    "# a bf16-split kernel issues six bf16 products per fp32 product --, rocprofv3 --pmc, own pass) against the host-side walk of every launch's tiles\n"
                    "# that bench.py's roofline numerators are made of (issued = whole tiles; executed = real rows x columns)\n")
            f.write("%-12s %16s %16s %10s %16s %10s\n" % ("family", "counter GFLOP", "host issued", "host/ctr", "host executed", "exec/ctr"))
            for k in ("conv_gemm", "conv_x6", "wino_conv", "wgrad"):
                c = cnt.get(k, 0.0)
                f.write("%-12s %16.2f %16.2f %10.4f %16.2f %10.4f\n" % (k, c / 1e9, host[k] / 1e9, host[k] / max(c, 1.0), execd[k] / 1e9, execd[k] / max(c, 1.0)))
                if c > 0:
                    assert abs(host[k] / c - 1.0) < 0.03, "host-issued FLOPs of %s disagree with the MFMA counter: %.4f" % (k, host[k] / c)
                    # executed / issued: whole-tile padding only -- the Winograd family pads most (28 of 32 tile slots per block in F(4x4,3x3)
                    # at 112 x 112 / 56 x 56, 49 of 64 in F(2x2,3x3) at 28 x 28, channel counts that are not multiples of 64)
                    lo = 0.75 if k == "wino_conv" else 0.85
                    assert execd[k] <= c * 1.0001 and execd[k] >= lo * c, "executed FLOPs of %s outside [%.2f, 1] x counter" % (k, lo)
        print(open(os.path.join(ROOT, "profiles", a.tag + "_mfma_counter_check.txt")).read())
    if a.sq:
        names = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
                 "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
        sq, nsq = load_counter(a.sq, set(names))
        with open(os.path.join(ROOT, "profiles", a.tag + "_pmc_sq_gemm.csv"), "w") as f:
            f.write("kernel,launches," + ",".join(names) + ",mfma_util,wait_any_frac,wait_inst_frac,active_inst_frac\n")
            for k in sorted(sq, key=lambda k: -sq[k]["SQ_BUSY_CYCLES"]):
                if not ("gemm" in k or "wgrad" in k or "em_" in k or "wino" in k or "conv_x6" in k):
                    continue
                v = sq[k]
                busy, wave = v["SQ_BUSY_CYCLES"] / 32.0, max(v["SQ_WAVE_CYCLES"], 1.0)
                f.write("%s,%d,%s,%.3f,%.3f,%.3f,%.3f\n" % (k.replace(",", " "), nsq[k], ",".join("%d" % v[n] for n in names),
                                                         v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / max(busy, 1.0), v["SQ_WAIT_ANY"] / wave,
                                                         v["SQ_WAIT_INST_ANY"] / wave, v["SQ_ACTIVE_INST_ANY"] / wave))
        print("SQ table written")


if __name__ == "__main__":
    main()
