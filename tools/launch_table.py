#!/usr/bin/env python3
"""Per-launch GEMM efficiency table: pairs the conv / wgrad ops of a single-lane plan (bench.py workload)
with the kernels of one step of a `rocprofv3 --kernel-trace` CSV taken with PICONS_LANES=1, and prints
issued TFLOP/s and the time each launch loses against its matrix-core peak: 157.3 TF/s for the fp32-MFMA kernels, 2 500 / 6 = 416.7 TF/s of
fp32-equivalent FLOPs for the bf16-split kernels (`*_x6_kernel`: six bf16 MFMA products per fp32 product).

    python tools/launch_table.py gpurun_out/prof_l1/l1_kernel_trace.csv [top_n]
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import picons_amd  # noqa: F401,E402
from picons_amd import capi, desc as D, step as pstep  # noqa: E402
from picons_amd.plan import Plan  # noqa: E402

PEAK = 157.3e9      # FLOP per ms, fp32 MFMA
PEAK_X6 = 2500e9 / 6  # FLOP per ms of fp32-equivalent work on the bf16 matrix cores, six products per fp32 product


def peak_of(kn):
    return PEAK_X6 if "_x6_kernel" in kn else PEAK

VEC = ("ostr", "ooff", "istr", "ntap", "ioff0", "istep", "wk0", "wkstep", "doff")


def unflat(i, fields):
    d, k = {}, 0
    for f in fields:
        if f in VEC:
            d[f] = i[k:k + 3]; k += 3
        else:
            d[f] = i[k]; k += 1
    return d


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    p = Plan(24, 224, n=8, groups=2, lanes=1)
    p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam()
    p.finalize()
    ops = [(name, op) for name in ("prep", "fwd", "loss", "bwd") for op in p.lists[name] if op[0] in (capi.OP_CONV, capi.OP_CONV_X6, capi.OP_WGRAD, capi.OP_WINO_CONV)]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    st = [r for r in rows[ad[-2] + 1:ad[-1] + 1] if any(k in r["Kernel_Name"] for k in ("conv_gemm", "conv_x6_kernel", "wgrad_kernel", "wgrad3_kernel", "wgrad4_kernel", "wgrad4_x6_kernel", "wgrad_x6_kernel", "wgrad3_x6_kernel", "wino_conv_kernel", "wino4_conv_kernel"))]
    def n_kernels(op):
        """pc_conv_wgrad gives a <=64-channel remainder of a deep grid its own 64-row-tile launch (pc_wgrad_work's launch count)."""
        return p.op_work[id(op[1])]["launches"] if op[0] == capi.OP_WGRAD else 1
    if len(st) != sum(n_kernels(op) for _n, op in ops):
        raise SystemExit("trace has %d GEMM launches per step, plan expects %d (lanes / version mismatch?)" %
                         (len(st), sum(n_kernels(op) for _n, op in ops)))
    out = []
    it = iter(st)
    for name, op in ops:
        rs = [next(it) for _ in range(n_kernels(op))]
        r = rs[0]
        dur = sum((int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e6 for q in rs)
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        blocks = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_X"])
        w = p.op_work[id(op[1])]          # host-side walk of the launch's tiles (pc_conv_work / pc_wgrad_work): MACs as the kernel runs them
        if op[0] == capi.OP_WINO_CONV:
            if "wino" not in kn:
                raise SystemExit("order mismatch: winograd op paired with " + kn)
            N_, T_, H_, W_, Ci_, _l, Co_ = op[1][:7]
            if (op[1][15] == 4) != ("wino4" in kn):
                raise SystemExit("order mismatch: winograd op with m = %d paired with %s" % (op[1][15], kn))
            what = "wino  M=%-7d Co=%-5d Ci=%-4d taps=3x(%s transform domain)" % (N_ * T_ * H_ * W_, Co_, Ci_, "6x6" if op[1][15] == 4 else "4x4")
        elif op[0] in (capi.OP_CONV, capi.OP_CONV_X6):
            if (op[0] == capi.OP_CONV_X6) != ("conv_x6" in kn):
                raise SystemExit("order mismatch: %s op paired with %s" % ("bf16-split" if op[0] == capi.OP_CONV_X6 else "fp32", kn))
            if "wgrad" in kn:
                raise SystemExit("order mismatch: conv op paired with " + kn)
            d = unflat(op[1], D.CONV_FIELDS)
            M = d["N"] * d["Tq"] * d["Hq"] * d["Wq"]
            what = "conv  M=%-7d Co=%-5d Ci=%-4d taps=%s" % (M, d["Co"], d["Ci"], "x".join(map(str, d["ntap"])))
        else:
            if "wgrad" not in kn:
                raise SystemExit("order mismatch: wgrad op paired with " + kn)
            d = unflat(op[1], D.WGRAD_FIELDS)
            M = d["N"] * d["Tq"] * d["Hq"] * d["Wq"]
            what = "wgrad K=%-7d Cd=%-5d Cs=%-4d taps=%s" % (M, d["Cd"], d["Cs"], "x".join(map(str, d["ntap"])))
        out.append((dur, 2 * w["executed"], 2 * w["issued"], 2 * w["valid"], name, kn, blocks, what))
    print("FLOPs per launch as the kernel runs it (host walk of its tiles, pc_conv_work / pc_wgrad_work): TF/s = executed (real rows x columns over the "
          "K each block walks; taps that are padding for a whole tile are skipped by the kernel and not counted), mfma = issued to the matrix "
          "cores (whole tiles), valid = non-padding MACs only; lost = time above the matrix-core peak time of the executed FLOPs (157.3 TF/s fp32 MFMA; 416.7 for *_x6_kernel)")
    print("%7s %7s %7s %7s %7s %-4s %6s %-34s %s" % ("ms", "TF/s", "mfma", "valid", "lost", "list", "blocks", "kernel", "launch (trimmed descriptor)"))
    for dur, fl, fm, fv, name, kn, blocks, what in sorted(out, key=lambda x: -(x[0] - x[1] / peak_of(x[5])))[:top]:
        print("%7.3f %7.1f %7.1f %7.1f %7.3f %-4s %6d %-34s %s" % (dur, fl / dur / 1e9, fm / dur / 1e9, fv / dur / 1e9, dur - fl / peak_of(kn), name, blocks, kn[:34], what))
    over = [x for x in out if x[2] / x[0] > peak_of(x[5])]
    if over:
        raise SystemExit("%d launches above their matrix-core peak even by issued MFMA FLOPs: accounting error" % len(over))
    tot = sum(x[0] for x in out)
    for label, sel in (("conv / dgrad fp32", lambda x: "conv_gemm" in x[5]), ("conv / dgrad bf16-split", lambda x: "conv_x6" in x[5]), ("winograd conv", lambda x: "wino" in x[5]),
                       ("weight grad fp32", lambda x: "wgrad" in x[5] and "_x6" not in x[5]), ("weight grad bf16-split", lambda x: "wgrad" in x[5] and "_x6" in x[5])):
        xs = [x for x in out if sel(x)]
        if not xs:
            continue
        t = sum(x[0] for x in xs)
        pk = peak_of(xs[0][5])
        print("%-24s %3d launches %6.2f ms/step: executed %.1f GF = %.1f TF/s (%.3f of peak), mfma-issued %.1f TF/s (%.3f), valid %.1f TF/s (%.3f)" %
              (label, len(xs), t, sum(x[1] for x in xs) / 1e9, sum(x[1] for x in xs) / t / 1e9, sum(x[1] for x in xs) / t / pk,
               sum(x[2] for x in xs) / t / 1e9, sum(x[2] for x in xs) / t / pk, sum(x[3] for x in xs) / t / 1e9, sum(x[3] for x in xs) / t / pk))
    print("GEMM launches: %d, %.2f ms/step, %.2f ms above the matrix-core peak time of the executed FLOPs (fp32 MFMA 157.3 TF/s; bf16-split 416.7)" %
          (len(out), tot, sum(x[0] - x[1] / peak_of(x[5]) for x in out)))


if __name__ == "__main__":
    main()
