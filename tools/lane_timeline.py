#!/usr/bin/env python3
"""Where is the chip under-filled during a multi-lane step?  Reads a rocprofv3 --kernel-trace csv of `bench.py` (four lanes),
takes one full step -- from the first kernel behind the previous step's last adam_kernel (the weight-layout prep in front of the clip's
to_ndhwc_kernel pair) to the step's own last adam_kernel; with the early Adam a step has TWO adam_kernel dispatches, so the windows are
found from the to_ndhwc pairs, not from the optimiser -- and prints

  * per stream: kernels, busy time, share of the step;
  * the step's wall time split by how many CU slots the kernels running at that instant could fill at most
    (sum over concurrent kernels of min(1, blocks / (256 CUs x resident blocks per CU)), resident blocks from the
    dispatch's LDS bytes / VGPRs / workgroup size);
  * the kernels that run while that sum is below --thresh, by accumulated under-filled time.

    python tools/lane_timeline.py gpurun_out/r02g/prof/r02_kernel_trace.csv [--thresh 0.6] [--out profiles/r02_lane_timeline.txt]
"""
import argparse
import collections
import csv
import sys

CUS, LDS_CU, VGPR_SIMD, WAVES_SIMD = 256, 160 * 1024, 512, 8


def clean(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()[:60]


# rocprofv3's dispatch record does not carry dynamic LDS and reports allocation granules, so the GEMM and EM kernels' resident
# blocks per CU come from their launch code (conv.hip: __launch_bounds__(256, 2), 3 for the small-tile wgrad variants; caps.hip)
OVERRIDE = (("conv_x6_kernel<256", 1), ("conv_x6_kernel<64, 64", 3), ("conv_x6_kernel", 2), ("wgrad4_kernel", 3), ("wgrad_kernel<64,", 3), ("wgrad_kernel<128, 256", 3), ("conv_gemm", 2), ("wgrad", 2), ("em_fwd", 1), ("em_bwd", 1))


def resident(r):
    n = clean(r["Kernel_Name"])
    for pre, v in OVERRIDE:
        if n.startswith(pre):
            return v
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    waves = max(1, (wg + 63) // 64)
    vg = max(1, int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
    wps = max(1, min(WAVES_SIMD, VGPR_SIMD // vg))
    by_vgpr = max(1, wps * 4 // waves)
    lds = int(r["LDS_Block_Size"])
    by_lds = LDS_CU // lds if lds else 64
    return max(1, min(by_vgpr, by_lds, 32 // waves if waves <= 32 else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace"); ap.add_argument("--thresh", type=float, default=0.6); ap.add_argument("--out")
    ap.add_argument("--window", type=int, default=-1, help="which step of the trace (-1 = the last complete one; the last step of a short profiled "
                    "run can carry host stalls of the profiler's own)")
    ap.add_argument("--expect-ms", type=float, default=0.0, help="the bench's ms_per_step: fail if the window found is shorter than half of it "
                    "(a window that is not a step proves nothing)")
    ap.add_argument("--by-lane", type=int, default=0, help="also print, per stream, the N kernels with the largest summed duration in this step")
    ap.add_argument("--sequence", type=int, default=-1, help="print the dispatches of this stream id in order (start offset, duration, kernel, blocks, "
                    "kernels of other streams live at its start)")
    ap.add_argument("--gaps", type=int, default=0, help="print the N longest intervals of the step in which NO kernel runs: offset, length, the kernel that ended "
                    "last before it and the one that starts after it (with their streams), and a histogram of all such intervals by length")
    ap.add_argument("--lane-gaps", type=int, default=-1, help="print the longest idle intervals of this stream id between two of its kernels and which other "
                    "stream's kernel ended last inside each (the dependency it most likely waited for)")
    a = ap.parse_args()
    rows = [r for r in csv.DictReader(open(a.trace)) if r["Kind"] == "KERNEL_DISPATCH"]
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    adams = sorted(r["e"] for r in rows if "adam_kernel" in r["Kernel_Name"])
    nd = [r["s"] for r in rows if "to_ndhwc_kernel" in r["Kernel_Name"]]
    heads = [t for i, t in enumerate(nd) if i == 0 or t - nd[i - 1] > 2_000_000]      # first to_ndhwc of every step (the pair is microseconds apart)
    assert len(heads) >= 2 and adams, "need two steps in the trace"
    wins = []
    for i in range(len(heads)):
        prev_adam = max([e for e in adams if e <= heads[i]], default=None)
        nxt_head = heads[i + 1] if i + 1 < len(heads) else None
        last_adam = max([e for e in adams if e > heads[i] and (nxt_head is None or e <= nxt_head)], default=None)
        if prev_adam is None or last_adam is None:
            continue
        start = min(r["s"] for r in rows if r["s"] >= prev_adam and r["s"] <= heads[i])
        wins.append((start, last_adam))
    assert wins, "no complete step in the trace"
    t0, t1 = wins[a.window]
    if a.expect_ms and (t1 - t0) / 1e6 < 0.5 * a.expect_ms:
        sys.exit("lane_timeline: the window found is %.3f ms, the step takes %.3f ms: not a step" % ((t1 - t0) / 1e6, a.expect_ms))
    step = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
    out = []
    out.append("step window %.3f ms, %d kernel dispatches" % ((t1 - t0) / 1e6, len(step)))
    per = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = (r["Queue_Id"], r["Stream_Id"])
        per[k][0] += 1; per[k][1] += r["e"] - r["s"]
    out.append("\nper stream (queue, stream): kernels, summed duration ms, share of the step")
    for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out.append("  %-12s %5d %8.3f %6.2f" % (k, n, d / 1e6, d / (t1 - t0)))
    ev = []
    for i, r in enumerate(step):
        blocks = 1
        for ax in "XYZ":
            blocks *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
        r["fill"] = min(1.0, blocks / float(CUS * resident(r)))
        ev.append((r["s"], 1, i)); ev.append((r["e"], 0, i))
    ev.sort()
    live = set()
    hist = collections.Counter()
    under = collections.Counter()
    alone = collections.Counter()
    prev = t0
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            f = sum(step[j]["fill"] for j in live)
            hist[min(10, int(f * 10))] += dt
            if f < a.thresh:
                for j in live:
                    under[clean(step[j]["Kernel_Name"])] += dt
                if len(live) == 1:
                    alone[clean(step[next(iter(live))]["Kernel_Name"])] += dt
                if not live:
                    under["(nothing running)"] += dt
        prev = t
        (live.add if kind else live.discard)(i)
    out.append("\nwall time by summed fill of the running kernels (1.0 = every resident-block slot of the chip could be taken)")
    for b in range(11):
        out.append("  fill %s %8.3f ms" % ("<%.1f" % ((b + 1) / 10) if b < 10 else ">=1.0", hist[b] / 1e6))
    out.append("\nkernels running while fill < %.1f (ms of such time; a kernel counts for every instant it is live)" % a.thresh)
    for k, d in under.most_common(25):
        out.append("  %8.3f  (alone %6.3f)  %s" % (d / 1e6, alone[k] / 1e6, k))
    if a.by_lane:
        for k, _ in sorted(per.items(), key=lambda kv: -kv[1][1]):
            agg = collections.defaultdict(lambda: [0, 0])
            for r in step:
                if (r["Queue_Id"], r["Stream_Id"]) == k:
                    v = agg[clean(r["Kernel_Name"])]; v[0] += 1; v[1] += r["e"] - r["s"]
            out.append("\nstream %s: kernel, dispatches, summed ms" % (k,))
            for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.by_lane]:
                out.append("  %8.3f %4d  %s" % (d / 1e6, c, n))
    if a.gaps:
        gaps, live_n, prev_t, last_end = [], 0, t0, None
        for t, kind, i in ev:
            if live_n == 0 and t > prev_t and kind == 1:
                gaps.append((t - prev_t, prev_t, last_end, i))
            if kind:
                live_n += 1
            else:
                live_n -= 1; last_end = i
            prev_t = t
        tot = sum(g[0] for g in gaps)
        out.append("\nintervals with no kernel running: %d, %.3f ms in total; by length:" % (len(gaps), tot / 1e6))
        for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1 << 30)):
            sel = [g[0] for g in gaps if lo * 1000 <= g[0] < hi * 1000]
            out.append("  %4s - %-5s us  %4d  %8.3f ms" % (lo, hi if hi < (1 << 30) else "", len(sel), sum(sel) / 1e6))
        out.append("the %d longest: offset ms, length us, ended before (stream) -> starts after (stream)" % a.gaps)
        for ln, at, le, nx in sorted(gaps, reverse=True)[:a.gaps]:
            b = "(step start)" if le is None else "%s (%s)" % (clean(step[le]["Kernel_Name"])[:44], step[le]["Stream_Id"])
            out.append("  %8.3f %7.1f  %s -> %s (%s)" % ((at - t0) / 1e6, ln / 1e3, b, clean(step[nx]["Kernel_Name"])[:44], step[nx]["Stream_Id"]))
    if a.lane_gaps >= 0:
        mine = sorted((r for r in step if int(r["Stream_Id"]) == a.lane_gaps), key=lambda r: r["s"])
        gl = []
        for u, v in zip(mine, mine[1:]):
            if v["s"] - u["e"] > 4000:
                gl.append((v["s"] - u["e"], u, v))
        out.append("\nstream %d: %d intervals of more than 4 us between its kernels, %.3f ms in total (the stream waits for another lane, or the host)"
                   % (a.lane_gaps, len(gl), sum(g[0] for g in gl) / 1e6))
        out.append("the 40 longest: offset ms, length us, its kernel before -> after | the kernel of another stream that ended last inside the interval")
        for ln, u, v in sorted(gl, key=lambda g: -g[0])[:40]:
            ended = [o for o in step if o["Stream_Id"] != u["Stream_Id"] and u["e"] <= o["e"] <= v["s"]]
            w = max(ended, key=lambda o: o["e"]) if ended else None
            out.append("  %8.3f %7.1f  %s -> %s | %s" % ((u["e"] - t0) / 1e6, ln / 1e3, clean(u["Kernel_Name"])[:36], clean(v["Kernel_Name"])[:36],
                                                     "-" if w is None else "%s (%s), %.1f us before" % (clean(w["Kernel_Name"])[:36], w["Stream_Id"], (v["s"] - w["e"]) / 1e3)))
    if a.sequence >= 0:
        out.append("\nstream %d in order: start ms, duration ms, blocks, kernel | live on other streams at its start" % a.sequence)
        for r in sorted(step, key=lambda r: r["s"]):
            if int(r["Stream_Id"]) != a.sequence:
                continue
            others = [clean(o["Kernel_Name"])[:28] for o in step if o is not r and o["s"] <= r["s"] < o["e"]]
            blocks = 1
            for ax in "XYZ":
                blocks *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
            out.append("  %8.3f %7.3f %6d  %-44s | %s" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, blocks, clean(r["Kernel_Name"])[:44], ", ".join(others)))
    text = "\n".join(out) + "\n"
    sys.stdout.write(text)
    if a.out:
        open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
