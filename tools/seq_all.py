#!/usr/bin/env python3
"""All lanes of one step in start order: start ms, duration, lane, blocks, kernel.  python tools/seq_all.py trace.csv [lo_ms hi_ms] [--window -2]"""
import csv, re, sys
a = [x for x in sys.argv[1:] if not x.startswith("--")]
win = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else -2
a = [x for x in a if x != str(win)] if "--window" in sys.argv else a
rows = [r for r in csv.DictReader(open(a[0])) if r["Kind"] == "KERNEL_DISPATCH"]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
ad = sorted(r["e"] for r in rows if "adam_kernel" in r["Kernel_Name"])
wins = list(zip(ad, ad[1:]))
t0, t1 = wins[win]
lo, hi = (float(a[1]), float(a[2])) if len(a) >= 3 else (0.0, 1e9)
for r in sorted((r for r in rows if r["s"] >= t0 and r["e"] <= t1), key=lambda r: r["s"]):
    s = (r["s"] - t0) / 1e6
    if lo <= s <= hi:
        blocks = 1
        for ax in "XYZ":
            blocks *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
        n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:52]
        print("%7.3f %6.3f  L%s %6d  %s" % (s, (r["e"] - r["s"]) / 1e6, r["Stream_Id"], blocks, n))
print("step %.3f ms" % ((t1 - t0) / 1e6))
