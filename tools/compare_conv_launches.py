#!/usr/bin/env python3
"""Side by side: the conv / dgrad launches of one step from two single-lane rocprofv3 kernel traces of the same plan (same launch order),
e.g. PICONS_SPLIT=1 against PICONS_SPLIT=0.
    python tools/compare_conv_launches.py <a_kernel_trace.csv> <b_kernel_trace.csv> <steps in each trace>"""
import csv, re, sys


def load(p):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
            for r in rows if "conv_gemm" in r["Kernel_Name"] or "conv_x6" in r["Kernel_Name"]]


def short(nm):
    m = re.search(r"(conv_\w+)<([^>]*)>", nm)
    return "%s<%s>" % (m.group(1).replace("conv_gemm_", "").replace("_kernel", ""), m.group(2).replace(" ", ""))


a, b, steps = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
assert len(a) == len(b) and len(a) % steps == 0, (len(a), len(b))
n = len(a) // steps
a, b = a[-n:], b[-n:]
ta = tb = ca = cb = 0.0
for (na, ua, ga), (nb, ub, gb) in zip(a, b):
    if na != nb:
        ca += ua; cb += ub
    ta += ua; tb += ub
    print("%-26s %8.1f us %5d blk | %-26s %8.1f us %5d blk | %.2fx" % (short(na), ua, ga, short(nb), ub, gb, ub / ua))
print("launches whose kernel differs: %.1f us against %.1f us; all %d conv launches: %.1f us against %.1f us" % (ca, cb, n, ta, tb))
