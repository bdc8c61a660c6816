#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter CSVs into the per-launch HBM traffic figure bench.py reports.

    export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing
    python tools/summarize_pmc.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/r01_traffic.json

(separate passes: TCC has 4 counter slots, FETCH_SIZE costs 3 and WRITE_SIZE 2.)  gfx950 corrections per
/opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-B requests as 64 B for 16-B-per-lane streams,
so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KiB.
"""
import collections
import csv
import json
import sys


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        per[name][0] += 1
        per[name][1] += float(r["Counter_Value"])
    return per


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md; per launch, bench.py workload (bs=8)",
           "kernels": {}}
    conv = [0, 0.0]
    for k in sorted(set(fetch) | set(write)):
        nl = max(fetch[k][0], write[k][0])
        b = (2.0 * fetch[k][1] + write[k][1]) * 1024.0
        out["kernels"][k] = {"launches": nl, "fetch_kib": fetch[k][1], "write_kib": write[k][1], "hbm_bytes_per_launch": b / max(nl, 1)}
        if k.startswith("conv_gemm"):
            conv[0] += nl
            conv[1] += b
    out["conv_gemm_hbm_bytes_per_launch"] = conv[1] / max(conv[0], 1)
    out["conv_gemm_launches"] = conv[0]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print("conv_gemm: %d launches, %.1f MB HBM per launch" % (conv[0], out["conv_gemm_hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
