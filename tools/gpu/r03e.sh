#!/bin/bash
# full GPU suite + default bench (+ PICONS_WINO=0 A/B)
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
PICONS_WINO=0 timeout 600 python3 bench.py --no-cpu-baseline --resident-inputs > $O/bench_nowino.json 2> $O/bench_nowino.err
tail -6 $O/tests.log
python3 - <<'PY'
import json
for f in ("bench","bench_nowino"):
    try:
        j=json.load(open("gpurun_out/r03e/%s.json"%f))
        print(f, "%.3f ms  %.1f clips/s  staged %s  conv frac %.3f kernel_ms %.2f  step frac %.3f  fam %s" % (j["ms_per_step"], j["value"], j["staged"] and "%.3f"%j["staged"]["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel_ms_per_step"], j["roofline_step"]["frac"], j["roofline_step"].get("gflop_by_family")))
    except Exception as e:
        print(f, "failed", e)
PY
