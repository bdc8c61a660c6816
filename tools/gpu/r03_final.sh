#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03fin
mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/tests.log; tail -2 $O/smoke.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r03fin/bench.json"))
print(j["ms_per_step"], j["value"], j["staged"]["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_reference_counted"], j["roofline_winograd"], j["roofline_step"]["frac"])
PY
