#!/bin/bash
# Round 4, last call on the final code: the whole GPU suite, smoke(), the default bench line.
export TMPDIR=/tmp
O=gpurun_out/r04fin
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -4 $O/tests.log; tail -2 $O/smoke.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r04fin/bench.json"))
print(j["ms_per_step"], j["value"], j["resident"]["ms_per_step"], j["dict_contract"]["ms_per_step"], j["split_off"]["ms_per_step"], j["roofline"]["frac"], j["roofline_fp32_conv"]["frac"], j["roofline_winograd"]["frac"], j["roofline_step"]["frac"])
PY
