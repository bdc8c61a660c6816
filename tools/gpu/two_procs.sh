export TMPDIR=/tmp
python3 bench.py --steps 200 --no-cpu-baseline --no-kernel-timing --resident-inputs > gpurun_out/two_a.json 2>/dev/null &
P1=$!
python3 bench.py --steps 200 --no-cpu-baseline --no-kernel-timing --resident-inputs > gpurun_out/two_b.json 2>/dev/null &
P2=$!
wait $P1 $P2
python3 -c "
import json
for f in ('a','b'):
    j=json.load(open('gpurun_out/two_%s.json'%f)); print(f, j['ms_per_step'], j['value'])
"
