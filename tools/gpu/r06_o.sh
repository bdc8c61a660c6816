#!/bin/bash
# per-kernel single-lane stats: round-5 tree against the current tree
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_o
mkdir -p $O
export TMPDIR=/tmp
for tree in _ab_r05 .; do
  tag=$(basename $(cd $R/$tree; pwd))
  cd /tmp
  PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o l1 -- python3 $R/$tree/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs > $O/prof_$tag.log 2>&1
  cp $O/prof_$tag/l1_kernel_stats.csv $O/stats_$tag.csv
done
python3 - <<PY
import csv
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        d[r['Name']]=(float(r['TotalDurationNs'])/3e6, int(r['Calls'])//3)
    return d
a=load("$O/stats__ab_r05.csv"); b=load("$O/stats_repo.csv")
print("total ms/step: r05 %.3f  now %.3f"%(sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
keys=sorted(set(a)|set(b), key=lambda k: -abs(b.get(k,(0,0))[0]-a.get(k,(0,0))[0]))
for k in keys[:28]:
    x=a.get(k,(0,0)); y=b.get(k,(0,0))
    print("%+8.3f ms  r05 %7.3f (%3d)  now %7.3f (%3d)  %s"%(y[0]-x[0], x[0], x[1], y[0], y[1], k.replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','')[:70]))
PY
