#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_r
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for x6 in 0 1; do
  PICONS_WGRAD_STEM_X6=$x6 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$x6 -o st -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $O/prof_$x6.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_$x6/st_kernel_trace.csv")))
ev=sorted([(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','')[:50], r.get('Stream_Id')) for r in rows])
w=[i for i,e in enumerate(ev) if e[2].startswith('wgrad4')]
print("x6=$x6: stem wgrad launches", len(w))
for i in w[2:8]:
    s,e_=ev[i][0],ev[i][1]
    # kernels overlapping the stem wgrad
    ov={}
    for q in ev:
        if q[1]>s and q[0]<e_ and q is not ev[i]:
            ov[q[2][:34]]=ov.get(q[2][:34],0)+(min(q[1],e_)-max(q[0],s))/1e3
    # next step's stem conv start
    nxt=[q for q in ev if q[0]>e_ and q[2].startswith('conv_gemm_glds_kernel<128, 64, 2, 2, 36')]
    gap=(nxt[0][0]-e_)/1e3 if nxt else -1
    print("  dur %.1f us; next stem conv starts %.1f us after; overlapping: %s"%((e_-s)/1e3, gap, {k:round(v,1) for k,v in sorted(ov.items(), key=lambda kv:-kv[1])[:5]}))
PY
done
