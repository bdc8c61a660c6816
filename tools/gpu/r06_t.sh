#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/r06_t
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for late in 1 0; do
  PICONS_BENCH_LATE_PREP=$late timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$late -o st -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $O/prof_$late.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_$late/st_kernel_trace.csv")))
ev=sorted([(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','')[:40], r.get('Stream_Id'), r.get('Queue_Id')) for r in rows])
stem=[e for e in ev if e[2].startswith('conv_gemm_glds_kernel<128, 64, 2, 2, 36')]
wg=[e for e in ev if e[2].startswith('wgrad4')]
clip=[e for e in ev if e[2].startswith('clip_from_u8')]
print("late_prep=$late: stem convs", len(stem), "clip kernels", len(clip), "queues of clip kernels", sorted({e[4] for e in clip}), "streams", sorted({e[3] for e in clip}))
# staged-leg steps: consecutive stem convs with period ~19ms
for a,b in list(zip(stem[:-1], stem[1:]))[3:9]:
    per=(b[0]-a[0])/1e6
    cl=[c for c in clip if a[0] <= c[0] < b[0]]
    w=[x for x in wg if a[0] <= x[0] < b[0]]
    if not cl or not w: continue
    print("  step period %.3f ms: clip kernels run at +%.2f .. +%.2f ms of the step (n=%d, busy %.0f us); stem wgrad +%.2f .. +%.2f; next stem conv starts %.0f us after the stem wgrad ends" % (per, (cl[0][0]-a[0])/1e6, (cl[-1][1]-a[0])/1e6, len(cl), sum(c[1]-c[0] for c in cl)/1e3, (w[-1][0]-a[0])/1e6, (w[-1][1]-a[0])/1e6, (b[0]-w[-1][1])/1e3))
PY
done
