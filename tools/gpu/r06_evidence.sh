#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
# Round-6 evidence on the final code: kernel stats (4 lanes / 1 lane), launch table, lane timeline, PMC passes (each in its own run with
# --kernel-trace only), the default bench line and its variants.  Summaries are copied into profiles/ by the caller.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06ev
mkdir -p $O
cd $R
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err
# the driver's exact command: its five roofline blocks must agree with the launch table below
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench.err
MS=$(python3 -c "import json; print(json.load(open('$O/bench.json'))['resident']['ms_per_step'])")
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --resident-inputs --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r06 -- $B > $O/prof.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l1 -o l1 -- $B > $O/prof_l1.log 2>&1
PICONS_SPLIT=0 PICONS_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l1_split0 -o l1s0 -- $B > $O/prof_l1_split0.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $B > $O/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $B > $O/pmc_w.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $B > $O/pmc_sq.log 2>&1
PICONS_LANES=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -o m -- $B > $O/pmc_m.log 2>&1
cd $R
python3 tools/launch_table.py $O/prof_l1/l1_kernel_trace.csv 70 > $O/launch_table.txt 2>&1
python3 tools/lane_timeline.py $O/prof/r06_kernel_trace.csv --window -1 --expect-ms $MS --by-lane 6 --gaps 30 > $O/lane_timeline.txt 2>&1
python3 tools/summarize_pmc.py --stats $O/prof_l1/l1_kernel_stats.csv --fetch $O/pmc_f/f_counter_collection.csv --write $O/pmc_w/w_counter_collection.csv \
    --sq $O/pmc_sq/sq_counter_collection.csv --mfma $O/pmc_m/m_counter_collection.csv --steps 3 --tag r06 > $O/summarize.log 2>&1
cp profiles/r06_traffic.json profiles/r06_hbm_table.md profiles/r06_pmc_sq_gemm.csv profiles/r06_mfma_counter_check.txt $O/ 2>/dev/null
cp $O/prof/r06_kernel_stats.csv $O/r06_kernel_stats.csv; cp $O/prof_l1/l1_kernel_stats.csv $O/r06_l1_kernel_stats.csv
timeout 300 python3 bench.py --gv --no-cpu-baseline --no-extra-legs > $O/bench_gv.json 2>> $O/bench.err
timeout 300 python3 bench.py --jhmdb --no-cpu-baseline --no-extra-legs > $O/bench_jhmdb.json 2>> $O/bench.err
timeout 300 python3 bench.py --epoch 12 --no-cpu-baseline --no-extra-legs > $O/bench_epoch12.json 2>> $O/bench.err
PICONS_FORCE_REDUCER=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 python3 bench.py --no-cpu-baseline --no-extra-legs > $O/bench_rccl_one_rank.json 2>> $O/bench.err
timeout 300 python3 bench.py --gpus 2 --dry-launch > $O/bench_dry_launch_2.json 2>> $O/bench.err
python3 tools/compare_conv_launches.py $O/prof_l1/l1_kernel_trace.csv $O/prof_l1_split0/l1s0_kernel_trace.csv 3 > $O/x6_launches.txt 2>&1
tail -14 $O/summarize.log; tail -8 $O/launch_table.txt; head -12 $O/lane_timeline.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06ev/bench*.json")):
    try:
        j = json.load(open(f)); print(f.split("/")[-1], j.get("ms_per_step"), j.get("value"), (j.get("resident") or {}).get("ms_per_step"), (j.get("dict_contract") or {}).get("ms_per_step"), (j.get("split_off") or {}).get("ms_per_step"), j.get("ranks_observed"))
    except Exception as e:
        print(f, "failed", e)
PY
