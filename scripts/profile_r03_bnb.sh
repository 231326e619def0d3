#!/bin/bash
# Round-3 profile set of the B&B paths (run on the GPU box from the repo root): writes gpurun_out/r03/*
set -o pipefail
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MVX_BNB_TIMING=1 python3 $R/scripts/bnbcuts.py 600 > $O/bnb_cuts_512x1024.jsonl 2> $O/bnb_cuts.err
grep "bnb window timing" $O/bnb_cuts.err > $O/bnb_phase_times.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tr1 -- python3 $R/scripts/bnbtrace.py > $O/bnbtrace.log 2>&1
{ grep "wall ms" $O/bnbtrace.log; python3 $R/scripts/kstats.py $O/tr1 | head -14; python3 $R/scripts/trace_busy.py $O/tr1 2>&1 | tail -2; } > $O/bnb_512x1024_window64_kernel_busy.txt
rm -rf $O/tr1
MVX_BNB_TIMING=1 python3 $R/scripts/bnbtrace.py 2>&1 | grep "bnb window timing\|wall ms" | tail -2 >> $O/bnb_phase_times.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tr2 -- python3 $R/scripts/config5time.py 64 > $O/c5.log 2>&1
{ grep nodes_per $O/c5.log; python3 $R/scripts/kstats.py $O/tr2 | head -12; python3 $R/scripts/trace_busy.py $O/tr2 2>&1 | tail -1; } > $O/config5_kernel_stats.txt
rm -rf $O/tr2
MVX_BNB_TIMING=1 python3 $R/scripts/config5time.py 64 2>&1 | grep "bnb window timing\|nodes_per" | tail -2 >> $O/bnb_phase_times.txt
python3 $R/bench.py > $O/bench_4096x8192.json 2> $O/bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_4096x8192_driver_form.json 2>> $O/bench.err
python3 $R/scripts/roundstats.py config5 > $O/config5_round_stats.txt 2>/dev/null
python3 $R/scripts/roundstats.py wide > $O/wide_round_stats.txt 2>/dev/null
python3 $R/scripts/childtime.py 4096 8192 4 > $O/child_solves_4096x8192.jsonl 2>/dev/null
python3 $R/scripts/bnbrepeat.py 10 2>/dev/null | grep -v "^first" > $O/bnb_determinism_soak.txt
ls $O
