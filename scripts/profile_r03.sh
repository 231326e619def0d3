#!/bin/bash
# Round-3 profile set of the headline path (run on the GPU box from the repo root): writes gpurun_out/r03/*
set -o pipefail
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_4096x8192.json 2> $O/bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_4096x8192_driver_form.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_4096x8192_kernel_stats.csv
python3 $R/scripts/kstats.py $O/stats > $O/bench_4096x8192_trace_summary.txt
rm -rf $O/stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scripts/trace_small.py 4096 8192 300 > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/scripts/trace_small.py 4096 8192 300 > /dev/null 2> $O/pmc_w.err
python3 $R/scripts/pmc_traffic.py $O/pmc_f $O/pmc_w 4096 8192 $O/k_fbc3_traffic_4096x8192.json k_fbc3 > /dev/null
rm -rf $O/pmc_f $O/pmc_w
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -- python3 $R/scripts/trace_small.py 4096 8192 300 > /dev/null 2> $O/pmc_sq.err
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$O/pmc_sq/**/*counter_collection.csv", recursive=True)
out = collections.defaultdict(lambda: collections.defaultdict(list))
if f:
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mvx::", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open("$O/pmc_sq_by_kernel.json", "w") as g:
        json.dump({k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in out.items() if k.startswith(("k_fbc3", "k_chain", "k_fpatch"))}, g, indent=1)
PY
rm -rf $O/pmc_sq
cd $R
python3 scripts/chaindbg.py 4096 8192 400 > $O/k_chain_phase_stamps_4096x8192.txt 2>/dev/null
python3 scripts/bulktime.py 4096 8192 > $O/bulk_pass_by_chain_4096x8192.jsonl 2>/dev/null
for s in "1024 2048" "1024 4096" "2048 4096" "3072 6144" "4096 8192" "8192 8192"; do for k in 0 8 16 32; do MVX_CHAIN=$k python3 scripts/chainsweep.py $s 2>/dev/null; done; done > $O/chain_sweep.jsonl
MVX_CLUSTER=0 python3 scripts/chainsweep.py 4096 8192 2>/dev/null > $O/two_launch_path.jsonl
MVX_CLUSTER=0 python3 scripts/chainsweep.py 1024 2048 2>/dev/null >> $O/two_launch_path.jsonl
python3 scripts/call20.py > $O/call20.txt 2>/dev/null
ls -la $O
