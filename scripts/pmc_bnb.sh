#!/bin/bash
# SQ counters per launch of the node-LP kernels (k_dsel, k_select, k_update) on the first 600 nodes of the config-5 tree
set -o pipefail
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_bnb -- python3 $R/scripts/config5prefix.py 600 > $O/pmc_bnb.log 2>&1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$O/pmc_bnb/**/*counter_collection.csv", recursive=True)
out = collections.defaultdict(lambda: collections.defaultdict(list))
if f:
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mvx::", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open("$O/pmc_sq_node_lp_kernels.json", "w") as g:
        json.dump({k: {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()} for k, d in out.items() if k.startswith(("k_dsel", "k_select", "k_update"))}, g, indent=1)
    print(open("$O/pmc_sq_node_lp_kernels.json").read()[:3000])
PY
rm -rf $O/pmc_bnb
