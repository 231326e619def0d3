#!/bin/bash
# The three B&B timings (wide tree, config-5 tree, cut modes) with the driver's phase times; extra env goes in front.
R=$PWD
export MVX_BNB_TIMING=1 MVX_COPY_TIMING=1
python3 $R/scripts/bnbtrace.py 2>&1 | grep "timing\|wall ms\|copy_prob" | tail -3
python3 $R/scripts/config5time.py 64 2>&1 | grep "timing\|nodes_per\|copy_prob" | tail -3 | cut -c1-260
python3 $R/scripts/bnbcuts.py 600 2>&1 | grep "window\": 64" | cut -c1-200
