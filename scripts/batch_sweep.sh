for cfg in "PREFIX=1 CHUNK=8" "PREFIX=0 CHUNK=1000" "PREFIX=1 CHUNK=4" "PREFIX=1 CHUNK=2" "PREFIX=1 CHUNK=16" "PREFIX=1 CHUNK=1000"; do
  set -- $cfg; echo "$cfg"
  env MVX_BNB_$1 MVX_BNB_$2 MVX_BATCH_PRED=0 timeout -k 10 150 bash scripts/bnb3.sh 2>&1 | grep "wall ms\|nodes_per_s" | cut -c1-160 || exit 1
done
