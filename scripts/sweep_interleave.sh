#!/bin/bash
# residency settings over index sizes: scripts/sweep_interleave.sh <outdir> "<rows or configs>" "<interleave:MiB list>" (MiB -1 = the default rule)
OUT=${1:-gpurun_out/interleave}
mkdir -p "$OUT"
COMMON="--no-cpu-baseline --no-recall --no-napi --no-raw --no-hbm-only --no-configs --no-parity --inprocess-shards 0 --latency-calls 100 --shared-sweep 0"
for rows in ${2:-10000000 4000000 20000000}; do
  case $rows in c*) sel="--config $rows"; steps=32;; *) sel="--rows $rows"; steps=$(( 80000000 / rows )); [ $steps -lt 4 ] && steps=4;; esac
  for spec in ${3:-1:224:100 1:224:50 1:224:25 1:240:100 1:192:100 0:224:100 1:224:0}; do
    IFS=: read il mb sp <<< "$spec"
    timeout -k 10 200 python bench.py $COMMON $sel --steps $steps --warmup 2 --opt resident_interleave=$il --opt resident_mb=$mb > "$OUT/h_${rows}_${il}_${mb}_$sp.json" 2> "$OUT/h_${rows}_${il}_${mb}_$sp.err" || exit 1
    python -c "import json,sys; d=json.loads(open('$OUT/h_${rows}_${il}_${mb}_$sp.json').read().strip().splitlines()[-1]); print('rows', '$rows', 'interleave', $il, 'mb', $mb, round(d['value']), 'e2e', round(d['end_to_end_hbm_frac'],4), 'kernel', round(d['roofline']['frac'],4), round(d['roofline']['cache_resident_frac_of_sweep'],3), 'lat', round(d['latency']['p50_ms'],4), round(d['latency']['min_ms'],4))"
  done
done
