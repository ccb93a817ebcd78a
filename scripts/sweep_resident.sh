#!/bin/bash
# the headline step and the config legs with different amounts of the index kept cache-resident (option resident_mb)
#   scripts/sweep_resident.sh <outdir> ["headline values"] ["c2 values"] ["c5 values"] ["c4 values"]
OUT=${1:-gpurun_out/resident}
mkdir -p "$OUT"
COMMON="--no-cpu-baseline --no-recall --no-napi --no-raw --no-hbm-only --no-configs --no-parity --inprocess-shards 0 --latency-calls 100 --shared-sweep 0"
for mb in ${2:-0 192 224 256 288 0 224}; do
  timeout -k 10 200 python bench.py $COMMON --steps 8 --warmup 2 --opt resident_mb=$mb > "$OUT/headline_$mb.json" 2> "$OUT/headline_$mb.err" || exit 1
  python -c "import json,sys; d=json.loads(open('$OUT/headline_$mb.json').read().strip().splitlines()[-1]); print('headline', $mb, round(d['value']), round(d['roofline']['frac'],4), d['latency']['p50_ms'])"
done
for cfg in c2 c5 c4; do
  case $cfg in c2) vals=${3:-0 -1 0 -1};; c5) vals=${4:-0 192 224 240 256 0};; c4) vals=${5:-0 224 0 224};; esac
  for mb in $vals; do
    timeout -k 10 200 python bench.py $COMMON --config $cfg --steps 32 --warmup 4 --opt resident_mb=$mb > "$OUT/${cfg}_$mb.json" 2> "$OUT/${cfg}_$mb.err" || exit 1
    python -c "import json,sys; d=json.loads(open('$OUT/${cfg}_$mb.json').read().strip().splitlines()[-1]); print('$cfg', $mb, round(d['value']), round(d['roofline']['frac'],4), d['latency']['p50_ms'])"
  done
done
