#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  rm -rf /tmp/ab_$v
  BBQ_LIB=$R/ab_tmp/libbbq_$v.so rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_$v -- python3 $R/bench.py --no-configs --no-napi --no-hbm-only --no-raw --no-cpu-baseline --inprocess-shards 0 --latency-calls 0 --no-recall --no-shard-shape --no-c1 --no-parity --steps 3 --warmup 1 --slots 1 > /dev/null 2>&1
  python3 - /tmp/ab_$v $v <<'PY'
import csv, glob, sys, collections
rows = [r for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") for r in csv.DictReader(open(f))]
by = collections.defaultdict(list)
for r in rows:
    if "mfma" in r["Kernel_Name"]:
        by[int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2], {k: round(sum(v) / len(v), 1) for k, v in sorted(by.items())})
PY
done
