#!/bin/bash
# A/B of two builds of libbbq.so on ONE box (the pool's boxes differ by a few per cent): the batched leg of bench.py with one pipeline
# slot under a kernel trace, average duration of the matrix-core sweep launches per grid size.
#   scripts/ab_kernel_trace.sh ab_tmp/libbbq_a.so ab_tmp/libbbq_b.so      (paths inside the repo: they travel with gpurun; *.so is git-ignored)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=/tmp/ab_$(basename "$lib" .so)
  rm -rf "$out"
  BBQ_LIB=$R/$lib rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 $R/bench.py --no-configs --no-napi --no-hbm-only --no-raw --no-cpu-baseline \
    --inprocess-shards 0 --latency-calls 0 --no-recall --no-shard-shape --no-c1 --no-parity --steps 3 --warmup 1 --slots 1 > /dev/null 2>&1
  python3 - "$out" "$lib" <<'PY'
import csv, glob, sys, collections
rows = [r for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") for r in csv.DictReader(open(f))]
by = collections.defaultdict(list)
for r in rows:
    if "mfma" in r["Kernel_Name"]:
        by[int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2], {k: round(sum(v) / len(v), 1) for k, v in sorted(by.items())})
PY
done
