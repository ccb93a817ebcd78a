#!/bin/bash
# development loop of the shared sweep: the bench's batched leg alone + a kernel trace of the same command
#   scripts/quick_batched.sh <outdir> [bench args]
OUT=${1:-gpurun_out/qb}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
COMMON="--no-configs --no-napi --no-hbm-only --no-raw --no-cpu-baseline --inprocess-shards 0 --latency-calls 0 --no-recall"
python3 $R/bench.py $COMMON "$@" > $R/$OUT/bench.json 2> $R/$OUT/bench.err || { tail -5 $R/$OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qb_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qb_tr -- python3 $R/bench.py $COMMON --no-parity --steps 3 --warmup 1 "$@" > /dev/null 2> $R/$OUT/trace.err
cp $(ls /tmp/qb_tr/*/*kernel_trace.csv | head -1) $R/$OUT/kernel_trace.csv
cp $(ls /tmp/qb_tr/*/*kernel_stats.csv | head -1) $R/$OUT/kernel_stats.csv
python3 - $R/$OUT <<'PY'
import json, sys, csv, collections
o = sys.argv[1]
d = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
b = d.get("batched") or {}
print("headline %.0f q/s; batched %.0f q/s identical %s" % (d["value"], b.get("value", 0), b.get("identical_to_unshared")))
rows = list(csv.DictReader(open(o + "/kernel_trace.csv")))
by = collections.defaultdict(list)
for r in rows:
    if "mfma" in r["Kernel_Name"] or "finalize" in r["Kernel_Name"]:
        by[(r["Kernel_Name"].split("(")[0][-30:], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    print("%-32s grid %8d  n %4d  avg %8.1f us  max %8.1f" % (k[0], k[1], len(v), sum(v) / len(v), max(v)))
PY
