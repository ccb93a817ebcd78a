#!/bin/bash
# average memory-side read latency of the dominant scan launch (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) for several cache-residency
# settings: cache hits shorten it.  scripts/pmc_ea_latency.sh <outdir> "<il:mb list>"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/${1:-gpurun_out/ea_latency}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-recall --no-cpu-baseline --no-parity --latency-calls 0 --shared-sweep 0 --no-configs --no-napi --no-raw --no-hbm-only --inprocess-shards 0"
for spec in ${2:-0:0 0:224 0:448 0:640 1:224 1:448}; do
  il=${spec%%:*}; mb=${spec##*:}
  rm -rf /tmp/rp_ea
  rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d /tmp/rp_ea -- python3 $R/bench.py --steps 2 --warmup 1 --batch 32 --slots 1 --opt resident_interleave=$il --opt resident_mb=$mb $COMMON > $OUT/bench_${il}_$mb.json 2> $OUT/err_${il}_$mb.txt || exit 1
  cp $(ls /tmp/rp_ea/*/*counter_collection.csv | head -1) $OUT/pmc_${il}_$mb.csv
  python3 - "$OUT/pmc_${il}_$mb.csv" $il $mb <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bbq_scan_kernel" in r["Kernel_Name"]]
big = max(int(r["Grid_Size"]) for r in rows)
acc = collections.defaultdict(list)
for r in rows:
    if int(r["Grid_Size"]) == big:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("interleave", sys.argv[2], "mb", sys.argv[3], "launches", len(acc["TCC_EA0_RDREQ_sum"]), "avg EA read latency (cycles)", round(m["TCC_EA0_RDREQ_LEVEL_sum"] / m["TCC_EA0_RDREQ_sum"], 1),
      "requests", int(m["TCC_EA0_RDREQ_sum"]), "dram credit stall", int(m.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0)))
PY
done
