#!/bin/bash
# L2 hits and misses (TCC_HIT / TCC_MISS, all XCDs) of the largest scan launch of the 1 M-row index with everything streamed, and of
# the probe with the same geometry: scripts/pmc_l2_hits.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/${1:-gpurun_out/l2_hits}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-recall --no-cpu-baseline --no-parity --latency-calls 0 --shared-sweep 0 --no-configs --no-napi --no-raw --no-hbm-only --inprocess-shards 0"
for mb in 0 -1; do
  rm -rf /tmp/rp_l2
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum --output-format csv -d /tmp/rp_l2 -- python3 $R/bench.py --config c2 --steps 2 --warmup 1 --slots 1 --opt resident_mb=$mb $COMMON > $OUT/bench_$mb.json 2> $OUT/err_$mb.txt || exit 1
  cp $(ls /tmp/rp_l2/*/*counter_collection.csv | head -1) $OUT/pmc_lib_$mb.csv
done
rm -rf /tmp/rp_l2p
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum --output-format csv -d /tmp/rp_l2p -- $R/scripts/ubench/scan_steps > $OUT/probe.txt 2> $OUT/probe_err.txt || exit 1
cp $(ls /tmp/rp_l2p/*/*counter_collection.csv | head -1) $OUT/pmc_probe.csv
python3 - $OUT <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
for name in ("pmc_lib_0.csv", "pmc_lib_-1.csv", "pmc_probe.csv"):
    rows = list(csv.DictReader(open(os.path.join(out, name))))
    by = collections.OrderedDict()
    for r in rows:
        key = (r["Dispatch_Id"], r["Kernel_Name"][:60], r["Grid_Size"])
        by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    scans = [(k, v) for k, v in by.items() if "scan_kernel" in k[1] or k[1].startswith("void k<")]
    if "lib" in name:
        big = max(int(k[2]) for k, v in scans)
        scans = [(k, v) for k, v in scans if int(k[2]) == big][-3:]
    else:
        scans = scans[:20]
    for k, v in scans:
        h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
        print(name, k[1][:40], "grid", k[2], "hit", int(h), "miss", int(m), "hit rate %.3f" % (h / max(1.0, h + m)), "req", int(v.get("TCC_REQ_sum", 0)), "streaming", int(v.get("TCC_STREAMING_REQ_sum", 0)))
PY
