#!/bin/bash
# how busy are the vector ALUs in the largest scan launch?  SQ counters + a kernel trace of the same command.
#   scripts/pmc_scan_valu.sh <outdir> [bench args, e.g. --config c2]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/${1:-gpurun_out/scan_valu}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-recall --no-cpu-baseline --no-parity --latency-calls 0 --shared-sweep 0 --no-configs --no-napi --no-raw --no-hbm-only --inprocess-shards 0 --steps 2 --warmup 1 --slots 1"
rm -rf /tmp/rp_v1 /tmp/rp_v2 /tmp/rp_vt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_vt -- python3 $R/bench.py "$@" $COMMON > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
cp $(ls /tmp/rp_vt/*/*kernel_trace.csv | head -1) $OUT/kernel_trace.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/rp_v1 -- python3 $R/bench.py "$@" $COMMON > /dev/null 2> $OUT/pmc1.err || exit 1
cp $(ls /tmp/rp_v1/*/*counter_collection.csv | head -1) $OUT/pmc1.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d /tmp/rp_v2 -- python3 $R/bench.py "$@" $COMMON > /dev/null 2> $OUT/pmc2.err || exit 1
cp $(ls /tmp/rp_v2/*/*counter_collection.csv | head -1) $OUT/pmc2.csv
python3 - $OUT <<'PY'
import csv, sys, os, collections
out = sys.argv[1]
tr = [r for r in csv.DictReader(open(os.path.join(out, "kernel_trace.csv"))) if "bbq_scan_kernel" in r["Kernel_Name"]]
def gsz(r): return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
big = max(gsz(r) for r in tr)
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr if gsz(r) == big]
dur = sum(d) / len(d) * 1e-9
pl = collections.defaultdict(list)
for f in ("pmc1.csv", "pmc2.csv"):
    rows = [r for r in csv.DictReader(open(os.path.join(out, f))) if "bbq_scan_kernel" in r["Kernel_Name"]]
    b = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == b: pl[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in pl.items()}
simd = dur * 1024 * 2.1e9
waves = m.get("SQ_WAVES", 1)
print("largest launch: %.1f us, %d launches, waves %.0f" % (dur * 1e6, len(d), waves))
print("VALU active share of SIMD time (x4 quad-cycles, 2.1 GHz): %.3f" % (m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd))
print("per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM_RD %.0f instructions" % (m.get("SQ_INSTS_VALU", 0) / waves, m.get("SQ_INSTS_SALU", 0) / waves, m.get("SQ_INSTS_LDS", 0) / waves, m.get("SQ_INSTS_VMEM_RD", 0) / waves))
print("cycles per VALU instruction: %.2f" % (m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(1, m.get("SQ_INSTS_VALU", 1))))
wc = max(1, m.get("SQ_WAVE_CYCLES", 1))
print("wave time: issuing %.3f  waiting for issue %.3f  waiting (any) %.3f" % (m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc))
print({k: round(v) for k, v in m.items()})
PY
