#!/bin/bash
# collects one configuration's rocprofv3 evidence on the GPU box into gpurun_out/prof_<tag>/ (then, back in the build container,
# `python scripts/summarize_profiles.py gpurun_out/prof_<tag> rNN <tag>` writes the committed summaries under profiles/)
#   scripts/collect_profiles.sh <tag> [bench.py arguments that select the workload, e.g. --config c5]
#   TRACE_ONLY=1 scripts/collect_profiles.sh <tag>_strict [...] --opt resident_mb=0     the same launches with nothing kept in the Infinity Cache
# The program after `--` is python3 itself (no env/bash hop: the profiler's library has initialised the GPU by then).
set -o pipefail
TAG=${1:-headline}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_kt /tmp/rp_f /tmp/rp_w
COMMON="--no-recall --no-cpu-baseline --no-parity --latency-calls 0 --shared-sweep 0 --no-configs --no-napi --no-raw --no-hbm-only --inprocess-shards 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_kt -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $COMMON > $OUT/bench_under_trace.json 2> $OUT/trace.err || exit 1
cp $(ls /tmp/rp_kt/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
cp $(ls /tmp/rp_kt/*/*kernel_trace.csv | head -1) $OUT/kernel_trace_full.csv
echo "kernel trace done"
if [ -n "$TRACE_ONLY" ]; then ls -la $OUT; exit 0; fi   # (the strict runs - resident_mb 0 - only need the trace)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_f -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --batch 32 --slots 1 $COMMON > $OUT/bench_under_fetch.json 2> $OUT/fetch.err || exit 1
cp $(ls /tmp/rp_f/*/*counter_collection.csv | head -1) $OUT/pmc_fetch_full.csv
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_w -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --batch 32 --slots 1 $COMMON > $OUT/bench_under_write.json 2> $OUT/write.err || exit 1
cp $(ls /tmp/rp_w/*/*counter_collection.csv | head -1) $OUT/pmc_write_full.csv
echo "write pass done"
ls -la $OUT
