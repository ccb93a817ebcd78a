#!/bin/bash
# PMC passes for the shared sweep on the matrix cores (bbq_scan_mfma_kernel): two rocprofv3 --pmc runs (8 SQ counters each) + a kernel
# trace of the same command; summaries go to gpurun_out/<dir>/, the numbers quoted in DESIGN.md are copied into profiles/.
#   scripts/profile_mfma.sh <outdir>
set -e
OUT=${1:-gpurun_out/mfma_pmc}
mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --no-recall --no-cpu-baseline --no-napi --no-raw --no-hbm-only --no-configs --no-shard-shape --no-c1 --no-parity --inprocess-shards 0 --latency-calls 0 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- $CMD > "$ROOT/$OUT/bench_trace.json" 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$ROOT/$OUT/pmc1" -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d "$ROOT/$OUT/pmc2" -- $CMD > /dev/null 2>&1
ls "$ROOT/$OUT"/*/* | head -20
