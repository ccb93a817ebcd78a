#!/bin/bash
# round 4's rocprofv3 evidence in one GPU call: kernel trace + the two PMC passes of the headline, c2, c4 and c5, the same launches
# traced with nothing kept in the Infinity Cache (resident_mb 0), the counters of the matrix-core shared sweep, the single-query timeline
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r4ev
for cfg in headline c2 c4 c5; do
  extra=""; [ "$cfg" != "headline" ] && extra="--config $cfg"
  bash $R/scripts/collect_profiles.sh $cfg $extra > $R/gpurun_out/r4ev/collect_$cfg.log 2>&1 && echo "$cfg profiles ok" || echo "$cfg profiles FAILED"
  TRACE_ONLY=1 bash $R/scripts/collect_profiles.sh ${cfg}_strict $extra --opt resident_mb=0 > $R/gpurun_out/r4ev/collect_${cfg}_strict.log 2>&1 && echo "$cfg strict trace ok" || echo "$cfg strict trace FAILED"
done
bash $R/scripts/profile_mfma.sh gpurun_out/r4ev_mfma > $R/gpurun_out/r4ev/mfma.log 2>&1 && echo "mfma counters ok"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_lat && rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_lat -- python3 $R/scripts/latency.py 10000000 only=1 > $R/gpurun_out/r4ev/latency_under_trace.txt 2>/dev/null && cp $(ls /tmp/rp_lat/*/*kernel_trace.csv | head -1) $R/gpurun_out/r4ev/latency_kernel_trace.csv ) && python3 $R/scripts/trace_one_call.py $R/gpurun_out/r4ev/latency_kernel_trace.csv > $R/gpurun_out/r4ev/single_query_timeline.txt && cat $R/gpurun_out/r4ev/single_query_timeline.txt
