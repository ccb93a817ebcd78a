set -o pipefail
mkdir -p gpurun_out/r3q
scripts/collect_profiles.sh headline > gpurun_out/r3q/collect_headline.log 2>&1 && echo "headline profiles ok"
scripts/collect_profiles.sh c2 --config c2 > gpurun_out/r3q/collect_c2.log 2>&1 && echo "c2 profiles ok"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_lat && rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_lat -- python3 $GRAFT_REPO_ROOT/scripts/latency.py 10000000 only=1 > $GRAFT_REPO_ROOT/gpurun_out/r3q/latency_under_trace.txt 2>/dev/null && cp $(ls /tmp/rp_lat/*/*kernel_trace.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/r3q/latency_kernel_trace.csv ) && python3 scripts/trace_one_call.py gpurun_out/r3q/latency_kernel_trace.csv > gpurun_out/r3q/single_query_timeline.txt && cat gpurun_out/r3q/single_query_timeline.txt
timeout -k 10 400 python bench.py > gpurun_out/r3q/bench_default.json 2> gpurun_out/r3q/bench_default.err; echo "bench rc=$?"
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r3q/gpu_tests.log 2>&1; tail -3 gpurun_out/r3q/gpu_tests.log
