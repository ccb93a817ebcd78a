"""rewrites the lines of profiles/README.md that quote numbers of the round's committed evidence (rNN_bench_n1.json, rNN_*_dominant_launch.json):
  python scripts/refresh_profiles_readme.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(ROOT, "profiles")
d = json.loads(open(os.path.join(P, rnd + "_bench_n1.json")).read().strip().splitlines()[-1])
h = json.load(open(os.path.join(P, rnd + "_headline_dominant_launch.json")))
c = json.load(open(os.path.join(P, rnd + "_c2_dominant_launch.json")))
path = os.path.join(P, "README.md")
s = open(path).read()


def replace_row(text, key, new_row):
    a = text.index("| `%s` |" % key)
    b = text.index("\n", a) + 1
    return text[:a] + new_row + text[b:]


s = replace_row(s, rnd + "_headline_dominant_launch.json",
                "| `%s_headline_dominant_launch.json` | the %d largest-segment launches of `bbq_scan_kernel<4, 6, 2, 1>` out of the trace: **avg %.0f µs for 25.29 GB = %.2f TB/s = %.3f of 8 TB/s** "
                "(bench.py's hipEvent average of the same profiled run: %.2f TB/s), 28 %% of the launch's row range served by the Infinity Cache (DESIGN §2); PMC: FETCH×2 + WRITE = **%.1f %%** of the "
                "algorithmic bytes (bytes that left the L2s: the Infinity Cache sits behind that counter). `vgpr_count_field_of_the_trace` is rocprofv3's field (28); the compiler's "
                "kernel-resource-usage remark says 50 VGPRs, occupancy 8, no scratch |\n"
                % (rnd, h["launches"], h["trace_avg_us"], h["trace_avg_GBps"] / 1e3, h["trace_avg_GBps"] / 8000, h["hipEvent_GBps_same_profiled_run"] / 1e3, 100 * h["pmc"]["traffic_over_algorithmic"]))
s = replace_row(s, rnd + "_c2_*",
                "| `%s_c2_*` | the same for BASELINE config 2 (1 M × 768, the whole index cache-resident; 64 queries per launch since the sub-batch also follows the call size): "
                "**%.2f TB/s = %.3f of the HBM peak** trace average over %d launches of %.2f ms - algorithmic bytes served by the Infinity Cache (FETCH×2 + WRITE = %.1f %%, see above) |\n"
                % (rnd, c["trace_avg_GBps"] / 1e3, c["trace_avg_GBps"] / 8000, c["launches"], c["trace_avg_us"] / 1e3, 100 * c["pmc"]["traffic_over_algorithmic"]))
cf, ho, na = d["configs"], d["hbm_only"], d["napi"]
s = replace_row(s, rnd + "_bench_n1.json",
                "| `%s_bench_n1.json` | `python bench.py` (un-profiled, the driver's default invocation, ≈ 60 s): %.2f K q/s (%.3f of the HBM peak end to end), dominant launch %.3f by HIP events, "
                "%.0f %% of the bytes a query sweeps cache-resident (`roofline.cache_resident_frac_of_sweep`), full-size parity true; **`hbm_only`** (the same step with `resident_mb` 0: everything "
                "streamed from HBM): %.2f K q/s, %.3f end to end, dominant launch %.3f, identical answers; **`configs`**: c2 %.1f K q/s (%.3f end to end, kernel %.3f, cache-resident), c4 %.2f K (%.3f), "
                "c5 %.1f K (%.3f), each with its own full-size parity check; **`latency`** p50 %.3f ms; **`raw_queries`** %.2f K q/s (`bbq_search_raw_batch`); **`napi`**: node over the saved 10 M-row "
                "index, %.2f K q/s batched (512 raw queries per call, %.1f of %.1f ms inside the addon) / %.3f ms per single call, identical to ctypes; `batched` %.1f K q/s with its roofline; "
                "`inprocess_multi` %.3f of the single index (4 shards on the one GPU); recall@100 %.3f; CPU baseline %.3f q/s |\n"
                % (rnd, d["value"] / 1e3, d["end_to_end_hbm_frac"], d["roofline"]["frac"], 100 * d["roofline"]["cache_resident_frac_of_sweep"], ho["value"] / 1e3, ho["end_to_end_hbm_frac"],
                   ho["roofline_frac_dominant_launch"], cf["c2"]["value"] / 1e3, cf["c2"]["end_to_end_hbm_frac"], cf["c2"]["roofline"]["frac"], cf["c4"]["value"] / 1e3,
                   cf["c4"]["end_to_end_hbm_frac"], cf["c5"]["value"] / 1e3, cf["c5"]["end_to_end_hbm_frac"], d["latency"]["p50_ms"], d["raw_queries"]["value"] / 1e3, na["value"] / 1e3,
                   na["batch_ms_per_call_inside_addon"], na["batch_ms_per_call"], na["p50_ms"], d["batched"]["value"] / 1e3, d["inprocess_multi"]["vs_single_index"], d["recall_at_100"],
                   d["cpu_baseline"]["value"]))
open(path, "w").write(s)
print("profiles/README.md refreshed from", rnd)
