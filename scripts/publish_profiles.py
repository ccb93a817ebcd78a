"""copies one round's evidence from gpurun_out/ (scratch) into profiles/ (committed) under the round's names:
  python scripts/publish_profiles.py r03
expects gpurun_out/prof_headline, prof_c2 (scripts/collect_profiles.sh), gpurun_out/<evidence dir>/ (scripts/round_evidence.sh)
and gpurun_out/<mfma dir>/ (scripts/profile_mfma.sh); the *_dominant_launch.json summaries and dominant_kernel.json come from
scripts/summarize_profiles.py, profiles/rNN_mfma_pmc.json from this script."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
ev = sys.argv[2] if len(sys.argv) > 2 else "r3q"
mf = sys.argv[3] if len(sys.argv) > 3 else "r3p"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))
        print("profiles/" + dst)


for tag in ("headline", "c2"):
    d = os.path.join(G, "prof_" + tag)
    cp(os.path.join(d, "kernel_stats.csv"), "%s_%s_kernel_stats.csv" % (rnd, tag))
    cp(os.path.join(d, "pmc_fetch_full.csv"), "%s_%s_pmc_fetch_size.csv" % (rnd, tag))
    cp(os.path.join(d, "pmc_write_full.csv"), "%s_%s_pmc_write_size.csv" % (rnd, tag))
    cp(os.path.join(d, "bench_under_trace.json"), "%s_%s_bench_under_trace.json" % (rnd, tag))
    if tag == "headline":
        cp(os.path.join(d, "kernel_trace_full.csv"), "%s_%s_kernel_trace.csv" % (rnd, tag))
cp(os.path.join(G, ev, "single_query_timeline.txt"), "%s_single_query_timeline.txt" % rnd)
cp(os.path.join(G, ev, "bench_default.json"), "%s_bench_n1.json" % rnd)

# the shared sweep on the matrix cores: counters of its dominant launches (largest grid), per launch and per 64-row tile and wave
out = {"how": "scripts/profile_mfma.sh: rocprofv3 --kernel-trace --stats and two --pmc passes (8 SQ counters each) of "
              "`python3 bench.py --steps 2 --warmup 1 ...` (10 M x 768, 32 queries per shared sweep); the dominant launches are the ones with the largest grid"}
for d in ("pmc1", "pmc2"):
    fs = glob.glob(os.path.join(G, mf, d, "*", "*counter_collection.csv"))
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "mfma" in r["Kernel_Name"]]
    g = max(int(r["Grid_Size"]) for r in rows)
    big = [r for r in rows if int(r["Grid_Size"]) == g]
    n = len({r["Dispatch_Id"] for r in big})
    agg = collections.defaultdict(float)
    for r in big:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    out.setdefault("per_launch", {}).update({c: v / n for c, v in agg.items()})
    out["grid_work_items"], out["launches_" + d] = g, n
    out["kernel"], out["vgpr_count_field"], out["lds_bytes"], out["scratch_bytes"] = big[0]["Kernel_Name"], int(big[0]["VGPR_Count"]), int(big[0]["LDS_Block_Size"]), int(big[0]["Scratch_Size"])
fs = glob.glob(os.path.join(G, mf, "trace", "*", "*kernel_trace.csv"))
if fs and "per_launch" in out:
    tr = [r for r in csv.DictReader(open(fs[0])) if "mfma" in r["Kernel_Name"]]
    g = max(int(r["Grid_Size_X"]) for r in tr)
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if int(r["Grid_Size_X"]) == g]
    out["trace_avg_us"], out["trace_launches"] = sum(us) / len(us), len(us)
    pl = out["per_launch"]
    # a workgroup = 512 work-items, persistent over 8 chunks of 512 rows; a tile = 64 rows of one wave
    tiles = g / 512 * 8 * 8
    out["rows_per_launch"], out["tiles_per_launch"] = tiles * 64, tiles
    out["per_tile_and_wave"] = {k_: pl[k_] / tiles for k_ in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU") if k_ in pl}
    simd_cycles = out["trace_avg_us"] * 1e-6 * 1024 * 2.1e9   # 1024 SIMDs at the ~2.1 GHz the chip holds under this load
    out["derived"] = {
        "valu_active_share_of_simd_time": pl.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles,
        "mfma_busy_share_of_simd_time": pl.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles,
        "cycles_per_valu_instruction": pl.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(pl.get("SQ_INSTS_VALU", 1), 1),
        "wave_time_split": {k_: pl.get(k_, 0) / max(pl.get("SQ_WAVE_CYCLES", 1), 1) for k_ in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
        "lds_bank_conflict_share": pl.get("SQ_LDS_BANK_CONFLICT", 0) / max(pl.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        "note": "SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md), SQ_VALU_MFMA_BUSY_CYCLES cycles; SIMD time = "
                "launch duration x 1024 SIMDs x 2.1 GHz",
    }
    json.dump(out, open(os.path.join(P, "%s_mfma_pmc.json" % rnd), "w"), indent=1)
    print("profiles/%s_mfma_pmc.json" % rnd)
    print(json.dumps(out["per_tile_and_wave"]), json.dumps(out["derived"]))
