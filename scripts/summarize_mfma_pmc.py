"""summary of scripts/profile_mfma.sh's output (kernel trace + two --pmc passes of the shared sweep on the matrix cores):
  python scripts/summarize_mfma_pmc.py gpurun_out/<dir> [profiles/rNN_mfma_pmc.json] [dim] [queries per launch]
The dominant launches are the bbq_scan_mfma_kernel launches with the largest grid; their tile count is taken from SQ_INSTS_MFMA
(30 MFMAs per 64-row tile and 32 queries at 768-d with 4-bit queries: 2 row groups x (12 of the contraction + 3 of the start values)),
so the summary does not depend on how many chunks a workgroup walks; "per tile and wave" is per tile AND group of 32 queries (a workgroup
serves two groups per tile load since round 4)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out = {"how": "scripts/profile_mfma.sh: rocprofv3 --kernel-trace --stats and two --pmc passes (8 SQ counters each) of "
              "`python3 bench.py --steps 2 --warmup 1 ...` (10 M x 768, 64 queries per launch chain, 32 per matrix-core group); the dominant launches are the ones with the largest grid"}
for p in ("pmc1", "pmc2"):
    fs = glob.glob(os.path.join(d, p, "*", "*counter_collection.csv"))
    if not fs:
        continue
    rows = [r for f in fs for r in csv.DictReader(open(f)) if "mfma" in r["Kernel_Name"]]   # (child processes of the run leave files of their own)
    g = max(int(r["Grid_Size"]) for r in rows)
    big = [r for r in rows if int(r["Grid_Size"]) == g]
    n = len({r["Dispatch_Id"] for r in big})
    agg = collections.defaultdict(float)
    for r in big:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    out.setdefault("per_launch", {}).update({c: v / n for c, v in agg.items()})
    out["grid_work_items"], out["launches_" + p] = g, n
    out["kernel"], out["vgpr_count_field"], out["lds_bytes"], out["scratch_bytes"] = big[0]["Kernel_Name"], int(big[0]["VGPR_Count"]), int(big[0]["LDS_Block_Size"]), int(big[0]["Scratch_Size"])
fs = glob.glob(os.path.join(d, "trace", "*", "*kernel_trace.csv"))
tr = [r for f in fs for r in csv.DictReader(open(f)) if "mfma" in r["Kernel_Name"]]
g = max(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) for r in tr)
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == g]
out["trace_avg_us"], out["trace_launches"] = sum(us) / len(us), len(us)
pl = out["per_launch"]
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 768
queries = int(sys.argv[4]) if len(sys.argv) > 4 else 64   # per launch: two groups of 32 share every tile load (32 before round 4's two-group kernel)
# MFMAs per 64-row tile and 32 queries: 2 row groups x (dim / 64 of the contraction in its FP6 x FP4 form + 3 of the start values)
tiles = pl["SQ_INSTS_MFMA"] / (2 * (dim / 64 + 3))
out["rows_per_launch"], out["tiles_per_launch"] = tiles * 64, tiles
out["per_tile_and_wave"] = {k_: pl[k_] / tiles for k_ in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU") if k_ in pl}
# the clock under the counters: SQ_BUSY_CYCLES is summed over the chip's 32 shader engines (8 XCDs x 4), and the SQs are busy for the whole
# launch - 1.8 GHz for this kernel, not the 2.4 GHz of the data sheet (a fixed 2.1 GHz here once made vector and matrix time look
# like they add up to the whole launch).  Cross-check: SQ_VALU_MFMA_BUSY_CYCLES per MFMA must be the instructions' own 32 / 64 cycles.
clock_hz = pl["SQ_BUSY_CYCLES"] / 32 / (out["trace_avg_us"] * 1e-6) if pl.get("SQ_BUSY_CYCLES") else 0.0
if not 1.2e9 <= clock_hz <= 2.5e9:   # (a pass whose launches overlapped with another kernel's: take what the clean passes of this kernel gave)
    clock_hz = 1.8e9
simd_cycles = out["trace_avg_us"] * 1e-6 * 1024 * clock_hz
valu = pl.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles
mfma = pl.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles
out["derived"] = {
    "clock_GHz_from_SQ_BUSY_CYCLES": clock_hz / 1e9,
    "mfma_busy_cycles_per_mfma": pl.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(pl.get("SQ_INSTS_MFMA", 1), 1),
    "cycles_per_tile_and_group": simd_cycles / tiles,
    "valu_active_share_of_simd_time": valu,
    "mfma_busy_share_of_simd_time": mfma,
    "overlap_of_the_two_at_least": max(0.0, valu + mfma - 1.0),
    "resident_waves_per_simd": pl.get("SQ_WAVE_CYCLES", 0) * 4 / simd_cycles,
    "cycles_per_valu_instruction": pl.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(pl.get("SQ_INSTS_VALU", 1), 1),
    "wave_time_split": {k_: pl.get(k_, 0) / max(pl.get("SQ_WAVE_CYCLES", 1), 1) for k_ in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
    "lds_bank_conflict_share": pl.get("SQ_LDS_BANK_CONFLICT", 0) / max(pl.get("SQ_LDS_IDX_ACTIVE", 1), 1),
    "queries_per_launch": queries,
    "queries_per_s_of_this_launch_alone": queries / (out["trace_avg_us"] * 1e-6),
    "note": "SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md), SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles; "
            "SIMD time = launch duration x 1024 SIMDs x the clock derived above",
}
if len(sys.argv) > 2 and sys.argv[2] != "-":
    json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["per_tile_and_wave"]), json.dumps(out["derived"], indent=1), out["trace_avg_us"], out.get("scratch_bytes"), out.get("vgpr_count_field"))
