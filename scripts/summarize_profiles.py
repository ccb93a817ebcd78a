"""Turns one collect_profiles.sh run (gpurun_out/prof*/) into the committed evidence of a round:

  profiles/rNN_<tag>_dominant_launch.json   the dominant scan launches picked out of the kernel trace (average / min / max, GB/s)
                                            next to bench.py's own hipEvent average of the same profiled run
  profiles/dominant_kernel.json             registry bench.py reads: per (rows, dim, indexBits, queryBits, bytes/row) the kernel-trace
                                            average and the PMC traffic per row, with where and how they were collected.  bench.py
                                            quotes an entry only for the kernel and layout it is actually running.

  python scripts/summarize_profiles.py <prof dir> <round, e.g. r02> <tag, e.g. headline>

HBM bytes from the PMC passes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE in separate
passes, KiB per dispatch; on gfx950 FETCH_SIZE tallies the 128-byte requests of a 16-B/lane coalesced streaming read at 64 bytes,
so it is doubled before it is compared with a byte count; WRITE_SIZE is exact for streaming stores.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dominant(rows, name_key, metric):
    scans = [r for r in rows if "bbq_scan_kernel" in r[name_key]]
    if not scans:
        raise SystemExit("no bbq_scan_kernel dispatch in the trace")
    biggest = max(metric(r) for r in scans)
    return [r for r in scans if metric(r) == biggest]


def main():
    prof, rnd, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.loads(open(os.path.join(prof, "bench_under_trace.json")).read().strip().splitlines()[-1])
    roof = bench["roofline"]
    trace = list(csv.DictReader(open(os.path.join(prof, "kernel_trace_full.csv"))))
    dom = dominant(trace, "Kernel_Name", lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))
    # warm-up launches of the same shape are part of the trace; bench.py's own figure covers the timed steps only, the trace all of them
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in dom]
    bpl = roof["bytes_per_launch"]
    out = {
        "kernel": "%s, largest-segment launches (grid %s x %s work-items)" % (dom[0]["Kernel_Name"], dom[0]["Grid_Size_X"], dom[0]["Grid_Size_Y"]),
        # rocprofv3's VGPR_Count field is not the compiler's register count (it reads 28 for the kernel that hipcc's
        # -Rpass-analysis=kernel-resource-usage reports at 50 VGPRs, occupancy 8, no scratch): kept under its own name
        "vgpr_count_field_of_the_trace": int(dom[0]["VGPR_Count"]), "vgprs_compiler_remark": 50 if "bbq_scan_kernel<4, 6, 2, 1>" in dom[0]["Kernel_Name"] else None,
        "lds_bytes": int(dom[0]["LDS_Block_Size"]), "scratch_bytes": int(dom[0]["Scratch_Size"]),
        "source": "kernel_trace of `rocprofv3 --kernel-trace --stats -- python3 bench.py %s` (scripts/collect_profiles.sh)" % bench.get("argv", ""),
        "workload": bench["config"]["workload"],
        "launches": len(us), "trace_avg_us": sum(us) / len(us), "trace_min_us": min(us), "trace_max_us": max(us),
        "algorithmic_bytes_per_launch": bpl,
        "trace_avg_GBps": bpl / (sum(us) / len(us) * 1e-6) / 1e9, "trace_min_us_GBps": bpl / (min(us) * 1e-6) / 1e9,
        "hipEvent_avg_ms_same_profiled_run": roof["avg_launch_ms"], "hipEvent_GBps_same_profiled_run": roof.get("achieved_algorithmic", roof.get("achieved")),
    }
    entry = {"rows": None, "dim": None, "index_bits": None, "query_bits": None, "bytes_per_row": bench["config"]["bytes_per_row"], "kernel": dom[0]["Kernel_Name"],
             "collected": "%s, %s" % (rnd, tag), "trace_avg_us": out["trace_avg_us"], "trace_launches": len(us), "trace_bytes_per_launch": bpl,
             "trace_avg_GBps": out["trace_avg_GBps"], "hbm_bytes_per_row": None, "pmc_how": None}
    # "<rows>x<dim>-dim <ib>-bit index, queryBits=<qb>, ..."
    w = bench["config"]["workload"]
    entry["rows"] = int(w.split("x")[0])
    entry["dim"] = int(w.split("x")[1].split("-")[0])
    entry["index_bits"] = int(w.split("-dim ")[1].split("-bit")[0])
    entry["query_bits"] = int(w.split("queryBits=")[1].split(",")[0])
    fpath, wpath = os.path.join(prof, "pmc_fetch_full.csv"), os.path.join(prof, "pmc_write_full.csv")
    if os.path.exists(fpath) and os.path.exists(wpath):
        fr = [r for r in csv.DictReader(open(fpath)) if r["Counter_Name"] == "FETCH_SIZE"]
        wr = [r for r in csv.DictReader(open(wpath)) if r["Counter_Name"] == "WRITE_SIZE"]
        fd = dominant(fr, "Kernel_Name", lambda r: int(r["Grid_Size"]))
        wd = dominant(wr, "Kernel_Name", lambda r: int(r["Grid_Size"]))
        fetch_kib = sum(float(r["Counter_Value"]) for r in fd) / len(fd)
        write_kib = sum(float(r["Counter_Value"]) for r in wd) / len(wd)
        bench_f = json.loads(open(os.path.join(prof, "bench_under_fetch.json")).read().strip().splitlines()[-1])
        bpl_f = bench_f["roofline"]["bytes_per_launch"]
        rows_f = bpl_f / bench_f["config"]["bytes_per_row"]
        traffic = fetch_kib * 1024 * 2 + write_kib * 1024
        out["pmc"] = {"fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib, "rows_x_queries_per_launch": rows_f,
                      "algorithmic_bytes_per_launch": bpl_f, "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / bpl_f,
                      "correction": "gfx950: FETCH_SIZE tallies the 128-B requests of 16-B/lane coalesced streaming reads at 64 B -> x2 "
                                    "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is"}
        entry["hbm_bytes_per_row"] = traffic / rows_f
        entry["pmc_how"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH x2 (gfx950 half-count of 16-B/lane streams) + WRITE"
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_%s_dominant_launch.json" % (rnd, tag)), "w"), indent=1)
    regp = os.path.join(ROOT, "profiles", "dominant_kernel.json")
    reg = json.load(open(regp)) if os.path.exists(regp) else {"entries": []}
    key = (entry["rows"], entry["dim"], entry["index_bits"], entry["query_bits"], entry["bytes_per_row"])
    if tag.endswith("_strict"):
        # the same launches traced with resident_mb 0 (nothing kept in the Infinity Cache): strict_* fields of the configuration's entry
        out["resident_mb"] = 0
        for e in reg["entries"]:
            if (e.get("rows"), e["dim"], e["index_bits"], e["query_bits"], e["bytes_per_row"]) == key:
                e["strict_trace_avg_us"], e["strict_trace_launches"], e["strict_trace_avg_GBps"] = out["trace_avg_us"], len(us), out["trace_avg_GBps"]
                e["strict_collected"] = "%s, %s" % (rnd, tag)
        json.dump(reg, open(regp, "w"), indent=1)
        print(json.dumps(out, indent=1))
        return
    reg["entries"] = [e for e in reg["entries"] if (e.get("rows"), e["dim"], e["index_bits"], e["query_bits"], e["bytes_per_row"]) != key] + [entry]
    json.dump(reg, open(regp, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
