"""prints the last N kernels of a rocprofv3 kernel trace (csv) with durations and the gaps between them"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -14:]
t0 = int(tail[0]["Start_Timestamp"])
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-56s start %8.1f us  dur %7.1f us  gap %6.1f us" % (r["Kernel_Name"][:56], (s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3 if prev else 0)))
    prev = e
