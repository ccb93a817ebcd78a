"""timeline of the LAST single-query call in a rocprofv3 kernel trace of scripts/latency.py (kernel durations and the gaps
between them): python scripts/trace_one_call.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# a call starts with its first launch: the pre-sampling kernel of the latency path (round 3), or - the general path - the dense
# first-segment sweep (scan kernel MODE 1 / 3) and the host-to-device copy right before it
def is_dense(name):
    if "bbq_lat_pre_kernel" in name:
        return True
    if "bbq_scan_kernel<" not in name:
        return False
    mode = name.split("bbq_scan_kernel<")[1].split(">")[0].split(",")[2].strip()
    return mode in ("1", "3")


starts = [i for i, e in enumerate(ev) if is_dense(e[2])]
if len(starts) < 3:
    raise SystemExit("fewer than three calls in the trace")
a, b = starts[-2], starts[-1]
while a > 0 and "copyBuffer" in ev[a - 1][2] and ev[a][0] - ev[a - 1][1] < 30_000:
    a -= 1
while b > a and "copyBuffer" in ev[b - 1][2] and ev[b][0] - ev[b - 1][1] < 30_000:
    b -= 1
call = ev[a:b]
t0 = call[0][0]
prev_end = t0
busy = 0
for s, e, name in call:
    short = name.split("(")[0].replace("void bbq::", "")[:70]
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short))
    busy += e - s
    prev_end = e
print("kernels %d, first start -> last end %.1f us, busy %.1f us" % (len(call), (call[-1][1] - t0) / 1e3, busy / 1e3))
