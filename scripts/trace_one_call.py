"""timeline of the LAST single-query call in a rocprofv3 kernel trace of scripts/latency.py (kernel durations and the gaps
between them): python scripts/trace_one_call.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# a call = the kernels between two idle gaps of more than 100 us
calls, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > 100_000:
        calls.append(cur)
        cur = []
    cur.append(e)
calls.append(cur)
call = calls[-2] if len(calls) > 1 else calls[-1]
t0 = call[0][0]
prev_end = t0
busy = 0
for s, e, name in call:
    short = name.split("(")[0].replace("void bbq::", "")[:70]
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short))
    busy += e - s
    prev_end = e
print("kernels %d, first start -> last end %.1f us, busy %.1f us" % (len(call), (call[-1][1] - t0) / 1e3, busy / 1e3))
