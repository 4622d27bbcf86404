"""Per-kernel timeline of the LAST graph replay in a rocprofv3 kernel trace: start offset, duration and the idle gap
before each kernel.  usage: trace_gaps.py <kernel_trace.csv> <kernels per replay>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    busy += e - s
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:70]))
    prev_end = e
tot = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
print("replay span %.1f us, kernels busy %.1f us, idle %.1f us" % (tot, busy / 1e3, tot - busy / 1e3))
