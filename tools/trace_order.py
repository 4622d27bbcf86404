"""Print the kernels of ONE eager box-head training step in launch order with durations, from a rocprofv3
--kernel-trace csv (argument: the *_kernel_trace.csv).  The traced program brackets the step with two marker
launches (a float64 fill of 12345 elements)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "FillFunctor<double>" in r["Kernel_Name"]]
lo, hi = marks[-2], marks[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo + 1:hi]:
    print("%9.1f us  %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:150]))
print("step span %.1f us, %d kernels" % ((int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, hi - lo - 1))
