"""Per-kernel sums of SQ counters from rocprofv3 --pmc CSV output directories (arguments)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], d)
            if key not in seen:
                seen.add(key)
                calls[(k, d)] += 1
for k, c in acc.items():
    if "conv3" not in k:
        continue
    print(k)
    for name, v in sorted(c.items()):
        print("   %-28s %.4g" % (name, v))
