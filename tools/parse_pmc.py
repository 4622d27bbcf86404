#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter per dispatch.
usage: parse_pmc.py DIR [DIR...] -> JSON on stdout"""
import collections, csv, glob, json, sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())}
       for k, cs in out.items() if k.startswith(("scan_", "cutout", "attn", "band", "nms", "rotate", "segment", "flow_"))}
print(json.dumps(res, indent=1))
