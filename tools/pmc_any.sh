#!/bin/bash
# usage: tools/pmc_any.sh OUTDIR KERNEL_PREFIX "GROUP1;GROUP2;..." script.py [args...]   (counter names separated by spaces inside a group)
set -u
OUT=$1; PREFIX=$2; GROUPS_STR=$3; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IFS=';' read -ra GROUPS_ <<< "$GROUPS_STR"
for grp in "${GROUPS_[@]}"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 "$@" > /dev/null 2> $OUT/$tag.err || echo "pass $tag failed: $(tail -2 $OUT/$tag.err)"
done
python3 - $OUT "$PREFIX" <<'PY'
import sys, glob, csv, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith(tuple(sys.argv[2].split(","))):
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %16.2f  n %d" % (c, sum(v) / len(v), len(v)))
PY
