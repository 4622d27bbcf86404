#!/bin/bash
# PMC passes over the headline launch (run on the GPU box): one counter group per pass, no tracing.
# usage: tools/pmc_headline.sh OUTDIR
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GROUPS_=("VALUBusy" "MemUnitStalled" "WriteUnitStalled" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY")
for grp in "${GROUPS_[@]}"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/$tag.err || echo "pass $tag failed"
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith(("scan_", "cutout")):
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s mean %14.2f  n %d" % (c, sum(v) / len(v), len(v)))
PY
