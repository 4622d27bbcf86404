#!/bin/bash
# Counter evidence for the TRAINING kernels from the EAGER step (the hipGraph-captured step is left alone: in round
# 2 counter collection aborted its queue).  Run on the GPU box: tools/pmc_train_eager.sh OUTDIR
set -u
OUT=${1:-gpurun_out/r3_trainpmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 tools/bench_drspaam.py 8 train > /dev/null 2> $OUT/$tag.err || echo "pass $tag failed"
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith(("conv3", "bn_", "attn", "cutout")):
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(out.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %16.2f  n %d" % (c, sum(v) / len(v), len(v)))
PY
