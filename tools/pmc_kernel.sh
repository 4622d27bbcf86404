#!/bin/bash
# PMC passes for one driver script (run on the GPU box): one counter group per pass, no tracing.
# usage: tools/pmc_kernel.sh OUTDIR KERNEL_PREFIX script.py [args...]
set -u
OUT=$1; PREFIX=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GROUPS_=("VALUBusy" "MemUnitStalled" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE GRBM_COUNT")
for grp in "${GROUPS_[@]}"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 "$@" > /dev/null 2> $OUT/$tag.err || echo "pass $tag failed"
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
        print("   %-24s mean %16.2f  n %d" % (c, sum(v) / len(v), len(v)))
PY
