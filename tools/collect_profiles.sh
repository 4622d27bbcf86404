#!/bin/bash
# Run on the GPU box (via gpurun): benchmark + rocprofv3 kernel trace + PMC traffic passes.
# Outputs under gpurun_out/$1/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
python3 tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core" > $OUT/host_cpu.txt
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/host_cpu.txt
cat $OUT/bench.json
