#!/bin/bash
# Run on the GPU box (via gpurun): PMC traffic passes, then the benchmark (which reads the fresh
# traffic file), then the rocprofv3 kernel trace of the same command.
# Outputs under gpurun_out/$1/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# counters in their own runs (no tracing), one counter per pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model > /dev/null 2> $OUT/pmc_write.err
python3 tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
cp $OUT/pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # bench.py reads the newest profiles/*_pmc_traffic.json
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pof_trace_$TAG -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
cp $(find /tmp/pof_trace_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core" > $OUT/host_cpu.txt
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/host_cpu.txt
cat $OUT/bench.json
