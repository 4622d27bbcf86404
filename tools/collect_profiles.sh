#!/bin/bash
# Run on the GPU box (via gpurun): PMC traffic passes, then the benchmark (which reads the fresh traffic file),
# then rocprofv3 kernel traces of the same command (default: 8 ring slots per launch, whose per-kernel duration is
# roofline.launch_ms of the bench line; and one slot per launch, comparable with single_batch_launches.ms_per_step).
# Outputs under gpurun_out/$1/ ; the summaries are copied into profiles/ with the tag as prefix.
set -u
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# counters in their own runs (no tracing), one counter per pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 40 --warmup 8 --repeats 1 --no-single --no-host-fed --no-cpu-baseline --no-model --no-train > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch pass done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 40 --warmup 8 --repeats 1 --no-single --no-host-fed --no-cpu-baseline --no-model --no-train > /dev/null 2> $OUT/pmc_write.err
echo "pmc passes done" >> $OUT/progress.txt
python3 tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
cp $OUT/pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # bench.py reads the newest profiles/*_pmc_traffic.json
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; echo "bench done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pof_trace_$TAG -- python3 bench.py --steps 24 --warmup 8 --no-single --no-host-fed --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
cp $(find /tmp/pof_trace_$TAG -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pof_trace1_$TAG -- python3 bench.py --steps 24 --warmup 8 --slots 1 --no-cpu-baseline --no-extra > $OUT/bench_traced_single_slot.json 2> $OUT/trace1.err
cp $(find /tmp/pof_trace1_$TAG -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_single_slot.csv
# configs[3] training step as one graph replay (HIP units / library modules), the ordered kernels of one eager step, and
# the Prototype's training + inference trace (no library convolution kernel may appear)
cd /tmp
for m in hip modules; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pof_bh_${m}_$TAG -- python3 $GRAFT_REPO_ROOT/tools/trace_boxhead.py $m > $GRAFT_REPO_ROOT/$OUT/boxhead_$m.log 2>&1
  cp $(find /tmp/pof_bh_${m}_$TAG -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$OUT/boxhead_graphed_${m}_kernel_stats.csv
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/pof_bhe_$TAG -- python3 $GRAFT_REPO_ROOT/tools/trace_boxhead_eager.py hip > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_order.py $(find /tmp/pof_bhe_$TAG -name "*kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/$OUT/boxhead_step_order.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pof_pt_$TAG -- python3 $GRAFT_REPO_ROOT/tools/trace_prototype.py > $GRAFT_REPO_ROOT/$OUT/prototype_trace.log 2>&1
cp $(find /tmp/pof_pt_$TAG -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$OUT/prototype_kernel_stats.csv
cd $GRAFT_REPO_ROOT
echo "model traces done" >> $OUT/progress.txt
(lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; nproc; rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8) > $OUT/host.txt
cat $OUT/bench.json | head -c 3000
