#!/bin/bash
# Headline kernel tuning sweep (run on the GPU box): threads per workgroup x runs per wave x slots per launch.
# usage: tools/tune_flat.sh [steps] > gpurun_out/tune_flat.txt
STEPS=${1:-400}
for tune in 128,1 64,1 256,1 128,2 64,2 256,2; do
  for slots in 1 4; do
    POF_FLAT_TUNE=$tune python bench.py --steps $STEPS --warmup 40 --slots $slots --no-extra --no-cpu-baseline 2>/dev/null | \
      python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']; f=d['roofline_f64']
print('tune %-6s slots %d  f32: %.2f us/step wall (%.3f)  %.2f us events (%.3f) | f64: %.2f us wall (%.3f) | epe %.2e %.2e' % ('$tune', $slots, d['ms_per_step']*1e3, r['frac'], r['events']['ms_per_step']*1e3, r['events']['frac'], f['ms_per_step']*1e3, f['frac'], d['epe_vs_oracle_m'], f['epe_vs_oracle_m']))
"
  done
done
