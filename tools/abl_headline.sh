#!/bin/bash
# Headline ablations (run on the GPU box): slots per launch, detections per sample, params rows on / off, and the
# driver's own command line.  usage: tools/abl_headline.sh > gpurun_out/abl.txt
line() {
  python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline())
except Exception as e:
    print('%-52s FAILED' % sys.argv[1]); sys.exit(0)
r=d['roofline']; f=d['roofline_f64']; s=d.get('single_batch_launches',{})
print('%-52s f32: %6.2f us/step wall (%.3f)  %6.2f ev | f64: %6.2f us (%.3f) | 1/launch: %6.2f us' % (sys.argv[1], d['ms_per_step']*1e3, r['frac'], r['events']['ms_per_step']*1e3, f['ms_per_step']*1e3, f['frac'], s.get('ms_per_step',0)*1e3))
" "$1"
}
B="python bench.py --no-extra --no-cpu-baseline"
for slots in 1 2 4 8; do $B --steps 400 --warmup 40 --slots $slots 2>/dev/null | line "steps 400 slots $slots"; done
$B --steps 400 --warmup 40 --slots 8 --no-params 2>/dev/null | line "steps 400 slots 8 no params rows"
for d in 0 3 8; do $B --steps 400 --warmup 40 --slots 8 --dets-per-sample $d 2>/dev/null | line "steps 400 slots 8 dets/sample $d"; done
for d in 0 3 8; do $B --steps 400 --warmup 40 --slots 8 --no-params --dets-per-sample $d 2>/dev/null | line "steps 400 slots 8 no params dets/sample $d"; done
for i in 1 2 3; do $B --steps 20 --warmup 5 2>/dev/null | line "DRIVER steps 20 warmup 5 (defaults) run $i"; done
