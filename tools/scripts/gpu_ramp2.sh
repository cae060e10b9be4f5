#!/bin/bash
# short timed regions (the driver's 20 steps) against the number of lanes
set -o pipefail
out=gpurun_out/ramp2; mkdir -p $out; : > $out/summary.txt
for lanes in 4 6 8 10 12 16; do for k in 20 20 240; do
  python3 bench.py --steps $k --warmup 5 --lanes $lanes --no-cpu-baseline --no-rlc-secondary --no-alone --no-tune > $out/l${lanes}_$k.json 2> $out/l${lanes}_$k.err || exit 1
  python3 -c "import json; d=json.load(open('$out/l${lanes}_$k.json')); print('lanes', $lanes, 'steps', $k, d['ms_per_step'], round(d['ms_per_step']*$k,2), d['value'])" | tee -a $out/summary.txt
done; done
