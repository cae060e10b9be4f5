#!/bin/bash
# mixed workload (two plans, two laned workspaces) against the size of the library's stream pool
set -o pipefail
O=gpurun_out/mixed_pool; mkdir -p $O
for pool in 16 24 32; do for rep in 1 2; do
  H2V_QUEUE_POOL=$pool timeout -k 10 300 python3 bench.py --workload lookup_atms_mixed --no-cpu-baseline --no-rlc-secondary --steps 240 --no-alone > $O/p${pool}_$rep.json 2> $O/p${pool}_$rep.err || { tail -5 $O/p${pool}_$rep.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/p${pool}_$rep.json')); print('pool', $pool, d['value'], d['ms_per_step'], [(t['ms_per_call_launchers_rule']) for t in d['config']['tuned_launch_shapes']])"
done; done
H2V_QUEUE_POOL=32 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-rlc-secondary --steps 240 --no-alone > $O/simple_p32.json 2> $O/simple_p32.err || { tail -5 $O/simple_p32.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/simple_p32.json')); print('simple_mul pool 32', d['value'], d['ms_per_step'])"
