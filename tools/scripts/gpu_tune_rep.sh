#!/bin/bash
# is the tuner's choice reproducible?  the small shapes, three runs each
set -o pipefail
O=gpurun_out/tunerep; mkdir -p $O
for cfg in "secp256k1 256" "secp256k1 512" "sha256 512"; do set -- $cfg
 for r in 1 2 3 4; do
  timeout -k 10 300 python3 bench.py --workload $1 --batch $2 --no-cpu-baseline --no-rlc-secondary --no-alone --steps 240 > $O/$1_$2_$r.json 2> $O/$1_$2_$r.err || { tail -3 $O/$1_$2_$r.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/$1_$2_$r.json')); t=d['config']['tuned_launch_shapes'][0]; print('$1 x $2', d['value'], d['ms_per_step'], 'engine', t['pairing_engine'], 'tpl', t['msm_terms_per_lane'], t['ms_per_call_launchers_rule'], t['ms_per_call_chosen'], t['configurations_measured'])"
 done; done
