#!/bin/bash
# the per-GPU shares of BASELINE configs[3] / [4] cut over 2 / 4 / 8 GPUs, as a stream of calls on one GPU
set -o pipefail
O=gpurun_out/shares; mkdir -p $O
for cfg in "sha256 128" "sha256 256" "sha256 512" "secp256k1 64" "secp256k1 128" "secp256k1 256" "simple_mul 512"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --workload $1 --batch $2 --no-cpu-baseline --no-rlc-secondary --no-alone --steps 480 $EXTRA > $O/$1_$2.json 2> $O/$1_$2.err || { tail -5 $O/$1_$2.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/$1_$2.json')); print('$1 x $2', d['value'], d['ms_per_step'], d['config'].get('steps_in_flight'), d.get('kernel_ms'))"
done
