#!/bin/bash
# does a larger chunk (more calls gathered per launch) pay?  ms per call against --chunk
set -o pipefail
O=gpurun_out/chunk; mkdir -p $O
for cfg in "lookup_mixed 2048 0" "lookup_mixed 2048 4096" "lookup_mixed 2048 8192" "atms_with_lookups 2048 0" "atms_with_lookups 2048 4096" "atms_with_lookups 2048 8192" "simple_mul 512 0" "simple_mul 512 8192" "lookup_mixed 256 0" "lookup_mixed 256 4096"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --workload $1 --batch $2 --chunk $3 --no-cpu-baseline --no-rlc-secondary --no-alone --steps 480 > $O/$1_$2_$3.json 2> $O/$1_$2_$3.err || { tail -5 $O/$1_$2_$3.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/$1_$2_$3.json')); print('$1 x $2 chunk $3', d['value'], d['ms_per_step'], d['config'].get('calls_coalesced_per_launch'))"
done
