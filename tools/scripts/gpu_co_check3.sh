#!/bin/bash
set -o pipefail
O=gpurun_out/cocheck3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/t.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t.log
for cfg in "lookup_atms_mixed 2048" "lookup_atms_mixed 2048 --shared-workspace" "sha256 1024" "secp256k1 512" "sha256 128" "secp256k1 64" "simple_mul 4096"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --workload $1 --batch $2 $3 --no-cpu-baseline --no-alone --steps 240 > $O/$1_$2$3.json 2> $O/$1_$2$3.err || { tail -5 $O/$1_$2$3.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/$1_$2$3.json')); print('$1 x $2 $3', d['value'], d['ms_per_step'], d['config'].get('calls_coalesced_per_launch'), (d.get('rlc_mode') or {}).get('value'))"
done
