#!/bin/bash
set -o pipefail
O=gpurun_out/lanemap; mkdir -p $O
run() { n=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --steps 240 --no-alone > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], (d.get('rlc_mode') or {}).get('value'), [(t['ms_per_call_launchers_rule']) for t in (d['config'].get('tuned_launch_shapes') or [])])"; }
run mixed1 --workload lookup_atms_mixed --no-rlc-secondary
run mixed2 --workload lookup_atms_mixed --no-rlc-secondary
run simple
run sha256_128 --workload sha256 --batch 128 --no-rlc-secondary
run secp_64 --workload secp256k1 --batch 64 --no-rlc-secondary
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tuned_shapes or chunking or mixed or null_stream" 2>&1 | tail -3
