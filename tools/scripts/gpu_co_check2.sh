#!/bin/bash
set -o pipefail
O=gpurun_out/cocheck; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_rlc.py -x -q -m gpu -k "coalesced or chunking or mixed_batch or rlc" > $O/t2.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t2.log
for cfg in "sha256 128" "secp256k1 64" "simple_mul 512"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --workload $1 --batch $2 --mode rlc --no-cpu-baseline --no-alone --steps 480 > $O/rlc_$1_$2.json 2> $O/rlc_$1_$2.err || { tail -5 $O/rlc_$1_$2.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/rlc_$1_$2.json')); print('rlc $1 x $2', d['value'], d['ms_per_step'])"
done
timeout -k 10 400 python3 tests/soak.py --minutes 2.5 --threads 4 --seed 31 --forms device_rlc,device,multi,host_rlc 2>&1 | tail -2
