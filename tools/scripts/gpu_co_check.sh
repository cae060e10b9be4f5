#!/bin/bash
set -o pipefail
O=gpurun_out/cocheck; mkdir -p $O
timeout -k 10 300 python3 bench.py --workload sha256 --batch 128 --steps 240 > $O/sha256_128.json 2> $O/sha256_128.err || { tail -5 $O/sha256_128.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/sha256_128.json')); print(d['value'], d['ms_per_step'], d['config'].get('calls_coalesced_per_launch'), d['roofline']['frac'], d.get('rlc_mode',{}).get('value'), d['cpu_baseline']['value'])"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/default20.json 2> $O/default20.err || { tail -5 $O/default20.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/default20.json')); print(d['value'], d['ms_per_step'], d['config'].get('calls_coalesced_per_launch'))"
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/t.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t.log
timeout -k 10 400 python3 tests/soak.py --minutes 3 --threads 4 --seed 21 2>&1 | tail -3
