#!/bin/bash
# what the driver runs at round end (GPU tests, smoke, the bench at its own step count), then three minutes of the soak
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $O/t.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?"
python3 -c "import json; d=json.load(open('$O/bench20.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['rlc_mode']['value'])"
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python3 -c "import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['steps'])"
timeout -k 10 400 python3 tests/soak.py --minutes 3 --threads 4 --seed 41 2>&1 | tail -2
