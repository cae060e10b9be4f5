#!/bin/bash
# RLC mode, 120 steps: all valid, one reject per batch, 1 % rejects; then the RLC parity tests
set -o pipefail
O=gpurun_out/rlcq; mkdir -p $O
for v in "" "--reject-count 1" "--reject-fraction 0.01"; do
  n=$(echo "$v" | tr -d ' -.'); n=${n:-clean}
  timeout -k 10 300 python3 bench.py --mode rlc $v --no-cpu-baseline --no-rlc-secondary --steps 120 --no-alone > $O/$n.json 2> $O/$n.err || exit 1
  python3 -c "import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'])"
done
if [ "$1" = "tests" ]; then timeout -k 10 600 python -m pytest tests/test_rlc.py -x -q -m gpu > $O/t.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t.log; fi
