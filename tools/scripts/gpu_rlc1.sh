#!/bin/bash
# where the one-reject RLC fall-back spends its time: kernel statistics of the timed steps, in flight and one call at a time
set -o pipefail
O=gpurun_out/rlc1; mkdir -p $O
A="--mode rlc --reject-count 1 --no-cpu-baseline --no-rlc-secondary"
timeout -k 10 300 python3 bench.py $A --steps 120 --no-alone > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py $A --steps 60 --warmup 5 --timed-only --no-alone > $O/prof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profa -- python3 bench.py $A --steps 20 --warmup 3 --timed-only --no-alone --pipeline streams --inflight 1 > $O/profa.log 2>&1 || exit 1
for d in prof profa; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; f=$(find $O/$d -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_trace.csv; done
rm -rf $O/prof $O/profa
python3 -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'])"
