#!/bin/bash
# what the pipeline's fill and drain cost a short timed region: the same step at 20 / 40 / 80 / 240 timed steps, and the
# kernel trace of the 20-step form (tools/trace_timeline.py reads it)
set -o pipefail
out=gpurun_out/ramp; mkdir -p $out
for k in 20 20 40 80 240; do
  python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-rlc-secondary --no-alone --no-tune > $out/steps_$k.json 2> $out/steps_$k.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$out/steps_$k.json')); print($k, d['ms_per_step'], round(d['ms_per_step']*$k,2), d['value'])" | tee -a $out/summary.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/trace -o t20 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rlc-secondary --no-alone --no-tune --timed-only > $GRAFT_REPO_ROOT/$out/trace.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py $f 40 > $out/timeline_last40ms.txt
wc -l $f
