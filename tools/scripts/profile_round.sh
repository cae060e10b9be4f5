#!/bin/bash
# One profile set on the GPU box (run through gpurun from the repo root):
#   tools/scripts/profile_round.sh TAG WORKLOAD BATCH MODE [extra bench.py flags]
# e.g. tools/scripts/profile_round.sh r02_v24 simple_mul 4096 per-proof
# writes, under gpurun_out/ (copy the ones to keep into profiles/):
#   TAG_bench.json                the bench line of the default-length run (CPU baseline included)
#   TAG_bench_under_rocprof.json  the line of the profiled run (--timed-only: the same launches rocprofv3 averages over)
#   TAG_kernel_stats.csv          rocprofv3 --kernel-trace --stats of that command
#   TAG_pmc_summary.txt           mean counters per launch, from separate --pmc passes (tools/pmc_summary.py); the first
#                                 line records what was profiled - bench.py only attaches traffic from a matching file
TAG=$1; WL=$2; BATCH=$3; MODE=$4; shift 4
ARGS="--workload $WL --batch $BATCH --mode $MODE --no-rlc-secondary $*"
O=gpurun_out
mkdir -p $O
timeout -k 10 400 python bench.py $ARGS > $O/b_$TAG.log 2>&1; grep "^{" $O/b_$TAG.log | tail -1 > $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the profiled command: the steps-in-flight count the run above settled on, and nothing but warm-up + timed steps, so
# that rocprofv3's per-kernel averages and the line's event-timed kernel_ms cover the same launches
INF=$(python -c "import json,sys; print(json.load(open('$O/${TAG}_bench.json'))['config']['steps_in_flight'])")
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 bench.py $ARGS --steps 60 --warmup 5 --inflight $INF --timed-only > $O/prof_$TAG.log 2>&1
grep "^{" $O/prof_$TAG.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${TAG}_$n -- python3 bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --hint $INF > $O/pmc_${TAG}_$n.log 2>&1 || exit 1
done
{ echo "# workload=$WL batch=$BATCH mode=$MODE tag=$TAG (rocprofv3 --pmc, separate passes; bench.py $ARGS --steps 2 --warmup 1 --inflight 1 --hint $INF)"; python tools/pmc_summary.py $O/pmc_${TAG}_*; } > $O/${TAG}_pmc_summary.txt
grep -E "INSTS_VALU|FETCH|WRITE_SIZE|WAIT_ANY|WAVE_CYCLES" $O/${TAG}_pmc_summary.txt
cat $O/prof_$TAG/*/*kernel_stats.csv | head -12
cp $O/prof_$TAG/*/*kernel_stats.csv $O/${TAG}_kernel_stats.csv
cut -c1-300 $O/${TAG}_bench.json
