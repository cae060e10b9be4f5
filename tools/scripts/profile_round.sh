#!/bin/bash
# One profile set on the GPU box (run through gpurun from the repo root):
#   tools/scripts/profile_round.sh TAG WORKLOAD BATCH MODE [extra bench.py flags]
# e.g. tools/scripts/profile_round.sh r03_v40 simple_mul 4096 per-proof
# writes, under gpurun_out/ (copy the ones to keep into profiles/):
#   TAG_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the TIMED steps (bench.py --timed-only --no-alone: the
#                                 library's lanes keep the steps in flight, so these averages are overlapped durations)
#   TAG_kernel_stats_alone.csv    the same kernels ONE STEP AT A TIME with the launch shapes of the timed run
#                                 (--pipeline streams --inflight 1 --hint N): each kernel's own duration - what `roofline` uses
#   TAG_pmc_summary.txt           mean counters per launch from separate --pmc passes of that one-step-at-a-time command
#                                 (tools/pmc_summary.py); its first line records what was profiled, and bench.py only
#                                 attaches `traffic` from a file whose workload / batch / mode match
#   TAG_bench.json                the bench line of the default-length run (CPU baseline included), taken LAST, with the
#                                 PMC summary already in profiles/ on the box, so that roofline.traffic is never null
TAG=$1; WL=$2; BATCH=$3; MODE=$4; shift 4
ARGS="--workload $WL --batch $BATCH --mode $MODE --no-rlc-secondary $*"
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. a first line, only to learn the steps in flight / the hint of the timed run
# every step must exit 0: a profile taken from a process that crashed (round 3: SIGSEGV at exit after the CSV had been
# written) is not evidence, and the script stops there
fail() { echo "profile_round.sh: step $1 exited with status $2 (see $3)" >&2; tail -20 $3 >&2; exit 1; }
timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --steps 10 > $O/b0_$TAG.log 2>&1 || fail 1 $? $O/b0_$TAG.log
grep "^{" $O/b0_$TAG.log | tail -1 > $O/${TAG}_bench0.json
INF=$(python -c "import json,sys; print(json.load(open('$O/${TAG}_bench0.json'))['config']['steps_in_flight'])")
# small calls run gathered (include/h2v.h, COALESCING): CO calls per launch.  The one-call-at-a-time passes below then run the
# LAUNCH size, BATCH x CO proofs (their kernels are the timed run's), and the timed passes CO times the steps (as many launches)
CO=$(python -c "
import json
c=json.load(open('$O/${TAG}_bench0.json'))['config'].get('calls_coalesced_per_launch') or 1
print(c[0] if isinstance(c, list) else c)")
ARGS_L="--workload $WL --batch $((BATCH * CO)) --mode $MODE --no-rlc-secondary $*"
# ... and what h2v_workspace_tune chose there (single-plan workloads): the one-step-at-a-time passes below have no lanes to tune in
# and force the same shapes, so that the counters belong to the kernels of the timed run
FORCE=$(python -c "
import json
t=json.load(open('$O/${TAG}_bench0.json'))['config'].get('tuned_launch_shapes') or []
print(('--pairing %d --msm-tpl %d' % (t[0]['pairing_engine'], t[0]['msm_terms_per_lane'])) if len(t) == 1 and (t[0]['pairing_engine'] or t[0]['msm_terms_per_lane']) else '')")
# 2. the timed steps under rocprofv3 (nothing but warm-up + timed steps)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 bench.py $ARGS $FORCE --steps $((60 * CO)) --warmup 5 --timed-only --no-alone > $O/prof_$TAG.log 2>&1 || fail 2 $? $O/prof_$TAG.log
cp $O/prof_$TAG/*/*kernel_stats.csv $O/${TAG}_kernel_stats.csv
# 3. one step at a time, same launch shapes: the kernels' own durations
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profa_$TAG -- python3 bench.py $ARGS_L --steps 20 --warmup 3 --timed-only --no-alone --pipeline streams --inflight 1 --hint $INF $FORCE > $O/profa_$TAG.log 2>&1 || fail 3 $? $O/profa_$TAG.log
cp $O/profa_$TAG/*/*kernel_stats.csv $O/${TAG}_kernel_stats_alone.csv
# 4. counters, separate passes, same one-step-at-a-time command
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${TAG}_$n -- python3 bench.py $ARGS_L --steps 2 --warmup 1 --timed-only --no-alone --pipeline streams --inflight 1 --hint $INF $FORCE > $O/pmc_${TAG}_$n.log 2>&1 || fail "4 ($n)" $? $O/pmc_${TAG}_$n.log
done
{ echo "# workload=$WL batch=$((BATCH * CO)) mode=$MODE pmc_steps=3 tag=$TAG calls_per_launch=$CO (rocprofv3 --pmc, separate passes; bench.py $ARGS_L --steps 2 --warmup 1 --timed-only --no-alone --pipeline streams --inflight 1 --hint $INF $FORCE)"; python tools/pmc_summary.py $O/pmc_${TAG}_*; } > $O/${TAG}_pmc_summary.txt
cp $O/${TAG}_pmc_summary.txt profiles/${TAG}_pmc_summary.txt    # (on the box: the final line below reads it)
# 5. the line
timeout -k 10 400 python bench.py $ARGS $FORCE --steps $((240 * CO)) > $O/b_$TAG.log 2>&1 || fail 5 $? $O/b_$TAG.log
grep "^{" $O/b_$TAG.log | tail -1 > $O/${TAG}_bench.json
grep -E "INSTS_VALU|FETCH|WRITE_SIZE" $O/${TAG}_pmc_summary.txt
head -8 $O/${TAG}_kernel_stats.csv | cut -d, -f1-5
head -8 $O/${TAG}_kernel_stats_alone.csv | cut -d, -f1-5
cut -c1-300 $O/${TAG}_bench.json
