#!/bin/bash
# Where a kernel's wave-cycles go: wider counter sets than profile_round.sh, same one-step-at-a-time command.
#   tools/scripts/pmc_diag.sh TAG WORKLOAD BATCH HINT  -> gpurun_out/TAG_pmc_diag.txt
TAG=$1; WL=$2; BATCH=$3; HINT=$4; shift 4
O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--workload $WL --batch $BATCH --mode per-proof --no-rlc-secondary --steps 2 --warmup 1 --timed-only --no-alone --pipeline streams --inflight 1 --hint $HINT $*"
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
         "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmcd_${TAG}_$i -- python3 bench.py $ARGS > $O/pmcd_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmcd_${TAG}_$i.log; exit 1; }
done
python tools/pmc_summary.py $O/pmcd_${TAG}_* > $O/${TAG}_pmc_diag.txt
grep -v "^k_vk" $O/${TAG}_pmc_diag.txt
