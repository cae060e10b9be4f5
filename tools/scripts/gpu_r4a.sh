#!/bin/bash
# round-4 check A: shutdown + mixed tests, the rocprofv3 lanes form must exit 0, the imad ubench v2
O=gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shutdown or chunking or mixed_batch" > $O/t.log 2>&1; echo "pytest rc=$?" | tee -a $O/t.log; tail -5 $O/t.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload simple_mul --batch 4096 --mode per-proof --no-rlc-secondary --steps 60 --warmup 5 --timed-only --no-alone > $O/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $O/t.log
tail -3 $O/prof.log | cut -c1-300
hipcc --offload-arch=gfx950 -O3 tools/ubench/imad.hip -o /tmp/imad > /dev/null 2>&1 && timeout -k 10 300 /tmp/imad > $O/imad.txt 2>&1; echo "imad rc=$?" | tee -a $O/t.log
head -40 $O/imad.txt
