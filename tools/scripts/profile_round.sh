V=$1
timeout -k 10 400 python bench.py > gpurun_out/b_$V.log 2>&1; grep "^{" gpurun_out/b_$V.log | tail -1 > gpurun_out/r01_${V}_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01_$V -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r01_$V.log 2>&1
grep "^{" gpurun_out/prof_r01_$V.log | tail -1 > gpurun_out/r01_${V}_bench_under_rocprof.json
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do n=$(echo $c | cut -d" " -f1); timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${V}_$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${V}_$n.log 2>&1 || exit 1; done
python tools/pmc_summary.py gpurun_out/pmc_${V}_* > gpurun_out/r01_${V}_pmc_summary.txt
grep -E "INSTS_VALU|FETCH|WRITE_SIZE|WAIT_ANY|WAVE_CYCLES" gpurun_out/r01_${V}_pmc_summary.txt
cat gpurun_out/prof_r01_$V/*/*kernel_stats.csv | head -8
cp gpurun_out/prof_r01_$V/*/*kernel_stats.csv gpurun_out/r01_${V}_kernel_stats.csv
cut -c1-200 gpurun_out/r01_${V}_bench.json
