run() { out=gpurun_out/sw.log; env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-rlc-secondary --no-alone "${@:2}" > $out 2>&1 || { echo FAIL; tail -2 $out; return; }; grep "^{" $out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('pairing_lanes_per_proof'))"; }
for cfg in "simple_mul 2048" "simple_mul 3072" "lookup_mixed 2048" "lookup_mixed 4096" "atms_with_lookups 4096"; do set -- $cfg
  echo -n "$1 x $2 default: "; run X=1 --workload $1 --batch $2
  echo -n "$1 x $2 six: "; run H2V_PAIRING_SIX=1 --workload $1 --batch $2
  echo -n "$1 x $2 twelve (SIX=0): "; run H2V_PAIRING_SIX=0 --workload $1 --batch $2
done
