run() { out=gpurun_out/sw.log; env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-rlc-secondary --no-alone "${@:2}" > $out 2>&1 || { echo FAIL; tail -2 $out; return; }; grep "^{" $out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for cfg in "sha256 1024" "secp256k1 512" "sha256 512" "sha256 256" "secp256k1 256" "sha256 128" "secp256k1 128" "secp256k1 64" "lookup_mixed 2048" "atms_with_lookups 2048" "simple_mul 4096" "simple_mul 512" "simple_mul 1024"; do set -- $cfg
  echo -n "$1 x $2: "; run X=1 --workload $1 --batch $2
done
echo -n "rlc: "; run X=1 --mode rlc
