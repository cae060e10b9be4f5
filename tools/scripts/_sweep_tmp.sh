run() { # name, env, args
  out=gpurun_out/sw_$1.log; shift
  env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-rlc-secondary --no-alone "${@:2}" > $out 2>&1 || { echo FAIL $out; tail -3 $out; exit 1; }
  grep "^{" $out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('msm_lanes_per_term'), d.get('msm_ladder_shape_of_a_split'))"
}
for t in 2 3 4; do echo -n "simple_mul TPL=$t: "; run t$t H2V_MSM_TPL=$t --steps 60; done
for t in 2 3 4; do echo -n "sha256 1024 TPL=$t: "; run s$t H2V_MSM_TPL=$t --steps 60 --workload sha256; done
for t in 2 4; do echo -n "atms 2048 TPL=$t: "; run a$t H2V_MSM_TPL=$t --steps 60 --workload atms_with_lookups; done
