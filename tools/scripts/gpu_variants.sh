#!/bin/bash
# EXPERIMENT helper: the headline bench under several values of one environment variable
#   tools/scripts/gpu_variants.sh VAR v1 v2 ...
VAR=$1; shift
O=gpurun_out/variants; mkdir -p $O
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 120 --no-cpu-baseline --no-rlc-secondary > $O/b_$v.log 2>&1; rc=$?
  python - "$v" "$rc" <<'PY'
import json,sys
v,rc=sys.argv[1],sys.argv[2]
l=[x for x in open('gpurun_out/variants/b_%s.log'%v) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(v, "rc", rc, "value", d['value'], "ms/step", d['ms_per_step'], "pairing alone", d['kernel_ms'].get('k_pairing_six'), "alone step", d['ms_per_step_one_step_at_a_time'], "ok", d['verdicts_as_expected_every_checked_step'])
else:
    print(v, "rc", rc, open('gpurun_out/variants/b_%s.log'%v).read()[-1500:])
PY
done
