#!/bin/bash
# EXPERIMENT helper: the headline bench under several command-line variants:  tools/scripts/gpu_sweep.sh "--lanes 6" "--lanes 8" ...
O=gpurun_out/sweep; mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 120 --no-cpu-baseline --no-rlc-secondary --no-alone $v > $O/b_$i.log 2>&1; rc=$?
  python - "$v" "$rc" "$O/b_$i.log" <<'PY'
import json,sys
v,rc,f=sys.argv[1:4]
l=[x for x in open(f) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("%-40s rc %s value %10.0f ms/step %.4f in_flight %s pairing %s msm %s/%s ok %s" % (v, rc, d['value'], d['ms_per_step'], d['config']['steps_in_flight'], d['pairing_lanes_per_proof'], d['msm_lanes_per_term'], d['msm_ladder_shape_of_a_split'], d['verdicts_as_expected_every_checked_step']))
else:
    print(v, "rc", rc, open(f).read()[-800:])
PY
done
