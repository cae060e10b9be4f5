#!/bin/bash
# EXPERIMENT helper: the headline bench under several command-line variants:  tools/scripts/gpu_sweep.sh "--lanes 6" "--lanes 8" ...
# (prints the step, the shapes, what the tuner chose and the kernels' own durations when the alone pass ran)
O=gpurun_out/sweep; mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 120 --no-cpu-baseline --no-rlc-secondary $v > $O/b_$i.log 2>&1; rc=$?
  python - "$v" "$rc" "$O/b_$i.log" <<'PY'
import json,sys
v,rc,f=sys.argv[1:4]
l=[x for x in open(f) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("%-46s rc %s value %9.0f ms/step %.4f in_flight %s pairing %s msm %s/%s ok %s tuned %s own %s" % (v, rc, d['value'], d['ms_per_step'], d['config']['steps_in_flight'], d['pairing_lanes_per_proof'], d['msm_lanes_per_term'], d['msm_ladder_shape_of_a_split'], d['verdicts_as_expected_every_checked_step'], [(t['pairing_engine'], t['msm_terms_per_lane']) for t in (d['config'].get('tuned_launch_shapes') or [])], {k[2:14]: round(x, 2) for k, x in d['kernel_ms'].items()} if d.get('ms_per_step_one_step_at_a_time') else None))
else:
    print(v, "rc", rc, open(f).read()[-800:])
PY
done
