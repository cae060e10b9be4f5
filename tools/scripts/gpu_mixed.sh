#!/bin/bash
# BASELINE configs[2] as one mixed workload: one laned workspace per plan against one for both plans (h2v_workspace_create_multi)
set -o pipefail
O=gpurun_out/mixed; mkdir -p $O
for v in "" "--shared-workspace" "" "--shared-workspace"; do
  n=${v:+shared}; n=${n:-two}
  timeout -k 10 300 python3 bench.py --workload lookup_atms_mixed $v --no-cpu-baseline --no-rlc-secondary --steps 240 --no-alone > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d['config'].get('tuned_launch_shapes'))"
done
timeout -k 10 300 python3 bench.py --workload lookup_atms_mixed --shared-workspace --mode rlc --no-cpu-baseline --steps 120 --no-alone > $O/shared_rlc.json 2> $O/shared_rlc.err || { tail -5 $O/shared_rlc.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload lookup_atms_mixed --mode rlc --no-cpu-baseline --steps 120 --no-alone > $O/two_rlc.json 2> $O/two_rlc.err || { tail -5 $O/two_rlc.err; exit 1; }
python3 -c "import json; [print(n, json.load(open('$O/%s.json' % n))['value'], json.load(open('$O/%s.json' % n))['ms_per_step']) for n in ('shared_rlc', 'two_rlc')]"
