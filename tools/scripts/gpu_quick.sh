#!/bin/bash
# quick check of a kernel change on the GPU box: the pairing / end-to-end parity tests, then the headline bench (120 steps)
O=gpurun_out/quick; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${1:-pairing or end_to_end or ivc_fold}" > $O/t.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t.log
timeout -k 10 300 python bench.py --steps 120 --no-cpu-baseline --no-rlc-secondary > $O/b.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/quick/b.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("value", d['value'], "ms/step", d['ms_per_step'], "kernel_ms", d['kernel_ms'], "alone step", d['ms_per_step_one_step_at_a_time'])
else:
    print(open('gpurun_out/quick/b.log').read()[-2000:])
PY
