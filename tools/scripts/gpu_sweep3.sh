#!/bin/bash
# bench variants with full flags given by the caller: prints value, ms/step
O=gpurun_out/sweep; mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-rlc-secondary --no-alone $v > $O/d_$i.log 2>&1; rc=$?
  python - "$v" "$rc" "$O/d_$i.log" <<'PY'
import json,sys
v,rc,f=sys.argv[1:4]
l=[x for x in open(f) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("%-60s rc %s value %9.0f ms/step %.4f ok %s" % (v, rc, d['value'], d['ms_per_step'], d['verdicts_as_expected_every_checked_step']))
else:
    print(v, "rc", rc, open(f).read()[-800:])
PY
done
