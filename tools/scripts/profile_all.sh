#!/bin/bash
# Every BASELINE configuration through tools/scripts/profile_round.sh under one tag prefix (run through gpurun from the repo
# root), then the single-GPU points the predicted strong-scaling curves of configs[3] / [4] are built from:
#   tools/scripts/profile_all.sh r03_v45
P=$1
S=tools/scripts/profile_round.sh
bash $S ${P} simple_mul 4096 per-proof > gpurun_out/pa_${P}.log 2>&1 || exit 1
bash $S ${P}_lookup_atms_mixed lookup_atms_mixed 2048 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1   # BASELINE configs[2] as named: both plans in flight
bash $S ${P}_lookup_mixed lookup_mixed 2048 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1
bash $S ${P}_atms atms_with_lookups 2048 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1
bash $S ${P}_sha256_1024 sha256 1024 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1
bash $S ${P}_secp256k1_512 secp256k1 512 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1
bash $S ${P}_sha256_128 sha256 128 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1
bash $S ${P}_secp256k1_64 secp256k1 64 per-proof >> gpurun_out/pa_${P}.log 2>&1 || exit 1
for cfg in "sha256 512" "sha256 256" "secp256k1 256" "secp256k1 128"; do set -- $cfg
  timeout -k 10 300 python bench.py --workload $1 --batch $2 --no-cpu-baseline --no-rlc-secondary --no-alone --steps $((240 * 4096 / $2 / 4)) > gpurun_out/b_${P}_$1_$2.log 2>&1 || exit 1   # (60 launches of 4096 gathered proofs)
  grep "^{" gpurun_out/b_${P}_$1_$2.log | tail -1 > gpurun_out/${P}_$1_$2_sweep_bench.json
done
for f in gpurun_out/${P}*_bench.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('steps_in_flight'), d['config'].get('pairing_lanes_per_proof'))
PY
done
