#!/bin/bash
# two ranks on ONE GPU over gloo (a rehearsal of the N > 1 code path, not a throughput figure: two processes time-slice the card)
set -o pipefail
O=gpurun_out/2rank; mkdir -p $O
run() { n=$1; shift
  H2V_BENCH_DEVICE=0 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --steps 60 --warmup 5 --no-cpu-baseline --no-alone "$@" > $O/$n.log 2> $O/$n.err || { tail -8 $O/$n.err; exit 1; }
  grep "^{" $O/$n.log | tail -1 > $O/$n.json
  python3 -c "import json; d=json.load(open('$O/$n.json')); c=d['config']; print('$n', d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], c.get('ranks_seen'), c.get('dist_backend'), c.get('calls_coalesced_per_launch'), d.get('accept_all_ranks_ok', d.get('all_accept')))"; }
run weak --no-rlc-secondary
run strong_sha256 --workload sha256 --batch 1024 --scaling strong --no-rlc-secondary
run strong_secp_rlc --workload secp256k1 --batch 512 --scaling strong --mode rlc
