#!/bin/bash
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coalesced or mixed_batch" 2>&1 | tail -3
timeout -k 10 400 python3 tests/soak.py --minutes 4 --threads 4 --seed 41 2>&1 | tee $O/soak.log | tail -3
