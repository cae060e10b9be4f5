#!/usr/bin/env python3
"""Development aid: runs the bucket MSM probe on n random terms (a few hundred distinct bases) so that a
`rocprofv3 --kernel-trace --stats` of this script shows the kernels of ONE bucket MSM in isolation."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from plutus_halo2_verifier_gen_amd import backend, bls12_381 as bls, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40966
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = random.Random(1)
fb = synth.fixed_base()
comp = [bls.g1_compress(fb.mul(rng.randrange(1, bls.R))) for _ in range(256)]
scal = [rng.randrange(bls.R) for _ in range(n)]
bases = [comp[rng.randrange(256)] for _ in range(n)]
for _ in range(reps):
    r = backend.probe_g1_msm_pippenger(scal, bases)
print("ok", r is not None)
