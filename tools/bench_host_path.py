import sys, time
sys.path.insert(0, '.')
from plutus_halo2_verifier_gen_amd import backend, plan as PL, synth, vk as V
vk, td = V.simple_mul_vk()
pl = PL.compile_plan(vk)
b = synth.forge_batch(vk, td, 4096, seed=1000, plan=pl, workers=16)
dp = backend.DevicePlan(pl.to_bytes(), 0)
ws = backend.Workspace(dp, 4096)
for _ in range(3):
    acc = dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, ws=ws)
t0 = time.perf_counter()
K = 20
for _ in range(K):
    acc = dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, ws=ws)
dt = (time.perf_counter() - t0) / K
print("host-buffer path: %.3f ms per 4096 proofs = %.0f proofs/s (all accepted: %s)" % (dt * 1e3, 4096 / dt, sum(acc) == 4096))
