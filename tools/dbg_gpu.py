import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plutus_halo2_verifier_gen_amd import backend, plan as PL, vk as V, synth
from oracle import binding as orc
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
vk, td = V.BUILDERS[name]()
pl = PL.compile_plan(vk)
print("plan", PL.plan_stats(pl), flush=True)
dp = backend.DevicePlan(pl.to_bytes(), 0)
print("loaded", flush=True)
b = synth.forge_batch(vk, td, n, seed=21, plan=pl, workers=1)
print("forged", flush=True)
got = dp.verify_batch(b.proofs, b.proof_off, b.instances, b.committed)
print("gpu", list(got), flush=True)
ov = orc.OracleVK(orc.vk_desc(json.loads(vk.to_json()), vk.omega, vk.omega_inv, vk.barycentric_weight))
print("cpu", list(ov.verify_batch(b.proofs, b.proof_off, b.instances, b.committed, threads=4)))
