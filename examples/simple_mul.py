#!/usr/bin/env python3
"""The verify half of the reference's examples/simple_mul.rs on the MI355X backend.

reference (examples/simple_mul.rs)                         here
  create_proof(.., &[&[&instance]], ..)   :70-83           synth.forge_batch (no prover in scope: a proof that satisfies
                                                           the verifier equation for the test SRS is forged from its
                                                           trapdoor - SURVEY.md section 7)
  invalid_proof[48*8 + 2] = !byte         :87-95           the same byte flip
  CTranscript::init_from_bytes(&proof)    :97              api.CircuitTranscript.init_from_bytes
  prepare(&vk, &[&[]], &[&[&instance]])   :98              api.prepare
  verifier.verify(&params)                :101-104         Guard.verify (raises VerifyError) / Guard.check
  export_*                                :128-140         wire.export_proof / serialize_proof / export_public_inputs

Needs the HIP library and a GPU (there is no CPU fallback):  python examples/simple_mul.py [out_dir]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from plutus_halo2_verifier_gen_amd import api, plan as PL, synth, vk as V, wire  # noqa: E402


def main() -> int:
    vk, trapdoor = V.simple_mul_vk()
    plan = PL.compile_plan(vk)
    batch = synth.forge_batch(vk, trapdoor, 1, seed=1, plan=plan, workers=1)
    proof = batch.proof(0)
    instance = batch.instance_ints(0, vk.n_public_inputs)      # [42, 42, 42], as in simple_mul.rs:68
    print("Public inputs:", instance)
    print("proof size", len(proof))

    invalid_proof = bytearray(proof)
    index = 48 * 8 + 2                                         # first scalar of the proof (simple_mul.rs:91)
    invalid_proof[index] ^= 0xFF

    params = api.ParamsVerifierKZG(bytes.fromhex(vk.s_g2))       # kzg_params.verifier_params(): s_g2 is all the verifier needs

    transcript = api.CircuitTranscript.init_from_bytes(proof)
    guard = api.prepare(vk, [[]], [[instance]], transcript)     # prepare(&vk, &[&[]], &[&[&instance]], &mut t)
    guard.verify(params)
    transcript.assert_empty()
    print("proof verified")

    transcript = api.CircuitTranscript.init_from_bytes(bytes(invalid_proof))
    try:
        api.prepare(vk, [[]], [[instance]], transcript).verify(params)
    except api.VerifyError as e:
        print("invalid proof rejected:", e)
    else:
        print("ERROR: the invalid proof verified")
        return 1

    # many proofs at once (the form the hardware wants), per proof and through the batch-accept fast path
    v = api.verifier_for(vk)
    many = [proof, bytes(invalid_proof)] * 8
    per_proof = v.verify_batch(many, [instance] * len(many))
    rlc = v.verify_batch(many, [instance] * len(many), mode="rlc")
    assert per_proof == rlc == [True, False] * 8
    print("batch of %d verified in both modes: %d accepted" % (len(many), sum(per_proof)))

    out = sys.argv[1] if len(sys.argv) > 1 else None
    if out:
        os.makedirs(out, exist_ok=True)
        wire.export_proof(os.path.join(out, "serialized_proof.hex"), proof)
        wire.serialize_proof(os.path.join(out, "serialized_proof.json"), proof)
        with open(os.path.join(out, "serialized_public_input.hex"), "w") as f:
            wire.export_public_inputs(instance, f)
        with open(os.path.join(out, "vk.json"), "w") as f:
            f.write(vk.to_json())
        print("exported to", out, "- verify again with:")
        print("  python -m plutus_halo2_verifier_gen_amd.verify_files --vk %s/vk.json --proof %s/serialized_proof.hex "
              "--public-inputs %s/serialized_public_input.hex" % (out, out, out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
