"""Verify exported artefacts on the GPU: the position of the reference's submitter / test harness that reads
`serialized_proof.{hex,json}`, `serialized_public_input.hex` and `serialized_committed_input.hex`
(examples/shared_utils/mod.rs:9-65) and calls the verifier.

  python -m plutus_halo2_verifier_gen_amd.verify_files --vk vk.json [--vk-constants verifier_key.ak] [--kzg-params kzg_params_K]
         --proof serialized_proof.hex [--proof more.hex ...] --public-inputs serialized_public_input.hex
         [--committed serialized_committed_input.hex]

All proofs share the public-input / committed-instance files when only one of each is given; otherwise give one per
proof.  Prints one line per proof (`accept` / `reject`) and exits 0 iff every proof verified.  Needs the HIP library
and a GPU: there is no CPU fallback.
"""
from __future__ import annotations

import argparse
import sys
from typing import List

from . import wire
from .plan import compile_plan
from .vk import BUILDERS, VerifyingKey


def load_vk(spec: str) -> VerifyingKey:
    if spec in BUILDERS:
        return BUILDERS[spec]()[0]
    with open(spec) as f:
        return VerifyingKey.from_json(f.read())


def assemble(vk: VerifyingKey, proofs: List[bytes], pis: List[List[int]], cis: List):
    """-> (proofs, proof_off, instances, committed) in the h2v_batch layout."""
    n = len(proofs)
    if len(pis) == 1:
        pis = pis * n
    if len(cis) <= 1:
        cis = (cis or [None]) * n
    if len(pis) != n or len(cis) != n:
        raise wire.WireError("need one public-input / committed file, or one per proof")
    off = [0]
    for p in proofs:
        off.append(off[-1] + len(p))
    for v in pis:
        if len(v) != vk.n_public_inputs:
            raise wire.WireError("expected %d public inputs, file has %d" % (vk.n_public_inputs, len(v)))
    inst = b"".join(wire.instances_to_abi(v) for v in pis)
    committed = None
    if vk.n_committed_instances:
        if any(c is None for c in cis):
            raise wire.WireError("the circuit has a committed instance: --committed is required")
        committed = b"".join(wire.committed_to_abi(c) for c in cis)
    return b"".join(proofs), off, inst, committed


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--vk", required=True, help="VerifyingKey JSON (vk.py) or a built-in circuit name")
    ap.add_argument("--vk-constants", help="generated verifier_key.ak / VKConstants.hs to take the constants from")
    ap.add_argument("--kzg-params", help="EXPERIMENTAL: kzg_params/kzg_params_{k} (src/kzg_params.rs): s_g2 is taken from its tail (wire.parse_kzg_params: "
                                         "self-validating, layout unpinned - the reference holds no sample) and must agree with the key's s_g2 "
                                         "(--vk-constants, or the vk.json's own)")
    ap.add_argument("--trust-kzg-params", action="store_true",
                    help="let the s_g2 read from --kzg-params REPLACE the key's instead of being checked against it")
    ap.add_argument("--proof", action="append", required=True)
    ap.add_argument("--public-inputs", action="append", required=True)
    ap.add_argument("--committed", action="append", default=[])
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    vk = load_vk(a.vk)
    if a.vk_constants:
        vk = vk.with_constants(wire.load_vk_constants(a.vk_constants))
    if a.kzg_params:
        # the file layout is guessed (six candidate encodings, accepted only when the element before last is the G2
        # generator): by default the value is a CROSS-CHECK of the key's s_g2, never its source
        kp = wire.load_kzg_params(a.kzg_params)
        if kp.s_g2 != vk.s_g2.lower():
            if a.vk_constants or not a.trust_kzg_params:
                raise wire.WireError("s_g2 of %s differs from the key's (%s); --trust-kzg-params takes the file's" % (
                    a.kzg_params, "VK constants" if a.vk_constants else "vk.json"))
            import dataclasses
            vk = dataclasses.replace(vk, s_g2=kp.s_g2)
    proofs = [wire.load_proof(p) for p in a.proof]
    pis = [wire.load_public_inputs(p) for p in a.public_inputs]
    cis = [wire.load_committed_inputs(p) for p in a.committed]
    buf, off, inst, committed = assemble(vk, proofs, pis, cis)
    from . import backend
    dp = backend.DevicePlan(compile_plan(vk).to_bytes(), a.device)
    acc = dp.verify_batch(buf, off, inst, committed)
    for path, ok in zip(a.proof, acc):
        print("%s %s" % ("accept" if ok else "reject", path))
    return 0 if all(acc) else 1


if __name__ == "__main__":
    sys.exit(main())
