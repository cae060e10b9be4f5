"""Host-side mirror of the reference's verify interface for the hot path, over the C-ABI (include/h2v.h).

The reference runs the path as two calls re-exported at /root/reference/src/lib.rs:9-15

    let mut transcript = CircuitTranscript::<CardanoFriendlyBlake2b>::init_from_bytes(&proof);   // simple_mul.rs:97
    let guard = prepare(&vk, &[&[]], &[&[&instance]], &mut transcript)?;                          // simple_mul.rs:98
    guard.verify(&kzg_params.verifier_params())?;                                                 // simple_mul.rs:101

(`DualMSM::check(&params) -> bool` in examples/ivc.rs:195-198, `transcript.assert_empty()` in ivc.rs:92-94).
Same names, argument meaning and error behaviour here; the work itself happens on the GPU when the guard is
consumed, and `verify_batch` is the batched form the hardware wants (thousands of independent proofs per call).

Error behaviour mirrors the reference: malformed encodings surface from `prepare`/`verify` as `VerifyError`
(Rust: `Err(midnight_proofs::plonk::Error)`), a failed pairing as `VerifyError` from `verify`.  API misuse and device
problems raise `backend.H2VError`.  There is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

from . import backend
from . import bls12_381 as bls
from .plan import Plan, compile_plan
from .vk import VerifyingKey

STATUS_TEXT = {
    backend.ST_BAD_SCALAR: "non-canonical scalar encoding in the proof",
    backend.ST_INVERSE_OF_ZERO: "division by zero in the verifier (x^n = 1 or x = omega^i)",
    backend.ST_SHORT_PROOF: "proof shorter than the verifier's read program",
    backend.ST_BAD_POINT: "invalid G1 encoding (malformed / off-curve / not in the subgroup)",
    backend.ST_PAIRING: "pairing check failed",
    backend.ST_RECURSION: "recursion: the verifying-key hash in the public inputs is not this key's",
}


class VerifyError(Exception):
    """The proof does not verify (Rust: Err(plonk::Error))."""

    def __init__(self, status: int):
        self.status = status
        reasons = [t for bit, t in STATUS_TEXT.items() if status & bit]
        super().__init__("; ".join(reasons) or "rejected")


@dataclass
class ParamsVerifierKZG:
    """`ParamsKZG::verifier_params()`: the verifier needs only s_g2 (and the G2 generator)."""
    s_g2: bytes  # 96-byte zcash-compressed G2


class CircuitTranscript:
    """CircuitTranscript<CardanoFriendlyBlake2b> on the verifier side: a cursor over the proof bytes.
    (/root/reference/src/plutus_gen/adjusted_types/mod.rs:30-72 is replayed on the GPU, not here.)"""

    def __init__(self, proof: bytes):
        self._proof = bytes(proof)
        self._consumed = 0

    @classmethod
    def init_from_bytes(cls, proof: bytes) -> "CircuitTranscript":
        return cls(proof)

    @property
    def proof(self) -> bytes:
        return self._proof

    def assert_empty(self) -> None:
        """examples/ivc.rs:92-94: every byte of the proof must have been read."""
        if self._consumed != len(self._proof):
            raise VerifyError(backend.ST_SHORT_PROOF if self._consumed > len(self._proof) else 0)


class Verifier:
    """A VerifyingKey compiled to a plan and loaded on one GPU (h2v_plan*): the object behind `prepare`."""

    def __init__(self, vk: VerifyingKey, device: int = 0, plan: Optional[Plan] = None):
        self.vk = vk
        self.plan = plan or compile_plan(vk)
        self.device_plan = backend.DevicePlan(self.plan.to_bytes(), device)
        self._ws: Optional[backend.Workspace] = None

    def _workspace(self, n: int) -> backend.Workspace:
        if self._ws is None or self._ws.max_batch < n:
            self._ws = backend.Workspace(self.device_plan, max(n, 64))
        return self._ws

    def verify_batch(self, proofs: Sequence[bytes], instances: Sequence[Sequence[int]],
                     committed: Optional[Sequence[Optional[bytes]]] = None, mode: str = "per-proof",
                     seed: Optional[bytes] = None) -> List[bool]:
        """accept[i] for n independent proofs of this circuit (instances[i]: the public-input scalars of proof i;
        committed[i]: its committed instance as 48 compressed bytes, when the circuit has one).
        mode="rlc": the batch-accept fast path (one bucket MSM + one pairing for the batch, per-proof kernels only if
        the batch check fails; same accept vector up to a 2^-128 soundness error over `seed`, drawn from the OS when None)."""
        n = len(proofs)
        if n == 0:
            return []
        if len(instances) != n:
            raise ValueError("one instance list per proof")
        n_pi = self.plan.n_pi
        for ins in instances:
            if len(ins) != n_pi:
                raise ValueError("expected %d public inputs per proof" % n_pi)
        off = [0]
        for p in proofs:
            off.append(off[-1] + len(p))
        inst = b"".join((int(v) % bls.R).to_bytes(32, "little") for ins in instances for v in ins)
        ci = None
        if self.plan.n_ci:
            if committed is None or len(committed) != n:
                raise ValueError("this circuit takes one committed instance per proof")
            ci = b"".join(bls.g1_compress(None) if c is None else bytes(c) for c in committed)
        if mode == "rlc":
            acc, _fell_back = self.device_plan.verify_batch_rlc(b"".join(proofs), off, inst, ci, ws=self._workspace(n), seed=seed)
        elif mode == "per-proof":
            acc = self.device_plan.verify_batch(b"".join(proofs), off, inst, ci, ws=self._workspace(n))
        else:
            raise ValueError("mode is 'per-proof' or 'rlc'")
        return [bool(a) for a in acc]


class Guard:
    """The VerificationGuard returned by `prepare`: consumed by `verify` (Guard::verify) or `check` (DualMSM::check)."""

    def __init__(self, verifier: Verifier, proof: bytes, instances: bytes, committed: Optional[bytes]):
        self._v = verifier
        self._proof, self._instances, self._committed = proof, instances, committed
        self._used = False

    def _run(self, params: Optional[ParamsVerifierKZG]) -> int:
        if self._used:
            raise RuntimeError("guard already consumed")  # Rust: moved value
        self._used = True
        if params is not None and bytes(params.s_g2) != bytes.fromhex(self._v.vk.s_g2):
            raise ValueError("verifier params do not match the verifying key's SRS (s_g2 differs)")
        tr = self._v.device_plan.trace(self._proof, self._instances, self._committed)
        return tr["status"] if not tr["accept"] else 0

    def verify(self, params: Optional[ParamsVerifierKZG] = None) -> None:
        st = self._run(params)
        if st:
            raise VerifyError(st)

    def check(self, params: Optional[ParamsVerifierKZG] = None) -> bool:
        return self._run(params) == 0


_VERIFIERS = {}


def verifier_for(vk: VerifyingKey, device: int = 0) -> Verifier:
    """One compiled plan per (verifying key CONTENT, device).  Keyed by the key's canonical JSON, not by id(vk): an
    object id is reused after garbage collection and would hand a new key a stale plan."""
    key = (vk.to_json(), device)
    if key not in _VERIFIERS:
        _VERIFIERS[key] = Verifier(vk, device)
    return _VERIFIERS[key]


def prepare(vk: VerifyingKey, committed_instances: Sequence[Sequence[Optional[bytes]]],
            instances: Sequence[Sequence[Sequence[int]]], transcript: CircuitTranscript, device: int = 0) -> Guard:
    """prepare(&vk, committed_instances: &[&[C]], instances: &[&[&[F]]], &mut transcript) for ONE proof
    (the outer slices have length 1, as at every call site of the reference)."""
    if len(instances) != 1 or len(committed_instances) != 1:
        raise ValueError("one proof per prepare() call; use Verifier.verify_batch for batches")
    cols = instances[0]
    pub = list(cols[0]) if len(cols) else []
    v = verifier_for(vk, device)
    if len(pub) != v.plan.n_pi:
        raise ValueError("expected %d public inputs" % v.plan.n_pi)
    cis = list(committed_instances[0])
    if len(cis) != v.plan.n_ci:
        raise ValueError("expected %d committed instances" % v.plan.n_ci)
    ci = None
    if cis:
        ci = bls.g1_compress(None) if cis[0] is None else bytes(cis[0])
    inst = b"".join((int(x) % bls.R).to_bytes(32, "little") for x in pub)
    transcript._consumed = v.plan.proof_len
    return Guard(v, transcript.proof, inst, ci)
