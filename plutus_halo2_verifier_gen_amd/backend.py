"""ctypes binding of the C-ABI library (include/h2v.h -> libh2v_hip.so, built in-tree by __graft_entry__.build()).

There is deliberately NO fallback: if the HIP library is missing, or no GPU is present, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libh2v_hip.so")

H2V_OK = 0
ST_BAD_SCALAR, ST_INVERSE_OF_ZERO, ST_SHORT_PROOF, ST_BAD_POINT, ST_PAIRING, ST_RECURSION = 1, 2, 4, 8, 16, 32


class H2VError(RuntimeError):
    pass


class Batch(C.Structure):
    _fields_ = [("n", C.c_uint64), ("proofs", C.c_void_p), ("proof_off", C.c_void_p), ("instances", C.c_void_p),
                ("committed", C.c_void_p)]


class Timings(C.Structure):
    _fields_ = [("transcript_combiner_ms", C.c_float), ("g1_decompress_ms", C.c_float), ("g1_msm_ms", C.c_float),
                ("pairing_ms", C.c_float), ("total_ms", C.c_float), ("launches", C.c_uint32),
                ("msm_lanes_per_term", C.c_uint32), ("pairing_lanes_per_proof", C.c_uint32),
                ("g1_msm_fixed_ms", C.c_float), ("msm_var_lanes_per_term", C.c_uint32)]


class RlcOpts(C.Structure):
    _fields_ = [("seed", C.c_uint8 * 32), ("flags", C.c_uint32)]


class RlcTimings(C.Structure):
    _fields_ = [("transcript_combiner_ms", C.c_float), ("g1_decompress_ms", C.c_float), ("prepare_ms", C.c_float),
                ("bucket_sort_ms", C.c_float), ("bucket_accumulate_ms", C.c_float), ("bucket_reduce_ms", C.c_float),
                ("pairing_ms", C.c_float), ("total_ms", C.c_float), ("msm_terms", C.c_uint32),
                ("window_bits", C.c_uint32), ("windows", C.c_uint32), ("max_chain", C.c_uint32)]


class PlanOpts(C.Structure):
    _fields_ = [("fixed_base_window_bits", C.c_uint32), ("reserved", C.c_uint32 * 7)]


class TuneReport(C.Structure):
    _fields_ = [("n_measured", C.c_uint32), ("calls_in_flight", C.c_uint32), ("default_ms", C.c_float), ("best_ms", C.c_float),
                ("pairing_engine", C.c_int32), ("msm_terms_per_lane", C.c_int32), ("reserved", C.c_uint32 * 4)]


# h2v_workspace_set_option / h2v_probe_set_option ids (include/h2v.h)
OPT_MSM_TERMS_PER_LANE, OPT_PAIRING_ENGINE, OPT_STREAMS, OPT_MSM_LANES_PER_TERM, OPT_MSM_BLOCK_SIZE, OPT_MSM_FIXED_SPLIT = 1, 2, 3, 4, 5, 6
OPT_COMBINER_SCHEDULE, OPT_COMBINER_PROOFS_PER_BLOCK, OPT_DECOMPRESS_FORM, OPT_PIPES, OPT_RLC_GROUP_STAGE, OPT_RLC_WINDOW_BITS, OPT_RLC_CHAIN, OPT_RLC_ROUTE, OPT_COALESCE = 7, 8, 9, 10, 11, 12, 13, 14, 15
OPT_COUNT = 16

RLC_SEED_GIVEN = 1
RLC_ONE_STREAM = 2
SUBMIT_RLC = 1


def _rlc_opts(seed, one_stream: bool = False):
    if seed is None and not one_stream:
        return None
    flags = RLC_ONE_STREAM if one_stream else 0
    if seed is None:
        return RlcOpts((C.c_uint8 * 32)(), flags)
    seed = bytes(seed)
    if len(seed) != 32:
        raise ValueError("the RLC seed is 32 bytes")
    return RlcOpts((C.c_uint8 * 32)(*seed), flags | RLC_SEED_GIVEN)


EXPORTS = [
    "h2v_plan_load", "h2v_plan_load_ex", "h2v_plan_free", "h2v_plan_info", "h2v_plan_compile", "h2v_blob_free", "h2v_workspace_create", "h2v_workspace_free",
    "h2v_workspace_timings", "h2v_workspace_hint_in_flight", "h2v_workspace_create_lanes", "h2v_workspace_create_multi", "h2v_workspace_defer_joins",
    "h2v_workspace_join", "h2v_workspace_lanes", "h2v_workspace_depth", "h2v_workspace_set_option", "h2v_workspace_get_option", "h2v_workspace_tune", "h2v_probe_set_option",
    "h2v_verify_batch", "h2v_verify_batch_submit", "h2v_verify_batch_wait", "h2v_verify_batch_device", "h2v_verify_batch_rlc", "h2v_verify_batch_rlc_device",
    "h2v_workspace_rlc_result", "h2v_probe_g1_msm_pippenger", "h2v_plan_trace_slots", "h2v_trace", "h2v_probe_field",
    "h2v_probe_blake2b", "h2v_probe_g1_decompress", "h2v_probe_g1_msm", "h2v_probe_g1_msm_fixed", "h2v_probe_quad_madd", "h2v_probe_pairing", "h2v_probe_pairing_ex",
    "h2v_last_error", "h2v_build_id",
    "h2v_device_count", "h2v_shutdown",
]

_lib = None


def lib():
    """Loads libh2v_hip.so.  Raises if it has not been built: the HIP extension is mandatory."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise H2VError("HIP backend library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm bundles its own copy of the HIP runtime under the system's soname.  If this library came first, the
        # process would hold /opt/rocm's runtime and a later `import torch` could no longer see the GPU ("No HIP GPUs are
        # available"): a Python host that may hand over torch tensors lets torch load its runtime first.
        if "torch" not in sys.modules and os.environ.get("H2V_NO_TORCH_PRELOAD") is None:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        L.h2v_last_error.restype = C.c_char_p
        L.h2v_build_id.restype = C.c_char_p
        L.h2v_plan_load.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.h2v_plan_load_ex.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(PlanOpts), C.POINTER(C.c_void_p)]
        L.h2v_plan_free.argtypes = [C.c_void_p]
        L.h2v_plan_compile.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.h2v_blob_free.argtypes = [C.c_void_p]
        L.h2v_plan_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
        L.h2v_workspace_create.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.h2v_workspace_free.argtypes = [C.c_void_p]
        L.h2v_workspace_create_lanes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.h2v_workspace_create_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.h2v_workspace_defer_joins.argtypes = [C.c_void_p, C.c_int]
        L.h2v_workspace_join.argtypes = [C.c_void_p, C.c_void_p]
        L.h2v_workspace_lanes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.h2v_workspace_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
        L.h2v_workspace_set_option.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
        L.h2v_workspace_get_option.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
        L.h2v_probe_set_option.argtypes = [C.c_uint32, C.c_int32]
        L.h2v_workspace_tune.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(TuneReport)]
        L.h2v_workspace_timings.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Timings)]
        L.h2v_verify_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p]
        L.h2v_verify_batch_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.POINTER(Timings)]
        L.h2v_verify_batch_submit.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_uint32, C.POINTER(RlcOpts)]
        L.h2v_verify_batch_wait.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.h2v_verify_batch_rlc.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.POINTER(RlcOpts),
                                           C.POINTER(C.c_int)]
        L.h2v_verify_batch_rlc_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.POINTER(RlcOpts)]
        L.h2v_workspace_rlc_result.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(RlcTimings)]
        L.h2v_probe_g1_msm_pippenger.argtypes = [C.c_int, C.c_uint32, C.c_char_p, C.c_char_p, C.c_void_p]
        L.h2v_plan_trace_slots.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
        L.h2v_trace.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.h2v_probe_field.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.h2v_probe_blake2b.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p]
        L.h2v_probe_g1_decompress.argtypes = [C.c_int, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p]
        L.h2v_probe_g1_msm.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_void_p]
        L.h2v_probe_g1_msm_fixed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.h2v_probe_pairing.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_void_p]
        L.h2v_probe_pairing_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
        L.h2v_shutdown.argtypes = [C.c_int]
        # h2v_shutdown before the interpreter goes down: the library's pool streams (hardware queues of their own) must not
        # outlive the HIP runtime / a profiler's tool library (include/h2v.h: library lifecycle).  Handles that Python still
        # holds afterwards are empty shells; their __del__ frees the host structs only.
        import atexit
        atexit.register(L.h2v_shutdown, -1)
        _lib = L
    return _lib


def check(rc: int):
    if rc != H2V_OK:
        raise H2VError("h2v error %d: %s" % (rc, (lib().h2v_last_error() or b"").decode()))


def probe_set_option(option: int, value: int) -> None:
    """h2v_probe_set_option: the launch-shape option the h2v_probe_* calls of this thread run with (0 = the launcher's choice)"""
    check(lib().h2v_probe_set_option(option, value))


def plan_compile(vk_json: str) -> bytes:
    """h2v_plan_compile: the C++ plan compiler behind the C-ABI (host-only; byte-identical with plan.compile_plan)."""
    raw = vk_json.encode()
    out, n = C.c_void_p(), C.c_size_t()
    check(lib().h2v_plan_compile(raw, len(raw), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        lib().h2v_blob_free(out)


def device_count() -> int:
    return lib().h2v_device_count()


def shutdown(device: int = -1) -> None:
    """h2v_shutdown: release everything the library owns on `device` (-1: every device); later calls raise H2VError."""
    check(lib().h2v_shutdown(device))


class DevicePlan:
    """A plan uploaded to one GPU (h2v_plan*)."""

    def __init__(self, plan_bytes: bytes, device: int = 0, fixed_base_window_bits: int = 0):
        """fixed_base_window_bits (h2v_plan_load_ex): 0 = the library's choice, or 4 / 8 / 12"""
        self._h = C.c_void_p()
        self.device = device
        if fixed_base_window_bits:
            opts = PlanOpts(fixed_base_window_bits, (C.c_uint32 * 7)())
            check(lib().h2v_plan_load_ex(plan_bytes, len(plan_bytes), device, C.byref(opts), C.byref(self._h)))
        else:
            check(lib().h2v_plan_load(plan_bytes, len(plan_bytes), device, C.byref(self._h)))
        v = [C.c_uint32() for _ in range(4)]
        check(lib().h2v_plan_info(self._h, *[C.byref(x) for x in v]))
        self.proof_len, self.n_pi, self.n_ci, self.n_terms = [x.value for x in v]
        n = C.c_uint32()
        check(lib().h2v_plan_trace_slots(self._h, None, 0, C.byref(n)))
        slots = (C.c_uint32 * max(1, n.value))()
        check(lib().h2v_plan_trace_slots(self._h, slots, n.value, C.byref(n)))
        self.trace_slots = list(slots[:n.value])

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            lib().h2v_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer batch verify
    def _host_batch(self, proofs: bytes, proof_off, instances: bytes, committed: Optional[bytes]):
        n = len(proof_off) - 1
        if n < 0 or any(proof_off[i + 1] < proof_off[i] for i in range(n)) or (n >= 0 and proof_off[0] < 0):
            raise H2VError("proof offsets must be non-decreasing")
        if n > 0 and proof_off[-1] > len(proofs):
            raise H2VError("proof_off[n] = %d exceeds the %d proof bytes handed over" % (proof_off[-1], len(proofs)))
        if self.n_pi and len(instances or b"") < 32 * self.n_pi * n:
            raise H2VError("instances: expected %d bytes" % (32 * self.n_pi * n))
        if self.n_ci and len(committed or b"") < 48 * n:
            raise H2VError("committed: expected %d bytes" % (48 * n))
        off = (C.c_uint64 * (n + 1))(*proof_off)
        pbuf = C.create_string_buffer(proofs, len(proofs)) if proofs else C.create_string_buffer(1)
        ibuf = C.create_string_buffer(instances, len(instances)) if instances else None
        cbuf = C.create_string_buffer(committed, len(committed)) if committed else None
        b = Batch(n, C.cast(pbuf, C.c_void_p), C.cast(off, C.c_void_p),
                  C.cast(ibuf, C.c_void_p) if ibuf is not None else None,
                  C.cast(cbuf, C.c_void_p) if cbuf is not None else None)
        return n, b, (off, pbuf, ibuf, cbuf)   # (the buffers must outlive the call)

    def verify_batch(self, proofs: bytes, proof_off, instances: bytes, committed: Optional[bytes], ws=None) -> bytes:
        n, b, _keep = self._host_batch(proofs, proof_off, instances, committed)
        acc = (C.c_uint8 * max(1, n))()
        check(lib().h2v_verify_batch(self._h, C.byref(b), acc, ws.handle if ws else None))
        return bytes(acc[:n])

    def host_batch(self, proofs: bytes, proof_off, instances: bytes, committed: Optional[bytes]):
        """The batch as ctypes buffers a caller owns (what a Rust / C caller hands over): (h2v_batch, keep-alive)."""
        n, b, keep = self._host_batch(proofs, proof_off, instances, committed)
        return b, keep

    def submit(self, batch: "Batch", ws, rlc: bool = False, seed: Optional[bytes] = None, one_stream: bool = False):
        """h2v_verify_batch_submit: copy + upload + verification + download enqueued on the workspace's stream."""
        opts = _rlc_opts(seed, one_stream)
        check(lib().h2v_verify_batch_submit(self._h, C.byref(batch), ws.handle, SUBMIT_RLC if rlc else 0,
                                            C.byref(opts) if opts is not None else None))

    def verify_batch_rlc(self, proofs: bytes, proof_off, instances: bytes, committed: Optional[bytes], ws=None,
                         seed: Optional[bytes] = None):
        """Batch-accept fast path (h2v_verify_batch_rlc): returns (accept bytes, fell_back)."""
        n, b, _keep = self._host_batch(proofs, proof_off, instances, committed)
        acc = (C.c_uint8 * max(1, n))()
        fb = C.c_int(0)
        opts = _rlc_opts(seed)
        check(lib().h2v_verify_batch_rlc(self._h, C.byref(b), acc, ws.handle if ws else None,
                                         C.byref(opts) if opts is not None else None, C.byref(fb)))
        return bytes(acc[:n]), bool(fb.value)

    def verify_batch_rlc_device(self, n, d_proofs, d_off, d_inst, d_ci, d_accept, d_status=None, ws=None, stream=None,
                                seed: Optional[bytes] = None, one_stream: bool = False):
        b = Batch(n, d_proofs, d_off, d_inst, d_ci)
        opts = _rlc_opts(seed, one_stream)
        check(lib().h2v_verify_batch_rlc_device(self._h, C.byref(b), d_accept, d_status, ws.handle if ws else None, stream,
                                                C.byref(opts) if opts is not None else None))

    # ---- device-resident batch verify (pointers = torch tensor data_ptr())
    def verify_batch_device(self, n, d_proofs, d_off, d_inst, d_ci, d_accept, d_status=None, ws=None, stream=None,
                            timings: bool = False):
        b = Batch(n, d_proofs, d_off, d_inst, d_ci)
        tm = Timings() if timings else None
        check(lib().h2v_verify_batch_device(self._h, C.byref(b), d_accept, d_status, ws.handle if ws else None, stream,
                                            C.byref(tm) if timings else None))
        return tm

    def trace(self, proof: bytes, instances: bytes, committed: Optional[bytes]):
        nt = len(self.trace_slots)
        sc = C.create_string_buffer(32 * max(1, nt))
        ms = C.create_string_buffer(32 * self.n_terms)
        el = C.create_string_buffer(96)
        er = C.create_string_buffer(96)
        st = C.c_uint32()
        acc = C.c_uint8()
        check(lib().h2v_trace(self._h, proof, len(proof), instances, committed, sc, ms, el, er, C.byref(st), C.byref(acc)))
        scal = {self.trace_slots[k]: int.from_bytes(sc.raw[32 * k:32 * k + 32], "little") for k in range(nt)}
        msm = [int.from_bytes(ms.raw[32 * k:32 * k + 32], "little") for k in range(self.n_terms)]

        def pt(b):
            return None if b == bytes(96) else (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))

        return {"scalars": scal, "msm_scalars": msm, "el": pt(el.raw), "er": pt(er.raw), "status": st.value,
                "accept": acc.value}


class Workspace:
    """h2v_workspace.  lanes / chunk given (0 = the library's choice): a LANED workspace (h2v_workspace_create_lanes) -
    calls on it are cut into chunks that are pipelined through library-owned lanes; plain Workspace(plan, n) is laned by
    itself from FOUR TIMES the plan's chunk size up (h2v_workspace_create), an ordinary workspace below that.  A laned
    workspace has at most 16 lanes (= host batches in flight through submit / wait), no trace buffer, and - with deferred
    joins - refuses the legacy NULL stream: its lanes run on blocking streams, which any NULL-stream work (PyTorch's
    default stream, a synchronous hipMemcpy) would drain (include/h2v.h: h2v_workspace_defer_joins)."""

    def __init__(self, plan: DevicePlan, max_batch: int, lanes: Optional[int] = None, chunk: Optional[int] = None):
        self._h = C.c_void_p()
        if lanes is None and chunk is None:
            check(lib().h2v_workspace_create(plan.handle, max_batch, C.byref(self._h)))
        else:
            check(lib().h2v_workspace_create_lanes(plan.handle, max_batch, lanes or 0, chunk or 0, C.byref(self._h)))
        self.max_batch = max_batch

    @classmethod
    def multi(cls, plans, max_batch: int, lanes: int = 0, chunk: int = 0) -> "Workspace":
        """h2v_workspace_create_multi: ONE laned workspace for several plans of one device (calls name their plan as usual)"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        arr = (C.c_void_p * len(plans))(*[p.handle for p in plans])
        check(lib().h2v_workspace_create_multi(arr, len(plans), max_batch, lanes, chunk, C.byref(self._h)))
        self.max_batch = max_batch
        return self

    def lanes(self):
        """(number of lanes, chunk size); (1, max_batch) for a workspace that is not laned"""
        a, b = C.c_uint32(), C.c_uint32()
        check(lib().h2v_workspace_lanes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def depth(self, n: int, rlc: bool = False) -> int:
        """h2v_workspace_depth: calls of n proofs the workspace keeps in flight before a call waits for a lane"""
        a = C.c_uint32()
        check(lib().h2v_workspace_depth(self._h, n, 1 if rlc else 0, C.byref(a)))
        return a.value

    OPT_MSM_TERMS_PER_LANE, OPT_PAIRING_ENGINE, OPT_STREAMS = 1, 2, 3

    def set_option(self, option: int, value: int) -> None:
        """h2v_workspace_set_option: a launch shape instead of the launcher's choice (0 = back to its choice)"""
        check(lib().h2v_workspace_set_option(self._h, option, value))

    def get_option(self, option: int) -> int:
        v = C.c_int32()
        check(lib().h2v_workspace_get_option(self._h, option, C.byref(v)))
        return v.value

    def tune(self, plan: "DevicePlan", n, d_proofs, d_off, d_inst, d_ci, stream) -> "TuneReport":
        """h2v_workspace_tune: measure the candidate launch shapes on this device-resident batch, keep the fastest"""
        b = Batch(n, d_proofs, d_off, d_inst, d_ci)
        rep = TuneReport()
        check(lib().h2v_workspace_tune(plan.handle, C.byref(b), self._h, stream, 0, C.byref(rep)))
        return rep

    def defer_joins(self, on: bool = True) -> None:
        """h2v_workspace_defer_joins: device-resident calls return without making the caller's stream wait, so that
        consecutive calls overlap in the lanes; join() orders a stream behind everything submitted so far."""
        check(lib().h2v_workspace_defer_joins(self._h, 1 if on else 0))

    def join(self, stream=None) -> None:
        check(lib().h2v_workspace_join(self._h, stream))

    @property
    def handle(self):
        return self._h

    def timings(self, calls_back: int = 0) -> Timings:
        tm = Timings()
        check(lib().h2v_workspace_timings(self._h, calls_back, C.byref(tm)))
        return tm

    def hint_in_flight(self, n: int) -> None:
        """h2v_workspace_hint_in_flight: the caller keeps n batches in flight (a tuning hint; results do not depend on it)."""
        check(lib().h2v_workspace_hint_in_flight(self._h, C.c_uint32(n)))

    def wait(self, n: int):
        """h2v_verify_batch_wait for the batch submitted on this workspace: (accept bytes, fell_back)."""
        acc = (C.c_uint8 * max(1, n))()
        fb = C.c_int(0)
        check(lib().h2v_verify_batch_wait(self._h, acc, C.byref(fb)))
        return bytes(acc[:n]), bool(fb.value)

    def rlc_result(self, calls_back: int = 0, timings: bool = True):
        """(batch check passed?, RlcTimings) of a past RLC call on this workspace; synchronise its stream first."""
        ok = C.c_uint32(0)
        tm = RlcTimings() if timings else None
        check(lib().h2v_workspace_rlc_result(self._h, calls_back, C.byref(ok), C.byref(tm) if timings else None))
        return bool(ok.value), tm

    def close(self):
        if getattr(self, "_h", None):
            lib().h2v_workspace_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- primitive probes (GPU parity tests)
class BatchStream:
    """Keeps up to `depth` host-buffer batches in flight on ONE laned workspace (h2v_workspace_create_lanes with `depth`
    lanes, one chunk per batch) through h2v_verify_batch_submit / _wait: push() hands a host batch over and returns the
    (accept bytes, fell_back) of the OLDEST batch once `depth` are in flight, or None; drain() returns the rest, oldest
    first.  The counterpart of h2v::BatchStream.  (Round 2 needed `depth` workspaces for this.)"""

    def __init__(self, plan: "DevicePlan", max_batch: int, depth: int, rlc: bool = False, seed: bytes = None):
        if depth < 1 or depth > 16:
            raise H2VError("BatchStream: depth must be 1 .. 16 (a laned workspace has at most 16 lanes)")
        self.plan, self.rlc, self.seed, self.depth = plan, rlc, seed, depth
        self.ws = Workspace(plan, max_batch, lanes=depth, chunk=max_batch)
        self._n = []          # proofs of the batches in flight, oldest first

    def push(self, host_batch, n: int):
        """host_batch: DevicePlan.host_batch(...)[0]; n: its number of proofs"""
        out = None
        if len(self._n) >= self.depth:     # every lane holds a batch: collect the oldest
            out = self.ws.wait(self._n.pop(0))
        self.plan.submit(host_batch, self.ws, rlc=self.rlc, seed=self.seed)
        self._n.append(n)
        return out

    def drain(self):
        out = []
        while self._n:
            out.append(self.ws.wait(self._n.pop(0)))
        return out

    def close(self):
        if self.ws is not None:
            self.ws.close()
            self.ws = None


def probe_field(op: int, a_list, b_list, device: int = 0):
    nl = 12 if op < 4 else 8
    n = len(a_list)
    nb = nl * 4

    def pack(vals):
        return b"".join(int(v).to_bytes(nb, "little") for v in vals)

    out = C.create_string_buffer(n * nb)
    pa, pb = pack(a_list), pack(b_list)
    check(lib().h2v_probe_field(device, op, n, pa, pb, out))
    return [int.from_bytes(out.raw[nb * i:nb * (i + 1)], "little") for i in range(n)]


def probe_blake2b(msgs, device: int = 0):
    n = len(msgs)
    ln = len(msgs[0])
    assert all(len(m) == ln for m in msgs)
    out = C.create_string_buffer(32 * n)
    check(lib().h2v_probe_blake2b(device, n, ln, b"".join(msgs), out))
    return [out.raw[32 * i:32 * i + 32] for i in range(n)]


def _unxy(b):
    return None if b == bytes(96) else (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))


def probe_g1_decompress(compressed_list, device: int = 0):
    n = len(compressed_list)
    out = C.create_string_buffer(96 * n)
    valid = C.create_string_buffer(n)
    check(lib().h2v_probe_g1_decompress(device, n, b"".join(compressed_list), out, valid))
    return [(valid.raw[i] == 1, _unxy(out.raw[96 * i:96 * i + 96])) for i in range(n)]


def probe_g1_msm(scalar_groups, base_groups, device: int = 0):
    """scalar_groups[i] = list of T ints, base_groups[i] = list of T 48-byte compressed points."""
    n = len(scalar_groups)
    T = len(scalar_groups[0])
    sc = b"".join(int(s).to_bytes(32, "little") for g in scalar_groups for s in g)
    bs = b"".join(b for g in base_groups for b in g)
    out = C.create_string_buffer(96 * n)
    check(lib().h2v_probe_g1_msm(device, n, T, sc, bs, out))
    return [_unxy(out.raw[96 * i:96 * i + 96]) for i in range(n)]


def probe_g1_msm_fixed(plan: DevicePlan, scalar_rows, bases_per_lane: int = 1):
    """sum_t s_t * B_t over the plan's VK-base terms through the all-window tables (k_g1_msm_fixed); scalar_rows[i] = the
    n_fix scalars of proof i (ints < r), in the order of the plan's VK-base terms.  n_fix() = probe_g1_msm_fixed(plan, None)."""
    nf = C.c_uint32()
    if scalar_rows is None:
        lib().h2v_probe_g1_msm_fixed(plan.handle, 0, 1, None, None, C.byref(nf))
        return nf.value
    n = len(scalar_rows)
    sc = b"".join(int(s).to_bytes(32, "little") for row in scalar_rows for s in row)
    out = C.create_string_buffer(96 * n)
    check(lib().h2v_probe_g1_msm_fixed(plan.handle, n, bases_per_lane, sc, out, C.byref(nf)))
    assert all(len(row) == nf.value for row in scalar_rows)
    return [_unxy(out.raw[96 * i:96 * i + 96]) for i in range(n)]


def probe_quad_madd(p_xy, q_xy, neg: bool, device: int = 0):
    """2P + (-)Q as Jacobian (X, Y, Z) integers: [one-lane result, quad lane 0, 1, 2, 3] (h2v_probe_quad_madd)."""
    def limbs(v):
        return [(v >> (32 * i)) & 0xffffffff for i in range(12)]
    pq = (C.c_uint32 * 48)(*(limbs(p_xy[0]) + limbs(p_xy[1]) + limbs(q_xy[0]) + limbs(q_xy[1])))
    out = (C.c_uint32 * 210)()
    check(lib().h2v_probe_quad_madd(device, pq, 1 if neg else 0, out))
    from . import bls12_381 as bls
    rinv = pow(1 << 392, -1, bls.P)

    def val(o):
        return sum(out[o + k] << (28 * k) for k in range(14)) * rinv % bls.P
    return [(val(42 * j), val(42 * j + 14), val(42 * j + 28)) for j in range(5)]


def probe_g1_msm_pippenger(scalars, bases_compressed, device: int = 0):
    """sum_n s_n * B_n through the bucket MSM (scalars: ints < r; bases: 48-byte compressed)."""
    n = len(scalars)
    sc = b"".join(int(s).to_bytes(32, "little") for s in scalars)
    out = C.create_string_buffer(96)
    check(lib().h2v_probe_g1_msm_pippenger(device, n, sc, b"".join(bases_compressed), out))
    return _unxy(out.raw)


def probe_pairing(plan: DevicePlan, p1_list, p2_list):
    n = len(p1_list)
    out = C.create_string_buffer(n)
    check(lib().h2v_probe_pairing(plan.handle, n, b"".join(p1_list), b"".join(p2_list), out))
    return [out.raw[i] for i in range(n)]


def probe_pairing_ex(plan: DevicePlan, p1_list, p2_list, impl: int, dump: bool = True):
    """returns (accept list, per proof [f_miller (12 ints), f_final (12 ints)]) with the chosen kernel"""
    n = len(p1_list)
    out = C.create_string_buffer(n)
    dbg = C.create_string_buffer(n * 24 * 48) if dump else None
    check(lib().h2v_probe_pairing_ex(plan.handle, n, b"".join(p1_list), b"".join(p2_list), out, impl, dbg))
    vals = []
    if dump:
        for i in range(n):
            row = [int.from_bytes(dbg.raw[(i * 24 + q) * 48:(i * 24 + q + 1) * 48], "little") for q in range(24)]
            vals.append((row[:12], row[12:]))
    return [out.raw[i] for i in range(n)], vals
