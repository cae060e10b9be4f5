"""include/h2v.hpp (the C++ mirror of the reference's prepare / verify interface) compiled with g++ and driven by
tests/cpp/h2v_cpp_driver.cpp.  CPU leg: it builds, links against the C-ABI library and fails loudly with
H2V_E_DEVICE on a box without a GPU (no CPU fallback).  GPU leg: per-proof prepare()/verify() and verify_batch() give
the construction's verdicts."""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "plutus_halo2_verifier_gen_amd")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    import __graft_entry__ as ge
    ge.build_hip()
    out = str(tmp_path_factory.mktemp("cpp") / "h2v_cpp_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "h2v_cpp_driver.cpp"), "-o", out,
                           "-L", PKG, "-lh2v_hip", "-pthread", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def _write_case(tmp_path, n=6, corrupt=(1, 4)):
    from plutus_halo2_verifier_gen_amd import plan as PL, synth, vk as V
    vk, td = V.simple_mul_vk()
    pl = PL.compile_plan(vk)
    b = synth.forge_batch(vk, td, n, seed=61, plan=pl, workers=1)
    proofs = [bytearray(b.proof(i)) for i in range(n)]
    for i in corrupt:
        proofs[i] = bytearray(synth.corrupt(pl, bytes(proofs[i]), b.instances[96 * i:96 * i + 96], "flip_last_scalar", None)[0])
    blob = struct.pack("<III", n, vk.n_public_inputs, 0)
    for i in range(n):
        blob += struct.pack("<I", len(proofs[i])) + bytes(proofs[i]) + b.instances[96 * i:96 * i + 96]
    (tmp_path / "plan.bin").write_bytes(pl.to_bytes())
    (tmp_path / "vk.json").write_text(vk.to_json())
    (tmp_path / "batch.bin").write_bytes(blob)
    return str(tmp_path / "plan.bin"), str(tmp_path / "batch.bin"), "".join("0" if i in corrupt else "1" for i in range(n))


def _gpu_present():
    try:
        from plutus_halo2_verifier_gen_amd import backend
        return backend.device_count() >= 1
    except Exception:
        return False


def test_cpp_header_builds_and_fails_loudly_without_gpu(driver, tmp_path):
    if _gpu_present():
        pytest.skip("a GPU is present: covered by the gpu leg")
    plan, batch, _ = _write_case(tmp_path, n=2, corrupt=())
    r = subprocess.run([driver, plan, batch], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, r.stdout + r.stderr
    assert r.stdout.startswith("error -3 ")          # H2V_E_DEVICE: no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["plan.bin", "vk.json"])
def test_cpp_prepare_verify_on_gpu(driver, tmp_path, source):
    """source = vk.json: the driver goes JSON -> h2v_plan_compile -> h2v_plan_load -> verify with no Python in the loop"""
    plan, batch, want = _write_case(tmp_path)
    if source == "vk.json":
        plan = os.path.join(os.path.dirname(plan), "vk.json")
    r = subprocess.run([driver, plan, batch], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())
    if source == "vk.json":
        assert int(lines["compiled_plan_bytes"]) == os.path.getsize(os.path.join(os.path.dirname(plan), "plan.bin"))
    assert lines["single"] == want and lines["batch"] == want
    assert lines["rlc"] == want and lines["stream0"] == want and lines["stream1"] == want
    assert lines["rlc_fell_back"] == "1"          # the corrupted proofs are caught only by the pairing
    assert lines["misuse_refused"] == "1"
    assert lines["batch_stream_per_proof"] == "1" and lines["batch_stream_rlc"] == "1"   # h2v::BatchStream, depth 3, seven batches
    assert lines["node_stream"] == "1"     # h2v::NodeStream over the device list [0, 0] (per proof) and [0, 0, 0] (RLC)
    assert lines["laned"] == "1"           # a laned workspace through the C++ wrapper
    assert lines["multi"] == "1"           # one laned workspace for two keys (h2v_workspace_create_multi), submit(vk, batch)
