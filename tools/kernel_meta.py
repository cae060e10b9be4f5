#!/usr/bin/env python3
"""Prints the register / scratch figures of every kernel in the shipped library's gfx950 code object (the numbers DESIGN.md
quotes): extracts the device ELF from libh2v_hip.so's fat binary and reads the .note metadata with llvm-readelf.
usage: kernel_meta.py [path to libh2v_hip.so]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "plutus_halo2_verifier_gen_amd", "libh2v_hip.so")
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(td, "dev.elf")])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", os.path.join(td, "dev.elf")], text=True)
rows, cur = [], {}
for line in notes.splitlines():
    m = re.match(r"(\s*-?\s*)\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|args):\s*(\S*)", line)
    if not m:
        continue
    lead, k, v = m.groups()
    if "-" in lead and k in ("agpr_count", "args"):     # (the first key of a kernel's map)
        if cur.get("name", "").startswith("k_"):
            rows.append(cur)
        cur = {}
    if k != "args":
        cur.setdefault(k, v)
if cur.get("name", "").startswith("k_"):
    rows.append(cur)
seen = set()
print("%-34s %5s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
for r in sorted(rows, key=lambda r: r["name"]):
    if r["name"] in seen or "vgpr_count" not in r:
        continue
    seen.add(r["name"])
    print("%-34s %5s %5s %5s %7s %7s %8s %7s" % (r["name"], r.get("vgpr_count"), r.get("agpr_count", "0"), r.get("sgpr_count"), r.get("vgpr_spill_count", "0"),
                                                  r.get("sgpr_spill_count", "0"), r.get("private_segment_fixed_size", "0"), r.get("group_segment_fixed_size", "0")))
