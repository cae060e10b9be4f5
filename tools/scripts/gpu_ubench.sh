#!/bin/bash
# the round's micro-benchmarks on the GPU box: gpurun_out/ubench/{imad,dfma}.txt (copy to profiles/rNN_*_ubench.txt)
O=gpurun_out/ubench; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/ubench/imad.hip -o /tmp/imad > /dev/null 2>&1 && timeout -k 10 400 /tmp/imad > $O/imad.txt 2>&1; echo "imad rc=$?"
hipcc --offload-arch=gfx950 -O3 tools/ubench/dfma_mont.hip -o /tmp/dfma > /dev/null 2>&1 && timeout -k 10 300 /tmp/dfma > $O/dfma.txt 2>&1; echo "dfma rc=$?"
cat $O/dfma.txt
grep -E "^---|mad_u64_u32|mad64|add_u32" $O/imad.txt | cut -c1-330
