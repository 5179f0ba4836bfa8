#!/bin/bash
# Round-4 third measurement set: FP16_OVFL probe, lab ablations of the K loop (no fragment reads / no DMA), GELU A/B, ledger, tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -o /tmp/fp16_ovfl_probe tools/fp16_ovfl_probe.hip 2>/dev/null && /tmp/fp16_ovfl_probe > $O/fp16_ovfl.txt 2>&1; cat $O/fp16_ovfl.txt
SHAPES=qkv:20800:2304:768,sq4096:4096:4096:4096 VARIANTS=1000,1004,3048,3052,1001 REPS=3 python tools/g8_lab.py 2>/dev/null > $O/lab_kloop.txt; cat $O/lab_kloop.txt
SHAPES=fc1,qkv ROUNDS=12 python tools/g8_lib_ab.py tools/_lab_old.so edgecape_amd/libedgecape_hip_lab.so 2>/dev/null > $O/gelu_ab.txt; cat $O/gelu_ab.txt
python tools/g8_ledger.py qkv fc1 fc2 sq4096 2>/dev/null > $O/ledger.txt; grep -v "XCD" $O/ledger.txt | head -60
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -n 15 $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err; python tools/bench_line.py < $O/bench.json | cut -c1-300
